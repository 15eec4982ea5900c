"""Physical objects crossed by the beam (mirror of CodePython/Sample.py:22-351).

`AnalyticalSample.setWave` / `setWaveRT` keep the reference signatures and run as HIP kernels on thickness maps held in
HBM as float32.  The phase k*delta*T reaches 1e2-1e3 rad, which float32 cannot carry to 1e-5, so
  * setWave forms the phase in float64 on the device, range-reduces, then takes a float32 sincos;
  * setWaveRT returns the phase as a float64 tensor;
  * the Experiment chains never materialise the phase at all: they hand the thickness maps and the per-material
    coefficients (`stack_wave` / `stack_rt`) to the consuming kernel, which fuses the transmission into its load.
"""
import numpy as np
import torch

from . import _xml, geometry, materials, ops
from ._lib import PsxError
from ._tensors import device, is_scalar, to_dev
from .getk import k_sample


class Sample:
    def __init__(self, xml_directory=None):
        self.xmlSampleFileName = "xmlFiles/Samples.xml"
        self._xml_directory = xml_directory
        self.myName = ""
        self.myType = ""
        self.myMaterials = []
        self.myGeometry = []
        self.geom_parameters = None
        self.myGeometryFunction = ""

    def defineCorrectValuesSample(self):
        """Sample.py:33-77: XML -> fields (optional tags depend on the geometry function)."""
        doc = _xml.parse(self._xml_directory or _xml.xml_dir(), "Samples.xml")
        node = _xml.find_named(doc, "sample", self.myName)
        if node is None:
            print(self.myName)
            raise ValueError("Sample not found in the xml file")
        self._node = node
        self.myType = _xml.child_text(node, "myType")
        self.myMaterials = list(_xml.child_text(node, "myMaterials").split(","))
        fn = self.myGeometryFunction = _xml.child_text(node, "myGeometryFunction")
        if fn == "getMembraneFromFile":
            self.myPMMAThickness = float(_xml.child_text(node, "myPMMAThickness"))
            if _xml.has_child(node, "myMembraneFile"):   # the reference never reads it (SAM:53-54 vs SAM:231)
                self.myMembraneFile = _xml.child_text(node, "myMembraneFile")
        if fn == "getMembraneSegmentedFromFile":
            self.myMeanSphereRadius = float(_xml.child_text(node, "myMeanSphereRadius"))
            self.myNbOfLayers = int(_xml.child_text(node, "myNbOfLayers"))
            self.myPMMAThickness = float(_xml.child_text(node, "myPMMAThickness"))
        if fn == "get_my_thickness" and self.myName != "air_volume":
            self.myThickness = float(_xml.child_text(node, "myThickness"))
        if fn == "getSampleFromFile":
            self.mySampleFile = _xml.child_text(node, "mySampleFile")
        if fn == "loadSampleGeometryFromImages":
            self.myGeometryFolder = _xml.child_text(node, "myGeometryFolder")
        for tag in ("myRadius", "myOrientation"):
            if _xml.has_child(node, tag):
                setattr(self, tag, float(_xml.child_text(node, tag)))

    def getDeltaBeta(self, sourceSpectrum):
        """Sample.py:83-152: per material, a list of (energy, value) for every energy of the spectrum."""
        self.materialProvenance = {}
        for material in self.myMaterials:
            if materials.has_table(material):            # TablesDeltaBeta branch, Sample.py:112-148
                delta, beta = materials.table_walk(material, sourceSpectrum)
                self.delta.append(delta)
                self.beta.append(beta)
                self.materialProvenance[material] = materials.provenance(material)
                continue
            db = [materials.delta_beta(material, e) for e, _ in sourceSpectrum]
            self.delta.append([(e, d) for (e, _), (d, _) in zip(sourceSpectrum, db)])
            self.beta.append([(e, b) for (e, _), (_, b) in zip(sourceSpectrum, db)])
            self.materialProvenance[material] = materials.provenance(material)
        if len(self.delta) != len(self.myMaterials):
            raise ValueError("One or more materials have not been found in delta beta tables")


class AnalyticalSample(Sample):
    def __init__(self, xml_directory=None):
        Sample.__init__(self, xml_directory)
        self.delta = []
        self.beta = []
        self._dev_geometry = None
        self._dev_src = None

    # ------------------------------------------------------------------------------------------ geometry
    def getMyGeometry(self, studyDimensions, studyPixelSize, oversamp, pointNum=0, number_of_positions=0):
        """Sample.py:163-245: thickness maps [material, x, y] in metres."""
        dimX, dimY = int(studyDimensions[0]), int(studyDimensions[1])
        fn = self.myGeometryFunction
        if self.myType == "sample_of_interest":
            if fn == "getSampleFromFile":
                self.myGeometry = np.load(self.mySampleFile)
                return
            if fn == "CreateSampleCylindre":
                self.myGeometry, self.geom_parameters = geometry.cylinder(dimX, dimY, studyPixelSize, self.myRadius,
                                                                          self.myOrientation)
                return
            if fn == "CreateSampleSphere":
                self.myGeometry, self.geom_parameters = geometry.sphere(dimX, dimY, studyPixelSize, self.myRadius)
                return
            if fn == "CreateYourSampleGeometry":                                # SAM:196-203
                self.myGeometry, self.geom_parameters = geometry.your_sample_geometry(dimX, dimY)
                return
            if fn == "CreateSampleSpheresInCylinder":                           # SAM:204-206
                self.myGeometry, self.geom_parameters = geometry.spheres_in_cylinder(dimX, dimY, studyPixelSize)
                return
            if fn == "CreateSampleSpheresInParallelepiped":                     # SAM:207-209
                self.myGeometry, self.geom_parameters = geometry.spheres_in_parallelepiped(dimX, dimY, studyPixelSize)
                return
            if fn == "loadSampleGeometryFromImages":                            # SAM:222-225
                geom, self.geom_parameters = geometry.load_sample_geometry_from_images(self.myGeometryFolder, dimX, dimY,
                                                                                       studyPixelSize)
                self.myGeometry = np.array(geom)
                return
        if self.myType == "membrane":
            if fn == "getMembraneFromFile":                                   # SAM:230-233
                from .Samples.getMembraneFromFile import getMembraneFromFile
                geom, self.geom_parameters = getMembraneFromFile(self.myMembraneFile, studyDimensions, pointNum,
                                                                 self.myPMMAThickness)
                self.myGeometry = np.asarray(geom)
                return
            if fn == "getMembraneSegmentedFromFile":                          # SAM:234-237, sphere splat on the GPU
                from .Samples.getMembraneFromFile import getMembraneSegmentedFromFile
                geom, self.geom_parameters = getMembraneSegmentedFromFile(self, dimX, dimY, studyPixelSize, pointNum,
                                                                          self.myPMMAThickness, stacked=True)
                self.myGeometry = geom[2]                      # [2, dimX, dimY]: membrane, support (a new tensor per position)
                return
        if fn == "get_my_thickness":
            self.myGeometry = np.full((1, dimX, dimY), self.myThickness * 1e-6, dtype=np.float32)   # SAM:239-243
            return
        raise ValueError("Could not define sample geometry")

    def geometry_dev(self):
        """The thickness stack as a float32 tensor in HBM (uploaded once per myGeometry object)."""
        g = self.myGeometry
        if self._dev_geometry is None or self._dev_src is not g:
            if isinstance(g, (list, tuple)):
                g = np.asarray(g)
            ndim = g.dim() if isinstance(g, torch.Tensor) else np.ndim(g)
            if ndim != 3:
                raise Exception("Sample Geometry has the wrong nb of dim [material, x, y]")   # SAM:263-264
            self._dev_geometry = to_dev(g, torch.float32)
            self._dev_src = self.myGeometry
        return self._dev_geometry

    # --------------------------------------------------------------------------------------- coefficients
    def _coeff(self, table, energy):
        """The reference looks delta/beta up by exact float equality on the energy and keeps 0 when nothing matches
        (Sample.py:266-277; the last matching row wins).  The rows of a material are indexed once (a 25-energy position
        otherwise walks every table for every energy: thousands of comparisons per position); the index follows the list it
        was built from (same object, same length)."""
        out = np.zeros(len(self.myMaterials))
        cache = self.__dict__.setdefault("_coeff_index", {})
        for imat in range(len(self.myMaterials)):
            rows = table[imat]
            hit = cache.get(id(rows))
            if hit is None or hit[0] is not rows or hit[1] != len(rows):
                hit = cache[id(rows)] = (rows, len(rows), {e: v for e, v in rows})
            out[imat] = hit[2].get(energy, 0.0)
        return out

    def stack_wave(self, energy, phase=True, att=True):
        """MaterialStack for a complex wave: cphase = -k delta, catt = -k beta (Sample.py:279)."""
        k = k_sample(energy)
        d, b = self._coeff(self.delta, energy), self._coeff(self.beta, energy)
        return ops.MaterialStack(self.geometry_dev(), cphase=(-k * d if phase else 0 * d), catt=(-k * b if att else 0 * b))

    def _df_model(self, imat):
        """(alveoli radius um, sphere volume fraction) when material `imat` scatters (Sample.py:322-344), else None.
        'cylinder_beeds' wins over "Lung" because the reference tests it second and overwrites."""
        if self.myType != "sample_of_interest":
            return None
        if self.myName == 'cylinder_beeds':
            return 15, 0.6
        if self.myMaterials[imat] == "Lung":
            return 47, 0.5
        return None

    def has_dark_field(self):
        return any(self._df_model(i) is not None for i in range(len(self.myMaterials)))

    def stack_rt(self, energy, phase=True, att=True):
        """MaterialStack for intensity + phase: cphase = -k delta, catt = -2 k beta (Sample.py:347-348).  Scattering
        materials enter with their thickness scaled by the sphere volume fraction (Sample.py:332,343), which is the same
        as scaling their two coefficients."""
        k = k_sample(energy)
        d, b = self._coeff(self.delta, energy), self._coeff(self.beta, energy)
        frac = np.array([1.0 if self._df_model(i) is None else self._df_model(i)[1] for i in range(len(self.myMaterials))])
        return ops.MaterialStack(self.geometry_dev(), cphase=(-k * d * frac if phase else 0 * d),
                                 catt=(-2 * k * b * frac if att else 0 * b))

    DF_CACHE_BYTES = 16 << 30  # width maps kept per sample: one float64 map per energy of a spectrum, least recently used out first

    def dark_field(self, energy):
        """newDf of setWaveRT (Sample.py:322-344): 2 delta sqrt(N_s) sqrt(ln(2/delta)+1) of the LAST scattering
        material (the reference overwrites, it does not accumulate); int 0 when nothing scatters."""
        # cache entry = (geometry object, what the map was computed from, map, exact maximum); validated by IDENTITY of the
        # geometry (an id() alone can be reused by a new object) and by value of everything else that enters the formula
        d = self._coeff(self.delta, energy)
        sig = (tuple(self.myMaterials), self.myName, self.myType, tuple(float(v) for v in d))
        cache = self.__dict__.setdefault("_df_cache", {})
        hit = cache.get(energy)
        if hit is not None and hit[0] is self.myGeometry and hit[1] == sig:
            cache[energy] = cache.pop(energy)         # most recently used last: a spectrum walked cyclically keeps its maps
            return hit[2]
        newDf = 0
        for imat in range(len(self.myMaterials)):
            model = self._df_model(imat)
            if model is None:
                continue
            radius, fraction = model
            NsphereVol = fraction * 3 / 4 / np.pi / (radius ** 3)
            geom = self.geometry_dev()[imat].to(torch.float64) * 1e6
            newDf = (2 * d[imat] * np.sqrt(np.log(2 / d[imat]) + 1)) * torch.sqrt(NsphereVol ** (1 / 3) * geom)
        if isinstance(newDf, torch.Tensor):
            # the map depends on the (static) thickness maps and the energy only: kept, with its maximum -- fastRefractionDF
            # sizes the displacement maps it returns by it (RF2:117) and would otherwise read it back on every call.
            # Bounded by BYTES (25 energies at 4096^2 = 3.4 GB; at least 4 maps, at most 256): the least recently used go.  (Round 4
            # evicted oldest-first at 32 entries: a spectrum of more than 32 energies then recomputed every map on every
            # position -- ADVICE r4.)
            cache.pop(energy, None)
            limit = max(4, min(256, self.DF_CACHE_BYTES // max(1, newDf.numel() * 8)))
            while len(cache) >= limit:
                cache.pop(next(iter(cache)))
            cache[energy] = (self.myGeometry, sig, newDf, float(newDf.max().item()))
        return newDf

    def dark_field_max(self, darkField):
        """The exact maximum (rad) of a map dark_field() returned, or None for any other array."""
        for _, _, t, mx in getattr(self, "_df_cache", {}).values():
            if t is darkField:
                return mx
        return None

    # ------------------------------------------------------------------------------------- reference API
    def setWave(self, incidentWave, energy):
        """Sample.py:248-282: disturbedWave = prod_m exp((-i k delta_m - k beta_m) T_m) * incidentWave."""
        stack = self.stack_wave(energy)
        wave = to_dev(incidentWave, torch.complex64)
        return ops.transmit_wave(wave, 1.0, stack)

    def setWaveRT(self, incidentIntensity, energy, incidentphi=0, incidentDf=0):
        """Sample.py:285-351: (I*exp(-2 k beta T), phi - k delta T, newDf); phi is float64; newDf is int 0
        unless a material scatters (then a float64 tensor of angles in rad)."""
        stack = self.stack_rt(energy)
        I = to_dev(incidentIntensity, torch.float32)
        phi_in = None if is_scalar(incidentphi) and incidentphi == 0 else (
            torch.full(I.shape, float(incidentphi), dtype=torch.float64, device=I.device) if is_scalar(incidentphi)
            else to_dev(incidentphi, torch.float64))
        I_out, phi_out = ops.transmit_rt(I, 1.0, stack, phi_in)
        return I_out, phi_out, self.dark_field(energy)
