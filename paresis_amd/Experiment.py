"""Experiment orchestration on the MI355X (mirror of CodePython/Experiment.py:25-607).

Same class surface as the reference: Experiment(exp_dict) reads the four XML files, owns mySource / myDetector /
myMembrane / mySampleofInterest / myAirVolume / myPlate, and exposes computeSampleAndReferenceImages_Fresnel,
computeSampleAndReferenceImages_RT, wavePropagation, refraction and saveAllParameters.

What is different under the hood (MI355X-first, not a translation):
  * every array lives in HBM (float32 / complex64 torch tensors); numpy never touches the hot loop;
  * object transmissions are not separate passes: the thickness maps and their per-energy coefficients are handed to the
    Fresnel / refraction kernels, which evaluate exp((-ik delta - k beta) T) while loading (K1/K2 fused);
  * the two propagations that share the membrane exit wave (Experiment.py:341 and :349) are one call (shared forward
    transform); |.|^2, plate attenuation and the sum over energies happen in the kernels' store phase;
  * `Experiment.from_objects` builds an experiment from injected objects (no XML), which is how the parity tests feed
    the same inputs to this class and to the oracle.
"""
import time

import numpy as np
import torch

from . import _lib, _xml, ops
from ._lib import PsxError
from ._tensors import device, is_scalar, to_dev
from .Detector import Detector
from .getk import getk, k_refraction
from .Sample import AnalyticalSample
from .Source import Source


def _rebased(base, rows):
    """ops.MaterialBatch of cached coefficient rows (cphase rows, catt rows) over this position's maps."""
    return ops.MaterialBatch(base, rows[0], rows[1])


class Experiment:
    def __init__(self, exp_dict):
        """Experiment.py:26-137: XML -> objects -> geometry; exp_dict carries experimentName, overSampling,
        nbExpPoints, simulation_type, filepath (and optionally xmlDir, noise, seed, fresnelEngine)."""
        self.xmlExperimentFileName = "xmlFiles/Experiment.xml"
        self._xml_directory = _xml.xml_dir(exp_dict)
        self.name = exp_dict['experimentName']
        self.exp_dict = exp_dict
        ed = self.exp_dict
        ed['studyPixelSize'] = 0.
        ed['studyDimensions'] = (0., 0.)
        ed['inVacuum'] = False
        ed['meanShotCount'] = 0
        ed['meanEnergy'] = 0
        ed['distSourceToMembrane'] = 0
        ed['distMembraneToObject'] = 0
        ed['distObjectToDetector'] = 0
        ed['studyPixelSize_unit'] = "um"
        ed['studyDimensions_unit'] = "pixels"
        ed['meanEnergy_unit'] = "keV"
        ed['distSourceToMembrane_unit'] = "m"
        ed['distMembraneToObject_unit'] = "m"
        ed['distObjectToDetector_unit'] = "m"
        self._init_state()
        if exp_dict.get('allowSyntheticMaterials'):
            from . import materials
            materials.allow_synthetic(True)

        self.defineCorrectValues(exp_dict)
        self.myDetector.defineCorrectValuesDetector()
        self.mySource.defineCorrectValuesSource()
        self.mySampleofInterest.defineCorrectValuesSample()
        self.myAirVolume.defineCorrectValuesSample()
        self.myAirVolume.myThickness = (ed['distSourceToMembrane'] + ed['distObjectToDetector'] + ed['distMembraneToObject']) * 1e6
        if self.myPlate is not None:
            self.myPlate.defineCorrectValuesSample()
        self.myMembrane.defineCorrectValuesSample()

        ed['magnification'] = (ed['distSourceToMembrane'] + ed['distObjectToDetector'] + ed['distMembraneToObject']) / (ed['distSourceToMembrane'] + ed['distMembraneToObject'])   # EXP:81
        self.getStudyDimensions()
        self.mySource.setMySpectrum(self.myDetector.det_param["photonCounting"])
        dims, pix, ov = ed['studyDimensions'], ed['studyPixelSize'], ed['overSampling']
        self.myAirVolume.getDeltaBeta(self.mySource.mySpectrum)
        self.myAirVolume.getMyGeometry(dims, pix, ov)
        if self.myPlate is not None:
            self.myPlate.getDeltaBeta(self.mySource.mySpectrum)
            self.myPlate.getMyGeometry(dims, pix, ov)
        self.mySampleofInterest.getDeltaBeta(self.mySource.mySpectrum)
        self.mySampleofInterest.getMyGeometry(dims, pix, ov)
        self.myMembrane.getDeltaBeta(self.mySource.mySpectrum)
        self.myMembrane.membranePixelSize = pix * ed['distSourceToMembrane'] / (ed['distSourceToMembrane'] + ed['distMembraneToObject'])   # EXP:96
        if self.myDetector.det_param['myScintillatorMaterial'] is not None:
            self.myDetector.getBeta(self.mySource.mySpectrum)
            self.myDetector.getSpectralEfficiency()
        if ed['simulation_type'] == "RayT" and ed["overSampling"] < 2:      # EXP:103-105
            print(f'/!\\/!\\ OVERSAMPLING FACTOR < MIN OVERSAMPLING FOR RAY-T MODEL: {ed["overSampling"]} < 2')
        if ed['simulation_type'] == "Fresnel":                              # EXP:106-110 (advisory print, host scalars)
            from .usefullScripts.getSamplingFactor import is_overSampling_ok
            kind = self.mySource.source_dict["myType"]
            if kind == "Polychromatic":                                     # judged at half the top energy of the spectrum
                is_overSampling_ok(ed, self.myDetector.det_param['myPixelSize'], self.mySource.mySpectrum[-1][0] / 2)
            elif kind == "Monochromatic":
                is_overSampling_ok(ed, self.myDetector.det_param['myPixelSize'], self.mySource.source_dict["Energy"])
        print('\nCurrent experiment:', self.name)
        print("  Magnification :", ed['magnification'])
        print(f'  Study dimensions: {ed["studyDimensions"]} pixels')
        print("  Sample pixel size =", ed["studyPixelSize"], "um")

    def _init_state(self):
        self.mySampleofInterest = None
        self.mySampleType = ""
        self.myDetector = None
        self.mySource = None
        self.myMembrane = None
        self.myPlate = None
        self.myAirVolume = None
        self.Dxreal = []
        self.Dyreal = []
        self.imageSampleBeforeDetection = []
        self.imageReferenceBeforeDetection = []
        self.imagePropagBeforeDetection = []
        self._fresnel_plan = None
        self._bins_ready = False      # thresholds validated and closed with the last spectrum energy (EXP:296-301)
        self._tables = None           # (state key, per-bin energy tables) of the batched chains
        self._etmp = None             # per-energy intensity scratch [energies of a bin][Nx][Ny] (batched energy chain)
        self._accs = None             # study-grid accumulators [4][Nx][Ny], allocated once
        self._tmp = None
        self._pending_means = []
        self._sums_pool, self._sums_next = None, 0
        self.darkFieldPropag = None
        self._df_tmp = None           # dark-field half of the fused sample hop
        self._zero_stack = None       # Propag / White of every position but 0
        self._halo = None             # refraction gather halo of this experiment (exp_dict['refractionHalo']: 4 | 6 | 8 | 'auto')

    @classmethod
    def from_objects(cls, exp_dict, source, detector, membrane, sample, air=None, plate=None):
        """Build an experiment from ready objects instead of XML (the route the parity tests and bench.py use)."""
        self = object.__new__(cls)
        self.name = exp_dict.get('experimentName', 'injected')
        self.exp_dict = exp_dict
        self._xml_directory = None
        self._init_state()
        self.mySource, self.myDetector = source, detector
        self.myMembrane, self.mySampleofInterest, self.myAirVolume, self.myPlate = membrane, sample, air, plate
        self.mySampleType = "AnalyticalSample"
        ed = exp_dict
        ed.setdefault('meanEnergy', 0)
        ed.setdefault('inVacuum', air is None)
        if 'magnification' not in ed:
            ed['magnification'] = (ed['distSourceToMembrane'] + ed['distObjectToDetector'] + ed['distMembraneToObject']) / (ed['distSourceToMembrane'] + ed['distMembraneToObject'])
        if 'studyDimensions' not in ed or 'studyPixelSize' not in ed:
            self.getStudyDimensions()
        return self

    def defineCorrectValues(self, exp_dict):
        """Experiment.py:140-197."""
        self.mySource = Source(self._xml_directory)
        self.myDetector = Detector(exp_dict, self._xml_directory)
        doc = _xml.parse(self._xml_directory, "Experiment.xml")
        node = _xml.find_named(doc, "experiment", self.name)
        if node is None:
            raise ValueError("experiment not found in xml file")
        ed = self.exp_dict
        ed['distSourceToMembrane'] = float(_xml.child_text(node, "distSourceToMembrane"))
        ed['distMembraneToObject'] = float(_xml.child_text(node, "distMembraneToObject"))
        ed['distObjectToDetector'] = float(_xml.child_text(node, "distObjectToDetector"))
        ed['meanShotCount'] = float(_xml.child_text(node, "meanShotCount"))
        if _xml.has_child(node, "inVacuum"):
            ed['inVacuum'] = _xml.child_text(node, "inVacuum") == "True"
        if _xml.has_child(node, "plateName"):
            self.myPlate = AnalyticalSample(self._xml_directory)
            self.myPlate.myName = _xml.child_text(node, "plateName")
        self.myAirVolume = AnalyticalSample(self._xml_directory)
        self.myAirVolume.myName = "air_volume"
        self.mySampleType = _xml.child_text(node, "sampleType")
        if self.mySampleType != "AnalyticalSample":
            raise Exception("sample type not defined")
        self.mySampleofInterest = AnalyticalSample(self._xml_directory)
        self.myMembrane = AnalyticalSample(self._xml_directory)
        self.myMembrane.myName = _xml.child_text(node, "membraneName")
        self.mySampleofInterest.myName = _xml.child_text(node, "sampleName")
        self.myDetector.myName = _xml.child_text(node, "detectorName")
        self.mySource.myName = _xml.child_text(node, "sourceName")

    def getText(self, node):
        return _xml.text(node)

    def getStudyDimensions(self):
        """Experiment.py:204-216."""
        ed, dp = self.exp_dict, self.myDetector.det_param
        self.precision = (dp["myPixelSize"] / ed['overSampling'] / ed['distObjectToDetector'])
        dims = np.asarray(dp["myDimensions"]) * int(ed['overSampling'])
        ed['studyDimensions'] = [int(dims[0]), int(dims[1])]
        ed['studyPixelSize'] = dp["myPixelSize"] / ed['overSampling'] / ed['magnification']

    # ---------------------------------------------------------------------------------------- propagators
    def _plan(self):
        Nx, Ny = (int(v) for v in self.exp_dict['studyDimensions'])
        p = self._fresnel_plan
        if p is None or (p.Nx, p.Ny) != (Nx, Ny):
            if p is not None:
                p.close()
            p = self._fresnel_plan = ops.FresnelPlan(Nx, Ny, max_dist=2, engine=int(self.exp_dict.get('fresnelEngine', 0)))
        return p

    def _fresnel_scalars(self, z, Energy, magnification):
        """a = z/(2kM), global phase kz/M (EXP:250) and the frequency step from the UN-padded grid (EXP:246-247)."""
        k = getk(Energy * 1000)
        Nx, Ny = self.exp_dict['studyDimensions']
        h = self.exp_dict['studyPixelSize'] * 1e-6
        return z / (2 * k * magnification), k * z / magnification, (2 * np.pi / (Nx * h), 2 * np.pi / (Ny * h))

    def wavePropagation(self, waveToPropagate, propagationDistance, Energy, magnification):
        """Experiment.py:219-252: Fresnel propagation of a complex wave over `propagationDistance` metres."""
        if propagationDistance == 0:
            return waveToPropagate                                                     # EXP:233-234
        a, g, du = self._fresnel_scalars(propagationDistance, Energy, magnification)
        return self._plan().propagate([a], [g], du, wave_in=to_dev(waveToPropagate, torch.complex64))[0]

    def refraction(self, intensityRefracted, phi, propagationDistance, Energy, magnification, darkField=0, _mutate=True,
                   _want_D=True):
        """Experiment.py:255-277.  (_mutate=False: the chain's own calls, whose input is a temporary, skip the in-place zeroing
        of clamped rays in the dark-field variant.)"""
        from .refractionFileNumba2 import fastRefraction, fastRefractionDF
        with ops.deterministic(self._reproducible(), scale=None):     # the caller's unit: the chain's own scope, or the user's setting
            if type(darkField) == int or type(darkField) == float:
                return fastRefraction(intensityRefracted, phi, propagationDistance, Energy, magnification,
                                      self.exp_dict["studyPixelSize"])
            known = self.mySampleofInterest.dark_field_max(darkField) if hasattr(self.mySampleofInterest, "dark_field_max") else None
            return fastRefractionDF(intensityRefracted, phi, propagationDistance, Energy, magnification,
                                    self.exp_dict["studyPixelSize"], darkField, darkFieldMax=known,
                                    check=not self.exp_dict.get('deferStatus'), mutate=_mutate, want_D=_want_D)

    def _reproducible(self):
        """exp_dict['reproducible'] (default True): the far rays of every refraction of this experiment are summed in fixed
        point (psx_set_deterministic), so an image is the same bits on every run and on any number of GPUs -- the reference's
        scatter is one raster-order loop (RF2:217-263) and has that property by construction.  False: float atomics in arrival
        order (last-bit differences that can flip a Poisson draw), a few per cent faster."""
        return bool(self.exp_dict.get('reproducible', True))

    def computeSampleAndReferenceImages(self, pointNum):
        """Dispatcher on exp_dict['simulation_type'] (main.py:68-73)."""
        if self.exp_dict['simulation_type'] == "Fresnel":
            return self.computeSampleAndReferenceImages_Fresnel(pointNum)
        if self.exp_dict['simulation_type'] == "RayT":
            return self.computeSampleAndReferenceImages_RT(pointNum)
        raise Exception("simulation Type not defined: ", self.exp_dict['simulation_type'])

    # ------------------------------------------------------------------------------------------- helpers
    def _begin(self, pointNum):
        """Bin thresholds (EXP:296-305), output stacks, accumulators.

        The reference validates the thresholds and appends the last spectrum energy when it computes position 0, which a
        serial run always does first.  Here a rank of a sharded run may never see position 0 (dist.my_positions), so it is
        done once per Experiment, whichever position comes first."""
        dp = self.myDetector.det_param
        nbins = self._close_bins()
        dev = device()
        n0, n1 = int(dp['myDimensions'][0]), int(dp['myDimensions'][1])
        # one allocation for the stacks; every slot of Sample/Reference is written by its bin's detection, Propag only at
        # position 0 and White is zero elsewhere (the reference detects an all-zero white there: Poisson(0) = 0).  Away from
        # position 0 the reference returns fresh zero arrays (EXP:363-375, 488-498), and so does this class: two zeroed stacks of
        # the position's own (a caller may write into them: `White[White == 0] = 1`).  A caller that only READS what it is
        # handed -- main.run, which saves or packs every stack, the bench's position loop -- sets exp_dict['sharedZeroStacks']:
        # Propag and White of every position but 0 are then ONE zero stack per experiment, not 33 MB filled per position (10 us
        # at 2048^2 detectors, the kernel trace of gpurun_out/r5s36) and half the memory kept per position.
        if pointNum != 0 and self.exp_dict.get('sharedZeroStacks', False):
            if self._zero_stack is None or tuple(self._zero_stack.shape) != (nbins, n0, n1) or self._zero_stack.device != dev:
                self._zero_stack = ops.fill(torch.empty((nbins, n0, n1), dtype=torch.float32, device=dev), 0.0)
            two = torch.empty((2, nbins, n0, n1), dtype=torch.float32, device=dev)
            out = [two[0], two[1], self._zero_stack, self._zero_stack]
        elif pointNum != 0:
            four = torch.empty((4, nbins, n0, n1), dtype=torch.float32, device=dev)
            ops.fill(four[2:], 0.0)
            out = [four[0], four[1], four[2], four[3]]
        else:
            four = torch.empty((4, nbins, n0, n1), dtype=torch.float32, device=dev)
            out = [four[0], four[1], four[2], four[3]]
        N = tuple(int(v) for v in self.exp_dict['studyDimensions'])
        # the four study-grid accumulators live as long as the experiment; the first energy of a bin STORES into them, so they
        # are never cleared (the reference re-allocates zeros after every bin, EXP:396-399)
        if self._accs is None or tuple(self._accs.shape[1:]) != N or self._accs.device != dev:
            self._accs = torch.empty((4,) + N, dtype=torch.float32, device=dev)
            self._tmp = torch.empty((3,) + N, dtype=torch.float32, device=dev)
        # [sum I_ref, sum E * I_ref] over the energies (EXP:360-361): one zeroed slot per position from a pool (a torch.zeros
        # per position would be a PyTorch fill kernel per position)
        if self._sums_pool is None or self._sums_next >= self._sums_pool.shape[0]:
            self._sums_pool = torch.zeros((256,) + tuple(ops.new_sums(dev).shape), dtype=torch.float64, device=dev)
            self._sums_next = 0
        sums = self._sums_pool[self._sums_next]
        self._sums_next += 1
        return out, [self._accs[0], self._accs[1], self._accs[2], self._accs[3]], N, dev, sums

    def reserve_outputs(self, n_positions, extras=True):
        """Lets the caching allocator own the output blocks of `n_positions` positions BEFORE the position loop: a caller that
        keeps every position's images (main.run, dist.PositionGatherer) otherwise sends the allocator to hipMalloc once per
        position -- a synchronous call of 0.2 to 2.5 ms depending on the box (gpurun_out/r5s6 against r5s8), in a loop whose
        positions take 0.6 to 1.1 ms.  Allocates the blocks and returns them to the allocator's pool; nothing is kept."""
        if not torch.cuda.is_available():          # the CPU rehearsals of the multi-rank path (gloo tests) compute elsewhere
            return
        dp = self.myDetector.det_param
        nbins = self._close_bins()
        n0, n1 = int(dp['myDimensions'][0]), int(dp['myDimensions'][1])
        # position 0 returns four stacks, every other position four of its own or, with exp_dict['sharedZeroStacks'], two (+ the
        # experiment's shared zero stack)
        per = 2 if self.exp_dict.get('sharedZeroStacks', False) else 4
        blocks = [torch.empty((per, nbins, n0, n1), dtype=torch.float32, device=device()) for _ in range(int(n_positions))]
        blocks.append(torch.empty((4, nbins, n0, n1), dtype=torch.float32, device=device()))
        if extras and self.exp_dict.get('simulation_type') == "RayT":
            # position 0 of the ray-tracing chain also returns two padded displacement maps and a dark-field map (EXP:488-498)
            N = tuple(int(v) for v in self.exp_dict['studyDimensions'])
            blocks += [torch.empty((N[0] + 30, N[1] + 30), dtype=torch.float32, device=device()) for _ in range(2)]
            blocks += [torch.empty(N, dtype=torch.float32, device=device())]
        del blocks

    def _close_bins(self):
        """EXP:296-301, once per Experiment: thresholds inside the spectrum, last spectrum energy appended.  Returns nbins."""
        thr, spec = self.myDetector.det_param["myBinsThersholds"], self.mySource.mySpectrum
        if not self._bins_ready:
            if any(e < spec[0][0] for e in thr) or any(e > spec[-1][0] for e in thr):
                raise Exception(f'At least one of your detector bin threshold is outside your source spectrum. \nYour source spectrum ranges from {spec[0][0]} to {spec[-1][0]}')
            thr.append(spec[-1][0])
            self._bins_ready = True
        return len(thr)

    def _incident(self, flux, energy, ie):
        """Scalar incident intensity per study pixel after the source window (EXP:308,320) and the scintillator
        efficiency (EXP:326-333 / 456-459); air attenuation is returned as a material stack to fuse."""
        ed = self.exp_dict
        I = ed['meanShotCount'] / ed['overSampling'] ** 2 * flux
        det = self.myDetector
        if det.det_param['myScintillatorMaterial'] is not None:
            if ed['simulation_type'] == "Fresnel" or not det.mySpectralEfficiency:
                beta = [b for e, b in det.beta if e == energy][-1]
                I *= 1 - np.exp(-2 * getk(energy * 1000) * det.det_param['myScintillatorThickness'] * 1e-6 * beta)
            else:
                for e, eff in det.mySpectralEfficiency:
                    if e == energy:
                        I *= eff
        return float(I)

    def _effective_source(self):
        ed = self.exp_dict
        return self.mySource.source_dict["mySize"] * ed['distObjectToDetector'] / (ed['distSourceToMembrane'] + ed['distMembraneToObject']) / self.myDetector.det_param['myPixelSize'] * ed['overSampling']   # EXP:380

    def _zero_unvisited_bins(self, stacks, nvisited, pointNum):
        """Bins no energy ever closed (thresholds denser than the spectrum) stay all-zero in the reference (np.zeros,
        EXP:302-305); the stacks here are uninitialised memory until a bin's detection writes them."""
        nbins = stacks[0].shape[0]
        if nvisited < nbins:
            for k in ((0, 1, 2, 3) if pointNum == 0 else (0, 1)):
                ops.fill(stacks[k][nvisited:], 0.0)

    def _detect_bin(self, ibin, pointNum, stacks, accs):
        """EXP:378-401 / 501-521: detection of the accumulated images of one energy bin.  The detector operator writes
        straight into the output stacks and the shot noise of the bin's images is ONE launch, each image under the key of
        what it is (seed, position, bin, kind) -- not of when it was drawn."""
        ess = self._effective_source()
        kinds = (0, 1, 2, 3) if pointNum == 0 else (0, 1)       # Propag / White exist at position 0 only
        self.myDetector.detect_many([accs[k] for k in kinds], ess, self.exp_dict, [stacks[k][ibin] for k in kinds],
                                    [(pointNum, ibin, k) for k in kinds])

    # -------------------------------------------------------------------------------------- Fresnel chain
    def _add_intensity(self, acc, img, plate_att, add=True):
        """acc (+)= img * plate attenuation (EXP:351-358); img is left untouched."""
        ops.accumulate(acc, img, 1.0, plate_att, add=add)

    def _white(self, white, I_scalar, air_rt, plate_att, first):
        """EXP:372-375 / 494-497: the flat-field image of one energy (uniform unless air/plate maps are not)."""
        att = ops.MaterialStack.concat(air_rt, plate_att)
        if first:
            ops.transmit_rt(None, I_scalar, att, want_phi=False, out=white)      # white = I * exp(-2 k beta T) (uniform without maps)
        else:
            w = ops.transmit_rt(None, I_scalar, att, want_phi=False, out=self._tmp[2])
            ops.accumulate(white, w[0], 1.0, None, add=True)

    def _finish_mean_energy(self, sums, npix):
        """EXP:360-361,403: intensity-weighted mean energy of the reference image.  One synchronising read of two float64
        sums per position -- or none, with exp_dict['deferMeanEnergy'] (main.run, bench.py): the sums stay in HBM and
        resolve_mean_energy() folds them in when the value is wanted, so the host keeps running ahead of the GPU."""
        ed = self.exp_dict
        if ed.get('deferMeanEnergy'):
            self._pending_means.append((sums, npix))
            return
        self._fold_mean(ops.fold_sums(sums).tolist(), npix)

    def _fold_mean(self, s, npix):
        ed = self.exp_dict
        ed['meanEnergy'] = (ed.get('meanEnergy', 0) + s[1] / npix) / (s[0] / npix)

    def resolve_mean_energy(self):
        """Fold the deferred per-position sums into exp_dict['meanEnergy'] in call order (one device-to-host copy)."""
        if self._pending_means:
            vals = torch.stack([t for t, _ in self._pending_means])[:, :, :2].sum(dim=1).tolist()
            for s, (_, npix) in zip(vals, self._pending_means):
                self._fold_mean(s, npix)
            self._pending_means = []
        return self.exp_dict.get('meanEnergy', 0)

    # Small study grids (the ones the reference itself is run on) cannot fill the chip: a line kernel is ~25 us of start-up
    # and drain whatever it computes, and a polychromatic position is 7 launches per energy.  There the energies of a
    # detector bin go through the chain TOGETHER (psx_fresnel_propagate_sources, psx_accumulate_many_f32): 7 launches per
    # bin.  On large grids a launch per energy costs nothing and the per-energy loop keeps its fused accumulation.
    BATCH_ENERGIES_MAX_PIXELS = 1137 * 1137        # lines of up to 1137 samples take the 2304-point transform, 8 to a round: fewer
                                                   # line groups than CUs, the case the batched library calls serve in one launch

    def _batch_energies(self, N, plan):
        flag = self.exp_dict.get('batchEnergies')
        if flag is not None:
            return bool(flag) and len(self.mySource.mySpectrum) > 1
        return (len(self.mySource.mySpectrum) > 1 and N[0] * N[1] <= self.BATCH_ENERGIES_MAX_PIXELS and
                (plan is None or plan.engine == _lib.ENGINE_LDS))

    def _bins_of_spectrum(self):
        """EXP:378: the energies of each detector bin, in spectrum order (a bin closes at the first energy above its
        threshold minus half a sampling step; energies after the last threshold are computed and never detected)."""
        thr, half = self.myDetector.det_param["myBinsThersholds"], self.mySource.source_dict["myEnergySampling"] / 2
        bins, cur, ibin = [], [], 0
        for ie, (E, flux) in enumerate(self.mySource.mySpectrum):
            cur.append((ie, E, flux))
            if ibin < len(thr) and E > thr[ibin] - half:
                bins.append(cur)
                cur, ibin = [], ibin + 1
        return bins, cur

    def _bin_tables(self, sim, plate, air):
        """Everything of the batched chains that depends on the energy only -- incident intensities, coefficient rows of every
        material stack the chain uses, chirp scalars / displacement scales -- per detector bin, built once per experiment
        state (the per-energy loop of the reference recomputes them for every membrane position) and re-based on each
        position's maps (ops.MaterialBatch.rebase)."""
        ed, spec = self.exp_dict, self.mySource.mySpectrum
        objs = (self.myMembrane, self.mySampleofInterest, air, plate)
        key = (sim, id(spec), len(spec), tuple(spec[0]), tuple(spec[-1]), tuple(self.myDetector.det_param["myBinsThersholds"]),
               self.mySource.source_dict["myEnergySampling"], ed['distSourceToMembrane'], ed['distMembraneToObject'],
               ed['distObjectToDetector'], ed['magnification'], ed['meanShotCount'], ed['overSampling'], ed['studyPixelSize'],
               tuple(ed['studyDimensions']), self.myDetector.det_param['myScintillatorMaterial'],
               self.myDetector.det_param['myScintillatorThickness'], id(self.myDetector.beta), len(self.myDetector.mySpectralEfficiency),
               tuple((id(o.delta), id(o.beta), len(o.delta), len(o.beta)) if o is not None else None for o in objs))
        if self._tables is not None and self._tables[0] == key:
            return self._tables[1]
        dSM, dMO, dOD, M = ed['distSourceToMembrane'], ed['distMembraneToObject'], ed['distObjectToDetector'], ed['magnification']
        rows = lambda stacks: ([st.cphase for st in stacks], [st.catt for st in stacks])
        bins, leftover = self._bins_of_spectrum()
        out = []
        for energies in bins:
            Es = [E for _, E, _ in energies]
            t = {"Es": Es, "I0": [self._incident(flux, E, ie) for ie, E, flux in energies]}
            t["amp"] = [float(np.sqrt(v)) for v in t["I0"]]                                  # EXP:334
            t["plate"] = rows([plate.stack_rt(E, phase=False) for E in Es]) if plate is not None else None
            t["air_rt"] = rows([air.stack_rt(E, phase=False) for E in Es]) if air is not None else None
            if sim == "Fresnel":
                air_w = [air.stack_wave(E, phase=False) if air is not None else None for E in Es]
                smp = [self.mySampleofInterest.stack_wave(E) for E in Es]
                t["mem"] = rows([ops.MaterialStack.concat(aw, self.myMembrane.stack_wave(E)) for aw, E in zip(air_w, Es)])
                t["smp"] = rows(smp)
                t["smp0"] = rows([ops.MaterialStack.concat(aw, sm) for aw, sm in zip(air_w, smp)])
                sc = [(self._fresnel_scalars(dMO, E, (dSM + dMO) / dSM), self._fresnel_scalars(dOD + dMO, E, M),
                       self._fresnel_scalars(dOD, E, M)) for E in Es]
                t["du"] = sc[0][0][2]
                t["aA"], t["gA"] = [[c[0][0], c[1][0]] for c in sc], [[c[0][1], c[1][1]] for c in sc]
                t["aB"], t["gB"] = [[c[2][0]] for c in sc], [[c[2][1]] for c in sc]
            else:
                air_rt = [air.stack_rt(E, phase=False) if air is not None else None for E in Es]
                mem = [self.myMembrane.stack_rt(E) for E in Es]
                smp = [self.mySampleofInterest.stack_rt(E) for E in Es]
                mem_phase = [m.with_coeffs(catt=[0.0] * m.n) for m in mem]
                t["mem_air"] = rows([ops.MaterialStack.concat(a_, m) for a_, m in zip(air_rt, mem)])
                t["mem_phase"] = rows(mem_phase)
                t["both"] = rows([ops.MaterialStack.concat(mp, sm) for mp, sm in zip(mem_phase, smp)])
                t["air_smp"] = rows([ops.MaterialStack.concat(a_, sm) for a_, sm in zip(air_rt, smp)])
                t["dsMO"] = [self._dscale(dMO, E) for E in Es]
                t["dsOD"] = [self._dscale(dOD, E) for E in Es]
            out.append(t)
        self._tables = (key, (out, leftover))
        return out, leftover

    def _fresnel_bins_batched(self, pointNum, stacks, accs, plan, plate, air, N, sums):
        """The Fresnel chain of one position (EXP:317-401) with the energies of each bin taken together.  Same operations on
        the same numbers as the per-energy loop; the per-energy intensities go through a scratch stack and are summed in
        spectrum order."""
        ed = self.exp_dict
        accS, accR, accP, white = accs
        dSM, dMO, dOD, M = ed['distSourceToMembrane'], ed['distMembraneToObject'], ed['distObjectToDetector'], ed['magnification']
        tables, leftover = self._bin_tables("Fresnel", plate, air)
        nmax = max([len(t["Es"]) for t in tables] + [1])
        if self._etmp is None or self._etmp.shape[0] < nmax or tuple(self._etmp.shape[1:]) != N:
            self._etmp = torch.empty((nmax,) + N, dtype=torch.float32, device=accS.device)
        # this position's maps under the cached coefficient rows (any energy serves: only the maps are taken)
        E0 = self.mySource.mySpectrum[0][0]
        air_w0 = air.stack_wave(E0, phase=False) if air is not None else None
        smp_b = self.mySampleofInterest.stack_wave(E0)
        mem_b = ops.MaterialStack.concat(air_w0, self.myMembrane.stack_wave(E0))
        smp0_b = ops.MaterialStack.concat(air_w0, smp_b)
        plate_b = plate.stack_rt(E0, phase=False) if plate is not None else None
        for ibin, t in enumerate(tables):
            Es, ne = t["Es"], len(t["Es"])
            plate_att = _rebased(plate_b, t["plate"]) if plate is not None else None
            tmp = [self._etmp[k] for k in range(ne)]
            # EXP:341 + EXP:349: membrane exit wave -> sample plane (complex field) and -> detector (|.|^2), every energy
            wbs = plan.propagate_sources(t["aA"], t["gA"], t["du"], amp=t["amp"], mats=_rebased(mem_b, t["mem"]),
                                         want_wave=[True, False], inten_out=[[None, x] for x in tmp])
            self.waveSampleBeforeSample = wbs[-1][0]
            # EXP:355-361: plate attenuation, sum over the bin's energies and the image sums for the mean energy
            ops.accumulate_many(accR, tmp, sums, Es, mats=plate_att, add=False)
            # EXP:344 + EXP:348: through the sample, on to the detector
            plan.propagate_sources(t["aB"], t["gB"], t["du"], wave_in=[w[0] for w in wbs], mats=_rebased(smp_b, t["smp"]),
                                   want_wave=[False], inten_out=[[x] for x in tmp])
            ops.accumulate_many(accS, tmp, None, [0.0] * ne, mats=plate_att, add=False)
            if pointNum == 0:                                                              # EXP:363-375
                plan.propagate_sources(t["aB"], t["gB"], t["du"], amp=t["amp"], mats=_rebased(smp0_b, t["smp0"]),
                                       want_wave=[False], inten_out=[[x] for x in tmp])
                ops.accumulate_many(accP, tmp, None, [0.0] * ne, mats=plate_att, add=False)
                for k, E in enumerate(Es):
                    air_rt = air.stack_rt(E, phase=False) if air is not None else None
                    self._white(white, t["I0"][k], air_rt, plate.stack_rt(E, phase=False) if plate is not None else None, k == 0)
            self._detect_bin(ibin, pointNum, stacks, accs)                                 # EXP:378-401
        if leftover:
            # energies above the last threshold: the reference still propagates them (the field it leaves behind is theirs)
            E = leftover[-1][1]
            I0 = self._incident(leftover[-1][2], E, leftover[-1][0])
            air_w = air.stack_wave(E, phase=False) if air is not None else None
            a1, g1, du = self._fresnel_scalars(dMO, E, (dSM + dMO) / dSM)
            self.waveSampleBeforeSample = plan.propagate([a1], [g1], du, amp=float(np.sqrt(I0)),
                                                         mats=ops.MaterialStack.concat(air_w, self.myMembrane.stack_wave(E)))[0]
        return len(tables)

    def computeSampleAndReferenceImages_Fresnel(self, pointNum):
        """Experiment.py:279-405.  Returns (SampleImage, ReferenceImage, PropagImage, detectedWhite), each
        [nbins, n, n] float32 in HBM."""
        ed = self.exp_dict
        stacks, accs, N, dev, sums = self._begin(pointNum)
        accS, accR, accP, white = accs
        plan = self._plan()
        plate, air = self.myPlate, (None if ed['inVacuum'] else self.myAirVolume)
        tmp = self._tmp[0]
        dSM, dMO, dOD, M = ed['distSourceToMembrane'], ed['distMembraneToObject'], ed['distObjectToDetector'], ed['magnification']
        if self._batch_energies(N, plan):
            nvisited = self._fresnel_bins_batched(pointNum, stacks, accs, plan, plate, air, N, sums)
            self._zero_unvisited_bins(stacks, nvisited, pointNum)
            self._finish_mean_energy(sums, N[0] * N[1])
            return tuple(stacks)
        ibin = 0
        first = True                     # first energy of the current bin: its images are stored, the later ones added
        for ie, (currentEnergy, flux) in enumerate(self.mySource.mySpectrum):
            I0 = self._incident(flux, currentEnergy, ie)
            amp = float(np.sqrt(I0))                                                      # EXP:334
            air_w = air.stack_wave(currentEnergy, phase=False) if air is not None else None   # sqrt(exp(-2 k beta T))
            air_rt = air.stack_rt(currentEnergy, phase=False) if air is not None else None
            plate_att = plate.stack_rt(currentEnergy, phase=False) if plate is not None else None
            mem = ops.MaterialStack.concat(air_w, self.myMembrane.stack_wave(currentEnergy))   # EXP:323,338
            smp = self.mySampleofInterest.stack_wave(currentEnergy)
            a1, g1, du = self._fresnel_scalars(dMO, currentEnergy, (dSM + dMO) / dSM)     # EXP:340 local magnification
            a2, g2, _ = self._fresnel_scalars(dOD + dMO, currentEnergy, M)
            a3, g3, _ = self._fresnel_scalars(dOD, currentEnergy, M)
            # EXP:341 + EXP:349 in one call: membrane exit wave -> sample plane (complex field) and -> detector (|.|^2)
            # (the first energy of a bin without a plate IS the bin's image so far: it is written in place and only summed)
            direct = first and plate_att is None
            wbs = plan.propagate([a1, a2], [g1, g2], du, amp=amp, mats=mem, want_wave=[True, False],
                                 inten_out=[None, accR if direct else tmp])[0]
            self.waveSampleBeforeSample = wbs
            # EXP:355-361: plate attenuation, sum over energies and the image sum for the mean energy in one pass
            if direct:
                ops.accumulate_sum(None, accR, sums, currentEnergy)
            else:
                ops.accumulate_sum(accR, tmp, sums, currentEnergy, mats=plate_att, add=not first)
            # EXP:344 + EXP:348: through the sample, on to the detector
            if plate_att is None:
                plan.propagate([a3], [g3], du, wave_in=wbs, mats=smp, want_wave=[False], inten_out=[accS], add=not first)
            else:
                plan.propagate([a3], [g3], du, wave_in=wbs, mats=smp, want_wave=[False], inten_out=[tmp])
                self._add_intensity(accS, tmp, plate_att, add=not first)
            if pointNum == 0:                                                             # EXP:363-375
                smp0 = ops.MaterialStack.concat(air_w, smp)
                if plate_att is None:
                    plan.propagate([a3], [g3], du, amp=amp, mats=smp0, want_wave=[False], inten_out=[accP], add=not first)
                else:
                    plan.propagate([a3], [g3], du, amp=amp, mats=smp0, want_wave=[False], inten_out=[tmp])
                    self._add_intensity(accP, tmp, plate_att, add=not first)
                self._white(white, I0, air_rt, plate_att, first)
            first = False
            if currentEnergy > self.myDetector.det_param["myBinsThersholds"][ibin] - self.mySource.source_dict["myEnergySampling"] / 2:
                self._detect_bin(ibin, pointNum, stacks, accs)                            # EXP:378-401
                ibin += 1
                first = True
        self._zero_unvisited_bins(stacks, ibin, pointNum)
        self._finish_mean_energy(sums, N[0] * N[1])
        return tuple(stacks)

    # ------------------------------------------------------------------------------------------- RT chain
    def _dscale(self, z, Energy):
        """D[pixels] = grad(phi)[rad/pixel] * dscale  (RF2:54-56; the RT chain uses the TOTAL magnification on
        every hop, EXP:466,473,474)."""
        h = self.exp_dict['studyPixelSize'] * 1e-6
        return z / k_refraction(Energy) / (h * self.exp_dict['magnification']) / h

    def _df_tables(self, DF, z, Energy, N):
        """What fastRefractionDF derives from the width map alone (RF2:114-117,135,171-178): the map in pixels, the per-source
        patch table of the re-splat, the largest patch half-size and the margin.  The map of a sample is static (thickness x a
        scalar of the energy), so one position computes them per energy and the others find them here: the split pass
        (0.10-0.20 ms at 4096^2) runs once per energy of an experiment, not once per position."""
        ed = self.exp_dict
        key = (float(Energy), float(z))
        cache = self.__dict__.setdefault("_df_tab_cache", {})
        hit = cache.get(key)
        if hit is not None and hit[0] is DF:
            cache[key] = cache.pop(key)
            return hit[1:]
        den = ed["studyPixelSize"] * 1e-6 * ed['magnification']      # RF2:114: DF * z / den, in that order
        limit = N[0] / 4                                             # RF2:135
        _, _, DF_px, prep, words = ops.darkfield_split(self._tmp[0], DF, z, den, limit)    # the split images are not used
        known = self.mySampleofInterest.dark_field_max(DF) if hasattr(self.mySampleofInterest, "dark_field_max") else None
        if known is not None and float(known) * z / den <= limit:
            maxDF = maxDFc = float(known) * z / den
        else:
            maxDF, maxDFc = ops.darkfield_maxima(words)
        margin2 = int(np.ceil(maxDF * 6))                            # RF2:117
        R = int(round(1.5 * maxDFc)) + 1
        limit = max(4, min(256, (16 << 30) // max(1, 12 * N[0] * N[1])))      # 12 bytes per pixel and entry, 16 GiB in all
        while len(cache) >= limit:
            cache.pop(next(iter(cache)))
        cache[key] = (DF, DF_px, prep, R, margin2)
        return DF_px, prep, R, margin2

    def _refraction_df_fused(self, Ibs, both, z, Energy, DF, N, clamp, out=None, add=False):
        """The chain's dark-field hop (EXP:469-473 -> RF2:88-196) without its intermediates: the reference forms the transmitted
        (I, phi) pair (SAM:347-348), splits I by the width map (RF2:147-150), refracts both halves and re-splats the dark-field
        half.  Here each halves come out of ONE refraction call straight from the thickness maps with the width map deciding
        a source's half (psx_refract_split_f32: a tile is staged once for both): no (I, phi) pair (12 bytes per pixel written and read twice), no split images, and the tables
        that depend on the width map alone come from _df_tables.  Same arithmetic per source pixel as the unfused calls."""
        DF_px, prep, R, margin2 = self._df_tables(DF, z, Energy, N)
        if margin2 < 1:      # a width map that is zero everywhere: the reference's margin-0 special case, through the literal path
            Ias, phis = ops.transmit_rt(Ibs, 1.0, both)
            img = self.refraction(Ias, phis, z, Energy, self.exp_dict['magnification'], DF, _mutate=False, _want_D=False)[0]
            if out is not None:
                ops.accumulate(out, img, 1.0, None, add=add)
                return out
            return img
        m = max(margin2, 8)
        dscale = self._dscale(z, Energy)
        if self._df_tmp is None or tuple(self._df_tmp.shape) != N:
            self._df_tmp = torch.empty(N, dtype=torch.float32, device=Ibs.device)
        I2, I2DF = ops.refract_split(N, both, dscale, clamp, DF_px, margin=m, I_in=Ibs, outs=[self._tmp[0], self._df_tmp])
        return ops.darkfield_blur_prepared(I2DF, DF_px, prep, I2, R, out=out, add=add)    # the NaN / inf scan (RF2:190-193) rides on its stores

    def _set_halo(self, N, clamp, air):
        """exp_dict['refractionHalo']: 'auto' (default), 4, 6 or 8 -- the gather halo of the refraction tiles is a speed knob whose
        best value depends on how far the rays of THIS experiment travel in study pixels (oversampling, distances, membrane).
        'auto' times the experiment's own longest hop with each halo once, on the first call (ops.tune_refract_halo: three host
        synchronisations), and keeps the winner for the life of the object -- unless the experiment is reproducible (the
        default), where a timing must not decide the bits of an image: see below."""
        if self._halo is None:
            want = self.exp_dict.get('refractionHalo', 'auto')
            if want == 'auto' and self._reproducible():
                # The halo decides which shares are gathered in the tiles and which are replayed, i.e. how a pixel's sum is
                # split into float(tile sum) + float(far sum): the last bit of an image depends on it.  A halo picked by TIMING
                # may differ between ranks and between runs, so a reproducible experiment takes it from a rule instead: how far
                # a ray of its longest hop travels in study pixels per radian of deflection, r = z / (h M).  Fitted to the
                # fixed-point replay's measured optima: the class's own chain at oversampling 2 (hops
                # 1.6 / 3.6 m, r = 1.2e6: halo 4 0.407 ms of refraction per position against 0.424 with 6) and at oversampling 4
                # (r = 2.4e6: 8 wins, 2.52 ms against 2.60 / 2.80 with 6 / 12); a 7.2 m batch at oversampling 2 (r = 2.4e6:
                # 6 or 8) and at oversampling 4 (r = 4.8e6: 12).
                ed = self.exp_dict
                r = max(ed['distMembraneToObject'], ed['distObjectToDetector']) / (ed['studyPixelSize'] * 1e-6 * ed['magnification'])
                want = 4 if r < 1.8e6 else (8 if r < 3.6e6 else 12)
            if want == 'auto':
                ed = self.exp_dict
                E = self.mySource.mySpectrum[-1][0]
                z = max(ed['distMembraneToObject'], ed['distObjectToDetector'])
                stack = ops.MaterialStack.concat(air.stack_rt(E, phase=False) if air is not None else None, self.myMembrane.stack_rt(E))
                out = self._tmp[0]
                self._halo, self._halo_times = ops.tune_refract_halo(
                    lambda: ops.refract(N, stack, self._dscale(z, E), clamp, I0=1.0, out=out),
                    halos=(4, 6, 8, 12, 16) if int(ed.get('overSampling', 1)) >= 4 else (4, 6, 8))
                ops.check_status(out.device, "refraction halo tuning")
            else:
                self._halo = int(want)
        ops.set_refract_halo(self._halo)

    def _rt_bins_batched(self, pointNum, stacks, accs, plate, air, N, sums, clamp):
        """The ray-tracing chain of one position (EXP:448-521) with the energies of each bin taken together
        (psx_refract_batch_f32: one launch per kernel for up to 8 energies); samples without dark field only."""
        ed = self.exp_dict
        accS, accR, accP, white = accs
        dMO, dOD = ed['distMembraneToObject'], ed['distObjectToDetector']
        tables, _ = self._bin_tables("RayT", plate, air)
        nmax = max([len(t["Es"]) for t in tables] + [1])
        if self._etmp is None or self._etmp.shape[0] < 2 * nmax or tuple(self._etmp.shape[1:]) != N:
            self._etmp = torch.empty((2 * nmax,) + N, dtype=torch.float32, device=accS.device)
        # this position's maps under the cached coefficient rows (any energy serves: only the maps are taken)
        E0 = self.mySource.mySpectrum[0][0]
        air_b = air.stack_rt(E0, phase=False) if air is not None else None
        mem_b, smp_b = self.myMembrane.stack_rt(E0), self.mySampleofInterest.stack_rt(E0)
        mem_air_b, both_b = ops.MaterialStack.concat(air_b, mem_b), ops.MaterialStack.concat(mem_b, smp_b)
        air_smp_b = ops.MaterialStack.concat(air_b, smp_b)
        plate_b = plate.stack_rt(E0, phase=False) if plate is not None else None
        for ibin, t in enumerate(tables):
            Es, ne, I0 = t["Es"], len(t["Es"]), t["I0"]
            plate_att = _rebased(plate_b, t["plate"]) if plate is not None else None
            Ibs = [self._etmp[k] for k in range(ne)]
            tmp = [self._etmp[nmax + k] for k in range(ne)]
            # EXP:463 + 466: membrane transmission fused into the first refraction, every energy
            ops.refract_batch(N, _rebased(mem_air_b, t["mem_air"]), t["dsMO"], clamp, I0=I0, outs=Ibs)
            self.IntensitySampleBeforeSample = Ibs[-1]
            # EXP:474 reference images: refracted again with the membrane phase only; EXP:480-486 their attenuated sum
            ops.refract_batch(N, _rebased(mem_b, t["mem_phase"]), t["dsOD"], clamp, I_in=Ibs, outs=tmp)
            ops.accumulate_many(accR, tmp, sums, Es, mats=plate_att, add=False)
            # EXP:469 + 473 sample images: sample attenuation and membrane+sample phase fused into the refraction
            ops.refract_batch(N, _rebased(both_b, t["both"]), t["dsOD"], clamp, I_in=Ibs, outs=tmp)
            ops.accumulate_many(accS, tmp, None, [0.0] * ne, mats=plate_att, add=False)
            if pointNum == 0:                                                             # EXP:488-498
                if ne > 1:
                    head = (t["air_smp"][0][:-1], t["air_smp"][1][:-1])
                    ops.refract_batch(N, _rebased(air_smp_b, head), t["dsOD"][:-1], clamp, I0=I0[:-1], outs=tmp[:-1])
                # the last energy of the bin leaves the displacement maps behind (EXP:492), like the per-energy loop
                last = air_smp_b.with_coeffs(cphase=t["air_smp"][0][-1], catt=t["air_smp"][1][-1])
                _, self.Dxreal, self.Dyreal = ops.refract(N, last, t["dsOD"][-1], clamp, I0=I0[-1], out=tmp[ne - 1], want_D=True)
                ops.accumulate_many(accP, tmp, None, [0.0] * ne, mats=plate_att, add=False)
                for k, E in enumerate(Es):
                    self._white(white, I0[k], air.stack_rt(E, phase=False) if air is not None else None,
                                plate.stack_rt(E, phase=False) if plate is not None else None, k == 0)
            self._detect_bin(ibin, pointNum, stacks, accs)                                # EXP:501-521
        return len(tables)

    def computeSampleAndReferenceImages_RT(self, pointNum):
        """Experiment.py:407-526.  Returns (SampleImage, ReferenceImage, PropagImage, detectedWhite, Dxreal, Dyreal,
        darkFieldPropag); Dxreal/Dyreal are the PADDED [N+30, N+30] maps of the last energy (point 0 only)."""
        # the replay mode is this experiment's (exp_dict['reproducible']), whoever calls -- main.run or a user of the class --
        # and the calling thread gets back the mode it had
        with ops.deterministic(self._reproducible(), scale=self._replay_scale()):
            return self._rt_chain(pointNum)

    def _replay_scale(self):
        """The intensity scale of this experiment's ray-tracing images: the incident intensity per study pixel of its strongest
        energy (EXP:308,320).  Every refraction of the chain then sums its far rays in one unit known before the call -- no
        maximum to measure, no memset node per refraction -- and the unit is a function of exp_dict alone, i.e. the same on
        every rank."""
        ed = self.exp_dict
        flux = max([f for _, f in self.mySource.mySpectrum] + [0.0])
        return float(ed['meanShotCount'] / ed['overSampling'] ** 2 * flux)

    def _rt_chain(self, pointNum):
        ed = self.exp_dict
        stacks, accs, N, dev, sums = self._begin(pointNum)
        accS, accR, accP, white = accs
        plate, air = self.myPlate, (None if ed['inVacuum'] else self.myAirVolume)
        Ibs, tmp = self._tmp[1], self._tmp[0]
        scattering = self.mySampleofInterest.has_dark_field()
        # EXP:444 re-zeros the dark-field map on EVERY call (it only builds up while position 0 is computed, EXP:491): a
        # scattering sample therefore returns zeros for pointNum > 0, whatever positions this object computed before
        # (EXP:444 allocates a NEW array: the map returned for an earlier position stays what it was, so a scattering sample
        # gets a fresh tensor per call -- filling the old one in place would clear what the caller still holds)
        if scattering or not isinstance(self.darkFieldPropag, torch.Tensor) or tuple(self.darkFieldPropag.shape) != N:
            self.darkFieldPropag = ops.fill(torch.empty(N, dtype=torch.float32, device=dev), 0.0)   # scalar dark field: stays zero
        dMO, dOD = ed['distMembraneToObject'], ed['distObjectToDetector']
        clamp = (N[0], N[1])                                                              # RF2:61-64
        self._set_halo(N, clamp, air)
        if not scattering and self._batch_energies(N, None):
            nvisited = self._rt_bins_batched(pointNum, stacks, accs, plate, air, N, sums, clamp)
            self._zero_unvisited_bins(stacks, nvisited, pointNum)
            if not ed.get('deferStatus'):
                ops.check_status(dev, "computeSampleAndReferenceImages_RT")               # RF2:81-82, checked once
            self._finish_mean_energy(sums, N[0] * N[1])
            if not ed.get('deferMeanEnergy'):
                print("Mean detected energy in reference image", ed['meanEnergy'])
            return stacks[0], stacks[1], stacks[2], stacks[3], self.Dxreal, self.Dyreal, self.darkFieldPropag
        ibin = 0
        first = True
        for ie, (currentEnergy, flux) in enumerate(self.mySource.mySpectrum):
            I0 = self._incident(flux, currentEnergy, ie)
            air_rt = air.stack_rt(currentEnergy, phase=False) if air is not None else None
            plate_att = plate.stack_rt(currentEnergy, phase=False) if plate is not None else None
            mem = self.myMembrane.stack_rt(currentEnergy)
            smp = self.mySampleofInterest.stack_rt(currentEnergy)
            # EXP:463 + 466: membrane transmission fused into the first refraction
            ops.refract(N, ops.MaterialStack.concat(air_rt, mem), self._dscale(dMO, currentEnergy), clamp, I0=I0, out=Ibs)
            self.IntensitySampleBeforeSample = Ibs
            mem_phase = mem.with_coeffs(catt=[0.0] * mem.n)
            # EXP:474 reference image: refracted again with the membrane phase only
            # (the first energy of a bin without a plate IS the bin's image so far: it is written in place and only summed)
            direct = first and plate_att is None
            ops.refract(N, mem_phase, self._dscale(dOD, currentEnergy), clamp, I_in=Ibs, out=accR if direct else tmp)
            # EXP:480-486: plate attenuation, sum over energies and the image sum for the mean energy in one pass
            if direct:
                ops.accumulate_sum(None, accR, sums, currentEnergy)
            else:
                ops.accumulate_sum(accR, tmp, sums, currentEnergy, mats=plate_att, add=not first)
            # EXP:469 + 473 sample image: sample attenuation and membrane+sample phase fused into the refraction
            both = ops.MaterialStack.concat(mem_phase, smp)
            if scattering:                                       # Lung / cylinder_beeds: fastRefractionDF (EXP:272-275)
                DF = self.mySampleofInterest.dark_field(currentEnergy)
                if plate_att is None:         # the sum over energies (EXP:478-483) rides on the re-splat's store
                    self._refraction_df_fused(Ibs, both, dOD, currentEnergy, DF, N, clamp, out=accS, add=not first)
                else:
                    img = self._refraction_df_fused(Ibs, both, dOD, currentEnergy, DF, N, clamp)
                    self._add_intensity(accS, img, plate_att, add=not first)
            elif plate_att is None:
                ops.refract(N, both, self._dscale(dOD, currentEnergy), clamp, I_in=Ibs, out=accS, add=not first)
            else:
                ops.refract(N, both, self._dscale(dOD, currentEnergy), clamp, I_in=Ibs, out=tmp)
                self._add_intensity(accS, tmp, plate_att, add=not first)
            if pointNum == 0:                                                             # EXP:488-498
                if scattering:
                    Ip, phip = ops.transmit_rt(None, I0, ops.MaterialStack.concat(air_rt, smp))
                    self.darkFieldPropag += (DF * flux).to(torch.float32)                 # EXP:491
                    img, self.Dxreal, self.Dyreal = self.refraction(Ip, phip, dOD, currentEnergy, ed['magnification'], DF,
                                                                    _mutate=False)
                    self._add_intensity(accP, img, plate_att, add=not first)
                else:
                    _, self.Dxreal, self.Dyreal = ops.refract(N, ops.MaterialStack.concat(air_rt, smp),
                                                              self._dscale(dOD, currentEnergy), clamp, I0=I0, out=tmp,
                                                              want_D=True)
                    self._add_intensity(accP, tmp, plate_att, add=not first)
                self._white(white, I0, air_rt, plate_att, first)
            first = False
            if currentEnergy > self.myDetector.det_param["myBinsThersholds"][ibin] - self.mySource.source_dict["myEnergySampling"] / 2:
                self._detect_bin(ibin, pointNum, stacks, accs)                            # EXP:501-521
                ibin += 1
                first = True
        self._zero_unvisited_bins(stacks, ibin, pointNum)
        if not ed.get('deferStatus'):
            ops.check_status(dev, "computeSampleAndReferenceImages_RT")                   # RF2:81-82, checked once
        self._finish_mean_energy(sums, N[0] * N[1])
        if not ed.get('deferMeanEnergy'):
            print("Mean detected energy in reference image", ed['meanEnergy'])
        return stacks[0], stacks[1], stacks[2], stacks[3], self.Dxreal, self.Dyreal, self.darkFieldPropag

    # ------------------------------------------------------------------------------------------- reporting
    def saveAllParameters(self, time0, expDict):
        """Experiment.py:530-607: text dump of every parameter dictionary next to the images."""
        fileName = expDict['filepath'] + self.name + '_' + str(expDict['expID']) + ".txt"
        print("file name: ", fileName)

        def dump(f, d):
            for cle, valeur in d.items():
                if cle.split('_')[-1] != 'unit':
                    unit = d.get(cle + "_unit")
                    f.write(f'\n    {cle}: {valeur} {unit}' if unit is not None else f'\n    {cle}: {valeur}')

        with open(fileName, "w+") as f:
            f.write("EXPERIMENT PARAMETERS - " + expDict['simulation_type'] + " - " + str(expDict['expID']))
            dump(f, self.exp_dict)
            f.write("\n\nEntire computing time: %gs" % (time.time() - time0))
            f.write("\n\nSource parameters:")
            f.write("\nSource name: %s" % self.mySource.myName)
            dump(f, self.mySource.source_dict)
            f.write("\n\nDetector parameters:")
            f.write("\nDetector name: %s" % self.myDetector.myName)
            dump(f, self.myDetector.det_param)
            f.write("\n\nSample informations")
            f.write("\nSample name: %s" % self.mySampleofInterest.myName)
            f.write("\nSample type: %s" % self.mySampleType)
            f.write("\n    materials: %s" % self.mySampleofInterest.myMaterials)
            f.write("\n    delta/beta source: %s" % getattr(self.mySampleofInterest, "materialProvenance", {}))
            for cle, valeur in (self.mySampleofInterest.geom_parameters or {}).items():
                f.write(f'\n    {cle}: {valeur[0]} {valeur[1]}')
            f.write("\n\nMembrane informations:")
            f.write("\nMembrane name: %s" % self.myMembrane.myName)
            f.write("\nMembrane type: %s" % self.myMembrane.myType)
            f.write("\n    materials: %s" % self.myMembrane.myMaterials)
            f.write("\n    delta/beta source: %s" % getattr(self.myMembrane, "materialProvenance", {}))
            f.write("\n    Membrane geometry function: %s" % self.myMembrane.myGeometryFunction)
            for cle, valeur in (self.myMembrane.geom_parameters or {}).items():
                f.write(f'\n    {cle}: {valeur[0]} {valeur[1]}')
            if self.myPlate is not None:
                f.write("\n\nDetectors protection Plate")
                f.write("\nPlate thickness: %s" % getattr(self.myPlate, "myThickness", None))
                f.write("\nPlate Material: %s" % self.myPlate.myMaterials)
                f.write("\n    delta/beta source: %s" % getattr(self.myPlate, "materialProvenance", {}))
