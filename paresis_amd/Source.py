"""Source model: XML -> source_dict -> spectrum [(E_keV, weight)]  (mirror of CodePython/Source.py:18-249).

Tube spectra need spekpy or an .xls reader, neither of which exists in this image, so a polychromatic source takes its
raw data injected and applies the reference's own processing to it: `source_dict['spectrum']` (or
`Source.spectrum_provider`) stands for spekpy's output (NaN -> 0, normalise, 1e-4 threshold: Source.py:108-123);
`source_dict['xlsRows']` with `spectrumFromXls` stands for the sheet rows (unit scaling, re-binning, 1e-3 threshold:
Source.py:132-233).  Without either, a documented Kramers-law stand-in is used (out of the hot path).
"""
import numpy as np

from . import _xml


class Source:
    spectrum_provider = None   # optional callable(source_dict, photon_counting) -> [(E_keV, weight)]

    def __init__(self, xml_directory=None):
        self.xmlSourcesFileName = "xmlFiles/Sources.xml"
        self._xml_directory = xml_directory
        self.myName = ""
        self.mySpectrum = []
        self.source_dict = {"mySize": 0., "myEnergySampling": 1, "myType": None}
        self.spectrumFromXls = False
        # units (Source.py:31-35)
        self.source_dict["mySize_unit"] = "um"
        self.source_dict["myEnergySampling_unit"] = "keV"
        self.source_dict["myVoltage_unit"] = "kVp"
        self.source_dict["Energy_unit"] = "keV"
        self.source_dict["filterThickness_unit"] = "mm"

    def defineCorrectValuesSource(self):
        """Source.py:38-77."""
        doc = _xml.parse(self._xml_directory or _xml.xml_dir(), "Sources.xml")
        node = _xml.find_named(doc, "source", self.myName)
        if node is None:
            raise ValueError("Source not found in the xml file")
        sd = self.source_dict
        sd["mySize"] = float(_xml.child_text(node, "mySize"))
        sd["myType"] = _xml.child_text(node, "myType")
        if sd["myType"] == "Polychromatic":
            sd["filterMaterial"] = None
            sd["myEnergySampling"] = float(_xml.child_text(node, "myEnergySampling"))
            if _xml.has_child(node, "sourceVoltage"):
                sd["myVoltage"] = float(_xml.child_text(node, "sourceVoltage"))
            if _xml.has_child(node, "spectrumFromXls"):
                self.spectrumFromXls = bool(_xml.child_text(node, "spectrumFromXls"))   # bool("False") is True: SRC:62
                for key in ("pathXlsSpectrum", "energyUnit", "energyColumnKey", "fluenceColumnKey"):
                    sd[key] = _xml.child_text(node, key)
            if _xml.has_child(node, "filterMaterial"):
                sd["filterMaterial"] = _xml.child_text(node, "filterMaterial")
                sd["filterThickness"] = float(_xml.child_text(node, "filterThickness"))
            if _xml.has_child(node, "myTargetMaterial"):
                sd["myTargetMaterial"] = _xml.child_text(node, "myTargetMaterial")
        if sd["myType"] == "Monochromatic":
            sd["myEnergySampling"] = 1
            sd["Energy"] = float(_xml.child_text(node, "myEnergy"))

    def setMySpectrum(self, flu_fluEn=True):
        """Source.py:79-240."""
        sd = self.source_dict
        if sd["myType"] == "Monochromatic":
            self.mySpectrum.append((sd["Energy"], 1))       # SRC:90-93
            return
        if sd["myType"] == "Polychromatic":
            if self.spectrumFromXls:
                if "xlsRows" not in sd:
                    sd["xlsRows"] = self._read_xls_rows(sd)        # Source.py:132-150 (needs xlrd)
                self.mySpectrum.extend(self._from_table_rows(sd))
                sd["spectrumModel"] = "tabulated (%s)" % sd.get("pathXlsSpectrum", "xlsRows injected")
                return
            if "spectrum" in sd:
                # NaN bins count as zero (Source.py:111-113)
                spec = [(float(e), 0.0 if np.isnan(w) else float(w)) for e, w in sd["spectrum"]]
            elif Source.spectrum_provider is not None:
                spec = list(Source.spectrum_provider(sd, flu_fluEn))
            else:
                spec = self._spekpy(sd, flu_fluEn)                  # Source.py:96-108, when spekpy is importable
                if spec is None:
                    import warnings
                    warnings.warn("polychromatic source %r: spekpy is not available and no spectrum was injected "
                                  "(source_dict['spectrum'] / Source.spectrum_provider); using a SYNTHETIC Kramers-law "
                                  "spectrum." % self.myName, UserWarning, stacklevel=2)
                    spec = self._kramers(sd)
                    sd["spectrumModel"] = "kramers-synthetic (spekpy/xlrd unavailable)"
                else:
                    sd["spectrumModel"] = "spekpy"
            tot = sum(w for _, w in spec)
            # normalise and drop bins below 1e-4 of the flux like Source.py:118-123
            self.mySpectrum.extend((e, w / tot) for e, w in spec if w / tot > 0.0001)
            return
        raise ValueError("unknown source type %r" % sd["myType"])

    @staticmethod
    def _spekpy(sd, flu_fluEn):
        """The reference's tube spectrum (Source.py:96-108) when spekpy is importable, else None."""
        try:
            import spekpy as sp
        except ImportError:
            return None
        sd.setdefault("myTargetMaterial", 'W')
        s = sp.Spek(kvp=sd["myVoltage"], th=12, targ=sd["myTargetMaterial"], dk=sd["myEnergySampling"])
        if sd.get("filterMaterial") is not None:
            s.filter(sd["filterMaterial"], sd["filterThickness"])
        energies, weights = s.get_spectrum(flu=flu_fluEn)
        return [(float(e), 0.0 if np.isnan(w) else float(w)) for e, w in zip(energies, weights)]

    @staticmethod
    def _read_xls_rows(sd):
        """Rows (energy, fluence) of the sheet named by pathXlsSpectrum / energyColumnKey / fluenceColumnKey
        (Source.py:132-150).  Needs xlrd like the reference; without it the rows must be injected as source_dict['xlsRows']."""
        try:
            import xlrd
        except ImportError as exc:
            raise ImportError("spectrumFromXls is set but xlrd is not available to read %r: inject the sheet rows as "
                              "source_dict['xlsRows'] = [(energy, fluence), ...]" % sd.get("pathXlsSpectrum")) from exc
        sh = xlrd.open_workbook(sd["pathXlsSpectrum"]).sheets()[0]
        cols = {str(sh.cell(0, c).value): c for c in range(sh.ncols)}
        ce, cf = cols[sd["energyColumnKey"]], cols[sd["fluenceColumnKey"]]
        return [(float(sh.cell(r, ce).value), float(sh.cell(r, cf).value)) for r in range(1, sh.nrows)]

    @staticmethod
    def _from_table_rows(sd):
        """Tabulated spectrum, Source.py:132-233 after the sheet has been read: `xlsRows` = [(energy, fluence)] in
        `energyUnit`, re-binned to `myEnergySampling` keV.  The reference's arithmetic is kept as is: the bin-width
        counter advances by the step of the first two rows, Nbin-1 full bins + one tail bin, the normalisation total
        leaves the tail bin out, bins at or under 0.001 are dropped (pinned by tests/golden/frontend.npz)."""
        scale = {"eV": 0.001, "MeV": 1000}.get(sd.get("energyUnit"), 1)
        spectrum = [[float(e) * scale, float(f)] for e, f in sd["xlsRows"]]
        sampling = sd["myEnergySampling"]
        den = spectrum[1][0] - spectrum[0][0]
        n_en = len(spectrum)
        n_bin = int((spectrum[-1][0] - spectrum[0][0]) // sampling)
        energies, weights = [], []
        n = 0
        tot = 0
        for _ in range(n_bin - 1):
            curr, w, eb = 0, 0, 0
            while curr < sampling:
                w += spectrum[n][1]
                eb += spectrum[n][1] * spectrum[n][0]
                n += 1
                curr = curr + den
            if w != 0:
                energies.append(eb / w)
                weights.append(w)
            tot += w
        w, eb = 0, 0
        while n < n_en:
            w += spectrum[n][1]
            eb += spectrum[n][1] * spectrum[n][0]
            n += 1
        if w != 0:
            energies.append(eb / w)
            weights.append(w)
        return [(e, wt / tot) for e, wt in zip(energies, weights) if wt / tot > 0.001]

    @staticmethod
    def _kramers(sd):
        kvp = float(sd.get("myVoltage", 50.0))
        dk = float(sd["myEnergySampling"])
        e = np.arange(dk * np.ceil(8.0 / dk), kvp, dk)
        w = (kvp / e - 1.0)
        if sd.get("filterMaterial") is not None:
            w = w * np.exp(-float(sd.get("filterThickness", 0.0)) * 50.0 * (10.0 / e) ** 3)
        return [(float(a), float(b)) for a, b in zip(e, w) if b > 0]
