"""Source model: XML -> source_dict -> spectrum [(E_keV, weight)]  (mirror of CodePython/Source.py:18-249).

Only the monochromatic branch (Source.py:90-93) is exact.  Tube spectra need spekpy or an .xls reader, neither of which
exists in this image; a polychromatic source therefore takes an injected spectrum (`source_dict['spectrum']` or
`Source.spectrum_provider`) and otherwise falls back to a documented Kramers-law stand-in (out of the hot path).
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
            if "spectrum" in sd:
                spec = [(float(e), float(w)) for e, w in sd["spectrum"]]
            elif Source.spectrum_provider is not None:
                spec = list(Source.spectrum_provider(sd, flu_fluEn))
            else:
                spec = self._kramers(sd)
                sd["spectrumModel"] = "kramers-synthetic (spekpy/xlrd unavailable)"
            tot = sum(w for _, w in spec)
            # normalise and drop bins below 1e-4 of the flux like Source.py:118-123
            self.mySpectrum.extend((e, w / tot) for e, w in spec if w / tot > 0.0001)
            return
        raise ValueError("unknown source type %r" % sd["myType"])

    @staticmethod
    def _kramers(sd):
        kvp = float(sd.get("myVoltage", 50.0))
        dk = float(sd["myEnergySampling"])
        e = np.arange(dk * np.ceil(8.0 / dk), kvp, dk)
        w = (kvp / e - 1.0)
        if sd.get("filterMaterial") is not None:
            w = w * np.exp(-float(sd.get("filterThickness", 0.0)) * 50.0 * (10.0 / e) ** 3)
        return [(float(a), float(b)) for a, b in zip(e, w) if b > 0]
