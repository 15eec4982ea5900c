#!/usr/bin/env python3
"""What the REFERENCE's own parsers read from its own four XML files (tests/fixtures/xml_ref/, copied from
/root/reference/CodePython/xmlFiles: configuration data) -> tests/golden/xml_ref_parsed.json; and the same for the files this package ships (paresis_amd/xmlFiles) ->
tests/golden/xml_pkg_parsed.json.

Run, in the build container only: every Source (Source.py:38-77 + setMySpectrum for the monochromatic one, SRC:90-93), every
Detector (Detector.py:44-76), every sample (Sample.py:33-77) and every experiment (Experiment.defineCorrectValues, EXP:138-197,
on a bare instance) of those files through the reference's code, plus the scalars Experiment.__init__ derives from them by
inline expressions (EXP:81, 96, 204-216) and the effective source size of the detection step (EXP:380), each for the
oversampling the reference's main.py sets (2).  Only the parsed values travel: a JSON file of names and numbers.
        python tests/golden/make_golden_xmlref.py
"""
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/CodePython"
sys.dont_write_bytecode = True
os.environ["MPLBACKEND"] = "Agg"
import numpy as np  # noqa: E402

np.int = int
np.float = float
_nb = types.ModuleType("numba")
_nb.jit = lambda *a, **k: (a[0] if len(a) == 1 and callable(a[0]) and not k else (lambda f: f))
sys.modules["numba"] = _nb
for _name in ["xlrd", "xraylib", "spekpy", "fabio", "fabio.edfimage", "fabio.tifimage", "cv2", "imutils", "skimage",
              "skimage.transform"]:
    sys.modules[_name] = types.ModuleType(_name)
sys.modules["skimage.transform"].radon = None
sys.modules["skimage.transform"].rescale = None
sys.modules["skimage"].transform = sys.modules["skimage.transform"]
sys.path.insert(0, REF)
os.chdir(REF)
from xml.dom import minidom  # noqa: E402

import Detector as DET  # noqa: E402
import Experiment as EXP  # noqa: E402
import Sample as SAM  # noqa: E402
import Source as SRC  # noqa: E402


def plain(v):
    if isinstance(v, dict):
        return {k: plain(x) for k, x in v.items()}
    if isinstance(v, (np.ndarray, tuple, list)):
        return [plain(x) for x in v]
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (np.floating,)):
        return float(v)
    return v


def names(path, element):
    doc = minidom.parse(path)
    return [n.getElementsByTagName("name")[0].childNodes[0].nodeValue for n in doc.documentElement.getElementsByTagName(element)]




def parse_all(root, path):
    """root: a directory holding xmlFiles/ (the reference's classes open "xmlFiles/<name>.xml" relative to the working directory)."""
    os.chdir(root)
    out = {"sources": {}, "detectors": {}, "samples": {}, "experiments": {}}
    for nm in names("xmlFiles/Sources.xml", "source"):
        s = SRC.Source()
        s.myName = nm
        s.defineCorrectValuesSource()
        e = {"source_dict": plain(dict(s.source_dict)), "spectrumFromXls": s.spectrumFromXls}
        if s.source_dict["myType"] == "Monochromatic":
            s.setMySpectrum()
            e["mySpectrum"] = plain(s.mySpectrum)
        out["sources"][nm] = e
    for nm in names("xmlFiles/Detectors.xml", "detector"):
        d = DET.Detector({})
        d.myName = nm
        d.defineCorrectValuesDetector()
        out["detectors"][nm] = {"det_param": plain(dict(d.det_param)), "myEnergyLimit": getattr(d, "myEnergyLimit", None)}
    for nm in names("xmlFiles/Samples.xml", "sample"):
        a = SAM.AnalyticalSample()
        a.myName = nm
        a.defineCorrectValuesSample()
        keep = {k: plain(v) for k, v in vars(a).items() if k not in ("xmldocSample", "xmlSampleFileName", "myGeometry", "geom_parameters",
                                                                      "delta", "beta")}
        out["samples"][nm] = keep
    OV = 2                                                    # main.py:33
    for nm in names("xmlFiles/Experiment.xml", "experiment"):
        x = object.__new__(EXP.Experiment)
        x.xmldoc = minidom.parse("xmlFiles/Experiment.xml")
        x.name = nm
        x.exp_dict = {"overSampling": OV, "inVacuum": False}          # EXP:44: the constructor's default before the parse
        x.myPlate = None
        EXP.Experiment.defineCorrectValues(x, x.exp_dict)
        x.myDetector.defineCorrectValuesDetector()
        x.mySource.defineCorrectValuesSource()
        ed = x.exp_dict
        ed['magnification'] = (ed['distSourceToMembrane'] + ed['distObjectToDetector'] + ed['distMembraneToObject']) / (
            ed['distSourceToMembrane'] + ed['distMembraneToObject'])                                     # EXP:81
        EXP.Experiment.getStudyDimensions(x)                                                            # EXP:204-216
        mem = ed['studyPixelSize'] * ed['distSourceToMembrane'] / (ed['distSourceToMembrane'] + ed['distMembraneToObject'])   # EXP:96
        eff = x.mySource.source_dict["mySize"] * ed['distObjectToDetector'] / (ed['distSourceToMembrane'] + ed['distMembraneToObject']) / \
            x.myDetector.det_param['myPixelSize'] * ed['overSampling']                                 # EXP:380 (FWHM, study pixels)
        out["experiments"][nm] = {
            "exp_dict": {k: plain(v) for k, v in ed.items()},
            "membraneName": x.myMembrane.myName, "sampleName": x.mySampleofInterest.myName, "detectorName": x.myDetector.myName,
            "sourceName": x.mySource.myName, "plateName": x.myPlate.myName if x.myPlate is not None else None,
            "airName": x.myAirVolume.myName, "sampleType": x.mySampleType,
            "membranePixelSize": float(mem), "effectiveSourceSize": float(eff), "precision": float(x.precision),
        }
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path, {k: len(v) for k, v in out.items()})


# the reference's own files, and the files this package ships (same schema, re-authored: the experiments the tests and the bench
# run) -- both through the reference's parsers
parse_all(REF, os.path.join(HERE, "xml_ref_parsed.json"))
parse_all(os.path.join(os.path.dirname(os.path.dirname(HERE)), "paresis_amd"), os.path.join(HERE, "xml_pkg_parsed.json"))
