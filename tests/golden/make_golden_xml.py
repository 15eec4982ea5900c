#!/usr/bin/env python3
"""Golden XML-derived scalars (tests/golden/xml_scalars.npz) from the reference itself: Experiment.getStudyDimensions
(Experiment.py:204-216) is RUN, unbound, on a stub carrying the numbers this package's xmlFiles hold for each shipped
experiment; the magnification (Experiment.py:81) and the membrane pixel size (Experiment.py:96) are inline expressions of
Experiment.__init__ (which cannot run here: xraylib / xlrd / the sphere list are absent) and are evaluated by
the same expressions on the same numbers, in the reference's association order.

Runs only in the build container (imports /root/reference with the shims of make_golden.py); nothing of the reference
travels -- only the numbers written here.      python tests/golden/make_golden_xml.py
"""
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
os.chdir(REF)
sys.path.insert(0, REF)
import Experiment as EXP  # noqa: E402

# (experiment name in paresis_amd/xmlFiles/Experiment.xml, detector dims, detector pixel um, dSM, dMO, dOD, oversampling)
CASES = [("Fil_Nylon_ID17", (200, 200), 6.0, 140.0, 1.6, 3.6, 2),
         ("Sphere_PMMA_plate", (300, 200), 6.0, 140.0, 1.6, 3.6, 1),
         ("Config1_512", (256, 256), 6.0, 140.0, 1.6, 3.6, 2),
         ("Bench_4096", (2048, 2048), 6.0, 140.0, 1.6, 3.6, 2)]
out = {"names": np.array([c[0] for c in CASES])}
for name, dims, pix, dSM, dMO, dOD, ov in CASES:
    stub = types.SimpleNamespace()
    stub.exp_dict = {"distSourceToMembrane": dSM, "distMembraneToObject": dMO, "distObjectToDetector": dOD, "overSampling": ov}
    stub.myDetector = types.SimpleNamespace(det_param={"myPixelSize": pix, "myDimensions": np.array(dims)})
    ed = stub.exp_dict
    ed['magnification'] = (ed['distSourceToMembrane'] + ed['distObjectToDetector'] + ed['distMembraneToObject']) / (
        ed['distSourceToMembrane'] + ed['distMembraneToObject'])                                     # EXP:81
    EXP.Experiment.getStudyDimensions(stub)                                                         # EXP:204-216, the reference's code
    mem = ed['studyPixelSize'] * ed['distSourceToMembrane'] / (ed['distSourceToMembrane'] + ed['distMembraneToObject'])   # EXP:96
    out[name + "/overSampling"] = np.array(ov)
    out[name + "/magnification"] = np.array(ed['magnification'])
    out[name + "/studyDimensions"] = np.array([int(v) for v in ed['studyDimensions']])
    out[name + "/studyPixelSize"] = np.array(ed['studyPixelSize'])
    out[name + "/membranePixelSize"] = np.array(mem)
    out[name + "/precision"] = np.array(stub.precision)
path = os.path.join(HERE, "xml_scalars.npz")
np.savez_compressed(path, **out)
print("wrote", path)
for k in sorted(out):
    print(k, out[k])
