"""Helpers shared by the CPU and GPU test files: golden loading and the error metric."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def relmax(a, ref):
    """max|a-ref| / max|ref|: the metric of SURVEY.md section 8d (pointwise-relative is meaningless in Fresnel minima)."""
    a = np.asarray(a)
    ref = np.asarray(ref)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    den = np.max(np.abs(ref))
    if den == 0:
        return float(np.max(np.abs(a)))
    return float(np.max(np.abs(a - ref)) / den)


def experiment_cfg(g, tag, Obj):
    """Rebuild the chain configuration of tests/golden/experiment.npz[tag] (tag = 'mono/RT', 'poly/Fresnel', ...)."""
    e = g[tag + "/exp"]
    det = g[tag + "/det"]
    src = g[tag + "/source"]

    def obj(nm):
        key = "%s/%s/geometry" % (tag, nm)
        if key not in g.files:
            return None
        return Obj(g[key], g["%s/%s/delta" % (tag, nm)], g["%s/%s/beta" % (tag, nm)])

    scint = None
    if tag + "/scint_beta" in g.files:
        scint = (float(g["scint/thickness_um"]), [(float(a), float(b)) for a, b in g[tag + "/scint_beta"]])
    return dict(scintillator=scint, dSM=float(e[0]), dMO=float(e[1]), dOD=float(e[2]), meanShotCount=float(e[3]), ov=int(e[4]),
                pix_um=float(e[5]), M=float(e[6]), inVacuum=bool(g[tag + "/inVacuum"]),
                N=tuple(int(v) for v in g[tag + "/studyDimensions"]),
                spectrum=[(float(a), float(b)) for a, b in g[tag + "/spectrum"]],
                source_size_um=float(src[0]), energy_sampling=float(src[1]),
                det_dims=(int(det[0]), int(det[1])), det_pix_um=float(det[2]), psf=float(det[3]),
                bins=[float(v) for v in g[tag + "/bins"]],
                membrane=obj("membrane"), sample=obj("sample"), air=obj("air"), plate=obj("plate"))
