#!/usr/bin/env python3
"""Generate the golden input/output vectors in tests/golden/*.npz from the reference itself.

Runs ONLY in the build container, where the reference lives at /root/reference (it never travels to the
GPU box; only the arrays written here do).  The reference is imported unmodified, function by function, with
the compatibility shims recorded in SURVEY.md section 8c:

  * `numba` is not installed: a stand-in module whose `jit` is the identity decorator, so the two @jit loops
    (refractionFileNumba2.py:198, Detector.py:185) run in the interpreter;
  * `np.int` / `np.float` (removed numpy aliases used at refractionFileNumba2.py:72-73) are restored;
  * empty module objects for the optional third-party imports the hot path never calls
    (xlrd, xraylib, spekpy, fabio, cv2, imutils, skimage);
  * Poisson shot noise (Detector.py:113-115, time-seeded) is bypassed by a proxy so that the recorded
    `detection` output is the deterministic pre-noise image.

Usage:  python tests/golden/make_golden.py            (writes next to this file)
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/CodePython"

sys.dont_write_bytecode = True
os.environ["MPLBACKEND"] = "Agg"
import numpy as np  # noqa: E402

np.int = int
np.float = float

_nb = types.ModuleType("numba")


def _jit(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


_nb.jit = _jit
sys.modules["numba"] = _nb
for _name in ["xlrd", "xraylib", "spekpy", "fabio", "fabio.edfimage", "fabio.tifimage", "cv2", "imutils",
              "skimage", "skimage.transform"]:
    sys.modules[_name] = types.ModuleType(_name)
sys.modules["skimage.transform"].radon = None
sys.modules["skimage.transform"].rescale = None
sys.modules["skimage"].transform = sys.modules["skimage.transform"]

sys.path.insert(0, REPO)
from paresis_amd import synth  # noqa: E402

_cwd = os.getcwd()
os.chdir(REF)  # the reference's constructors parse xmlFiles/*.xml relative to cwd
sys.path.insert(0, REF)
import refractionFileNumba2 as RF2  # noqa: E402
import refractionFileNumba as RF1  # noqa: E402
import Experiment as EXP  # noqa: E402
import Detector as DET  # noqa: E402
import Sample as SAM  # noqa: E402
import Source as SRC  # noqa: E402
import getk as GETK  # noqa: E402


class _NoNoiseRandomState:
    def __init__(self, seed=None):
        pass

    def poisson(self, lam):
        return lam


class _NpProxy:
    """numpy with random.RandomState(seed).poisson(x) == x (Detector.py:113-115)."""

    class random:  # noqa: N801
        RandomState = _NoNoiseRandomState

    def __getattr__(self, name):
        return getattr(np, name)


DET.np = _NpProxy()

rng = np.random.Generator(np.random.PCG64(20261004))


def save(name, d):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **d)
    print("wrote", path, "%.1f KiB" % (os.path.getsize(path) / 1024.0))


# ---------------------------------------------------------------------------------------------------------
def gold_scalars():
    d = {}
    E = np.array([25000.0, 52000.0, 17.5e3, 100e3, 1.0])
    d["getk/E_eV"] = E
    d["getk/k"] = np.array([GETK.getk(e) for e in E])
    sig = np.array([0.5, 0.8333, 1.0, 1.2, 1.5, 10 * 3.6 / 141.6 / 6 * 2 / 2.355, 2.5, 0.1667, 0.25])
    d["gauss/sigma"] = sig
    for i, s in enumerate(sig):
        d["gauss/%d/det" % i] = DET.create_gaussian_shape(s)
        d["gauss/%d/rf2" % i] = RF2.gaussian_shape(s)
    k = 0
    for shape, (sx, sy) in [((12, 12), (12, 12)), ((12, 8), (6, 4)), ((16, 12), (4, 3)), ((9, 9), (3, 3)),
                            ((10, 20), (5, 10))]:
        a = rng.uniform(0, 5, shape)
        d["resize/%d/in" % k] = a
        d["resize/%d/size" % k] = np.array([sx, sy])
        d["resize/%d/out" % k] = np.array(DET.resize(a.copy(), sx, sy))
        k += 1
    d["resize/n"] = np.array(k)
    save("scalars.npz", d)


def make_sample(name, mtype, materials, geometry, energies, deltas, betas):
    s = object.__new__(SAM.AnalyticalSample)
    s.myName = name
    s.myType = mtype
    s.myMaterials = list(materials)
    s.myGeometry = np.asarray(geometry, dtype=np.float64)
    s.delta = [[(e, dl[i]) for i, e in enumerate(energies)] for dl in deltas]
    s.beta = [[(e, bl[i]) for i, e in enumerate(energies)] for bl in betas]
    s.geom_parameters = None
    s.myGeometryFunction = "injected"
    return s


def gold_transmission():
    d = {}
    Nx, Ny = 40, 36
    pix = 3e-6
    T = np.stack([synth.sphere_membrane(Nx, Ny, pix, 3).astype(np.float64), synth.slab(Nx, Ny, 6e-3).astype(np.float64)])
    energies = [52.0, 30.0]
    deltas = [[6.2e-7, 1.9e-6], [9.87e-8, 2.9e-7]]
    betas = [[4.0e-9, 3.0e-8], [4.5e-11, 1.6e-10]]
    s = make_sample("mem", "membrane", ["CuSn", "PMMA"], T, energies, deltas, betas)
    wave = (rng.uniform(0.5, 1.5, (Nx, Ny)) * np.exp(1j * rng.uniform(-3, 3, (Nx, Ny))))
    I0 = rng.uniform(100, 200, (Nx, Ny))
    phi0 = rng.uniform(-2, 2, (Nx, Ny))
    d["T"] = T
    d["energies"] = np.array(energies)
    d["delta"] = np.array(deltas)
    d["beta"] = np.array(betas)
    d["wave_in"] = wave
    d["I_in"] = I0
    d["phi_in"] = phi0
    for ie, E in enumerate(energies):
        d["setWave/%d" % ie] = s.setWave(wave.copy(), E)
        I, phi, df = s.setWaveRT(I0.copy(), E, phi0.copy())
        d["setWaveRT/%d/I" % ie] = I
        d["setWaveRT/%d/phi" % ie] = phi
        d["setWaveRT/%d/df" % ie] = np.array(df)
        I, phi, df = s.setWaveRT(I0.copy(), E)  # default scalar phase 0
        d["setWaveRT0/%d/I" % ie] = I
        d["setWaveRT0/%d/phi" % ie] = phi
    save("transmission.npz", d)


class _ExpStub:
    pass


def gold_fresnel():
    d = {}
    cases = []
    k = 0
    for (Nx, Ny) in [(64, 64), (65, 65), (96, 80), (50, 71)]:
        pix_um = 6.0 / 2 / (145.2 / 141.6)
        T = synth.sphere_membrane(Nx, Ny, pix_um * 1e-6, 7 + k).astype(np.float64)
        amp = np.sqrt(7500.0) * np.exp(-2.63e11 * 4.0e-9 * T)
        wave = amp * np.exp(-1j * 2.63e11 * 6.2e-7 * T)
        for (z, E, M) in [(1.6, 52.0, 141.6 / 140.0), (3.6, 52.0, 145.2 / 141.6), (5.2, 25.0, 145.2 / 141.6),
                          (0.0, 52.0, 1.0), (0.35, 17.5, 2.0)]:
            stub = _ExpStub()
            stub.exp_dict = {"studyDimensions": [Nx, Ny], "studyPixelSize": pix_um}
            out = EXP.Experiment.wavePropagation(stub, wave.copy(), z, E, M)
            d["%d/wave" % k] = wave
            d["%d/params" % k] = np.array([z, E, M, pix_um])
            d["%d/out" % k] = out
            cases.append(k)
            k += 1
    d["n"] = np.array(k)
    save("fresnel.npz", d)


def refraction_inputs(Nx, Ny, pix_um, seed, delta=6.2e-7, E=52.0):
    kk = 2 * np.pi * E * 1000 * 1.6e-19 / (6.626e-34 * 2.998e8)
    T = synth.sphere_membrane(Nx, Ny, pix_um * 1e-6, seed).astype(np.float64)
    phi = -kk * delta * T - kk * 9.87e-8 * 6e-3
    I = 7500.0 * np.exp(-2 * kk * 4.0e-9 * T) * (1.0 + 0.05 * np.sin(np.arange(Nx)[:, None] * 0.3 + np.arange(Ny)[None, :] * 0.17))
    return I, phi, T


def gold_refraction():
    d = {}
    k = 0
    pix_um = 6.0 / 2 / (145.2 / 141.6)
    # (shape, z, M, note): small / medium / leaving the margin (|D|>15) / clamp (|D|>N)
    for (Nx, Ny), z, M, note in [((64, 64), 1.6, 145.2 / 141.6, "small"),
                                 ((96, 80), 3.6, 145.2 / 141.6, "medium"),
                                 ((65, 65), 3.6, 145.2 / 141.6, "odd"),
                                 ((64, 64), 40.0, 145.2 / 141.6, "leave_margin"),
                                 ((48, 40), 2500.0, 1.0, "clamp"),
                                 ((33, 47), 14.0, 1.3, "ragged")]:
        I, phi, T = refraction_inputs(Nx, Ny, pix_um, 11 + k)
        for ver, mod in (("v2", RF2), ("v1", RF1)):
            Iin = I.copy()
            out, Dx, Dy = mod.fastRefraction(Iin, phi.copy(), z, 52.0, M, pix_um)
            d["%d/%s/out" % (k, ver)] = out
            d["%d/%s/Dx" % (k, ver)] = Dx
            d["%d/%s/Dy" % (k, ver)] = Dy
            d["%d/%s/I_after" % (k, ver)] = Iin  # the reference zeroes clamped entries in place (RF2:61-62)
        d["%d/I" % k] = I
        d["%d/phi" % k] = phi
        d["%d/T" % k] = T
        d["%d/params" % k] = np.array([z, 52.0, M, pix_um])
        print("refraction case", k, note, "max|Dx|", np.abs(d["%d/v2/Dx" % k]).max(), "max|Dy|", np.abs(d["%d/v2/Dy" % k]).max())
        k += 1
    d["n"] = np.array(k)

    # raw loop on hand-made displacement fields (exercises every sign/|D|>1 branch and the border rules)
    Nx, Ny = 20, 17
    I = rng.uniform(1, 2, (Nx, Ny))
    Dx = rng.uniform(-3.5, 3.5, (Nx, Ny))
    Dy = rng.uniform(-3.5, 3.5, (Nx, Ny))
    Dx[2, 3] = 0.0
    Dy[2, 3] = 0.0
    Dx[4, 4] = 1.0
    Dy[4, 4] = -1.0
    Dx[5, 5] = 0.0
    Dy[5, 5] = 0.75
    Dx[6, 6] = -2.0
    Dy[6, 6] = 0.0
    Dx[0, :] = -0.5
    Dx[-1, :] = 0.5
    Dy[:, 0] = -0.25
    Dy[:, -1] = 0.25
    I2 = np.zeros((Nx, Ny))
    out = RF2.fastloopNumba(Nx, Ny, I.copy(), I2, Dy.copy(), Dx.copy(), Dx.astype(int), Dy.astype(int))
    d["loop/I"] = I
    d["loop/Dx"] = Dx
    d["loop/Dy"] = Dy
    d["loop/out"] = np.array(out)
    save("refraction.npz", d)


def make_detector(dims, pix_um, psf, bins=None):
    det = object.__new__(DET.Detector)
    det.myName = "synthetic"
    det.det_param = {"myDimensions": np.array(dims), "myPixelSize": float(pix_um), "myPSF": float(psf),
                     "myBinsThersholds": list(bins or []), "myScintillatorMaterial": None,
                     "myScintillatorThickness": 0.0, "photonCounting": True}
    det.mySpectralEfficiency = []
    det.beta = []
    return det


def gold_detector():
    d = {}
    k = 0
    for dims, ov, fwhm, psf in [((24, 20), 1, 0.0, 0.0), ((24, 20), 2, 0.0, 0.0), ((24, 20), 2, 0.3595, 0.0),
                                ((20, 24), 4, 1.5 * 2.355, 1.0), ((16, 16), 2, 1.2, 1.2), ((24, 20), 1, 0.9, 1.0),
                                ((12, 14), 3, 2.0, 0.5)]:
        det = make_detector(dims, 6.0, psf)
        N = (dims[0] * ov, dims[1] * ov)
        img = rng.uniform(50, 150, N) * (1 + 0.3 * np.sin(np.arange(N[0])[:, None] * 0.5))
        out = det.detection(img.copy(), fwhm, {"overSampling": ov})
        d["%d/in" % k] = img
        d["%d/params" % k] = np.array([dims[0], dims[1], ov, fwhm, psf])
        d["%d/out" % k] = np.asarray(out, dtype=np.float64)
        k += 1
    d["n"] = np.array(k)
    save("detector.npz", d)


def build_experiment(det_dims, ov, spectrum, bins, energy_sampling, psf, src_size, in_vacuum, with_plate, seed0):
    dSM, dMO, dOD = 140.0, 1.6, 3.6
    exp = object.__new__(EXP.Experiment)
    M = (dSM + dMO + dOD) / (dSM + dMO)
    det_pix = 6.0
    N = [det_dims[0] * ov, det_dims[1] * ov]
    pix_um = det_pix / ov / M
    exp.exp_dict = {"experimentName": "synthetic", "overSampling": ov, "nbExpPoints": 2, "simulation_type": "RayT",
                    "studyPixelSize": pix_um, "studyDimensions": N, "inVacuum": in_vacuum, "meanShotCount": 30000.0,
                    "meanEnergy": 0, "distSourceToMembrane": dSM, "distMembraneToObject": dMO,
                    "distObjectToDetector": dOD, "magnification": M}
    src = object.__new__(SRC.Source)
    src.myName = "synthetic"
    src.mySpectrum = list(spectrum)
    src.source_dict = {"mySize": src_size, "myEnergySampling": energy_sampling,
                       "myType": "Monochromatic" if len(spectrum) == 1 else "Polychromatic"}
    exp.mySource = src
    exp.myDetector = make_detector(det_dims, det_pix, psf, bins)
    energies = [e for e, _ in spectrum]
    # synthetic delta/beta ~ E^-2 / E^-3 scaling from the 52 keV values
    def db(name):
        d0, b0 = synth.DELTA_BETA_52KEV[name]
        return [d0 * (52.0 / e) ** 2 for e in energies], [b0 * (52.0 / e) ** 3 for e in energies]
    mem_pix = pix_um * dSM / (dSM + dMO)
    def membrane_geom(point):
        return np.stack([synth.sphere_membrane(N[0], N[1], mem_pix * 1e-6, seed0 + point).astype(np.float64),
                         synth.slab(N[0], N[1], 6e-3).astype(np.float64)])
    dl, bl = zip(db("CuSn"), db("PMMA"))
    exp.myMembrane = make_sample("membrane", "membrane", ["CuSn", "PMMA"], membrane_geom(0), energies, list(dl), list(bl))
    dl, bl = db("Nylon")
    exp.mySampleofInterest = make_sample("sample", "sample_of_interest", ["Nylon"],
                                         synth.cylinder_sample(N[0], N[1], pix_um * 1e-6).astype(np.float64)[None],
                                         energies, [dl], [bl])
    dl, bl = db("air")
    exp.myAirVolume = make_sample("air_volume", "air", ["air"], synth.slab(N[0], N[1], dSM + dMO + dOD).astype(np.float64)[None],
                                  energies, [dl], [bl])
    exp.myPlate = None
    if with_plate:
        dl, bl = db("C")
        exp.myPlate = make_sample("plate", "plate", ["C"], synth.slab(N[0], N[1], 1e-3).astype(np.float64)[None],
                                  energies, [dl], [bl])
    exp.Dxreal = []
    exp.Dyreal = []
    return exp, membrane_geom


def record_inputs(d, tag, exp):
    d[tag + "/exp"] = np.array([exp.exp_dict[k] for k in ("distSourceToMembrane", "distMembraneToObject",
                                                           "distObjectToDetector", "meanShotCount", "overSampling",
                                                           "studyPixelSize", "magnification")], dtype=np.float64)
    d[tag + "/inVacuum"] = np.array(bool(exp.exp_dict["inVacuum"]))
    d[tag + "/studyDimensions"] = np.array(exp.exp_dict["studyDimensions"])
    d[tag + "/spectrum"] = np.array(exp.mySource.mySpectrum, dtype=np.float64)
    d[tag + "/source"] = np.array([exp.mySource.source_dict["mySize"], exp.mySource.source_dict["myEnergySampling"]], dtype=np.float64)
    d[tag + "/det"] = np.array([exp.myDetector.det_param["myDimensions"][0], exp.myDetector.det_param["myDimensions"][1],
                                exp.myDetector.det_param["myPixelSize"], exp.myDetector.det_param["myPSF"]], dtype=np.float64)
    d[tag + "/bins"] = np.array(exp.myDetector.det_param["myBinsThersholds"], dtype=np.float64)
    for nm, s in (("membrane", exp.myMembrane), ("sample", exp.mySampleofInterest), ("air", exp.myAirVolume), ("plate", exp.myPlate)):
        if s is None:
            continue
        d[tag + "/" + nm + "/geometry"] = s.myGeometry
        d[tag + "/" + nm + "/delta"] = np.array([[v for _, v in l] for l in s.delta])
        d[tag + "/" + nm + "/beta"] = np.array([[v for _, v in l] for l in s.beta])


def gold_experiment():
    d = {}
    configs = [
        # tag, det_dims, ov, spectrum, bins, sampling, psf, src_size_um, in_vacuum, plate
        ("mono", (48, 40), 2, [(52.0, 1)], [], 1, 0.0, 10.0, True, False),
        ("poly", (32, 36), 2, [(20.0, 0.25), (24.0, 0.45), (28.0, 0.30)], [24.0], 4.0, 1.2, 50.0, False, True),
    ]
    for tag, dims, ov, spec, bins, samp, psf, ssz, vac, plate in configs:
        for sim in ("RT", "Fresnel"):
            exp, membrane_geom = build_experiment(dims, ov, spec, bins, samp, psf, ssz, vac, plate, 40)
            t = "%s/%s" % (tag, sim)
            record_inputs(d, t, exp)
            for point in (0, 1):
                exp.myMembrane.myGeometry = membrane_geom(point)
                d["%s/p%d/membrane" % (t, point)] = exp.myMembrane.myGeometry
                exp.exp_dict["meanEnergy"] = 0
                if sim == "RT":
                    S, R, Pg, W, Dx, Dy, DF = exp.computeSampleAndReferenceImages_RT(point)
                    d["%s/p%d/Dx" % (t, point)] = np.asarray(Dx, dtype=np.float64)
                    d["%s/p%d/Dy" % (t, point)] = np.asarray(Dy, dtype=np.float64)
                    d["%s/p%d/DF" % (t, point)] = np.asarray(DF, dtype=np.float64)
                else:
                    S, R, Pg, W = exp.computeSampleAndReferenceImages_Fresnel(point)
                d["%s/p%d/Sample" % (t, point)] = np.asarray(S, dtype=np.float64)
                d["%s/p%d/Reference" % (t, point)] = np.asarray(R, dtype=np.float64)
                d["%s/p%d/Propag" % (t, point)] = np.asarray(Pg, dtype=np.float64)
                d["%s/p%d/White" % (t, point)] = np.asarray(W, dtype=np.float64)
                d["%s/p%d/meanEnergy" % (t, point)] = np.array(exp.exp_dict["meanEnergy"])
            d["%s/bins_after" % t] = np.array(exp.myDetector.det_param["myBinsThersholds"], dtype=np.float64)
    save("experiment.npz", d)


def gold_darkfield():
    """Dark-field branch: AnalyticalSample.setWaveRT with a "Lung" material (Sample.py:322-344), fastRefractionDF
    (refractionFileNumba2.py:88-196) and the RT chain that routes through them (Experiment.py:469-473, 490-492)."""
    d = {}
    Nx, Ny = 48, 40
    pix_um = 6.0 / 2 / (145.2 / 141.6)
    M = 145.2 / 141.6
    E = 52.0
    I, phi, T = refraction_inputs(Nx, Ny, pix_um, 21)
    # dark-field angle map (rad): a cylinder-shaped region, zero elsewhere
    yy = np.arange(Ny, dtype=np.float64) - Ny / 2 + 0.5
    prof = np.sqrt(np.clip(12.0 ** 2 - yy ** 2, 0, None)) / 12.0
    for k, (z, amp) in enumerate([(3.6, 2.5e-6), (1.6, 0.6e-6), (3.6, 0.0)]):
        df = np.broadcast_to(amp * prof[None, :], (Nx, Ny)).copy()
        Iin = I.copy()
        out, Dx, Dy = RF2.fastRefractionDF(Iin, phi.copy(), z, E, M, pix_um, df.copy())
        d["rf/%d/params" % k] = np.array([z, E, M, pix_um])
        d["rf/%d/df" % k] = df
        d["rf/%d/out" % k] = out
        d["rf/%d/Dx" % k] = Dx
        d["rf/%d/Dy" % k] = Dy
        print("fastRefractionDF case", k, "max DF px", (df * z / (pix_um * 1e-6 * M)).max(), "Dx shape", Dx.shape)
    d["rf/I"] = I
    d["rf/phi"] = phi
    d["rf/n"] = np.array(3)
    # Lung sample through setWaveRT
    geom = np.stack([synth.cylinder_sample(Nx, Ny, pix_um * 1e-6).astype(np.float64), T])
    s = make_sample("lungs", "sample_of_interest", ["Lung", "PMMA"], geom, [E], [[3.1e-7], [9.87e-8]], [[1.6e-10], [4.5e-11]])
    I1, phi1, df1 = s.setWaveRT(I.copy(), E, phi.copy())
    d["lung/geometry"] = geom
    d["lung/I"] = I1
    d["lung/phi"] = phi1
    d["lung/df"] = np.asarray(df1, dtype=np.float64)
    s2 = make_sample("cylinder_beeds", "sample_of_interest", ["PMMA"], geom[:1], [E], [[9.87e-8]], [[4.5e-11]])
    I2, phi2, df2 = s2.setWaveRT(I.copy(), E, phi.copy())
    d["beeds/I"] = I2
    d["beeds/phi"] = phi2
    d["beeds/df"] = np.asarray(df2, dtype=np.float64)
    # full RT chain with a Lung sample (mono-energetic, points 0 and 1)
    exp, membrane_geom = build_experiment((24, 20), 2, [(52.0, 1)], [], 1, 0.0, 10.0, True, False, 60)
    Ns = exp.exp_dict["studyDimensions"]
    lung = synth.cylinder_sample(Ns[0], Ns[1], exp.exp_dict["studyPixelSize"] * 1e-6, radius_frac=0.3).astype(np.float64)[None]
    exp.mySampleofInterest = make_sample("lungs", "sample_of_interest", ["Lung"], lung, [52.0], [[3.1e-7]], [[1.6e-10]])
    record_inputs(d, "chain", exp)
    for point in (0, 1):
        exp.myMembrane.myGeometry = membrane_geom(point)
        d["chain/p%d/membrane" % point] = exp.myMembrane.myGeometry
        exp.exp_dict["meanEnergy"] = 0
        S, R, Pg, W, Dx, Dy, DF = exp.computeSampleAndReferenceImages_RT(point)
        for nm, a in (("Sample", S), ("Reference", R), ("Propag", Pg), ("White", W), ("Dx", Dx), ("Dy", Dy), ("DF", DF)):
            d["chain/p%d/%s" % (point, nm)] = np.asarray(a, dtype=np.float64)
    save("darkfield.npz", d)


def gold_membrane():
    """getMembraneSegmentedFromFile (Samples/getMembraneFromFile.py:60-171) on a synthetic sphere list written as
    Samples/Membranes/CuSn.txt in a scratch directory (the real file is not distributed); numpy's global generator is
    seeded so that the layer offsets (getMembraneFromFile.py:139-140) are reproducible."""
    import json
    import tempfile
    import Samples.getMembraneFromFile as GM
    d = {}
    cases = [  # tag, dimX, dimY, pixSize_um, meanRadius_um, layers, support_um, n_spheres, seed
        ("plain", 72, 64, 3.0, 12.0, 2, 6000.0, None, 7),
        ("stitch", 60, 52, 40.0, 60.0, 1, 4000.0, 3000, 8),
    ]
    here = os.getcwd()
    for tag, dimX, dimY, pix, meanR, layers, support, nmax, seed in cases:
        lst = synth.sphere_list(n_max=nmax)
        with tempfile.TemporaryDirectory() as tmp:
            os.makedirs(os.path.join(tmp, "Samples", "Membranes"))
            with open(os.path.join(tmp, "Samples", "Membranes", "CuSn.txt"), "w") as f:
                json.dump(lst.tolist(), f)
            os.chdir(tmp)
            try:
                smp = types.SimpleNamespace(myMeanSphereRadius=meanR, myNbOfLayers=layers)
                np.random.seed(seed)
                geom, params = GM.getMembraneSegmentedFromFile(smp, dimX, dimY, pix, 0, support)
            finally:
                os.chdir(here)
        d[tag + "/params"] = np.array([dimX, dimY, pix, meanR, layers, support, -1 if nmax is None else nmax, seed], dtype=np.float64)
        d[tag + "/membrane"] = np.asarray(geom[0], dtype=np.float64)
        d[tag + "/support"] = np.asarray(geom[1], dtype=np.float64)
        print("membrane", tag, "max thickness", d[tag + "/membrane"].max(), "coverage", (d[tag + "/membrane"] > 0).mean())
    save("membrane.npz", d)


# ---------------------------------------------------------------------------------------------------------
# Polychromatic front-end: the reference's table walks and spectrum resampling, run on SYNTHETIC
# tables.  xlrd and spekpy are absent, so the two readers are replaced by in-memory stand-ins that only SERVE data
# (sheet.cell(r, c).value / Spek.get_spectrum()); every line of parsing, interpolation, thresholding and resampling
# that is recorded below is the reference's own (Sample.py:83-152, Detector.py:131-158, Source.py:79-240).
class _Cell:
    def __init__(self, v):
        self.value = v


class _Sheet:
    def __init__(self, grid):
        self.grid = grid
        self.nrows = len(grid)
        self.ncols = max(len(r) for r in grid)

    def cell(self, r, c):
        row = self.grid[r]
        return _Cell(row[c] if c < len(row) else "")


class _Workbook:
    def __init__(self, sheets):
        self._sheets = sheets

    def sheets(self):
        return self._sheets


def _fake_xlrd(books):
    m = types.ModuleType("xlrd")
    m.open_workbook = lambda path: _Workbook([_Sheet(g) for g in books[path]])
    return m


def _table_sheet(tables):
    """TablesDeltaBeta.xls layout: per material 3 columns (E_eV, delta, beta), name in row 0, data from row 3."""
    nrows = 3 + max(len(t[1]) for t in tables)
    grid = [["" for _ in range(3 * len(tables))] for _ in range(nrows)]
    for m, (name, E, dl, bt) in enumerate(tables):
        grid[0][3 * m] = name
        grid[1][3 * m], grid[1][3 * m + 1], grid[1][3 * m + 2] = "Energy", "delta", "beta"
        grid[2][3 * m] = "eV"
        for r in range(len(E)):
            grid[3 + r][3 * m], grid[3 + r][3 * m + 1], grid[3 + r][3 * m + 2] = float(E[r]), float(dl[r]), float(bt[r])
    return grid


def gold_frontend():
    d = {}
    # ---- delta/beta tables: two materials on different (irregular) energy grids
    E1 = np.concatenate([np.arange(8000.0, 30000.0, 1500.0), np.arange(30000.0, 130001.0, 5000.0)])
    E2 = np.geomspace(12000.0, 150000.0, 37)
    tabs = [("SynthNylon", E1, 2.4e-7 * (25000.0 / E1) ** 2, 9e-11 * (25000.0 / E1) ** 3.1),
            ("SynthGadox", E2, 1.9e-6 * (25000.0 / E2) ** 2, 6e-8 * (25000.0 / E2) ** 2.7)]
    for m, (name, E, dl, bt) in enumerate(tabs):
        d["tab/%d/E_eV" % m], d["tab/%d/delta" % m], d["tab/%d/beta" % m] = E, dl, bt
    d["tab/names"] = np.array([t[0] for t in tabs])
    books = {"Samples/DeltaBeta/TablesDeltaBeta.xls": [_table_sheet(tabs)]}
    spectrum = [(7.5, 0.02), (9.0, 0.05), (12.0, 0.1), (17.3, 0.2), (25.0, 0.25), (31.0, 0.18), (52.0, 0.12), (88.8, 0.06),
                (129.9, 0.02)]
    d["spectrum"] = np.array(spectrum)
    SAM.xlrd = _fake_xlrd(books)
    smp = object.__new__(SAM.AnalyticalSample)
    smp.myMaterials = ["SynthNylon", "SynthGadox"]
    smp.delta, smp.beta = [], []
    smp.getDeltaBeta(spectrum)                                    # Sample.py:83-152, table branch
    d["sample/delta"] = np.array(smp.delta)                       # [nmat][nE][2]
    d["sample/beta"] = np.array(smp.beta)
    DET.xlrd = _fake_xlrd(books)
    det = object.__new__(DET.Detector)
    det.det_param = {"myScintillatorMaterial": "SynthGadox", "myScintillatorThickness": 150.0}
    det.beta, det.mySpectralEfficiency = [], []
    det.getBeta(spectrum)                                         # Detector.py:131-158
    det.getSpectralEfficiency()                                   # Detector.py:161-170
    d["det/beta"] = np.array(det.beta)
    d["det/efficiency"] = np.array(det.mySpectralEfficiency)

    # ---- tube spectrum through the spekpy branch (Source.py:98-130): NaN bins, 1e-4 threshold
    Es = np.arange(10.0, 80.5, 0.5)
    flu = np.maximum(80.0 / Es - 1.0, 0.0) * np.exp(-((12.0 / Es) ** 3)) * 1e6
    flu[[3, 40]] = np.nan
    flu[100:] *= 1e-4

    class _Spek:
        def __init__(self, kvp, th, targ, dk):
            d["spek/args"] = np.array([kvp, th, dk])

        def filter(self, mat, thick):
            d["spek/filter_thickness"] = np.array(thick)

        def get_spectrum(self, flu=True):
            return [Es.copy(), globals_flu.copy()]

    globals_flu = flu
    fake_sp = types.ModuleType("spekpy")
    fake_sp.Spek = _Spek
    SRC.sp = fake_sp
    src = object.__new__(SRC.Source)
    src.mySpectrum = []
    src.spectrumFromXls = False
    src.source_dict = {"myType": "Polychromatic", "myVoltage": 80.0, "myEnergySampling": 0.5, "filterMaterial": "Al",
                       "filterThickness": 1.5}
    src.setMySpectrum()
    d["spek/E"], d["spek/fluence"] = Es, flu
    d["spek/out"] = np.array(src.mySpectrum)

    # ---- tabulated spectrum through the xls branch (Source.py:132-233): unit scaling, re-binning, 1e-3 threshold
    for case, (unit, E0, step, n, sampling) in enumerate([("keV", 10.0, 0.5, 141, 2.0), ("eV", 8000.0, 250.0, 200, 1.0),
                                                          ("keV", 15.0, 1.0, 60, 5.0)]):
        Ex = E0 + step * np.arange(n)
        scale = {"keV": 1.0, "eV": 0.001}[unit]
        fx = np.maximum((Ex * scale)[-1] * 1.02 / (Ex * scale) - 1.0, 0.0) * np.exp(-((14.0 / (Ex * scale)) ** 3)) * 3e5
        grid = [["comment", "", ""], ["E", "N", "other"]] + [[float(a), float(b), 0.0] for a, b in zip(Ex, fx)]
        SRC.xlrd = _fake_xlrd({"spectrum.xls": [grid]})
        src = object.__new__(SRC.Source)
        src.mySpectrum = []
        src.spectrumFromXls = True
        src.source_dict = {"myType": "Polychromatic", "myEnergySampling": sampling, "energyUnit": unit,
                           "pathXlsSpectrum": "spectrum.xls", "energyColumnKey": "E", "fluenceColumnKey": "N",
                           "filterMaterial": None}
        src.setMySpectrum()
        d["xls/%d/E" % case], d["xls/%d/fluence" % case] = Ex, fx
        d["xls/%d/unit_is_eV" % case] = np.array(unit == "eV")
        d["xls/%d/sampling" % case] = np.array(sampling)
        d["xls/%d/out" % case] = np.array(src.mySpectrum)
    d["xls/n"] = np.array(3)
    save("frontend.npz", d)


def gold_frontend_chain():
    """Both chains with everything the polychromatic front-end produces going through the reference's own code: the
    spectrum from the tabulated branch of Source.setMySpectrum, the sample's delta/beta and the scintillator's beta from
    the table walks, the scintillator efficiency (Experiment.py:326-332 / :456-459), air, plate, two energy bins."""
    g = np.load(os.path.join(HERE, "frontend.npz"))
    tabs = [(str(n), g["tab/%d/E_eV" % m], g["tab/%d/delta" % m], g["tab/%d/beta" % m]) for m, n in enumerate(g["tab/names"])]
    books = {"Samples/DeltaBeta/TablesDeltaBeta.xls": [_table_sheet(tabs)]}
    Ex = 15.0 + np.arange(60)
    fx = np.maximum(76.0 / Ex - 1.0, 0.0) * np.exp(-((20.0 / Ex) ** 3)) * 3e5
    grid = [["E", "N"]] + [[float(a), float(b)] for a, b in zip(Ex, fx)]
    books["spectrum.xls"] = [grid]
    SRC.xlrd = SAM.xlrd = DET.xlrd = _fake_xlrd(books)
    src = object.__new__(SRC.Source)
    src.mySpectrum = []
    src.spectrumFromXls = True
    src.source_dict = {"myType": "Polychromatic", "myEnergySampling": 10.0, "energyUnit": "keV",
                       "pathXlsSpectrum": "spectrum.xls", "energyColumnKey": "E", "fluenceColumnKey": "N",
                       "filterMaterial": None, "mySize": 30.0}
    src.setMySpectrum()
    spectrum = list(src.mySpectrum)
    d = {"xls/E": Ex, "xls/fluence": fx, "xls/sampling": np.array(10.0), "scint/thickness_um": np.array(120.0)}
    for sim in ("RT", "Fresnel"):
        exp, membrane_geom = build_experiment((24, 28), 2, spectrum, [40.0], 10.0, 1.0, 30.0, False, True, 70)
        smp = exp.mySampleofInterest
        smp.myMaterials, smp.delta, smp.beta = ["SynthNylon"], [], []
        smp.getDeltaBeta(spectrum)                                       # table walk, Sample.py:112-148
        det = exp.myDetector
        det.det_param["myScintillatorMaterial"] = "SynthGadox"
        det.det_param["myScintillatorThickness"] = 120.0
        det.getBeta(spectrum)                                            # Detector.py:131-158
        det.getSpectralEfficiency()
        t = "chain/%s" % sim
        record_inputs(d, t, exp)
        d[t + "/scint_beta"] = np.array(det.beta)
        for point in (0, 1):
            exp.myMembrane.myGeometry = membrane_geom(point)
            d["%s/p%d/membrane" % (t, point)] = exp.myMembrane.myGeometry
            exp.exp_dict["meanEnergy"] = 0
            if sim == "RT":
                S, R, Pg, W, Dx, Dy, DF = exp.computeSampleAndReferenceImages_RT(point)
            else:
                S, R, Pg, W = exp.computeSampleAndReferenceImages_Fresnel(point)
            for nm, a in (("Sample", S), ("Reference", R), ("Propag", Pg), ("White", W)):
                d["%s/p%d/%s" % (t, point, nm)] = np.asarray(a, dtype=np.float64)
            d["%s/p%d/meanEnergy" % (t, point)] = np.array(exp.exp_dict["meanEnergy"])
    save("frontend_chain.npz", d)


def gold_geometry():
    """Analytic sample generators of Samples/createSampGeom.py (the deterministic ones that need no imutils):
    CreateSampleSphere (:15-53, radius from xmlFiles/Samples.xml), CreateSampleSpheresInCylinder (:108-171) and
    CreateYourSampleGeometry (:289-318)."""
    from Samples import createSampGeom as CSG
    d = {}
    geom, par = CSG.CreateSampleSphere("PMMA_sphere", 96, 130, 25.0)       # radius 1000 um = 40 px
    d["sphere/args"] = np.array([96, 130, 25.0])
    d["sphere/geom"] = geom
    d["sphere/radius_um"] = np.array(par["Sphere_radius"][0])
    geom, par = CSG.CreateSampleSpheresInCylinder("spheres_in_cylinder", 210, 120, 20.0)
    d["sic/args"] = np.array([210, 120, 20.0])
    d["sic/geom"] = geom
    d["sic/params"] = np.array([par["Spheres_radius"][0], par["Cylinder_radius"][0], par["Position_Sphere_1"][0],
                                par["Position_Sphere_2"][0]])
    geom, par = CSG.CreateSampleSpheresInCylinder("spheres_in_cylinder", 561, 333, 7.7)    # odd sizes, fractional radius
    d["sic2/args"] = np.array([561, 333, 7.7])
    d["sic2/geom"] = geom.astype(np.float32)
    geom, par = CSG.CreateYourSampleGeometry("x", 33, 47, 1.0)
    d["your/args"] = np.array([33, 47, 1.0])
    d["your/geom"] = geom
    d["your/thickness"] = np.array(par["geometry thickness"][0])
    save("geometry.npz", d)


if __name__ == "__main__":
    if len(sys.argv) > 1:                    # regenerate only the named groups:  make_golden.py geometry membrane
        for g in sys.argv[1:]:
            globals()["gold_" + g]()
        os.chdir(_cwd)
        sys.exit(0)
    gold_scalars()
    gold_transmission()
    gold_fresnel()
    gold_refraction()
    gold_detector()
    gold_experiment()
    gold_membrane()
    gold_darkfield()
    gold_frontend()
    gold_frontend_chain()
    gold_geometry()
    os.chdir(_cwd)
