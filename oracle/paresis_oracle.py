"""CPU oracle for the PARESIS hot path -- TEST INFRASTRUCTURE, not product code.

A float64 restatement (numpy + scipy.signal, plus two scalar C loops in oracle_loops.c for what the reference
JIT-compiles with Numba) of the per-energy, per-membrane-position image formation of quenotl/PARESIS.  Every function
cites the reference lines it follows (paths relative to /root/reference/CodePython).  The oracle is PINNED: it is
checked against golden vectors produced by the reference itself (tests/golden/make_golden.py ->
tests/golden/*.npz; tests/test_oracle_golden.py).

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import this module.  The product
(paresis_amd) never does, and fails loudly when its HIP library is missing.
"""
import ctypes
import os

import numpy as np
from numpy.fft import fft2, fftshift, ifft2, ifftshift
from scipy.signal import fftconvolve

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle_loops.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle C loops not built: run `make -C oracle` (or __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        dp = ctypes.POINTER(ctypes.c_double)
        lib.oracle_fastloop.argtypes = [ctypes.c_int64, ctypes.c_int64, dp, dp, dp, dp]
        lib.oracle_fastloop.restype = None
        lib.oracle_resize.argtypes = [ctypes.c_int64, ctypes.c_int64, dp, ctypes.c_int64, ctypes.c_int64, dp]
        lib.oracle_resize.restype = None
        lib.oracle_membrane_splat.argtypes = [ctypes.c_int64, dp, dp, dp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                              ctypes.c_int64, dp]
        lib.oracle_membrane_splat.restype = None
        _LIB = lib
    return _LIB


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


# ------------------------------------------------------------------------------------------------- scalars
H_PLANCK = 6.626e-34   # getk.py:16  (deliberately not CODATA: the reference's constants are the spec)
C_LIGHT = 2.998e8      # getk.py:17
E_CHARGE = 1.6e-19     # getk.py:18


def getk(energy_eV):
    """getk.py:12-20."""
    return 2 * np.pi * energy_eV * E_CHARGE / (H_PLANCK * C_LIGHT)


def k_sample(energy_keV):
    """Sample.py:265 / Sample.py:300 (same constants, different association order)."""
    return 2 * np.pi * energy_keV * 1000 * 1.6e-19 / (6.626e-34 * 2.998e8)


def k_refraction(energy_keV):
    """refractionFileNumba2.py:47-48."""
    lam = 6.626 * 1e-34 * 2.998e8 / (energy_keV * 1000 * 1.6e-19)
    return 2 * np.pi / lam


# -------------------------------------------------------------------------------------------- transmission
def set_wave(wave, geometry, delta, beta, energy_keV):
    """AnalyticalSample.setWave, Sample.py:248-282.  geometry [nmat,Nx,Ny] (m); delta/beta per material."""
    geometry = np.asarray(geometry)
    if geometry.ndim != 3:
        raise Exception("Sample Geometry has the wrong nb of dim [material, x, y]")
    k = k_sample(energy_keV)
    out = wave
    for m in range(geometry.shape[0]):
        out = np.exp((-1j * k * delta[m] - k * beta[m]) * geometry[m]) * out      # SAM:279
    return out


def set_wave_rt(intensity, geometry, delta, beta, energy_keV, phi=0, materials=None, my_type=None, name=None):
    """AnalyticalSample.setWaveRT, Sample.py:285-351.  `materials`, `my_type`, `name` switch on the dark-field sample
    model (Sample.py:322-344): a "Lung" material of a sample_of_interest, or any material of the sample named
    'cylinder_beeds', scatters: newDf = 2 delta sqrt(N_s) sqrt(ln(2/delta)+1) and the thickness is scaled by the
    sphere volume fraction.  newDf is overwritten (not accumulated) per material, like the reference."""
    k = k_sample(energy_keV)
    I = intensity
    new_df = 0
    geometry = np.asarray(geometry)
    for m in range(geometry.shape[0]):
        geom = geometry[m]
        if my_type == "sample_of_interest":
            for hit, radius, fraction in ((materials is not None and materials[m] == "Lung", 47, 0.5),       # SAM:324-333
                                          (name == "cylinder_beeds", 15, 0.6)):                               # SAM:335-343
                if hit:
                    n_vol = fraction * 3 / 4 / np.pi / (radius ** 3)
                    n_sphere = n_vol ** (1 / 3) * (geometry[m] * 1e6)
                    new_df = 2 * delta[m] * (n_sphere) ** (1 / 2) * np.sqrt(np.log(2 / delta[m]) + 1)
                    geom = geom * fraction
        I = np.exp(-2 * k * beta[m] * geom) * I               # SAM:347
        phi = phi - k * delta[m] * geom                       # SAM:348
    return I, phi, new_df


# ------------------------------------------------------------------------------------------------- Fresnel
def wave_propagation(wave, z, energy_keV, magnification, study_dims, pix_um):
    """Experiment.wavePropagation, Experiment.py:219-252."""
    if z == 0:
        return wave                                            # EXP:233-234
    margin = 15
    w = np.pad(wave, margin, mode="reflect")                   # EXP:237
    k = getk(energy_keV * 1000)                                # EXP:239
    Nx, Ny = w.shape
    # EXP:243-248: uv_sqr[i, j] = u_m(i)^2 + v_m(j)^2, frequency step from the UN-padded study dimensions
    u_m = (np.arange(Nx) - (Nx // 2)) * 2 * np.pi / (study_dims[0] * pix_um * 1e-6)
    v_m = (np.arange(Ny) - (Ny // 2)) * 2 * np.pi / (study_dims[1] * pix_um * 1e-6)
    uv_sqr = u_m[:, None] ** 2 + v_m[None, :] ** 2
    out = np.exp(1j * k * z / magnification) * ifft2(
        ifftshift(np.exp(-1j * z * uv_sqr / (2 * k * magnification)) * fftshift(fft2(w))))   # EXP:250
    return out[margin:Nx - margin, margin:Ny - margin]          # EXP:251


# ---------------------------------------------------------------------------------------------- refraction
def fastloop(I, Dx, Dy, I2=None):
    """fastloopNumba, refractionFileNumba2.py:198-263 (C loop).  Accumulates into and returns I2."""
    I = np.ascontiguousarray(I, dtype=np.float64)
    Dx = np.ascontiguousarray(Dx, dtype=np.float64)
    Dy = np.ascontiguousarray(Dy, dtype=np.float64)
    if I2 is None:
        I2 = np.zeros_like(I)
    assert I2.flags.c_contiguous and I2.dtype == np.float64 and I.shape == Dx.shape == Dy.shape == I2.shape
    _lib().oracle_fastloop(I.shape[0], I.shape[1], _dp(I), _dp(I2), _dp(Dx), _dp(Dy))
    return I2


def fast_refraction(intensity, phi, z, energy_keV, magnification, pix_um, variant="v2"):
    """fastRefraction: refractionFileNumba2.py:25-86 (variant "v2", the one Experiment.py:22 imports) or
    refractionFileNumba.py:11-68 (variant "v1": margin 10, clamp |D|>1e3).

    Mutates `intensity` in place exactly like the reference (RF2:61-62).  Returns (I2[N,N], Dx[P,P], Dy[P,P]).
    """
    k = k_refraction(energy_keV)
    Nx, Ny = intensity.shape
    if variant == "v2":
        margin2, limx, limy = 15, Nx, Ny                       # RF2:50, 61-64
    elif variant == "v1":
        margin2, limx, limy = 10, 1e3, 1e3                     # RF1:36, 46-49
    else:
        raise ValueError(variant)
    h = pix_um * 1e-6
    dphix, dphiy = np.gradient(phi, h, edge_order=2)           # RF2:54
    Dx = dphix * z / k / (h * magnification)                   # RF2:55
    Dy = dphiy * z / k / (h * magnification)                   # RF2:56
    Dx[abs(Dx) < 1e-12] = 0                                    # RF2:59-60
    Dy[abs(Dy) < 1e-12] = 0
    intensity[abs(Dx) > limx] = 0                              # RF2:61-62 (in place!)
    intensity[abs(Dy) > limy] = 0
    Dx[abs(Dx) > limx] = 0                                     # RF2:63-64
    Dy[abs(Dy) > limy] = 0
    Dx = np.pad(Dx, margin2, mode="constant")                  # RF2:65-67
    Dy = np.pad(Dy, margin2, mode="constant")
    Ipad = np.pad(intensity, margin2, mode="constant")
    I2 = fastloop(Ipad, Dx, Dy)                                # RF2:70-77
    I2 = I2[margin2:Nx + margin2, margin2:Ny + margin2]        # RF2:78
    if np.isnan(I2).any() or np.any(abs(I2) > 1e50):           # RF2:81-82
        raise Exception("The calculated intensity refractive includes some nans or insane values")
    return I2, Dx, Dy


def fast_refraction_df(intensity, phi, z, energy_keV, magnification, pix_um, dark_field):
    """fastRefractionDF, refractionFileNumba2.py:88-196 (without its matplotlib pop-ups).  Mutates `intensity` and
    `dark_field` in place like the reference.  Returns (I3[N,N], Dx, Dy) with Dx, Dy padded by ceil(6*max DF)."""
    k = k_refraction(energy_keV)
    Nx, Ny = intensity.shape
    h = pix_um * 1e-6
    dark_field = dark_field * z / (h * magnification)                          # RF2:114 (rad -> pixels)
    max_df = np.max(dark_field)
    margin2 = int(np.ceil(max_df * 6))                                         # RF2:117
    dphix, dphiy = np.gradient(phi, h, edge_order=2)
    Dx = dphix * z / k / (h * magnification)
    Dy = dphiy * z / k / (h * magnification)
    Dx[abs(Dx) < 1e-12] = 0
    Dy[abs(Dy) < 1e-12] = 0
    intensity[abs(Dx) > Nx] = 0
    intensity[abs(Dy) > Ny] = 0
    Dx[abs(Dx) > Nx] = 0
    Dy[abs(Dy) > Ny] = 0
    Dx = np.pad(Dx, margin2, mode="constant")
    Dy = np.pad(Dy, margin2, mode="constant")
    dark_field[dark_field > Nx / 4] = 0                                        # RF2:135
    dark_field = np.pad(dark_field, margin2, mode="constant")
    Ipad = np.pad(intensity, margin2, mode="constant")
    I_nodf = np.copy(Ipad); I_nodf[dark_field != 0] = 0                        # RF2:147-150
    I_df = np.copy(Ipad); I_df[dark_field == 0] = 0
    I2 = fastloop(I_nodf, Dx, Dy)
    I2df = fastloop(I_df, Dx, Dy)
    I3 = np.zeros_like(I2)
    for i in range(margin2, Nx + margin2):                                     # RF2:171-184
        for j in range(margin2, Ny + margin2):
            if I2df[i, j] != 0:
                if dark_field[i, j] != 0:
                    patch = create_gaussian_shape(dark_field[i, j] / 2)
                    s2 = patch.shape[0] // 2
                    I3[i - s2:i + s2 + 1, j - s2:j + s2 + 1] += patch * I2df[i, j]
                else:
                    I3[i, j] += I2df[i, j]
    I3 += I2
    I3 = I3[margin2:Nx + margin2, margin2:Ny + margin2]
    if np.any(abs(I3) > 1e50) or np.isnan(I3).any():
        raise Exception("The calculated intensity refractive includes some nans or insane values")
    return I3, Dx, Dy


# ------------------------------------------------------------------------------------------------ detector
def py_round(x):
    """Python 3 round(): round-half-to-even, as used at Detector.py:212 / refractionFileNumba2.py:15."""
    return int(round(float(x)))


def create_gaussian_shape(sigma):
    """create_gaussian_shape, Detector.py:201-220 (== gaussian_shape, refractionFileNumba2.py:14-23)."""
    dim = py_round(sigma * 3) * 2 + 1
    q = np.arange(0, dim) - np.floor(dim / 2)
    Qx, Qy = np.meshgrid(q, q)
    g = np.exp(-((Qx ** 2) / 2. / sigma ** 2 + (Qy ** 2) / 2. / sigma ** 2))
    return g / np.sum(g)


def resize(img, sizeX, sizeY):
    """resize, Detector.py:185-198 (block SUM, same factor on both axes)."""
    Nx, Ny = img.shape
    if Nx == sizeX and Ny == sizeY:
        return img
    img = np.ascontiguousarray(img, dtype=np.float64)
    out = np.empty((sizeX, sizeY))
    _lib().oracle_resize(Nx, Ny, _dp(img), sizeX, sizeY, _dp(out))
    return out


def detection(img, eff_source_fwhm_px, over_sampling, det_dims, psf_sigma_px):
    """Detector.detection, Detector.py:79-119, WITHOUT the time-seeded Poisson draw (DET:113-115): returns the
    float64 expectation image that the reference feeds to RandomState.poisson."""
    margins = 15
    x = np.pad(img, margins * over_sampling, mode="reflect")                   # DET:93
    if eff_source_fwhm_px != 0:                                                # DET:96-99
        x = fftconvolve(x, create_gaussian_shape(eff_source_fwhm_px / 2.355), mode="same")
    x = resize(x, det_dims[0] + margins * 2, det_dims[1] + margins * 2)        # DET:103
    if psf_sigma_px != 0:                                                      # DET:106-110
        x = fftconvolve(x, create_gaussian_shape(psf_sigma_px), mode="same")
    return x[margins:det_dims[0] + margins, margins:det_dims[1] + margins]      # DET:118


# ------------------------------------------------------------------------------------------ full chains
class Obj:
    """A thickness stack with its per-energy index decrements: geometry [nmat,Nx,Ny] (m), delta/beta [nmat][nE]."""

    def __init__(self, geometry, delta, beta, materials=None, my_type=None, name=None):
        self.geometry = np.asarray(geometry, dtype=np.float64)
        self.delta = np.asarray(delta, dtype=np.float64)
        self.beta = np.asarray(beta, dtype=np.float64)
        self.materials, self.my_type, self.name = materials, my_type, name


def _bins(cfg, point):
    # EXP:296-301 / EXP:425-430: at point 0 the last spectrum energy is appended as the closing threshold
    thr = list(cfg["bins"])
    spec = cfg["spectrum"]
    if point == 0:
        if any(t < spec[0][0] for t in thr) or any(t > spec[-1][0] for t in thr):
            raise Exception("At least one of your detector bin threshold is outside your source spectrum.")
        thr.append(spec[-1][0])
        cfg["bins"] = thr
    return thr


def _eff_source(cfg):
    # EXP:380 / EXP:503
    return cfg["source_size_um"] * cfg["dOD"] / (cfg["dSM"] + cfg["dMO"]) / cfg["det_pix_um"] * cfg["ov"]


def compute_fresnel(cfg, point):
    """Experiment.computeSampleAndReferenceImages_Fresnel, Experiment.py:279-405.

    cfg keys: dSM,dMO,dOD,meanShotCount,ov,pix_um,M,inVacuum,N(2),spectrum[(E,w)],source_size_um,energy_sampling,
    det_dims(2),det_pix_um,psf,bins(list, mutated at point 0 like the reference),membrane/sample/air/plate (Obj|None),
    optional scintillator = (thickness_um, [(E, beta)]).
    """
    thr = _bins(cfg, point)
    nb = len(thr)
    n0, n1 = cfg["det_dims"]
    N0, N1 = cfg["N"]
    S = np.zeros((nb, n0, n1)); R = np.zeros((nb, n0, n1)); Pg = np.zeros((nb, n0, n1)); W = np.zeros((nb, n0, n1))
    I0 = np.ones((N0, N1)) * (cfg["meanShotCount"] / cfg["ov"] ** 2)            # EXP:308
    accS = np.zeros((N0, N1)); accR = np.zeros((N0, N1)); accP = np.zeros((N0, N1)); white = np.zeros((N0, N1))
    sumI = 0.0; meanE = 0.0; ibin = 0
    mem, smp, air, plate = cfg["membrane"], cfg["sample"], cfg["air"], cfg["plate"]
    prop = lambda w, z, E, M: wave_propagation(w, z, E, M, (N0, N1), cfg["pix_um"])
    for ie, (E, flux) in enumerate(cfg["spectrum"]):
        I = I0 * flux                                                          # EXP:320
        if not cfg["inVacuum"]:
            I, _, _ = set_wave_rt(I, air.geometry, air.delta[:, ie], air.beta[:, ie], E)     # EXP:323
        if cfg.get("scintillator") is not None:                                # EXP:326-332: beta matched by energy equality
            thick, betas = cfg["scintillator"]
            for e_data, b_en in betas:
                if e_data == E:
                    sc_beta = b_en
            I = I * (1 - np.exp(-2 * getk(E * 1000) * thick * 1e-6 * sc_beta))
        w0 = np.sqrt(I)                                                        # EXP:334
        wm = set_wave(w0, mem.geometry, mem.delta[:, ie], mem.beta[:, ie], E)  # EXP:338
        magMemObj = (cfg["dSM"] + cfg["dMO"]) / cfg["dSM"]                     # EXP:340
        wbs = prop(wm, cfg["dMO"], E, magMemObj)                               # EXP:341
        was = set_wave(wbs, smp.geometry, smp.delta[:, ie], smp.beta[:, ie], E)  # EXP:344
        wS = prop(was, cfg["dOD"], E, cfg["M"])                                # EXP:348
        wR = prop(wm, cfg["dOD"] + cfg["dMO"], E, cfg["M"])                    # EXP:349
        IS = abs(wS) ** 2; IR = abs(wR) ** 2                                   # EXP:351,354
        if plate is not None:
            IS, _, _ = set_wave_rt(IS, plate.geometry, plate.delta[:, ie], plate.beta[:, ie], E)
            IR, _, _ = set_wave_rt(IR, plate.geometry, plate.delta[:, ie], plate.beta[:, ie], E)
        accS += IS; accR += IR                                                 # EXP:357-358
        sumI += np.mean(IR); meanE += E * np.mean(IR)                          # EXP:360-361
        if point == 0:                                                         # EXP:363-375
            wp = prop(set_wave(w0, smp.geometry, smp.delta[:, ie], smp.beta[:, ie], E), cfg["dOD"], E, cfg["M"])
            IP = abs(wp) ** 2
            Iw = w0 ** 2
            if plate is not None:
                IP, _, _ = set_wave_rt(IP, plate.geometry, plate.delta[:, ie], plate.beta[:, ie], E)
                Iw, _, _ = set_wave_rt(w0 ** 2, plate.geometry, plate.delta[:, ie], plate.beta[:, ie], E)
            accP += IP; white = white + Iw
        if E > thr[ibin] - cfg["energy_sampling"] / 2:                         # EXP:378
            ess = _eff_source(cfg)
            det = lambda im: detection(im, ess, cfg["ov"], cfg["det_dims"], cfg["psf"])
            S[ibin] = det(accS); R[ibin] = det(accR)                           # EXP:388-390
            if point == 0:
                Pg[ibin] = det(accP)
            W[ibin] = det(white)                                               # EXP:394
            accS = np.zeros((N0, N1)); accR = np.zeros((N0, N1)); accP = np.zeros((N0, N1)); white = np.zeros((N0, N1))
            ibin += 1
    return S, R, Pg, W, meanE / sumI


def compute_rt(cfg, point, variant="v2"):
    """Experiment.computeSampleAndReferenceImages_RT, Experiment.py:407-526 (scalar dark field only)."""
    thr = _bins(cfg, point)
    nb = len(thr)
    n0, n1 = cfg["det_dims"]
    N0, N1 = cfg["N"]
    S = np.zeros((nb, n0, n1)); R = np.zeros((nb, n0, n1)); Pg = np.zeros((nb, n0, n1)); W = np.zeros((nb, n0, n1))
    I0 = np.ones((N0, N1)) * (cfg["meanShotCount"] / cfg["ov"] ** 2)            # EXP:438
    phi0 = np.zeros((N0, N1))
    accS = np.zeros((N0, N1)); accR = np.zeros((N0, N1)); accP = np.zeros((N0, N1)); white = np.zeros((N0, N1))
    sumI = 0.0; meanE = 0.0; ibin = 0
    Dxreal = []; Dyreal = []
    mem, smp, air, plate = cfg["membrane"], cfg["sample"], cfg["air"], cfg["plate"]
    def refr(I, phi, z, E, df=0):                                              # Experiment.refraction, EXP:255-277
        if type(df) == int or type(df) == float:
            return fast_refraction(abs(I), phi, z, E, cfg["M"], cfg["pix_um"], variant)
        return fast_refraction_df(abs(I), phi, z, E, cfg["M"], cfg["pix_um"], df)
    dfp = np.zeros((N0, N1))
    sm = dict(materials=getattr(smp, "materials", None), my_type=getattr(smp, "my_type", None), name=getattr(smp, "name", None))
    for ie, (E, flux) in enumerate(cfg["spectrum"]):
        I = I0 * flux                                                          # EXP:451
        if not cfg["inVacuum"]:
            I, _, _ = set_wave_rt(I, air.geometry, air.delta[:, ie], air.beta[:, ie], E)
        if cfg.get("scintillator") is not None:                                # EXP:456-459: tabulated efficiency
            for e_data, eff in spectral_efficiency(cfg["scintillator"][1], cfg["scintillator"][0]):
                if e_data == E:
                    I = I * eff
        Im, phim, _ = set_wave_rt(I, mem.geometry, mem.delta[:, ie], mem.beta[:, ie], E, phi0)   # EXP:463
        Ibs, _, _ = refr(Im, phim, cfg["dMO"], E)                              # EXP:466 (total magnification!)
        Ias, phis, DF = set_wave_rt(Ibs, smp.geometry, smp.delta[:, ie], smp.beta[:, ie], E, phim, **sm)  # EXP:469
        IS, _, _ = refr(Ias, phis, cfg["dOD"], E, DF)                          # EXP:473
        IR, _, _ = refr(Ibs, phim, cfg["dOD"], E)                              # EXP:474
        if plate is not None:                                                  # EXP:478-480
            IS, _, _ = set_wave_rt(IS, plate.geometry, plate.delta[:, ie], plate.beta[:, ie], E)
            IR, _, _ = set_wave_rt(IR, plate.geometry, plate.delta[:, ie], plate.beta[:, ie], E)
        accS += IS; accR += IR
        sumI += np.mean(IR); meanE += E * np.mean(IR)
        if point == 0:                                                         # EXP:488-498
            Ip, phip, DFp = set_wave_rt(I, smp.geometry, smp.delta[:, ie], smp.beta[:, ie], E, phi0, **sm)
            dfp = dfp + DFp * flux                                             # EXP:491
            IP, Dxreal, Dyreal = refr(Ip, phip, cfg["dOD"], E, DFp)
            if plate is not None:
                IP, _, _ = set_wave_rt(IP, plate.geometry, plate.delta[:, ie], plate.beta[:, ie], E)
                I, _, _ = set_wave_rt(I, plate.geometry, plate.delta[:, ie], plate.beta[:, ie], E)
            white = white + I
            accP += IP
        if E > thr[ibin] - cfg["energy_sampling"] / 2:                         # EXP:501
            ess = _eff_source(cfg)
            det = lambda im: detection(im, ess, cfg["ov"], cfg["det_dims"], cfg["psf"])
            S[ibin] = det(accS); R[ibin] = det(accR)
            if point == 0:
                Pg[ibin] = det(accP)
            W[ibin] = det(white)
            accS = np.zeros((N0, N1)); accR = np.zeros((N0, N1)); accP = np.zeros((N0, N1)); white = np.zeros((N0, N1))
            ibin += 1
    cfg["_darkFieldPropag"] = dfp
    return S, R, Pg, W, Dxreal, Dyreal, meanE / sumI


# ------------------------------------------------------------------------------------- membrane synthesis
def membrane_sphere_layers(sphere_list, dimX, dimY, pix_um, mean_radius_um, n_layers, rand_state):
    """Host part of getMembraneSegmentedFromFile (Samples/getMembraneFromFile.py:79-142): scale the sphere list to the
    requested mean radius, move the origin to the top-left corner, stitch copies until the list covers the study grid,
    draw one random offset per layer.  Returns (margin, margin2, [(xfloat, yfloat, radFloat) per layer]) in pixels of the
    margin-extended grid.  `rand_state` is a numpy RandomState: the reference uses the unseeded global one (:139-140)."""
    margin = int(np.ceil(10 * mean_radius_um / pix_um))                      # :80
    margin2 = int(np.floor(margin / 2))                                      # :81
    corr = mean_radius_um / 12.8                                             # :85
    size_x = int(np.floor(8102)) * corr + mean_radius_um                     # :86
    size_y = int(np.floor(9740)) * corr + mean_radius_um                     # :87
    par = np.asarray(sphere_list, dtype=np.float64) * corr                   # :93-94
    par[:, 1] += size_x / 2                                                  # :97-98
    par[:, 0] += size_y / 2
    par0, sx0, sy0 = par.copy(), size_x, size_y
    while size_x / pix_um - dimX < 0:                                        # :107-113 stitching along x
        st = par0.copy()
        st[:, 1] += size_x
        par = np.concatenate((par, st), axis=0)
        size_x += sx0
    par0 = par.copy()
    while size_y / pix_um - dimY < 0:                                        # :115-122 stitching along y
        st = par0.copy()
        st[:, 0] += size_y
        par = np.concatenate((par, st), axis=0)
        size_y += sy0
    layers = []
    for _ in range(n_layers):                                                # :135-142
        max_ox = size_x / pix_um - dimX
        max_oy = size_y / pix_um - dimY
        ox = rand_state.randint(margin2, max_ox - margin2)
        oy = rand_state.randint(margin2, max_oy - margin2)
        layers.append((par[:, 1] / pix_um - ox, par[:, 0] / pix_um - oy, par[:, 2] / pix_um))
    return margin, margin2, layers


def membrane_segmented(sphere_list, dimX, dimY, pix_um, mean_radius_um, n_layers, support_um, seed):
    """getMembraneSegmentedFromFile, Samples/getMembraneFromFile.py:60-171, with np.random.seed(seed) semantics.
    Returns [membrane_m, support_m] (float64)."""
    margin, margin2, layers = membrane_sphere_layers(sphere_list, dimX, dimY, pix_um, mean_radius_um, n_layers,
                                                     np.random.RandomState(seed))
    mem = np.zeros((dimX + 2 * margin, dimY + 2 * margin))
    for xf, yf, rad in layers:
        xf = np.ascontiguousarray(xf); yf = np.ascontiguousarray(yf); rad = np.ascontiguousarray(rad)
        _lib().oracle_membrane_splat(len(rad), _dp(xf), _dp(yf), _dp(rad), dimX, dimY, margin, margin2, _dp(mem))
    mem = mem[margin:-margin, margin:-margin]                                # :161
    return [mem * pix_um * 1e-6, np.ones(mem.shape) * support_um * 1e-6]     # :167-169


# ------------------------------------------------------------------------- polychromatic front-end
def table_walk(spectrum, table_E_eV, table_delta, table_beta):
    """The delta/beta table walk of Sample.getDeltaBeta (Sample.py:112-148) and Detector.getBeta (Detector.py:139-158)
    for ONE material column: table rows (E_eV, delta, beta) ascending, spectrum [(E_keV, w)] ascending.  The row
    pointer is NOT reset between energies (:121 before the loop), energies under the current row give (0, 1) (:125-129),
    otherwise linear interpolation between the bracketing rows (:131-144).  Returns ([(E, delta)], [(E, beta)])."""
    row = 0
    delta, beta = [], []
    for energy, _ in spectrum:
        cur = table_E_eV[row]
        if energy * 1000 < cur:
            delta.append((energy, 0))
            beta.append((energy, 1))
            continue
        nxt = table_E_eV[row + 1]
        while nxt < energy * 1e3:
            row += 1
            cur = table_E_eV[row]
            nxt = table_E_eV[row + 1]
        step = nxt - cur
        d = abs(nxt - energy * 1e3) / step * table_delta[row] + abs(cur - energy * 1e3) / step * table_delta[row + 1]
        b = abs(nxt - energy * 1e3) / step * table_beta[row] + abs(cur - energy * 1e3) / step * table_beta[row + 1]
        delta.append((energy, d))
        beta.append((energy, b))
    return delta, beta


def spectral_efficiency(beta, thickness_um):
    """Detector.getSpectralEfficiency, Detector.py:161-170."""
    return [(e, 1 - np.exp(-2 * getk(e * 1000) * thickness_um * 1e-6 * b)) for e, b in beta]


def tube_spectrum(energies, fluence):
    """Source.setMySpectrum, generated-spectrum branch after the spekpy call (Source.py:108-123): NaN -> 0,
    normalise by the total, keep the bins above 1e-4."""
    fluence = np.where(np.isnan(fluence), 0.0, np.asarray(fluence, dtype=np.float64))
    fl = 0
    for v in fluence:
        fl += v
    return [(energies[i], fluence[i] / fl) for i in range(len(energies)) if fluence[i] / fl > 0.0001]


def xls_spectrum(rows_E, rows_fluence, unit_scale, sampling):
    """Source.setMySpectrum, tabulated branch (Source.py:132-233): rows in file units (scaled by unit_scale to keV),
    re-binned to `sampling` keV.  Kept quirks: the bin width counter advances by the step of the FIRST two rows (:194),
    only Nbin-1 full bins are formed and the rest is one tail bin (:200-222), the normalisation total does not include
    the tail bin (:213 vs :215-222), bins at or under 0.001 are dropped (:228)."""
    spectrum = [[e * unit_scale, f] for e, f in zip(rows_E, rows_fluence)]
    den = spectrum[1][0] - spectrum[0][0]
    n_en = len(spectrum)
    n_bin = int((spectrum[-1][0] - spectrum[0][0]) // sampling)
    energyplot, weightplot = [], []
    n = 0
    tot_weight = 0
    for _ in range(n_bin - 1):
        curr = 0
        w = 0
        eb = 0
        while curr < sampling:
            w += spectrum[n][1]
            eb += spectrum[n][1] * spectrum[n][0]
            n += 1
            curr = curr + den
        if w != 0:
            energyplot.append(eb / w)
            weightplot.append(w)
        tot_weight += w
    w = 0
    eb = 0
    while n < n_en:
        w += spectrum[n][1]
        eb += spectrum[n][1] * spectrum[n][0]
        n += 1
    if w != 0:
        energyplot.append(eb / w)
        weightplot.append(w)
    out = []
    for i in range(len(energyplot)):
        flux = weightplot[i] / tot_weight
        if flux > 0.001:
            out.append((energyplot[i], flux))
    return out
