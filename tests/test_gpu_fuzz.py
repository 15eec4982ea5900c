"""Seeded random differential tests: the HIP path (through the C ABI) against the CPU oracle on randomly drawn, mostly
awkward geometries and parameters -- odd and prime sizes, non-square grids, every halo / margin / clamp / accumulate
combination, thickness maps or explicit (I, phi), one call or a distance batch.  The draws are deterministic (seed = case
number); `PSX_FUZZ=k` multiplies the number of cases (the committed default keeps the file under a minute on one MI355X).

Tolerance as everywhere: max|out-ref| / max|ref| <= 1e-5 (fp32 device arithmetic against the fp64 reference restatement)."""
import os

import numpy as np
import pytest
import torch

from oracle import paresis_oracle as orc
from tests._golden import relmax

pytestmark = pytest.mark.gpu

TOL = 1e-5
MULT = max(1, int(os.environ.get("PSX_FUZZ", "1")))


@pytest.fixture(scope="module")
def ops():
    from paresis_amd import ops as _ops
    from paresis_amd._lib import lib
    assert lib().psx_device_ok() == 1, lib().psx_last_error()
    yield _ops
    _ops.set_refract_halo(4)


def dev(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).cuda()


def _size(rng, lo, hi):
    """A grid length: mostly arbitrary, sometimes right at a boundary of the kernels' tilings."""
    edges = [lo, lo + 1, 55, 56, 57, 63, 64, 65, 111, 112, 113, 127, 128, 129, 255, 256, 257, 545, 546, 547, 1136, 1137, 1138]
    edges = [e for e in edges if lo <= e <= hi]
    if edges and rng.random() < 0.3:
        return int(rng.choice(edges))
    return int(rng.integers(lo, hi + 1))


def _smooth(rng, shape, cells):
    """A smooth random field in [-1, 1]: bilinear-ish blobs `cells` pixels wide (separable sums of a few cosines)."""
    x = np.arange(shape[0])[:, None] / cells
    y = np.arange(shape[1])[None, :] / cells
    f = np.zeros(shape)
    for _ in range(4):
        kx, ky = rng.uniform(0.3, 2.0, 2)
        px, py = rng.uniform(0, 2 * np.pi, 2)
        f += rng.uniform(0.3, 1.0) * np.cos(kx * x + px) * np.cos(ky * y + py)
    return f / np.max(np.abs(f))


# ------------------------------------------------------------------------------------------------ Fresnel
@pytest.mark.parametrize("case", range(16 * MULT))
def test_fuzz_fresnel(ops, case):
    rng = np.random.default_rng(10_000 + case)
    big = rng.random() < 0.2
    Nx, Ny = _size(rng, 16, 1300 if big else 400), _size(rng, 16, 700 if big else 400)
    if rng.random() < 0.5:
        Nx, Ny = Ny, Nx
    E = float(rng.uniform(12.0, 90.0))
    M = float(rng.uniform(1.0, 3.0))
    pix = float(rng.choice([0.25, 0.5, 1.0, 2.0, 6.5]) * rng.uniform(0.8, 1.25))
    nd = int(rng.integers(1, 6))
    zs = [float(z) for z in rng.uniform(0.02, 9.0, nd)]
    nmat = int(rng.integers(0, 4))
    has_wave = nmat == 0 or rng.random() < 0.5
    amp = float(rng.uniform(0.5, 2.0))
    engine = 2 if max(Nx, Ny) <= 4593 else 0
    w_in = (rng.normal(size=(Nx, Ny)) + 1j * rng.normal(size=(Nx, Ny))).astype(np.complex64) if has_wave else None
    T = delta = beta = None
    m = None
    k = orc.k_sample(E)
    w0 = amp * (w_in.astype(np.complex128) if has_wave else np.ones((Nx, Ny), dtype=np.complex128))
    if nmat:
        T = np.stack([(rng.uniform(0, 1) + _smooth(rng, (Nx, Ny), rng.uniform(3, 40))) * rng.uniform(1e-6, 3e-4)
                      for _ in range(nmat)]).astype(np.float32)
        T = np.abs(T)
        delta = list(rng.uniform(5e-8, 8e-7, nmat))
        beta = list(rng.uniform(1e-11, 5e-9, nmat))
        m = ops.MaterialStack(dev(T, torch.float32), cphase=[-k * d for d in delta], catt=[-k * b for b in beta])
        w0 = orc.set_wave(w0, T.astype(np.float64), delta, beta, E)
    kk = orc.getk(E * 1000)
    du = (2 * np.pi / (Nx * pix * 1e-6), 2 * np.pi / (Ny * pix * 1e-6))
    plan = ops.FresnelPlan(Nx, Ny, max_dist=max(nd, int(rng.integers(1, 6))), engine=engine)
    if engine == 2:
        assert plan.engine == 2
    add = rng.random() < 0.3
    scales = [float(s) for s in rng.uniform(0.2, 3.0, nd)]
    want = [bool(rng.random() < 0.7) for _ in range(nd)]
    base = rng.uniform(0.0, 2.0, (Nx, Ny)).astype(np.float32)
    inten = [dev(base, torch.float32) if (not want[d] or rng.random() < 0.6) else None for d in range(nd)]
    outs = plan.propagate([z / (2 * kk * M) for z in zs], [kk * z / M for z in zs], du,
                          wave_in=dev(w_in, torch.complex64) if has_wave else None, amp=amp, mats=m, want_wave=want,
                          inten_out=inten, inten_scale=scales, add=add)
    what = dict(case=case, shape=(Nx, Ny), nd=nd, nmat=nmat, has_wave=has_wave, add=add, E=E, M=M, pix=pix, zs=zs)
    for d in range(nd):
        ref = orc.wave_propagation(w0, zs[d], E, M, (Nx, Ny), pix)
        if want[d]:
            assert relmax(outs[d].cpu().numpy(), ref) < TOL, (what, d)
        if inten[d] is not None:
            ri = scales[d] * np.abs(ref) ** 2 + (base.astype(np.float64) if add else 0.0)
            assert relmax(inten[d].cpu().numpy(), ri) < TOL, (what, d, "intensity")
    plan.close()


@pytest.mark.parametrize("case", range(5 * MULT))
def test_fuzz_fresnel_long_lines(ops, case):
    """Lines longer than one LDS transform (partitioned, coupled and two-round kernels: 4594 ... 18402 samples) on either
    axis of a thin grid, whole images against the oracle."""
    rng = np.random.default_rng(15_000 + case)
    edges = [4594, 4600, 9202, 9203, 12278, 12279, 12280, 16384, 18402]
    long = int(rng.choice(edges)) if rng.random() < 0.4 else int(rng.integers(4594, 18403))
    short = _size(rng, 16, 200)
    Nx, Ny = (long, short) if rng.random() < 0.5 else (short, long)
    E = float(rng.uniform(15.0, 80.0))
    M = float(rng.uniform(1.0, 2.0))
    pix = float(rng.uniform(0.2, 2.0))
    nd = int(rng.integers(1, 4))
    zs = [float(z) for z in rng.uniform(0.05, 8.0, nd)]
    amp = float(rng.uniform(0.5, 2.0))
    k = orc.k_sample(E)
    T = np.abs((rng.uniform(0, 1) + _smooth(rng, (Nx, Ny), rng.uniform(5, 60))) * rng.uniform(1e-6, 2e-4))[None].astype(np.float32)
    delta, beta = [float(rng.uniform(5e-8, 8e-7))], [float(rng.uniform(1e-11, 5e-9))]
    m = ops.MaterialStack(dev(T, torch.float32), cphase=[-k * delta[0]], catt=[-k * beta[0]])
    w0 = orc.set_wave(np.full((Nx, Ny), amp, dtype=np.complex128), T.astype(np.float64), delta, beta, E)
    kk = orc.getk(E * 1000)
    du = (2 * np.pi / (Nx * pix * 1e-6), 2 * np.pi / (Ny * pix * 1e-6))
    plan = ops.FresnelPlan(Nx, Ny, max_dist=nd, engine=2)
    assert plan.engine == 2
    inten = [torch.zeros((Nx, Ny), dtype=torch.float32, device="cuda") for _ in zs]
    outs = plan.propagate([z / (2 * kk * M) for z in zs], [kk * z / M for z in zs], du, amp=amp, mats=m, inten_out=inten)
    for d in range(nd):
        ref = orc.wave_propagation(w0, zs[d], E, M, (Nx, Ny), pix)
        assert relmax(outs[d].cpu().numpy(), ref) < TOL, (case, Nx, Ny, nd, d)
        assert relmax(inten[d].cpu().numpy(), np.abs(ref) ** 2) < TOL, (case, Nx, Ny, nd, d, "intensity")
    plan.close()


# --------------------------------------------------------------------------------------------- refraction
@pytest.mark.parametrize("case", range(24 * MULT))
def test_fuzz_refraction(ops, case):
    rng = np.random.default_rng(20_000 + case)
    big = rng.random() < 0.15
    Nx, Ny = _size(rng, 8, 700 if big else 260), _size(rng, 8, 500 if big else 260)
    if rng.random() < 0.5:
        Nx, Ny = Ny, Nx
    ver = "v2" if rng.random() < 0.75 else "v1"
    margin = 15 if ver == "v2" else 10
    clamp = (Nx, Ny) if ver == "v2" else (1e3, 1e3)
    E = float(rng.uniform(15.0, 80.0))
    M = float(rng.uniform(1.0, 2.5))
    pix = float(rng.uniform(0.5, 4.0))
    h = pix * 1e-6
    halo = int(rng.choice([4, 6, 8, 12, 16]))
    ops.set_refract_halo(halo)
    nd = int(rng.integers(1, 5))
    zs = [float(z) for z in rng.uniform(0.05, 4.0, nd)]
    kr = orc.k_refraction(E)
    ks = orc.k_sample(E)
    nmat = int(rng.integers(0, 6))
    I0 = float(rng.uniform(0.5, 9000.0))
    # the phase: a smooth field whose steepest slope moves a ray by `reach` pixels at the longest distance, plus (sometimes)
    # a few hard steps that send single rays a long way (far rays, clamps)
    reach = float(rng.choice([0.3, 2.0, 5.0, 11.0, 30.0]))
    cells = float(rng.uniform(3.0, 25.0))
    steps = rng.random() < 0.35
    I_in = phi_in = None
    m = None
    if nmat:
        delta = list(rng.uniform(1e-7, 6e-7, nmat))
        beta = list(rng.uniform(1e-11, 4e-9, nmat))
        # phi = -k sum delta T: slope of T_0 per pixel for `reach`: D = dphi/dpix * z / (k h M) / h ... in pixels
        t_amp = reach * cells * kr * h * M * h / (max(zs) * ks * delta[0])
        T = [np.abs(t_amp * (1.0 + _smooth(rng, (Nx, Ny), cells)))]
        for _ in range(1, nmat):
            T.append(np.abs(rng.uniform(0, 0.2) * t_amp * delta[0] / delta[len(T)] * (1.0 + _smooth(rng, (Nx, Ny), cells))))
        T = np.stack(T).astype(np.float32)
        if steps:
            for _ in range(6):
                i, j = rng.integers(0, Nx), rng.integers(0, Ny)
                T[0, i:i + rng.integers(1, 4), j:j + rng.integers(1, 4)] *= rng.uniform(1.5, 40.0)
        m = ops.MaterialStack(dev(T, torch.float32), cphase=[-ks * d for d in delta], catt=[-2 * ks * b for b in beta])
    use_I = nmat == 0 or rng.random() < 0.4
    use_phi = nmat == 0 or rng.random() < 0.4
    if use_I:
        I_in = rng.uniform(0.0, 2.0, (Nx, Ny)).astype(np.float32)
        if rng.random() < 0.4:
            I_in[rng.random((Nx, Ny)) < 0.5] = 0.0
    if use_phi:
        p_amp = reach * cells * kr * h * M * h / max(zs)
        phi_in = p_amp * _smooth(rng, (Nx, Ny), cells)
        if steps and nmat == 0:
            for _ in range(6):
                i, j = rng.integers(0, Nx), rng.integers(0, Ny)
                phi_in[i:i + rng.integers(1, 4), j:j + rng.integers(1, 4)] += p_amp * rng.uniform(-60.0, 60.0)
    # the oracle's inputs: what the device forms from the float32 maps
    I_src = I0 * (I_in.astype(np.float64) if use_I else np.ones((Nx, Ny)))
    phi_src = phi_in.copy() if use_phi else 0
    if nmat:
        I_src, phi_src, _ = orc.set_wave_rt(I_src, T.astype(np.float64), delta, beta, E, phi_src)
    elif not use_phi:
        phi_src = np.zeros((Nx, Ny))
    add = rng.random() < 0.3
    out_scale = float(rng.uniform(0.3, 2.0))
    base = rng.uniform(0.0, 2.0, (Nx, Ny)).astype(np.float32)
    dsc = [z / kr / (h * M) / h for z in zs]
    kw = dict(margin=margin, I_in=dev(I_in, torch.float32) if use_I else None, I0=I0,
              phi_in=dev(phi_in, torch.float64) if use_phi else None, out_scale=out_scale, add=add)
    if nd == 1:
        out, Dx, Dy = ops.refract((Nx, Ny), m, dsc[0], clamp, out=dev(base, torch.float32) if add else None, want_D=True, **kw)
        outs = [out]
    else:
        outs = ops.refract_multi((Nx, Ny), m, dsc, clamp, outs=[dev(base, torch.float32) for _ in zs] if add else None, **kw)
        Dx = None
    ops.check_status(outs[0].device)
    what = dict(case=case, shape=(Nx, Ny), ver=ver, halo=halo, nd=nd, nmat=nmat, use_I=use_I, use_phi=use_phi, reach=reach,
                steps=steps, add=add)
    for d in range(nd):
        ref, Dxr, Dyr = orc.fast_refraction(np.array(I_src, dtype=np.float64), np.array(phi_src, dtype=np.float64), zs[d], E, M,
                                            pix, variant=ver)
        ref = out_scale * ref + (base.astype(np.float64) if add else 0.0)
        scale = max(np.max(np.abs(ref)), out_scale * np.max(np.abs(I_src)))     # a refraction can empty the whole frame
        err = np.max(np.abs(outs[d].cpu().numpy() - ref)) / scale if scale > 0 else 0.0
        assert err < TOL, (what, d, err)
        if Dx is not None:
            dmax = max(np.max(np.abs(Dxr)), np.max(np.abs(Dyr)), 1e-30)
            assert np.max(np.abs(Dx.cpu().numpy() - Dxr)) / dmax < 2e-6, (what, "Dx")
            assert np.max(np.abs(Dy.cpu().numpy() - Dyr)) / dmax < 2e-6, (what, "Dy")


# ------------------------------------------------------------------------------------------------ fastloop
@pytest.mark.parametrize("case", range(6 * MULT))
def test_fuzz_fastloop(ops, case):
    rng = np.random.default_rng(30_000 + case)
    Nx, Ny = _size(rng, 4, 300), _size(rng, 4, 300)
    I = rng.uniform(0, 3, (Nx, Ny)).astype(np.float32)
    reach = float(rng.choice([0.4, 3.0, 20.0, 400.0]))
    Dx = (reach * rng.normal(size=(Nx, Ny))).astype(np.float32)
    Dy = (reach * rng.normal(size=(Nx, Ny))).astype(np.float32)
    if rng.random() < 0.5:
        Dx[rng.random((Nx, Ny)) < 0.3] = np.round(Dx[rng.random((Nx, Ny)) < 0.3].mean())    # exact integers: weight-0 shares
    base = rng.uniform(0, 1, (Nx, Ny)).astype(np.float32)
    I2 = dev(base, torch.float32)
    ops.fastloop(dev(I, torch.float32), dev(Dx, torch.float32), dev(Dy, torch.float32), I2)
    ref = orc.fastloop(I, Dx, Dy, base.astype(np.float64).copy())
    assert relmax(I2.cpu().numpy(), ref) < TOL, (case, Nx, Ny, reach)


# ------------------------------------------------------------------------------------------------ detector
@pytest.mark.parametrize("case", range(12 * MULT))
def test_fuzz_detector(ops, case):
    rng = np.random.default_rng(40_000 + case)
    ov = int(rng.choice([1, 2, 3, 4]))
    nx, ny = _size(rng, 20, 330), _size(rng, 20, 330)
    Nx, Ny = nx * ov, ny * ov
    fwhm = float(rng.choice([0.0, 0.4, 1.3, 3.0, 8.0]) * rng.uniform(0.8, 1.2))
    psf = float(rng.choice([0.0, 0.5, 1.2, 2.5]) * rng.uniform(0.8, 1.2))
    img = rng.uniform(0.2, 2.0, (Nx, Ny)).astype(np.float32)
    img[rng.integers(0, Nx), rng.integers(0, Ny)] = 50.0                    # a hot pixel: the band edges show
    plan = ops.DetectorPlan(Nx, Ny, ov, nx, ny, fwhm / 2.355, psf)
    out = plan.detect(dev(img, torch.float32))
    plan.close()
    ref = orc.detection(img.astype(np.float64), fwhm, ov, (nx, ny), psf)
    assert relmax(out.cpu().numpy(), ref) < TOL, (case, Nx, Ny, ov, fwhm, psf)


# ---------------------------------------------------------------------------------------------- dark field
@pytest.mark.parametrize("case", range(6 * MULT))
def test_fuzz_darkfield_refraction(ops, case):
    from paresis_amd import refractionFileNumba2 as RF2
    rng = np.random.default_rng(50_000 + case)
    Nx, Ny = _size(rng, 24, 150), _size(rng, 24, 150)
    E = float(rng.uniform(20.0, 70.0))
    M = float(rng.uniform(1.0, 2.0))
    pix = float(rng.uniform(1.0, 4.0))
    z = float(rng.uniform(0.2, 2.0))
    h = pix * 1e-6
    kr = orc.k_refraction(E)
    reach = float(rng.choice([0.5, 3.0, 9.0]))
    cells = float(rng.uniform(4.0, 20.0))
    phi = reach * cells * kr * h * M * h / z * _smooth(rng, (Nx, Ny), cells)
    I = rng.uniform(0.5, 2.0, (Nx, Ny)).astype(np.float32)
    max_px = float(rng.choice([0.8, 2.0, 5.0, 12.0, 25.0]))     # up to 5: the LDS-tiled gather; beyond: bands of source rows
    df = np.clip(_smooth(rng, (Nx, Ny), cells), 0.0, None) * max_px * h * M / z           # radians
    if rng.random() < 0.5:
        df[:, : Ny // 2] = 0.0
    out, Dx, Dy = RF2.fastRefractionDF(I.copy(), phi, z, E, M, pix, df.copy())
    ref, Dxr, Dyr = orc.fast_refraction_df(I.astype(np.float64), phi.copy(), z, E, M, pix, df.copy())
    assert tuple(Dx.shape) == Dxr.shape
    assert relmax(out.cpu().numpy(), ref) < TOL, (case, Nx, Ny, reach, max_px)


# --------------------------------------------------------------------------------------------- whole chains
def _chain_cfg(rng, sim):
    """A random, small, injected experiment: spectrum, detector bins, scintillator / plate / air on or off, odd grids."""
    ov = int(rng.choice([1, 2, 3]))
    n0, n1 = _size(rng, 24, 90), _size(rng, 24, 90)
    N0, N1 = n0 * ov, n1 * ov
    dSM, dMO, dOD = float(rng.uniform(0.3, 140.0)), float(rng.uniform(0.05, 2.0)), float(rng.uniform(0.1, 4.0))
    M = (dSM + dMO + dOD) / (dSM + dMO)
    det_pix = float(rng.choice([6.5, 12.0, 24.0, 50.0]))
    pix = det_pix / ov / M
    h = pix * 1e-6
    nE = int(rng.choice([1, 1, 2, 3, 5]))
    e0 = float(rng.uniform(15.0, 60.0))
    step = float(rng.choice([1.0, 2.0, 5.0]))
    energies = [e0 + step * i for i in range(nE)]
    flux = rng.uniform(0.2, 1.0, nE)
    flux = flux / flux.sum()
    spectrum = [(float(e), float(f)) for e, f in zip(energies, flux)]
    nbins = int(rng.integers(0, min(3, nE)))                   # thresholds inside the spectrum; the chain appends the last energy
    bins = sorted(float(energies[i]) for i in rng.choice(nE - 1, nbins, replace=False)) if nbins and nE > 1 else []
    cells = float(rng.uniform(4.0, 18.0))
    reach = float(rng.choice([0.3, 1.5, 4.0])) if sim == "RT" else float(rng.choice([0.3, 1.0]))

    def obj(nmat, amp, d_lo, d_hi, rough):
        geom = np.stack([np.abs(amp[m] * (1.0 + (0.9 * _smooth(rng, (N0, N1), cells) if rough else np.zeros((N0, N1)))))
                         for m in range(nmat)])
        geom = geom.astype(np.float32).astype(np.float64)          # both sides see float32 thickness values
        delta = rng.uniform(d_lo, d_hi, (nmat, nE))
        # absorption index such that the thickest point of each map removes between 5 % and 80 % of the intensity
        beta = np.array([[rng.uniform(0.05, 1.5) / (2 * orc.k_sample(e) * geom[m].max()) for e in energies] for m in range(nmat)])
        return orc.Obj(geom, delta, beta)

    zmax = dMO + dOD
    d_mem = 1.2e-6
    t_mem = reach * cells * h * M * h / (zmax * d_mem)
    membrane = obj(2, [t_mem, 0.3 * t_mem], 0.6 * d_mem, d_mem, True)
    sample = obj(1, [rng.uniform(0.5, 3.0) * t_mem], 2e-7, 6e-7, True)
    in_vac = bool(rng.random() < 0.5)
    air = None if in_vac else obj(1, [dSM + zmax], 1e-10, 4e-10, False)
    plate = obj(1, [rng.uniform(1e-4, 1e-3)], 2e-7, 5e-7, False) if rng.random() < 0.5 else None
    scint = None
    if rng.random() < 0.4:
        scint = (float(rng.uniform(50.0, 400.0)), [(float(e), float(rng.uniform(1e-9, 2e-8))) for e in energies])
    return dict(scintillator=scint, dSM=dSM, dMO=dMO, dOD=dOD, meanShotCount=float(rng.uniform(50.0, 9000.0)), ov=ov, pix_um=pix,
                M=M, inVacuum=in_vac, N=(N0, N1), spectrum=spectrum, source_size_um=float(rng.choice([0.0, 1.0, 8.0, 40.0])),
                energy_sampling=step, det_dims=(n0, n1), det_pix_um=det_pix, psf=float(rng.choice([0.0, 0.6, 1.4])), bins=bins,
                membrane=membrane, sample=sample, air=air, plate=plate)


@pytest.mark.parametrize("sim", ["Fresnel", "RT"])
@pytest.mark.parametrize("case", range(5 * MULT))
def test_fuzz_chain(ops, case, sim):
    import copy
    from tests._build import build_experiment
    rng = np.random.default_rng(60_000 + case + (500 if sim == "RT" else 0))
    cfg = _chain_cfg(rng, sim)
    ref_cfg = copy.deepcopy(cfg)
    exp = build_experiment(cfg, sim)
    ops.set_refract_halo(int(rng.choice([4, 6, 8, 12, 16])))
    what = dict(case=case, sim=sim, N=cfg["N"], ov=cfg["ov"], nE=len(cfg["spectrum"]), bins=cfg["bins"], vac=cfg["inVacuum"],
                plate=cfg["plate"] is not None, scint=cfg["scintillator"] is not None, src=cfg["source_size_um"], psf=cfg["psf"])
    for point in (0, 1):
        exp.exp_dict["meanEnergy"] = 0
        out = exp.computeSampleAndReferenceImages(point)
        ref = orc.compute_fresnel(ref_cfg, point) if sim == "Fresnel" else orc.compute_rt(ref_cfg, point)
        names = ("Sample", "Reference", "Propag", "White")
        for k, nm in enumerate(names):
            if point == 1 and nm == "Propag":
                continue
            got = out[k].cpu().numpy()
            if point == 1 and nm == "White" and not np.any(got):
                continue
            assert relmax(got, ref[k]) < TOL, (what, point, nm, relmax(got, ref[k]))
        assert abs(exp.exp_dict["meanEnergy"] - ref[-1]) < 1e-4 * ref[-1], (what, point)
        if sim == "RT" and point == 0:
            dmax = max(np.max(np.abs(ref[4])), 1e-30)
            assert np.max(np.abs(out[4].cpu().numpy() - ref[4])) / dmax < 2e-6, (what, "Dx")


# --------------------------------------------------------------------------------------- membrane synthesis
@pytest.mark.parametrize("case", range(6 * MULT))
def test_fuzz_membrane(ops, case):
    """getMembraneSegmentedFromFile on random sphere lists, grids that need stitching along either axis, 1-4 layers."""
    import types
    from paresis_amd.Samples.getMembraneFromFile import getMembraneSegmentedFromFile
    rng = np.random.default_rng(70_000 + case)
    n = int(rng.choice([40, 900, 6000]))
    lst = np.stack([rng.uniform(-4870.0, 4870.0, n), rng.uniform(-4051.0, 4051.0, n), rng.uniform(6.0, 19.0, n)], axis=1)
    dimX, dimY = _size(rng, 40, 600), _size(rng, 40, 600)
    meanR = float(rng.uniform(3.0, 22.0))
    pix = float(rng.uniform(0.6, 5.0))
    layers = int(rng.integers(1, 5))
    support = float(rng.uniform(0.0, 8000.0))
    seed = int(rng.integers(0, 2 ** 31 - 1))
    smp = types.SimpleNamespace(myMeanSphereRadius=meanR, myNbOfLayers=layers)
    try:
        ref = orc.membrane_segmented(lst, dimX, dimY, pix, meanR, layers, support, seed)
    except ValueError:
        # a stitched list that leaves no room for the random layer offset (getMembraneFromFile.py:139-140: randint(low >= high)):
        # the reference raises, so must the mirror (case 323 at PSX_FUZZ=100)
        with pytest.raises(ValueError):
            getMembraneSegmentedFromFile(smp, dimX, dimY, pix, 0, support, seed=seed, sphere_list=lst)
        return
    geom, _ = getMembraneSegmentedFromFile(smp, dimX, dimY, pix, 0, support, seed=seed, sphere_list=lst)
    what = (case, n, dimX, dimY, meanR, pix, layers)
    assert relmax(geom[0].cpu().numpy(), ref[0]) < 1e-6, what
    assert relmax(geom[1].cpu().numpy(), ref[1]) < 1e-6, what


@pytest.mark.parametrize("case", range(3 * MULT))
def test_fuzz_chain_darkfield(ops, case):
    """The ray-tracing chain with a scattering sample (Sample.py:322-344: 'Lung' material or the sample named
    'cylinder_beeds'): fastRefractionDF inside the energy loop, the dark-field map of position 0."""
    import copy
    from tests._build import build_experiment
    rng = np.random.default_rng(80_000 + case)
    cfg = _chain_cfg(rng, "RT")
    lung = bool(rng.random() < 0.5)
    N0, N1 = cfg["N"]
    nE = len(cfg["spectrum"])
    h = cfg["pix_um"] * 1e-6
    delta = rng.uniform(2e-7, 6e-7, (1, nE))
    radius, fraction = (47, 0.5) if lung else (15, 0.6)
    # thickness for a dark field of `px` pixels at the detector (SAM:324-343 solved for the thickness, first energy)
    px = float(rng.choice([0.7, 1.9, 3.3]))
    d0 = delta[0, 0]
    c = (fraction * 3 / 4 / np.pi / radius ** 3) ** (1 / 3)
    t_um = (px * h * cfg["M"] / (cfg["dOD"] * 2 * d0 * np.sqrt(np.log(2 / d0) + 1))) ** 2 / c
    shape = np.clip(_smooth(rng, (N0, N1), float(rng.uniform(8.0, 30.0))) + 0.2, 0.0, None)
    geom = (t_um * 1e-6 * shape / shape.max())[None].astype(np.float32).astype(np.float64)
    beta = np.array([[rng.uniform(0.05, 1.0) / (2 * orc.k_sample(e) * geom.max()) for e, _ in cfg["spectrum"]]])
    name = "lungs" if lung else "cylinder_beeds"
    mats = ["Lung"] if lung else ["PMMA"]
    cfg["sample"] = orc.Obj(geom, delta, beta, materials=mats, my_type="sample_of_interest", name=name)
    ref_cfg = copy.deepcopy(cfg)
    exp = build_experiment(cfg, "RT", sample_materials=tuple(mats), sample_name=name)
    what = dict(case=case, N=cfg["N"], lung=lung, px=px, nE=nE, bins=cfg["bins"])
    for point in (0, 1):
        exp.exp_dict["meanEnergy"] = 0
        out = exp.computeSampleAndReferenceImages(point)
        ref = orc.compute_rt(ref_cfg, point)
        for k, nm in enumerate(("Sample", "Reference", "Propag", "White")):
            if point == 1 and nm in ("Propag", "White"):
                continue
            assert relmax(out[k].cpu().numpy(), ref[k]) < TOL, (what, point, nm, relmax(out[k].cpu().numpy(), ref[k]))
        if point == 0:
            assert tuple(out[4].shape) == ref[4].shape, (what, "Dx shape")
            assert relmax(out[6].cpu().numpy(), ref_cfg["_darkFieldPropag"]) < 2e-6, (what, "dark-field map")
        else:
            assert float(out[6].abs().max()) == 0.0
