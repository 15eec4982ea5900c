"""GPU parity of each HIP kernel family (through the C ABI) against the golden vectors and the CPU oracle.

Tolerance: max|out-ref| / max|ref| <= 1e-5 (BASELINE.json north_star, fp32 vs the fp64 reference)."""
import numpy as np
import pytest
import torch

from oracle import paresis_oracle as orc
from tests._golden import load, relmax

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def ops():
    from paresis_amd import ops as _ops
    from paresis_amd._lib import lib
    assert lib().psx_device_ok() == 1, lib().psx_last_error()
    return _ops


def dev(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).cuda()


def test_transmission_golden(ops):
    g = load("transmission.npz")
    T = dev(g["T"], torch.float32)
    for ie, E in enumerate(g["energies"]):
        k = orc.k_sample(E)
        dl, bl = g["delta"][:, ie], g["beta"][:, ie]
        m = ops.MaterialStack(T, cphase=-k * dl, catt=-k * bl)
        out = ops.transmit_wave(dev(g["wave_in"], torch.complex64), 1.0, m)
        assert relmax(out.cpu().numpy(), g["setWave/%d" % ie]) < TOL
        m2 = ops.MaterialStack(T, cphase=-k * dl, catt=-2 * k * bl)
        I, phi = ops.transmit_rt(dev(g["I_in"], torch.float32), 1.0, m2, dev(g["phi_in"], torch.float64))
        assert relmax(I.cpu().numpy(), g["setWaveRT/%d/I" % ie]) < TOL
        assert relmax(phi.cpu().numpy(), g["setWaveRT/%d/phi" % ie]) < 1e-12   # float64 on the device
        # unit wave / ones intensity through NULL inputs
        out1 = ops.transmit_wave(None, 2.0, m)
        ref1 = orc.set_wave(np.full(T.shape[1:], 2.0 + 0j), g["T"], dl, bl, E)
        assert relmax(out1.cpu().numpy(), ref1) < TOL
        # in place
        w = dev(g["wave_in"], torch.complex64)
        ops.transmit_wave(w, 1.0, m, out=w)
        assert relmax(w.cpu().numpy(), g["setWave/%d" % ie]) < TOL


def test_accumulate(ops):
    rng = np.random.default_rng(1)
    img = rng.uniform(1, 2, (33, 47)); acc0 = rng.uniform(1, 2, (33, 47)); T = rng.uniform(0, 1e-3, (1, 33, 47))
    acc = dev(acc0, torch.float32)
    m = ops.MaterialStack(dev(T, torch.float32), catt=[-700.0])
    ops.accumulate(acc, dev(img, torch.float32), 0.5, m, add=True)
    ref = acc0 + 0.5 * img * np.exp(-700.0 * T[0].astype(np.float32).astype(np.float64))
    assert relmax(acc.cpu().numpy(), ref) < 1e-6
    ops.accumulate(acc, dev(img, torch.float32), 2.0, None, add=False)
    assert relmax(acc.cpu().numpy(), 2.0 * img) < 1e-6


@pytest.mark.parametrize("engine", [1, 0])
def test_fresnel_golden(ops, engine):
    g = load("fresnel.npz")
    worst = 0.0
    for k in range(int(g["n"])):
        z, E, M, pix = g["%d/params" % k]
        w = g["%d/wave" % k]
        Nx, Ny = w.shape
        plan = ops.FresnelPlan(Nx, Ny, engine=engine)
        kk = orc.getk(E * 1000)
        a = z / (2 * kk * M)
        du = (2 * np.pi / (Nx * pix * 1e-6), 2 * np.pi / (Ny * pix * 1e-6))
        inten = torch.zeros((Nx, Ny), dtype=torch.float32, device="cuda")
        out = plan.propagate([a], [kk * z / M], du, wave_in=dev(w, torch.complex64), inten_out=[inten])[0]
        err = relmax(out.cpu().numpy(), g["%d/out" % k])
        worst = max(worst, err)
        assert err < TOL, (k, err)
        assert relmax(inten.cpu().numpy(), np.abs(g["%d/out" % k]) ** 2) < TOL
        plan.close()
    print("fresnel engine", engine, "worst rel err", worst)


@pytest.mark.parametrize("engine", [1, 0])
def test_fresnel_fused_transmission_and_shared_forward(ops, engine):
    """K1 fused into the padded load; two distances share one forward transform (EXP:341 and EXP:349)."""
    g = load("transmission.npz")
    T64 = g["T"]
    Nx, Ny = T64.shape[1:]
    E, pix = 52.0, 2.9
    k = orc.k_sample(E)
    dl, bl = g["delta"][:, 0], g["beta"][:, 0]
    m = ops.MaterialStack(dev(T64, torch.float32), cphase=-k * dl, catt=-k * bl)
    w0 = orc.set_wave(np.full((Nx, Ny), 86.6 + 0j), T64, dl, bl, E)
    plan = ops.FresnelPlan(Nx, Ny, engine=engine)
    zs = [(1.6, 141.6 / 140), (5.2, 145.2 / 141.6), (0.0, 1.0)]
    kk = orc.getk(E * 1000)
    du = (2 * np.pi / (Nx * pix * 1e-6), 2 * np.pi / (Ny * pix * 1e-6))
    acc = [torch.ones((Nx, Ny), dtype=torch.float32, device="cuda") for _ in zs]
    outs = plan.propagate([z / (2 * kk * M) for z, M in zs], [kk * z / M for z, M in zs], du, amp=86.6, mats=m,
                          inten_out=acc, inten_scale=[1.0, 0.5, 2.0], add=True)
    for (z, M), o, ac, sc in zip(zs, outs, acc, [1.0, 0.5, 2.0]):
        ref = orc.wave_propagation(w0, z, E, M, (Nx, Ny), pix)
        assert relmax(o.cpu().numpy(), ref) < TOL
        assert relmax(ac.cpu().numpy(), 1.0 + sc * np.abs(ref) ** 2) < TOL
    plan.close()


@pytest.mark.parametrize("shape", [(16, 40), (97, 131), (131, 97), (33, 1000), (4593, 33), (600, 34)])
def test_fresnel_lds_engine_edge_geometries(ops, shape):
    """Ragged line groups, every line length class (M = 1152 ... 9216, incl. the longest line the engine takes), lines
    straddling the last group, an input wave AND four materials through the transposing pre-pass."""
    Nx, Ny = shape
    rng = np.random.default_rng(7 + Nx)
    E, pix, z, M = 52.0, 2.9, 2.3, 1.02
    T64 = rng.uniform(0.0, 3e-5, size=(4, Nx, Ny))
    dl = np.array([6.2e-7, 9.9e-8, 3e-7, 1e-7])
    bl = np.array([4e-9, 4.5e-11, 1e-9, 2e-10])
    w_in = rng.normal(size=(Nx, Ny)) + 1j * rng.normal(size=(Nx, Ny))
    k = orc.k_sample(E)
    m = ops.MaterialStack(dev(T64, torch.float32), cphase=-k * dl, catt=-k * bl)
    T32 = T64.astype(np.float32).astype(np.float64)          # both sides see the fp32 thickness values
    w0 = orc.set_wave(0.75 * w_in.astype(np.complex64).astype(np.complex128), T32, dl, bl, E)
    ref = orc.wave_propagation(w0, z, E, M, (Nx, Ny), pix)
    plan = ops.FresnelPlan(Nx, Ny, engine=2)
    assert plan.engine == 2, "the LDS engine must take this geometry"
    kk = orc.getk(E * 1000)
    du = (2 * np.pi / (Nx * pix * 1e-6), 2 * np.pi / (Ny * pix * 1e-6))
    out = plan.propagate([z / (2 * kk * M)], [kk * z / M], du, wave_in=dev(w_in, torch.complex64), amp=0.75, mats=m)[0]
    assert relmax(out.cpu().numpy(), ref) < TOL
    plan.close()


@pytest.mark.parametrize("shape,nd", [((512, 40), 1), ((40, 512), 2), ((1024, 500), 3), ((498, 2048), 1), ((2048, 505), 2), ((4096, 36), 1),
                                      ((36, 4096), 2), ((4090, 1021), 3), ((1100, 33), 1), ((2200, 20), 2), ((300, 4081), 1),
                                      ((16384, 36), 2), ((36, 16384), 1), ((16370, 20), 3), ((24, 16356), 2), ((16384, 520), 4)])
def test_fresnel_power_of_two_lines_and_wrapped_outputs(ops, shape, nd):
    """Lines of N samples through ONE M = 256 R1-point transform with M just below N + P - 1 (fresnel_p2.hip): every radix
    (M = 1024 ... 8192), on either axis, with 29 / 5 / 1 / 0 wrapped outputs put right by the taps they missed, one distance
    and the shared-forward rounds of several; the line lengths that stay on the 576 R3-point transforms (1100, 2200); and lines
    of about 16384 samples, one 32768-point convolution in two coupled rounds (fresnel_p2x.hip: 29 / 1 wrapped outputs, either
    axis, a line fetched once for the rounds of all its distances).
    Whole complex fields against the float64 oracle; the same plan with the switch "no_p2" must agree as well."""
    from paresis_amd._lib import lib
    Nx, Ny = shape
    rng = np.random.default_rng(11 + Nx + 3 * Ny)
    E, pix, M = 52.0, 2.9, 1.03
    zs = (2.3, 7.2, 0.4)[:nd]
    T64 = rng.uniform(0.0, 3e-5, size=(2, Nx, Ny))
    dl, bl = np.array([6.2e-7, 9.9e-8]), np.array([4e-9, 4.5e-11])
    w_in = rng.normal(size=(Nx, Ny)) + 1j * rng.normal(size=(Nx, Ny))
    k = orc.k_sample(E)
    m = ops.MaterialStack(dev(T64, torch.float32), cphase=-k * dl, catt=-k * bl)
    T32 = T64.astype(np.float32).astype(np.float64)
    w0 = orc.set_wave(1.25 * w_in.astype(np.complex64).astype(np.complex128), T32, dl, bl, E)
    kk = orc.getk(E * 1000)
    du = (2 * np.pi / (Nx * pix * 1e-6), 2 * np.pi / (Ny * pix * 1e-6))
    refs = [orc.wave_propagation(w0, z, E, M, (Nx, Ny), pix) for z in zs]
    for no_p2 in (0, 1):
        assert lib().psx_debug_switch(b"no_p2", no_p2) == 0
        try:
            plan = ops.FresnelPlan(Nx, Ny, max_dist=nd, engine=2)
        finally:
            lib().psx_debug_switch(b"no_p2", 0)
        assert plan.engine == 2
        inten = [torch.zeros((Nx, Ny), dtype=torch.float32, device="cuda") for _ in zs]
        outs = plan.propagate([z / (2 * kk * M) for z in zs], [kk * z / M for z in zs], du, wave_in=dev(w_in, torch.complex64),
                              amp=1.25, mats=m, inten_out=inten)
        for z, o, it, ref in zip(zs, outs, inten, refs):
            assert relmax(o.cpu().numpy(), ref) < TOL, (shape, z, no_p2)
            assert relmax(it.cpu().numpy(), np.abs(ref) ** 2) < TOL, (shape, z, no_p2)
        plan.close()


def test_fresnel_known_answers(ops):
    # uniform wave keeps its modulus (reflect pad of a constant is constant: only the DC bin, chirp(0)=1)
    plan = ops.FresnelPlan(40, 44)
    w = torch.full((40, 44), 3.0 + 0j, dtype=torch.complex64, device="cuda")
    out = plan.propagate([1.3e-9], [0.7], (5e4, 4e4), wave_in=w)[0]
    assert np.allclose(np.abs(out.cpu().numpy()), 3.0, rtol=2e-6)


@pytest.mark.parametrize("ver", ["v2", "v1"])
def test_refraction_golden_phi64(ops, ver):
    """Signature-faithful path: explicit float64 phase array."""
    g = load("refraction.npz")
    margin = 15 if ver == "v2" else 10
    for k in range(int(g["n"])):
        z, E, M, pix = g["%d/params" % k]
        I = g["%d/I" % k]; phi = g["%d/phi" % k]
        Nx, Ny = I.shape
        h = pix * 1e-6
        dscale = z / orc.k_refraction(E) / (h * M) / h
        clamp = (Nx, Ny) if ver == "v2" else (1e3, 1e3)
        Iin = dev(I, torch.float32)
        out, Dx, Dy = ops.refract((Nx, Ny), None, dscale, clamp, margin=margin, I_in=Iin, phi_in=dev(phi, torch.float64),
                                  want_D=True, I_mut=Iin)
        ops.check_status(out.device)
        assert relmax(out.cpu().numpy(), g["%d/%s/out" % (k, ver)]) < TOL, (k, ver)
        assert relmax(Dx.cpu().numpy(), g["%d/%s/Dx" % (k, ver)]) < 1e-6
        assert relmax(Dy.cpu().numpy(), g["%d/%s/Dy" % (k, ver)]) < 1e-6
        assert relmax(Iin.cpu().numpy(), g["%d/%s/I_after" % (k, ver)]) < 1e-6     # clamped rays zeroed in place


def test_refraction_from_thickness_maps(ops):
    """Fused path: phase and attenuation formed on the device from float32 thickness maps (never phi in fp32)."""
    g = load("refraction.npz")
    for k in range(int(g["n"])):
        z, E, M, pix = g["%d/params" % k]
        T = g["%d/T" % k]
        Nx, Ny = T.shape
        kk = orc.k_sample(E)
        slab = np.full_like(T, 6e-3)
        geom = np.stack([T, slab])
        I_ref, phi_ref, _ = orc.set_wave_rt(np.full((Nx, Ny), 7500.0), geom, [6.2e-7, 9.87e-8], [4.0e-9, 4.5e-11], E, 0)
        ref, Dxr, Dyr = orc.fast_refraction(I_ref.copy(), phi_ref, z, E, M, pix)
        m = ops.MaterialStack(dev(geom, torch.float32), cphase=[-kk * 6.2e-7, -kk * 9.87e-8],
                              catt=[-2 * kk * 4.0e-9, -2 * kk * 4.5e-11])
        h = pix * 1e-6
        out, Dx, Dy = ops.refract((Nx, Ny), m, z / orc.k_refraction(E) / (h * M) / h, (Nx, Ny), I0=7500.0, want_D=True)
        assert relmax(out.cpu().numpy(), ref) < TOL, k
        assert relmax(Dx.cpu().numpy(), Dxr) < 1e-6
        # accumulate + scale
        acc = torch.full((Nx, Ny), 3.0, dtype=torch.float32, device="cuda")
        ops.refract((Nx, Ny), m, z / orc.k_refraction(E) / (h * M) / h, (Nx, Ny), I0=7500.0, out=acc, out_scale=0.5, add=True)
        assert relmax(acc.cpu().numpy(), 3.0 + 0.5 * ref) < TOL


@pytest.mark.parametrize("nmat", [5, 8])
def test_refraction_more_than_four_maps(ops, nmat):
    """The nmat = 8 instantiations (5..8 maps; staged in two batches, free of scratch memory since round 3): the membrane map
    split into `nmat` slices with their own coefficients against the oracle on the equivalent (I, phi), on a grid with
    interior AND border tiles, one distance and a distance batch."""
    g = load("refraction.npz")
    z, E, M, pix = g["0/params"]
    T0 = g["0/T"]
    reps = (int(np.ceil(200 / T0.shape[0])), int(np.ceil(200 / T0.shape[1])))
    T = np.tile(T0, reps)[:200, :200].astype(np.float32)
    Nx, Ny = T.shape
    kk = orc.k_sample(E)
    rng = np.random.default_rng(nmat)
    frac = rng.uniform(0.5, 1.5, nmat)
    geom = np.stack([(T * f).astype(np.float32) for f in frac])
    delta = list(rng.uniform(1e-7, 6e-7, nmat))
    beta = list(rng.uniform(1e-10, 4e-9, nmat))
    I_ref, phi_ref, _ = orc.set_wave_rt(np.full((Nx, Ny), 7500.0), geom.astype(np.float64), delta, beta, E, 0)
    m = ops.MaterialStack(dev(geom, torch.float32), cphase=[-kk * d for d in delta], catt=[-2 * kk * b for b in beta])
    h = pix * 1e-6
    zs = (z, 0.5 * z)
    dsc = [zz / orc.k_refraction(E) / (h * M) / h for zz in zs]
    outs = ops.refract_multi((Nx, Ny), m, dsc, (Nx, Ny), I0=7500.0)
    for zz, ds, o in zip(zs, dsc, outs):
        ref, _, _ = orc.fast_refraction(I_ref.copy(), phi_ref, zz, E, M, pix)
        single, _, _ = ops.refract((Nx, Ny), m, ds, (Nx, Ny), I0=7500.0)
        assert relmax(single.cpu().numpy(), ref) < TOL, (nmat, zz)
        assert relmax(o.cpu().numpy(), ref) < TOL, (nmat, zz)
    ops.check_status(outs[0].device)


def test_refraction_known_answers(ops):
    rng = np.random.default_rng(3)
    Nx, Ny = 70, 131
    I = rng.uniform(1, 2, (Nx, Ny)).astype(np.float32)
    # phi = const -> D == 0 -> identity (RF2:222-224)
    out, _, _ = ops.refract((Nx, Ny), None, 1.0, (Nx, Ny), I_in=dev(I, torch.float32),
                            phi_in=torch.full((Nx, Ny), 123.456, dtype=torch.float64, device="cuda"))
    assert np.array_equal(out.cpu().numpy(), I)
    # phi linear along axis 0 -> uniform integer shift by s pixels
    s = 3
    phi = (np.arange(Nx, dtype=np.float64) * s)[:, None] * np.ones((1, Ny))
    out, Dx, _ = ops.refract((Nx, Ny), None, 1.0, (Nx, Ny), I_in=dev(I, torch.float32), phi_in=dev(phi, torch.float64), want_D=True)
    o = out.cpu().numpy()
    assert np.allclose(o[s:], I[:-s], rtol=1e-6) and np.all(o[:s] == 0)
    assert np.allclose(Dx.cpu().numpy()[15:-15, 15:-15], s)
    # flux conservation when nothing leaves the grid: shift by a fraction inside a zero border
    I2 = np.zeros((Nx, Ny), np.float32); I2[20:50, 20:100] = I[20:50, 20:100]
    phi = 0.37 * np.arange(Nx, dtype=np.float64)[:, None] - 1.6 * np.arange(Ny, dtype=np.float64)[None, :]
    out, _, _ = ops.refract((Nx, Ny), None, 1.0, (Nx, Ny), I_in=dev(I2, torch.float32), phi_in=dev(phi, torch.float64))
    assert abs(float(out.sum().item()) / float(I2.sum()) - 1) < 1e-5


def test_refraction_far_rays_and_border_rules(ops):
    """Displacements far beyond the gather halo (and beyond the margin) go through the far-ray replay."""
    rng = np.random.default_rng(5)
    Nx, Ny = 150, 97
    I = rng.uniform(1, 2, (Nx, Ny))
    phi = np.cumsum(rng.uniform(-30, 30, (Nx, Ny)), axis=0) + np.cumsum(rng.uniform(-30, 30, (Nx, Ny)), axis=1)
    I32 = I.astype(np.float32).astype(np.float64)
    ref, Dxr, Dyr = orc.fast_refraction(I32.copy(), phi.copy(), 1.0, 52.0, 1.0, 1.0)   # arbitrary scale
    k = orc.k_refraction(52.0)
    dscale = 1.0 / k / (1e-6 * 1.0) / 1e-6
    out, Dx, Dy = ops.refract((Nx, Ny), None, dscale, (Nx, Ny), I_in=dev(I32, torch.float32), phi_in=dev(phi, torch.float64), want_D=True)
    assert np.abs(Dxr).max() > 20
    assert relmax(Dx.cpu().numpy(), Dxr) < 1e-6
    assert relmax(out.cpu().numpy(), ref) < TOL


@pytest.mark.parametrize("halo", [4, 6, 8, 12, 16])
def test_refraction_distance_batch(ops, halo):
    """psx_refract_multi_f32: the distances of a call share one staged tile; every image equals the oracle's and the
    one-distance call's (bitwise where no ray is far, i.e. where no global float atomics are involved)."""
    from paresis_amd._lib import lib
    g = load("refraction.npz")
    zs = (0.3, 1.6, 3.6, 7.2, 40.0)          # the last one sends most rays through the far replay
    try:
        assert lib().psx_refract_set_halo(halo) == 0
        for k in range(int(g["n"])):
            _, E, M, pix = g["%d/params" % k]
            T = g["%d/T" % k]
            Nx, Ny = T.shape
            kk = orc.k_sample(E)
            geom = np.stack([T, np.full_like(T, 6e-3)])
            I_ref, phi_ref, _ = orc.set_wave_rt(np.full((Nx, Ny), 7500.0), geom, [6.2e-7, 9.87e-8], [4.0e-9, 4.5e-11], E, 0)
            m = ops.MaterialStack(dev(geom, torch.float32), cphase=[-kk * 6.2e-7, -kk * 9.87e-8],
                                  catt=[-2 * kk * 4.0e-9, -2 * kk * 4.5e-11])
            h = pix * 1e-6
            dsc = [z / orc.k_refraction(E) / (h * M) / h for z in zs]
            outs = ops.refract_multi((Nx, Ny), m, dsc, (Nx, Ny), I0=7500.0)
            for z, d, o in zip(zs, dsc, outs):
                ref, Dxr, Dyr = orc.fast_refraction(I_ref.copy(), phi_ref, z, E, M, pix)
                assert relmax(o.cpu().numpy(), ref) < TOL, (k, z)
                single, _, _ = ops.refract((Nx, Ny), m, d, (Nx, Ny), I0=7500.0)
                if max(np.abs(Dxr).max(), np.abs(Dyr).max()) < halo - 1:
                    assert torch.equal(o, single), (k, z)
                else:
                    assert relmax(o.cpu().numpy(), single.cpu().numpy()) < 1e-6, (k, z)
            # explicit (I, phi) source, accumulate + scale into existing images
            I32 = dev(I_ref, torch.float32)
            accs = [torch.full((Nx, Ny), 3.0 + i, dtype=torch.float32, device="cuda") for i in range(2)]
            ops.refract_multi((Nx, Ny), None, dsc[1:3], (Nx, Ny), I_in=I32, phi_in=dev(phi_ref, torch.float64), outs=accs,
                              out_scale=0.5, add=True)
            for i, z in enumerate(zs[1:3]):
                ref, _, _ = orc.fast_refraction(I32.cpu().numpy().astype(np.float64), phi_ref, z, E, M, pix)
                assert relmax(accs[i].cpu().numpy(), 3.0 + i + 0.5 * ref) < TOL, (k, z)
        # argument checks: more distances than a call takes, two distances sharing an image
        from paresis_amd._lib import PsxError
        with pytest.raises(PsxError):
            ops.refract_multi((Nx, Ny), m, [1.0] * 9, (Nx, Ny))
        o = torch.empty((Nx, Ny), dtype=torch.float32, device="cuda")
        with pytest.raises(PsxError):
            ops.refract_multi((Nx, Ny), m, [1.0, 2.0], (Nx, Ny), outs=[o, o])
    finally:
        lib().psx_refract_set_halo(4)


@pytest.mark.parametrize("halo", [4, 6, 8, 12, 16])
def test_refraction_both_tile_geometries(ops, halo):
    """The gather halo (4 ... 16 pixels) is a speed knob: every golden case passes with each of the five tile geometries."""
    from paresis_amd._lib import lib
    g = load("refraction.npz")
    try:
        assert lib().psx_refract_set_halo(halo) == 0
        for k in range(int(g["n"])):
            z, E, M, pix = g["%d/params" % k]
            I = g["%d/I" % k]; phi = g["%d/phi" % k]
            Nx, Ny = I.shape
            h = pix * 1e-6
            out, Dx, Dy = ops.refract((Nx, Ny), None, z / orc.k_refraction(E) / (h * M) / h, (Nx, Ny),
                                      I_in=dev(I, torch.float32), phi_in=dev(phi, torch.float64), want_D=True)
            assert relmax(out.cpu().numpy(), g["%d/v2/out" % k]) < TOL, (halo, k)
        # bitwise reproducible when no ray is far (fixed-point tile sums do not depend on the atomics' order)
        z, E, M, pix = g["0/params"]
        I = dev(g["0/I"], torch.float32); phi = dev(g["0/phi"] * 0.2, torch.float64)
        h = pix * 1e-6
        a1, _, _ = ops.refract(I.shape, None, z / orc.k_refraction(E) / (h * M) / h, I.shape, I_in=I, phi_in=phi)
        a2, _, _ = ops.refract(I.shape, None, z / orc.k_refraction(E) / (h * M) / h, I.shape, I_in=I, phi_in=phi)
        assert torch.equal(a1, a2)
    finally:
        lib().psx_refract_set_halo(4)
    assert lib().psx_refract_set_halo(5) != 0


def test_refraction_status_flag(ops):
    Nx, Ny = 64, 64
    I = torch.full((Nx, Ny), float("inf"), dtype=torch.float32, device="cuda")
    ops.refract((Nx, Ny), None, 1.0, (Nx, Ny), I_in=I, phi_in=torch.zeros((Nx, Ny), dtype=torch.float64, device="cuda"))
    with pytest.raises(Exception, match="nans or insane"):
        ops.check_status(I.device)
    ops.check_status(I.device)   # cleared


def test_fastloop_golden(ops):
    g = load("refraction.npz")
    I2 = torch.zeros(g["loop/I"].shape, dtype=torch.float32, device="cuda")
    ops.fastloop(dev(g["loop/I"], torch.float32), dev(g["loop/Dx"], torch.float32), dev(g["loop/Dy"], torch.float32), I2)
    assert relmax(I2.cpu().numpy(), g["loop/out"]) < TOL


def test_detector_golden(ops):
    g = load("detector.npz")
    for k in range(int(g["n"])):
        d0, d1, ov, fwhm, psf = g["%d/params" % k]
        img = g["%d/in" % k]
        plan = ops.DetectorPlan(img.shape[0], img.shape[1], int(ov), int(d0), int(d1), fwhm / 2.355, psf)
        out = plan.detect(dev(img, torch.float32))
        assert relmax(out.cpu().numpy(), g["%d/out" % k]) < TOL, k
        plan.close()


@pytest.mark.parametrize("case", [(1100, 1204, 2, 1.3, 1.2), (1101, 1203, 3, 0.0, 0.8), (1600, 2052, 4, 2.6, 0.0),
                                  (700, 520, 1, 0.9, 1.7), (2400, 600, 2, 9.0, 3.1), (1024, 1536, 2, 0.5, 0.0),
                                  (520, 2052, 4, 0.3, 2.0), (2048, 2048, 2, 0.036, 1.2), (1500, 1300, 2, 0.7, 2.4),
                                  (1028, 772, 4, 0.0, 1.7)])
def test_detector_multi_block_grids(ops, case):
    """The banded detector kernels on grids of several 256-output blocks: the fused (contiguous axis, axis 0) pairs where
    the rows are 16-byte aligned and the bands fit (front only, front + PSF, the bench geometry) and the four-pass form
    elsewhere (unaligned rows, wide source blur), with and without each blur; the PSF stage as a stencil (k_psf_tile: 5, 9, 11, 13
    and 15 taps, its three instantiations) and as a banded pair (19 taps): against the dense composite operator of the
    host builder (the one the CPU suite holds to the oracle), applied in float64."""
    Nx, Ny, ov, sig_src, sig_psf = case
    nx, ny = Nx // ov, Ny // ov
    rng = np.random.default_rng(Nx + Ny)
    img = rng.uniform(0.5, 2.0, (Nx, Ny)).astype(np.float32)

    def dense(N, n):
        start, w = ops.detector_operator_host(N, ov, n, sig_src, sig_psf)
        C = np.zeros((n, N))
        for r in range(n):
            k = min(w.shape[1], N - start[r])
            C[r, start[r]:start[r] + k] = w[r, :k]
        return C

    ref = dense(Nx, nx) @ img.astype(np.float64) @ dense(Ny, ny).T
    plan = ops.DetectorPlan(Nx, Ny, ov, nx, ny, sig_src, sig_psf)
    out = plan.detect(dev(img, torch.float32))
    plan.close()
    assert out.shape == (nx, ny)
    assert relmax(out.cpu().numpy(), ref) < 2e-6, case


@pytest.mark.parametrize("case", [(1100, 1204, 2, 1.3, 1.2), (1600, 2052, 4, 2.6, 0.0), (1101, 1203, 3, 0.0, 0.8),
                                  (2048, 2048, 2, 0.036, 1.2), (2400, 600, 2, 9.0, 3.1), (1500, 1300, 2, 0.7, 2.4)])
def test_detector_images_of_a_bin_in_one_call(ops, case):
    """psx_detect_multi_f32 (the two to four images of an energy bin, EXP:388-394): with both stages fused the images share
    each launch; in every geometry -- fused front + PSF, fused front alone, the four-pass form (unaligned rows, wide blur) --
    and for 2 ... 5 images (5: two calls) image k is bit for bit what detect() of it alone writes, misaligned inputs and
    reused outputs included."""
    Nx, Ny, ov, sig_src, sig_psf = case
    nx, ny = Nx // ov, Ny // ov
    g = torch.Generator(device="cuda").manual_seed(Nx * 7 + Ny)
    plan = ops.DetectorPlan(Nx, Ny, ov, nx, ny, sig_src, sig_psf)
    imgs = [torch.rand((Nx, Ny), generator=g, device="cuda") * (3.0 + k) for k in range(5)]
    single = [plan.detect(im).clone() for im in imgs]
    for n in (2, 3, 4, 5):
        outs = plan.detect_many(imgs[:n])
        for k in range(n):
            assert torch.equal(outs[k], single[k]), (case, n, k)
    # one image that is not 16-byte aligned sends the call down the one-by-one path; outputs handed in are filled in place
    buf = torch.empty(Nx * Ny + 4, dtype=torch.float32, device="cuda")
    mis = buf[1:1 + Nx * Ny].view(Nx, Ny)
    mis.copy_(imgs[1])
    outs = [torch.full((nx, ny), -1.0, device="cuda") for _ in range(3)]
    res = plan.detect_many([imgs[0], mis, imgs[2]], outs)
    for k in range(3):
        assert res[k] is outs[k] and torch.equal(outs[k], single[k]), (case, k)
    with pytest.raises(Exception, match="same image"):
        plan.detect_many([imgs[0], imgs[1]], [outs[0], outs[0]])
    plan.close()
    ops.check_status(imgs[0].device, "detector")


def test_resize_golden(ops):
    g = load("scalars.npz")
    for k in range(int(g["resize/n"])):
        sx, sy = (int(v) for v in g["resize/%d/size" % k])
        out = ops.resize(dev(g["resize/%d/in" % k], torch.float32), sx, sy)
        assert relmax(out.cpu().numpy(), g["resize/%d/out" % k]) < 1e-6


def test_poisson_statistics(ops):
    for lam in (0.5, 4.0, 30.0, 7500.0):
        x = ops.poisson(torch.full((400, 500), lam, dtype=torch.float32, device="cuda"), seed=1234).cpu().numpy().astype(np.float64)
        n = x.size
        assert abs(x.mean() - lam) < 6 * np.sqrt(lam / n), (lam, x.mean())
        assert abs(x.var() / lam - 1) < 0.02, (lam, x.var())
        assert np.all(x == np.floor(x)) and x.min() >= 0
    a = ops.poisson(torch.full((64, 64), 20.0, dtype=torch.float32, device="cuda"), seed=7)
    b = ops.poisson(torch.full((64, 64), 20.0, dtype=torch.float32, device="cuda"), seed=7)
    c = ops.poisson(torch.full((64, 64), 20.0, dtype=torch.float32, device="cuda"), seed=8)
    assert torch.equal(a, b) and not torch.equal(a, c)
    # a draw is a function of (pixel, key) alone: the 16-byte path (four pixels per thread, the pending ones dealt out over
    # the wave) and the scalar path (a misaligned image) give the same image, whatever the size does to the last wave
    g = torch.Generator(device="cuda").manual_seed(2)
    for n in (4 * 64 * 3 + 4 * 17, 1 << 20, 4 * 5 + 3, 300007):
        lam = torch.rand(n, generator=g, device="cuda") * torch.tensor([0.0, 3.0, 40.0, 30000.0], device="cuda")[torch.randint(0, 4, (n,), generator=g, device="cuda")]
        buf = torch.empty(n + 4, dtype=torch.float32, device="cuda")
        mis = buf[1:1 + n]
        mis.copy_(lam)
        x = ops.poisson(lam.clone(), seed=99)
        ops.poisson_multi([mis], [99])
        assert torch.equal(x, mis), n


def test_bad_arguments_raise(ops):
    from paresis_amd._lib import PsxError
    with pytest.raises(PsxError):
        ops.transmit_wave(torch.zeros((4, 4), dtype=torch.complex64), 1.0, None)   # CPU tensor: no CPU path
    with pytest.raises(PsxError):
        ops.FresnelPlan(8, 8, margin=15)                                           # reflect margin larger than grid
    with pytest.raises(PsxError):
        ops.refract((2, 2), None, 1.0, (2, 2), phi_in=torch.zeros((2, 2), dtype=torch.float64, device="cuda"))


def test_deterministic_order_mode(ops):
    """psx_set_deterministic: the scatter paths that use float atomics -- the far-ray replay and
    psx_fastloop_f32 -- deposit through order-independent fixed-point accumulators; two runs are bitwise equal, the result
    stays within the oracle tolerance and within float rounding of the default mode."""
    rng = np.random.default_rng(11)
    Nx, Ny = 300, 260
    I = rng.uniform(1, 2, (Nx, Ny))
    phi = np.cumsum(rng.uniform(-30, 30, (Nx, Ny)), axis=0) + np.cumsum(rng.uniform(-30, 30, (Nx, Ny)), axis=1)
    I32 = I.astype(np.float32).astype(np.float64)
    ref, Dxr, _ = orc.fast_refraction(I32.copy(), phi.copy(), 1.0, 52.0, 1.0, 1.0)
    assert np.abs(Dxr).max() > 20                                   # nearly every ray is a far ray
    dscale = 1.0 / orc.k_refraction(52.0) / (1e-6 * 1.0) / 1e-6
    It, pt = dev(I32, torch.float32), dev(phi, torch.float64)
    run = lambda: ops.refract((Nx, Ny), None, dscale, (Nx, Ny), I_in=It, phi_in=pt)[0]
    plain = run()
    g = load("refraction.npz")
    loop_in = [dev(g["loop/" + k], torch.float32) for k in ("I", "Dx", "Dy")]
    # a larger scatter with many collisions per target for the raw loop
    Il = torch.rand((512, 384), device="cuda") + 0.5
    Dxl = (torch.rand((512, 384), device="cuda") - 0.5) * 40
    Dyl = (torch.rand((512, 384), device="cuda") - 0.5) * 40
    try:
        ops.set_deterministic(True)
        a, b = run(), run()
        assert torch.equal(a, b)
        assert relmax(a.cpu().numpy(), ref) < TOL
        assert float((a - plain).abs().max() / plain.abs().max()) < 2e-6
        outs = []
        for _ in range(2):
            I2 = torch.zeros_like(Il)
            ops.fastloop(Il, Dxl, Dyl, I2)
            outs.append(I2)
        assert torch.equal(outs[0], outs[1])
        I2 = torch.zeros(g["loop/I"].shape, dtype=torch.float32, device="cuda")
        ops.fastloop(*loop_in, I2)
        assert relmax(I2.cpu().numpy(), g["loop/out"]) < TOL
    finally:
        ops.set_deterministic(False)
    I2 = torch.zeros_like(Il)
    ops.fastloop(Il, Dxl, Dyl, I2)
    assert float((I2 - outs[0]).abs().max() / outs[0].abs().max()) < 2e-6


def test_order_independent_far_replay_needs_no_workspace_state(ops):
    """the allocation-free order-independent replay (two passes over the far lists, scratch words
    inside the caller's workspace, cleared by the tile kernel as it lists a ray).  The scratch has NO initial state: the workspace is filled with garbage before every call and
    shared between calls of different shapes (one distance, a distance batch, an energy batch, accumulate mode); every image is
    bitwise reproducible, equal to the one-distance call's, and within float rounding of the float-atomic replay."""
    import paresis_amd.ops as O
    rng = np.random.default_rng(5)
    Nx, Ny = 333, 290
    T = dev(np.stack([np.cumsum(rng.uniform(0, 2e-5, (Nx, Ny)), axis=0), rng.uniform(0, 3e-4, (Nx, Ny))]), torch.float32)
    k = 2.6e11
    mats = ops.MaterialStack(T, cphase=[-k * 6.2e-7, -k * 9.9e-8], catt=[-2 * k * 4e-9, -2 * k * 4.5e-11])
    ds = [1.0, 2.2, 3.1, 4.5]            # displacement scale: rad/pixel of phase gradient -> pixels
    plain = ops.refract_multi((Nx, Ny), mats, ds, (Nx, Ny), I0=100.0)
    torch.cuda.synchronize()

    def trash():
        for buf in O._workspaces.values():
            buf.random_(0, 255)

    try:
        ops.set_deterministic(True)
        runs = []
        for _ in range(2):
            trash()
            runs.append([t.clone() for t in ops.refract_multi((Nx, Ny), mats, ds, (Nx, Ny), I0=100.0)])
        for d in range(4):
            assert torch.equal(runs[0][d], runs[1][d]), d
            err = float((runs[0][d] - plain[d]).abs().max() / plain[d].abs().max())
            assert err < 2e-6, (d, err)
            trash()
            one = ops.refract((Nx, Ny), mats, ds[d], (Nx, Ny), I0=100.0)[0]
            assert torch.equal(one, runs[0][d]), d                # the batch is the one-distance call, bit for bit
        # accumulate mode (the energy sum of the chain): out += refraction, twice -> deterministic as well
        acc = []
        for _ in range(2):
            o = torch.full((Nx, Ny), 3.0, dtype=torch.float32, device="cuda")
            trash()
            ops.refract((Nx, Ny), mats, ds[3], (Nx, Ny), I0=100.0, out=o, add=True)
            acc.append(o)
        assert torch.equal(acc[0], acc[1])
        assert float((acc[0] - 3.0 - runs[0][3]).abs().max() / runs[0][3].abs().max()) < 1e-6
        # energy batch (psx_refract_batch_f32): 3 coefficient sets over the same maps, one launch per pass
        stacks = [mats.with_coeffs(cphase=[c * f for c in mats.cphase], catt=mats.catt) for f in (1.0, 0.8, 1.3)]
        eb = []
        for _ in range(2):
            trash()
            eb.append([t.clone() for t in ops.refract_batch((Nx, Ny), stacks, [ds[2]] * 3, (Nx, Ny), I0=[100.0, 90.0, 80.0])])
        for e in range(3):
            assert torch.equal(eb[0][e], eb[1][e]), e
            trash()
            one = ops.refract((Nx, Ny), stacks[e], ds[2], (Nx, Ny), I0=[100.0, 90.0, 80.0][e])[0]
            assert torch.equal(one, eb[0][e]), e
    finally:
        ops.set_deterministic(False)
    ops.check_status(T.device)
    # the case must have far rays at all: the float-atomic replay differs from the fixed-point one somewhere, or at least the
    # far lists are not empty (displacements beyond the 4-pixel halo)
    gx = np.abs(np.gradient(T[0].cpu().numpy().astype(np.float64) * mats.cphase[0] + T[1].cpu().numpy().astype(np.float64) * mats.cphase[1], axis=0))
    assert (gx * ds[3]).max() > 6


@pytest.mark.parametrize("case", ["masked", "black_beside_bright", "pile_up"])
@pytest.mark.parametrize("halo", [4, 8, 16])
def test_order_independent_replay_has_no_float_fallback(ops, case, halo):
    """round 4's replay took its fixed-point unit from the TARGET tile and fell back to arrival-order float
    atomics where that tile had none (all-zero window: masked inputs, the zero halves of the dark-field split) or where a share
    was too large for it (a bright tile beside a near-black one).  The unit is now one per call (2^-30 of the power of two above
    the largest staged intensity), so every far share is summed as an integer.  Each case puts MANY far shares on pixels of dark
    tiles; the image must be bitwise repeatable over runs with a garbage-filled workspace, equal to the oracle within the
    tolerance, and -- the far sums being exact integers -- independent of the ORDER of the atomics, which the test provokes by
    running once alone and once while another stream keeps the GPU busy."""
    import paresis_amd.ops as O
    rng = np.random.default_rng({"masked": 21, "black_beside_bright": 22, "pile_up": 23}[case])
    Nx, Ny = 236, 301
    I = rng.uniform(0.5, 1.5, (Nx, Ny))
    if case == "masked":
        I[:, : Ny // 2] = 0.0                                      # the left half stages nothing but zeros: tiles without a unit
        I[60:120, Ny // 2:] = 0.0
    elif case == "black_beside_bright":
        I[:, : Ny // 2] *= 1e-9                                    # near-black tiles ...
        I[:, Ny // 2:] *= 1e4                                      # ... beside bright ones: a share is 2^43 of the target tile's largest
    else:
        I *= 100.0
    if case == "pile_up":
        # a lens: every ray of a 90-pixel disc is sent to (almost) the same few pixels, far beyond any halo
        x, y = np.meshgrid(np.arange(Nx) - 118.0, np.arange(Ny) - 150.0, indexing="ij")
        phi = -0.5 * (x * x + y * y) * (np.hypot(x, y) < 90) * 0.97
        dscale = 1.0
    else:
        # rays travel 10..60 pixels towards smaller j (from the bright half into the dark one), a few pixels along i
        phi = np.cumsum(rng.uniform(-60, -10, (Nx, Ny)), axis=1) + np.cumsum(rng.uniform(-4, 4, (Nx, Ny)), axis=0)
        dscale = 1.0
    I32 = I.astype(np.float32).astype(np.float64)
    h = 1e-6
    z = dscale * orc.k_refraction(52.0) * h * h
    ref, Dxr, Dyr = orc.fast_refraction(I32.copy(), phi.copy(), z, 52.0, 1.0, 1.0)
    assert max(np.abs(Dxr).max(), np.abs(Dyr).max()) > 9           # far beyond the widest halo
    It, pt = dev(I32, torch.float32), dev(phi, torch.float64)
    run = lambda: ops.refract((Nx, Ny), None, dscale, (Nx, Ny), I_in=It, phi_in=pt)[0].clone()

    def trash():
        for buf in O._workspaces.values():
            buf.random_(0, 255)

    ops.set_refract_halo(halo)
    try:
        ops.set_deterministic(True)
        trash()
        a = run()
        trash()
        b = run()
        # the same call while a second stream hammers the memory system: another arrival order of the atomics
        side = torch.cuda.Stream()
        junk = torch.empty(1 << 26, dtype=torch.float32, device="cuda")
        with torch.cuda.stream(side):
            for _ in range(20):
                junk.add_(1.0)
        c = run()
        side.synchronize()
        ops.check_status(It.device)
        assert torch.equal(a, b) and torch.equal(a, c)
        assert relmax(a.cpu().numpy(), ref) < TOL
        if case == "pile_up":
            assert ref.max() > 50 * I32.max()                       # hundreds of shares on one pixel
        else:
            dark = ref[:, Ny // 2 - 40: Ny // 2 - 8]
            assert dark.max() > 0.1 * I32.max()                    # the dark tiles did receive far shares
    finally:
        ops.set_deterministic(False)
        ops.set_refract_halo(4)


def test_clock_probe_reads_a_plausible_shader_clock(ops):
    """psx_clock_probe: the shader-clock counter against the constant 100 MHz one over a 30 us spin on every CU -- MI355X runs
    between 0.5 and 2.5 GHz; two probes in a row agree within 15 % (the second one is under the load of the first)."""
    a, b = ops.clock_probe(), ops.clock_probe()
    assert 400.0 < a < 3000.0 and 400.0 < b < 3000.0, (a, b)
    assert abs(a - b) < 0.15 * max(a, b), (a, b)


def test_pack_counts_roundtrip_and_overflow(ops):
    """psx_pack_counts_u16 / psx_unpack_counts_u16 (the gather of the per-position stacks moves photon counts as 16-bit
    integers): exact round trip for every count 0..65534 and, through the exception table, for larger counts up to 2^24;
    vector body and scalar tail, unaligned starts, a non-zero index base; the flag is raised by a fraction, a negative,
    2^24 + 2, NaN and inf -- wherever it sits -- and by a full table, and by nothing else."""
    from paresis_amd._lib import PsxError
    g = torch.Generator(device="cuda").manual_seed(5)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    for n, off in ((65536 + 8 * 300 + 5, 0), (4099, 1), (7, 0), (1 << 20, 3)):
        src = torch.randint(0, 65535, (n + off,), generator=g, device="cuda").to(torch.float32)
        if n >= 65536:
            src[off:off + 65536] = torch.arange(65536, device="cuda", dtype=torch.float32)
        s = src[off:]
        bright = torch.randint(0, n, (min(20, n),), generator=g, device="cuda")
        s[bright] = torch.randint(65535, 1 << 24, (bright.numel(),), generator=g, device="cuda").to(torch.float32)
        s[n - 1] = 16777216.0
        exc = torch.zeros((64, 2), dtype=torch.int32, device="cuda")
        packed = torch.empty(n + off, dtype=torch.int16, device="cuda")[off:]
        cnt.zero_()
        ops.pack_counts(s, packed, 0, exc, cnt, flag)
        assert int(flag.item()) == 0 and 1 <= int(cnt.item()) <= 22
        small = s < 65535
        assert torch.equal((packed.to(torch.int32) & 0xFFFF)[small], s.to(torch.int32)[small])
        assert bool(((packed.to(torch.int32) & 0xFFFF)[~small] == 65535).all())
        back = ops.unpack_counts(packed, torch.empty(n + off, dtype=torch.float32, device="cuda")[off:], exc, cnt)
        assert torch.equal(back, s)
        # two images into one buffer: the second one's exceptions carry its index base
        both = torch.empty(2 * n, dtype=torch.int16, device="cuda")
        cnt.zero_()
        ops.pack_counts(s, both[:n], 0, exc, cnt, flag)
        ops.pack_counts(s.flip(0).contiguous(), both[n:], n, exc, cnt, flag)
        back = ops.unpack_counts(both, torch.empty(2 * n, dtype=torch.float32, device="cuda"), exc, cnt)
        assert int(flag.item()) == 0 and torch.equal(back[:n], s) and torch.equal(back[n:], s.flip(0))
        for bad in (0.5, -1.0, 16777218.0, float("nan"), float("inf"), 1e30):
            for where in (0, n // 2, n - 1):
                t = s.clone()
                t[where] = bad
                flag.zero_(); cnt.zero_()
                ops.pack_counts(t, packed, 0, exc, cnt, flag)
                assert int(flag.item()) == 1, (bad, where)
        flag.zero_(); cnt.zero_()
        ops.pack_counts(s, packed, 0, exc[:1], cnt, flag)                 # a one-entry table: full
        assert int(flag.item()) == (1 if int(cnt.item()) > 1 else 0) and int(cnt.item()) >= 2
        flag.zero_()
    # an image full of caustics (the 4096^2 XML experiment: 3.3 % of the pixels above 65534): every wave has escapes in most of
    # its pixel slots, the table takes them all -- each exactly once -- and the round trip is exact
    n = (1 << 21) + 13
    s = torch.randint(0, 60000, (n,), generator=g, device="cuda").to(torch.float32)
    hot = torch.rand(n, generator=g, device="cuda") < 0.05
    s[hot] = torch.randint(65535, 300000, (int(hot.sum()),), generator=g, device="cuda").to(torch.float32)
    exc = torch.zeros((n // 16, 2), dtype=torch.int32, device="cuda")
    packed = torch.empty(n, dtype=torch.int16, device="cuda")
    flag.zero_(); cnt.zero_()
    ops.pack_counts(s, packed, 0, exc, cnt, flag)
    assert int(flag.item()) == 0 and int(cnt.item()) == int(hot.sum())
    idx = exc[:int(cnt.item()), 0].to(torch.int64)
    assert torch.equal(torch.sort(idx).values, torch.nonzero(hot).view(-1)) and torch.equal(exc[:int(cnt.item()), 1].float(), s[idx])
    assert torch.equal(ops.unpack_counts(packed, torch.empty(n, dtype=torch.float32, device="cuda"), exc, cnt), s)
    with pytest.raises(PsxError):
        ops.pack_counts(torch.zeros(4, device="cuda"), torch.zeros(5, dtype=torch.int16, device="cuda"), 0, exc, cnt, flag)


@pytest.mark.parametrize("case", [(200, 312, 5, 2, 2), (512, 512, 25, 1, 2), (96, 130, 3, 2, 1), (640, 384, 18, 2, 2)])
def test_propagate_sources_matches_one_call_per_source(ops, case):
    """psx_fresnel_propagate_sources (the energies of a detector bin in three launches on small grids): every (source,
    distance) result is bit for bit what psx_fresnel_propagate gives for that source -- own input wave, amplitude,
    coefficients, chirp, phase and scale; more sources than one launch holds; a z = 0 pair (one-source path); both engines."""
    Nx, Ny, ns, nd, engine = case
    g = torch.Generator(device="cuda").manual_seed(Nx + ns)
    T = torch.rand((2, Nx, Ny), generator=g, device="cuda") * 1e-4
    mats = [ops.MaterialStack(T, cphase=[-3e4 * (1 + 0.1 * s), 2e4 / (1 + s)], catt=[-30.0 * (1 + s), -5.0]) for s in range(ns)]
    waves_in = [torch.complex(torch.rand((Nx, Ny), generator=g, device="cuda"), torch.rand((Nx, Ny), generator=g, device="cuda"))
                if s % 2 else None for s in range(ns)]
    amp = [1.0 + 0.25 * s for s in range(ns)]
    a = [[2.0e-12 * (1 + s) * (1 + 0.5 * d) for d in range(nd)] for s in range(ns)]
    if nd == 2 and ns == 3:
        a[1][0] = 0.0                                      # z == 0 (EXP:233-234): the batch falls back to one source at a time
    gph = [[1.0e9 * (s + 1) + 0.3 * d for d in range(nd)] for s in range(ns)]
    du = (2 * np.pi / (Nx * 1e-6), 2 * np.pi / (Ny * 1e-6))
    scale = [[0.5 + s + d for d in range(nd)] for s in range(ns)]
    plan = ops.FresnelPlan(Nx, Ny, max_dist=2, engine=engine)
    ref_w, ref_i = [], []
    for s in range(ns):
        io = [torch.empty((Nx, Ny), device="cuda") for _ in range(nd)]
        ref_w.append(plan.propagate(a[s], gph[s], du, wave_in=waves_in[s], amp=amp[s], mats=mats[s], want_wave=[True] * nd,
                                    inten_out=io, inten_scale=scale[s]))
        ref_i.append(io)
    io = [[torch.empty((Nx, Ny), device="cuda") for _ in range(nd)] for _ in range(ns)]
    got = plan.propagate_sources(a, gph, du, wave_in=waves_in, amp=amp, mats=mats, want_wave=[True] * nd, inten_out=io,
                                 inten_scale=scale)
    for s in range(ns):
        for d in range(nd):
            assert torch.equal(got[s][d], ref_w[s][d]), (s, d)
            assert torch.equal(io[s][d], ref_i[s][d]), (s, d)
    # intensity only, no complex output
    io2 = [[torch.empty((Nx, Ny), device="cuda") for _ in range(nd)] for _ in range(ns)]
    plan.propagate_sources(a, gph, du, wave_in=waves_in, amp=amp, mats=mats, want_wave=[False] * nd, inten_out=io2,
                           inten_scale=scale)
    assert all(torch.equal(io2[s][d], ref_i[s][d]) for s in range(ns) for d in range(nd))
    plan.close()


def test_accumulate_many_matches_one_call_per_image(ops):
    """psx_accumulate_many_f32: the float32 image is bit for bit the chain of psx_accumulate_sum_f32 calls, the float64 sums
    agree to rounding; more images than one launch holds, aligned and ragged sizes, with and without attenuation maps."""
    g = torch.Generator(device="cuda").manual_seed(4)
    for shape, ne, with_maps in (((300, 412), 5, True), ((127, 33), 20, True), ((256, 256), 3, False)):
        T = torch.rand((2,) + shape, generator=g, device="cuda") * 1e-3
        imgs = [torch.rand(shape, generator=g, device="cuda") * 100 for _ in range(ne)]
        mats = [ops.MaterialStack(T, catt=[-40.0 * (e + 1), -3.0]) if with_maps else None for e in range(ne)]
        weights = [20.0 + 2 * e for e in range(ne)]
        scales = [1.0 + 0.1 * e for e in range(ne)]
        for add in (False, True):
            a1 = torch.full(shape, 7.0, device="cuda")
            a2 = a1.clone()
            s1, s2 = ops.new_sums(a1.device), ops.new_sums(a1.device)
            for e in range(ne):
                ops.accumulate_sum(a1, imgs[e], s1, weights[e], scale=scales[e], mats=mats[e], add=add or e > 0)
            ops.accumulate_many(a2, imgs, s2, weights, scales=scales, mats=mats, add=add)
            assert torch.equal(a1, a2), (shape, add)
            f1, f2 = ops.fold_sums(s1), ops.fold_sums(s2)
            assert torch.allclose(f1, f2, rtol=1e-12, atol=0), (f1, f2)


def test_refract_batch_matches_one_call_per_refraction(ops):
    """psx_refract_batch_f32 (the energies of a detector bin, one launch per kernel for eight of them): every image equals
    the psx_refract_f32 result for that refraction -- bit for bit while all rays stay inside the gather halo, within float
    rounding once far rays are replayed with float atomics; uniform input and per-refraction input images; more refractions
    than one launch (and one call) holds."""
    g = torch.Generator(device="cuda").manual_seed(8)
    Nx, Ny = 210, 333
    x = torch.linspace(-1, 1, Nx, device="cuda")[:, None]
    y = torch.linspace(-1, 1, Ny, device="cuda")[None, :]
    T = torch.stack([(1 - x * x).clamp(min=0) * (1 - y * y).clamp(min=0) * 2e-4,
                     torch.rand((Nx, Ny), generator=g, device="cuda") * 2e-6]).to(torch.float32).contiguous()
    for ne, strength in ((5, 1.0), (19, 1.0), (3, 60.0)):
        mats = [ops.MaterialStack(T, cphase=[-4e4 * strength / (1 + 0.2 * e), -3e4 / (1 + e)], catt=[-50.0 * (e + 1), -10.0])
                for e in range(ne)]
        dsc = [0.8 + 0.1 * e for e in range(ne)]
        I0 = [100.0 + e for e in range(ne)]
        Iin = [torch.rand((Nx, Ny), generator=g, device="cuda") + 0.5 for _ in range(ne)]
        for with_in in (False, True):
            ref = [ops.refract((Nx, Ny), mats[e], dsc[e], (Nx, Ny), I_in=Iin[e] if with_in else None, I0=I0[e])[0] for e in range(ne)]
            got = ops.refract_batch((Nx, Ny), mats, dsc, (Nx, Ny), I_in=Iin if with_in else None, I0=I0)
            for e in range(ne):
                if strength == 1.0:
                    assert torch.equal(got[e], ref[e]), (ne, with_in, e)
                else:
                    assert float((got[e] - ref[e]).abs().max() / ref[e].abs().max()) < 2e-6, (ne, with_in, e)
    ops.check_status(T.device, "refract batch")


@pytest.mark.parametrize("case", [(200, 312, 2), (1300, 900, 1), (2100, 2100, 4), (2100, 2100, 3), (640, 384, 2)])
def test_work_queue_gives_the_same_images(ops, case):
    """psx_fresnel_plan_work_queue: line groups handed out through a queue (per-XCD atomic counters) instead of static
    shares -- small grids (fewer groups than CUs: some workgroups find the queue empty), the shared-forward rounds of a
    several-distance pass 1 (even and odd counts), a batch of sources; bit for bit the same images, repeatedly (the last
    workgroup re-arms the queue for the next launch)."""
    Nx, Ny, nd = case
    g = torch.Generator(device="cuda").manual_seed(Nx)
    T = torch.rand((1, Nx, Ny), generator=g, device="cuda") * 1e-4
    mats = ops.MaterialStack(T, cphase=[-2e4], catt=[-20.0])
    a = [3.0e-12 * (1 + 0.7 * d) for d in range(nd)]
    gph = [1.0e8 + d for d in range(nd)]
    du = (2 * np.pi / (Nx * 1e-6), 2 * np.pi / (Ny * 1e-6))
    plan = ops.FresnelPlan(Nx, Ny, max_dist=4, engine=2)

    def run():
        io = [torch.empty((Nx, Ny), device="cuda") for _ in range(nd)]
        w = plan.propagate(a, gph, du, amp=2.0, mats=mats, want_wave=[True] * nd, inten_out=io)
        return w, io

    w0, i0 = run()
    plan.work_queue(True)
    for rep in range(3):
        w1, i1 = run()
        for d in range(nd):
            assert torch.equal(w0[d], w1[d]) and torch.equal(i0[d], i1[d]), (case, rep, d)
    if nd <= 2:
        ms = [ops.MaterialStack(T, cphase=[-2e4 * (1 + s)], catt=[-20.0]) for s in range(5)]
        aa = [[v * (1 + s) for v in a] for s in range(5)]
        gg = [gph] * 5
        q = plan.propagate_sources(aa, gg, du, amp=[1.0] * 5, mats=ms)
        plan.work_queue(False)
        r = plan.propagate_sources(aa, gg, du, amp=[1.0] * 5, mats=ms)
        assert all(torch.equal(q[s][d], r[s][d]) for s in range(5) for d in range(nd))
    plan.close()


def test_darkfield_front_kernels_against_numpy(ops):
    """psx_darkfield_split_f32 / merge / repad (round 3: the array glue of fastRefractionDF, RF2:114-150,128-129,139-140, as
    kernels) against the same operations in numpy, including the DF > Nx/4 rule, both maxima and a grid that is not a
    multiple of anything."""
    rng = np.random.default_rng(11)
    Nx, Ny = 93, 70
    I = rng.uniform(1.0, 9.0, (Nx, Ny)).astype(np.float32)
    DFrad = np.where(rng.uniform(size=(Nx, Ny)) < 0.4, 0.0, rng.uniform(1e-7, 3e-6, (Nx, Ny)))
    DFrad[5, 7] = 1.0e-4                       # becomes > Nx/4 pixels: zeroed by the rule, but counts for the margin
    num, den = 3.6, 2.9 * 1e-6 * 1.02
    a, b, dfpx, prep, words = ops.darkfield_split(dev(I, torch.float32), dev(DFrad, torch.float64), num, den, Nx / 4)
    px = DFrad * num / den                     # RF2:114, the reference's order of operations
    mx0 = px.max()
    pxc = np.where(px > Nx / 4, 0.0, px)
    m0, m1 = ops.darkfield_maxima(words)
    assert m0 == mx0 and m1 == pxc.max()                       # float64, bit for bit
    df32 = pxc.astype(np.float32)
    assert np.array_equal(dfpx.cpu().numpy(), df32)
    assert np.array_equal(a.cpu().numpy(), np.where(df32 != 0, 0.0, I).astype(np.float32))
    assert np.array_equal(b.cpu().numpy(), np.where(df32 != 0, I, 0.0).astype(np.float32))
    # patch table: (half-size, 1/normalisation) of gaussian_shape(DF/2) (RF2:14-23: side round(3 sigma)*2+1, banker's rounding)
    tab = prep.view(torch.float32).view(Nx, Ny, 2).cpu().numpy()
    for (i, j) in [(0, 0), (5, 7), (40, 33), (92, 69), (17, 2)]:
        s = float(pxc[i, j]) / 2                # RF2:174: the patch side comes from the float64 width
        if s == 0:
            assert tab[i, j, 0] == 0 and tab[i, j, 1] == 1
            continue
        g = orc.create_gaussian_shape(s)
        h = (g.shape[0] - 1) // 2
        raw = np.exp(-(np.arange(-h, h + 1) ** 2) / 2.0 / s ** 2)
        assert tab[i, j, 0] == h
        assert abs(tab[i, j, 1] * raw.sum() ** 2 - 1) < 1e-6
    out = torch.empty((Nx, Ny), dtype=torch.float32, device="cuda")
    ops.darkfield_merge(out, a, b)
    assert np.array_equal(out.cpu().numpy(), I)
    src = rng.uniform(-1, 1, (Nx + 16, Ny + 16)).astype(np.float32)
    for md in (0, 3, 11):
        got = ops.repad(dev(src, torch.float32), 8, md, (Nx, Ny)).cpu().numpy()
        assert np.array_equal(got, np.pad(src[8:8 + Nx, 8:8 + Ny], md))


@pytest.mark.parametrize("max_df", [1.1, 2.4, 6.6, 9.6, 20.0, 41.0])
def test_darkfield_resplat_all_tile_shapes(ops, max_df):
    """The variable-width Gaussian re-splat (RF2:168-186) through psx_darkfield_split_f32 + psx_darkfield_blur_prepared_f32
    against the reference's literal per-source patch loop in float64, for widths that take the LDS-tiled gather (window
    <= 60 KB: R <= 14) and, beyond, the gather over bands of source rows (R = 15: one band; 31: three; 63: six, patches wider
    than the image); sources with and without dark field, zero sources, patches clipped by the image border, a grid that is
    no multiple of the tile."""
    rng = np.random.default_rng(int(max_df * 10))
    Nx, Ny = 75, 58
    I2DF = np.where(rng.uniform(size=(Nx, Ny)) < 0.2, 0.0, rng.uniform(1.0, 9.0, (Nx, Ny))).astype(np.float32)
    I2 = rng.uniform(0.0, 3.0, (Nx, Ny)).astype(np.float32)
    DFpx = np.where(rng.uniform(size=(Nx, Ny)) < 0.3, 0.0, rng.uniform(0.3, max_df, (Nx, Ny)))
    DFpx[10, 10] = max_df
    # psx_darkfield_split_f32 with scale 1 turns the width map into the float32 pixels + patch table the gather uses
    _, _, DF32, prep, words = ops.darkfield_split(dev(I2DF, torch.float32), dev(DFpx, torch.float64), 1.0, 1.0, 1e9)
    R = int(round(1.5 * max_df)) + 1
    out = ops.darkfield_blur_prepared(dev(I2DF, torch.float32), DF32, prep, dev(I2, torch.float32), R)
    ops.check_status(out.device)
    df = DF32.cpu().numpy().astype(np.float64)
    m = int(np.ceil(6 * max_df))
    ref = np.zeros((Nx + 2 * m, Ny + 2 * m))
    for i in range(Nx):
        for j in range(Ny):
            v = float(I2DF[i, j])
            if v == 0:
                continue
            if df[i, j] != 0:
                patch = orc.create_gaussian_shape(df[i, j] / 2)
                h = patch.shape[0] // 2
                ref[m + i - h:m + i + h + 1, m + j - h:m + j + h + 1] += patch * v
            else:
                ref[m + i, m + j] += v
    ref = ref[m:m + Nx, m:m + Ny] + I2
    assert relmax(out.cpu().numpy(), ref) < 2e-6, max_df


@pytest.mark.parametrize("halo", [4, 12])
@pytest.mark.parametrize("det", [False, True])
def test_refract_split_equals_two_masked_refractions(ops, halo, det):
    """psx_refract_split_f32 (round 5): ONE call refracts the sources where a map is zero and those where it is not (the
    two halves of fastRefractionDF's split, RF2:147-154) on one staging per tile.  Against two plain refractions of the
    pre-masked intensity (same thickness maps, same phase) and against the oracle on each half; mask patterns: a half plane
    (most tiles see one side only), a checkerboard of 5-pixel cells (every tile sees both) and an all-zero map."""
    rng = np.random.default_rng(77 + halo)
    Nx, Ny = 211, 187
    T = dev(np.stack([np.cumsum(rng.uniform(0, 2.5e-5, (Nx, Ny)), axis=0), rng.uniform(0, 3e-4, (Nx, Ny))]), torch.float32)
    k = 2.6e11
    delta, beta = [6.2e-7, 9.9e-8], [4e-9, 4.5e-11]
    mats = ops.MaterialStack(T, cphase=[-k * d for d in delta], catt=[-2 * k * b for b in beta])
    I = rng.uniform(0.5, 2.0, (Nx, Ny)).astype(np.float32)
    It = dev(I, torch.float32)
    dscale = 2.4
    ii, jj = np.meshgrid(np.arange(Nx), np.arange(Ny), indexing="ij")
    masks = {"half": (jj >= Ny // 2) * 1.5, "checker": (((ii // 5) + (jj // 5)) % 2) * 0.7, "none": np.zeros((Nx, Ny))}
    # oracle inputs: transmitted intensity and phase of the same float32 maps
    T64 = T.cpu().numpy().astype(np.float64)
    phi = -k * (delta[0] * T64[0] + delta[1] * T64[1])                                  # SAM:348 with the test's round k
    I_t = 3.0 * I.astype(np.float64) * np.exp(-2 * k * (beta[0] * T64[0] + beta[1] * T64[1]))   # SAM:347
    h = 1e-6
    z = dscale * orc.k_refraction(52.0) * h * h
    ops.set_refract_halo(halo)
    try:
        ops.set_deterministic(det)
        for name, m in masks.items():
            mt = dev(m, torch.float32)
            a0, a1 = ops.refract_split((Nx, Ny), mats, dscale, (Nx, Ny), mt, I_in=It, I0=3.0)
            a0, a1 = a0.clone(), a1.clone()
            for side, got in ((0, a0), (1, a1)):
                keep = (m != 0) == bool(side)
                plain, _, _ = ops.refract((Nx, Ny), mats, dscale, (Nx, Ny), I_in=dev(I * keep, torch.float32), I0=3.0)
                assert float((got - plain).abs().max()) <= 2e-6 * max(1.0, float(plain.abs().max())), (name, side)
                ref, _, _ = orc.fast_refraction(I_t * keep, phi.copy(), z, 52.0, 1.0, 1.0)
                assert relmax(got.cpu().numpy(), ref) < TOL or ref.max() == 0, (name, side)
                if not keep.any():
                    assert float(got.abs().max()) == 0.0
            # accumulate mode: both halves added to existing images
            b0, b1 = torch.full_like(a0, 2.0), torch.full_like(a1, 5.0)
            ops.refract_split((Nx, Ny), mats, dscale, (Nx, Ny), mt, I_in=It, I0=3.0, outs=[b0, b1], add=True)
            assert float((b0 - 2.0 - a0).abs().max()) < 1e-5 * max(1.0, float(a0.max()))
            assert float((b1 - 5.0 - a1).abs().max()) < 1e-5 * max(1.0, float(a1.max()))
        ops.check_status(T.device)
        with pytest.raises(Exception):
            ops.refract_split((Nx, Ny), mats, dscale, (Nx, Ny), mt, I_in=It, phi_in=torch.zeros((Nx, Ny), dtype=torch.float64, device="cuda"))
    finally:
        ops.set_deterministic(False)
        ops.set_refract_halo(4)


def test_replay_unit_from_the_callers_scale(ops):
    """psx_set_deterministic_scale (round 5): the order-independent replay takes its fixed-point unit from the caller's
    intensity scale instead of measuring the call's maximum (no memset node, no atomicMax).  Bitwise repeatable, within float
    rounding of the measured-unit result and of the oracle; a scale far too small (a share beyond 2^40 units) raises the status
    word instead of overflowing a sum."""
    from paresis_amd._lib import PsxError
    rng = np.random.default_rng(31)
    Nx, Ny = 240, 199
    I = rng.uniform(50.0, 150.0, (Nx, Ny))
    phi = np.cumsum(rng.uniform(-25, 25, (Nx, Ny)), axis=0) + np.cumsum(rng.uniform(-25, 25, (Nx, Ny)), axis=1)
    I32 = I.astype(np.float32).astype(np.float64)
    h = 1e-6
    ref, Dxr, _ = orc.fast_refraction(I32.copy(), phi.copy(), orc.k_refraction(52.0) * h * h, 52.0, 1.0, 1.0)
    assert np.abs(Dxr).max() > 15
    It, pt = dev(I32, torch.float32), dev(phi, torch.float64)
    run = lambda: ops.refract((Nx, Ny), None, 1.0, (Nx, Ny), I_in=It, phi_in=pt)[0].clone()
    with ops.deterministic(True):
        measured = run()
    with ops.deterministic(True, scale=100.0):
        a, b = run(), run()
        ops.check_status(It.device)
    assert torch.equal(a, b)
    assert relmax(a.cpu().numpy(), ref) < TOL
    assert float((a - measured).abs().max() / measured.abs().max()) < 1e-6
    assert not ops.get_deterministic()
    with ops.deterministic(True, scale=0.05):                 # shares 3000 times the scale: still inside the 2^10 x 64 of room
        c = run()
        ops.check_status(It.device)
    assert float((c - measured).abs().max() / measured.abs().max()) < 1e-6
    with ops.deterministic(True, scale=1e-20):                # 1e22 times the scale: refused, not wrapped
        run()
        with pytest.raises(PsxError):
            ops.check_status(It.device)
