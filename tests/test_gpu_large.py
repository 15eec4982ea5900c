"""BASELINE.json's full sizes on the GPU, checked through size-independent properties (the oracle would take minutes):
2048^2 / 4096^2 on the LDS Fresnel engine, 16384^2 on the rocFFT engine, plus an oracle spot check at 1024^2."""
import numpy as np
import pytest
import torch

from oracle import paresis_oracle as orc
from tests._golden import relmax

pytestmark = pytest.mark.gpu


def _membrane(N, seed=0):
    from paresis_amd import synth
    g = synth.bench_geometry(N, pointNum=seed)
    return g, torch.from_numpy(g["membrane"]).cuda()


def _stacks(ops, T, E=52.0):
    from paresis_amd import synth
    from paresis_amd.getk import k_sample
    k = k_sample(E)
    d = [synth.DELTA_BETA_52KEV[m] for m in ("CuSn", "PMMA")]
    wave = ops.MaterialStack(T, cphase=[-k * x[0] for x in d], catt=[-k * x[1] for x in d])
    rt = ops.MaterialStack(T, cphase=[-k * x[0] for x in d], catt=[-2 * k * x[1] for x in d])
    return wave, rt


@pytest.mark.parametrize("N,engine", [(2048, 2), (4096, 2), (4096, 1)])
def test_fresnel_properties_full_size(N, engine):
    from paresis_amd import ops
    from paresis_amd.getk import getk
    g, T = _membrane(N)
    wave, _ = _stacks(ops, T)
    plan = ops.FresnelPlan(N, N, max_dist=2, engine=engine)
    assert plan.engine == engine
    kk = getk(52000.0)
    h = g["pix_um"] * 1e-6
    du = (2 * np.pi / (N * h),) * 2
    a = [3.6 / (2 * kk * g["M"]), 7.2 / (2 * kk * g["M"])]
    # (1) uniform wave keeps its modulus
    u = torch.full((N, N), 2.0 + 1.0j, dtype=torch.complex64, device="cuda")
    out = plan.propagate(a[:1], [0.3], du, wave_in=u)[0]
    assert float((out.abs() - abs(2 + 1j)).abs().max()) < 2e-5
    # (2) linearity: P(alpha*w1 + w2) == alpha*P(w1) + P(w2)
    w1 = ops.transmit_wave(None, 50.0, wave)
    w2 = torch.roll(w1, (37, -111), (0, 1)).contiguous()
    p1, p2 = plan.propagate(a[:1], [0.0], du, wave_in=w1)[0], plan.propagate(a[:1], [0.0], du, wave_in=w2)[0]
    p12 = plan.propagate(a[:1], [0.0], du, wave_in=(0.5 * w1 + w2).contiguous())[0]
    assert float((p12 - (0.5 * p1 + p2)).abs().max() / p1.abs().max()) < 2e-5
    # (3) fused transmission == separate transmission; two distances in one call == one by one; |.|^2 output
    f2 = plan.propagate(a, [0.0, 0.0], du, amp=50.0, mats=wave)
    assert float((f2[0] - p1).abs().max() / p1.abs().max()) < 2e-6
    inten = torch.zeros((N, N), dtype=torch.float32, device="cuda")
    plan.propagate(a[1:], [0.0], du, wave_in=w1, want_wave=[False], inten_out=[inten])
    assert float((inten - f2[1].abs() ** 2).abs().max() / inten.max()) < 2e-6
    # (4) composition of the periodic operator: propagating by a then by a equals propagating by 2a away from the frame
    #     (the crop/re-pad between the two hops only touches a band near the border)
    q = plan.propagate(a[:1], [0.0], du, wave_in=p1)[0]
    m = 600
    err = (q - f2[1])[m:-m, m:-m].abs().max() / f2[1].abs().max()
    assert float(err) < 5e-3
    plan.close()


def test_both_engines_agree_4096():
    from paresis_amd import ops
    from paresis_amd.getk import getk
    N = 4096
    g, T = _membrane(N, 1)
    wave, _ = _stacks(ops, T)
    kk = getk(52000.0)
    h = g["pix_um"] * 1e-6
    du = (2 * np.pi / (N * h),) * 2
    a = [1.6 / (2 * kk * 141.6 / 140)]
    outs = []
    for eng in (1, 2):
        plan = ops.FresnelPlan(N, N, engine=eng)
        outs.append(plan.propagate(a, [kk * 1.6 / (141.6 / 140)], du, amp=86.6, mats=wave)[0])
        plan.close()
    assert float((outs[0] - outs[1]).abs().max() / outs[0].abs().max()) < 3e-6


@pytest.mark.parametrize("shape", [(5000, 5000), (4704, 1500), (1200, 6008), (4594, 4594), (9800, 320), (320, 12000),
                                   (12400, 320), (320, 16384), (16384, 200), (13001, 200), (18402, 136), (12279, 72), (72, 12278)])
def test_partitioned_lds_engine_matches_rocfft_and_oracle(shape):
    """Lines longer than one LDS transform (N > 4593) run as a partitioned convolution -- output blocks x kernel segments,
    partial sums in HBM, the two LDS lines coupled into one 18432-point transform -- on one or both axes: one product per
    line (5000, 4704, 6008, 4594), one segment x two blocks (9800, 12000); lines of 12280..18402 samples as ONE convolution
    of 36864 points split by a radix-2 decimation-in-frequency step over two coupled rounds (12400, 16384 on either axis,
    13001: odd period, the partner of a sample sits in the other LDS line; 18402: the longest such line, period = 18432;
    12279: the shortest, next to 12278, the longest line the two-segment partition still takes in two rounds);
    checked against the rocFFT engine (complex wave and accumulated
    intensity, 2 distances in one call) and, on a 300-pixel-wide cut, against the oracle's FFT of the same lines."""
    from paresis_amd import ops
    from paresis_amd.getk import getk
    Nx, Ny = shape
    gen = torch.Generator(device="cuda").manual_seed(Nx + Ny)
    w = torch.complex(1.0 + 0.2 * torch.randn(Nx, Ny, device="cuda", generator=gen),
                      0.2 * torch.randn(Nx, Ny, device="cuda", generator=gen)).to(torch.complex64)
    kk = getk(52000.0)
    h = 2.9e-6
    du = (2 * np.pi / (Nx * h), 2 * np.pi / (Ny * h))
    zs = (1.6, 7.2)
    a = [z / (2 * kk * 1.01) for z in zs]
    gp = [kk * z / 1.01 for z in zs]
    res = []
    for eng in (1, 2):
        plan = ops.FresnelPlan(Nx, Ny, max_dist=2, engine=eng)
        assert plan.engine == eng
        acc = [torch.full((Nx, Ny), 0.5, dtype=torch.float32, device="cuda") for _ in zs]
        waves = plan.propagate(a, gp, du, wave_in=w, want_wave=[True, True], inten_out=acc, inten_scale=[2.0, 3.0], add=True)
        only_i = [torch.empty((Nx, Ny), dtype=torch.float32, device="cuda") for _ in zs]
        plan.propagate(a, gp, du, wave_in=w, want_wave=[False, False], inten_out=only_i)
        res.append((waves, acc, only_i))
        plan.close()
    for d in range(len(zs)):
        ref_w, ref_a, ref_i = res[0][0][d], res[0][1][d], res[0][2][d]
        assert float((res[1][0][d] - ref_w).abs().max() / ref_w.abs().max()) < 3e-6, (shape, d)
        assert float((res[1][1][d] - ref_a).abs().max() / ref_a.abs().max()) < 5e-6, (shape, d)
        assert float((res[1][2][d] - ref_i).abs().max() / ref_i.abs().max()) < 5e-6, (shape, d)
        assert float((res[1][2][d] - res[1][0][d].abs() ** 2).abs().max() / ref_i.abs().max()) < 5e-6
    # oracle on a narrow strip: a separable operator acts on axis 0 alone when the wave does not vary along axis 1
    col = w[:, :1].expand(Nx, 64).contiguous()
    plan = ops.FresnelPlan(Nx, 64, engine=2)
    out = plan.propagate(a[:1], gp[:1], (du[0], 2 * np.pi / (64 * h)), wave_in=col)[0]
    plan.close()
    ref = orc.wave_propagation(col.cpu().numpy().astype(np.complex128), zs[0], 52.0, 1.01, (Nx, 64), h * 1e6)
    assert relmax(out.cpu().numpy(), ref) < 1e-5
    # ... and on axis 1 alone when it does not vary along axis 0: long lines ALONG AXIS 1 are pass 2 of the engine (strided
    # reads of the blocked intermediate; for 12 280 <= Ny <= 18 402 the <16, false, PART, PAIR, ., ., DIF> instance, config 5's
    # dominant kernel), which the axis-0 strip never reaches
    if Ny > 4593:
        row = w[:1, :].expand(64, Ny).contiguous()
        plan = ops.FresnelPlan(64, Ny, engine=2)
        out = plan.propagate(a[:1], gp[:1], (2 * np.pi / (64 * h), du[1]), wave_in=row)[0]
        plan.close()
        ref = orc.wave_propagation(row.cpu().numpy().astype(np.complex128), zs[0], 52.0, 1.01, (64, Ny), h * 1e6)
        assert relmax(out.cpu().numpy(), ref) < 1e-5, ("axis 1", shape)


@pytest.mark.parametrize("N", [1024])
def test_oracle_spot_check(N):
    from paresis_amd import ops, synth
    from paresis_amd.getk import getk, k_refraction
    g, T = _membrane(N, 2)
    wave, rt = _stacks(ops, T)
    d = [synth.DELTA_BETA_52KEV[m] for m in ("CuSn", "PMMA")]
    delta, beta = [x[0] for x in d], [x[1] for x in d]
    g64 = g["membrane"].astype(np.float64)
    kk = getk(52000.0)
    h = g["pix_um"] * 1e-6
    plan = ops.FresnelPlan(N, N)
    out = plan.propagate([3.6 / (2 * kk * g["M"])], [kk * 3.6 / g["M"]], (2 * np.pi / (N * h),) * 2, amp=86.6, mats=wave)[0]
    ref = orc.wave_propagation(orc.set_wave(np.full((N, N), 86.6 + 0j), g64, delta, beta, 52.0), 3.6, 52.0, g["M"], (N, N), g["pix_um"])
    assert relmax(out.cpu().numpy(), ref) < 1e-5
    I, phi, _ = orc.set_wave_rt(np.full((N, N), 7500.0), g64, delta, beta, 52.0, 0)
    r_ref, _, _ = orc.fast_refraction(I, phi, 7.2, 52.0, g["M"], g["pix_um"])
    r, _, _ = ops.refract((N, N), rt, 7.2 / k_refraction(52.0) / (h * g["M"]) / h, (N, N), I0=7500.0)
    ops.check_status(r.device)
    assert relmax(r.cpu().numpy(), r_ref) < 1e-5


def test_refraction_flux_and_halos_4096():
    """Flux that stays on the grid is conserved; both gather halos give the same image at full size."""
    from paresis_amd import ops
    from paresis_amd._lib import lib
    from paresis_amd.getk import k_refraction
    N = 4096
    g, T = _membrane(N, 3)
    _, rt = _stacks(ops, T)
    h = g["pix_um"] * 1e-6
    dsc = 5.2 / k_refraction(52.0) / (h * g["M"]) / h
    outs = []
    try:
        for halo in (4, 6, 8, 12, 16):
            lib().psx_refract_set_halo(halo)
            out, _, _ = ops.refract((N, N), rt, dsc, (N, N), I0=7500.0)
            outs.append(out)
    finally:
        lib().psx_refract_set_halo(4)
    assert max(float((outs[0] - o).abs().max() / outs[0].max()) for o in outs[1:]) < 2e-6
    I_in, _ = ops.transmit_rt(None, 7500.0, rt, want_phi=False)
    m = 64   # rays near the frame may leave; compare the bulk
    lost = abs(float(outs[1][m:-m, m:-m].sum(dtype=torch.float64) / I_in[m:-m, m:-m].sum(dtype=torch.float64)) - 1)
    assert lost < 1e-3


def test_refraction_distance_batch_4096():
    """The bench's refraction call (4 distances, one launch per kernel) against four one-distance calls at full size, and
    flux conservation of every image of the batch."""
    from paresis_amd import ops
    from paresis_amd.getk import k_refraction
    N = 4096
    g, T = _membrane(N, 5)
    _, rt = _stacks(ops, T)
    h = g["pix_um"] * 1e-6
    dsc = [z / k_refraction(52.0) / (h * g["M"]) / h for z in (1.6, 3.6, 5.2, 7.2)]
    outs = ops.refract_multi((N, N), rt, dsc, (N, N), I0=7500.0)
    I_in, _ = ops.transmit_rt(None, 7500.0, rt, want_phi=False)
    m = 64
    tot = float(I_in[m:-m, m:-m].sum(dtype=torch.float64))
    for d, o in zip(dsc, outs):
        single, _, _ = ops.refract((N, N), rt, d, (N, N), I0=7500.0)
        assert float((o - single).abs().max() / single.max()) < 1e-6
        assert abs(float(o[m:-m, m:-m].sum(dtype=torch.float64)) / tot - 1) < 1e-3
    ops.check_status(outs[0].device)


def test_16384_partitioned_engine_and_detector():
    """Config 5: 16384^2 study grid (detector 4096^2 x oversampling 4, PSF 1.2 px, 10 um source) resident in HBM."""
    from paresis_amd import ops
    N, ov, n = 16384, 4, 4096
    plan = ops.FresnelPlan(N, N, max_dist=1)
    assert plan.engine == 2                      # a 32797-point line does not fit one LDS transform: partitioned convolution
    u = torch.full((N, N), 1.5 + 0j, dtype=torch.complex64, device="cuda")
    inten = torch.zeros((N, N), dtype=torch.float32, device="cuda")
    plan.propagate([2e-12], [0.1], (3e5, 3e5), wave_in=u, want_wave=[False], inten_out=[inten])
    assert float((inten - 2.25).abs().max()) < 1e-4
    del u
    # a random wave through both engines at the full size (5 output blocks x 3 kernel segments per line on either axis)
    gen = torch.Generator(device="cuda").manual_seed(7)
    w = torch.complex(1.0 + 0.2 * torch.randn(N, N, device="cuda", generator=gen),
                      0.2 * torch.randn(N, N, device="cuda", generator=gen)).to(torch.complex64)
    lds = plan.propagate([2e-12], [0.1], (3e5, 3e5), wave_in=w)[0]
    plan.close()
    ref_plan = ops.FresnelPlan(N, N, max_dist=1, engine=1)
    ref = ref_plan.propagate([2e-12], [0.1], (3e5, 3e5), wave_in=w)[0]
    ref_plan.close()
    err = float((lds - ref).abs().max() / ref.abs().max())
    del w, lds, ref
    torch.cuda.empty_cache()
    assert err < 3e-6, err
    det = ops.DetectorPlan(N, N, ov, n, n, 10 * 3.6 / 141.6 / 6 * ov / 2.355, 1.2)
    img = det.detect(inten)
    assert img.shape == (n, n)
    assert float((img[40:-40, 40:-40] - 2.25 * ov * ov).abs().max()) < 1e-3     # bin SUM of a uniform image
    noisy = ops.poisson(img * 100, seed=11)
    assert abs(float(noisy.mean()) / (225.0 * ov * ov) - 1) < 1e-3


# ---------------------------------------------------------------------------------------------------------------------
# Round 2: the places where a regression could hide at the BASELINE sizes ("weak" item 1)
def test_bench_fresnel_call_4096_four_distances():
    """bench.py's exact Fresnel call -- 4096^2, ONE call to 4 distances, transmission fused, |.|^2 outputs (the only user of
    pass 1's distance-inner rounds with 4 distances) -- every one of the 4 images against the rocFFT engine (<= 3e-6), the
    image of the longest distance WHOLE against the float64 CPU restatement, and all four against it on a 64-wide strip
    (EXP:219-252)."""
    from oracle import cpu_baseline as cb
    from paresis_amd import ops, synth
    from paresis_amd.getk import getk
    N, E = 4096, 52.0
    zs = (1.6, 3.6, 5.2, 7.2)
    g, T = _membrane(N, 0)
    wave, _ = _stacks(ops, T)
    kk = getk(E * 1000)
    h = g["pix_um"] * 1e-6
    du = (2 * np.pi / (N * h),) * 2
    a = [z / (2 * kk * g["M"]) for z in zs]
    gp = [kk * z / g["M"] for z in zs]
    amp = float(np.sqrt(7500.0))
    res = []
    for eng in (2, 1):
        plan = ops.FresnelPlan(N, N, max_dist=4, engine=eng)
        assert plan.engine == eng
        outs = [torch.empty((N, N), dtype=torch.float32, device="cuda") for _ in zs]
        plan.propagate(a, gp, du, amp=amp, mats=wave, want_wave=[False] * 4, inten_out=outs)
        res.append(outs)
        plan.close()
    for d in range(4):
        err = float((res[0][d] - res[1][d]).abs().max() / res[1][d].abs().max())
        assert err < 3e-6, (d, err)
    # one WHOLE image (the longest distance) against the float64 restatement, not only the strip below
    import os
    d_b = [synth.DELTA_BETA_52KEV[m] for m in ("CuSn", "PMMA")]
    nt = max(1, min(32, (os.cpu_count() or 1) // 2))
    full = cb.fresnel_intensity(g["membrane"], [x[0] for x in d_b], [x[1] for x in d_b], amp, zs[3], E, g["M"], g["pix_um"], nt)
    assert relmax(res[0][3].cpu().numpy(), full) < 1e-5
    del full
    # strip: a wave that does not vary along axis 1 -- the separable operator then acts on axis 0 alone; 64 columns of the
    # membrane's first column, all 4 distances in one call on the LDS engine, against the float64 restatement
    strip = np.repeat(g["membrane"][:, :, :1], 64, axis=2).copy()
    Ts = torch.from_numpy(strip).cuda()
    ws, _ = _stacks(ops, Ts)
    plan = ops.FresnelPlan(N, 64, max_dist=4, engine=2)
    outs = [torch.empty((N, 64), dtype=torch.float32, device="cuda") for _ in zs]
    plan.propagate(a, gp, (du[0], 2 * np.pi / (64 * h)), amp=amp, mats=ws, want_wave=[False] * 4, inten_out=outs)
    plan.close()
    for d, z in enumerate(zs):
        ref = cb.fresnel_intensity(strip, [x[0] for x in d_b], [x[1] for x in d_b], amp, z, E, g["M"], g["pix_um"], 4)
        assert relmax(outs[d].cpu().numpy(), ref) < 1e-5, d


def test_refraction_2048_against_cpu_restatement():
    """BASELINE config 2 (2048^2, 1 distance): the refraction against the float64 C++ restatement of RF2:25-86 (about a
    second of CPU at this size), plus the Fresnel propagation of the same inputs."""
    from oracle import cpu_baseline as cb
    from paresis_amd import ops, synth
    from paresis_amd.getk import getk, k_refraction
    N, E, z = 2048, 52.0, 3.6
    g, T = _membrane(N, 4)
    wave, rt = _stacks(ops, T)
    d_b = [synth.DELTA_BETA_52KEV[m] for m in ("CuSn", "PMMA")]
    delta, beta = [x[0] for x in d_b], [x[1] for x in d_b]
    h = g["pix_um"] * 1e-6
    r, _, _ = ops.refract((N, N), rt, z / k_refraction(E) / (h * g["M"]) / h, (N, N), I0=7500.0)
    ops.check_status(r.device)
    ref = cb.refraction_intensity(g["membrane"], delta, beta, 7500.0, z, E, g["M"], g["pix_um"], 8)
    assert relmax(r.cpu().numpy(), ref) < 1e-5
    kk = getk(E * 1000)
    plan = ops.FresnelPlan(N, N, max_dist=1)
    inten = torch.empty((N, N), dtype=torch.float32, device="cuda")
    plan.propagate([z / (2 * kk * g["M"])], [kk * z / g["M"]], (2 * np.pi / (N * h),) * 2, amp=float(np.sqrt(7500.0)), mats=wave,
                   want_wave=[False], inten_out=[inten])
    plan.close()
    ref = cb.fresnel_intensity(g["membrane"], delta, beta, float(np.sqrt(7500.0)), z, E, g["M"], g["pix_um"], 8)
    assert relmax(inten.cpu().numpy(), ref) < 1e-5


@pytest.mark.parametrize("det", [False, True])
def test_refraction_4096_whole_image_against_cpu_restatement(det):
    """the HEADLINE configuration's refraction (4096^2, the longest distance of its batch, 7.2 m) as a WHOLE
    image against the float64 C++ restatement of RF2:25-86 + 198-263 (raster-order scatter, ~4 s on 8 host threads) -- with the
    narrowest and the widest gather halo (another split of the same sums between tile gathers and far-ray replay), and with the far
    rays summed by float atomics (det False) or by the order-independent fixed-point replay (det True, the Experiment default)."""
    from oracle import cpu_baseline as cb
    from paresis_amd import ops, synth
    from paresis_amd.getk import k_refraction
    N, E, z = 4096, 52.0, 7.2
    g, T = _membrane(N, 2)
    _, rt = _stacks(ops, T)
    d_b = [synth.DELTA_BETA_52KEV[m] for m in ("CuSn", "PMMA")]
    delta, beta = [x[0] for x in d_b], [x[1] for x in d_b]
    h = g["pix_um"] * 1e-6
    ref = cb.refraction_intensity(g["membrane"], delta, beta, 7500.0, z, E, g["M"], g["pix_um"], 8)
    try:
        ops.set_deterministic(det)
        for halo in (4, 8, 16):
            ops.set_refract_halo(halo)
            r, _, _ = ops.refract((N, N), rt, z / k_refraction(E) / (h * g["M"]) / h, (N, N), I0=7500.0)
            ops.check_status(r.device)
            err = relmax(r.cpu().numpy(), ref)
            assert err < 1e-5, (halo, det, err)
    finally:
        ops.set_refract_halo(4)
        ops.set_deterministic(False)


def test_refraction_16384_properties():
    """BASELINE config 5's grid through the refraction: flux that stays on the grid is conserved, the distance batch equals
    the one-distance calls, and a 512-row band equals the same band computed on its own (the gradient stencil and the gather
    are local: rows far from the band's edges cannot tell the difference)."""
    from paresis_amd import ops, synth
    from paresis_amd.getk import k_refraction, k_sample
    N = 16384
    pix_um = 6.0 / 4 / (145.2 / 141.6)
    gen = torch.Generator(device="cuda").manual_seed(5)
    # smooth random membrane-like thickness: low-resolution noise upsampled (a few um of CuSn, gradients of a few pixels)
    low = torch.rand((1, 1, N // 64, N // 64), device="cuda", generator=gen)
    Tm = torch.nn.functional.interpolate(low, size=(N, N), mode="bicubic", align_corners=False)[0, 0].clamp_(0, 1).mul_(30e-6).contiguous()
    T = torch.stack([Tm, torch.full_like(Tm, 6e-3)])
    del low, Tm
    k = k_sample(52.0)
    d = [synth.DELTA_BETA_52KEV[m] for m in ("CuSn", "PMMA")]
    rt = ops.MaterialStack(T, cphase=[-k * x[0] for x in d], catt=[-2 * k * x[1] for x in d])
    h = pix_um * 1e-6
    M = 145.2 / 141.6
    dsc = [z / k_refraction(52.0) / (h * M) / h for z in (1.6, 7.2)]
    outs = ops.refract_multi((N, N), rt, dsc, (N, N), I0=470.0)
    ops.check_status(outs[0].device)
    I_in, _ = ops.transmit_rt(None, 470.0, rt, want_phi=False)
    m = 128
    tot = float(I_in[m:-m, m:-m].sum(dtype=torch.float64))
    del I_in
    for o in outs:
        assert abs(float(o[m:-m, m:-m].sum(dtype=torch.float64)) / tot - 1) < 1e-3
    single, _, _ = ops.refract((N, N), rt, dsc[1], (N, N), I0=470.0)
    assert float((outs[1] - single).abs().max() / single.max()) < 1e-6
    del single
    # a band on its own
    r0, r1, guard = 6000, 6512, 96
    band = ops.MaterialStack(T[:, r0:r1].contiguous(), cphase=rt.cphase, catt=rt.catt)
    ob, _, _ = ops.refract((r1 - r0, N), band, dsc[1], (r1 - r0, N), I0=470.0)
    err = float((ob[guard:-guard] - outs[1][r0 + guard:r1 - guard]).abs().max() / outs[1].max())
    assert err < 1e-6, err


def test_integration_md_binding_runs():
    """The ctypes stub printed in INTEGRATION.md section 2 is executed as printed and compared with the reference's own
    fastRefraction output (tests/golden/refraction.npz)."""
    import os
    import re
    from paresis_amd import _lib
    from tests._golden import load
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"## 2\..*?```python\n(.*?)```", text, flags=re.S).group(1)
    assert "def fastRefraction" in code
    code = code.replace('ctypes.CDLL("libparesis_hip.so")', 'ctypes.CDLL(%r)' % _lib.LIB_PATH)
    ns = {}
    exec(compile(code, "INTEGRATION.md#2", "exec"), ns)
    g = load("refraction.npz")
    for k in range(int(g["n"])):
        z, E, M, pix = g["%d/params" % k]
        I = torch.from_numpy(g["%d/I" % k].astype(np.float32)).cuda()
        phi = torch.from_numpy(g["%d/phi" % k]).cuda()
        out, Dx, Dy = ns["fastRefraction"](I, phi, z, E, M, pix)
        assert relmax(out.cpu().numpy(), g["%d/v2/out" % k]) < 1e-5, k
        assert relmax(Dx.cpu().numpy(), g["%d/v2/Dx" % k]) < 1e-5
        assert np.array_equal(I.cpu().numpy() == 0, g["%d/v2/I_after" % k] == 0)        # in-place zeroing of clamped rays


@pytest.mark.parametrize("N", [3000, 2000, 1100])
def test_shared_forward_rounds_odd_and_even_distance_counts(N):
    """Pass 1 of a multi-distance call shares the forward transform between the two distances of a round (one image line per
    round at N >= 2305, two at N >= 1153, four below); 3 or 5 distances leave a half-empty last pair, 2 and 4 none: every
    image equals the one-distance call's, bit for bit (the arithmetic of a (line, distance) result does not depend on which
    round computes it)."""
    from paresis_amd import ops
    from paresis_amd.getk import getk
    g, T = _membrane(N, 6)
    wave, _ = _stacks(ops, T)
    kk = getk(52000.0)
    h = g["pix_um"] * 1e-6
    du = (2 * np.pi / (N * h),) * 2
    zs = (1.6, 3.6, 5.2, 7.2, 9.0)
    a = [z / (2 * kk * g["M"]) for z in zs]
    gp = [kk * z / g["M"] for z in zs]
    plan = ops.FresnelPlan(N, N, max_dist=5)
    singles = [plan.propagate([a[i]], [gp[i]], du, amp=50.0, mats=wave)[0] for i in range(5)]
    for nd in (2, 3, 4, 5):
        outs = plan.propagate(a[:nd], gp[:nd], du, amp=50.0, mats=wave)
        for i in range(nd):
            assert torch.equal(outs[i], singles[i]), (nd, i)
    plan.close()


@pytest.mark.parametrize("shape,nd", [((2400, 600), 3), ((1200, 1100), 2), ((700, 1300), 4), ((2310, 300), 2), ((300, 2310), 3)])
def test_fresnel_multi_distance_ragged_grids_against_the_oracle(shape, nd):
    """Non-square grids through a multi-distance call: the two axes use different transform sizes, pass 1 pairs distances
    (or not: too few line groups to fill the chip at 300 lines) -- every complex field against the float64 oracle."""
    from paresis_amd import ops
    from paresis_amd.getk import getk
    Nx, Ny = shape
    gen = torch.Generator(device="cuda").manual_seed(Nx * 7 + Ny)
    w = torch.complex(1.0 + 0.3 * torch.randn(Nx, Ny, device="cuda", generator=gen),
                      0.3 * torch.randn(Nx, Ny, device="cuda", generator=gen)).to(torch.complex64)
    kk = getk(52000.0)
    pix = 2.9
    h = pix * 1e-6
    du = (2 * np.pi / (Nx * h), 2 * np.pi / (Ny * h))
    zs = (0.8, 3.6, 7.2, 11.0)[:nd]
    M = 1.02
    plan = ops.FresnelPlan(Nx, Ny, max_dist=nd)
    outs = plan.propagate([z / (2 * kk * M) for z in zs], [kk * z / M for z in zs], du, wave_in=w)
    plan.close()
    w64 = w.cpu().numpy().astype(np.complex128)
    for z, o in zip(zs, outs):
        ref = orc.wave_propagation(w64, z, 52.0, M, (Nx, Ny), pix)
        assert relmax(o.cpu().numpy(), ref) < 1e-5, (shape, z)
