#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the Fresnel + refraction step on synthetic grids (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 4096] [--engine auto|rocfft|lds] [--no-cpu-baseline]

One STEP = one pass of the hot path over one membrane position of the 4096x4096 workload (BASELINE.json configs[2],
SURVEY.md section 8d): the membrane exit wave (2-material transmission fused into the load) is Fresnel-propagated to the
4 distances z = {1.6, 3.6, 5.2, 7.2} m (one call: transmission evaluated once) and the ray-tracing refraction (2-material
transmission fused) is run at the same 4 distances (one call: each tile's window staged once): 4 units of "Fresnel
propagation + refraction" on N^2 pixels.
value = units * N^2 * n_gpus / time  [Mpixel/s], inputs resident in HBM before the timed region.

N GPUs: one process per GPU (torchrun), each rank runs its own membrane position (seed 1000+rank): weak scaling, no
data-path collective; the final image gather over RCCL is done once after the timed region and reported as gather_ms.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DISTANCES = (1.6, 3.6, 5.2, 7.2)
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--engine", default="auto", choices=["auto", "rocfft", "lds"])
    ap.add_argument("--halo", type=int, default=4, choices=[4, 6, 8], help="refraction gather halo (speed knob)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph")
    ap.add_argument("--overlap", action="store_true",
                    help="issue the step's refractions on a second stream (no gain since the Fresnel call became two long "
                         "persistent launches; kept for experiments)")
    ap.add_argument("--no-overlap", action="store_true", help=argparse.SUPPRESS)   # former default switch, accepted and ignored
    ap.add_argument("--refract-per-distance", action="store_true",
                    help="one refraction call per distance instead of the distance batch (for comparison)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1: nccl (= RCCL over xGMI, the real thing) or gloo (rehearsal of "
                         "the multi-rank control flow with several ranks on ONE GPU; collectives then go through host copies)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    return ap.parse_args()


def main():
    a = parse()
    import torch
    import torch.distributed as td
    from paresis_amd import _lib, ops, synth
    from paresis_amd.getk import getk, k_refraction, k_sample

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d" % (a.gpus, a.gpus))
    torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
    if world > 1:
        td.init_process_group(backend=a.backend, rank=rank, world_size=world)
    dev = torch.device("cuda", torch.cuda.current_device())
    lib = _lib.lib()
    assert lib.psx_device_ok() == 1, lib.psx_last_error()
    _lib.check(lib.psx_refract_set_halo(a.halo), "psx_refract_set_halo")

    N = a.size
    E = 52.0
    geo = synth.bench_geometry(N, pointNum=rank)
    M, pix = geo["M"], geo["pix_um"]
    h = pix * 1e-6
    db = [synth.DELTA_BETA_52KEV[m] for m in geo["membrane_materials"]]
    delta, beta = [d for d, _ in db], [b for _, b in db]
    k = k_sample(E)
    T = torch.from_numpy(geo["membrane"]).to(dev)
    I0 = 30000.0 / 4
    wave_mats = ops.MaterialStack(T, cphase=[-k * d for d in delta], catt=[-k * b for b in beta])
    rt_mats = ops.MaterialStack(T, cphase=[-k * d for d in delta], catt=[-2 * k * b for b in beta])
    engine = {"auto": _lib.ENGINE_AUTO, "rocfft": _lib.ENGINE_ROCFFT, "lds": _lib.ENGINE_LDS}[a.engine]
    plan = ops.FresnelPlan(N, N, max_dist=len(DISTANCES), engine=engine)
    kk = getk(E * 1000)
    aa = [z / (2 * kk * M) for z in DISTANCES]
    gp = [kk * z / M for z in DISTANCES]
    du = (2 * np.pi / (N * h), 2 * np.pi / (N * h))
    dsc = [z / k_refraction(E) / (h * M) / h for z in DISTANCES]
    fres = [torch.empty((N, N), dtype=torch.float32, device=dev) for _ in DISTANCES]
    refr = [torch.empty((N, N), dtype=torch.float32, device=dev) for _ in DISTANCES]
    amp = float(np.sqrt(I0))

    # The two models of a step are independent and could share the GPU from two HIP streams.  That paid (+3.8 %) while a
    # Fresnel call was 9 launches with idle tails; with the distances merged into two persistent launches it does not.
    side = torch.cuda.Stream() if a.overlap else None

    def step(overlap=True):
        if side is not None and overlap:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for i in range(len(DISTANCES)):
                    ops.refract((N, N), rt_mats, dsc[i], (N, N), I0=I0, out=refr[i])
            plan.propagate(aa, gp, du, amp=amp, mats=wave_mats, want_wave=[False] * len(DISTANCES), inten_out=fres)
            torch.cuda.current_stream().wait_stream(side)
            return
        plan.propagate(aa, gp, du, amp=amp, mats=wave_mats, want_wave=[False] * len(DISTANCES), inten_out=fres)
        if a.refract_per_distance:
            for i in range(len(DISTANCES)):
                ops.refract((N, N), rt_mats, dsc[i], (N, N), I0=I0, out=refr[i])
        else:       # the call's distances in one launch per kernel, like the Fresnel call above
            ops.refract_multi((N, N), rt_mats, dsc, (N, N), I0=I0, outs=refr)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    run_step = step
    if a.graph:                 # every library call is asynchronous on the current stream, so a step captures as is
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        run_step = graph.replay
        run_step()
        barrier()
    # timed region: exactly K un-instrumented steps between barriers
    lib.psx_profile_enable(0)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run_step()
    barrier()
    dt = time.perf_counter() - t0
    ops.check_status(dev, "bench")
    # per-kernel durations for the roofline: the same K steps once more with the library recording a HIP event pair
    # around each of its launches on the launch stream (the event records cost ~4 % of a step, so they stay out of `value`)
    kern = {}
    if not a.no_kernel_timing:
        import ctypes
        lib.psx_profile_enable(1)
        for _ in range(a.steps):
            step(overlap=False)         # one stream: every kernel has the GPU to itself while its events are recorded
        barrier()
        buf = ctypes.create_string_buffer(1 << 16)
        _lib.check(lib.psx_profile_summary(buf, len(buf)), "psx_profile_summary")
        for line in buf.value.decode().splitlines():
            nm, cnt, tot = line.split()
            kern[nm] = (int(cnt), float(tot))
        lib.psx_profile_enable(0)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        dt = float(tt.item())

    # final image gather (RCCL over xGMI), outside the timed region
    gather_ms = None
    if world > 1:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        img = fres[1] if a.backend == "nccl" else fres[1].cpu()
        bucket = [torch.empty_like(img) for _ in range(world)] if rank == 0 else None
        td.gather(img, bucket, dst=0)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - t1) * 1e3

    units = len(DISTANCES)
    ms_per_step = dt / a.steps * 1e3
    value = units * N * N * world / (dt / a.steps) / 1e6

    out = {"metric": "Mpixels/s, 4096^2 Fresnel+refraction step", "value": round(value, 1), "unit": "Mpixel/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "%dx%d fp32 study grid, 1 membrane position per GPU per step, 4 propagation distances "
                                  "z={1.6,3.6,5.2,7.2} m: 4 x (Fresnel propagation + refraction) from the 2-material "
                                  "membrane thickness maps (transmission evaluated inside the step); 52 keV, "
                                  "dSM/dMO/dOD=140/1.6/3.6 m" % (N, N),
                      "units_per_step": units, "fresnel_engine": {1: "rocfft", 2: "lds"}[plan.engine],
                      "streams": 1 if side is None else 2,
                      "parallelism": "positions sharded, 1 per GPU" if world > 1 else "single GPU"}}
    if gather_ms is not None:
        out["gather_ms"] = round(gather_ms, 3)

    if rank == 0:
        P = N + 30
        nmat = 2
        # algorithmic bytes of each timed kernel per UNIT (one distance) -- DESIGN.md "Roofline accounting"; BASELINE.md
        # section 4 -- turned into bytes per launch with the launches the kernel really had (the LDS engine covers all
        # distances of a step in one launch per pass)
        alg_unit = {
            "k_refract_near": (12 + 4 * nmat) * P * P,
            "rocfft_forward": 32 * P * P,
            "rocfft_inverse": 32 * P * P,
            "k_fresnel_rows": 32 * P * P,
            "k_fresnel_cols": 32 * P * P,
        }
        alg = {nm: b * units * a.steps // kern[nm][0] for nm, b in alg_unit.items() if nm in kern}
        per = {nm: tot / cnt for nm, (cnt, tot) in kern.items()}
        step_share = {nm: tot / a.steps for nm, (cnt, tot) in kern.items()}
        out["kernel_ms_per_step"] = {nm: round(v, 4) for nm, v in sorted(step_share.items(), key=lambda kv: -kv[1])}
        out["kernel_timing"] = ("HIP event pairs recorded by the library around each launch, on a second pass of the same K "
                            "steps issued on ONE stream")
        dom = None
        for nm, v in sorted(step_share.items(), key=lambda kv: -kv[1]):
            if nm in alg:
                dom = nm
                break
        if dom is not None:
            ach = alg[dom] / (per[dom] * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(dom, N),
                               "ms_per_launch": round(per[dom], 4), "algorithmic_bytes_per_launch": alg[dom]}
            # the same figure for every kernel that has an algorithmic price (the step has three of similar weight)
            out["roofline"]["by_kernel"] = {
                nm: {"ms_per_launch": round(per[nm], 4), "achieved": round(alg[nm] / (per[nm] * 1e-3) / 1e9, 1),
                     "frac": round(alg[nm] / (per[nm] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(nm, N)}
                for nm in sorted(per) if nm in alg}
            # the bytes the kernels REALLY move (PMC counters of the committed profile) over the live launch time: what HBM
            # sees, next to the algorithmic figure above -- these kernels are bound by instruction issue, not by HBM
            for nm, e in out["roofline"]["by_kernel"].items():
                if e["traffic"]:
                    e["hbm_gbs_measured"] = round(e["traffic"] / (per[nm] * 1e-3) / 1e9, 1)
            # whole-step view with the same accounting: 4 x (64 + 12 + 4*nmat) bytes per padded pixel
            step_bytes = units * (64 + 12 + 4 * nmat) * P * P
            out["roofline"]["step_achieved"] = round(step_bytes / (dt / a.steps) / 1e9, 1)
            out["roofline"]["step_frac"] = round(step_bytes / (dt / a.steps) / 1e9 / HBM_PEAK_GBS, 4)
        if not a.no_cpu_baseline:
            out["cpu_baseline"], out["parity"] = cpu_baseline(N, geo, delta, beta, E, M, pix, I0, fres, refr)
        print(json.dumps(out))
    if world > 1:
        td.barrier()
        td.destroy_process_group()


def pmc_traffic(kernel, N):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r01_pmc_summary.json:
    FETCH_SIZE and WRITE_SIZE collected in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950).  Only valid for the configuration it was collected on (N = 4096); None otherwise."""
    if N != 4096:
        return None
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    # pass 2 (k_fresnel_rows) is the <16, false> instance of the line kernel (strided reads), pass 1 the <16, true> one
    key = {"k_fresnel_rows": "k_fresnel_lines<16, false", "k_fresnel_cols": "k_fresnel_lines<16, true",
           "k_refract_near": "k_refract_near<"}.get(kernel)
    if not key:
        return None
    for name, e in prof.items():
        if name.startswith(key):
            return e.get("hbm_bytes_per_launch")
    return None


def cpu_baseline(N, geo, delta, beta, E, M, pix, I0, fres, refr):
    """The oracle (fp64 numpy/pocketfft + scalar C loop, ONE thread = what the reference's numpy.fft + Numba @jit use)
    timed on this box's host cores for a bounded sample: ONE of the step's 4 units (z = 3.6 m) on the same inputs.
    Also returns the fp32 error of the GPU images against it."""
    import torch
    from oracle import paresis_oracle as orc
    torch.set_num_threads(1)
    g64 = geo["membrane"].astype(np.float64)
    zi = 1
    z = DISTANCES[zi]
    t0 = time.perf_counter()
    w = orc.set_wave(np.full((N, N), np.sqrt(I0) + 0j), g64, delta, beta, E)
    Fi = np.abs(orc.wave_propagation(w, z, E, M, (N, N), pix)) ** 2
    t1 = time.perf_counter()
    I, phi, _ = orc.set_wave_rt(np.full((N, N), I0), g64, delta, beta, E, 0)
    Ri, _, _ = orc.fast_refraction(I, phi, z, E, M, pix)
    t2 = time.perf_counter()
    ef = float(np.max(np.abs(fres[zi].cpu().numpy() - Fi)) / np.max(np.abs(Fi)))
    er = float(np.max(np.abs(refr[zi].cpu().numpy() - Ri)) / np.max(np.abs(Ri)))
    cb = {"value": round(N * N / (t2 - t0) / 1e6, 3), "unit": "Mpixel/s", "cores": 1, "kind": "port",
          "sample": "1 of the step's 4 units (z=3.6 m) on the same %dx%d inputs: transmission + Fresnel propagation "
                    "(%.2f s) + transmission + refraction (%.2f s), fp64, 1 thread" % (N, N, t1 - t0, t2 - t1),
          "host_cpus": os.cpu_count()}
    return cb, {"metric": "max|gpu-oracle|/max|oracle|", "fresnel": ef, "refraction": er, "tolerance": 1e-5}


if __name__ == "__main__":
    main()
