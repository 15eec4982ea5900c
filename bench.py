#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the Fresnel + refraction step on synthetic grids (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 4096] [--positions 64] [--no-cpu-baseline] ...

One STEP = one pass of the hot path over one membrane position of the 4096x4096 workload (BASELINE.json configs[2],
SURVEY.md section 8d): the membrane exit wave (2-material transmission fused into the load) is Fresnel-propagated to the
4 distances z = {1.6, 3.6, 5.2, 7.2} m (one call: transmission evaluated once) and the ray-tracing refraction (2-material
transmission fused) is run at the same 4 distances (one call: each tile's window staged once): 4 units of "Fresnel
propagation + refraction" on N^2 pixels.
value = units * N^2 * n_gpus / time  [Mpixel/s], inputs resident in HBM before the timed region.

N GPUs: one process per GPU, each rank runs its own membrane position (seed 1000+rank): weak scaling, no data-path
collective.  `python bench.py --gpus N` with no RANK in the environment starts its own N ranks (a child
`python -m torch.distributed.run`, decided before anything touches the GPU); under torchrun it is one of the ranks.

Config 4 of BASELINE.json (the 64-position membrane batch) is measured in the SAME run, after the timed steps, and reported
in the `positions_batch` object of the JSON line: the whole position loop of main.py:63-110 -- membrane synthesis with
seed(pointNum), the image-formation chain, detection, shot noise -- for 64 positions strided over the ranks, the RCCL
gather of all 64 x 2 detector stacks onto rank 0 INSIDE its timed region; rank 0 then re-computes positions it did not
own and checks the gathered images bit for bit.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DISTANCES = (1.6, 3.6, 5.2, 7.2)
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_GINST = 1228.8     # wave64 vector instructions per second, x1e9: 256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles (same guide)
PARITY_TOL = 1e-5            # BASELINE.json north_star: max|out-ref|/max|ref|


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
    ap.add_argument("--gather", default="overlap", choices=["overlap", "final"],
                    help="positions batch on several ranks: gather round by round behind the computation (default) or once at the end")
    ap.add_argument("--positions", type=int, default=64,
                    help="membrane positions of the config-4 batch measured after the timed steps (0 = skip)")
    ap.add_argument("--positions-size", type=int, default=0, help="study grid of the batch (default: --size, at most 4096)")
    ap.add_argument("--positions-trace", action="store_true",
                    help="record a HIP event after every position of the batch and report the per-position times of rank 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    return ap.parse_args()


def spawn_ranks(a):
    """`--gpus N` outside torchrun: start N ranks as a child torch.distributed.run and leave with its exit code.  Runs before
    torch.cuda / HIP is touched in this process (never exec or fork a process that has initialised the GPU)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.call(cmd, env=env))


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" not in os.environ and a.gpus > 1:
        spawn_ranks(a)
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d; launch one rank per GPU (python bench.py --gpus N starts "
                         "them itself)" % (a.gpus, world))
    import torch
    import torch.distributed as td
    from paresis_amd import _lib, ops, synth
    from paresis_amd.getk import getk, k_refraction, k_sample

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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
    if os.environ.get("PSX_WORK_QUEUE"):            # A/B of the line kernels' work queue on the step itself (default: static shares)
        plan.work_queue(True)
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
    # The device reaches its steady step time only after ~40 ms of continuous load (tools/step_ramp.py: the first ten steps
    # of a process take 1.36-1.62 ms, steps 30+ 1.25 ms), so a short timed region right after W = 5 warm-up steps reads a few
    # per cent high.  `value` stays what the contract says -- W warm-up steps, then exactly K timed steps -- and the same K
    # un-instrumented steps are timed once more here, after the event pass, as `steady`.
    barrier()
    t1 = time.perf_counter()
    for _ in range(a.steps):
        run_step()
    barrier()
    dt_steady = time.perf_counter() - t1
    cpu_dev = dev if a.backend == "nccl" else torch.device("cpu")
    ranks_seen = 1
    if world > 1:
        tt = torch.tensor([dt, dt_steady], dtype=torch.float64, device=cpu_dev)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        dt, dt_steady = float(tt[0].item()), float(tt[1].item())
        one = torch.ones(1, dtype=torch.int64, device=cpu_dev)
        td.all_reduce(one)                                  # every rank really took part in the collective
        ranks_seen = int(one.item())

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
                      "parallelism": "positions sharded, 1 per GPU" if world > 1 else "single GPU"},
           "ranks_seen": ranks_seen,
           "steady": {"ms_per_step": round(dt_steady / a.steps * 1e3, 4),
                      "value": round(units * N * N * world / (dt_steady / a.steps) / 1e6, 1),
                      "note": "the same K un-instrumented steps timed a second time, after the per-kernel event pass: the device "
                              "needs ~40 ms of load to reach its steady step time (tools/step_ramp.py)"}}

    # ---- BASELINE.json config 4: the membrane-position batch, its own timed region (all ranks take part)
    if a.positions > 0:
        import contextlib
        pn = a.positions_size or min(N, 4096)
        with contextlib.redirect_stdout(sys.stderr):     # the mirrors print like the reference; stdout carries the JSON line only
            out["positions_batch"] = {}
            for sim in ("Fresnel", "RayT"):
                try:
                    out["positions_batch"][sim] = positions_batch(a, sim, pn, rank, world, dev)
                except Exception as exc:              # the step's line above is measured already: keep it, report the batch as failed
                    import traceback
                    traceback.print_exc()
                    out["positions_batch"][sim] = {"error": "%s: %s" % (type(exc).__name__, exc)}

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
            prof = pmc_profile(N)
            ach = alg[dom] / (per[dom] * 1e-3) / 1e9
            traffic = pmc_value(prof, dom, "hbm_bytes_per_launch")
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                               "ms_per_launch": round(per[dom], 4), "algorithmic_bytes_per_launch": alg[dom],
                               "pricing": "64 B per padded pixel and propagation (BASELINE.md section 4): 32*P^2 per distance "
                                          "and line kernel; (12+4*nmat)*P^2 per refraction"}
            if traffic:
                # what HBM really sees: the PMC bytes of the committed profile over the LIVE launch time
                out["roofline"]["hbm_frac_measured"] = round(traffic / (per[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            # the same figures for every kernel that has an algorithmic price (the step has three of similar weight)
            by = {}
            for nm in sorted(per):
                if nm not in alg:
                    continue
                e = {"ms_per_launch": round(per[nm], 4), "achieved": round(alg[nm] / (per[nm] * 1e-3) / 1e9, 1),
                     "frac": round(alg[nm] / (per[nm] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "traffic": pmc_value(prof, nm, "hbm_bytes_per_launch")}
                if e["traffic"]:
                    e["hbm_gbs_measured"] = round(e["traffic"] / (per[nm] * 1e-3) / 1e9, 1)
                    e["hbm_frac_measured"] = round(e["hbm_gbs_measured"] / HBM_PEAK_GBS, 4)
                valu = pmc_value(prof, nm, "SQ_INSTS_VALU")
                if valu:
                    e["valu_ginst_s"] = round(valu / (per[nm] * 1e-3) / 1e9, 1)
                    e["valu_frac"] = round(e["valu_ginst_s"] / VALU_PEAK_GINST, 4)
                by[nm] = e
            out["roofline"]["by_kernel"] = by
            # SURVEY.md section 8(d)'s price for a distance batch on ONE input wave: the forward half is shared,
            # (32 + 32 d) P^2 for the whole Fresnel call (pre-pass + both line kernels), against 64 d P^2 above
            fres_ms = sum(step_share.get(nm, 0.0) for nm in ("k_source_transposed", "k_fresnel_cols", "k_fresnel_rows",
                                                             "k_pad_transmit", "rocfft_forward", "k_chirp_mul",
                                                             "rocfft_inverse", "k_crop_out"))
            shared_bytes = (32 + 32 * units) * P * P
            if fres_ms > 0:
                out["roofline"]["frac_shared"] = round(shared_bytes / (fres_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                out["roofline"]["frac_shared_note"] = ("whole Fresnel call (pre-pass + pass 1 + pass 2, %.4f ms) priced at "
                                                       "(32+32*d)*P^2 = %d bytes (SURVEY.md 8d, shared forward transform)"
                                                       % (fres_ms, shared_bytes))
            # whole-step view, both prices
            step_bytes = units * (64 + 12 + 4 * nmat) * P * P
            step_bytes_shared = shared_bytes + units * (12 + 4 * nmat) * P * P
            out["roofline"]["step_achieved"] = round(step_bytes / (dt / a.steps) / 1e9, 1)
            out["roofline"]["step_frac"] = round(step_bytes / (dt / a.steps) / 1e9 / HBM_PEAK_GBS, 4)
            out["roofline"]["step_frac_shared"] = round(step_bytes_shared / (dt / a.steps) / 1e9 / HBM_PEAK_GBS, 4)
            # second bound: these kernels are limited by vector instruction issue, not by HBM
            valu = pmc_value(prof, dom, "SQ_INSTS_VALU")
            if valu:
                gi = valu / (per[dom] * 1e-3) / 1e9
                busy = pmc_value(prof, dom, "SQ_BUSY_CYCLES")
                out["roofline_valu"] = {"bound": "valu", "kernel": dom, "achieved": round(gi, 1), "peak": VALU_PEAK_GINST,
                                        "unit": "G wave64-instructions/s", "frac": round(gi / VALU_PEAK_GINST, 4),
                                        "SQ_INSTS_VALU_per_launch": valu, "SQ_BUSY_CYCLES_per_launch": busy,
                                        "note": "SQ_INSTS_VALU of the committed PMC pass over the live launch time; peak = "
                                                "1024 SIMD-32 x 2.4 GHz / 2 cycles per plain wave64 instruction -- most of the "
                                                "engine's instructions are packed fp32 (v_pk_*), which hold the pipe twice as "
                                                "long, so the pipe is busier than this fraction says",
                                        "profile": prof.get("_file") if prof else None}
        rc = 0
        if not a.no_cpu_baseline:
            out["cpu_baseline"], out["parity"] = cpu_baseline(N, geo, delta, beta, E, M, pix, I0, fres, refr)
            if not out["parity"]["ok"]:
                rc = 3
        pb = out.get("positions_batch", {})
        for sim, e in pb.items():
            if e.get("check", {}).get("ok") is False:
                rc = 4
        print(json.dumps(out))
        if rc:
            sys.stderr.write("bench.py: parity check FAILED (exit %d)\n" % rc)
    else:
        rc = 0
    if world > 1:
        code = torch.tensor([rc], dtype=torch.int64, device=cpu_dev)
        td.broadcast(code, src=0)
        rc = int(code.item())
        td.barrier()
        td.destroy_process_group()
    sys.exit(rc)


def positions_batch(a, sim, N, rank, world, dev):
    """BASELINE.json config 4: `--positions` membrane positions strided over the ranks, the full loop of main.py:63-110 per
    position (membrane synthesis with seed(pointNum), chain, detection, shot noise) and the RCCL gather of every position's
    Sample/Reference stacks onto rank 0 -- round by round behind the computation (--gather overlap, default) or once at the
    end (--gather final) -- all inside the timed region (barrier + synchronize on both sides, MAX over ranks)."""
    import torch
    import torch.distributed as td
    from paresis_amd import dist, ops, synth

    P = a.positions
    exp, place = synth.bench_experiment(N, sim, noise=True, seed=7)
    mine = dist.my_positions(P, rank, world)
    cpu_dev = dev if (world == 1 or a.backend == "nccl") else torch.device("cpu")

    def position(p):
        place(p)
        return exp.computeSampleAndReferenceImages(p)[:2]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    for p in (P + 1 + rank, P + 1 + world + rank):       # untimed: plans, sphere list on the GPU, allocator pools
        position(p)
    nbins = exp._close_bins()
    dims = exp.myDetector.det_param["myDimensions"]
    stack_shape = (nbins, int(dims[0]), int(dims[1]))
    overlap = world > 1 and a.gather == "overlap"
    if overlap and sim == "Fresnel":
        exp._plan().work_queue(True)      # the transfer's copy kernels share the GPU with the line kernels from here on

    mode = {"overlap": overlap}

    def gather_all(positions_fn):
        """Computes this rank's positions and brings every position's stacks to rank 0: round by round behind the computation
        (dist.PositionGatherer) or in one gather at the end.  Returns (gathered, seconds of computation issued + finished)."""
        t1 = time.perf_counter()
        if mode["overlap"]:
            gat = dist.PositionGatherer(P, rank, world, to_host=False, shape=stack_shape)
            for p in mine:
                gat.add(p, positions_fn(p))
            ev = torch.cuda.Event()
            ev.record()
            ev.synchronize()              # this rank's computation (not the rounds in flight on RCCL's stream)
            tc = time.perf_counter() - t1
            return gat.finish(), tc
        results = {p: positions_fn(p) for p in mine}
        torch.cuda.synchronize()
        tc = time.perf_counter() - t1
        return dist.gather_positions(results, P, rank, world, to_host=False), tc

    if world > 1:
        # the whole gather path once at its full size, untimed: communicator and peer connections, the packing kernels, and
        # the caching allocator's blocks for the staging buffers (a first-time hipMalloc of ~2 GiB on rank 0 would land in
        # the timed region)
        warm = position(P + 1 + rank)
        try:
            gather_all(lambda p: warm)
        except Exception:                  # the overlapped form failed where every rank fails alike (an API it lacks): one gather at the end
            if not mode["overlap"]:
                raise
            import traceback
            traceback.print_exc()
            mode["overlap"] = False
            gather_all(lambda p: warm)
        del warm
    # The interpreter's cyclic garbage collector would otherwise run a full collection somewhere in the first positions
    # (hundreds of thousands of objects allocated by the set-up above: ~40-60 ms of host time with the GPU idle -- the
    # one-off stall DESIGN.md round 1 could not explain).  Collect now, then keep the survivors out of later collections.
    import gc
    gc.collect()
    gc.freeze()
    barrier()
    t0 = time.perf_counter()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(len(mine) + 1)] if a.positions_trace else None
    if marks:
        marks[0].record()
    done = [0]

    def timed_position(p):
        out = position(p)
        if marks:
            done[0] += 1
            marks[done[0]].record()
        return out

    gathered, t_comp = gather_all(timed_position)
    barrier()
    dt = time.perf_counter() - t0
    exp.resolve_mean_energy()
    ops.check_status(dev, "positions batch")
    times = torch.tensor([dt, t_comp, dt - t_comp], dtype=torch.float64, device=cpu_dev)
    per_rank = [times]
    if world > 1:
        per_rank = [torch.empty_like(times) for _ in range(world)]
        td.all_gather(per_rank, times)
    if rank != 0:
        if world > 1:                                    # rank 0 recomputes a few positions meanwhile: wait for it
            td.barrier()
        return None
    dt_max = max(float(t[0]) for t in per_rank)
    n_det = int(exp.myDetector.det_param["myDimensions"][0])
    res = {"positions": P, "n_gpus": world, "ranks_seen": len(per_rank), "study_grid": N, "detector": n_det,
           "ms_total": round(dt_max * 1e3, 3), "ms_per_position": round(dt_max * 1e3 / P, 4),
           "positions_per_s": round(P / dt_max, 1), "Mpixel_per_s": round(P * N * N / dt_max / 1e6, 1),
           "per_rank_compute_ms": [round(float(t[1]) * 1e3, 3) for t in per_rank],
           "gather_ms": round(max(float(t[2]) for t in per_rank) * 1e3, 3),
           "gathered_bytes": int(sum(v[0].numel() + v[1].numel() for v in gathered.values()) * 4),
           "gather_wire_bytes": dist.last_gather.get("wire_bytes"), "gather_packed_u16": dist.last_gather.get("packed"),
           "gather_overlapped": bool(dist.last_gather.get("overlapped")),
           "timed_region": "synthesis + chain + detection + shot noise of every position + the gather onto rank 0 (images stay "
                           "in rank 0's HBM)", "backend": a.backend if world > 1 else None}
    if marks:
        res["per_position_ms_rank0"] = [round(marks[i].elapsed_time(marks[i + 1]), 3) for i in range(len(mine))]
    # rank 0 re-computes positions it did not own (every position when it is alone) and compares with what arrived
    others = [p for p in range(P) if p % world != 0] if world > 1 else list(range(P))
    sample = sorted(set(others[:2] + others[-1:])) if others else []
    worst, equal = 0.0, True
    for p in sample:
        S, R = position(p)
        for mine_t, got in ((S, gathered[p][0]), (R, gathered[p][1])):
            got = got.to(mine_t.device)
            equal = equal and bool(torch.equal(mine_t, got))
            worst = max(worst, float((mine_t - got).abs().max() / got.abs().max()))
    torch.cuda.synchronize()
    # Fresnel chain: no float atomics anywhere -> bit for bit; ray tracing: far rays are replayed with float atomics in
    # arbitrary order, which the Poisson draw may turn into a different count at a few pixels
    res["check"] = {"positions_recomputed_on_rank0": sample, "bit_equal": equal, "max_rel_diff": worst,
                    "ok": bool(equal) if sim == "Fresnel" else bool(worst < 1e-3)}
    if world > 1:
        td.barrier()
    return res


def pmc_profile(N):
    """The newest committed rocprofv3 PMC summary for study grid N (profiles/rNN_pmc_summary[_N].json: FETCH_SIZE, WRITE_SIZE
    and the SQ counters each collected in its own run; hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024, FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950).  Only valid for the configuration it was collected on."""
    import glob
    suffix = "" if N == 4096 else "_%d" % N
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_summary%s.json" % suffix)))
    for f in reversed(files):
        try:
            prof = json.load(open(f))["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        prof["_file"] = os.path.relpath(f, ROOT)
        return prof
    return None


def pmc_value(prof, kernel, field):
    if not prof:
        return None
    for name, e in prof.items():
        if not isinstance(e, dict):
            continue
        if kernel in ("k_fresnel_rows", "k_fresnel_cols"):
            # pass 2 (k_fresnel_rows) is the <R3, false, .> instance of the line kernel (strided reads), pass 1 <R3, true, .>
            if not name.startswith("k_fresnel_lines<"):
                continue
            contig = name.split(",")[1].strip()
            if contig != ("true" if kernel == "k_fresnel_cols" else "false"):
                continue
            return e.get(field)
        if name.startswith(kernel + "<") or name == kernel:
            return e.get(field)
    return None


def cpu_baseline(N, geo, delta, beta, E, M, pix, I0, fres, refr):
    """The build's CPU restatement (oracle/cpu_baseline.{cpp,py}: float64, the reference's algorithm and operation order,
    golden-checked) timed on this box's host cores on ALL 4 units of the step, at 1 thread (the stand-in for the reference:
    numpy.fft and a non-parallel Numba @jit are single-threaded) and at all cores (OpenMP + pocketfft workers).  Also the
    fp32 error of all 8 GPU images against it; a failure makes bench.py exit non-zero."""
    import torch
    from oracle import cpu_baseline as cb
    torch.set_num_threads(1)
    ncpu = os.cpu_count() or 1
    try:
        ncpu = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    tf1, tr1, F, R = cb.time_units(geo["membrane"], delta, beta, I0, DISTANCES, E, M, pix, 1)
    # all cores: hardware threads and half of them (one per physical core where SMT is on), the faster one is reported
    best = None
    for nt in sorted({ncpu, max(1, ncpu // 2)}, reverse=True):
        tf, tr, _, _ = cb.time_units(geo["membrane"], delta, beta, I0, DISTANCES, E, M, pix, nt)
        if best is None or tf + tr < best[0] + best[1]:
            best = (tf, tr, nt)
    tfa, tra, nta = best
    ef = [float(np.max(np.abs(fres[i].cpu().numpy() - F[i])) / np.max(np.abs(F[i]))) for i in range(len(DISTANCES))]
    er = [float(np.max(np.abs(refr[i].cpu().numpy() - R[i])) / np.max(np.abs(R[i]))) for i in range(len(DISTANCES))]
    units = len(DISTANCES)
    cb1 = units * N * N / (tf1 + tr1) / 1e6
    cba = units * N * N / (tfa + tra) / 1e6
    out = {"value": round(cb1, 3), "unit": "Mpixel/s", "cores": 1, "kind": "port",
           "sample": "all %d units of one step on the same %dx%d inputs: %d x (transmission + Fresnel propagation) %.2f s + "
                     "%d x (transmission + refraction) %.2f s, fp64, 1 thread" % (units, N, N, units, tf1, units, tr1),
           "all_cores": {"value": round(cba, 3), "unit": "Mpixel/s", "cores": nta,
                         "sample": "the same %d units with OpenMP + pocketfft workers on %d threads (of %d hardware threads; the "
                                   "faster of all / half): Fresnel %.2f s, refraction %.2f s" % (units, nta, ncpu, tfa, tra)},
           "host_cpus": os.cpu_count(), "cpu_model": cb.cpu_model(),
           "implementation": "oracle/cpu_baseline.cpp (C++17 -O3 x86-64-v3, OpenMP) + pocketfft via scipy.fft for the 2-D FFTs"}
    ok = max(ef + er) <= PARITY_TOL
    return out, {"metric": "max|gpu-cpu_fp64|/max|cpu_fp64| per image, all %d distances" % units, "fresnel": ef,
                 "refraction": er, "tolerance": PARITY_TOL, "ok": bool(ok)}


if __name__ == "__main__":
    main()
