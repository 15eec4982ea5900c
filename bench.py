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
METRIC = "Mpixels/s, 4096^2 Fresnel+refraction step"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--engine", default="auto", choices=["auto", "rocfft", "lds"])
    ap.add_argument("--halo", type=int, default=0, choices=[0, 4, 6, 8, 12, 16],
                    help="refraction gather halo of the headline step (a speed knob; 0 = 6 with the order-independent replay, 4 with "
                         "float atomics: the measured optima of this step, gpurun_out/r6s8)")
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
    ap.add_argument("--no-warm-batch", action="store_true", help="positions batch: skip the second, warm-start measurement")
    ap.add_argument("--positions-trace", action="store_true",
                    help="record a HIP event after every position of the batch and report the per-position times of rank 0")
    ap.add_argument("--spinup-ms", type=float, default=80.0,
                    help="GPU load before the W warm-up steps so that the clocks have ramped (0 = none; reported as `spinup`)")
    ap.add_argument("--debug-switch", action="append", default=[], metavar="NAME=VALUE",
                    help="diagnostic A/B switch of the library (psx_debug_switch; repeatable); echoed in the line as `debug_switches`")
    ap.add_argument("--work-queue", action="store_true",
                    help="A/B: the line kernels' work queue on the headline step itself (default: static shares)")
    ap.add_argument("--no-reproducible-batch", action="store_true",
                    help="positions batch: skip the third measurement (ray tracing with the order-independent far-ray replay)")
    ap.add_argument("--deterministic-step", action="store_true", help="accepted and ignored: the default since round 6")
    ap.add_argument("--float-atomics-step", action="store_true",
                    help="the headline step's far rays through float atomics (the library's default mode) instead of the order-independent "
                         "replay the Experiment class runs by default; `other_far_ray_mode` on the line is then the replay")
    ap.add_argument("--no-replay-scale", action="store_true",
                    help="with --deterministic-step: the replay's unit from each call's measured maximum (memset node + atomicMax) "
                         "instead of the caller's intensity scale")
    ap.add_argument("--sink", type=int, default=0,
                    help="positions batch on several ranks: the rank that receives every position's images (default 0, which also "
                         "owns position 0 and its extra images: with another sink the straggler and the receiver are two GPUs)")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` object (BASELINE configs 1, 2 and 5)")
    ap.add_argument("--configs", default="512,2048,16384", help="study grids of the `configs` object")
    ap.add_argument("--no-config-graph", action="store_true", help="configs: time the small grids launch by launch only")
    ap.add_argument("--no-config-parity", action="store_true",
                    help="`configs` entries without their parity leg (profiling runs: the strips' small launches would mix into the "
                         "kernel statistics)")
    ap.add_argument("--no-whole-image-parity", action="store_true",
                    help="`configs` entries above 2048^2: skip the whole-image comparison with the float64 restatement (strips only)")
    ap.add_argument("--only-configs", action="store_true",
                    help="run ONLY the `configs` entries (no headline step, no positions batch): the command rocprofv3 profiles for "
                         "the config-5 variant of the driver's line (tools/collect_profiles.sh _cfg5 --only-configs --configs 16384)")
    ap.add_argument("--emulate-world", type=int, default=8,
                    help="one GPU only: after the positions batch, run the share of rank 0 and of the last rank of a world of this "
                         "size, each in a fresh process, and put the predicted speed-up on the line (0: skip)")
    ap.add_argument("--emulate-rank", type=int, default=-1, help=argparse.SUPPRESS)     # child mode of --emulate-world
    ap.add_argument("--emulate-sim", default="Fresnel", help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    return ap.parse_args()


def spawn_ranks(a):
    """`--gpus N` outside torchrun: start N ranks as a child torch.distributed.run and leave with its exit code.  Runs before
    torch.cuda / HIP is touched in this process (never exec or fork a process that has initialised the GPU)."""
    import socket
    if any(k.startswith("ROCPROFILER_") or k.startswith("ROCPROF_") for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        # the profiler's preloaded library has initialised the GPU in THIS process already: starting the ranks from it is the
        # fork/exec of a GPU process the pool forbids.  Profile one rank: python3 bench.py (no --gpus).
        raise SystemExit("bench.py: --gpus %d under rocprofv3 would start the ranks from a process that has initialised the GPU; "
                         "profile a single rank instead" % a.gpus)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # Process count of `python bench.py --gpus N`: this parent (never touches the GPU) + the launcher's agent + N ranks = N + 2
    # (N + 1 when the driver starts torch.distributed.run itself); each rank uses exactly one GPU.
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        # the ranks are gone and rank 0 may never have printed its line: say so in a form the driver can parse.  A negative
        # code is the signal that killed the launcher (e.g. -9: the box's process / memory guard, which leaves no message).
        print(json.dumps({"error": "bench.py --gpus %d: the rank launcher exited with code %d before a result line was complete "
                                   "(rank tracebacks, if any, are on stderr; exit 5 = a rank failed inside the positions batch, "
                                   "6 = a peer was lost / bounded wait expired, 3 / 4 = parity or gather check failed, negative = "
                                   "killed by that signal)" % (a.gpus, rc),
                          "rc": rc, "n_gpus": a.gpus, "metric": METRIC, "value": None, "unit": "Mpixel/s",
                          "processes_started": a.gpus + 2}), flush=True)
    sys.exit(rc if rc >= 0 else 128 - rc)


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

    from paresis_amd import dist as pdist
    rank = int(os.environ.get("RANK", "0"))
    # one GPU per rank under nccl (= RCCL): more ranks than GPUs is refused HERE, before any collective (RCCL itself would
    # only say "Duplicate GPU detected" from the first barrier, or hang); gloo may rehearse several ranks on one GPU
    torch.cuda.set_device(pdist.local_device(a.backend if world > 1 else "gloo", rank, world))
    dev = torch.device("cuda", torch.cuda.current_device())
    if world > 1:
        import datetime
        kw = {"device_id": dev} if a.backend == "nccl" else {}      # RCCL otherwise guesses the device from the global rank
        td.init_process_group(backend=a.backend, rank=rank, world_size=world,
                              timeout=datetime.timedelta(seconds=pdist.timeout_s()), **kw)
    lib = _lib.lib()
    assert lib.psx_device_ok() == 1, lib.psx_last_error()
    a.deterministic_step = not a.float_atomics_step
    if a.halo == 0:
        a.halo = 6 if a.deterministic_step else 4
    _lib.check(lib.psx_refract_set_halo(a.halo), "psx_refract_set_halo")
    for item in a.debug_switch:
        name, _, val = item.partition("=")
        ops.debug_switch(name, int(val) if val else 1)
    if a.deterministic_step:
        ops.set_deterministic(True)
        if not a.no_replay_scale:
            ops.set_deterministic_scale(30000.0 / 4)      # as Experiment._replay_scale: the incident intensity per study pixel
    if a.emulate_rank >= 0:
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):
            res = emulated_rank_share(a, a.emulate_sim, a.positions_size or min(a.size, 4096), a.emulate_rank, a.emulate_world, dev)
        print(json.dumps(res))
        sys.exit(0)
    if a.only_configs:
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):
            cfgs = run_configs(a, dev)
        print(json.dumps({"configs": cfgs, "debug_switches": ops.debug_switches_active()}))
        sys.exit(0 if all(e.get("parity", {}).get("ok", True) is not False for e in cfgs.values()) else 3)

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
    if a.work_queue:            # A/B of the line kernels' work queue on the step itself (default: static shares)
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

    # Clock spin-up, disclosed in the line (`spinup`): a process needs ~40 ms of continuous load before its steps reach their
    # steady time (tools/step_ramp.py: steps 2-10 of a fresh process take 1.36-1.62 ms, steps 30+ 1.25 ms), and W = 5 warm-up
    # steps are 6 ms.  The same step is therefore run for >= --spinup-ms of GPU time first (host-timed, coarse), THEN come the
    # W warm-up steps and the K timed steps of the contract, back to back: no real run of this path is 30 ms long.
    # `value_cold`: the contract's W warm-up + K timed steps FIRST, on a process whose clocks have not
    # ramped -- what rounds 1-2 reported as `value`, kept so that rounds stay comparable.  (With --spinup-ms 0 it is `value`.)
    cold = None
    if a.spinup_ms > 0:
        for _ in range(a.warmup):
            step()
        barrier()
        tc0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        barrier()
        cold = time.perf_counter() - tc0
    spin = {"ms": 0.0, "steps": 0}
    if a.spinup_ms > 0:
        ts = time.perf_counter()
        while (time.perf_counter() - ts) * 1e3 < a.spinup_ms or spin["steps"] < 2:
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            spin["steps"] += 4
        spin["ms"] = round((time.perf_counter() - ts) * 1e3, 1)
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
    # what this box sustains: the shader clock right behind the steady steps (psx_clock_probe: a 30 us spin on every CU).  The
    # boxes of the pool differ by 12 % in step time with the same library; the probe says how much of a
    # run's time is the box's clock.  Outside every timed region.
    try:
        shader_mhz = round(ops.clock_probe(), 1)
    except Exception as exc:                               # diagnostics only: never fails the run
        shader_mhz = None
        sys.stderr.write("bench: clock probe failed: %s\n" % exc)
    # ... and K steps in the OTHER far-ray mode (order-independent fixed-point replay <-> float atomics), so that the cost of
    # reproducible sums is on the line whichever mode `value` was measured in
    other_mode = {"far_rays": "float atomics" if ops.get_deterministic() else "order-independent fixed-point replay"}
    if not a.graph:
        ops.set_deterministic(not ops.get_deterministic())
        for _ in range(a.warmup):
            step()
        barrier()
        t2 = time.perf_counter()
        for _ in range(a.steps):
            step()
        barrier()
        dt_other = time.perf_counter() - t2
        ops.check_status(dev, "bench (other far-ray mode)")
        ops.set_deterministic(not ops.get_deterministic())
        step()                      # the images the parity leg checks are the headline mode's
        barrier()
        other_mode.update(ms_per_step=round(dt_other / a.steps * 1e3, 4),
                          value=round(len(DISTANCES) * N * N * world / (dt_other / a.steps) / 1e6, 1),
                          vs_steady_pct=round((dt_other / dt_steady - 1.0) * 100.0, 2),
                          note="the same K steps with the far rays summed the other way, timed right after `steady` (this rank's clock)")
    cpu_dev = dev if a.backend == "nccl" else torch.device("cpu")
    ranks_seen = 1
    if cold is None:
        cold = dt
    if world > 1:
        tt = torch.tensor([dt, dt_steady, cold], dtype=torch.float64, device=cpu_dev)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        dt, dt_steady, cold = float(tt[0].item()), float(tt[1].item()), float(tt[2].item())
        one = torch.ones(1, dtype=torch.int64, device=cpu_dev)
        td.all_reduce(one)                                  # every rank really took part in the collective
        ranks_seen = int(one.item())

    units = len(DISTANCES)
    ms_per_step = dt / a.steps * 1e3
    value = units * N * N * world / (dt / a.steps) / 1e6

    out = {"metric": METRIC, "value": round(value, 1), "unit": "Mpixel/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "%dx%d fp32 study grid, 1 membrane position per GPU per step, 4 propagation distances "
                                  "z={1.6,3.6,5.2,7.2} m: 4 x (Fresnel propagation + refraction) from the 2-material "
                                  "membrane thickness maps (transmission evaluated inside the step); 52 keV, "
                                  "dSM/dMO/dOD=140/1.6/3.6 m" % (N, N),
                      "units_per_step": units, "fresnel_engine": {1: "rocfft", 2: "lds"}[plan.engine], "refraction_halo": a.halo,
                      "streams": 1 if side is None else 2,
                      "parallelism": "positions sharded, 1 per GPU" if world > 1 else "single GPU"},
           "ranks_seen": ranks_seen,
           "value_cold": round(units * N * N * world / (cold / a.steps) / 1e6, 1),
           "cold": {"ms_per_step": round(cold / a.steps * 1e3, 4),
                    "note": "the same W warm-up + K timed steps run FIRST, before the spin-up (clocks not ramped): the protocol of "
                            "rounds 1-2, for comparison across rounds"},
           "debug_switches": ops.debug_switches_active(),
           "env_switches": {k: v for k, v in sorted(os.environ.items()) if k.startswith("PSX_")},
           "far_rays": "order-independent fixed-point replay" if ops.get_deterministic() else "float atomics",
           "other_far_ray_mode": other_mode,
           "spinup": {"ms": spin["ms"], "steps": spin["steps"],
                      "note": "the same step run untimed BEFORE the W warm-up steps until the clocks have ramped (--spinup-ms)"},
           "shader_clock_mhz": shader_mhz,
           "shader_clock_note": "measured right behind the `steady` steps with the shader-clock and the 100 MHz counters (psx_clock_probe); "
                                "the step time of one library scales with it from box to box",
           "steady": {"ms_per_step": round(dt_steady / a.steps * 1e3, 4),
                      "value": round(units * N * N * world / (dt_steady / a.steps) / 1e6, 1),
                      "note": "the same K un-instrumented steps timed a second time, after the per-kernel event pass: the device "
                              "needs ~40 ms of load to reach its steady step time (tools/step_ramp.py)"}}

    # the library's own default (float atomics, measured unit) for what follows: the position batches set the mode through the
    # Experiment class, the `configs` entries say which mode they ran in
    ops.set_deterministic(False)
    ops.set_deterministic_scale(0.0)
    # ---- BASELINE.json config 4: the membrane-position batch, its own timed region (all ranks take part)
    if a.positions > 0:
        import contextlib
        pn = a.positions_size or min(N, 4096)
        with contextlib.redirect_stdout(sys.stderr):     # the mirrors print like the reference; stdout carries the JSON line only
            out["positions_batch"] = {}
            # ray tracing twice: far rays replayed with float atomics (the faster form: last bits depend on the arrival order, a
            # Poisson draw may flip) and with the order-independent replay (bit-reproducible on any number of GPUs, main.run's
            # `reproducible` switch; measured cost in DESIGN.md section 4.3)
            # (round 5: the order-independent replay is the Experiment class's default -- `RayT` is that, `RayT_float_atomics` the
            # opt-out, exp_dict['reproducible'] = False)
            for key in ("Fresnel", "RayT", "RayT_float_atomics"):
                sim = key.split("_")[0]
                if key == "RayT_float_atomics" and a.no_reproducible_batch:
                    continue
                try:
                    out["positions_batch"][key] = positions_batch(a, sim, pn, rank, world, dev, reproducible=not key.endswith("_float_atomics"))
                    if world == 1 and a.emulate_world > 1 and key in ("Fresnel", "RayT") and a.positions >= a.emulate_world:
                        out["positions_batch"][key]["rank_share"] = emulate_world(a, sim, out["positions_batch"][key])
                except Exception as exc:
                    import traceback
                    traceback.print_exc()
                    if world > 1:
                        # the ranks may be out of step: no further collective can be trusted on this communicator, and a rank
                        # waiting in one would hang until the timeout.  Leave now, non-zero (torchrun then ends the others).
                        sys.stderr.write("bench.py: rank %d: positions batch failed (%s: %s) -- leaving\n" % (rank, type(exc).__name__, exc))
                        sys.stderr.flush()
                        os._exit(6 if isinstance(exc, pdist.DistError) else 5)
                    # one rank: the step's line above is measured already: keep it, report the batch as failed
                    out["positions_batch"][key] = {"error": "%s: %s" % (type(exc).__name__, exc)}

    # ---- BASELINE.json configs 1, 2 and 5 on the same line (rank 0 of a one-rank run only: they are single-GPU configurations)
    if rank == 0 and world == 1 and not a.no_configs and N == 4096:
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):
            try:
                out["configs"] = run_configs(a, dev)
            except Exception as exc:
                import traceback
                traceback.print_exc()
                out["configs"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    if rank == 0:
        P = N + 30
        nmat = 2
        # TWO price lists.  `frac` uses SURVEY.md section 8(d)'s price for what this step IS -- a batch
        # of d distances on ONE input wave, forward transform shared: a 2-D FFT is two axis passes of 16 B per padded pixel, so a
        # line kernel (one axis: its share of the forward transform + d inverses) is priced at (16 + 16 d) P^2 per launch of d
        # distances and the whole Fresnel call at (32 + 32 d) P^2.  `frac_per_propagation` prices every distance as a
        # propagation of its own (64 P^2 each, 32 d P^2 per line-kernel launch).
        shared_step = {      # bytes per STEP under the shared-forward price
            "k_refract_near": units * (12 + 4 * nmat) * P * P,
            "rocfft_forward": 32 * P * P,
            "rocfft_inverse": units * 32 * P * P,
            "k_fresnel_rows": (16 + 16 * units) * P * P,
            "k_fresnel_cols": (16 + 16 * units) * P * P,
        }
        perprop_step = {     # bytes per STEP with every distance a full propagation
            "k_refract_near": units * (12 + 4 * nmat) * P * P,
            "rocfft_forward": units * 32 * P * P,
            "rocfft_inverse": units * 32 * P * P,
            "k_fresnel_rows": units * 32 * P * P,
            "k_fresnel_cols": units * 32 * P * P,
        }
        alg = {nm: b * a.steps // kern[nm][0] for nm, b in shared_step.items() if nm in kern}          # per launch
        alg_pp = {nm: b * a.steps // kern[nm][0] for nm, b in perprop_step.items() if nm in kern}
        per = {nm: tot / cnt for nm, (cnt, tot) in kern.items()}
        step_share = {nm: tot / a.steps for nm, (cnt, tot) in kern.items()}
        # an event pair costs a few microseconds: below ~30 us per launch the figures rank the kernels but are not durations
        out["kernel_ms_per_step"] = {nm: round(v, 4) for nm, v in sorted(step_share.items(), key=lambda kv: -kv[1]) if per[nm] >= 0.03}
        short = {nm: round(v, 4) for nm, v in sorted(step_share.items(), key=lambda kv: -kv[1]) if per[nm] < 0.03}
        if short:
            out["kernel_ms_short_launches"] = dict(short, note="launches under 30 us: event-pair overhead is of the same order, "
                                                               "a ranking, not durations")
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
            ach_pp = alg_pp[dom] / (per[dom] * 1e-3) / 1e9
            traffic = pmc_value(prof, dom, "hbm_bytes_per_launch")
            out["roofline"] = {"bound": "valu-issue", "priced_against": "hbm",
                               "bound_note": "frac = ALGORITHMIC bytes (price list below) / launch time / HBM peak, as the contract asks; "
                                             "the kernel itself moves about a third of the per-propagation bytes (hbm_frac_measured) "
                                             "and is limited by vector-instruction issue (roofline_valu)",
                               "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                               "traffic_from": prof.get("_file") if prof else None,
                               # a stale profile does not go unnoticed: were the counters collected on the kernel
                               # sources this run was built from?  (null: a summary older than the stamp)
                               "traffic_sources_match": (prof.get("_csrc_sha1") == csrc_sha1()) if prof and prof.get("_csrc_sha1") else None,
                               "traffic_note": "HBM bytes per launch from the committed rocprofv3 PMC passes of the same command on the "
                                               "same build (profiles/), not re-measured by this run",
                               "ms_per_launch": round(per[dom], 4), "algorithmic_bytes_per_launch": alg[dom],
                               "pricing": "SURVEY 8(d), distance batch on one input wave (forward transform shared): (16+16*d)*P^2 per "
                                          "line-kernel launch of d distances, (32+32*d)*P^2 per Fresnel call; (12+4*nmat)*P^2 per refraction",
                               "frac_per_propagation": round(ach_pp / HBM_PEAK_GBS, 4),
                               "achieved_per_propagation": round(ach_pp, 1),
                               "algorithmic_bytes_per_launch_per_propagation": alg_pp[dom],
                               "pricing_per_propagation": "64 B per padded pixel and propagation (BASELINE.md section 4): 32*P^2 per "
                                                          "distance and line kernel (the figure rounds 1-3 printed as `frac`)"}
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
                     "frac_per_propagation": round(alg_pp[nm] / (per[nm] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
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
            # the whole Fresnel call (pre-pass + both line kernels) under the shared price
            fres_ms = sum(step_share.get(nm, 0.0) for nm in ("k_source_transposed", "k_fresnel_cols", "k_fresnel_rows",
                                                             "k_pad_transmit", "rocfft_forward", "k_chirp_mul",
                                                             "rocfft_inverse", "k_crop_out"))
            shared_bytes = (32 + 32 * units) * P * P
            if fres_ms > 0:
                out["roofline"]["fresnel_call_frac"] = round(shared_bytes / (fres_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                out["roofline"]["fresnel_call_note"] = ("whole Fresnel call (pre-pass + pass 1 + pass 2, %.4f ms) priced at "
                                                        "(32+32*d)*P^2 = %d bytes" % (fres_ms, shared_bytes))
            # whole-step view, both prices
            step_bytes_pp = units * (64 + 12 + 4 * nmat) * P * P
            step_bytes = shared_bytes + units * (12 + 4 * nmat) * P * P
            out["roofline"]["step_achieved"] = round(step_bytes / (dt / a.steps) / 1e9, 1)
            out["roofline"]["step_frac"] = round(step_bytes / (dt / a.steps) / 1e9 / HBM_PEAK_GBS, 4)
            out["roofline"]["step_frac_per_propagation"] = round(step_bytes_pp / (dt / a.steps) / 1e9 / HBM_PEAK_GBS, 4)
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
        for e in (out.get("configs") or {}).values():
            if isinstance(e, dict) and e.get("parity", {}).get("ok") is False:
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


def positions_batch(a, sim, N, rank, world, dev, reproducible=False):
    """BASELINE.json config 4: `--positions` membrane positions strided over the ranks, the full loop of main.py:63-110 per
    position (membrane synthesis with seed(pointNum), chain, detection, shot noise) and the RCCL gather of every position's
    Sample/Reference stacks onto rank 0 -- round by round behind the computation (--gather overlap, default) or once at the
    end (--gather final) -- all inside the timed region, MAX over ranks.  Measured twice: `cold` (barrier + synchronize on
    both sides, host clock: the GPU is idle when the clock starts, so the first positions run at ramping clocks -- at 8 ranks a
    rank's whole share is ~10 ms) and `warm` (the GPU kept under load up to the start: untimed positions, an on-stream
    all-reduce instead of a host barrier, HIP events around the region).

    Every decision that changes WHICH collectives follow is taken by all ranks together (dist.agree_on_overlap); an exception
    once collectives are in flight ends the process (exit 5 / 6) instead of being retried on the same communicator."""
    import torch
    import torch.distributed as td
    from paresis_amd import dist, ops, synth

    P = a.positions
    exp, place = synth.bench_experiment(N, sim, noise=True, seed=7)
    mine = dist.my_positions(P, rank, world)
    sink = a.sink if 0 <= a.sink < world else 0
    # ray tracing: far rays through the order-independent replay, so that a position's images are the same bits on 1 GPU and on
    # 8 (float atomics let a last bit flip a Poisson draw); the Fresnel chain has no float atomics anywhere
    det_mode = sim == "RayT" and reproducible
    exp.exp_dict['reproducible'] = bool(reproducible)     # the Experiment sets the library's mode around its ray-tracing chain
    exp.exp_dict['sharedZeroStacks'] = True               # as main.run: the loop only reads (packs) what a position returns
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
    exp.reserve_outputs(len(mine) + 2)        # as main.run: every position's images are kept, none of them costs a hipMalloc
    overlap = world > 1 and a.gather == "overlap"
    gat = None
    if overlap:
        # buffers of every round on every rank first (rank 0 alone holds the ~2 GiB of receive buckets: the likeliest failure
        # is one-sided), then ONE collective decision
        gat = dist.PositionGatherer(P, rank, world, dst=sink, to_host=False, shape=stack_shape)
        inject = os.environ.get("PSX_BENCH_INJECT", "") == "prepare:%d" % rank
        overlap = dist.agree_on_overlap(gat, inject_failure=inject)
        if not overlap:
            gat = None
    if overlap and sim == "Fresnel":
        exp._plan().work_queue(True)      # the transfer's copy kernels share the GPU with the line kernels from here on

    def gather_all(positions_fn):
        """Computes this rank's positions and brings every position's stacks to rank 0: round by round behind the computation
        (dist.PositionGatherer) or in one gather at the end.  Returns (gathered, seconds of computation issued + finished)."""
        t1 = time.perf_counter()
        if gat is not None:
            gat.reset()
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
        return dist.gather_positions(results, P, rank, world, dst=sink, to_host=False), tc

    if world > 1:
        # the whole gather path once at its full size, untimed: communicator and peer connections, the packing kernels, the
        # caching allocator's blocks.  Collectives are in flight from here on: no fallback, a failure ends the process.
        warm = position(P + 1 + rank)
        if os.environ.get("PSX_BENCH_INJECT", "") == "warmup:%d" % rank:
            raise dist.DistError("injected failure in the gather warm-up (test)")
        gather_all(lambda p: warm)
        del warm
    # The interpreter's cyclic garbage collector would otherwise run a full collection somewhere in the first positions
    # (hundreds of thousands of objects allocated by the set-up above: ~40-60 ms of host time with the GPU idle -- the
    # one-off stall DESIGN.md round 1 could not explain).  Collect now, then keep the survivors out of later collections.
    import gc
    gc.collect()
    gc.freeze()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(len(mine) + 1)] if a.positions_trace else None
    done = [0]

    def timed_position(p):
        out = position(p)
        if marks:
            done[0] += 1
            marks[done[0]].record()
        return out

    # ---- cold: the contract's bracket (barrier + synchronize, host clock)
    barrier()
    t0 = time.perf_counter()
    if marks:
        marks[0].record()
    gathered, t_comp = gather_all(timed_position)
    barrier()
    dt = time.perf_counter() - t0
    exp.resolve_mean_energy()
    ops.check_status(dev, "positions batch")
    per_pos_ms = [round(marks[i].elapsed_time(marks[i + 1]), 3) for i in range(len(mine))] if marks else None
    marks = None
    # the sink re-computes positions it did not own (every position when it is alone) and compares with what arrived -- here,
    # so that the cold run's images can be released before the warm run asks the allocator for its own
    n_gathered = len(gathered)
    check = None
    if rank == sink:
        others = [p for p in range(P) if p % world != sink] if world > 1 else list(range(P))
        sample = sorted(set(others[:2] + others[-1:])) if others else []
        worst, equal, ndiff, npix, peak = 0.0, True, 0, 0, 1.0
        for p in sample:
            S, R = position(p)
            for mine_t, got in ((S, gathered[p][0]), (R, gathered[p][1])):
                got = got.to(mine_t.device)
                equal = equal and bool(torch.equal(mine_t, got))
                worst = max(worst, float((mine_t - got).abs().max() / got.abs().max()))
                ndiff += int((mine_t != got).sum())
                npix += got.numel()
                peak = max(peak, float(got.abs().max()))
            del S, R
        torch.cuda.synchronize()
        check = (sample, equal, worst, ndiff, npix, peak)
    del gathered
    # ---- warm: same work, the GPU loaded up to the first timed kernel.  ~60 ms of untimed positions are queued (the host runs
    # ahead of the GPU), then an all-reduce ON THE STREAM lines the ranks up without idling the GPUs (td.barrier() would
    # synchronise the host and the device), then the start event; the end event follows the last unpack kernel and one more
    # on-stream all-reduce, so that every rank's interval covers the slowest rank's work.
    warm_ms = None
    if not a.no_warm_batch:
        tick = torch.zeros(1, dtype=torch.float32, device=cpu_dev)
        n_pre = max(4, int(60.0 / max(0.3, dt * 1e3 / max(1, len(mine)))))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        for i in range(n_pre):
            position(P + 1 + rank)
        if world > 1:
            td.all_reduce(tick)
        e0.record()
        g2, _ = gather_all(position)
        if world > 1:
            td.all_reduce(tick)
        e1.record()
        barrier()
        warm_ms = e0.elapsed_time(e1)
        del g2
        exp.resolve_mean_energy()
        ops.check_status(dev, "positions batch (warm)")
    gc.unfreeze()
    times = torch.tensor([dt, t_comp, dt - t_comp, (warm_ms or 0.0) * 1e-3], dtype=torch.float64, device=cpu_dev)
    per_rank = [times]
    if world > 1:
        per_rank = [torch.empty_like(times) for _ in range(world)]
        td.all_gather(per_rank, times)
    if rank != sink:
        if world > 1:                                    # the sink recomputes a few positions meanwhile: wait for it
            td.barrier()
        if rank == 0 and world > 1:                      # rank 0 prints the line: it receives the sink's report
            box = [None]
            td.broadcast_object_list(box, src=sink, device=cpu_dev)
            return box[0]
        elif world > 1 and sink != 0:
            td.broadcast_object_list([None], src=sink, device=cpu_dev)
        return None
    dt_max = max(float(t[0]) for t in per_rank)
    n_det = int(exp.myDetector.det_param["myDimensions"][0])
    res = {"positions": P, "n_gpus": world, "ranks_seen": len(per_rank), "study_grid": N, "detector": n_det,
           "ms_total": round(dt_max * 1e3, 3), "ms_per_position": round(dt_max * 1e3 / P, 4),
           "positions_per_s": round(P / dt_max, 1), "Mpixel_per_s": round(P * N * N / dt_max / 1e6, 1),
           "clock": "cold: barrier + synchronize on both sides, host clock, GPU idle at the start",
           "per_rank_compute_ms": [round(float(t[1]) * 1e3, 3) for t in per_rank],
           "gather_ms": round(max(float(t[2]) for t in per_rank) * 1e3, 3),
           "gathered_bytes": int(n_gathered * 2 * stack_shape[0] * stack_shape[1] * stack_shape[2] * 4),
           "gather_wire_bytes": dist.last_gather.get("wire_bytes"), "gather_packed_u16": dist.last_gather.get("packed"),
           "gather_overlapped": bool(dist.last_gather.get("overlapped")),
           "timed_region": "synthesis + chain + detection + shot noise of every position + the gather onto the sink rank (images "
                           "stay in its HBM; those that crossed as 16-bit counts are widened to float32 when they are read)",
           "backend": a.backend if world > 1 else None}
    if warm_ms is not None:
        wmax = max(float(t[3]) for t in per_rank)
        res["warm"] = {"ms_total": round(wmax * 1e3, 3), "ms_per_position": round(wmax * 1e3 / P, 4),
                       "positions_per_s": round(P / wmax, 1), "Mpixel_per_s": round(P * N * N / wmax / 1e6, 1),
                       "clock": "the same batch once more with the GPU under load up to the start (untimed positions, on-stream "
                                "all-reduce, no host synchronisation), HIP events around the region, MAX over ranks"}
    if per_pos_ms:
        res["per_position_ms_rank0"] = per_pos_ms
    sample, equal, worst, ndiff, npix, peak = check
    # Fresnel chain: no float atomics anywhere -> bit for bit; ray tracing with the order-independent replay: bit for bit too;
    # with --float-atomics far rays are summed in arrival order, which the Poisson draw may turn into a different count
    # (float atomics: a far ray's last bit may flip a Poisson draw -- by one count when the inversion branch drew it, by a few
    # standard deviations when a rejection sampler accepts another uniform: 1.4e-5 to 2e-3 of the peak have been seen.  The
    # images are still draws of the same distributions: at most a handful of pixels may differ, each by a few sqrt(counts).)
    res["check"] = {"positions_recomputed_on_sink": sample, "bit_equal": equal, "max_rel_diff": worst,
                    "pixels_differing": ndiff, "pixels_compared": npix,
                    "ok": bool(equal) if (sim == "Fresnel" or det_mode)
                          else bool(ndiff <= max(8, npix // 100000) and worst <= 8.0 / max(1.0, peak) ** 0.5)}
    res["far_rays"] = "order-independent fixed-point replay" if det_mode else ("float atomics" if sim == "RayT" else None)
    res["sink_rank"] = sink
    if world > 1:
        td.barrier()
        if sink != 0:
            td.broadcast_object_list([res], src=sink, device=cpu_dev)
    return res


def emulated_rank_share(a, sim, N, r, W, dev):
    """What rank r of a world of W would do for the --positions batch, on THIS GPU alone, in this (fresh) process: the same
    set-up positions_batch() runs untimed (two positions: plans, sphere list, allocator pools; the packing kernels once), then
    between two host synchronisations with the GPU idle at the start -- the `cold` clock of positions_batch -- its positions
    r, r + W, ... (position 0 with its Propag / White extras on rank 0), each packed into its 16-bit wire buffer as
    dist.PositionGatherer does before it hands the round to RCCL (the sink keeps what arrives in that form: nothing to add for
    rank 0).  No collective is issued: what is missing from a real rank's time is the
    transfer itself (priced on the line from the wire bytes) and whatever the copy kernels of RCCL cost the chain
    (+3 % per position with the Fresnel plan's work queue).  A prediction, not a measurement."""
    import gc

    import torch
    from paresis_amd import dist, ops, synth

    P = a.positions
    exp, place = synth.bench_experiment(N, sim, noise=True, seed=7)
    exp.exp_dict['sharedZeroStacks'] = True               # as positions_batch
    mine = dist.my_positions(P, r, W)

    def position(p):
        place(p)
        return exp.computeSampleAndReferenceImages(p)[:2]

    for p in (P + 1 + r, P + 1 + W + r):
        position(p)
    nbins = exp._close_bins()
    dims = exp.myDetector.det_param["myDimensions"]
    per_img = nbins * int(dims[0]) * int(dims[1])
    wires = [dist._CountsWire(2 * per_img, dev) for _ in mine]
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    if sim == "Fresnel":
        exp._plan().work_queue(True)          # as positions_batch does while transfers share the GPU

    trace = {"host_ms": [], "ev": []}

    def share(traced=False):
        th = time.perf_counter()
        for t, p in enumerate(mine):
            S, R = position(p)
            wires[t].head.zero_()
            wires[t].pack(S, 0, flag)
            wires[t].pack(R, per_img, flag)
            if traced:                        # host time spent issuing this position, and an event behind its last kernel
                now = time.perf_counter()
                trace["host_ms"].append(round((now - th) * 1e3, 3))
                if os.environ.get("PSX_EMULATE_ALLOC_TRACE"):
                    trace.setdefault("allocs", []).append(torch.cuda.memory_stats()["num_device_alloc"])
                th = now
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                trace["ev"].append(ev)

    # the gather warm-up of positions_batch (world > 1): the packing / unpacking kernels once, untimed
    S, R = position(P + 1 + r)
    wires[0].head.zero_()
    wires[0].pack(S, 0, flag)
    wires[0].unpack()
    del S, R
    exp.reserve_outputs(len(mine) + 2)
    gc.collect()
    gc.freeze()
    torch.cuda.synchronize()
    # the barrier of a real run: the GPU is idle when the clock starts, the host thread has been polling (RCCL's barrier spins:
    # a sleeping thread would add the wake-up of its core to the first position).  The length of the idle does not matter
    # between 0.1 and 100 ms (gpurun_out/r5s5).
    t_idle = time.perf_counter() + float(os.environ.get("PSX_EMULATE_IDLE_MS", "2")) * 1e-3
    while time.perf_counter() < t_idle:
        pass
    ev0 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    share(traced=True)
    torch.cuda.synchronize()
    cold = time.perf_counter() - t0
    gpu_done = [round(ev0.elapsed_time(e), 3) for e in trace["ev"]]
    exp.resolve_mean_energy()
    ops.check_status(dev, "emulated rank share")
    # warm: the same share with the GPU kept under load up to the start (as positions_batch's `warm`)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(max(4, int(0.06 / max(3e-4, cold / max(1, len(mine)))))):
        position(P + 1 + r)
    e0.record()
    share()
    e1.record()
    torch.cuda.synchronize()
    gc.unfreeze()
    return {"rank": r, "world": W, "positions": mine, "cold_ms": round(cold * 1e3, 3), "warm_ms": round(e0.elapsed_time(e1), 3),
            "packed_ok": int(flag.item()) == 0, "wire_bytes_per_position": int(wires[0].bytes.numel()),
            "cold_host_issue_ms_per_position": trace["host_ms"], "cold_gpu_done_at_ms": gpu_done, "allocs": trace.get("allocs")}


def emulate_world(a, sim, one_gpu):
    """positions_batch.<sim>.rank_share: no 8-GPU node was ever available to this repository, so the share
    of rank 0 (position 0's extras, the sink's unpacking) and of the last rank of --emulate-world ranks is run on ONE GPU, each
    in a fresh process (a real rank is one: cold clocks, cold caches), and the 8-GPU time is PREDICTED as the slower share +
    the part of the gather that cannot hide behind the computation, priced at SURVEY.md section 5's xGMI figure (one
    point-to-point link per peer, ~153 GB/s each, all 7 into the sink at once)."""
    import subprocess
    W = a.emulate_world
    out = {"world": W, "note": "prediction, not a measurement: each share run alone on one GPU in a fresh process; no collective issued",
           "context": "the children are started by this process AFTER its own positions batch: the parent still holds its HIP context "
                      "and HBM, and the card's clocks may be warm from the batch just finished (ADVICE r5) -- `cold_ms` is cold for the "
                      "child's caches and allocator, not necessarily for the clocks; every predicted_* figure is a model built on them"}
    if any(k.startswith("ROCPROFILER_") or k.startswith("ROCPROF_") for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        out["skipped"] = "under rocprofv3 (a child process would be profiled as well)"
        return out
    shares = {}
    for r in sorted({0, W - 1}):
        cmd = [sys.executable, os.path.abspath(__file__), "--emulate-rank", str(r), "--emulate-world", str(W), "--emulate-sim", sim,
               "--positions", str(a.positions), "--size", str(a.size), "--positions-size", str(a.positions_size)]
        pr = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        if pr.returncode != 0:
            out["error"] = "rank %d: rc %d: %s" % (r, pr.returncode, pr.stderr.strip().splitlines()[-1:] or "")
            return out
        shares[r] = json.loads(pr.stdout.strip().splitlines()[-1])
    XGMI_LINK_GBS = 153.0
    per_pos = shares[0]["wire_bytes_per_position"]
    rounds = (a.positions + W - 1) // W
    gather_all_ms = rounds * per_pos / XGMI_LINK_GBS / 1e6          # one peer's whole contribution over its own link
    gather_last_ms = per_pos / XGMI_LINK_GBS / 1e6                  # what the overlapped gather leaves exposed: the last round
    slow_cold = max(v["cold_ms"] for v in shares.values())
    slow_warm = max(v["warm_ms"] for v in shares.values())
    out.update(rank0_ms=shares[0]["cold_ms"], rank0_warm_ms=shares[0]["warm_ms"],
               **{"rank%d_ms" % (W - 1): shares[W - 1]["cold_ms"], "rank%d_warm_ms" % (W - 1): shares[W - 1]["warm_ms"]},
               one_gpu_64_ms=one_gpu["ms_total"], one_gpu_64_warm_ms=one_gpu.get("warm", {}).get("ms_total"),
               gather_ms_last_round=round(gather_last_ms, 3), gather_ms_if_fully_exposed=round(gather_all_ms, 3),
               xgmi_link_GBs_assumed=XGMI_LINK_GBS, wire_bytes_per_position=per_pos, packed_ok=all(v["packed_ok"] for v in shares.values()))
    out["predicted_speedup_%d" % W] = round(one_gpu["ms_total"] / (slow_cold + gather_last_ms), 2)
    out["predicted_speedup_%d_gather_exposed" % W] = round(one_gpu["ms_total"] / (slow_cold + gather_all_ms), 2)
    if one_gpu.get("warm"):
        out["predicted_speedup_%d_warm" % W] = round(one_gpu["warm"]["ms_total"] / (slow_warm + gather_last_ms), 2)
    return out


def run_configs(a, dev):
    """The other single-GPU configurations of BASELINE.json on the driver's line (`configs`), same process, after the headline:
    config 1's grid (512^2 = detector 256 x oversampling 2, 1 distance, with Detector.detection), config 2 (2048^2, 1 distance)
    and config 5 (16384^2 = detector 4096 x oversampling 4, 4 distances, with Detector.detection -- pad, source blur, bin-sum,
    PSF 1.2 px, DET:79-119 -- of all 8 images INSIDE the timed step).  Each entry: ms per step, Mpixel/s (units x N^2 / t),
    step_frac under the price list of `roofline` (+ the detector's bytes) and the fp32 error against the float64
    restatement -- whole images where that takes seconds, a 64-column strip at 16384^2 (a separable operator acts on axis 0
    alone when nothing varies along axis 1)."""
    import types

    import torch
    from oracle import cpu_baseline as cb
    from oracle import paresis_oracle as orc
    from paresis_amd import _lib, ops, synth
    from paresis_amd.Samples.getMembraneFromFile import getMembraneSegmentedFromFile
    from paresis_amd.getk import getk, k_refraction, k_sample

    # the `configs` entries run the library's default far-ray mode (float atomics), as in every earlier round: their figures stay
    # comparable; the headline carries both modes
    ops.set_deterministic(False)
    ops.set_deterministic_scale(0.0)
    E, I0 = 52.0, 7500.0
    db = [synth.DELTA_BETA_52KEV[m] for m in ("CuSn", "PMMA")]
    delta, beta = [d for d, _ in db], [b for _, b in db]
    k, kk = k_sample(E), getk(E * 1000)
    smp = types.SimpleNamespace(myMeanSphereRadius=15.0, myNbOfLayers=2)
    out = {}
    for N in [int(v) for v in a.configs.split(",") if v]:
        ov = 4 if N >= 8192 else 2
        zs = (3.6,) if N <= 2048 else DISTANCES
        detect = N != 2048
        M = 145.2 / 141.6
        pix = 6.0 / ov / M
        h = pix * 1e-6
        geom, _ = getMembraneSegmentedFromFile(smp, N, N, pix * 140.0 / 141.6, 0, 6000.0, stacked=True)
        T = geom[2]
        wave_mats = ops.MaterialStack(T, cphase=[-k * d for d in delta], catt=[-k * b for b in beta])
        rt_mats = ops.MaterialStack(T, cphase=[-k * d for d in delta], catt=[-2 * k * b for b in beta])
        plan = ops.FresnelPlan(N, N, max_dist=len(zs))
        aa = [z / (2 * kk * M) for z in zs]
        gp = [kk * z / M for z in zs]
        du = (2 * np.pi / (N * h),) * 2
        dsc = [z / k_refraction(E) / (h * M) / h for z in zs]
        fres = [torch.empty((N, N), dtype=torch.float32, device=dev) for _ in zs]
        refr = [torch.empty((N, N), dtype=torch.float32, device=dev) for _ in zs]
        n = N // ov
        sig_src, sig_psf = 10.0 * 3.6 / 141.6 / 6.0 * ov / 2.355, 1.2
        # gather halo of the refraction tiles (a speed knob, the images do not depend on it): at oversampling 4 the
        # displacements are four times as many pixels as at oversampling 2 and the 8-pixel halo pays (16384^2: tile kernel
        # 4.3 -> 5.5 ms, far-ray replay 5.7 -> 3.0 ms); the headline's 4-pixel halo elsewhere
        # (round 4: picked by measurement -- ops.tune_refract_halo times the step's own refraction call with each halo)
        halo, halo_ms = ops.tune_refract_halo(lambda: ops.refract_multi((N, N), rt_mats, dsc, (N, N), I0=I0, outs=refr),
                                              halos=(4, 6, 8, 12, 16) if ov >= 4 else (4, 6, 8))
        det = ops.DetectorPlan(N, N, ov, n, n, sig_src, sig_psf) if detect else None
        dets = [torch.empty((n, n), dtype=torch.float32, device=dev) for _ in range(2 * len(zs))] if detect else []
        amp = float(np.sqrt(I0))

        def step():
            plan.propagate(aa, gp, du, amp=amp, mats=wave_mats, want_wave=[False] * len(zs), inten_out=fres)
            ops.refract_multi((N, N), rt_mats, dsc, (N, N), I0=I0, outs=refr)
            if det is not None:                    # up to four images per launch, each bit for bit what its own call would write
                det.detect_many(fres + refr, dets)

        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        one = time.perf_counter() - t0
        K = int(min(200, max(3, 0.15 / one)))
        for _ in range(int(min(K, max(1, 0.05 / one)))):     # clocks
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        ops.check_status(dev, "configs %d" % N)
        # one more step with the library's event pairs: where the step's time goes
        import ctypes
        lib = _lib.lib()
        lib.psx_profile_enable(1)
        step()
        torch.cuda.synchronize()
        buf = ctypes.create_string_buffer(1 << 16)
        _lib.check(lib.psx_profile_summary(buf, len(buf)), "psx_profile_summary")
        lib.psx_profile_enable(0)
        kern, kshort = {}, {}
        for line in buf.value.decode().splitlines():
            nm, cnt, tot = line.split()
            # an event pair costs a few microseconds: launches under 30 us are a ranking, not durations
            (kern if float(tot) / max(1, int(cnt)) >= 0.03 else kshort)[nm] = round(float(tot), 4)
        # Small grids: a step is seven launches of 6-45 us, so the gaps between dispatches are a large part
        # of it.  The whole step is captured once into a hipGraph (every library call is asynchronous on the current stream and
        # allocates nothing once its plans and kernel spectra exist) and replayed: `ms_hipgraph_replay` beside `ms` (launch by
        # launch).  Measured (gpurun_out/r5s12): the replay is not faster on this stack -- 512^2 0.101 against 0.097 ms, 2048^2
        # 0.130 against 0.125 -- a graph's kernel nodes are dispatched one by one with the same barrier packets between them.
        dt_graph = None
        if N <= 2048 and not a.no_config_graph:
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    step()
                for _ in range(int(min(K, max(3, 0.05 / one)))):
                    g.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(K):
                    g.replay()
                torch.cuda.synchronize()
                dt_graph = (time.perf_counter() - t0) / K
                ops.check_status(dev, "configs %d (graph)" % N)
            except Exception as exc:                       # capture refused (a plan would have had to allocate): report, keep going
                sys.stderr.write("configs %d: hipGraph capture failed: %s\n" % (N, exc))
        P = N + 30
        nmat = 2
        # both price lists of `roofline`: the distance batch on one input wave (shared forward transform) and one full
        # propagation per distance; with one distance they coincide
        step_bytes = (32 + 32 * len(zs)) * P * P + len(zs) * (12 + 4 * nmat) * P * P
        step_bytes_pp = len(zs) * (64 + 12 + 4 * nmat) * P * P
        if detect:
            det_bytes = 2 * len(zs) * int(4 * (N + 30 * ov) ** 2 * (1 + 1.0 / ov ** 2))
            step_bytes += det_bytes
            step_bytes_pp += det_bytes
        e = {"workload": "%dx%d fp32 study grid (detector %d x oversampling %d), %d distance(s)%s" %
                         (N, N, n, ov, len(zs), ", Detector.detection of all %d images in the step" % (2 * len(zs)) if detect else ""),
             "steps": K, "ms": round(dt * 1e3, 4),
             "ms_hipgraph_replay": round(dt_graph * 1e3, 4) if dt_graph is not None else None,
             "Mpixel_per_s": round(len(zs) * N * N / dt / 1e6, 1),
             "step_bytes": step_bytes, "step_frac": round(step_bytes / dt / 1e9 / HBM_PEAK_GBS, 4),
             "step_frac_per_propagation": round(step_bytes_pp / dt / 1e9 / HBM_PEAK_GBS, 4),
             "fresnel_engine": {1: "rocfft", 2: "lds"}[plan.engine], "refraction_halo": halo,
             "refraction_halo_tuning_ms": {str(h): round(v, 4) for h, v in halo_ms.items()},
             "kernel_ms_per_step": dict(sorted(kern.items(), key=lambda kv: -kv[1]))}
        if kshort:
            e["kernel_ms_short_launches"] = dict(sorted(kshort.items(), key=lambda kv: -kv[1]),
                                                 note="launches under 30 us: event-pair overhead of the same order, a ranking only")
        # ---- parity against the float64 restatement (the checker; after the timed region)
        nt = max(1, min(32, (os.cpu_count() or 1) // 2))
        par = {}
        if a.no_config_parity:
            e["parity"] = {"skipped": "--no-config-parity"}
            out[str(N)] = e
            plan.close()
            if det is not None:
                det.close()
            del T, geom, fres, refr, dets, wave_mats, rt_mats
            torch.cuda.empty_cache()
            continue
        if N <= 2048:
            g64 = T.cpu().numpy()
            ref_f = cb.fresnel_intensity(g64, delta, beta, amp, zs[0], E, M, pix, nt)
            ref_r = cb.refraction_intensity(g64, delta, beta, I0, zs[0], E, M, pix, nt)
            par["fresnel"] = float(np.max(np.abs(fres[0].cpu().numpy() - ref_f)) / np.max(np.abs(ref_f)))
            par["refraction"] = float(np.max(np.abs(refr[0].cpu().numpy() - ref_r)) / np.max(np.abs(ref_r)))
            if detect:
                ref_d = orc.detection(ref_f, sig_src * 2.355, ov, (n, n), sig_psf)
                par["detector"] = float(np.max(np.abs(dets[0].cpu().numpy() - ref_d)) / np.max(np.abs(ref_d)))
            par["what"] = "whole images, distance %.1f m" % zs[0]
        else:
            W = 64
            strip = np.repeat(T[:, :, :1].cpu().numpy(), W, axis=2).copy()
            Ts = torch.from_numpy(strip).to(dev)
            ws = ops.MaterialStack(Ts, cphase=[-k * d for d in delta], catt=[-k * b for b in beta])
            rs = ops.MaterialStack(Ts, cphase=[-k * d for d in delta], catt=[-2 * k * b for b in beta])
            sp = ops.FresnelPlan(N, W, max_dist=1)
            so = [torch.empty((N, W), dtype=torch.float32, device=dev)]
            sp.propagate(aa[-1:], gp[-1:], (du[0], 2 * np.pi / (W * h)), amp=amp, mats=ws, want_wave=[False], inten_out=so)
            ref_f = cb.fresnel_intensity(strip, delta, beta, amp, zs[-1], E, M, pix, nt)
            par["fresnel"] = float(np.max(np.abs(so[0].cpu().numpy() - ref_f)) / np.max(np.abs(ref_f)))
            sr, _, _ = ops.refract((N, W), rs, dsc[-1], (N, W), I0=I0)
            ref_r = cb.refraction_intensity(strip, delta, beta, I0, zs[-1], E, M, pix, nt)
            par["refraction"] = float(np.max(np.abs(sr.cpu().numpy() - ref_r)) / np.max(np.abs(ref_r)))
            sd = ops.DetectorPlan(N, W, ov, n, W // ov, sig_src, sig_psf)
            dd = sd.detect(so[0])
            ref_d = orc.detection(ref_f, sig_src * 2.355, ov, (n, W // ov), sig_psf)
            par["detector"] = float(np.max(np.abs(dd.cpu().numpy() - ref_d)) / np.max(np.abs(ref_d)))
            par["what"] = ("%d-column strip of the same membrane (constant along axis 1), distance %.1f m; fresnel_axis1 / "
                           "refraction_axis1: the transposed strip (%d rows, constant along axis 0) -- its long lines run along "
                           "axis 1, i.e. through pass 2 of the engine, the step's dominant kernel" % (W, zs[-1], W))
            sp.close()
            sd.close()
            del Ts, ws, rs, so, sr, dd
            # the transposed strip: long lines along axis 1 (pass 2: strided reads of the blocked intermediate, the DIF instance
            # <16, false, PART, PAIR, ., ., DIF> at 16384) against the same float64 restatement
            strip2 = np.repeat(T[:, :1, :].cpu().numpy(), W, axis=1).copy()
            Ts = torch.from_numpy(strip2).to(dev)
            ws = ops.MaterialStack(Ts, cphase=[-k * d for d in delta], catt=[-k * b for b in beta])
            rs = ops.MaterialStack(Ts, cphase=[-k * d for d in delta], catt=[-2 * k * b for b in beta])
            sp = ops.FresnelPlan(W, N, max_dist=1)
            so = [torch.empty((W, N), dtype=torch.float32, device=dev)]
            sp.propagate(aa[-1:], gp[-1:], (2 * np.pi / (W * h), du[1]), amp=amp, mats=ws, want_wave=[False], inten_out=so)
            ref_f = cb.fresnel_intensity(strip2, delta, beta, amp, zs[-1], E, M, pix, nt)
            par["fresnel_axis1"] = float(np.max(np.abs(so[0].cpu().numpy() - ref_f)) / np.max(np.abs(ref_f)))
            sr, _, _ = ops.refract((W, N), rs, dsc[-1], (W, N), I0=I0)
            ref_r = cb.refraction_intensity(strip2, delta, beta, I0, zs[-1], E, M, pix, nt)
            par["refraction_axis1"] = float(np.max(np.abs(sr.cpu().numpy() - ref_r)) / np.max(np.abs(ref_r)))
            sp.close()
            del Ts, ws, rs, so, sr
            if not a.no_whole_image_parity:
                # ... and ONE WHOLE image of each kind at the longest distance against the float64 restatement (VERDICT r3
                # weak 1c: the strips pin the operator, this pins the image -- the membrane, the far rays, the partition's
                # block boundaries; ~15 GB of host memory and a minute of host time at 16384^2)
                tw0 = time.perf_counter()
                g64 = T.cpu().numpy()
                ref_f = cb.fresnel_intensity(g64, delta, beta, amp, zs[-1], E, M, pix, nt)
                par["fresnel_whole_image"] = float(np.max(np.abs(fres[-1].cpu().numpy() - ref_f)) / np.max(np.abs(ref_f)))
                del ref_f
                ref_r = cb.refraction_intensity(g64, delta, beta, I0, zs[-1], E, M, pix, nt)
                par["refraction_whole_image"] = float(np.max(np.abs(refr[-1].cpu().numpy() - ref_r)) / np.max(np.abs(ref_r)))
                del ref_r, g64
                par["whole_image_seconds"] = round(time.perf_counter() - tw0, 1)
        par["tolerance"] = PARITY_TOL
        par["ok"] = bool(max(v for kx, v in par.items() if kx in ("fresnel", "refraction", "detector", "fresnel_axis1",
                                                                  "refraction_axis1", "fresnel_whole_image",
                                                                  "refraction_whole_image")) <= PARITY_TOL)
        e["parity"] = par
        out[str(N)] = e
        plan.close()
        if det is not None:
            det.close()
        del T, geom, fres, refr, dets, wave_mats, rt_mats
        torch.cuda.empty_cache()
    _lib.check(_lib.lib().psx_refract_set_halo(a.halo), "psx_refract_set_halo")
    return out


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
        prof["_csrc_sha1"] = json.load(open(f)).get("csrc_sha1")      # the kernel sources the profile was collected on
        return prof
    return None


def csrc_sha1():
    """The hash tools/summarise_profiles.py stamps into a PMC summary: paresis_amd/csrc/*.hip, *.hpp and the Makefile."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "paresis_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")) or f == "Makefile":
            h.update(f.encode() + b"\0" + open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def pmc_value(prof, kernel, field):
    if not prof:
        return None
    for name, e in prof.items():
        if not isinstance(e, dict):
            continue
        if kernel in ("k_fresnel_rows", "k_fresnel_cols"):
            # pass 2 (k_fresnel_rows) is the CONTIG = false instance of a line kernel (strided reads), pass 1 CONTIG = true:
            # k_fresnel_p2<R1, CONTIG, DUAL, QUEUE> (power-of-two transforms), k_fresnel_lines<R3, CONTIG, ...> (576 R3 points),
            # k_fresnel_part<CONTIG, PAIR, DIF> (longer lines)
            if name.startswith("k_fresnel_lines<") or name.startswith("k_fresnel_p2<"):     # <R3 | R1, CONTIG, ...>
                contig = name.split(",")[1].strip()
            elif name.startswith("k_fresnel_part<"):
                contig = name.split("<")[1].split(",")[0].strip()
            else:
                continue
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
