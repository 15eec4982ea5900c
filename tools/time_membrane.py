#!/usr/bin/env python3
"""Diagnostic: the membrane synthesis of a position (seeded offsets + k_membrane_layers) alone: the kernel by the library's own
event pairs (psx_profile_*), and the loop's cadence -- which is the HOST's (about 0.067 ms of Python per call, gpurun_out/r5s62: it
does not change when the kernel does nothing), not the kernel's.
    python tools/time_membrane.py [N] [NPOS]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paresis_amd import _lib, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
NPOS = int(sys.argv[2]) if len(sys.argv) > 2 else 200
lib = _lib.lib()
exp, place = synth.bench_experiment(N, "RT", noise=False, seed=3)
for p in range(8): place(p)
torch.cuda.synchronize()
t0 = time.perf_counter()
for p in range(NPOS): place(p)
torch.cuda.synchronize()
cadence = (time.perf_counter() - t0) / NPOS * 1e3
lib.psx_profile_enable(1)
for p in range(NPOS): place(p)
torch.cuda.synchronize()
buf = ctypes.create_string_buffer(1 << 14); lib.psx_profile_summary(buf, len(buf)); lib.psx_profile_enable(0)
ks = {l.split()[0]: float(l.split()[2]) / int(l.split()[1]) for l in buf.value.decode().splitlines()}
print("membrane synthesis %dx%d: kernel %.4f ms (event pairs around each launch, %d positions); loop cadence %.4f ms per position (host)"
      % (N, N, ks.get("k_membrane", float("nan")), NPOS, cadence))
