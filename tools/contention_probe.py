#!/usr/bin/env python3
"""Diagnostic: what a few foreign workgroups cost the position loop -- `c` spinning workgroups with a little LDS (what the copy
kernels of an RCCL transfer look like to the dispatcher) run on a side stream while 4096^2 positions are computed.  The
Fresnel line kernels are 256 persistent workgroups that need a whole CU each (all of its LDS).
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libspin_occupy.so tools/spin_occupy.hip
    python tools/contention_probe.py [N] [sim] [queue]      # queue: psx_fresnel_plan_work_queue on"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from paresis_amd import synth

here = os.path.dirname(os.path.abspath(__file__))
spin = ctypes.CDLL(os.path.join(here, "libspin_occupy.so"))
spin.spin_occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sim = sys.argv[2] if len(sys.argv) > 2 else "Fresnel"
exp, place = synth.bench_experiment(N, sim, noise=True, seed=3)


def position(p):
    place(p)
    return exp.computeSampleAndReferenceImages(p)


for p in range(3):
    position(p)
if len(sys.argv) > 3 and sim == "Fresnel":
    exp._plan().work_queue(True)
    print("work queue on")
torch.cuda.synchronize()
side = torch.cuda.Stream()
sink = torch.zeros(4, device="cuda")
NPOS = 16
for c in (0, 1, 4, 8, 16, 32, 64):
    torch.cuda.synchronize()
    if c:
        spin.spin_occupy(c, 256, 4096, 60000.0, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(side.cuda_stream))   # 60 ms
        time.sleep(0.002)
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for p in range(1, NPOS + 1):
        position(p)
    e1.record()
    e1.synchronize()
    print("%3d foreign workgroups: %.3f ms per position" % (c, e0.elapsed_time(e1) / NPOS))
    torch.cuda.synchronize()
