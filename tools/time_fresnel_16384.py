#!/usr/bin/env python3
"""Diagnostic: per-kernel times of ONE 4-distance Fresnel call at N^2 (default 16384), library event pairs."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paresis_amd import _lib, ops
import _switches                      # PSX_SWITCHES="no_dif=1 ..." -> psx_debug_switch (the library reads no environment)
_switches.apply()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ND = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lib = _lib.lib()
plan = ops.FresnelPlan(N, N, max_dist=ND)
T = torch.rand((1, N, N), device="cuda") * 1e-4
m = ops.MaterialStack(T, cphase=[-3e5], catt=[-3.0])
outs = [torch.empty((N, N), dtype=torch.float32, device="cuda") for _ in range(ND)]
aa = [2e-12 * (k + 1) for k in range(ND)]
f = lambda: plan.propagate(aa, [0.1] * ND, (3e5, 3e5), amp=10.0, mats=m, want_wave=[False] * ND, inten_out=outs)
f(); f(); torch.cuda.synchronize()
lib.psx_profile_enable(1)
for _ in range(3): f()
torch.cuda.synchronize()
buf = ctypes.create_string_buffer(1 << 16)
lib.psx_profile_summary(buf, len(buf))
for line in buf.value.decode().splitlines():
    nm, cnt, tot = line.split()
    print("%-24s %8.3f ms per launch" % (nm, float(tot) / int(cnt)))
