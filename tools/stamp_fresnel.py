#!/usr/bin/env python3
"""Diagnostic (never a timed run): where does a workgroup of the LDS Fresnel row pass spend its wall time?"""
import ctypes
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from paresis_amd import _lib, ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ND = int(sys.argv[2]) if len(sys.argv) > 2 else 1      # with PSX_STAMP_PASS1=1: distances sharing pass 1
lib = _lib.lib()
plan = ops.FresnelPlan(N, N, max_dist=ND)
w = (torch.randn(N, N, device="cuda") + 1j * torch.randn(N, N, device="cuda")).to(torch.complex64)
out = torch.empty((N, N), dtype=torch.float32, device="cuda")
du = 2 * np.pi / (N * 2.9e-6)
outs = [torch.empty((N, N), dtype=torch.float32, device="cuda") for _ in range(ND)]
aa = [3e-12 * (k + 1) for k in range(ND)]
def run():
    plan.propagate(aa, [0.0] * ND, (du, du), wave_in=w, want_wave=[False] * ND, inten_out=outs)
for _ in range(3):
    run()
nblk = N
buf = torch.zeros((nblk, 16), dtype=torch.int64, device="cuda")
lib.psx_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
run()
torch.cuda.synchronize()
lib.psx_debug_stamps(None)
s = buf.cpu().numpy().astype(np.float64)
s = s[s[:, 11] > 0]
names = ["load+spread", "barrier", "fwd A", "barrier", "fwd B", "barrier", "C + xH + invC", "barrier", "inv B", "barrier", "inv A + store (last dist; earlier ones fold in here)"]
d = np.diff(s[:, :12], axis=1) * 10.0   # 100 MHz ticks -> ns
print("workgroups:", len(s), " mean total %.2f us" % (d.sum(1).mean() / 1e3))
for n, v in zip(names, d.mean(0)):
    print("  %-16s %7.2f us  %5.1f %%" % (n, v / 1e3, 100 * v / d.sum(1).mean()))
span = (s[:, 11].max() - s[:, 0].min()) * 10.0 / 1e3
print("kernel span %.1f us; sum of workgroup times / 256 CUs = %.1f us" % (span, d.sum() / 1e3 / 256))
