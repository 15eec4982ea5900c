#!/usr/bin/env python3
"""Diagnostic (never a timed run): phase shares of one workgroup of the refraction tile kernel."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from paresis_amd import _lib, ops, synth
from paresis_amd.getk import k_refraction, k_sample

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
z = float(sys.argv[2]) if len(sys.argv) > 2 else 3.6
lib = _lib.lib()
geo = synth.bench_geometry(N)
k = k_sample(52.0)
db = [synth.DELTA_BETA_52KEV[m] for m in geo["membrane_materials"]]
m = ops.MaterialStack(torch.from_numpy(geo["membrane"]).cuda(), cphase=[-k * d for d, _ in db], catt=[-2 * k * b for _, b in db])
h = geo["pix_um"] * 1e-6
dsc = z / k_refraction(52.0) / (h * geo["M"]) / h
out = torch.empty((N, N), dtype=torch.float32, device="cuda")
for _ in range(3):
    ops.refract((N, N), m, dsc, (N, N), I0=7500.0, out=out)
buf = torch.zeros((1 << 16, 16), dtype=torch.int64, device="cuda")
lib.psx_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
ops.refract((N, N), m, dsc, (N, N), I0=7500.0, out=out)
torch.cuda.synchronize()
lib.psx_debug_stamps(None)
s = buf.cpu().numpy().astype(np.float64)
s = s[s[:, 5] > 0]
names = ["stage phi/I", "zero+max reduce", "deposit loop", "barrier", "store tile"]
d = np.diff(s[:, :6], axis=1) * 10.0
print("workgroups:", len(s), " mean total %.2f us" % (d.sum(1).mean() / 1e3))
for n, v in zip(names, d.mean(0)):
    print("  %-16s %7.2f us  %5.1f %%" % (n, v / 1e3, 100 * v / d.sum(1).mean()))
print("kernel span %.1f us" % ((s[:, 5].max() - s[:, 0].min()) * 10.0 / 1e3))
