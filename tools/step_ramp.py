#!/usr/bin/env python3
"""Diagnostic: duration of each of the first steps of the bench workload in a fresh process (HIP events around every step):
how long the device takes to reach its steady step time after start-up."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from paresis_amd import _lib, ops, synth
from paresis_amd.getk import getk, k_refraction, k_sample
N, E = 4096, 52.0
DIST = (1.6, 3.6, 5.2, 7.2)
geo = synth.bench_geometry(N)
M, pix = geo["M"], geo["pix_um"]; h = pix * 1e-6
db = [synth.DELTA_BETA_52KEV[m] for m in geo["membrane_materials"]]
k = k_sample(E)
T = torch.from_numpy(geo["membrane"]).cuda()
wm = ops.MaterialStack(T, cphase=[-k * d for d, _ in db], catt=[-k * b for _, b in db])
rm = ops.MaterialStack(T, cphase=[-k * d for d, _ in db], catt=[-2 * k * b for _, b in db])
plan = ops.FresnelPlan(N, N, max_dist=4)
kk = getk(E * 1000)
aa = [z / (2 * kk * M) for z in DIST]; gp = [kk * z / M for z in DIST]; du = (2 * np.pi / (N * h),) * 2
dsc = [z / k_refraction(E) / (h * M) / h for z in DIST]
fres = [torch.empty((N, N), dtype=torch.float32, device="cuda") for _ in DIST]
refr = [torch.empty((N, N), dtype=torch.float32, device="cuda") for _ in DIST]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
torch.cuda.synchronize()
ev[0].record()
for i in range(n):
    plan.propagate(aa, gp, du, amp=86.6, mats=wm, want_wave=[False] * 4, inten_out=fres)
    ops.refract_multi((N, N), rm, dsc, (N, N), I0=7500.0, outs=refr)
    ev[i + 1].record()
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print("first 12 steps (ms):", " ".join("%.3f" % v for v in t[:12]))
for a in range(0, n, 10):
    print("steps %3d-%3d: mean %.4f ms" % (a, a + 9, float(np.mean(t[a:a + 10]))))
