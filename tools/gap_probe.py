#!/usr/bin/env python3
"""Diagnostic: wall time per call of each half of the bench step against the sum of its kernels' event-pair times."""
import os, sys, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from paresis_amd import _lib, ops, synth
from paresis_amd.getk import getk, k_refraction, k_sample
import bench
N = 4096; E = 52.0
lib = _lib.lib()
geo = synth.bench_geometry(N, pointNum=0)
M, pix = geo["M"], geo["pix_um"]; h = pix * 1e-6
db = [synth.DELTA_BETA_52KEV[m] for m in geo["membrane_materials"]]
k = k_sample(E)
T = torch.from_numpy(geo["membrane"]).cuda()
wave_mats = ops.MaterialStack(T, cphase=[-k * d for d, _ in db], catt=[-k * b for _, b in db])
rt_mats = ops.MaterialStack(T, cphase=[-k * d for d, _ in db], catt=[-2 * k * b for _, b in db])
Z = bench.DISTANCES
plan = ops.FresnelPlan(N, N, max_dist=len(Z))
kk = getk(E * 1000)
aa = [z / (2 * kk * M) for z in Z]; gp = [kk * z / M for z in Z]
du = (2 * np.pi / (N * h),) * 2
dsc = [z / k_refraction(E) / (h * M) / h for z in Z]
fres = [torch.empty((N, N), dtype=torch.float32, device="cuda") for _ in Z]
refr = [torch.empty((N, N), dtype=torch.float32, device="cuda") for _ in Z]

def fr():
    plan.propagate(aa, gp, du, amp=86.6, mats=wave_mats, want_wave=[False] * len(Z), inten_out=fres)
def rf():
    ops.refract_multi((N, N), rt_mats, dsc, (N, N), I0=7500.0, outs=refr)
def both():
    fr(); rf()

def wall(f, n=40):
    for _ in range(5): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

def events(f, n=40):
    lib.psx_profile_enable(1)
    for _ in range(n): f()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    lib.psx_profile_summary(buf, len(buf))
    lib.psx_profile_enable(0)
    return {l.split()[0]: float(l.split()[2]) / n * 1e3 for l in buf.value.decode().splitlines()}

for name, f in (("fresnel call", fr), ("refraction call", rf), ("both", both)):
    w = wall(f); e = events(f); w2 = wall(f)
    print("%-16s wall %.1f / %.1f us per call, kernels by events %.1f us: %s" % (name, w, w2, sum(e.values()), {k: round(v, 1) for k, v in e.items()}))
