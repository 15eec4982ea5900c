#!/usr/bin/env python3
"""Diagnostic: k_refract_near / k_refract_far time per distance of the bench workload for both tile geometries."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from paresis_amd import _lib, ops, synth
from paresis_amd.getk import k_refraction, k_sample
import bench
N = 4096; E = 52.0
lib = _lib.lib()
geo = synth.bench_geometry(N, pointNum=0)
M, pix = geo["M"], geo["pix_um"]; h = pix * 1e-6
db = [synth.DELTA_BETA_52KEV[m] for m in geo["membrane_materials"]]
k = k_sample(E)
T = torch.from_numpy(geo["membrane"]).cuda()
mats = ops.MaterialStack(T, cphase=[-k * d for d, _ in db], catt=[-2 * k * b for _, b in db])
out = torch.empty((N, N), dtype=torch.float32, device="cuda")
for halo in (4, 6, 8):
    _lib.check(lib.psx_refract_set_halo(halo), "halo")
    for z in bench.DISTANCES:
        dsc = z / k_refraction(E) / (h * M) / h
        for _ in range(3):
            ops.refract((N, N), mats, dsc, (N, N), I0=7500.0, out=out)
        torch.cuda.synchronize()
        lib.psx_profile_enable(1)
        for _ in range(10):
            ops.refract((N, N), mats, dsc, (N, N), I0=7500.0, out=out)
        torch.cuda.synchronize()
        buf = ctypes.create_string_buffer(1 << 16)
        lib.psx_profile_summary(buf, len(buf))
        lib.psx_profile_enable(0)
        t = {l.split()[0]: float(l.split()[2]) / int(l.split()[1]) for l in buf.value.decode().splitlines()}
        print("halo %d  z=%.1f m: near %.1f us  far %.1f us" % (halo, z, 1e3 * t.get("k_refract_near", 0), 1e3 * t.get("k_refract_far", 0)))

# the distance batch (one launch per kernel for all distances) and the far-ray population per distance
for halo in (4, 6, 8):
    _lib.check(lib.psx_refract_set_halo(halo), "halo")
    dscs = [z / k_refraction(E) / (h * M) / h for z in bench.DISTANCES]
    outs = [torch.empty((N, N), dtype=torch.float32, device="cuda") for _ in dscs]
    for _ in range(3):
        ops.refract_multi((N, N), mats, dscs, (N, N), I0=7500.0, outs=outs)
    torch.cuda.synchronize()
    lib.psx_profile_enable(1)
    for _ in range(10):
        ops.refract_multi((N, N), mats, dscs, (N, N), I0=7500.0, outs=outs)
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    lib.psx_profile_summary(buf, len(buf))
    lib.psx_profile_enable(0)
    t = {l.split()[0]: float(l.split()[2]) / int(l.split()[1]) for l in buf.value.decode().splitlines()}
    print("halo %d  batch of %d: near %.1f us  far %.1f us" % (halo, len(dscs), 1e3 * t.get("k_refract_near", 0), 1e3 * t.get("k_refract_far", 0)))
    tile = {4: 56, 6: 52, 8: 48}[halo]
    nt = ((N + tile - 1) // tile) ** 2
    ws = ops._workspaces[(0, "ws")]
    cnt = ws[:4 * nt * len(dscs)].view(torch.int32).cpu().numpy().reshape(len(dscs), nt)
    for z, c in zip(bench.DISTANCES, cnt):
        print("   z=%.1f m: %d far rays in %d of %d tiles, busiest tile %d" % (z, c.sum(), (c > 0).sum(), nt, c.max()))
