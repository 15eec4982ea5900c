#!/usr/bin/env python3
"""Diagnostic: wall time per membrane position of the two image-formation chains at the bench size (4096^2 study grid,
oversampling 2, mono-energetic, in vacuum), with per-kernel event times -- where a position's time goes beyond the
bench's hot step (detections, accumulations, host code)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import types
from tests._build import build_experiment
from paresis_amd import _lib, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = _lib.lib()
geo = synth.bench_geometry(N, pointNum=0)
d = synth.DELTA_BETA_52KEV
obj = lambda g, mats: types.SimpleNamespace(geometry=g, delta=[[d[m][0]] for m in mats], beta=[[d[m][1]] for m in mats])
cfg = dict(dSM=140.0, dMO=1.6, dOD=3.6, meanShotCount=30000.0, ov=2, pix_um=geo["pix_um"], M=geo["M"], inVacuum=True,
           N=(N, N), spectrum=[(52.0, 1.0)], source_size_um=10.0, energy_sampling=1.0, det_dims=(N // 2, N // 2),
           det_pix_um=6.0, psf=1.2, bins=[], membrane=obj(geo["membrane"], geo["membrane_materials"]),
           sample=obj(geo["sample"], ["Nylon"]), air=None, plate=None, scintillator=None)
for sim in ("Fresnel", "RT"):
    exp = build_experiment(cfg, sim)
    f = (lambda p: exp.computeSampleAndReferenceImages_Fresnel(p)) if sim == "Fresnel" else (lambda p: exp.computeSampleAndReferenceImages_RT(p))
    f(0); f(1); torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n): f(1)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    lib.psx_profile_enable(1)
    for _ in range(n): f(1)
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    lib.psx_profile_summary(buf, len(buf)); lib.psx_profile_enable(0)
    ks = {l.split()[0]: (int(l.split()[1]) / n, float(l.split()[2]) / n) for l in buf.value.decode().splitlines()}
    print("%s chain, %dx%d, position 1: %.2f ms wall per position; kernels %.2f ms:" % (sim, N, N, wall, sum(v[1] for v in ks.values())))
    for k, (c, t) in sorted(ks.items(), key=lambda kv: -kv[1][1]):
        print("    %-22s %5.1f launches  %7.3f ms" % (k, c, t))
