#!/usr/bin/env python3
"""Diagnostic: the membrane-position loop of main.py:63-110 on ONE GPU at the bench size -- per position: membrane
synthesis (seeded offsets, sphere splat on the GPU) + the image-formation chain + detection; images stay in HBM.
BASELINE.json config 4 is 64 such positions over 8 GPUs (8 per GPU, no data-path collective)."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import types
from tests._build import build_experiment
from paresis_amd import synth
from paresis_amd.Samples.getMembraneFromFile import getMembraneSegmentedFromFile
_args = [a for a in sys.argv[1:] if not a.startswith('--')]
N = int(_args[0]) if len(_args) > 0 else 4096
NPOS = int(_args[1]) if len(_args) > 1 else 16
geo = synth.bench_geometry(N, pointNum=0)
d = synth.DELTA_BETA_52KEV
obj = lambda g, mats: types.SimpleNamespace(geometry=g, delta=[[d[m][0]] for m in mats], beta=[[d[m][1]] for m in mats])
cfg = dict(dSM=140.0, dMO=1.6, dOD=3.6, meanShotCount=30000.0, ov=2, pix_um=geo["pix_um"], M=geo["M"], inVacuum=True,
           N=(N, N), spectrum=[(52.0, 1.0)], source_size_um=10.0, energy_sampling=1.0, det_dims=(N // 2, N // 2),
           det_pix_um=6.0, psf=1.2, bins=[], membrane=obj(geo["membrane"], geo["membrane_materials"]),
           sample=obj(geo["sample"], ["Nylon"]), air=None, plate=None, scintillator=None)
smp = types.SimpleNamespace(myMeanSphereRadius=15.0, myNbOfLayers=2)
mpix = geo["pix_um"] * 140.0 / 141.6
for sim in ("Fresnel", "RT"):
    exp = build_experiment(cfg, sim, noise=True)
    def position(p):
        geom, _ = getMembraneSegmentedFromFile(smp, N, N, mpix, p, 6000.0)
        exp.myMembrane.myGeometry = torch.stack(geom)
        return exp.computeSampleAndReferenceImages(p)
    position(0); position(1); torch.cuda.synchronize()
    t0 = time.perf_counter()
    keep, each = [], []
    for p in range(1, NPOS + 1):
        t1 = time.perf_counter()
        na = torch.cuda.memory_stats()["num_device_alloc"]
        keep.append(position(p)[:2])
        each.append((time.perf_counter() - t1) * 1e3)
        if torch.cuda.memory_stats()["num_device_alloc"] != na:
            each[-1] = -each[-1]              # printed negative: the caching allocator went to hipMalloc during this position
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("   per position (host, ms; negative: the caching allocator called hipMalloc):", " ".join("%.1f" % t for t in each))
    print("   median %.2f ms per position" % float(np.median(np.abs(each))))
    print("%s: %d positions of %dx%d (detector %dx%d) in %.1f ms = %.2f ms per position (%.0f Mpixel/s of study grid)"
          % (sim, NPOS, N, N, N // 2, N // 2, dt * 1e3, dt / NPOS * 1e3, NPOS * N * N / dt / 1e6))
    import ctypes
    from paresis_amd import _lib
    lib = _lib.lib()
    lib.psx_profile_enable(1)
    for p in range(1, 5): position(p)
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    lib.psx_profile_summary(buf, len(buf)); lib.psx_profile_enable(0)
    ks = {l.split()[0]: (int(l.split()[1]) / 4, float(l.split()[2]) / 4) for l in buf.value.decode().splitlines()}
    print("   library kernels per position %.2f ms: %s" % (sum(v[1] for v in ks.values()),
          ", ".join("%s x%.0f %.3f" % (k, c, t) for k, (c, t) in sorted(ks.items(), key=lambda kv: -kv[1][1]))))
