#!/usr/bin/env python3
"""Diagnostic: far-ray fraction and kernel times of the refraction distance batch for each gather halo, on the GPU-synthesised
membrane of bench.py's `configs` entries (15 um spheres, two layers).   python tools/halo_sweep.py N ov [reproducible]"""
import ctypes, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from paresis_amd import _lib, ops, synth
from paresis_amd.Samples.getMembraneFromFile import getMembraneSegmentedFromFile
from paresis_amd.getk import k_refraction, k_sample
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ov = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rep = len(sys.argv) > 3 and sys.argv[3] == "reproducible"
lib = _lib.lib()
import _switches                      # PSX_SWITCHES="far_stride=7919" -> psx_debug_switch
_switches.apply()
E, I0 = 52.0, 7500.0
db = [synth.DELTA_BETA_52KEV[m] for m in ("CuSn", "PMMA")]
k = k_sample(E)
M = 145.2 / 141.6
pix = 6.0 / ov / M
h = pix * 1e-6
smp = types.SimpleNamespace(myMeanSphereRadius=15.0, myNbOfLayers=2)
geom, _ = getMembraneSegmentedFromFile(smp, N, N, pix * 140.0 / 141.6, 0, 6000.0, stacked=True)
rt = ops.MaterialStack(geom[2], cphase=[-k * d for d, _ in db], catt=[-2 * k * b for _, b in db])
zs = (1.6, 3.6, 5.2, 7.2)
dsc = [z / k_refraction(E) / (h * M) / h for z in zs]
outs = [torch.empty((N, N), dtype=torch.float32, device="cuda") for _ in zs]
ops.set_deterministic(rep)
tiles = {4: 56, 6: 52, 8: 48, 12: 40, 16: 32}
for halo in ((4, 6, 8, 12, 16) if ov >= 4 or '--all' in sys.argv else (4, 6, 8)):
    ops.set_refract_halo(halo)
    f = lambda: ops.refract_multi((N, N), rt, dsc, (N, N), I0=I0, outs=outs)
    f(); f(); torch.cuda.synchronize()
    lib.psx_profile_enable(1)
    for _ in range(5): f()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 14); lib.psx_profile_summary(buf, len(buf)); lib.psx_profile_enable(0)
    kern = {l.split()[0]: float(l.split()[2]) / int(l.split()[1]) for l in buf.value.decode().splitlines()}
    nt = ((N + tiles[halo] - 1) // tiles[halo]) ** 2
    ws = next(iter(ops._workspaces.values()))
    counts = ws[:4 * len(zs) * nt].view(torch.int32).to(torch.int64).view(len(zs), nt).sum(dim=1).tolist()
    print("N %d ov %d halo %d%s: far rays per distance %s = %s %% of the pixels; kernels (ms per launch) %s; total %.3f ms" %
          (N, ov, halo, " (order-independent replay)" if rep else "", counts, ["%.2f" % (100.0 * c / N / N) for c in counts],
           {a: round(b, 4) for a, b in kern.items()}, sum(kern.values())))
ops.set_deterministic(False)
