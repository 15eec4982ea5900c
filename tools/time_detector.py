#!/usr/bin/env python3
"""Diagnostic: the detector operator on the two images of a position (4096^2 -> 2048^2, the bench's geometry), by the library's
event pairs: front stage alone (a plan without PSF) and both stages.     python tools/time_detector.py [N] [ov] [nimg]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paresis_amd import _lib, ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ov = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nimg = int(sys.argv[3]) if len(sys.argv) > 3 else 2
n = N // ov
lib = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(1)
imgs = [torch.rand((N, N), generator=g, device="cuda") * 7500.0 for _ in range(nimg)]
outs = [torch.empty((n, n), device="cuda") for _ in range(nimg)]
res = {}
for name, psf in (("front", 0.0), ("both", 1.2)):
    plan = ops.DetectorPlan(N, N, ov, n, n, 0.036, psf)
    for _ in range(5): plan.detect_many(imgs, outs)
    torch.cuda.synchronize()
    lib.psx_profile_enable(1)
    for _ in range(200): plan.detect_many(imgs, outs)
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 14); lib.psx_profile_summary(buf, len(buf)); lib.psx_profile_enable(0)
    ks = {l.split()[0]: float(l.split()[2]) / 200 for l in buf.value.decode().splitlines()}
    res[name] = ks.get("k_band_pair", 0.0) + ks.get("k_psf_tile", 0.0)
    plan.close()
print("detector %dx%d -> %dx%d, %d images per call: front stage %.4f ms, PSF stage %.4f ms (event pairs, 200 calls)"
      % (N, N, n, n, nimg, res["front"], res["both"] - res["front"]))
