#!/usr/bin/env python3
"""Diagnostic: the shot noise of a position's two detector images (2048^2, mean 7500 counts) by the library's event pairs.
    python tools/time_poisson.py [n] [mean]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paresis_amd import _lib, ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
mean = float(sys.argv[2]) if len(sys.argv) > 2 else 7500.0
lib = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(1)
lam = [(torch.rand((n, n), generator=g, device="cuda") * 0.5 + 0.75) * mean for _ in range(2)]
bufs = [t.clone() for t in lam]
for _ in range(5): ops.poisson_multi([b.copy_(l) for b, l in zip(bufs, lam)], [11, 12])
torch.cuda.synchronize()
lib.psx_profile_enable(1)
for k in range(100): ops.poisson_multi([b.copy_(l) for b, l in zip(bufs, lam)], [2 * k, 2 * k + 1])
torch.cuda.synchronize()
buf = ctypes.create_string_buffer(1 << 14); lib.psx_profile_summary(buf, len(buf)); lib.psx_profile_enable(0)
ks = {l.split()[0]: float(l.split()[2]) / int(l.split()[1]) for l in buf.value.decode().splitlines()}
x = bufs[0].double()
print("k_poisson, two %dx%d images of mean %.0f: %.4f ms per launch (event pairs); mean %.2f, variance / mean %.4f"
      % (n, n, mean, ks.get("k_poisson", float("nan")), float(x.mean()), float(x.var() / x.mean())))
