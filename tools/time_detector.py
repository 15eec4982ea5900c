#!/usr/bin/env python3
"""Diagnostic: the detector operator alone (Detector.detection without noise) at the bench size; kernels by name under
rocprofv3 --kernel-trace --stats.    python tools/time_detector.py [N] [ov] [sigma_src, study pixels; bench: 0.036]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ctypes
from paresis_amd import _lib, ops
import _switches                      # PSX_SWITCHES="no_dif=1 ..." -> psx_debug_switch (the library reads no environment)
_switches.apply()

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ov = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sig = float(sys.argv[3]) if len(sys.argv) > 3 else 1.3
img = torch.rand((N, N), device="cuda") + 0.5
plan = ops.DetectorPlan(N, N, ov, N // ov, N // ov, sig, 1.2, device=img.device)
out = torch.empty((N // ov, N // ov), device="cuda")
for _ in range(3):
    plan.detect(img, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    plan.detect(img, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 50
print("detector %dx%d -> %dx%d: %.1f us per image (%.2f TB/s of the %d MB that must move)"
      % (N, N, N // ov, N // ov, dt * 1e6, (img.numel() + out.numel()) * 4 / dt / 1e12, (img.numel() + out.numel()) * 4 >> 20))
lib = _lib.lib()
lib.psx_profile_enable(1)
for _ in range(10):
    plan.detect(img, out=out)
torch.cuda.synchronize()
buf = ctypes.create_string_buffer(1 << 16); lib.psx_profile_summary(buf, len(buf)); lib.psx_profile_enable(0)
print("   library kernels per image (us):", ", ".join("%s x%d %.1f" % (l.split()[0], int(l.split()[1]) // 10, float(l.split()[2]) * 100 / 1)
                                                      for l in buf.value.decode().splitlines()))
