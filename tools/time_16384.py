#!/usr/bin/env python3
"""Diagnostic: wall time of the config-5 pieces (16384^2 study grid) on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from paresis_amd import ops
import _switches                      # PSX_SWITCHES="no_dif=1 ..." -> psx_debug_switch (the library reads no environment)
_switches.apply()
N, ov, n = 16384, 4, 4096
def wall(f, n=3):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
plan = ops.FresnelPlan(N, N, max_dist=1)
u = (torch.randn(N, N, device="cuda") * 0.1 + 1.0).to(torch.complex64)
inten = torch.zeros((N, N), dtype=torch.float32, device="cuda")
t = wall(lambda: plan.propagate([2e-12], [0.1], (3e5, 3e5), wave_in=u, want_wave=[False], inten_out=[inten]))
P = N + 30
print("fresnel 16384^2 engine %d: %.1f ms  (%.2f Gpixel/s; 64 B/px algorithmic -> %.2f TB/s)" % (plan.engine, t, N * N / t / 1e6, 64 * P * P / t / 1e9))
del u; plan.close()
T = torch.rand((1, N, N), device="cuda") * 1e-4
m = ops.MaterialStack(T, cphase=[-3e5], catt=[-3.0])
out = torch.empty((N, N), dtype=torch.float32, device="cuda")
t = wall(lambda: ops.refract((N, N), m, 2e-3, (N, N), I0=100.0, out=out))
print("refraction 16384^2: %.1f ms (%.2f Gpixel/s)" % (t, N * N / t / 1e6))
det = ops.DetectorPlan(N, N, ov, n, n, 10 * 3.6 / 141.6 / 6 * ov / 2.355, 1.2)
t = wall(lambda: det.detect(inten))
print("detector 16384^2 -> 4096^2: %.2f ms (%.2f TB/s of input)" % (t, 4 * N * N / t / 1e9))
import ctypes
from paresis_amd import _lib
lib = _lib.lib()
def kernels(f, n=5):
    lib.psx_profile_enable(1)
    for _ in range(n): f()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    lib.psx_profile_summary(buf, len(buf))
    lib.psx_profile_enable(0)
    return buf.value.decode()
print(kernels(lambda: det.detect(inten)))
del det, inten, out, T
torch.cuda.empty_cache()
# the bench-sized detector: 4096^2 study grid, oversampling 2, source blur 1.5 study px, PSF 1.2 px
N2, ov2, n2 = 4096, 2, 2048
img = torch.rand((N2, N2), device="cuda")
for ss, sp in ((0.0, 0.0), (1.5, 0.0), (0.0, 1.2), (1.5, 1.2)):
    d2 = ops.DetectorPlan(N2, N2, ov2, n2, n2, ss, sp)
    t = wall(lambda: d2.detect(img), n=20)
    print("detector 4096^2 -> 2048^2, sigma_src %.1f sigma_psf %.1f: %.3f ms (%.2f TB/s of input)" % (ss, sp, t, 4 * N2 * N2 / t / 1e9))
print(kernels(lambda: d2.detect(img), n=20))
