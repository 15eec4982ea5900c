#!/usr/bin/env python3
"""Diagnostic: does the time of the 4096^2 line kernels depend on WHERE the plan's buffers live?  Session r4s17 showed pass 1 at
0.388 or 0.405 ms depending on nothing but the size of an unrelated code object of the library.  Here: in ONE process (code
fixed), plans created after dummy allocations of different sizes (which shift every later hipMalloc), 20 calls each, times from
the library's own event pairs; the addresses of the input wave and of the images are printed beside them."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from paresis_amd import _lib, ops
from paresis_amd.getk import getk
lib = _lib.lib()
N = 4096
kk = getk(52000.0)
h = 2.9e-6
zs = [1.6, 3.6, 5.2, 7.2]
a = [z / (2 * kk * 1.01) for z in zs]
gp = [kk * z / 1.01 for z in zs]
du = (2 * np.pi / (N * h), 2 * np.pi / (N * h))
gen = torch.Generator(device="cuda").manual_seed(1)
w = torch.complex(1.0 + 0.2 * torch.randn(N, N, device="cuda", generator=gen), 0.2 * torch.randn(N, N, device="cuda", generator=gen)).to(torch.complex64)
outs = [torch.zeros((N, N), dtype=torch.float32, device="cuda") for _ in zs]
keep = []
def timed(plan):
    for _ in range(60):                                  # spin-up
        plan.propagate(a, gp, du, wave_in=w, want_wave=[False] * 4, inten_out=outs)
    torch.cuda.synchronize()
    lib.psx_profile_enable(1)
    for _ in range(20):
        plan.propagate(a, gp, du, wave_in=w, want_wave=[False] * 4, inten_out=outs)
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16); lib.psx_profile_summary(buf, len(buf)); lib.psx_profile_enable(0)
    ks = {l.split()[0]: float(l.split()[2]) / int(l.split()[1]) for l in buf.value.decode().splitlines()}
    return ks
pads = [0, 1 << 20, 3 << 20, 37 << 20, 64 << 20, 100 << 20, (1 << 30) + (5 << 20), 2 << 20, 17 << 20, 513 << 20]
for trial, pad in enumerate(pads):
    if pad:
        keep.append(torch.empty(pad, dtype=torch.uint8, device="cuda"))
    plan = ops.FresnelPlan(N, N, max_dist=4, engine=2)
    ks = timed(plan)
    print("trial %d pad %5d MiB: rows %.4f cols %.4f pre %.4f ms   wave@%x out0@%x pad@%x" % (
        trial, pad >> 20, ks.get("k_fresnel_rows", 0), ks.get("k_fresnel_cols", 0), ks.get("k_source_transposed", 0),
        w.data_ptr(), outs[0].data_ptr(), keep[-1].data_ptr() if keep else 0), flush=True)
    plan.close()
