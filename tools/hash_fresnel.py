#!/usr/bin/env python3
"""Diagnostic: sha256 of the images a set of Fresnel calls produces -- every line-kernel variant (whole lines at R3 = 4 / 8 / 16,
shared-forward rounds, work queue, partitioned, coupled, DIF on either axis).  Run it with two builds of the library in place and
compare the output: a refactoring of the kernels must leave every hash unchanged."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from paresis_amd import ops
import _switches                      # PSX_SWITCHES="no_park=1 ..." -> psx_debug_switch (the library reads no environment)
_switches.apply()
from paresis_amd.getk import getk
kk = getk(52000.0)
h = 2.9e-6
def run(Nx, Ny, nd, queue=False, inten=False):
    gen = torch.Generator(device="cuda").manual_seed(Nx * 7 + Ny)
    w = torch.complex(1.0 + 0.2 * torch.randn(Nx, Ny, device="cuda", generator=gen), 0.2 * torch.randn(Nx, Ny, device="cuda", generator=gen)).to(torch.complex64)
    plan = ops.FresnelPlan(Nx, Ny, max_dist=nd, engine=2)
    if queue:
        plan.work_queue(True)
    zs = [1.6, 3.6, 5.2, 7.2, 9.0][:nd]
    a = [z / (2 * kk * 1.01) for z in zs]
    gp = [kk * z / 1.01 for z in zs]
    du = (2 * np.pi / (Nx * h), 2 * np.pi / (Ny * h))
    if inten:
        outs = [torch.full((Nx, Ny), 0.25, dtype=torch.float32, device="cuda") for _ in zs]
        plan.propagate(a, gp, du, wave_in=w, want_wave=[False] * nd, inten_out=outs, inten_scale=[1.5] * nd, add=True)
    else:
        outs = plan.propagate(a, gp, du, wave_in=w)
    torch.cuda.synchronize()
    hsh = hashlib.sha256()
    for o in outs:
        hsh.update(o.cpu().numpy().tobytes())
    plan.close()
    return hsh.hexdigest()[:16]
cases = [(512, 512, 1), (300, 700, 3), (1100, 900, 2), (2048, 2048, 1), (2048, 2048, 4), (4096, 4096, 1), (4096, 4096, 4), (4096, 4096, 3),
         (5000, 5000, 2), (4704, 1500, 2), (9800, 320, 1), (320, 12000, 2), (16384, 128, 2), (128, 16384, 2), (13001, 200, 1)]
for c in cases:
    print(c, run(*c), run(*c, inten=True))
for c in [(512, 512, 2), (2048, 2048, 3), (4096, 4096, 4)]:
    print(c, "queue", run(*c, queue=True))
