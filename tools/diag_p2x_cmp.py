#!/usr/bin/env python3
"""Diagnostic for the PSX_X_CMP build of fresnel_p2x.hip: which legs of round O's parked input differ from what round O forms from
LDS?  Words 24 (bit masks, low half: even engine waves, high half: odd ones) and 25 (threads that saw a difference) of every
workgroup's stamps.   python tools/diag_p2x_cmp.py Nx Ny nd"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import paresis_oracle as orc
from paresis_amd import ops
from paresis_amd._lib import lib
Nx, Ny, nd = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(5)
E, pix, M = 52.0, 2.9, 1.03
zs = (2.3, 7.2, 0.4, 5.0)[:nd]
w_in = (rng.normal(size=(Nx, Ny)) + 1j * rng.normal(size=(Nx, Ny))).astype(np.complex64)
kk = orc.getk(E * 1000)
du = (2 * np.pi / (Nx * pix * 1e-6), 2 * np.pi / (Ny * pix * 1e-6))
plan = ops.FresnelPlan(Nx, Ny, max_dist=nd, engine=2)
buf = torch.zeros(512 * 32, dtype=torch.int64, device="cuda")
lib().psx_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
outs = plan.propagate([z / (2 * kk * M) for z in zs], [kk * z / M for z in zs], du, wave_in=torch.from_numpy(w_in).cuda())
torch.cuda.synchronize()
lib().psx_debug_stamps(None)
b = buf.cpu().numpy().reshape(512, 32)
for wg in range(512):
    m, c = int(b[wg, 24]), int(b[wg, 25])
    if m or c:
        print("wg %3d  even waves %08x  odd waves %08x  threads %d" % (wg, m & 0xffffffff, (m >> 32) & 0xffffffff, c & 0xffffffff))
flat = buf.cpu().numpy()
per = flat[64 * 32:64 * 32 + 512]
bad = np.nonzero(per)[0]
print("threads of workgroup 0 with differing legs:", len(bad))
for t in bad[:200]:
    nA = 32 * (t >> 6) + (t & 31)
    print("  t %3d wave %d lane %2d lineA %d nA %3d n0 %3d  legs %08x" % (t, t >> 6, t & 63, (t >> 5) & 1, nA, 2 * nA + ((t >> 5) & 1), int(per[t]) & 0xffffffff))
ref = orc.wave_propagation(w_in.astype(np.complex128), zs[0], E, M, (Nx, Ny), pix)
print("max err %.2e" % (np.abs(outs[0].cpu().numpy() - ref).max() / np.abs(ref).max()))
