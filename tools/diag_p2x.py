#!/usr/bin/env python3
"""Diagnostic: where does the two-round kernel (fresnel_p2x.hip) differ from the float64 oracle?  Rows of a (16384, Ny) grid."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import paresis_oracle as orc
from paresis_amd import ops
Nx, Ny, nd = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(5)
E, pix, M = 52.0, 2.9, 1.03
zs = (2.3, 7.2, 0.4, 5.0)[:nd]
w_in = (rng.normal(size=(Nx, Ny)) + 1j * rng.normal(size=(Nx, Ny))).astype(np.complex64)
kk = orc.getk(E * 1000)
du = (2 * np.pi / (Nx * pix * 1e-6), 2 * np.pi / (Ny * pix * 1e-6))
plan = ops.FresnelPlan(Nx, Ny, max_dist=nd, engine=2)
for rep in range(2):
    outs = plan.propagate([z / (2 * kk * M) for z in zs], [kk * z / M for z in zs], du, wave_in=torch.from_numpy(w_in).cuda())
    for z, o in zip(zs, outs):
        ref = orc.wave_propagation(w_in.astype(np.complex128), z, E, M, (Nx, Ny), pix)
        err = np.abs(o.cpu().numpy() - ref) / np.abs(ref).max()
        long_axis = 0 if Nx > Ny else 1
        per_line = err.max(axis=long_axis)          # per short-axis index (= per long line)
        per_pos = err.max(axis=1 - long_axis)       # per position along the long lines
        bad_lines = np.nonzero(per_line > 1e-5)[0]
        bad_pos = np.nonzero(per_pos > 1e-5)[0]
        if rep == 0 and len(bad_pos):
            col = err[:, 0] if long_axis == 0 else err[0, :]
            bp = np.nonzero(col > 1e-5)[0]
            n0 = bp - 8163
            oo = o.cpu().numpy()
            colo = oo[:, 0] if long_axis == 0 else oo[0, :]
            colr = ref[:, 0] if long_axis == 0 else ref[0, :]
            for i in list(bp[:10]) + [int(bp[0]) - 1, int(bp[7]) + 1]:
                print("   i", i, "out", colo[i], "ref", colr[i], "ratio", colo[i] / colr[i])
            print("line 0 bad positions", [int(v) for v in bp[:16]], "\n n0 = i - 8163:", list(n0), "\n nA", list(n0 // 2), "lineA", list(n0 % 2))
        if len(bad_pos):
            # leg q x engine wave w of the stage-A butterfly that produced output i: m' = i + (P - 1 - 16384), q = m' // 512, w = (m' % 512) // 64
            N = max(Nx, Ny)
            shx = N + 30 - 1 - 16384
            col = (err[:, :].max(axis=1) if long_axis == 0 else err[:, :].max(axis=0)) > 1e-5
            mp = np.arange(N) + shx
            tab = np.zeros((33, 8), int)
            for i in np.nonzero(col)[0]:
                tab[min(mp[i] // 512, 32), (mp[i] % 512) // 64] += 1
            print("bad outputs by leg (rows) x wave (columns):")
            for q in range(33):
                if tab[q].any():
                    print("  leg %2d " % q + " ".join("%3d" % v for v in tab[q]))
        print("rep", rep, "z", z, "max err %.2e" % err.max(), "bad lines", bad_lines[:12], len(bad_lines), "bad positions: n", len(bad_pos),
              "range", (bad_pos.min(), bad_pos.max()) if len(bad_pos) else None, "first", bad_pos[:8])
