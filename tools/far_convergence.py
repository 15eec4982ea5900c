#!/usr/bin/env python3
"""How much the far-ray shares of a refraction converge (VERDICT r4 item 3 ii): for the GPU-synthesised membrane of bench.py's
`configs` entries, counts per distance the far rays, their shares, the DISTINCT target pixels (what the fold pass walks) and the
distinct (source tile, target pixel) pairs (what would be left of the atomics if a source tile combined its shares in LDS first).

    python tools/far_convergence.py N ov [halo]          # e.g. 16384 4 8   |   4096 2 4
"""
import json
import sys
import types

import torch

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from paresis_amd import ops, synth                                        # noqa: E402
from paresis_amd.Samples.getMembraneFromFile import getMembraneSegmentedFromFile   # noqa: E402
from paresis_amd.getk import k_refraction, k_sample                       # noqa: E402

N, ov = int(sys.argv[1]), int(sys.argv[2])
H = int(sys.argv[3]) if len(sys.argv) > 3 else (8 if ov >= 4 else 4)
TH = {4: 56, 6: 52, 8: 48}[H]
E, I0, M = 52.0, 7500.0, 145.2 / 141.6
pix = 6.0 / ov / M
h = pix * 1e-6
db = [synth.DELTA_BETA_52KEV[m] for m in ("CuSn", "PMMA")]
k = k_sample(E)
smp = types.SimpleNamespace(myMeanSphereRadius=15.0, myNbOfLayers=2)
geom, _ = getMembraneSegmentedFromFile(smp, N, N, pix * 140.0 / 141.6, 0, 6000.0, stacked=True)
T = geom[2]
rt = ops.MaterialStack(T, cphase=[-k * d for d, _ in db], catt=[-2 * k * b for _, b in db])
ops.set_refract_halo(H)
out = {"N": N, "ov": ov, "halo": H, "tile": TH}
for z in (1.6, 3.6, 5.2, 7.2):
    dsc = z / k_refraction(E) / (h * M) / h
    _, Dx, Dy = ops.refract((N, N), rt, dsc, (N, N), I0=I0, want_D=True)
    Dx, Dy = Dx[15:-15, 15:-15], Dy[15:-15, 15:-15]
    fx, fy = torch.floor(Dx).to(torch.int32), torch.floor(Dy).to(torch.int32)
    near = (fx >= -H) & (fx < H) & (fy >= -H) & (fy < H)
    idx = torch.nonzero(~near)
    del near
    i, j = idx[:, 0].to(torch.int32), idx[:, 1].to(torch.int32)
    bi, bj = i + fx[idx[:, 0], idx[:, 1]], j + fy[idx[:, 0], idx[:, 1]]
    del fx, fy, Dx, Dy, idx
    src_tile = (i // TH).to(torch.int64) * ((N + TH - 1) // TH) + (j // TH)
    keys_px, keys_tile = [], []
    nshares = 0
    for di in (0, 1):
        for dj in (0, 1):
            ti, tj = bi + di, bj + dj
            ok = (ti >= 0) & (ti < N) & (tj >= 0) & (tj < N)
            r0, c0 = (ti // TH) * TH, (tj // TH) * TH
            gathered = (i >= r0 - H) & (i < r0 + TH + H) & (j >= c0 - H) & (j < c0 + TH + H)   # the target tile's own window
            ok &= ~gathered
            p = ti.to(torch.int64)[ok] * N + tj.to(torch.int64)[ok]
            nshares += int(p.numel())
            keys_px.append(p)
            keys_tile.append(src_tile[ok] * (N * N) + p)
    px = torch.cat(keys_px)
    n_px = int(torch.unique(px).numel())
    del px, keys_px
    kt = torch.cat(keys_tile)
    n_tile = int(torch.unique(kt).numel())
    del kt, keys_tile
    per_list = torch.bincount(src_tile)
    out["z=%.1f" % z] = {"far_rays": int(i.numel()), "far_frac_of_pixels": round(i.numel() / N / N, 4), "shares": nshares,
                         "distinct_target_pixels": n_px, "shares_per_target": round(nshares / max(1, n_px), 2),
                         "distinct_after_source_tile_combine": n_tile,
                         "atomics_saved_by_tile_combine": round(1 - n_tile / max(1, nshares), 3),
                         "records_per_nonempty_list_mean": round(float(per_list[per_list > 0].float().mean()), 1),
                         "records_per_list_max": int(per_list.max())}
    del i, j, bi, bj, src_tile
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
