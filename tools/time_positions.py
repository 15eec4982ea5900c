#!/usr/bin/env python3
"""Diagnostic: the membrane-position loop of main.py:63-110 on ONE GPU at the bench size -- per position: membrane
synthesis (seeded offsets, sphere splat on the GPU) + the image-formation chain + detection + shot noise; images stay in HBM.
BASELINE.json config 4 is 64 such positions over 8 GPUs (8 per GPU, no data-path collective).

    python tools/time_positions.py [N] [NPOS] [--poly 25]      # --poly E: a tube spectrum of E energies (polychromatic position)
                                   [--float-atomics] [--halo 4|6|8] [--sim RT|Fresnel] [--scatter [--thin F]] [--ov 2|4]
"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from paresis_amd import _lib, ops, synth
import _switches                      # PSX_SWITCHES="near_lds_pad=8" -> psx_debug_switch
_switches.apply()

def _opt(name, default=None):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


_skip = set()
for _o in ('--poly', '--halo', '--sim', '--thin', '--ov'):
    if _o in sys.argv:
        _skip.add(sys.argv.index(_o) + 1)
_args = [a for i, a in enumerate(sys.argv) if i > 0 and not a.startswith('--') and i not in _skip]
N = int(_args[0]) if len(_args) > 0 else 4096
NPOS = int(_args[1]) if len(_args) > 1 else 16
npoly = int(sys.argv[sys.argv.index('--poly') + 1]) if '--poly' in sys.argv else 0
spectrum = None
if npoly:
    e = np.linspace(20.0, 20.0 + 2.0 * (npoly - 1), npoly)
    w = np.exp(-0.5 * ((e - e.mean()) / (0.3 * (e[-1] - e[0] + 1))) ** 2)
    spectrum = [(float(a), float(b)) for a, b in zip(e, w / w.sum())]
lib = _lib.lib()
for sim in ([_opt('--sim')] if _opt('--sim') else ["Fresnel", "RT"]):
    exp, place = synth.bench_experiment(N, sim, noise=True, seed=3, spectrum=spectrum, ov=int(_opt('--ov', 2)))
    exp.exp_dict['reproducible'] = '--float-atomics' not in sys.argv
    if _opt('--halo'):
        exp.exp_dict['refractionHalo'] = int(_opt('--halo'))
    if '--scatter' in sys.argv:               # the sample as a scattering one (SAM:322-344): fastRefractionDF on the chain's sample hop
        exp.mySampleofInterest.myName = 'cylinder_beeds'
        if _opt('--thin'):                    # ... of 1/F the thickness: the dark-field width goes with its square root
            exp.mySampleofInterest.myGeometry = (np.asarray(exp.mySampleofInterest.myGeometry) / float(_opt('--thin'))).astype(np.float32)

    def position(p):
        place(p)
        return exp.computeSampleAndReferenceImages(p)

    position(0); position(1); torch.cuda.synchronize()
    t0 = time.perf_counter()
    keep, each = [], []
    for p in range(1, NPOS + 1):
        t1 = time.perf_counter()
        na = torch.cuda.memory_stats()["num_device_alloc"]
        keep.append(position(p)[:2])
        each.append((time.perf_counter() - t1) * 1e3)
        if torch.cuda.memory_stats()["num_device_alloc"] != na:
            each[-1] = -each[-1]              # printed negative: the caching allocator went to hipMalloc during this position
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    exp.resolve_mean_energy(); ops.check_status(keep[0][0].device, "positions")
    print("   per position (host, ms; negative: the caching allocator called hipMalloc):", " ".join("%.1f" % t for t in each))
    print("   median %.2f ms per position (host issue time)" % float(np.median(np.abs(each))))
    print("%s%s: %d positions of %dx%d (detector %dx%d) in %.1f ms = %.2f ms per position (%.0f Mpixel/s of study grid)"
          % (sim, " poly%d" % npoly if npoly else "", NPOS, N, N, N // 2, N // 2, dt * 1e3, dt / NPOS * 1e3, NPOS * N * N / dt / 1e6))
    lib.psx_profile_enable(1)
    for p in range(1, 5): position(p)
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    lib.psx_profile_summary(buf, len(buf)); lib.psx_profile_enable(0)
    ks = {l.split()[0]: (int(l.split()[1]) / 4, float(l.split()[2]) / 4) for l in buf.value.decode().splitlines()}
    print("   library kernels per position %.2f ms: %s" % (sum(v[1] for v in ks.values()),
          ", ".join("%s x%.0f %.3f" % (k, c, t) for k, (c, t) in sorted(ks.items(), key=lambda kv: -kv[1][1]))))
