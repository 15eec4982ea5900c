#!/usr/bin/env python3
"""Diagnostic: wall and kernel time of fastRefractionDF at 4096^2 (dark field of ~2.5 px inside a cylinder)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from paresis_amd import _lib, ops, synth
from paresis_amd import refractionFileNumba2 as RF2
lib = _lib.lib()
N = 4096
geo = synth.bench_geometry(N)
pix, M = geo["pix_um"], geo["M"]
k = 2 * np.pi * 52e3 * 1.6e-19 / (6.626e-34 * 2.998e8)
phi = torch.from_numpy(-k * 6.2e-7 * geo["membrane"][0].astype(np.float64)).cuda()
I = torch.full((N, N), 7500.0, dtype=torch.float32, device="cuda")
WHERE = sys.argv[3] if len(sys.argv) > 3 else "sample"         # sample | all | none | <float: a disc of that fraction of the width>
mask = geo["sample"][0] > 0 if WHERE == "sample" else (np.ones((N, N), bool) if WHERE == "all" else np.zeros((N, N), bool))
df = torch.from_numpy(np.where(mask, 2.0e-6, 0.0 if WHERE != "none" else 1e-12)).cuda()
dfmax = float(df.max().item())
Iw = I.clone()
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
MUTATE = not (len(sys.argv) > 2 and sys.argv[2] == "chain")      # "chain": as Experiment calls it (input is a temporary)
def f():
    # (the call zeroes CLAMPED rays in its input, |D| > N: there are none here, so the same array serves every repetition and
    # nothing but the call runs between the clocks -- under `rocprofv3 --kernel-trace --stats` every kernel that appears REPS
    # times or more belongs to the call: all of them are the library's)
    return RF2.fastRefractionDF(Iw, phi, 3.6, 52.0, M, pix, df, darkFieldMax=dfmax, check=False, mutate=MUTATE)
f(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(REPS): f()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / REPS * 1e3
lib.psx_profile_enable(1); f(); torch.cuda.synchronize()
buf = ctypes.create_string_buffer(1 << 16); lib.psx_profile_summary(buf, len(buf)); lib.psx_profile_enable(0)
print("fastRefractionDF %dx%d (mutate=%s, dark field: %s, %.0f %% of the pixels): %.2f ms wall; library kernels: %s" % (N, N, MUTATE, WHERE, 100 * mask.mean(), wall, buf.value.decode().replace("\n", "; ")))
