#!/usr/bin/env python3
"""Diagnostic: wall time of one membrane synthesis (getMembraneSegmentedFromFile: host binning + k_membrane) at 4096^2."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from paresis_amd import _lib, synth
from paresis_amd.Samples.getMembraneFromFile import getMembraneSegmentedFromFile
lib = _lib.lib()
class S: pass
for N, pix, rad, layers in ((4096, 2.9252, 15.0, 2), (4096, 2.9252, 50.0, 3), (16384, 1.46, 15.0, 2)):
    s = S(); s.myMeanSphereRadius = rad; s.myNbOfLayers = layers
    getMembraneSegmentedFromFile(s, N, N, pix, 0, 6000.0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for p in range(1, 4): getMembraneSegmentedFromFile(s, N, N, pix, p, 6000.0)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 3 * 1e3
    lib.psx_profile_enable(1)
    getMembraneSegmentedFromFile(s, N, N, pix, 5, 6000.0); torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16); lib.psx_profile_summary(buf, len(buf)); lib.psx_profile_enable(0)
    print("N=%d pix=%.2f um radius %.0f um x %d layers: %.1f ms wall per position; kernels: %s" % (N, pix, rad, layers, wall, buf.value.decode().replace("\n", "; ")))
