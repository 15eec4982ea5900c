#!/bin/bash
# Round 5, GPU session 81: the detector tests with the added PSF widths (k_psf_tile's 13- and 17-tap instantiations).
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s81
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -p no:cacheprovider -k "detector" > $OUT/tests.out 2>&1; echo "tests rc $?"; tail -3 $OUT/tests.out
