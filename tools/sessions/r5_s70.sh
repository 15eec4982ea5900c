#!/bin/bash
# Round 5, GPU session 70: the PSF stage as a register-tiled stencil (k_psf_tile): detector tests (goldens,
# multi-block grids, images of a bin, fuzz), then the two stages by event pairs, 4096^2 -> 2048^2 and 16384^2 -> 4096^2.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s70
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_experiment.py tests/test_gpu_main.py -m gpu -x -q -p no:cacheprovider -k "detector or chain or xml or main" > $OUT/tests.out 2>&1; rc=$?; echo "tests rc $rc"; tail -2 $OUT/tests.out
[ $rc -eq 0 ] || exit $rc
for r in 1 2; do timeout -k 10 200 python tools/time_detector.py 2>&1 | tail -1; done | tee $OUT/det4096.out
timeout -k 10 300 python tools/time_detector.py 16384 4 4 2>&1 | tail -1 | tee $OUT/det16384.out
