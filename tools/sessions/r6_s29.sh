#!/bin/bash
# Round 6, GPU session 29: config-4 test with the wider escape table + the pack kernel's dense case; then where k_refract_near's
# time goes (build-time experiments with wrong images: conflict-free deposits / no stencil reads / both).
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s29
mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_gpu_main.py tests/test_gpu_kernels.py -x -q -m gpu -k "config4 or pack_counts or gather" > $OUT/tests.out 2>&1; rc=$?; echo "tests rc $rc"; tail -4 $OUT/tests.out
[ $rc -ne 0 ] && exit $rc
bash tools/ab_run.sh $OUT base own nosten both
