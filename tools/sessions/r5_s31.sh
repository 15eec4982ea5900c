#!/bin/bash
# Round 5, GPU session 31: the direct test of psx_refract_split_f32.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s31
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -p no:cacheprovider -k "split" > $OUT/tests.out 2>&1; rc=$?; tail -15 $OUT/tests.out
exit $rc
