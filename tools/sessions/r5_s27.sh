#!/bin/bash
# Round 5, GPU session 27: banded dark-field gather with a staged entry shared by a thread's four outputs: tests, then timings.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s27
mkdir -p $OUT
PSX_FUZZ=5 timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_experiment.py -m gpu -x -q -p no:cacheprovider -k "darkfield" > $OUT/tests.out 2>&1; rc=$?; tail -3 $OUT/tests.out
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/time_positions.py 4096 16 --sim RT --scatter > $OUT/scatter_mono.out 2>&1; grep -v "per position (host" $OUT/scatter_mono.out | tail -2
timeout -k 10 600 python tools/time_positions.py 4096 3 --sim RT --poly 25 --scatter > $OUT/scatter.out 2>&1; grep -v "per position (host" $OUT/scatter.out | tail -2
