#!/bin/bash
# Round 5, GPU session 74: k_poisson with the unsettled pixels carried over on a per-wave list (one full-wave step at a time): the
# sampler's tests (statistics; 16-byte path == scalar path draw for draw), keys, chains; then the kernel by event pairs.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s74
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_main.py tests/test_gpu_experiment.py tests/test_gpu_large.py -m gpu -x -q -p no:cacheprovider -k "poisson or main or chain or noise or key or xml" > $OUT/tests.out 2>&1; rc=$?; echo "tests rc $rc"; tail -2 $OUT/tests.out
[ $rc -eq 0 ] || exit $rc
for r in 1 2; do timeout -k 10 200 python tools/time_poisson.py 2>&1 | tail -1; done | tee $OUT/poisson.out
timeout -k 10 200 python tools/time_poisson.py 2048 30 2>&1 | tail -1 | tee -a $OUT/poisson.out
timeout -k 10 200 python tools/time_poisson.py 2048 4 2>&1 | tail -1 | tee -a $OUT/poisson.out
