#!/bin/bash
# Round 6, GPU session 27: the rest of the GPU suite (from test_gpu_main on), smoke, the default bench.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s27
mkdir -p $OUT
timeout -k 10 700 python -m pytest tests/test_gpu_main.py tests/test_dist_gloo.py -x -q -m gpu > $OUT/tests.out 2>&1; rc=$?; echo "gpu main rc $rc"; tail -4 $OUT/tests.out
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python __graft_entry__.py smoke > $OUT/smoke.out 2>&1; echo "smoke rc $?"; tail -2 $OUT/smoke.out
timeout -k 10 500 python bench.py > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc $?"; tail -c 600 $OUT/bench.out
