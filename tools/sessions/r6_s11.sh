#!/bin/bash
# Round 6, GPU session 11: first run of the two-round power-of-two kernel for ~16384-sample lines (fresnel_p2x.hip): parity, then
# config 5's timing.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s11
mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -x -q -k "power_of_two" > $OUT/t1.out 2>&1; rc=$?; echo "p2 tests rc $rc"; tail -15 $OUT/t1.out
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_large.py -x -q -k "16384 or partitioned" > $OUT/t2.out 2>&1; rc=$?; echo "large rc $rc"; tail -5 $OUT/t2.out
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py --only-configs --configs 16384 > $OUT/cfg5.out 2> $OUT/cfg5.err; echo "cfg5 rc $?"; tail -c 1800 $OUT/cfg5.out
