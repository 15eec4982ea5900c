#!/bin/bash
# Round 5, GPU session 23: the whole GPU suite on the final build, then the seeded differential tests at 30x their committed size.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s23
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/tests.out 2>&1; rc=$?; tail -3 $OUT/tests.out
[ $rc -eq 0 ] || exit $rc
PSX_FUZZ=30 timeout -k 10 1100 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider > $OUT/fuzz30.out 2>&1; echo "fuzz rc $?"; tail -4 $OUT/fuzz30.out
