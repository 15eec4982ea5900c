#!/bin/bash
# Round 6, GPU session 37: the whole GPU suite, smoke() and the default bench on the final build; then the seeded differential
# tests at 30 x their committed size.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s37
mkdir -p $OUT
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $OUT/tests.out 2>&1; rc=$?; echo "gpu suite rc $rc"; tail -4 $OUT/tests.out
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python __graft_entry__.py smoke > $OUT/smoke.out 2>&1; echo "smoke rc $?"; tail -2 $OUT/smoke.out
timeout -k 10 500 python bench.py > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc $?"
PSX_FUZZ=30 timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > $OUT/fuzz30.out 2>&1; echo "fuzz x30 rc $?"; tail -3 $OUT/fuzz30.out
