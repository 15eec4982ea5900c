#!/bin/bash
# Round 6, GPU session 6: the whole GPU suite and smoke() on the power-of-two line kernels (state of r6s4), then the default bench.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s6
mkdir -p $OUT
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $OUT/tests.out 2>&1; rc=$?; echo "gpu suite rc $rc"; tail -8 $OUT/tests.out
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python __graft_entry__.py smoke > $OUT/smoke.out 2>&1; echo "smoke rc $?"; tail -2 $OUT/smoke.out
timeout -k 10 500 python bench.py > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc $?"; tail -c 1500 $OUT/bench.out
