#!/bin/bash
# Round 5, GPU session 17: whole GPU suite on the current build, then the dark-field position timings with the accumulating re-splat.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s17
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/tests.out 2>&1; rc=$?; tail -3 $OUT/tests.out
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/time_positions.py 4096 6 --sim RT --poly 25 > $OUT/plain.out 2>&1; grep -v "per position (host" $OUT/plain.out | tail -2
timeout -k 10 300 python tools/time_positions.py 4096 6 --sim RT --poly 25 --scatter --thin 200 > $OUT/thin.out 2>&1; grep -v "per position (host" $OUT/thin.out | tail -2
timeout -k 10 300 python tools/time_positions.py 4096 16 --sim RT > $OUT/plain_mono.out 2>&1; grep -v "per position (host" $OUT/plain_mono.out | tail -2
timeout -k 10 300 python tools/time_positions.py 4096 16 --sim RT --scatter --thin 30 > $OUT/thin_mono.out 2>&1; grep -v "per position (host" $OUT/thin_mono.out | tail -2
