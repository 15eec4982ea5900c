#!/bin/bash
# Round 4, GPU session 36: kernels of a membrane position at 4096^2 (both chains), current build.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s36
mkdir -p $OUT
timeout -k 10 400 python tools/time_positions.py 4096 16 > $OUT/pos.out 2> $OUT/pos.err || { echo FAILED; tail -5 $OUT/pos.err; exit 1; }
cat $OUT/pos.out
