#!/bin/bash
# Round 4, GPU session 30: phase shares of the refraction tile kernel (stamps), 4096^2 one distance.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s30
mkdir -p $OUT
timeout -k 10 300 python tools/stamp_refract.py 4096 3.6 > $OUT/stamp.out 2> $OUT/stamp.err || { echo FAILED; tail -5 $OUT/stamp.err; exit 1; }
cat $OUT/stamp.out
