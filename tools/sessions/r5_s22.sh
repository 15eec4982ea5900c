#!/bin/bash
# Round 5, GPU session 22: the default bench line on the final build (as the driver runs it), wall clock included.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s22
mkdir -p $OUT
t0=$(date +%s)
timeout -k 10 1000 python bench.py > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc $? in $(( $(date +%s) - t0 )) s"
tail -2 $OUT/bench.err
