#!/bin/bash
# Round 5, GPU session 14: baseline of the dark-field branch in a 25-energy ray-tracing position (before energy batching).
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s14
mkdir -p $OUT
timeout -k 10 300 python tools/time_positions.py 4096 6 --sim RT --poly 25 > $OUT/plain.out 2>&1; grep -v "per position (host" $OUT/plain.out | tail -3
timeout -k 10 300 python tools/time_positions.py 4096 6 --sim RT --poly 25 --scatter > $OUT/scatter.out 2>&1; grep -v "per position (host" $OUT/scatter.out | tail -3
timeout -k 10 300 python tools/time_positions.py 4096 16 --sim RT --scatter > $OUT/scatter_mono.out 2>&1; grep -v "per position (host" $OUT/scatter_mono.out | tail -3
