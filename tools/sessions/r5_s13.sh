#!/bin/bash
# Round 5, GPU session 13: A/B of the membrane plan's cell side (8 / 16 / 32 pixels) on one box: k_membrane per position.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s13
mkdir -p $OUT
cp paresis_amd/libparesis_hip.so /tmp/keep.so
for rep in 1 2; do
for v in 32 16 8; do
  cp tools/ab/libparesis_hip_mc$v.so paresis_amd/libparesis_hip.so
  timeout -k 10 200 python tools/time_positions.py 4096 32 --sim RT > $OUT/pos_mc$v.out 2>&1; echo "cell $v:" $(grep -o "k_membrane x1 [0-9.]*" $OUT/pos_mc$v.out) $(grep -o "= [0-9.]* ms per position" $OUT/pos_mc$v.out) | tee -a $OUT/ab.out
done
done
cp /tmp/keep.so paresis_amd/libparesis_hip.so
