#!/bin/bash
# Round 6, GPU session 24: is the committed two-round kernel robust?  The same source perturbed (a scheduling barrier + s_sleep at the
# top of the round loop; -O2) against the oracle, and the tree build.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s24
mkdir -p $OUT
cp paresis_amd/libparesis_hip.so $OUT/.saved.so
for v in v1p v1o2; do
  cp tools/ab/libparesis_hip_$v.so paresis_amd/libparesis_hip.so
  timeout -k 10 300 python tools/diag_p2x.py 16384 36 2 > $OUT/$v.out 2>&1; echo $v; grep "^rep" $OUT/$v.out | cut -c1-150
done
cp $OUT/.saved.so paresis_amd/libparesis_hip.so
timeout -k 10 300 python tools/diag_p2x.py 16384 36 2 > $OUT/tree.out 2>&1; echo tree; grep "^rep" $OUT/tree.out | cut -c1-150
