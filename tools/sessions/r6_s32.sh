#!/bin/bash
# Round 6, GPU session 32: step 2 of the parked round-O input (round O reads it back) fails -- with the loads' cache policy bits,
# with the memory counter drained either side, with the two halves of the workgroup's buffer swapped?
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s32
mkdir -p $OUT
cp paresis_amd/libparesis_hip.so $OUT/.lib_saved.so
for t in x2 x2aux x2aux3 x2drain x2swap; do
  cp tools/ab/libparesis_hip_$t.so paresis_amd/libparesis_hip.so
  echo "== $t"
  timeout -k 10 120 python tools/diag_p2x.py 36 16384 1 > $OUT/diag_b_$t.out 2>&1; grep -A34 "^bad outputs" $OUT/diag_b_$t.out | head -36; grep "^rep" $OUT/diag_b_$t.out | cut -c1-200
done
cp $OUT/.lib_saved.so paresis_amd/libparesis_hip.so
