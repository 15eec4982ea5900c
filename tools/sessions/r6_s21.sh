#!/bin/bash
# Round 6, GPU session 22 (step B on the committed kernel: the round compiled twice): bisecting the two-round kernel's failure: ye requested in one piece behind the butterfly (x1), the line
# fetched in one piece (x2), both (x12) -- each on top of the diagnostic form whose round O takes its input from LDS again.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s22
mkdir -p $OUT
cp paresis_amd/libparesis_hip.so $OUT/.saved.so
for v in sB; do
  cp tools/ab/libparesis_hip_$v.so paresis_amd/libparesis_hip.so
  timeout -k 10 300 python tools/diag_p2x.py 16384 36 1 > $OUT/$v.out 2>&1; echo $v; grep "^rep" $OUT/$v.out | cut -c1-170
done
cp $OUT/.saved.so paresis_amd/libparesis_hip.so
