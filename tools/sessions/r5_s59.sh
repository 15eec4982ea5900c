#!/bin/bash
# Round 5, GPU session 59: membrane splat variants (PSX_MEMBRANE_SPLAT 0 / 1 / 2) timed alone over 300 positions, four rounds on one box.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s59
mkdir -p $OUT
cp paresis_amd/libparesis_hip.so $OUT/../.product.so
for round in 1 2 3 4; do
  for v in 0 1 2; do
    cp tools/ab/libparesis_hip_minc$v.so paresis_amd/libparesis_hip.so
    echo "variant $v:" $(timeout -k 10 200 python tools/time_membrane.py 4096 300 2>&1 | tail -1) | tee -a $OUT/ab.out
  done
done
cp $OUT/../.product.so paresis_amd/libparesis_hip.so; rm -f $OUT/../.product.so
