#!/bin/bash
# Round 5, GPU session 62: where the membrane kernel's time goes -- timing experiments (wrong images, never shipped): PSX_ML_OFF
# 0 whole kernel, 1 no splat, 2 no stores, 4 no search for spheres, 6 neither search nor stores (launch + zeroing alone).
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s62
mkdir -p $OUT
cp paresis_amd/libparesis_hip.so $OUT/../.product.so
for round in 1 2; do
  for v in 0 1 2 4 6; do
    cp tools/ab/libparesis_hip_moff$v.so paresis_amd/libparesis_hip.so
    echo "off $v:" $(timeout -k 10 200 python tools/time_membrane.py 4096 300 2>&1 | tail -1) | tee -a $OUT/ab.out
  done
done
cp $OUT/../.product.so paresis_amd/libparesis_hip.so; rm -f $OUT/../.product.so
