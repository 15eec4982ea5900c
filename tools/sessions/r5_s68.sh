#!/bin/bash
# Round 5, GPU session 68: where k_band_pair's time goes -- timing experiments (PSX_BP_OFF, wrong images, never shipped): 0 whole
# kernels, 1 no column operator, 2 no row operator / stores, 3 neither (fetch + staging alone).
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s68
mkdir -p $OUT
cp paresis_amd/libparesis_hip.so $OUT/../.product.so
for round in 1 2; do
  for v in 0 1 2 3; do
    cp tools/ab/libparesis_hip_bp$v.so paresis_amd/libparesis_hip.so
    echo "off $v:" $(timeout -k 10 200 python tools/time_detector.py 2>&1 | tail -1) | tee -a $OUT/ab.out
  done
done
cp $OUT/../.product.so paresis_amd/libparesis_hip.so; rm -f $OUT/../.product.so
