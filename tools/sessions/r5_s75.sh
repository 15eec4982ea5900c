#!/bin/bash
# Round 5, GPU session 73: where k_poisson's time goes -- timing experiments (PSX_POISSON_OFF, wrong draws, never shipped): 0 whole kernel,
# 1 no exact test (the squeeze's candidate taken), 2 one Philox round instead of ten, 3 both.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s75
mkdir -p $OUT
cp paresis_amd/libparesis_hip.so $OUT/../.product.so
for round in 1 2; do
  for v in 0 4; do
    cp tools/ab/libparesis_hip_po$v.so paresis_amd/libparesis_hip.so
    echo "off $v:" $(timeout -k 10 200 python tools/time_poisson.py 2>&1 | tail -1) | tee -a $OUT/ab.out
  done
done
cp $OUT/../.product.so paresis_amd/libparesis_hip.so; rm -f $OUT/../.product.so
