#!/bin/bash
# Round 5, GPU session 60: the membrane kernel without its splat (timing experiment, wrong images: variant 3) against variant 2, two rounds.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s60
mkdir -p $OUT
cp paresis_amd/libparesis_hip.so $OUT/../.product.so
for round in 1 2; do
  for v in 2 3; do
    cp tools/ab/libparesis_hip_minc$v.so paresis_amd/libparesis_hip.so
    echo "variant $v:" $(timeout -k 10 200 python tools/time_membrane.py 4096 300 2>&1 | tail -1) | tee -a $OUT/ab.out
  done
done
cp $OUT/../.product.so paresis_amd/libparesis_hip.so; rm -f $OUT/../.product.so
