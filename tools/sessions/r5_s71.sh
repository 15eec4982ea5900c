#!/bin/bash
# Round 5, GPU session 71: k_psf_tile with 16 against 32 output rows per workgroup (25 against 42 KB of LDS), two rounds.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s71
mkdir -p $OUT
cp paresis_amd/libparesis_hip.so $OUT/../.product.so
for round in 1 2; do
  for v in 16 32; do
    cp tools/ab/libparesis_hip_tr$v.so paresis_amd/libparesis_hip.so
    echo "rows $v:" $(timeout -k 10 200 python tools/time_detector.py 2>&1 | tail -1) | tee -a $OUT/ab.out
    echo "rows $v:" $(timeout -k 10 200 python tools/time_detector.py 16384 4 4 2>&1 | tail -1) | tee -a $OUT/ab.out
  done
done
cp $OUT/../.product.so paresis_amd/libparesis_hip.so; rm -f $OUT/../.product.so
