#!/bin/bash
# Round 5, GPU session 61: the membrane layers kernel's tile (32x32 / 64x32 / 32x64 / 64x64 / 128x32), timed alone over 300 positions,
# two rounds on one box; membrane parity tests on each variant.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s61
mkdir -p $OUT
cp paresis_amd/libparesis_hip.so $OUT/../.product.so
for round in 1 2; do
  for v in 32x32 64x32 32x64 64x64 128x32; do
    cp tools/ab/libparesis_hip_mt$v.so paresis_amd/libparesis_hip.so
    echo "tile $v:" $(timeout -k 10 200 python tools/time_membrane.py 4096 300 2>&1 | tail -1) | tee -a $OUT/ab.out
  done
done
for v in 64x32 32x64 64x64 128x32; do
  cp tools/ab/libparesis_hip_mt$v.so paresis_amd/libparesis_hip.so
  timeout -k 10 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "membrane" > $OUT/tests_$v.out 2>&1; echo "membrane tests on $v: rc $?" $(tail -1 $OUT/tests_$v.out)
done
cp $OUT/../.product.so paresis_amd/libparesis_hip.so; rm -f $OUT/../.product.so
