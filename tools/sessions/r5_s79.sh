#!/bin/bash
# Round 5, GPU session 79: the membrane chord's reciprocal square root as v_rsq_f64 (no float32 round trip) -- A/B, two rounds; membrane
# parity tests on the variant.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s79
mkdir -p $OUT
cp paresis_amd/libparesis_hip.so $OUT/../.product.so
for round in 1 2; do
  for v in 0 1; do
    cp tools/ab/libparesis_hip_rsq$v.so paresis_amd/libparesis_hip.so
    echo "rsq64 $v:" $(timeout -k 10 200 python tools/time_membrane.py 4096 300 2>&1 | tail -1) | tee -a $OUT/ab.out
  done
done
cp tools/ab/libparesis_hip_rsq1.so paresis_amd/libparesis_hip.so
timeout -k 10 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "membrane" > $OUT/tests.out 2>&1; echo "membrane tests on rsq64: rc $?" $(tail -1 $OUT/tests.out)
cp $OUT/../.product.so paresis_amd/libparesis_hip.so; rm -f $OUT/../.product.so
