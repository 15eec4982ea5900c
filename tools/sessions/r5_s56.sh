#!/bin/bash
# Round 5, GPU session 56: membrane splat, the row offset by repeated addition (PSX_MEMBRANE_INC) -- A/B as whole libraries, two
# rounds on one box; membrane parity tests on the variant.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s56
mkdir -p $OUT
cp paresis_amd/libparesis_hip.so $OUT/../.product.so
for round in 1 2; do
  for v in 0 1; do
    cp tools/ab/libparesis_hip_minc$v.so paresis_amd/libparesis_hip.so
    timeout -k 10 200 python tools/time_positions.py 4096 32 --sim RT > $OUT/pos_inc${v}_$round.out 2>&1; echo "inc $v:" $(grep -o "k_membrane x1 [0-9.]*" $OUT/pos_inc${v}_$round.out) $(grep -o "= [0-9.]* ms per position" $OUT/pos_inc${v}_$round.out) | tee -a $OUT/ab.out
  done
done
cp tools/ab/libparesis_hip_minc1.so paresis_amd/libparesis_hip.so
timeout -k 10 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "membrane" > $OUT/tests_inc1.out 2>&1; echo "membrane tests on inc 1: rc $?"; tail -2 $OUT/tests_inc1.out
cp $OUT/../.product.so paresis_amd/libparesis_hip.so; rm -f $OUT/../.product.so
