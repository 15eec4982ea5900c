#!/bin/bash
# Round 5, GPU session 57: membrane splat variants (PSX_MEMBRANE_SPLAT 0 / 1 / 2: conversion per row / repeated addition / lengths in the accumulator's unit) -- A/B as whole libraries, two
# rounds on one box; membrane parity tests on the variant.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s57
mkdir -p $OUT
cp paresis_amd/libparesis_hip.so $OUT/../.product.so
for round in 1 2; do
  for v in 0 1 2; do
    cp tools/ab/libparesis_hip_minc$v.so paresis_amd/libparesis_hip.so
    timeout -k 10 200 python tools/time_positions.py 4096 32 --sim RT > $OUT/pos_inc${v}_$round.out 2>&1; echo "inc $v:" $(grep -o "k_membrane x1 [0-9.]*" $OUT/pos_inc${v}_$round.out) $(grep -o "= [0-9.]* ms per position" $OUT/pos_inc${v}_$round.out) | tee -a $OUT/ab.out
  done
done
cp tools/ab/libparesis_hip_minc2.so paresis_amd/libparesis_hip.so
timeout -k 10 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "membrane" > $OUT/tests_inc2.out 2>&1; echo "membrane tests on variant 2: rc $?"; tail -2 $OUT/tests_inc2.out
cp $OUT/../.product.so paresis_amd/libparesis_hip.so; rm -f $OUT/../.product.so
