#!/bin/bash
# Round 6, GPU session 31: the parked round-O input of the two-round kernel, ONE change at a time (tools/ab libraries x1..x4,
# PSX_X_STEP): which step breaks parity?  The XL shapes of the power-of-two test + the diagnostic, per library.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s31
mkdir -p $OUT
cp paresis_amd/libparesis_hip.so $OUT/.lib_saved.so
for t in x1 x2 x3 x4; do
  cp tools/ab/libparesis_hip_$t.so paresis_amd/libparesis_hip.so
  timeout -k 10 240 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "power_of_two and (shape11 or shape12 or shape13 or shape14 or shape15)" > $OUT/tests_$t.out 2>&1; echo "$t tests rc $?"; tail -2 $OUT/tests_$t.out
  timeout -k 10 120 python tools/diag_p2x.py 16384 36 2 > $OUT/diag_a_$t.out 2>&1; grep "^rep" $OUT/diag_a_$t.out | cut -c1-200
  timeout -k 10 120 python tools/diag_p2x.py 36 16384 1 > $OUT/diag_b_$t.out 2>&1; grep "^rep" $OUT/diag_b_$t.out | cut -c1-200
done
cp $OUT/.lib_saved.so paresis_amd/libparesis_hip.so
