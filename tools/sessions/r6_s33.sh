#!/bin/bash
# Round 6, GPU session 33: which legs of the parked round-O input differ from the LDS-derived ones (PSX_X_CMP)?
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s33
mkdir -p $OUT
cp paresis_amd/libparesis_hip.so $OUT/.lib_saved.so
cp tools/ab/libparesis_hip_x1cmp.so paresis_amd/libparesis_hip.so
timeout -k 10 120 python tools/diag_p2x_cmp.py 36 16384 1 > $OUT/cmp_b.out 2>&1; grep -v "^wg" $OUT/cmp_b.out | head -150
timeout -k 10 120 python tools/diag_p2x_cmp.py 16384 36 2 > $OUT/cmp_a.out 2>&1; head -50 $OUT/cmp_a.out
cp $OUT/.lib_saved.so paresis_amd/libparesis_hip.so
