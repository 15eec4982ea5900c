#!/bin/bash
# Round 6, GPU session 18: two-round kernel, failure of r6s13 / r6s16: does it come from the parked round-O input?  The diagnostic
# build (PSX_P2X_DBG=2: as 1, and round E does not store the parked input at all) against the build in the tree.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s18
mkdir -p $OUT
timeout -k 10 300 python tools/diag_p2x.py 16384 36 1 > $OUT/tree.out 2>&1; grep "^rep" $OUT/tree.out
cp paresis_amd/libparesis_hip.so $OUT/.saved.so
cp tools/ab/libparesis_hip_xdbg2.so paresis_amd/libparesis_hip.so
timeout -k 10 300 python tools/diag_p2x.py 16384 36 1 > $OUT/dbg2.out 2>&1; grep "^rep" $OUT/dbg2.out
cp $OUT/.saved.so paresis_amd/libparesis_hip.so
