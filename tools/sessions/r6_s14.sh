#!/bin/bash
# Round 6, GPU session 14: diagnostic of the reworked two-round kernel's failure (r6s13): which lines / positions are wrong?
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s14
mkdir -p $OUT
timeout -k 10 300 python tools/diag_p2x.py 16384 36 1 > $OUT/d1.out 2>&1; grep -v amdgpu $OUT/d1.out
timeout -k 10 300 python tools/diag_p2x.py 16384 36 2 > $OUT/d2.out 2>&1; grep -v amdgpu $OUT/d2.out
timeout -k 10 300 python tools/diag_p2x.py 36 16384 2 > $OUT/d3.out 2>&1; grep -v amdgpu $OUT/d3.out
timeout -k 10 300 python tools/diag_p2x.py 16384 520 2 > $OUT/d4.out 2>&1; grep -v amdgpu $OUT/d4.out
