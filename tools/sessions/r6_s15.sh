#!/bin/bash
# Round 6, GPU session 15: the exact bad positions of a line (two-round kernel, r6s13's failure)
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s15
mkdir -p $OUT
timeout -k 10 300 python tools/diag_p2x.py 16384 36 1 > $OUT/d1.out 2>&1; grep -v amdgpu $OUT/d1.out | cut -c1-1500
