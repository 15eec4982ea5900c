#!/bin/bash
# Round 6, GPU session 12: where do the rounds of the two-round kernel spend their time at 16384^2 (stamps of pass 2 and pass 1)?
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s12
mkdir -p $OUT
timeout -k 10 300 python tools/stamp_fresnel.py 16384 4 > $OUT/stamp_p2.out 2>&1; echo "stamp rc $?"; grep -v amdgpu.ids $OUT/stamp_p2.out
PSX_SWITCHES="stamp_pass1=1" timeout -k 10 300 python tools/stamp_fresnel.py 16384 4 > $OUT/stamp_p1.out 2>&1; echo "stamp pass1 rc $?"; grep -v amdgpu.ids $OUT/stamp_p1.out
PSX_SWITCHES="stamp_pass1=1 stamp_round=0" timeout -k 10 300 python tools/stamp_fresnel.py 16384 4 > $OUT/stamp_p1_r0.out 2>&1; echo "stamp pass1 round 0 rc $?"; grep -A14 "workgroups:" $OUT/stamp_p1_r0.out
