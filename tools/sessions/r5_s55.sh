#!/bin/bash
# Round 5, GPU session 55: the differential fuzz at 30x on the round's final build (after psx_detect_multi_f32, the in-place reference
# image and v_exp_f32 in the refraction staging).
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s55
mkdir -p $OUT
PSX_FUZZ=30 timeout -k 10 1150 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider > $OUT/fuzz30.out 2>&1; echo "fuzz30 rc $?"; tail -3 $OUT/fuzz30.out
