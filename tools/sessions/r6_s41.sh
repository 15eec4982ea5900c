#!/bin/bash
# Round 6, GPU session 41: the seeded differential tests at 100 x their committed size on the final build (8800 cases).
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s41
mkdir -p $OUT
PSX_FUZZ=100 timeout -k 10 1150 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > $OUT/fuzz100.out 2>&1; echo "fuzz x100 rc $?"; tail -3 $OUT/fuzz100.out
