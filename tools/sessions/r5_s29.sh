#!/bin/bash
# Round 5, GPU session 29: the seeded differential tests at 100x their committed size on the final build (in two halves, progress lines).
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s29
mkdir -p $OUT
PSX_FUZZ=100 timeout -k 10 1150 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider -k "refraction or fastloop or darkfield or membrane or detector" > $OUT/fuzz100_a.out 2>&1; echo "a rc $?"; tail -3 $OUT/fuzz100_a.out
