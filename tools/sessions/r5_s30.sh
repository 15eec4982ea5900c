#!/bin/bash
# Round 5, GPU session 30: the other half of the 100x differential tests (Fresnel families and the whole chains).
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s30
mkdir -p $OUT
PSX_FUZZ=100 timeout -k 10 1150 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider -k "not (refraction or fastloop or darkfield or membrane or detector)" > $OUT/fuzz100_b.out 2>&1; echo "b rc $?"; tail -3 $OUT/fuzz100_b.out
