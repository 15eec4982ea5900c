#!/bin/bash
# Round 5, GPU session 65: hardware rcp / sqrt / log in the Poisson sampler; the membrane kernel's new tile in the position loop:
# Poisson + detector + membrane + main tests, position loops.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s65
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_main.py tests/test_gpu_experiment.py -m gpu -x -q -p no:cacheprovider > $OUT/tests.out 2>&1; rc=$?; echo "tests rc $rc"; tail -2 $OUT/tests.out
[ $rc -eq 0 ] || exit $rc
for sim in RT Fresnel; do timeout -k 10 300 python tools/time_positions.py 4096 48 --sim $sim > $OUT/pos_$sim.out 2>&1; grep -E "positions of|library kernels" $OUT/pos_$sim.out; done
