#!/bin/bash
# Round 5, GPU session 37: one shared zero stack for Propag / White away from position 0: chain + main tests, position timings.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s37
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_experiment.py tests/test_gpu_main.py tests/test_gpu_fuzz.py -m gpu -x -q -p no:cacheprovider -k "chain or main or xml or ranks or position or rccl or reproducible" > $OUT/tests.out 2>&1; rc=$?; tail -3 $OUT/tests.out
[ $rc -eq 0 ] || exit $rc
for sim in RT Fresnel; do timeout -k 10 300 python tools/time_positions.py 4096 48 --sim $sim > $OUT/pos_$sim.out 2>&1; grep -v "per position (host" $OUT/pos_$sim.out | tail -2; done
