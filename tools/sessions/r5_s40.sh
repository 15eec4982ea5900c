#!/bin/bash
# Round 5, GPU session 40: the replay's unit from the caller's scale (Experiment: incident intensity): tests, position timings.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s40
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_experiment.py tests/test_gpu_main.py tests/test_gpu_fuzz.py -m gpu -x -q -p no:cacheprovider -k "replay or order or deterministic or chain or reproducible or ranks or xml or split" > $OUT/tests.out 2>&1; rc=$?; tail -4 $OUT/tests.out
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/time_positions.py 4096 48 --sim RT > $OUT/pos_det.out 2>&1; grep -v "per position (host" $OUT/pos_det.out | tail -2
timeout -k 10 300 python tools/time_positions.py 4096 48 --sim RT --float-atomics > $OUT/pos_float.out 2>&1; grep -v "per position (host" $OUT/pos_float.out | tail -2
