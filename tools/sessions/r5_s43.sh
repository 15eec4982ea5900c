#!/bin/bash
# Round 5, GPU session 43: halo rule from the rays' reach + the replay unit from the experiment's scale: chain / main tests, RT positions both modes.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s43
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_experiment.py tests/test_gpu_main.py tests/test_gpu_fuzz.py -m gpu -x -q -p no:cacheprovider -k "chain or main or xml or ranks or reproducible or halo" > $OUT/tests.out 2>&1; rc=$?; tail -3 $OUT/tests.out
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
timeout -k 10 300 python tools/time_positions.py 4096 64 --sim RT > $OUT/pos_det.out 2>&1; echo "reproducible:" $(grep -o "= [0-9.]* ms per position" $OUT/pos_det.out) | tee -a $OUT/ab.out
timeout -k 10 300 python tools/time_positions.py 4096 64 --sim RT --float-atomics > $OUT/pos_float.out 2>&1; echo "float atomics:" $(grep -o "= [0-9.]* ms per position" $OUT/pos_float.out) | tee -a $OUT/ab.out
done
