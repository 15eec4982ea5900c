#!/bin/bash
# Round 5, GPU session 16: the chain's dark-field hop fused (masked refractions from the thickness maps, width-map tables cached per
# energy): dark-field tests + chain fuzz, then 25-energy positions: plain, scattering (thin sample: narrow patches), scattering.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s16
mkdir -p $OUT
PSX_FUZZ=5 timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_experiment.py -m gpu -x -q -p no:cacheprovider -k "darkfield or chain_rt" > $OUT/tests.out 2>&1; rc=$?; tail -3 $OUT/tests.out
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/time_positions.py 4096 6 --sim RT --poly 25 > $OUT/plain.out 2>&1; grep -v "per position (host" $OUT/plain.out | tail -2
timeout -k 10 300 python tools/time_positions.py 4096 6 --sim RT --poly 25 --scatter --thin 200 > $OUT/thin.out 2>&1; grep -v "per position (host" $OUT/thin.out | tail -2
timeout -k 10 300 python tools/time_positions.py 4096 16 --sim RT --scatter --thin 30 > $OUT/thin_mono.out 2>&1; grep -v "per position (host" $OUT/thin_mono.out | tail -2
timeout -k 10 300 python tools/time_positions.py 4096 16 --sim RT --scatter > $OUT/scatter_mono.out 2>&1; grep -v "per position (host" $OUT/scatter_mono.out | tail -2
