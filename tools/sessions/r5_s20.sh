#!/bin/bash
# Round 5, GPU session 20: halo 12 / 16 tile geometries: refraction tests + fuzz, halo sweep on config 5's grid in both replay modes.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s20
mkdir -p $OUT
PSX_FUZZ=4 timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_large.py tests/test_host_cpu.py -x -q -p no:cacheprovider -k "refract or order or spill or symbol" > $OUT/tests.out 2>&1; rc=$?; tail -3 $OUT/tests.out
[ $rc -eq 0 ] || exit $rc
for mode in float reproducible; do
  timeout -k 10 400 python tools/halo_sweep.py 16384 4 $mode > $OUT/halo_16384_$mode.out 2>&1 && grep "^N " $OUT/halo_16384_$mode.out
  timeout -k 10 300 python tools/halo_sweep.py 8192 4 $mode > $OUT/halo_8192_$mode.out 2>&1 && grep "^N " $OUT/halo_8192_$mode.out
done
