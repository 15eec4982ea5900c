#!/bin/bash
# Round 5, GPU session 32: A/B -- waves of the tile kernel whose 64 sources all miss the tile skip their deposits (halo >= 8).
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s32
mkdir -p $OUT
cp paresis_amd/libparesis_hip.so /tmp/keep.so
cp tools/ab/libparesis_hip_skip1.so paresis_amd/libparesis_hip.so
PSX_FUZZ=3 timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -x -q -p no:cacheprovider -k "refract or order" > $OUT/tests_skip1.out 2>&1; rc=$?; tail -2 $OUT/tests_skip1.out
for rep in 1 2; do
for v in skip0 skip1; do
  cp tools/ab/libparesis_hip_$v.so paresis_amd/libparesis_hip.so
  for mode in float reproducible; do
    timeout -k 10 400 python tools/halo_sweep.py 16384 4 $mode > $OUT/halo_${v}_$mode.out 2>&1; grep "halo 8\|halo 12\|halo 16" $OUT/halo_${v}_$mode.out | sed "s/^/$v /" | cut -c1-60,150-400 | tee -a $OUT/ab.out
  done
done
done
cp /tmp/keep.so paresis_amd/libparesis_hip.so
exit $rc
