#!/bin/bash
# Round 5, GPU session 34: does the replay run faster when concurrently running waves work on tiles far apart (psx_debug_switch
# "far_stride": lists walked in a stride coprime to their number)?  16384^2 and 4096^2, float atomics and fixed point.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s34
mkdir -p $OUT
for st in 0 7919 100003 37; do
  for mode in float reproducible; do
    PSX_SWITCHES="far_stride=$st" timeout -k 10 400 python tools/halo_sweep.py 16384 4 $mode > $OUT/h16384_${st}_$mode.out 2>&1; grep "halo 8\|halo 12" $OUT/h16384_${st}_$mode.out | sed "s/^/stride $st /" | cut -c1-50,150-420 | tee -a $OUT/ab.out
  done
  PSX_SWITCHES="far_stride=$st" timeout -k 10 200 python tools/halo_sweep.py 4096 2 float > $OUT/h4096_${st}.out 2>&1; grep "halo 4:" $OUT/h4096_${st}.out | sed "s/^/stride $st /" | cut -c1-40,150-420 | tee -a $OUT/ab.out
done
