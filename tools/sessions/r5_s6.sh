#!/bin/bash
# Round 5, GPU session 6: per-position host issue time and GPU completion time inside a rank's cold 8-position share.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s6
mkdir -p $OUT
for sim in Fresnel RayT; do
  for r in 7 0; do
    PSX_EMULATE_IDLE_MS=1 timeout -k 10 200 python bench.py --emulate-rank $r --emulate-world 8 --emulate-sim $sim 2>/dev/null | tail -1 | tee -a $OUT/trace.out
  done
done
