#!/bin/bash
# Round 4, GPU session 29: detector kernels by name at the bench geometries (source blur 0.036 px, PSF 1.2 px).
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s29
mkdir -p $OUT
for a in "4096 2 0.036" "4096 2 1.3" "2048 1 0.036" "8192 4 0.036" "16384 4 0.072"; do
  set -- $a
  timeout -k 10 300 python tools/time_detector.py $1 $2 $3 > $OUT/det_$1_$2_$3.out 2> $OUT/det_$1_$2_$3.err || { echo "FAILED $a"; exit 1; }
  cat $OUT/det_$1_$2_$3.out
done
