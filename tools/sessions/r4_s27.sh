#!/bin/bash
# Round 4, GPU session 27: where the dark-field gather's time goes: width map everywhere / inside the sample / nowhere.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s27
mkdir -p $OUT
for w in sample all none; do
  timeout -k 10 300 python tools/time_darkfield.py 20 chain $w > $OUT/df_$w.out 2> $OUT/df_$w.err || { echo "FAILED $w"; exit 1; }
  cat $OUT/df_$w.out
done
