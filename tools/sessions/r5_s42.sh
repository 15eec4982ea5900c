#!/bin/bash
# Round 5, GPU session 42: the halo of a reproducible experiment at oversampling 4 (8192^2 study grid, hops 1.6 / 3.6 m): RT positions, halo 6 / 8 / 12.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s42
mkdir -p $OUT
for h in 4 6 8 12; do
  timeout -k 10 300 python tools/time_positions.py 8192 16 --sim RT --ov 4 --halo $h > $OUT/pos_h$h.out 2>&1; echo "ov 4 halo $h:" $(grep -o "= [0-9.]* ms per position" $OUT/pos_h$h.out) $(grep -o "k_refract_near x3 [0-9.]*" $OUT/pos_h$h.out) $(grep -o "k_refract_far_add x3 [0-9.]*" $OUT/pos_h$h.out) $(grep -o "k_refract_far_fold x3 [0-9.]*" $OUT/pos_h$h.out) | tee -a $OUT/ab.out
done
