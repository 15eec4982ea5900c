#!/bin/bash
# Round 5, GPU session 41: which halo should a reproducible experiment at oversampling 2 take?  RT positions, fixed-point replay, halo 4 / 6 / 8, twice.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s41
mkdir -p $OUT
for rep in 1 2; do for h in 4 6 8; do
  timeout -k 10 300 python tools/time_positions.py 4096 64 --sim RT --halo $h > $OUT/pos_h$h.out 2>&1; echo "halo $h:" $(grep -o "= [0-9.]* ms per position" $OUT/pos_h$h.out) $(grep -o "k_refract_near x3 [0-9.]*" $OUT/pos_h$h.out) $(grep -o "k_refract_far_add x3 [0-9.]*" $OUT/pos_h$h.out) $(grep -o "k_refract_far_fold x3 [0-9.]*" $OUT/pos_h$h.out) | tee -a $OUT/ab.out
done; done
