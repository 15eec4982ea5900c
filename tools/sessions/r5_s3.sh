#!/bin/bash
# Round 5, GPU session 3: where the order-independent replay's cost sits in a ray-tracing POSITION (4096^2, detector 2048^2):
# per-kernel breakdown with float atomics and with the fixed-point replay, halo 4 and 6.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s3
mkdir -p $OUT
for h in 4 6; do
  timeout -k 10 200 python tools/time_positions.py 4096 32 --sim RT --halo $h --float-atomics > $OUT/pos_float_h$h.out 2>&1 && grep -v "per position (host" $OUT/pos_float_h$h.out | tail -3
  timeout -k 10 200 python tools/time_positions.py 4096 32 --sim RT --halo $h > $OUT/pos_det_h$h.out 2>&1 && grep -v "per position (host" $OUT/pos_det_h$h.out | tail -3
done
