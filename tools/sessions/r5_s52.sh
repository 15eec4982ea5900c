#!/bin/bash
# Round 5, GPU session 52: what ONE refraction workgroup per CU costs (psx_debug_switch near_lds_pad: LDS padded so that the second
# workgroup no longer fits) -- the price a two-phase "sample + reference in one launch" tile kernel would pay for its LDS.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s52
mkdir -p $OUT
for pad in 0 8 0 8; do
  PSX_SWITCHES="near_lds_pad=$pad" timeout -k 10 300 python tools/time_positions.py 4096 32 --sim RT > $OUT/pos_pad$pad.out 2>&1 || { echo "failed"; tail -3 $OUT/pos_pad$pad.out; exit 1; }
  echo "pad $pad KiB:"; grep -E "positions of|library kernels" $OUT/pos_pad$pad.out
done
for pad in 0 8; do
  echo "halo sweep (4 distances), pad $pad KiB:"; PSX_SWITCHES="near_lds_pad=$pad" timeout -k 10 300 python tools/halo_sweep.py 4096 2 > $OUT/sweep_pad$pad.out 2>&1; tail -4 $OUT/sweep_pad$pad.out
done
