#!/bin/bash
# Round 5, GPU session 7: is the 2.5 ms host stall at a share's second position the caching allocator going to hipMalloc?
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s7
mkdir -p $OUT
for res in "" 4; do
  for r in 7 0; do
    PSX_EMULATE_RESERVE=$res PSX_EMULATE_IDLE_MS=1 timeout -k 10 200 python bench.py --emulate-rank $r --emulate-world 8 --emulate-sim RayT 2>/dev/null | tail -1 | tee -a $OUT/trace.out
  done
done
