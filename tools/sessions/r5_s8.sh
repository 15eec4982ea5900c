#!/bin/bash
# Round 5, GPU session 8: the 2.5 ms host stall at a share's second position -- with and without the allocator query, twice each,
# on one box.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s8
mkdir -p $OUT
for rep in 1 2; do
  for tr in "" 1; do
    PSX_EMULATE_ALLOC_TRACE=$tr PSX_EMULATE_IDLE_MS=1 timeout -k 10 200 python bench.py --emulate-rank 7 --emulate-world 8 --emulate-sim RayT 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('trace=$tr', d['cold_ms'], d['warm_ms'], d['cold_host_issue_ms_per_position'], d['allocs'])" | tee -a $OUT/trace.out
  done
done
