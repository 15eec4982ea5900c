#!/bin/bash
# Round 5, GPU session 10 (side stream warmed in prepare()): sink widens each round behind its arrival (side stream), output blocks reserved before the loop:
# the RCCL / two-rank tests, then rank 0's and rank 7's emulated share, twice.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s10
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_main.py -m gpu -x -q -k "rccl or ranks" -p no:cacheprovider > $OUT/tests.out 2>&1; rc=$?; tail -3 $OUT/tests.out
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
  for sim in Fresnel RayT; do
    for r in 0 7; do
      PSX_EMULATE_IDLE_MS=1 timeout -k 10 200 python bench.py --emulate-rank $r --emulate-world 8 --emulate-sim $sim 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$sim rank $r', d['cold_ms'], d['warm_ms'], d['cold_host_issue_ms_per_position'])" | tee -a $OUT/trace.out
    done
  done
done
