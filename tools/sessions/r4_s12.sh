#!/bin/bash
# Round 4, GPU session 12: seeded random differential tests (tests/test_gpu_fuzz.py), default size then PSX_FUZZ=6.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s12
mkdir -p $OUT
step() {
  local name=$1 to=$2; shift 2
  echo "== $name" | tee -a $OUT/progress.log
  timeout -k 10 $to "$@" > $OUT/$name.out 2> $OUT/$name.err
  local rc=$?
  echo "$name rc=$rc" | tee -a $OUT/progress.log
  if [ $rc -ge 124 ]; then echo "ABORT after $name" | tee -a $OUT/progress.log; exit 1; fi
  return 0
}
step fuzz1 500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --durations=8
tail -15 $OUT/fuzz1.out
PSX_FUZZ=6 step fuzz6 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q
grep -E "^FAILED|passed|failed" $OUT/fuzz6.out | head -60
