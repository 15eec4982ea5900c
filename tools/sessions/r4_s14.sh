#!/bin/bash
# Round 4, GPU session 14: dark-field split in the reference's order of operations; fuzz file incl. chains; dark-field tests.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s14
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
step df_tests 500 python -m pytest tests -m gpu -q -k "darkfield"
tail -3 $OUT/df_tests.out
step fuzz1 500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --durations=5
grep -E "^FAILED|passed|failed" $OUT/fuzz1.out | head -40
PSX_FUZZ=8 step fuzz8 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -q
grep -E "^FAILED|passed|failed" $OUT/fuzz8.out | head -60
step dftime 300 python tools/time_darkfield.py
tail -12 $OUT/dftime.out
