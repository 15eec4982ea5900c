#!/bin/bash
# Round 4, GPU session 20: after the refraction edit: whole GPU suite; long-line Fresnel fuzz x6.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s20
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
step suite 1000 python -m pytest tests -m gpu -q -x --durations=10
tail -16 $OUT/suite.out
PSX_FUZZ=6 step long6 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "long_lines"
grep -E "^FAILED|passed|failed" $OUT/long6.out | head -40
