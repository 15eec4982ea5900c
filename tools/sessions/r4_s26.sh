#!/bin/bash
# Round 4, GPU session 26: dark-field gather with 16-byte LDS reads (was narrowed to ds_read_b96): tests, timing.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s26
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
step tests 600 python -m pytest tests -m gpu -q -x -k "darkfield or refract or fuzz"
tail -3 $OUT/tests.out
step dftime1 300 python tools/time_darkfield.py
step dftime2 300 python tools/time_darkfield.py 20 chain
cat $OUT/dftime1.out $OUT/dftime2.out
