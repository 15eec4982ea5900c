#!/bin/bash
# Round 4, GPU session 35: checkpoint -- whole GPU suite and the default bench on the current build.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s35
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
step suite 1000 python -m pytest tests -m gpu -q -x
tail -3 $OUT/suite.out
step smoke 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
tail -1 $OUT/smoke.out
step bench 900 python bench.py
tail -c 3000 $OUT/bench.out | head -c 1500
