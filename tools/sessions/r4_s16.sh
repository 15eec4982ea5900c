#!/bin/bash
# Round 4, GPU session 16: k_refract_near without the extras branch and the hoisted lane mask: tests, fuzz x4 (with the new
# membrane and scattering-chain families), step timing twice.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s16
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
step tests 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_experiment.py -m gpu -q -x
tail -3 $OUT/tests.out
PSX_FUZZ=4 step fuzz4 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q
grep -E "^FAILED|passed|failed" $OUT/fuzz4.out | head -40
step bench_a 300 python bench.py --no-cpu-baseline --positions 0 --no-configs
step bench_b 300 python bench.py --no-cpu-baseline --positions 0 --no-configs
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4s16/bench_*.out")):
    j=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    print(os.path.basename(f), j["ms_per_step"], j["value"], j["kernel_ms_per_step"], j["parity"]["ok"])
PY
