#!/bin/bash
# Round 4, GPU session 38: float far replay with neighbour-merged shares: tests, fuzz x3, timing at 4096^2 and config 5.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s38
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
step tests 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_experiment.py tests/test_gpu_main.py tests/test_gpu_large.py -m gpu -q -x -k "refract or chain or main or darkfield or 16384"
tail -3 $OUT/tests.out
PSX_FUZZ=3 step fuzz3 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "refraction or chain or darkfield"
tail -2 $OUT/fuzz3.out
step bench 300 python bench.py --no-cpu-baseline --positions 0 --no-configs
step cfg5 400 python bench.py --only-configs --configs 16384 --no-whole-image-parity
python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/r4s38/bench.out").read().splitlines() if l.startswith("{")][-1])
print("4096:", j["ms_per_step"], j["kernel_ms_per_step"], j["parity"]["refraction"])
j=json.loads([l for l in open("gpurun_out/r4s38/cfg5.out").read().splitlines() if l.startswith("{")][-1])
e=j["configs"]["16384"]; print("cfg5", e["ms"], e["step_frac_per_propagation"], e["kernel_ms_per_step"], e["refraction_halo_tuning_ms"], e["parity"]["refraction"], e["parity"]["refraction_axis1"])
PY
