#!/bin/bash
# Round 4, GPU session 39: DUAL rounds with the upper engine waves' inverse stage A deferred into the next round's forward stages:
# hashes against the previous library, Fresnel tests, timing A/B twice.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s39
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
cp paresis_amd/libparesis_hip.so $OUT/lib_new.so
step hash_new 400 python tools/hash_fresnel.py
cp tools/ab/libparesis_hip_prev.so paresis_amd/libparesis_hip.so
step hash_prev 400 python tools/hash_fresnel.py
diff $OUT/hash_new.out $OUT/hash_prev.out > $OUT/hash_diff.txt && echo "HASHES IDENTICAL" | tee -a $OUT/progress.log || { echo "HASHES DIFFER" | tee -a $OUT/progress.log; cat $OUT/hash_diff.txt; cp $OUT/lib_new.so paresis_amd/libparesis_hip.so; rm -f $OUT/lib_new.so; exit 1; }
for r in 1 2; do
  cp tools/ab/libparesis_hip_prev.so paresis_amd/libparesis_hip.so
  step b_prev_$r 300 python bench.py --no-cpu-baseline --positions 0 --no-configs
  cp $OUT/lib_new.so paresis_amd/libparesis_hip.so
  step b_new_$r 300 python bench.py --no-cpu-baseline --positions 0 --no-configs
done
cp $OUT/lib_new.so paresis_amd/libparesis_hip.so; rm -f $OUT/lib_new.so
step tests 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_large.py tests/test_gpu_fuzz.py tests/test_gpu_experiment.py -m gpu -q -x -k "fresnel or propagate or work_queue or chain"
tail -3 $OUT/tests.out
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4s39/b_*.out")):
    j=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    print(os.path.basename(f), j["ms_per_step"], j["kernel_ms_per_step"])
PY
