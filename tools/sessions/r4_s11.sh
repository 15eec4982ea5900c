#!/bin/bash
# Round 4, GPU session 11: DIF loaders keeping the line in registers: hashes vs the previous build, tests, timings, KEEP A/B.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s11
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
cp paresis_amd/libparesis_hip.so $OUT/lib_main.so
step hash_new 400 python tools/hash_fresnel.py
step tests_new 600 python -m pytest tests/test_gpu_large.py -m gpu -x -q -k "partitioned or 16384"
tail -2 $OUT/tests_new.out
step b16k_keep48 300 python bench.py --no-cpu-baseline --positions 0 --no-configs --size 16384 --steps 4 --warmup 1
for v in nokeep keep40 keep56; do
  cp tools/ab/libparesis_hip_$v.so paresis_amd/libparesis_hip.so
  if [ $v = nokeep ]; then step hash_prev 400 python tools/hash_fresnel.py; else
    step ${v}_tests 600 python -m pytest tests/test_gpu_large.py -m gpu -x -q -k "partitioned"
    if ! grep -q passed $OUT/${v}_tests.out || grep -q failed $OUT/${v}_tests.out; then echo "$v tests not clean" | tee -a $OUT/progress.log; continue; fi
  fi
  step b16k_$v 300 python bench.py --no-cpu-baseline --positions 0 --no-configs --size 16384 --steps 4 --warmup 1
done
cp $OUT/lib_main.so paresis_amd/libparesis_hip.so
rm -f $OUT/lib_main.so
diff $OUT/hash_new.out $OUT/hash_prev.out > $OUT/hash_diff.txt && echo "HASHES IDENTICAL" | tee -a $OUT/progress.log || (echo "HASHES DIFFER" | tee -a $OUT/progress.log; cat $OUT/hash_diff.txt)
step b16k_keep48_again 300 python bench.py --no-cpu-baseline --positions 0 --no-configs --size 16384 --steps 4 --warmup 1
step cfg5 400 python bench.py --only-configs --configs 16384
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4s11/b16k*.out")):
    j=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    print(os.path.basename(f), j["ms_per_step"], j["kernel_ms_per_step"])
j=json.loads([l for l in open("gpurun_out/r4s11/cfg5.out").read().splitlines() if l.startswith("{")][-1])
e=j["configs"]["16384"]; print("cfg5", e["ms"], e["step_frac"], e["step_frac_per_propagation"], e["kernel_ms_per_step"], e["parity"])
PY
