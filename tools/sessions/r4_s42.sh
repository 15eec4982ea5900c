#!/bin/bash
# Round 4, GPU session 42: first fetch of a line launch requested before the twiddle-table copy: hashes, small-grid timing A/B.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s42
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
for r in 1 2 3; do
  cp tools/ab/libparesis_hip_prev.so paresis_amd/libparesis_hip.so
  step c_prev_$r 300 python bench.py --only-configs --configs 512,2048 --no-config-parity
  cp $OUT/lib_new.so paresis_amd/libparesis_hip.so
  step c_new_$r 300 python bench.py --only-configs --configs 512,2048 --no-config-parity
done
cp $OUT/lib_new.so paresis_amd/libparesis_hip.so; rm -f $OUT/lib_new.so
step bench 300 python bench.py --no-cpu-baseline --positions 0 --no-configs
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4s42/c_*.out")):
    j=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    print(os.path.basename(f), {k: (v["ms"], v["step_frac"]) for k, v in j["configs"].items()})
j=json.loads([l for l in open("gpurun_out/r4s42/bench.out").read().splitlines() if l.startswith("{")][-1])
print("4096", j["ms_per_step"], j["kernel_ms_per_step"])
PY
