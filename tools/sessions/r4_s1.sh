#!/bin/bash
# Round 4, GPU session 1: tests, default bench, refraction accumulator A/B, order-independent replay cost.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s1
mkdir -p $OUT
step() {   # step NAME TIMEOUT CMD... ; stops the session when a step is killed or times out
  local name=$1 to=$2; shift 2
  echo "== $name" | tee -a $OUT/progress.log
  timeout -k 10 $to "$@" > $OUT/$name.out 2> $OUT/$name.err
  local rc=$?
  echo "$name rc=$rc" | tee -a $OUT/progress.log
  if [ $rc -ge 124 ]; then echo "ABORT after $name" | tee -a $OUT/progress.log; exit 1; fi
  return 0
}
step pytest 900 python -m pytest tests -m gpu -x -q
tail -5 $OUT/pytest.out
step bench_default 400 python bench.py
B="python bench.py --no-cpu-baseline --positions 0 --no-configs --steps 50"
cp paresis_amd/libparesis_hip.so $OUT/lib_main.so
for rep in 1 2; do
  for v in p58_m0 p58_m1 p64_m1 p64_m2; do
    cp tools/ab/libparesis_hip_$v.so paresis_amd/libparesis_hip.so
    step ab_${v}_$rep 200 $B
  done
done
for v in p58_m0 p58_m1 p64_m1 p64_m2; do
  cp tools/ab/libparesis_hip_$v.so paresis_amd/libparesis_hip.so
  step ab16k_${v} 300 python bench.py --no-cpu-baseline --positions 0 --no-configs --size 16384 --steps 3 --warmup 1
done
cp $OUT/lib_main.so paresis_amd/libparesis_hip.so
step det_step 200 $B --deterministic-step
step plain_step 200 $B
step cfg5_plain 400 python bench.py --only-configs --configs 16384
step cfg5_det 400 python bench.py --only-configs --configs 16384 --deterministic-step
rm -f $OUT/lib_main.so
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4s1/*.out")):
    try:
        line=[l for l in open(f).read().splitlines() if l.startswith("{")][-1]
        j=json.loads(line)
    except Exception as e:
        continue
    k=j.get("kernel_ms_per_step",{})
    print(os.path.basename(f), j.get("ms_per_step"), {a:k[a] for a in k}, j.get("kernel_ms_short_launches",{}))
    if "configs" in j:
        for n,e in j["configs"].items():
            print("   cfg",n,e.get("ms"),e.get("step_frac"),e.get("kernel_ms_per_step"),e.get("parity"))
PY
