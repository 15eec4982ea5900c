#!/bin/bash
# Round 4, GPU session 8: the line kernel split in two: identical images (hashes against the previous build), tests, bench.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s8
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
cp tools/ab/libparesis_hip_prev.so paresis_amd/libparesis_hip.so
step hash_prev 400 python tools/hash_fresnel.py
step bench_prev 200 python bench.py --no-cpu-baseline --positions 0 --no-configs --steps 50
step bench16k_prev 300 python bench.py --no-cpu-baseline --positions 0 --no-configs --size 16384 --steps 3 --warmup 1
cp $OUT/lib_main.so paresis_amd/libparesis_hip.so
rm -f $OUT/lib_main.so
diff $OUT/hash_new.out $OUT/hash_prev.out > $OUT/hash_diff.txt && echo "HASHES IDENTICAL" | tee -a $OUT/progress.log || (echo "HASHES DIFFER" | tee -a $OUT/progress.log; cat $OUT/hash_diff.txt)
step bench_new 200 python bench.py --no-cpu-baseline --positions 0 --no-configs --steps 50
step bench16k_new 300 python bench.py --no-cpu-baseline --positions 0 --no-configs --size 16384 --steps 3 --warmup 1
step bench2k_new 200 python bench.py --no-cpu-baseline --positions 0 --no-configs --size 2048 --steps 100
step pytest 900 python -m pytest tests -m gpu -q -x
tail -4 $OUT/pytest.out
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4s8/bench*.out")):
    j=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    print(os.path.basename(f), j["ms_per_step"], j["kernel_ms_per_step"])
PY
cat $OUT/hash_new.out
