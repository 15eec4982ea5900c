#!/bin/bash
# Round 4, GPU session 9: DIF instance A/B: round-O twiddle requested early, more window positions in flight during the transform.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s9
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
for v in main w4 nha56 nha60 main2; do
  case $v in main|main2) cp $OUT/lib_main.so paresis_amd/libparesis_hip.so;; *) cp tools/ab/libparesis_hip_$v.so paresis_amd/libparesis_hip.so;; esac
  if [ $v != main ] && [ $v != main2 ]; then
    step ${v}_tests 600 python -m pytest tests/test_gpu_large.py -m gpu -x -q -k "partitioned"
    if ! grep -q passed $OUT/${v}_tests.out || grep -q failed $OUT/${v}_tests.out; then echo "$v tests not clean" | tee -a $OUT/progress.log; continue; fi
  fi
  step ${v}_16384 300 python bench.py --no-cpu-baseline --positions 0 --no-configs --size 16384 --steps 4 --warmup 1
done
cp $OUT/lib_main.so paresis_amd/libparesis_hip.so
rm -f $OUT/lib_main.so
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4s9/*_16384.out")):
    j=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    print(os.path.basename(f), j["ms_per_step"], j["kernel_ms_per_step"])
PY
grep -h passed $OUT/*_tests.out
