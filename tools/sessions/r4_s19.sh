#!/bin/bash
# Round 4, GPU session 19: the two modes of pass 1 in bench.py (0.388 / 0.405 ms by library variant): does a dummy allocation made
# before the library loads move a build from one mode to the other?
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s19
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
for v in r0_e0 r1_e0; do
  cp tools/ab/libparesis_hip_$v.so paresis_amd/libparesis_hip.so
  for pad in 0 64 4096 2048000; do
    step p_${v}_$pad 300 python tools/pad_bench.py $pad --no-cpu-baseline --positions 0 --no-configs
  done
done
cp $OUT/lib_main.so paresis_amd/libparesis_hip.so; rm -f $OUT/lib_main.so
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4s19/p_*.out")):
    j=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    print(os.path.basename(f), j["ms_per_step"], j["kernel_ms_per_step"])
PY
