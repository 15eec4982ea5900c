#!/bin/bash
# Round 4, GPU session 18: does buffer placement decide the two modes of pass 1 (0.388 / 0.405 ms)?  tools/placement_probe.py
# with the main library and with the variant that showed the slow mode.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s18
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
step probe_main 400 python tools/placement_probe.py
cat $OUT/probe_main.out
cp tools/ab/libparesis_hip_r1_e0.so paresis_amd/libparesis_hip.so
step probe_r1e0 400 python tools/placement_probe.py
cat $OUT/probe_r1e0.out
cp $OUT/lib_main.so paresis_amd/libparesis_hip.so; rm -f $OUT/lib_main.so
