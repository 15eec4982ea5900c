#!/bin/bash
# Round 4, GPU session 28: detector pair kernel with aligned 16-byte tap reads: tests, then timing against the previous library.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s28
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
step tests 600 python -m pytest tests -m gpu -q -x -k "detector or fuzz_chain or fuzz_detector or chain or main"
tail -3 $OUT/tests.out
step det16k_new 300 python tools/time_detector_16384.py
step det_new 300 python tools/time_detector.py
cp paresis_amd/libparesis_hip.so $OUT/lib_main.so
cp tools/ab/libparesis_hip_detbase.so paresis_amd/libparesis_hip.so
step det16k_old 300 python tools/time_detector_16384.py
step det_old 300 python tools/time_detector.py
cp $OUT/lib_main.so paresis_amd/libparesis_hip.so; rm -f $OUT/lib_main.so
for f in det16k_old det16k_new det_old det_new; do echo "--- $f"; cat $OUT/$f.out; done
