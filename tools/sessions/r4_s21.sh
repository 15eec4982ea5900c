#!/bin/bash
# Round 4, GPU session 21: phase reads of the deposit loop as four ds_read_b64 instead of two ds_read2_b64: A/B x 3 on one box,
# then the refraction tests on the new form.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s21
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
for round in 1 2 3; do
  for v in PSX_PHI_SINGLE0 PSX_PHI_SINGLE1; do
    cp tools/ab/libparesis_hip_$v.so paresis_amd/libparesis_hip.so
    step b_${v}_$round 300 python bench.py --no-cpu-baseline --positions 0 --no-configs
  done
done
cp tools/ab/libparesis_hip_PSX_PHI_SINGLE1.so paresis_amd/libparesis_hip.so
step tests 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -m gpu -q -x -k "refract or fastloop or chain or darkfield"
tail -3 $OUT/tests.out
cp $OUT/lib_main.so paresis_amd/libparesis_hip.so; rm -f $OUT/lib_main.so
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4s21/b_*.out")):
    j=json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    print(os.path.basename(f), j["ms_per_step"], j["kernel_ms_per_step"])
PY
