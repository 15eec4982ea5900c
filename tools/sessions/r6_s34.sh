#!/bin/bash
# Round 6, GPU session 34: the parked round-O input with its stores addressed through the vector offset only (the store-data
# hazard of r6s33): the compare build, then steps 2-4 against the XL shapes, then config 5 timed with step 4.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s34
mkdir -p $OUT
cp paresis_amd/libparesis_hip.so $OUT/.lib_saved.so
cp tools/ab/libparesis_hip_x1cmp.so paresis_amd/libparesis_hip.so
timeout -k 10 120 python tools/diag_p2x_cmp.py 36 16384 1 > $OUT/cmp_b.out 2>&1; grep -v "^wg" $OUT/cmp_b.out | head -8
for t in x2 x3 x4; do
  cp tools/ab/libparesis_hip_$t.so paresis_amd/libparesis_hip.so
  timeout -k 10 240 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "power_of_two and (shape11 or shape12 or shape13 or shape14 or shape15)" > $OUT/tests_$t.out 2>&1; echo "$t tests rc $?"; tail -2 $OUT/tests_$t.out
done
cp $OUT/.lib_saved.so paresis_amd/libparesis_hip.so
timeout -k 10 300 python bench.py --only-configs --configs 16384 --no-config-parity > $OUT/cfg5_base.out 2>$OUT/cfg5_base.err; echo "base rc $?"
cp tools/ab/libparesis_hip_x4.so paresis_amd/libparesis_hip.so
timeout -k 10 300 python bench.py --only-configs --configs 16384 --no-config-parity > $OUT/cfg5_x4.out 2>$OUT/cfg5_x4.err; echo "x4 rc $?"
cp $OUT/.lib_saved.so paresis_amd/libparesis_hip.so
python - <<PY
import json
for t in ("base", "x4"):
    try:
        d = json.loads(open("$OUT/cfg5_%s.out" % t).read().strip().splitlines()[-1])
        c = d["configs"]["16384"]
        print(t, c["ms"], c.get("kernel_ms_per_step"))
    except Exception as e:
        print(t, "no line", e)
PY
