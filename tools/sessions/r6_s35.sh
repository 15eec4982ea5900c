#!/bin/bash
# Round 6, GPU session 35: the intermediate blocked by pass 1's lines for 16384^2-class grids (both passes on the two-round kernel) +
# the parked round-O input (step 4): parity (XL shapes, the 16384^2 tests), then config 5 timed: layout x step.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s35
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_large.py -x -q -m gpu -k "power_of_two or 16384 or partitioned" > $OUT/tests.out 2>&1; rc=$?; echo "tests rc $rc"; tail -3 $OUT/tests.out
[ $rc -ne 0 ] && exit $rc
cp paresis_amd/libparesis_hip.so $OUT/.lib_saved.so
run() {   # tag, library, extra flags
  cp $2 paresis_amd/libparesis_hip.so
  timeout -k 10 300 python bench.py --only-configs --configs 16384 --no-config-parity $3 > $OUT/cfg5_$1.out 2>$OUT/cfg5_$1.err; echo "$1 rc $?"
}
run s4_byl $OUT/.lib_saved.so ""
run s4_old $OUT/.lib_saved.so "--debug-switch no_xl_layout=1"
run s3_byl tools/ab/libparesis_hip_s3.so ""
run s0_byl tools/ab/libparesis_hip_s0.so ""
run s0_old tools/ab/libparesis_hip_s0.so "--debug-switch no_xl_layout=1"
run s4_byl2 $OUT/.lib_saved.so ""
cp $OUT/.lib_saved.so paresis_amd/libparesis_hip.so
python - <<PY
import json
for t in ("s4_byl", "s4_old", "s3_byl", "s0_byl", "s0_old", "s4_byl2"):
    try:
        d = json.loads(open("$OUT/cfg5_%s.out" % t).read().strip().splitlines()[-1])
        c = d["configs"]["16384"]
        k = c.get("kernel_ms_per_step", {})
        print("%-8s step %.3f  pass2 %.3f pass1 %.3f near %.3f" % (t, c["ms"], k.get("k_fresnel_rows", 0), k.get("k_fresnel_cols", 0), k.get("k_refract_near", 0)))
    except Exception as e:
        print(t, "no line", e)
PY
