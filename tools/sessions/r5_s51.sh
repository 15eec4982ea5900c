#!/bin/bash
# Round 5, GPU session 51: k_band_pair instantiated for 4 and 12 taps; the configs' detections through detect_many: detector tests,
# kernel trace of the RT position loop, config 5 and 512 of the bench.
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5s51
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -m gpu -x -q -p no:cacheprovider -k "detector or chain" > $OUT/tests.out 2>&1; rc=$?; echo "tests rc $rc"; tail -2 $OUT/tests.out
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --only-configs --configs 512,16384 > $OUT/configs.out 2> $OUT/configs.err; echo "configs rc $?"
python3 - <<PY
import json
d = json.loads(open("$OUT/configs.out").read().strip().splitlines()[-1])
for k, v in d["configs"].items(): print(k, v.get("ms"), v.get("step_frac"), v.get("step_frac_per_propagation"), v.get("parity"))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- python3 $ROOT/tools/time_positions.py 4096 48 --sim RT > $OUT/trace.log 2>&1 || { echo "rocprof failed"; tail -3 $OUT/trace.log; exit 1; }
python3 $ROOT/tools/trace_gaps.py $(ls $OUT/tr/*kernel_trace.csv $OUT/tr/*/*kernel_trace.csv 2>/dev/null | head -1) | tee $OUT/gapsRT.txt
rm -rf $OUT/tr
