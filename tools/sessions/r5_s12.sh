#!/bin/bash
# Round 5, GPU session 12: membrane plan with 16-pixel cells (tests, fuzz, kernel time); configs 512 / 2048 as captured hipGraphs.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s12
mkdir -p $OUT
PSX_FUZZ=20 timeout -k 10 600 python -m pytest tests/test_gpu_main.py tests/test_gpu_fuzz.py -m gpu -x -q -p no:cacheprovider -k "membrane" > $OUT/tests.out 2>&1; rc=$?; tail -3 $OUT/tests.out
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/time_membrane.py > $OUT/membrane.out 2>&1; grep "^N=" $OUT/membrane.out
timeout -k 10 600 python bench.py --only-configs --configs 512,2048 > $OUT/configs.out 2> $OUT/configs.err; echo "rc $?"; python - <<PY
import json
d = json.loads(open("$OUT/configs.out").read().strip().splitlines()[-1])
for k, v in d["configs"].items():
    print(k, v["ms"], v["ms_plain_launches"], v["launch"], v["step_frac"], v["kernel_ms_per_step"], v.get("kernel_ms_short_launches"), v["parity"].get("ok"))
PY
