#!/bin/bash
# Round 5, GPU session 11: lazily widened results on the sink, reserved output blocks, busy-wait barrier in the emulation:
# multi-rank GPU tests, then the bench line's rank_share objects.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s11
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_main.py -m gpu -x -q -p no:cacheprovider > $OUT/tests.out 2>&1; rc=$?; tail -3 $OUT/tests.out
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python bench.py --no-configs --no-cpu-baseline > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc $?"; tail -3 $OUT/bench.err; python - <<PY
import json
d = json.loads(open("$OUT/bench.out").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "far_rays", "other_far_ray_mode")})
for k, v in d["positions_batch"].items():
    print(k, v["ms_total"], v.get("warm", {}).get("ms_total"), v["check"]["bit_equal"], json.dumps(v.get("rank_share")))
PY
