#!/bin/bash
# Round 5, GPU session 4: bench.py --emulate-world 8 (rank 0's and rank 7's share of the 64-position batch, each in a fresh
# process on one GPU) + the changed tests.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s4
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_experiment.py tests/test_gpu_main.py -m gpu -x -q -p no:cacheprovider > $OUT/tests.out 2>&1; rc=$?; tail -3 $OUT/tests.out
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python bench.py --no-configs --no-cpu-baseline > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc $?"; tail -3 $OUT/bench.err; python - <<PY
import json
d = json.loads(open("$OUT/bench.out").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "far_rays", "other_far_ray_mode")})
for k, v in d["positions_batch"].items():
    print(k, v["ms_total"], v.get("warm", {}).get("ms_total"), v["check"]["bit_equal"], json.dumps(v.get("rank_share"), indent=1))
PY
