#!/bin/bash
# Round 6, GPU session 44: the default bench and the rocprofv3 collections once more (r6s43 landed on one of the pool's slow boxes).
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s44
mkdir -p $OUT
timeout -k 10 500 python bench.py > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc $?"
python - <<PY
import json
d = json.loads(open("$OUT/bench.out").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], d["kernel_ms_per_step"])
PY
bash tools/collect_profiles.sh && bash tools/collect_profiles.sh _cfg5 --only-configs --configs 16384
