#!/bin/bash
# Round 5, GPU session 26: 4-rank gloo rehearsal again (the float-atomics batch check now allows a flipped Poisson draw its few sqrt(counts)).
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s26
mkdir -p $OUT
timeout -k 10 500 python bench.py --gpus 4 --backend gloo --positions 16 --no-configs --no-cpu-baseline --steps 5 --warmup 2 > $OUT/n4.out 2> $OUT/n4.err; echo "n=4 rc $?"
python - <<PY
import json
txt = open("$OUT/n4.out").read()
d = json.loads([l for l in txt.splitlines() if l.startswith('{"metric"')][0])
for k, v in d.get("positions_batch", {}).items():
    print(k, v["ms_total"], v["check"])
PY
