#!/bin/bash
# Round 6, GPU session 48: the shader-clock probe next to the step time (three short bench runs on this box).
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s48
mkdir -p $OUT
for i in 1 2; do
  timeout -k 10 200 python bench.py --no-configs --no-cpu-baseline --positions 0 > $OUT/b$i.out 2> $OUT/b$i.err; echo "rc $?"
  python - <<PY
import json
d = json.loads(open("$OUT/b$i.out").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "steady", d["steady"]["ms_per_step"], "shader MHz", d["shader_clock_mhz"], "ms x GHz", round(d["steady"]["ms_per_step"] * d["shader_clock_mhz"] / 1000, 4))
PY
done
