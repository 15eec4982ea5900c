#!/bin/bash
# Round 6, GPU session 8: the headline step with the far rays summed either way and halos 4 / 6 / 8, all on ONE box (boxes differ
# by up to 10 %): what does the Experiment's default mode (order-independent replay, caller's scale) cost the step today?
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s8
mkdir -p $OUT
run() {
  tag=$1; shift
  timeout -k 10 200 python bench.py --no-configs --no-cpu-baseline --positions 0 --steps 30 "$@" > $OUT/$tag.out 2> $OUT/$tag.err
  python - <<PY
import json
d = json.loads(open("$OUT/$tag.out").read().strip().splitlines()[-1])
k = d.get("kernel_ms_per_step", {})
print("%-18s ms/step %.4f  steady %.4f  other mode %s  near %.4f far %s" % ("$tag", d["ms_per_step"], d["steady"]["ms_per_step"] if isinstance(d.get("steady"), dict) else -1,
      d.get("other_far_ray_mode", {}).get("ms_per_step"), k.get("k_refract_near", 0), {n: v for n, v in list(k.items()) + list(d.get("kernel_ms_short_launches", {}).items()) if "far" in n}))
PY
}
run float_h4 --halo 4
run det_h4 --halo 4 --deterministic-step
run det_h6 --halo 6 --deterministic-step
run det_h8 --halo 8 --deterministic-step
run float_h6 --halo 6
run det_h4_noscale --halo 4 --deterministic-step --no-replay-scale
run float_h4_again --halo 4
