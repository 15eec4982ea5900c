#!/bin/bash
# Round 5, GPU session 1: the two-pass order-independent replay (unit per call, PREP folded into the append, fold tables) --
# whole GPU suite, kernel times in both modes at 4096^2 (ov 2) and 16384^2 (ov 4), convergence statistics of the far shares,
# the default bench line.
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5s1
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/tests.out 2>&1; rc=$?; tail -5 $OUT/tests.out
[ $rc -eq 0 ] || exit $rc
for mode in float reproducible; do
  timeout -k 10 200 python tools/halo_sweep.py 4096 2 $mode > $OUT/halo_4096_$mode.out 2>&1 && cat $OUT/halo_4096_$mode.out | tail -3
  timeout -k 10 300 python tools/halo_sweep.py 16384 4 $mode > $OUT/halo_16384_$mode.out 2>&1 && cat $OUT/halo_16384_$mode.out | tail -3
done
timeout -k 10 300 python tools/far_convergence.py 16384 4 8 > $OUT/conv_16384.out 2>&1; tail -60 $OUT/conv_16384.out
timeout -k 10 200 python tools/far_convergence.py 4096 2 4 > $OUT/conv_4096.out 2>&1; tail -20 $OUT/conv_4096.out
timeout -k 10 600 python bench.py > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc $?"; python - <<PY
import json
d = json.loads(open("$OUT/bench.out").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "value_cold", "far_rays", "other_far_ray_mode")})
print(d["roofline"]["frac"], d.get("kernel_ms_per_step"))
for k, v in d["positions_batch"].items():
    print(k, v["ms_total"], v.get("warm", {}).get("ms_total"), v["check"])
for k, v in d["configs"].items():
    print(k, v["ms"], v["step_frac"], v["step_frac_per_propagation"], v["refraction_halo"], v["kernel_ms_per_step"], v["parity"])
PY
