#!/bin/bash
# Round 5, GPU session 2: exact far-ray listing criterion (a ray is listed only if the replay will deposit one of its shares):
# refraction tests + fuzz, kernel times in both modes, the step both ways.
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5s2
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_large.py -m gpu -x -q -p no:cacheprovider -k "refract or deterministic or order or fastloop or chain" > $OUT/tests.out 2>&1; rc=$?; tail -3 $OUT/tests.out
[ $rc -eq 0 ] || exit $rc
for mode in float reproducible; do
  timeout -k 10 200 python tools/halo_sweep.py 4096 2 $mode > $OUT/halo_4096_$mode.out 2>&1 && grep "^N " $OUT/halo_4096_$mode.out
  timeout -k 10 300 python tools/halo_sweep.py 16384 4 $mode > $OUT/halo_16384_$mode.out 2>&1 && grep "^N " $OUT/halo_16384_$mode.out
done
timeout -k 10 600 python bench.py --no-configs --positions 0 --no-cpu-baseline > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc $?"; python - <<PY
import json
d = json.loads(open("$OUT/bench.out").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "value_cold", "far_rays", "other_far_ray_mode", "steady")})
print(d["roofline"]["frac"], d.get("kernel_ms_per_step"))
PY
