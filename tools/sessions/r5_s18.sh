#!/bin/bash
# Round 5, GPU session 19 (split mode in its own instantiations): psx_refract_split_f32 (one staging for both halves of the dark-field split): refraction + dark-field
# tests, the headline step (did the split's uniform branch cost the plain call anything?), dark-field position timings.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s18
mkdir -p $OUT
PSX_FUZZ=3 timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_experiment.py tests/test_host_cpu.py -x -q -p no:cacheprovider -k "darkfield or refract or chain or spill or symbol" > $OUT/tests.out 2>&1; rc=$?; tail -3 $OUT/tests.out
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --no-configs --positions 0 --no-cpu-baseline > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc $?"; python - <<PY
import json
d = json.loads(open("$OUT/bench.out").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "value_cold", "far_rays", "other_far_ray_mode")}, d.get("kernel_ms_per_step"))
PY
timeout -k 10 300 python tools/time_positions.py 4096 6 --sim RT --poly 25 > $OUT/plain.out 2>&1; grep -v "per position (host" $OUT/plain.out | tail -2
timeout -k 10 300 python tools/time_positions.py 4096 6 --sim RT --poly 25 --scatter --thin 200 > $OUT/thin.out 2>&1; grep -v "per position (host" $OUT/thin.out | tail -2
timeout -k 10 300 python tools/time_positions.py 4096 16 --sim RT > $OUT/plain_mono.out 2>&1; grep -v "per position (host" $OUT/plain_mono.out | tail -2
timeout -k 10 300 python tools/time_positions.py 4096 16 --sim RT --scatter --thin 30 > $OUT/thin_mono.out 2>&1; grep -v "per position (host" $OUT/thin_mono.out | tail -2
