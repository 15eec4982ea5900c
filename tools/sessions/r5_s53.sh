#!/bin/bash
# Round 5, GPU session 53: v_exp_f32 itself in the refraction tile kernel's staging (exp2f() wraps it in five more instructions):
# refraction tests, then the halo sweep (4 distances) and the RT position loop.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s53
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_large.py -m gpu -x -q -p no:cacheprovider -k "refract or fastloop or chain or replay" > $OUT/tests.out 2>&1; rc=$?; echo "tests rc $rc"; tail -2 $OUT/tests.out
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/halo_sweep.py 4096 2 > $OUT/sweep.out 2>&1; tail -3 $OUT/sweep.out
for r in 1 2; do timeout -k 10 300 python tools/time_positions.py 4096 32 --sim RT > $OUT/pos$r.out 2>&1; grep -E "positions of|library kernels" $OUT/pos$r.out; done
