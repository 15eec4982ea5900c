#!/bin/bash
# Round 4, GPU session 37: the fuzz file at 40x its committed size (about 3000 cases), once.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s37
mkdir -p $OUT
PSX_FUZZ=40 timeout -k 10 1100 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x --maxfail=20 > $OUT/fuzz40.out 2> $OUT/fuzz40.err
echo "rc=$?" | tee $OUT/progress.log
grep -E "^FAILED|passed|failed" $OUT/fuzz40.out | head -40
