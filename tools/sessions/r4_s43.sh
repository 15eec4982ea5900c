#!/bin/bash
# Round 4, GPU session 43: the fuzz file at 100x its committed size (8800 cases), once, on the final build.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s43
mkdir -p $OUT
PSX_FUZZ=100 timeout -k 10 1150 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --maxfail=20 -p no:cacheprovider > $OUT/fuzz100.out 2> $OUT/fuzz100.err
echo "rc=$?" | tee $OUT/progress.log
grep -E "^FAILED|passed|failed" $OUT/fuzz100.out | head -40
