#!/bin/bash
# Round 5, GPU session 33: final build -- GPU suite, default bench, and config 5's rocprofv3 set again (its tile kernel changed in s32).
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=$PWD
OUT=$PWD/gpurun_out/r5s33
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/tests.out 2>&1; echo "tests rc $?"; tail -2 $OUT/tests.out
t0=$(date +%s); timeout -k 10 1000 python bench.py > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc $? in $(( $(date +%s) - t0 )) s"
bash tools/collect_profiles.sh _cfg5 --only-configs --configs 16384 --no-config-parity > $OUT/prof.log 2>&1 && echo "cfg5 profiles done"
