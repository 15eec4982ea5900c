#!/bin/bash
# Round 6, GPU session 50: final build -- the whole GPU suite, smoke(), the default bench; then the rocprofv3 collections for profiles/.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s50
mkdir -p $OUT
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $OUT/tests.out 2>&1; rc=$?; echo "gpu suite rc $rc"; tail -4 $OUT/tests.out
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python __graft_entry__.py smoke > $OUT/smoke.out 2>&1; echo "smoke rc $?"; tail -2 $OUT/smoke.out
timeout -k 10 500 python bench.py > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc $?"
bash tools/collect_profiles.sh && bash tools/collect_profiles.sh _cfg5 --only-configs --configs 16384
