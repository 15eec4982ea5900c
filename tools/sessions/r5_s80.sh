#!/bin/bash
# Round 5, GPU session 80 (final build of the round): what the driver runs at round end, in its order: build check, smoke(), the GPU suite, the default bench.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s80
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $OUT/smoke.out 2>&1; echo "smoke rc $?"; tail -2 $OUT/smoke.out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/tests.out 2>&1; echo "tests rc $?"; tail -2 $OUT/tests.out
t0=$(date +%s); timeout -k 10 1000 python bench.py > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc $? in $(( $(date +%s) - t0 )) s"
