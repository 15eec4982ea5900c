#!/bin/bash
# Round 5, GPU session 35: kernel trace of the small-grid config steps (512^2, 2048^2): durations and the idle time between kernels.
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5s35
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 512 2048; do
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace$n -o t -- python3 $ROOT/bench.py --only-configs --configs $n --no-config-parity --no-config-graph > $OUT/trace$n.log 2>&1 || { echo "rocprof failed"; tail -3 $OUT/trace$n.log; exit 1; }
  f=$(ls $OUT/trace$n/*kernel_trace.csv $OUT/trace$n/*/*kernel_trace.csv 2>/dev/null | head -1)
  echo "== $n ($f)"; python3 $ROOT/tools/trace_gaps.py $f | tee $OUT/gaps$n.txt
  rm -rf $OUT/trace$n      # the raw trace is large; the summary stays
done
