#!/bin/bash
# Round 5, GPU session 49: one Fresnel position of the loop, dispatch by dispatch (the medians of trace_gaps mix the one- and
# two-distance launches of the same line kernel).
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5s49
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- python3 $ROOT/tools/time_positions.py 4096 24 --sim Fresnel > $OUT/trace.log 2>&1 || { echo "rocprof failed"; tail -3 $OUT/trace.log; exit 1; }
python3 $ROOT/tools/trace_gaps.py $(ls $OUT/tr/*kernel_trace.csv $OUT/tr/*/*kernel_trace.csv 2>/dev/null | head -1) --dump 26 | tee $OUT/dump.txt
rm -rf $OUT/tr
