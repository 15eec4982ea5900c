#!/bin/bash
# Round 5, GPU session 46: kernel trace of the position loops (4096^2) after psx_detect_multi_f32 / the in-place reference image.
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5s46
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for sim in RT Fresnel; do
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace$sim -o t -- python3 $ROOT/tools/time_positions.py 4096 48 --sim $sim > $OUT/trace$sim.log 2>&1 || { echo "rocprof failed"; tail -3 $OUT/trace$sim.log; exit 1; }
  f=$(ls $OUT/trace$sim/*kernel_trace.csv $OUT/trace$sim/*/*kernel_trace.csv 2>/dev/null | head -1)
  echo "== $sim"; python3 $ROOT/tools/trace_gaps.py $f | tee $OUT/gaps$sim.txt
  rm -rf $OUT/trace$sim
done
