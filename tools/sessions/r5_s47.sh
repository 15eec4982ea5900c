#!/bin/bash
# Round 5, GPU session 47: which HIP call is the copyBuffer (+20 us idle) that every second Fresnel position shows in the kernel trace.
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5s47
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tr -o t -- python3 $ROOT/tools/time_positions.py 4096 12 --sim Fresnel > $OUT/trace.log 2>&1 || { echo "rocprof failed"; tail -3 $OUT/trace.log; exit 1; }
ls $OUT/tr $OUT/tr/* | head -20
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/tr/**/*hip_api_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
c = collections.Counter(r["Function"] for r in rows)
for k, v in c.most_common(40): print(v, k)
m = glob.glob("$OUT/tr/**/*memory_copy_trace.csv", recursive=True)
if m:
    rows = list(csv.DictReader(open(m[0])))
    print(len(rows), "copies; columns", list(rows[0].keys()) if rows else None)
    for r in rows[-12:]: print(r)
PY
# keep only the small files
find $OUT/tr -name "*kernel_trace.csv" -delete
