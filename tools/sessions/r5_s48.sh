#!/bin/bash
# Round 5, GPU session 48: the copyBuffer dispatches of the Fresnel position loop: neighbours in the kernel trace and the HIP call
# with the same correlation id.
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5s48
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --hip-trace --kernel-trace --output-format csv -d $OUT/tr -o t -- python3 $ROOT/tools/time_positions.py 4096 12 --sim Fresnel > $OUT/trace.log 2>&1 || { echo "rocprof failed"; tail -3 $OUT/trace.log; exit 1; }
python3 - <<PY
import csv, glob
k = list(csv.DictReader(open(glob.glob("$OUT/tr/**/*kernel_trace.csv", recursive=True)[0])))
a = list(csv.DictReader(open(glob.glob("$OUT/tr/**/*hip_api_trace.csv", recursive=True)[0])))
print("kernel columns", list(k[0].keys()))
byc = {}
for r in a: byc.setdefault(r["Correlation_Id"], []).append(r["Function"])
k.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(k) if "copyBuffer" in r["Kernel_Name"]]
print(len(idx), "copyBuffer dispatches of", len(k))
for i in idx[-6:]:
    for j in range(max(0, i - 2), min(len(k), i + 3)):
        r = k[j]
        print("  " if j != i else "->", r["Kernel_Name"][:60], "corr", r["Correlation_Id"], byc.get(r["Correlation_Id"]), "grid", r.get("Grid_Size_X", r.get("Grid_Size")), "dur", int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print()
PY
rm -rf $OUT/tr
