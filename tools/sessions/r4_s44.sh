#!/bin/bash
# Round 4, GPU session 44: counters of k_membrane_layers (what bounds the synthesis of a position's membrane?) + membrane fuzz case 323.
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r4s44
mkdir -p $OUT
PSX_FUZZ=100 timeout -k 10 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k "membrane" -p no:cacheprovider > $OUT/fuzz_membrane.out 2>&1; tail -2 $OUT/fuzz_membrane.out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT \
    --output-format csv -d $OUT/pmc -o runc -- python3 $ROOT/tools/time_membrane.py > $OUT/pmc.log 2>&1 || { echo "rocprof failed"; tail -5 $OUT/pmc.log; exit 1; }
python3 - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(sorted(glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True))[-1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "k_membrane" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"].split("(")[0][-30:], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: "%.3g" % (sum(v) / len(v)) for c, v in d.items()})
PY
tail -4 $OUT/pmc.log
