#!/bin/bash
# Round 6, GPU session 23: two-round kernel, single-compiled round loop (the committed form) + round O's input parked by the unit's
# first round E and read back at round O's start + the next line fetched over the unit's last two rounds: parity, config 5, stamps.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s23
mkdir -p $OUT
timeout -k 10 300 python tools/diag_p2x.py 16384 36 2 > $OUT/diag.out 2>&1; grep "^rep" $OUT/diag.out | cut -c1-150
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py -x -q -k "power_of_two" > $OUT/t1.out 2>&1; rc=$?; echo "p2 tests rc $rc"; tail -3 $OUT/t1.out
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_large.py -x -q -k "16384 or partitioned" > $OUT/t2.out 2>&1; rc=$?; echo "large rc $rc"; tail -3 $OUT/t2.out
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py --only-configs --configs 16384 > $OUT/cfg5.out 2> $OUT/cfg5.err; echo "cfg5 rc $?"
python - <<PY
import json
d = json.loads(open("$OUT/cfg5.out").read().strip().splitlines()[-1])["configs"]["16384"]
print(d["ms"], d["step_frac"], d["kernel_ms_per_step"], d["parity"]["ok"])
PY
timeout -k 10 300 python tools/stamp_fresnel.py 16384 4 > $OUT/stamp_p2.out 2>&1; echo "stamp rc $?"; grep -A22 "workgroups:" $OUT/stamp_p2.out
PSX_SWITCHES="stamp_round=0" timeout -k 10 300 python tools/stamp_fresnel.py 16384 4 > $OUT/stamp_p2_r0.out 2>&1; grep -A22 "workgroups:" $OUT/stamp_p2_r0.out
PSX_SWITCHES="stamp_pass1=1" timeout -k 10 300 python tools/stamp_fresnel.py 16384 4 > $OUT/stamp_p1.out 2>&1; echo "stamp pass1 rc $?"; grep -A22 "workgroups:" $OUT/stamp_p1.out
