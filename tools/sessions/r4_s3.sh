#!/bin/bash
# Round 4, GPU session 3: IB A/B (after the stride fix), dark-field timing, full test pass, default bench.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s3
mkdir -p $OUT
step() {
  local name=$1 to=$2; shift 2
  echo "== $name" | tee -a $OUT/progress.log
  timeout -k 10 $to "$@" > $OUT/$name.out 2> $OUT/$name.err
  local rc=$?
  echo "$name rc=$rc" | tee -a $OUT/progress.log
  if [ $rc -ge 124 ]; then echo "ABORT after $name" | tee -a $OUT/progress.log; exit 1; fi
  return 0
}
step pytest 900 python -m pytest tests -m gpu -x -q
tail -3 $OUT/pytest.out
step darkfield 200 python tools/time_darkfield.py 20
step bench_default 500 python bench.py
B="python bench.py --no-cpu-baseline --positions 0 --no-configs --steps 50"
cp paresis_amd/libparesis_hip.so $OUT/lib_main.so
for ib in 16 4; do
  cp tools/ab/libparesis_hip_ib$ib.so paresis_amd/libparesis_hip.so
  step ib${ib}_tests 900 python -m pytest tests/test_gpu_large.py tests/test_gpu_kernels.py -m gpu -x -q -k "fresnel or partitioned or engines or ragged or shared_forward"
  if ! grep -q passed $OUT/ib${ib}_tests.out || grep -q failed $OUT/ib${ib}_tests.out; then echo "ib$ib tests not clean: skipping its timings" | tee -a $OUT/progress.log; continue; fi
  step ib${ib}_4096 200 $B
  step ib${ib}_16384 300 python bench.py --no-cpu-baseline --positions 0 --no-configs --size 16384 --steps 3 --warmup 1
  step ib${ib}_2048 200 python bench.py --no-cpu-baseline --positions 0 --no-configs --size 2048 --steps 100
done
cp $OUT/lib_main.so paresis_amd/libparesis_hip.so
step ib8_4096 200 $B
step ib8_16384 300 python bench.py --no-cpu-baseline --positions 0 --no-configs --size 16384 --steps 3 --warmup 1
step ib8_2048 200 python bench.py --no-cpu-baseline --positions 0 --no-configs --size 2048 --steps 100
rm -f $OUT/lib_main.so
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4s3/*.out")):
    try:
        line=[l for l in open(f).read().splitlines() if l.startswith("{")][-1]
        j=json.loads(line)
    except Exception as e:
        print(os.path.basename(f), open(f).read()[-400:].replace("\n"," | "))
        continue
    k=j.get("kernel_ms_per_step",{})
    print(os.path.basename(f), j.get("ms_per_step"), j.get("value"), j.get("value_cold"), {a:k[a] for a in k}, j.get("kernel_ms_short_launches",{}))
    for sim,e in (j.get("positions_batch") or {}).items():
        print("   batch",sim,e.get("ms_total"),(e.get("warm") or {}).get("ms_total"),e.get("check"),e.get("far_rays"))
    if "configs" in j:
        for n,e in j["configs"].items():
            print("   cfg",n,e.get("ms"),e.get("step_frac"),e.get("step_frac_per_propagation"),e.get("kernel_ms_per_step"),e.get("kernel_ms_short_launches"))
    if "roofline" in j:
        r=j["roofline"]; print("   roofline", {k:r[k] for k in ("kernel","frac","frac_per_propagation","step_frac","step_frac_per_propagation","fresnel_call_frac") if k in r})
PY
