#!/bin/bash
# Round 4, GPU session 2: order-independent replay v2 (ownership by returning atomic), new accumulator default, IB A/B.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s2
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
step pytest_refr 600 python -m pytest tests -m gpu -x -q -k "deterministic or order_independent or refraction or chain_rt or darkfield"
tail -3 $OUT/pytest_refr.out
B="python bench.py --no-cpu-baseline --positions 0 --no-configs --steps 50"
step plain_step 200 $B
step det_step 200 $B --deterministic-step
step cfg5_det 400 python bench.py --only-configs --configs 16384 --deterministic-step
step batch_det 400 python bench.py --no-cpu-baseline --no-configs --steps 20
step batch_float 400 python bench.py --no-cpu-baseline --no-configs --steps 20 --float-atomics
cp paresis_amd/libparesis_hip.so $OUT/lib_main.so
for ib in 4 16; do
  cp tools/ab/libparesis_hip_ib$ib.so paresis_amd/libparesis_hip.so
  step ib${ib}_tests 900 python -m pytest tests/test_gpu_large.py tests/test_gpu_kernels.py -m gpu -x -q -k "fresnel or partitioned or engines or ragged or shared_forward"
  step ib${ib}_4096 200 $B
  step ib${ib}_16384 300 python bench.py --no-cpu-baseline --positions 0 --no-configs --size 16384 --steps 3 --warmup 1
  step ib${ib}_2048 200 python bench.py --no-cpu-baseline --positions 0 --no-configs --size 2048 --steps 100
done
cp $OUT/lib_main.so paresis_amd/libparesis_hip.so
step ib8_16384 300 python bench.py --no-cpu-baseline --positions 0 --no-configs --size 16384 --steps 3 --warmup 1
step ib8_2048 200 python bench.py --no-cpu-baseline --positions 0 --no-configs --size 2048 --steps 100
rm -f $OUT/lib_main.so
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4s2/*.out")):
    try:
        line=[l for l in open(f).read().splitlines() if l.startswith("{")][-1]
        j=json.loads(line)
    except Exception as e:
        print(os.path.basename(f), open(f).read()[-300:].replace("\n"," | "))
        continue
    k=j.get("kernel_ms_per_step",{})
    print(os.path.basename(f), j.get("ms_per_step"), {a:k[a] for a in k}, j.get("kernel_ms_short_launches",{}))
    for sim,e in (j.get("positions_batch") or {}).items():
        print("   batch",sim,e.get("ms_total"),(e.get("warm") or {}).get("ms_total"),e.get("check"),e.get("far_rays"))
    if "configs" in j:
        for n,e in j["configs"].items():
            print("   cfg",n,e.get("ms"),e.get("step_frac"),e.get("kernel_ms_per_step"),e.get("kernel_ms_short_launches"))
PY
