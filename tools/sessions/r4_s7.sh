#!/bin/bash
# Round 4, GPU session 7: lanes per far-ray list A/B (4096^2 step, config 5, positions batch), refraction tests on the default.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s7
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
step pytest_refr 600 python -m pytest tests -m gpu -x -q -k "refraction or deterministic or order_independent or chain_rt or fastloop or reproducible"
tail -3 $OUT/pytest_refr.out
B="python bench.py --no-cpu-baseline --positions 0 --no-configs --steps 50"
cp paresis_amd/libparesis_hip.so $OUT/lib_main.so
for v in 64 32 16 8 4; do
  if [ $v = 16 ]; then cp $OUT/lib_main.so paresis_amd/libparesis_hip.so; else cp tools/ab/libparesis_hip_sub$v.so paresis_amd/libparesis_hip.so; fi
  step sub${v}_4096 200 $B
  step sub${v}_4096det 200 $B --deterministic-step
  step sub${v}_halo16k 300 python tools/halo_sweep.py 16384 4
  step sub${v}_halo4k 300 python tools/halo_sweep.py 4096 2
done
cp $OUT/lib_main.so paresis_amd/libparesis_hip.so
rm -f $OUT/lib_main.so
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/r4s7/sub*.out")):
    txt=open(f).read()
    try:
        j=json.loads([l for l in txt.splitlines() if l.startswith("{")][-1])
        k=dict(j.get("kernel_ms_per_step",{})); k.update({a:b for a,b in (j.get("kernel_ms_short_launches") or {}).items() if a!="note"})
        print(os.path.basename(f), j.get("ms_per_step"), {a:k[a] for a in k if "refract" in a})
    except Exception:
        for l in txt.splitlines():
            if l.startswith("N "): print(os.path.basename(f), l[:40], l[l.index("kernels"):][:200])
PY
