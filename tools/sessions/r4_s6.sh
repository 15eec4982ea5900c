#!/bin/bash
# Round 4, GPU session 6: hardware sin/cos in the transmission: full tests + bench.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s6
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
step pytest 900 python -m pytest tests -m gpu -q
tail -8 $OUT/pytest.out
step bench_default 500 python bench.py
python - <<'PY'
import json
j=json.loads([l for l in open("gpurun_out/r4s6/bench_default.out").read().splitlines() if l.startswith("{")][-1])
print(j["value"], j["value_cold"], j["ms_per_step"], j["kernel_ms_per_step"], j.get("kernel_ms_short_launches"))
print(j["parity"])
r=j["roofline"]; print({k:r[k] for k in ("frac","frac_per_propagation","step_frac","step_frac_per_propagation")})
for n,e in j["configs"].items(): print(n, e["ms"], e["step_frac"], e.get("step_frac_per_propagation"), e["kernel_ms_per_step"], e.get("kernel_ms_short_launches"), e["parity"], e.get("refraction_halo"), e.get("refraction_halo_tuning_ms"))
for s,e in j["positions_batch"].items(): print(s, e.get("ms_total"), (e.get("warm") or {}).get("ms_total"), e.get("check"))
PY
