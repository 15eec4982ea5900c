#!/bin/bash
# Round 6, GPU session 42: the refraction tile kernel with the distances taken two at a time on interior tiles (PSX_NEAR_PAIR: the staged phase as a second accumulator):
# parity (every refraction test), then the A/B at 4096^2 and on config 5.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s42
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "refract or chain or fastloop or darkfield or deterministic or replay or bench_ or xml_experiment" > $OUT/tests.out 2>&1; rc=$?; echo "tests rc $rc"; tail -3 $OUT/tests.out
[ $rc -ne 0 ] && exit $rc
bash tools/ab_run.sh $OUT pair0 pair1
cp paresis_amd/libparesis_hip.so $OUT/.lib_saved2.so
for t in pair0 pair1 pair0 pair1; do
  cp tools/ab/libparesis_hip_$t.so paresis_amd/libparesis_hip.so
  timeout -k 10 300 python bench.py --only-configs --configs 16384 --no-config-parity > $OUT/cfg5_$t.out 2>$OUT/cfg5_$t.err
  python - <<PY
import json
c = json.loads(open("$OUT/cfg5_$t.out").read().strip().splitlines()[-1])["configs"]["16384"]
print("$t", "config 5 step %.3f near %.3f" % (c["ms"], c["kernel_ms_per_step"]["k_refract_near"]), c.get("refraction_halo_tuning_ms"))
PY
done
cp $OUT/.lib_saved2.so paresis_amd/libparesis_hip.so
