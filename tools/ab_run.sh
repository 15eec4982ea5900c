#!/bin/bash
# On the GPU box: times the headline step's kernels with each library of tools/ab/ named on the command line, twice, interleaved
# (boxes differ by up to 10 %, and a box drifts by a per cent or two: only figures of ONE call compare).
#   tools/ab_run.sh OUTDIR tag1 tag2 ...        (extra bench flags through AB_FLAGS)
OUT=$1; shift
mkdir -p $OUT
cp paresis_amd/libparesis_hip.so $OUT/.lib_saved.so
for rep in 1 2; do
  for tag in "$@"; do
    cp tools/ab/libparesis_hip_$tag.so paresis_amd/libparesis_hip.so
    timeout -k 10 200 python bench.py --no-configs --no-cpu-baseline --positions 0 --steps 30 $AB_FLAGS > $OUT/${tag}_$rep.out 2> $OUT/${tag}_$rep.err || echo "$tag rc $?"
    python - <<PY
import json
d = json.loads(open("$OUT/${tag}_$rep.out").read().strip().splitlines()[-1])
k = d.get("kernel_ms_per_step", {})
print("%-14s run $rep  ms/step %.4f steady %.4f  rows %.4f cols %.4f near %.4f  parity-free" % ("$tag", d["ms_per_step"], d["steady"]["ms_per_step"],
      k.get("k_fresnel_rows", 0), k.get("k_fresnel_cols", 0), k.get("k_refract_near", 0)))
PY
  done
done
cp $OUT/.lib_saved.so paresis_amd/libparesis_hip.so
