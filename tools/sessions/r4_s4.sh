#!/bin/bash
# Round 4, GPU session 4: 2048^2 stamps, halo sweeps, dark field (chain form), tests of the new pieces, profiles r04.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4s4
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
step pytest_new 600 python -m pytest tests -m gpu -x -q -k "halo or darkfield or reproducible or chain_rt"
tail -3 $OUT/pytest_new.out
step darkfield_chain 200 python tools/time_darkfield.py 20 chain
step stamp2048_p2 200 python tools/stamp_fresnel.py 2048 1
PSX_SWITCHES="stamp_pass1=1" step stamp2048_p1 200 python tools/stamp_fresnel.py 2048 1
step stamp4096_p2 200 python tools/stamp_fresnel.py 4096 1
step halo_4096_2 300 python tools/halo_sweep.py 4096 2
step halo_8192_4 300 python tools/halo_sweep.py 8192 4
step halo_16384_4 400 python tools/halo_sweep.py 16384 4
step halo_16384_4_rep 400 python tools/halo_sweep.py 16384 4 reproducible
bash tools/collect_profiles.sh > $OUT/collect_4096.log 2>&1; echo "collect_4096 rc=$?" | tee -a $OUT/progress.log
bash tools/collect_profiles.sh _cfg5 --only-configs --configs 16384 > $OUT/collect_cfg5.log 2>&1; echo "collect_cfg5 rc=$?" | tee -a $OUT/progress.log
for f in $OUT/*.out; do echo "--- $f"; tail -25 $f | cut -c1-600; done
