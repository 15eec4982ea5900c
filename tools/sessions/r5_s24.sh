#!/bin/bash
# Round 5, GPU session 24: A/B of the tile kernel with the phase gradients formed once per tile (float32 pairs in the phase's LDS
# words) against the shipped form: refraction tests on the variant, the headline step, single-distance positions, config 5's sweep.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s24
mkdir -p $OUT
cp paresis_amd/libparesis_hip.so /tmp/keep.so
cp tools/ab/libparesis_hip_pre1.so paresis_amd/libparesis_hip.so
PSX_FUZZ=3 timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_large.py -x -q -p no:cacheprovider -k "refract or order or darkfield or chain" > $OUT/tests_pre1.out 2>&1; rc=$?; tail -3 $OUT/tests_pre1.out
for rep in 1 2; do
for v in pre0 pre1; do
  cp tools/ab/libparesis_hip_$v.so paresis_amd/libparesis_hip.so
  timeout -k 10 300 python bench.py --no-configs --positions 0 --no-cpu-baseline > $OUT/bench_$v.out 2> $OUT/bench_$v.err
  python - <<PY | tee -a $OUT/ab.out
import json
d = json.loads(open("$OUT/bench_$v.out").read().strip().splitlines()[-1])
print("$v", d["ms_per_step"], d["kernel_ms_per_step"], d["parity"]["refraction"])
PY
  timeout -k 10 200 python tools/time_positions.py 4096 32 --sim RT > $OUT/pos_$v.out 2>&1; echo "$v positions:" $(grep -o "k_refract_near x3 [0-9.]*" $OUT/pos_$v.out) $(grep -o "= [0-9.]* ms per position" $OUT/pos_$v.out) | tee -a $OUT/ab.out
done
done
for v in pre0 pre1; do
  cp tools/ab/libparesis_hip_$v.so paresis_amd/libparesis_hip.so
  timeout -k 10 400 python tools/halo_sweep.py 16384 4 float > $OUT/halo_$v.out 2>&1; grep "halo 8\|halo 12" $OUT/halo_$v.out | sed "s/^/$v /" | tee -a $OUT/ab.out
done
cp /tmp/keep.so paresis_amd/libparesis_hip.so
exit $rc
