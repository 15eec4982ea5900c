#!/bin/bash
# Round 5, GPU session 67: the deposit's eight weight products as four v_pk_mul_f32 (PSX_PK_WEIGHTS) -- A/B as whole libraries,
# two rounds on one box (4-distance launch by event pairs; the RT position loop); refraction tests + fuzz on the packed form.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s67
mkdir -p $OUT
cp paresis_amd/libparesis_hip.so $OUT/../.product.so
for round in 1 2; do
  for v in 0 1; do
    cp tools/ab/libparesis_hip_pk$v.so paresis_amd/libparesis_hip.so
    timeout -k 10 300 python tools/halo_sweep.py 4096 2 > $OUT/sweep_pk${v}_$round.out 2>&1; echo "pk $v:" $(grep -o "halo 4.*k_refract_near': [0-9.]*" $OUT/sweep_pk${v}_$round.out | grep -o "k_refract_near': [0-9.]*") | tee -a $OUT/ab.out
    timeout -k 10 300 python tools/time_positions.py 4096 48 --sim RT > $OUT/pos_pk${v}_$round.out 2>&1; echo "pk $v:" $(grep -o "k_refract_near x3 [0-9.]*" $OUT/pos_pk${v}_$round.out) $(grep -o "= [0-9.]* ms per position" $OUT/pos_pk${v}_$round.out) | tee -a $OUT/ab.out
  done
done
cp $OUT/../.product.so paresis_amd/libparesis_hip.so; rm -f $OUT/../.product.so
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_large.py -m gpu -x -q -p no:cacheprovider -k "refract or fastloop or replay or chain" > $OUT/tests.out 2>&1; echo "tests rc $?"; tail -2 $OUT/tests.out
