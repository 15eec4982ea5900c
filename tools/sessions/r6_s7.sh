#!/bin/bash
# Round 6, GPU session 7: DUAL rounds at R1 = 32: waves 4-7 pass barrier (4) first and finish the round during the next forward stage A; ABI 9 library.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s7
mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -k "fresnel" > $OUT/t1.out 2>&1; rc=$?; echo "kernels fresnel rc $rc"; tail -5 $OUT/t1.out
[ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python -m pytest tests/test_gpu_large.py -x -q -k "fresnel or shared_forward or engines_agree or oracle_spot" > $OUT/t2.out 2>&1; rc=$?; echo "large rc $rc"; tail -5 $OUT/t2.out
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/stamp_fresnel.py 4096 1 > $OUT/stamp_p2.out 2>&1; echo "stamp rc $?"; cat $OUT/stamp_p2.out
PSX_SWITCHES="stamp_pass1=1" timeout -k 10 300 python tools/stamp_fresnel.py 4096 4 > $OUT/stamp_p1.out 2>&1; echo "stamp pass1 rc $?"; cat $OUT/stamp_p1.out
timeout -k 10 500 python bench.py --no-configs --no-cpu-baseline --positions 0 --steps 20 > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc $?"; tail -c 3000 $OUT/bench.out
