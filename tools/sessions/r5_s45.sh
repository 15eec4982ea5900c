#!/bin/bash
# Round 5, GPU session 45: the images of an energy bin detected in one launch per stage (psx_detect_multi_f32) and the
# reference image of a bin's first energy written in place (sum-only pass): tests, then the position loops with the kernels
# of a position.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s45
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -p no:cacheprovider -k "detector or poisson" > $OUT/tests_det.out 2>&1; echo "detector tests rc $?"; tail -2 $OUT/tests_det.out
timeout -k 10 900 python -m pytest tests/test_gpu_experiment.py tests/test_gpu_fuzz.py -m gpu -x -q -p no:cacheprovider > $OUT/tests_chain.out 2>&1; echo "chain tests rc $?"; tail -2 $OUT/tests_chain.out
timeout -k 10 300 python tools/time_positions.py 4096 32 > $OUT/positions.out 2>&1; echo "positions rc $?"; grep -E "positions of|library kernels" $OUT/positions.out
