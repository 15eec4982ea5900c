#!/bin/bash
# Round 6, GPU session 9: A/B on one box of (i) skipping the stores of the legs before sample 0, (ii) static priority for the
# younger engine waves, (iii) stage A's twiddle powers read with the inputs; then the Fresnel parity tests on the build in the tree.
cd "$(dirname "$0")/../.."
tools/ab_run.sh $PWD/gpurun_out/r6s9 base noskip prio twearly
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py -x -q -k "fresnel" > gpurun_out/r6s9/t1.out 2>&1; echo "kernels fresnel rc $?"; tail -2 gpurun_out/r6s9/t1.out
