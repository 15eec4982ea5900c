#!/bin/bash
# Round 6, GPU session 47: pass 1's later distance pairs take the line's spectrum from the workgroup's park buffer (PSX_P2_REUSE):
# parity (every Fresnel test + the chains), then the A/B at 4096^2.
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r6s47
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "fresnel or chain or propagate or bench_ or xml_experiment or power_of_two or work_queue or fuzz" > $OUT/tests.out 2>&1; rc=$?; echo "tests rc $rc"; tail -3 $OUT/tests.out
[ $rc -ne 0 ] && exit $rc
bash tools/ab_run.sh $OUT reuse0 reuse1
