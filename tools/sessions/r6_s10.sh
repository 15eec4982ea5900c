#!/bin/bash
# Round 6, GPU session 10: A/B on one box: the lower half of the legs not stored (two compiled forms behind one uniform branch),
# + static priority for engine waves 4-7, + twiddle powers read early, + loader priority during the fetch.
cd "$(dirname "$0")/../.."
tools/ab_run.sh $PWD/gpurun_out/r6s10 noskip half halfprio combo comboLd
