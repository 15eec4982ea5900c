#!/bin/bash
# Round 5, GPU session 38: what would a uniform support layer passed as a scalar be worth?  Timing experiment: the position loops with
# the membrane's uniform support map left out (physics differs: a timing probe only).
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s38
mkdir -p $OUT
timeout -k 10 400 python tools/_tmp_onemap.py 2>/dev/null | tee $OUT/onemap.out
