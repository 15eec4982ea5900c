#!/bin/bash
# Round 5, GPU session 5: how long may a rank's GPU idle at the barrier before its 8-position share pays for cold clocks?
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5s5
mkdir -p $OUT
for idle in 0.1 1 5 20 100; do
  for sim in Fresnel RayT; do
    PSX_EMULATE_IDLE_MS=$idle timeout -k 10 200 python bench.py --emulate-rank 7 --emulate-world 8 --emulate-sim $sim 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('idle $idle ms', '$sim', 'cold', d['cold_ms'], 'warm', d['warm_ms'])" | tee -a $OUT/idle.out
  done
done
