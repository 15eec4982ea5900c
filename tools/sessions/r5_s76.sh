#!/bin/bash
# Round 5, GPU session 76: instruction counts of k_poisson with and without the exact test's arithmetic (variants po0 / po4), SQ counters.
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5s76
mkdir -p $OUT
cp paresis_amd/libparesis_hip.so $OUT/../.product.so
cd /tmp && export TMPDIR=/tmp
for v in 0 4; do
  cp $ROOT/tools/ab/libparesis_hip_po$v.so $ROOT/paresis_amd/libparesis_hip.so
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_TRANS \
      --output-format csv -d $OUT/p$v -o t -- python3 $ROOT/tools/time_poisson.py > $OUT/p$v.log 2>&1 || { echo "pmc $v failed"; tail -3 $OUT/p$v.log; }
  echo "== variant $v"; python3 $ROOT/tools/pmc_positions.py $(ls $OUT/p$v/*counter_collection.csv $OUT/p$v/*/*counter_collection.csv 2>/dev/null) | grep -A10 "k_poisson" | tee -a $OUT/pmc.txt
  rm -rf $OUT/p$v
done
cp $OUT/../.product.so $ROOT/paresis_amd/libparesis_hip.so; rm -f $OUT/../.product.so
