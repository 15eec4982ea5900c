#!/bin/bash
# Round 5, GPU session 58: SQ counters of the RT position loop's kernels (what bounds k_membrane_layers: the splat's instruction
# count did not -- s57), two --pmc passes, nothing else traced.
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5s58
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \
    --output-format csv -d $OUT/p1 -o t -- python3 $ROOT/tools/time_positions.py 4096 16 --sim RT > $OUT/p1.log 2>&1 || { echo "pass 1 failed"; tail -3 $OUT/p1.log; exit 1; }
timeout -k 10 500 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU \
    --output-format csv -d $OUT/p2 -o t -- python3 $ROOT/tools/time_positions.py 4096 16 --sim RT > $OUT/p2.log 2>&1 || { echo "pass 2 failed"; tail -3 $OUT/p2.log; }
python3 $ROOT/tools/pmc_positions.py $(ls $OUT/p1/*counter_collection.csv $OUT/p1/*/*counter_collection.csv $OUT/p2/*counter_collection.csv $OUT/p2/*/*counter_collection.csv 2>/dev/null) | tee $OUT/pmc.txt
rm -rf $OUT/p1 $OUT/p2
