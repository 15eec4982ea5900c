#!/bin/bash
# Collects the rocprofv3 evidence for profiles/ on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash tools/collect_profiles.sh'
# then, back in the container:  python tools/summarise_profiles.py r01
# Kernel statistics and every PMC group are separate runs (counters are never combined with a trace), the profiled
# program is `python3 bench.py ...` itself (nothing between `--` and it).
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
CMD="python3 $ROOT/bench.py --no-cpu-baseline --positions 0 ${BENCH_ARGS:-}"      # the default bench run (50 timed steps + the per-kernel event pass), minus the CPU leg
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/fin_stats $OUT/fin_fetch $OUT/fin_write $OUT/fin_sq
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fin_stats -o runc -- $CMD > $OUT/fin_stats.log 2>&1
echo stats >> $OUT/progress.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fin_fetch -o runc -- $CMD > $OUT/fin_fetch.log 2>&1
echo fetch >> $OUT/progress.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/fin_write -o runc -- $CMD > $OUT/fin_write.log 2>&1
echo write >> $OUT/progress.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT \
    --output-format csv -d $OUT/fin_sq -o runc -- $CMD > $OUT/fin_sq.log 2>&1
echo sq >> $OUT/progress.log
ls $OUT/fin_stats $OUT/fin_fetch $OUT/fin_write $OUT/fin_sq
