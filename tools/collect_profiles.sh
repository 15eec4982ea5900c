#!/bin/bash
# Collects the rocprofv3 evidence for profiles/ on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash tools/collect_profiles.sh'                      # the default bench run (4096^2)
#   gpurun --timeout 900 -- 'bash tools/collect_profiles.sh _16384 --size 16384 --steps 3 --warmup 1'
#   gpurun --timeout 900 -- 'bash tools/collect_profiles.sh _cfg5 --only-configs --configs 16384'   # config 5 as on the driver's line
# then, back in the container:  python tools/summarise_profiles.py r02 [_16384]
# Kernel statistics and every PMC group are separate runs (counters are never combined with a trace), the profiled
# program is `python3 bench.py ...` itself (nothing between `--` and it).
set -e
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
SFX=${1:-}
shift || true
for arg in "$@"; do
    case "$arg" in
        --gpus|--gpus=*) echo "collect_profiles.sh: profile ONE rank (bench.py --gpus N would start its ranks from a process the profiler's library has already initialised the GPU in)" >&2; exit 2;;
    esac
done
CMD="python3 $ROOT/bench.py --no-cpu-baseline --positions 0 --no-configs $*"      # the bench run (timed steps + the per-kernel event pass), minus the CPU leg and the positions batch
case " $* " in
    *" --only-configs "*) CMD="python3 $ROOT/bench.py $*";;     # a `configs` entry alone, as the driver's line runs it (GPU membrane, halo 8, detector)
esac
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/fin_stats$SFX $OUT/fin_fetch$SFX $OUT/fin_write$SFX $OUT/fin_sq$SFX
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fin_stats$SFX -o runc -- $CMD > $OUT/fin_stats$SFX.log 2>&1
echo stats$SFX >> $OUT/progress.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fin_fetch$SFX -o runc -- $CMD > $OUT/fin_fetch$SFX.log 2>&1
echo fetch$SFX >> $OUT/progress.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/fin_write$SFX -o runc -- $CMD > $OUT/fin_write$SFX.log 2>&1
echo write$SFX >> $OUT/progress.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT \
    --output-format csv -d $OUT/fin_sq$SFX -o runc -- $CMD > $OUT/fin_sq$SFX.log 2>&1
echo sq$SFX >> $OUT/progress.log
ls $OUT/fin_stats$SFX $OUT/fin_fetch$SFX $OUT/fin_write$SFX $OUT/fin_sq$SFX
