#!/bin/bash
# Round 5, GPU session 78: the rocprofv3 evidence for profiles/r05_* again, on the round's final build: the default bench run (4096^2)
# and config 5 as the driver's line runs it (kernel statistics and each PMC group in separate runs).
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=$PWD
bash tools/collect_profiles.sh "" && echo "set 1 done"
bash tools/collect_profiles.sh _cfg5 --only-configs --configs 16384 --no-config-parity && echo "set 3 done"
