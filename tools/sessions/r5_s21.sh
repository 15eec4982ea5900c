#!/bin/bash
# Round 5, GPU session 21: rocprofv3 evidence for profiles/r05_*: the default bench run (4096^2), the same with the fixed-point
# replay on the step (--deterministic-step), and config 5 as the driver's line runs it.
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=$PWD
bash tools/collect_profiles.sh "" && echo "set 1 done"
bash tools/collect_profiles.sh _det --deterministic-step && echo "set 2 done"
bash tools/collect_profiles.sh _cfg5 --only-configs --configs 16384 --no-config-parity && echo "set 3 done"
