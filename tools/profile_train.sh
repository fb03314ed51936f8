#!/bin/bash
# rocprofv3 --kernel-trace --stats of the training bench (gpurun, repository root): tools/profile_train.sh <tag> [bench.py flags]
set -o pipefail
TAG=${1:-train}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --mode train --batch 32 --steps 6 --warmup 3 "$@" > $OUT/bench.json.log 2>&1 || exit 1
find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
