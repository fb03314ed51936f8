#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py configuration (run through gpurun from the repository root):
#   tools/profile_config.sh <tag> <bench.py flags...>   ->  gpurun_out/prof_<tag>/{bench.json.log,stats/...}
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline --streams 1 "$@" > $OUT/bench.json.log 2>&1 || exit 1
find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
tail -c 600 $OUT/bench.json.log
