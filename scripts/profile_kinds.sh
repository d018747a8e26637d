#!/bin/bash
# rocprofv3 --kernel-trace --stats for the other configs (SURVEY 8d C, D, E + Information).
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profk_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 scripts/bench_kinds.py > $OUT/bench_kinds.jsonl 2> $OUT/stats.log
cat $OUT/bench_kinds.jsonl
find $OUT -name "*kernel_stats.csv" -exec head -20 {} \;
