#!/bin/bash
# per-kernel durations of the SRIF 12/6 fp32 step (time + measurement kernels)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_srif
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 scripts/bench_kinds.py srif > $OUT/bench.jsonl 2> $OUT/stats.log
head -1 $OUT/bench.jsonl
find $OUT -name "*kernel_stats.csv" -exec head -6 {} \; | cut -c1-200
