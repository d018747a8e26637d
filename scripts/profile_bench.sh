#!/bin/bash
# Collects the rocprofv3 evidence for the headline bench line on the GPU box (run via gpurun):
#   1. --kernel-trace --stats   -> per-kernel average duration
#   2. --pmc FETCH_SIZE         -> HBM read traffic   (own pass, see MI355X_MICROARCH.md "HBM")
#   3. --pmc WRITE_SIZE         -> HBM write traffic  (own pass)
# Usage: scripts/profile_bench.sh <tag>   (outputs under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
ARGS="--steps 40 --warmup 5 --no-cpu-baseline --fused-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/bench_stats.json 2> $OUT/stats.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.log
find $OUT -name "*.csv" | head -20
python3 scripts/summarise_profile.py $OUT $TAG
