#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box (run via gpurun); every --pmc pass is its own run with no
# tracing flags beside it (MI355X_MICROARCH.md, "HBM" / "rocprofv3 PMC slots").
#   scripts/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/ ; then scripts/summarise_round.py <tag> writes profiles/<tag>/
# Passes:  headline bench (kernel-trace --stats, FETCH_SIZE, WRITE_SIZE, VALU counters of the fused kernel),
#          other configs (kernel-trace --stats), SRIF (FETCH_SIZE, WRITE_SIZE, SQ counters), MC and chi-square (SQ counters).
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
python3 -c "import sys; sys.path.insert(0, '$ROOT'); from gokalman_amd import roofline as rl; print(rl.kernel_source_hash('$ROOT'))" > $OUT/source_hash.txt
export TMPDIR=/tmp
cd $ROOT
B="--full-out= --steps 40 --warmup 5 --repeat 1 --no-cpu-baseline --no-parity --no-host-path --ooc-filters 0 --mc-runs 0 --chisq-runs 0 --hybrid-filters 0 --sqrt-filters 0 --srif-filters 0 --shared-filters 0 --split-filters 0"
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"
# ONLY=<prefix> repeats the passes whose name starts with it (e.g. ONLY=kinds after a fix to scripts/bench_kinds.py), keeping the others
run() { local name=$1; shift; if [ -n "${ONLY:-}" ] && [[ "$name" != ${ONLY}* ]]; then return; fi; echo "== $name"; "$@" > $OUT/$name.out 2> $OUT/$name.log; }
run bench_plain   python3 bench.py --steps 2000 --warmup 100 --full-out $OUT/bench_plain.full.json
run bench_stats   rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -- python3 bench.py $B --full-out $OUT/bench_stats.full.json
run bench_fetch   rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/bench_fetch -- python3 bench.py $B --fused-steps 0
run bench_write   rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/bench_write -- python3 bench.py $B --fused-steps 0
run bench_sq      rocprofv3 --pmc $SQ --output-format csv -d $OUT/bench_sq -- python3 bench.py $B
run kinds_plain   python3 scripts/bench_kinds.py
run kinds_stats   rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kinds_stats -- python3 scripts/bench_kinds.py
run kinds_fetch   rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/kinds_fetch -- python3 scripts/bench_kinds.py vsplit vpad sqsplit infsplit vnoise vshared sqrt info srif srifpad hybrid --srif-shapes=16x6,14x4,11x4,7x3
run kinds_write   rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/kinds_write -- python3 scripts/bench_kinds.py vsplit vpad sqsplit infsplit vnoise vshared sqrt info srif srifpad hybrid --srif-shapes=16x6,14x4,11x4,7x3
run kinds_sq      rocprofv3 --pmc $SQ --output-format csv -d $OUT/kinds_sq -- python3 scripts/bench_kinds.py vsplit vpad sqsplit infsplit vnoise vshared sqrt info srif srifpad hybrid mc --srif-shapes=16x6,14x4,11x4,7x3
run chisq_plain   python3 scripts/bench_chisq.py
run chisq_sq      rocprofv3 --pmc $SQ --output-format csv -d $OUT/chisq_sq -- python3 scripts/bench_chisq.py
if [ -z "${ONLY:-}" ]; then
hipcc --offload-arch=gfx950 -O3 scripts/diag_stream.hip -o /tmp/diag_stream 2> /dev/null && /tmp/diag_stream > $OUT/diag_stream.out 2>&1
hipcc --offload-arch=gfx950 -O3 scripts/diag_lanepair.hip -o /tmp/diag_lanepair 2> /dev/null && /tmp/diag_lanepair > $OUT/diag_lanepair.out 2>&1
hipcc --offload-arch=gfx950 -O3 scripts/diag_lanequad.hip -o /tmp/diag_lanequad 2> /dev/null && /tmp/diag_lanequad > $OUT/diag_lanequad.out 2>&1
hipcc --offload-arch=gfx950 -O2 scripts/diag_launch_latency.hip -o /tmp/diag_launch_latency 2> /dev/null && timeout 120 /tmp/diag_launch_latency > $OUT/diag_launch_latency.out 2>&1
hipcc -std=c++17 -O2 -Iinclude scripts/latency_n1.cpp -Lgokalman_amd -lgokalman_amd -Wl,-rpath,$ROOT/gokalman_amd -o /tmp/latency_n1 2> /dev/null && timeout 200 /tmp/latency_n1 > $OUT/latency_n1.out 2>&1
fi
find $OUT -name "*.csv" | wc -l
# keep what travels back small: the per-dispatch CSVs are condensed on the box
python3 scripts/summarise_round.py $TAG --condense
du -sh $OUT
