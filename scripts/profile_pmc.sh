#!/bin/bash
# rocprofv3 --pmc pass over one bench_kinds configuration (own run, no tracing flags beside it):
#   scripts/profile_pmc.sh <tag> <bench_kinds config> "<COUNTER COUNTER ...>"
# writes gpurun_out/pmc_<tag>/ and prints per-kernel means.
set -u
TAG=$1; CFG=$2; CTRS=$3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
rocprofv3 --pmc $CTRS --output-format csv -d $OUT -- python3 scripts/bench_kinds.py $CFG > $OUT/bench.jsonl 2> $OUT/log.txt
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if "srif" in k or "vanilla" in k or "mc_kernel" in k or "chisq" in k or "squareroot" in k:
            print(k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
