#!/bin/bash
# rocprofv3 evidence for SURVEY 8d config E (256k SRIF 12/6 fp32): kernel durations, then HBM traffic of the
# fused kernel (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, MI355X_MICROARCH.md "HBM").
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_srif
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 scripts/bench_kinds.py srif > $OUT/bench.jsonl 2> $OUT/stats.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 scripts/bench_kinds.py srif > /dev/null 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 scripts/bench_kinds.py srif > /dev/null 2> $OUT/write.log
head -1 $OUT/bench.jsonl
find $OUT -name "*kernel_stats.csv" -exec head -6 {} \; | cut -c1-200
python3 - <<'PY'
import csv, glob, os, statistics
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/prof_srif"
for tag, col in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in glob.glob(out + "/" + tag + "/**/*counter_collection.csv", recursive=True):
        vals = {}
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == col and "srif_fused_kernel<float" in r.get("Kernel_Name", ""):
                vals.setdefault(r["Dispatch_Id"], 0.0)
                vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
        if vals:
            print(tag, col, "per launch: median", statistics.median(vals.values()), "n", len(vals))
PY
