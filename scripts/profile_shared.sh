#!/bin/bash
# SQ counters and HBM-side traffic of the shared-model legs (scripts/bench_kinds.py vshared sshared): separate --pmc passes, csv only.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_shared
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"
rocprofv3 --pmc $SQ --output-format csv -d $OUT/sq -- python3 scripts/bench_kinds.py vshared sshared > $OUT/sq.out 2> $OUT/sq.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 scripts/bench_kinds.py vshared sshared > $OUT/fetch.out 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 scripts/bench_kinds.py vshared sshared > $OUT/write.out 2> $OUT/write.log
python3 - <<'PY'
import collections, csv, glob, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "prof_shared")
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "true>" in r["Kernel_Name"] and ("vanilla_reg" in r["Kernel_Name"] or "squareroot_reg" in r["Kernel_Name"] or "information_reg" in r["Kernel_Name"]):
            rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "shared_counters.txt"), "w") as fo:
    for kn, cs in rows.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        line = kn.replace("void kb::", "")[:100] + " launches %d" % len(next(iter(cs.values())))
        if "SQ_WAVES" in m:
            w, wc = m["SQ_WAVES"], m["SQ_WAVE_CYCLES"]
            line += " | VALU/wave %.0f SALU/wave %.0f active %.0f%% issue-stalled %.0f%% waiting %.0f%%" % (
                m["SQ_INSTS_VALU"] / w, m["SQ_INSTS_SALU"] / w, 100 * m["SQ_ACTIVE_INST_ANY"] / wc, 100 * m["SQ_WAIT_INST_ANY"] / wc, 100 * m["SQ_WAIT_ANY"] / wc)
        if "FETCH_SIZE" in m:
            line += " | read B %.0f" % (m["FETCH_SIZE"] * 1024 * 2)
        if "WRITE_SIZE" in m:
            line += " | written B %.0f" % (m["WRITE_SIZE"] * 1024)
        fo.write(line + "\n")
        print(line)
PY
find $OUT -name "*.csv" -delete
