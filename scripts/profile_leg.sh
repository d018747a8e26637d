#!/bin/bash
# usage (inside a gpurun command): scripts/profile_leg.sh OUTDIR KERNEL_SUBSTRING leg [leg ...]
# Kernel-trace statistics, SQ / LDS counters and HBM-side traffic of bench_kinds.py legs, each in its own rocprofv3 pass
# (--pmc never together with a trace domain); writes OUTDIR/{stats.txt,counters.txt}.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/$1; SUB=$2; shift 2
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/bench_kinds.py "$@" > $OUT/trace.out 2> $OUT/trace.log
pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 scripts/bench_kinds.py $LEGS > $OUT/$name.out 2> $OUT/$name.log || echo "pass $name failed" >> $OUT/errors.txt; }
LEGS="$*"
pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
pass lds SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES
pass vmem SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAVE_CYCLES
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 - "$OUT" "$SUB" <<'PY'
import collections, csv, glob, os, sys
out, sub = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            rows[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "counters.txt"), "w") as fo:
    for kn, cs in rows.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        line = kn.replace("void kb::", "")[:110] + " launches %d\n " % len(next(iter(cs.values())))
        w, wc = m.get("SQ_WAVES"), m.get("SQ_WAVE_CYCLES")
        for c, v in sorted(m.items()):
            line += " %s=%.4g" % (c, v)
        if w and wc:
            line += "\n  per wave: VALU %.0f SALU %.0f" % (m.get("SQ_INSTS_VALU", 0) / w, m.get("SQ_INSTS_SALU", 0) / w)
            for c in ("SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
                if c in m: line += " %s %.0f" % (c[9:], m[c] / w)
            line += " | of wave time:"
            for c, lbl in (("SQ_ACTIVE_INST_ANY", "active"), ("SQ_WAIT_INST_ANY", "issue-stalled"), ("SQ_WAIT_ANY", "waiting"), ("SQ_ACTIVE_INST_VALU", "valu"),
                           ("SQ_ACTIVE_INST_LDS", "lds"), ("SQ_ACTIVE_INST_VMEM", "vmem"), ("SQ_WAIT_INST_LDS", "wait-lds"), ("SQ_LDS_BANK_CONFLICT", "lds-conflict")):
                if c in m: line += " %s %.1f%%" % (lbl, 100 * m[c] / wc)
        if "FETCH_SIZE" in m: line += "\n  read B %.0f" % (m["FETCH_SIZE"] * 1024 * 2)
        if "WRITE_SIZE" in m: line += " written B %.0f" % (m["WRITE_SIZE"] * 1024)
        fo.write(line + "\n"); print(line)
with open(os.path.join(out, "stats.txt"), "w") as fo:
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Name"]:
                s = "%s calls %s avg %.1f us min %.1f max %.1f" % (r["Name"].replace("void kb::", "")[:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3)
                fo.write(s + "\n"); print(s)
PY
find $OUT -name "*.csv" -delete; find $OUT -name "*.db" -delete
