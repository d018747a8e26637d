"""Per-kernel means of the counters in a rocprofv3 rocpd database (--pmc run without --output-format csv).
usage: python scripts/pmc_db.py results.db [kernel-substring]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
acc = defaultdict(lambda: defaultdict(list))
for kn, cn, val in db.execute("select %s, counter_name, value from counters_collection" % kcol):
    if sub in kn:
        acc[kn][cn].append(val)
for kn, cs in acc.items():
    n = len(next(iter(cs.values())))
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    line = "%s: launches %d" % (kn[:80], n)
    w = m.get("SQ_WAVES")
    for c, v in sorted(m.items()):
        line += " | %s %.0f" % (c, v)
    if w and "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"]
        line += " || quad-cycles/wave %.0f" % (wc / w)
        if "SQ_ACTIVE_INST_ANY" in m:
            line += " active %.0f%%" % (100 * m["SQ_ACTIVE_INST_ANY"] / wc)
        if "SQ_WAIT_INST_ANY" in m:
            line += " issue-stalled %.0f%%" % (100 * m["SQ_WAIT_INST_ANY"] / wc)
        if "SQ_WAIT_ANY" in m:
            line += " waiting %.0f%%" % (100 * m["SQ_WAIT_ANY"] / wc)
        if "SQ_INSTS_VALU" in m:
            line += " VALU/wave %.0f" % (m["SQ_INSTS_VALU"] / w)
    print(line)
