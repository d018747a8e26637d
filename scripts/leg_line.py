"""Prints 'config  us' for the bench_kinds.py JSON lines on stdin (A/B runs: scripts/ab_variants.sh "python scripts/bench_kinds.py vsplit | python scripts/leg_line.py" base v1 ...)."""
import json
import sys

for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        if "ms_per_step" in d:
            print("%-60s %8.1f us  errors %s" % (d["config"][:60], d["ms_per_step"] * 1e3, d.get("errors")))
