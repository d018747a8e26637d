"""Prints 'config µs errors' for the SRIF legs of scripts/bench_kinds.py (A/B runs: scripts/ab_variants.sh "python scripts/srif_line.py" base v1 ...)."""
import json
import subprocess
import sys

out = subprocess.run([sys.executable, "scripts/bench_kinds.py", "srif"] + sys.argv[1:], capture_output=True, text=True).stdout
for line in out.splitlines():
    if line.startswith("{"):
        d = json.loads(line)
        print("%-20s %7.1f us  errors %d" % (d["config"][:20], d["ms_per_step"] * 1e3, d["errors"]))
