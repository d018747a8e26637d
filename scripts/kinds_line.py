"""Prints 'config µs' for the given legs of scripts/bench_kinds.py (A/B runs: scripts/ab_variants.sh "python scripts/kinds_line.py vnoise" base v1 ...)."""
import json
import subprocess
import sys

out = subprocess.run([sys.executable, "scripts/bench_kinds.py"] + sys.argv[1:], capture_output=True, text=True).stdout
for line in out.splitlines():
    if line.startswith("{"):
        d = json.loads(line)
        if "ms_per_step" in d:
            print("%-60s %8.1f us  errors %s" % (d["config"][:60], d["ms_per_step"] * 1e3, d.get("errors")))
        else:
            print("%-60s %8.2f ms  %.1f G run-steps/s" % (d["config"][:60], d["seconds"] * 1e3, d["run_steps_per_s"] / 1e9))
