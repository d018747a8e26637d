"""One line for scripts/bench_chisq.py (A/B: scripts/ab_variants.sh "python scripts/chisq_line.py" base v1 ...)."""
import json
import subprocess
import sys

out = subprocess.run([sys.executable, "scripts/bench_chisq.py"] + sys.argv[1:], capture_output=True, text=True).stdout
for line in out.splitlines():
    if line.startswith("{"):
        d = json.loads(line)
        print("chi-square %d runs x %d steps: %.2f ms  %.1f G run-steps/s  NIS %.4f NEES %.4f" % (d["runs"], d["steps"], d["seconds"] * 1e3, d["run_steps_per_s"] / 1e9, d["nis_mean"], d["nees_mean"]))
