"""stdin: JSON lines of scripts/bench_kinds.py / bench_split_shapes.py -> one `config  us per step` line each."""
import json
import sys

for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        ms = (d.get("roofline") or {}).get("kernel_ms") or d.get("ms_per_step") or 0.0
        print("%-70s %8.1f us  errors %s" % (d.get("config", "")[:70], ms * 1e3, d.get("errors", "-")))
