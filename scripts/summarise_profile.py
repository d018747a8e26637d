"""Condenses the rocprofv3 CSVs from scripts/profile_bench.sh into profiles/<tag>_summary.md/json."""
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {"tag": tag}


def find(pattern):
    m = glob.glob(os.path.join(out, pattern), recursive=True)
    return m[0] if m else None


stats = find("stats/**/*kernel_stats.csv")
lines = []
if stats:
    rows = list(csv.DictReader(open(stats)))
    res["kernel_stats"] = rows[:8]
    lines.append("## rocprofv3 --kernel-trace --stats (top kernels)\n")
    lines.append("| kernel | calls | avg ns | total ns | % |\n|---|---|---|---|---|")
    for r in rows[:8]:
        lines.append("| %s | %s | %s | %s | %s |" % (r.get("Name", "")[:90], r.get("Calls"), r.get("AverageNs"),
                                                   r.get("TotalDurationNs"), r.get("Percentage")))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = find("pmc_%s/**/*counter_collection.csv" % ("fetch" if ctr == "FETCH_SIZE" else "write"))
    if not f:
        continue
    vals = {}
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != ctr:
            continue
        vals.setdefault(r.get("Kernel_Name", ""), []).append(float(r.get("Counter_Value", 0)))
    res[ctr] = {kname[:90]: {"launches": len(v), "mean": sum(v) / len(v)} for kname, v in vals.items()}
    lines.append("\n## --pmc %s (mean per launch, raw counter units = KiB)\n" % ctr)
    for kname, v in vals.items():
        lines.append("- `%s`: %d launches, mean %.1f" % (kname[:90], len(v), sum(v) / len(v)))
# HBM traffic of the dominant kernel, corrected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE under-reports wide coalesced reads by exactly 2x on gfx950; WRITE_SIZE is exact; unit KiB.
dom = None
for kname in res.get("FETCH_SIZE", {}):
    if "vanilla_reg_kernel" in kname:
        dom = kname
if dom:
    rd = res["FETCH_SIZE"][dom]["mean"] * 1024.0
    wr = res.get("WRITE_SIZE", {}).get(dom, {}).get("mean", 0.0) * 1024.0
    res["dominant_kernel"] = dom
    res["fetch_bytes_raw"] = rd
    res["write_bytes_raw"] = wr
    res["hbm_bytes_per_launch"] = 2.0 * rd + wr
    lines.append("\n## HBM traffic per launch of `%s`\n" % dom)
    lines.append("- FETCH_SIZE raw %.0f B -> x2 (gfx950 128-B requests tallied at 64 B) = %.0f B read" % (rd, 2 * rd))
    lines.append("- WRITE_SIZE %.0f B written" % wr)
    lines.append("- total %.0f B per launch" % (2 * rd + wr))
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(out, "%s_summary.json" % tag), "w"), indent=1)
open(os.path.join(out, "%s_summary.md" % tag), "w").write("# rocprofv3 summary %s\n\n" % tag + "\n".join(lines) + "\n")
print("\n".join(lines))
