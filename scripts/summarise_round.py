"""Condenses gpurun_out/prof_<tag>/ (scripts/profile_round.sh) into profiles/<tag>/ and refreshes
profiles/traffic_latest.json and profiles/valu_latest.json (read by bench.py for roofline.traffic / fused.roofline).

    python scripts/summarise_round.py <tag> --condense     (on the GPU box: CSVs -> condensed.json, CSVs removed)
    python scripts/summarise_round.py <tag>                (here: condensed.json -> profiles/<tag>/*.md, *.json)
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gokalman_amd import roofline as rl   # noqa: E402  (kernel_source_hash: the counter files name the sources they were measured on)
tag = sys.argv[1]
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
DST = os.path.join(ROOT, "profiles", tag)


def short(name):
    return name.replace("void kb::", "").replace("(kb::StepArgs)", "").strip()


def condense():
    # the sources the counters were measured on: profile_round.sh writes the hash BEFORE its first pass (on the GPU box, where the
    # snapshot cannot change under it); a hash taken later would stamp edited sources as measured
    hf = os.path.join(SRC, "source_hash.txt")
    if not os.path.exists(hf):
        sys.exit("summarise_round: %s missing -- counters without the hash of the sources they were collected on are not summarised" % hf)
    out = {"stats": {}, "pmc": {}, "source_hash": open(hf).read().strip()}
    for d in sorted(glob.glob(os.path.join(SRC, "*"))):
        if not os.path.isdir(d):
            continue
        key = os.path.basename(d)
        for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
            out["stats"][key] = [dict(r) for r in list(csv.DictReader(open(f)))[:12]]
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = collections.defaultdict(lambda: collections.defaultdict(list))
            regs = {}
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
                regs[r["Kernel_Name"]] = {kk: r.get(kk) for kk in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size")}
            out["pmc"][key] = {kn: {"launches": len(next(iter(cs.values()))), "regs": regs[kn],
                                   "mean": {c: sum(v) / len(v) for c, v in cs.items()},
                                   "last": {c: v[-1] for c, v in cs.items()}} for kn, cs in acc.items()}
    json.dump(out, open(os.path.join(SRC, "condensed.json"), "w"), indent=1)
    for d in glob.glob(os.path.join(SRC, "*")):
        if os.path.isdir(d):
            shutil.rmtree(d)


def jsonl(path):
    out = []
    if os.path.exists(path):
        for line in open(path):
            if line.startswith("{"):
                out.append(json.loads(line))
    return out


WARMER = "vanilla_reg_kernel<double, 6, 3, 0, false, false, false, false, false>"   # warm_clocks() of bench_kinds.py / bench_chisq.py


def main():
    if "--condense" in sys.argv:
        condense()
        return
    c = json.load(open(os.path.join(SRC, "condensed.json")))
    os.makedirs(DST, exist_ok=True)
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
    try:   # re-summarising an existing round keeps the commit the profile was TAKEN at
        head = json.load(open(os.path.join(DST, "summary.json")))["head"]
    except Exception:
        pass
    for name in ("bench_plain", "bench_stats", "kinds_plain", "chisq_plain", "diag_stream", "diag_lanepair", "diag_lanequad", "diag_launch_latency", "latency_n1"):
        p = os.path.join(SRC, name + ".out")
        if os.path.exists(p):
            shutil.copy(p, os.path.join(DST, {"bench_plain": "bench.json", "bench_stats": "bench_under_rocprof.json", "kinds_plain": "bench_kinds.jsonl",
                                             "chisq_plain": "bench_chisq.json", "diag_stream": "diag_stream.txt", "diag_lanepair": "diag_lanepair.txt",
                                             "diag_lanequad": "diag_lanequad.txt", "diag_launch_latency": "diag_launch_latency.txt", "latency_n1": "latency_n1.txt"}[name]))
    for name, dst in (("bench_plain.full.json", "bench_full.json"), ("bench_stats.full.json", "bench_full_under_rocprof.json")):
        if os.path.exists(os.path.join(SRC, name)):   # the full documents behind the compact stdout lines (gokalman_amd/benchline.py)
            shutil.copy(os.path.join(SRC, name), os.path.join(DST, dst))
    md = ["# rocprofv3 summary %s (head %s)\n" % (tag, head),
          "Collected by `scripts/profile_round.sh %s` on one MI355X through gpurun; every `--pmc` pass is a separate run.\n" % tag]
    # ---- kernel stats
    for key, title in (("bench_stats", "headline bench (`bench.py --steps 40`)"), ("kinds_stats", "other configs (`scripts/bench_kinds.py`)")):
        rows = c["stats"].get(key, [])
        md.append("\n## --kernel-trace --stats: %s\n\n| kernel | calls | avg ns | total ns | %% |\n|---|---|---|---|---|" % title)
        for r in rows[:12]:
            if key == "kinds_stats" and WARMER in r.get("Name", ""):
                continue   # bench_kinds.py's clock warmer (256k filters of the headline kernel, untimed)
            md.append("| `%s` | %s | %s | %s | %s |" % (short(r.get("Name", ""))[:100], r.get("Calls"), r.get("AverageNs"), r.get("TotalDurationNs"), r.get("Percentage")))
    # ---- traffic
    traffic = []
    md.append("\n## HBM-side traffic per launch (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE; KiB counters -> bytes)\n")
    md.append("| kernel | launches | read B | written B | total B | per filter |\n|---|---|---|---|---|---|")
    filters = {"vanilla_reg_kernel<double, 6, 3, 0, false, false, false, false, true": 1 << 20,   # AWGN (bench_kinds vnoise)
               "vanilla_reg_kernel<double, 6, 3, 0": 1 << 20, "squareroot_reg_kernel<double, 6, 3": 1 << 20, "information_reg_kernel<double, 6, 3": 1 << 20,
               "hybrid_reg_kernel<double, 6, 2": 1 << 20, "srif_pair_kernel<float, 12, 6": 1 << 18, "srif_pair_kernel<double, 12, 6": 1 << 18,
               "vanilla_split_kernel<double, 12, 6": 1 << 18, "squareroot_split_kernel<double, 12, 6": 1 << 18, "information_split_kernel<double, 12, 6": 1 << 18,
               "vanilla_split_kernel<double, 12, 4": 1 << 18, "vanilla_split_kernel<double, 16, 4": 1 << 18, "vanilla_split_kernel<double, 12, 8": 1 << 18, "vanilla_split_kernel<double, 16, 8": 1 << 18, "squareroot_split_kernel<double, 8, 4": 1 << 18, "information_split_kernel<double, 8, 4": 1 << 18,
               "srif_split_kernel<": 1 << 18}
    for fkey, wkey in (("bench_fetch", "bench_write"), ("kinds_fetch", "kinds_write")):
        fe, wr = c["pmc"].get(fkey, {}), c["pmc"].get(wkey, {})
        for kn, e in fe.items():
            if fkey == "kinds_fetch" and WARMER in kn:
                continue
            sub = next((s for s in filters if s in kn), None)
            if sub is None or "FETCH_SIZE" not in e["mean"]:
                continue
            rd = 2.0 * e["mean"]["FETCH_SIZE"] * 1024.0
            w = wr.get(kn, {}).get("mean", {}).get("WRITE_SIZE", 0.0) * 1024.0
            n = filters[sub]
            md.append("| `%s` | %d | %.0f | %.0f | %.0f | %.1f |" % (short(kn)[:90], e["launches"], rd, w, rd + w, (rd + w) / n))
            traffic.append({"tag": tag, "head": head, "kernel": kn, "filters": n, "hbm_bytes_per_launch": rd + w,
                            "fetch_bytes_raw": e["mean"]["FETCH_SIZE"] * 1024.0, "write_bytes": w})
    # ---- SQ counters
    md.append("\n## SQ counters per launch (means; SQ_WAVE_CYCLES / WAIT / ACTIVE count quad-cycles summed over waves)\n")
    md.append("(register / LDS / scratch allocation per kernel: resource_usage.md, from the code objects' metadata)\n")
    md.append("| kernel | waves | VALU / wave | SALU / wave | wave quad-cycles / wave | active | issue-stalled | waiting | LDS B / workgroup |\n|---|---|---|---|---|---|---|---|---|")
    valu = {}
    for key in ("bench_sq", "kinds_sq", "chisq_sq"):
        for kn, e in c["pmc"].get(key, {}).items():
            m = e["last"] if ("mc_kernel" in kn or "chisq_kernel" in kn) else e["mean"]   # multi-step kernels: the timed launch, see below
            if "SQ_WAVES" not in m or m["SQ_WAVES"] < 64 or m.get("SQ_INSTS_VALU", 0) / m["SQ_WAVES"] < 200:
                continue
            wv = m["SQ_WAVES"]
            wc = m["SQ_WAVE_CYCLES"]
            if "rocclr" in kn or "at::native" in kn or (key != "bench_sq" and WARMER in kn):
                continue
            md.append("| `%s` | %.0f | %.0f | %.0f | %.0f | %.0f%% | %.0f%% | %.0f%% | %s |" % (
                short(kn)[:90], wv, m["SQ_INSTS_VALU"] / wv, m["SQ_INSTS_SALU"] / wv, wc / wv, 100 * m["SQ_ACTIVE_INST_ANY"] / wc,
                100 * m["SQ_WAIT_INST_ANY"] / wc, 100 * m["SQ_WAIT_ANY"] / wc, e["regs"].get("LDS_Block_Size")))
            valu[kn] = {"waves": wv, "valu_insts_per_wave": m["SQ_INSTS_VALU"] / wv}
            if "mc_kernel" in kn or "chisq_kernel" in kn:
                # these run `steps` filter steps per launch and the scripts warm up with a SHORT launch first: the mean over the
                # dispatches is not the count of the timed launch (r03a and earlier took the mean: mc read 302 045 instead of
                # 599 573 VALU per wave, which halved its issue-roofline fraction).  The timed launch is the last one.
                la = e["last"]
                valu[kn] = {"waves": la["SQ_WAVES"], "valu_insts_per_wave": la["SQ_INSTS_VALU"] / la["SQ_WAVES"], "dispatch": "last (the timed launch)"}
    named = {}
    for kn, v in valu.items():
        # the time-fused kernels by their FULL template argument list <T, NS, NM, NC, FULL, PREDICT, FUSED, PAD, NOISE, SHARED>: the
        # prefix "...0, false, false, true" matches the Noiseless AND the AWGN instantiation since round 5, and the last one won
        # (r05g priced the Noiseless leg with the AWGN kernel's 290 019 VALU per wave: frac 1.43)
        for name, full in (("vanilla_fused", "vanilla_reg_kernel<double, 6, 3, 0, false, false, true, false, false, false>"),
                           ("vanilla_fused_awgn", "vanilla_reg_kernel<double, 6, 3, 0, false, false, true, false, true, false>")):
            if full in kn:
                T = 100   # bench.py's --fused-steps default since round 4
                try:
                    full_doc = os.path.join(DST, "bench_full_under_rocprof.json")
                    if os.path.exists(full_doc):
                        bj = json.load(open(full_doc))
                    else:
                        bj = [json.loads(l) for l in open(os.path.join(DST, "bench_under_rocprof.json")) if l.startswith("{")][-1]
                    T = int(bj["fused"]["steps_per_launch"])
                except Exception:
                    pass
                named[name] = dict(v, kernel=kn, steps_per_launch=T, valu_insts_per_wave_per_step=v["valu_insts_per_wave"] / float(T))
        if "mc_kernel<double, 4, 2" in kn:
            named["mc"] = dict(v, kernel=kn, steps_per_launch=1086)
        if "chisq_kernel" in kn:
            named["chisq"] = dict(v, kernel=kn, steps_per_launch=1086)
        if "srif_pair_kernel<float, 12, 6, false, true" in kn:
            named["srif_pair_f32"] = dict(v, kernel=kn)
        if "srif_pair_kernel<double, 12, 6, false, true" in kn:
            named["srif_pair_f64"] = dict(v, kernel=kn)
    open(os.path.join(DST, "summary.md"), "w").write("\n".join(md) + "\n")
    json.dump({"tag": tag, "head": head, "condensed": c}, open(os.path.join(DST, "summary.json"), "w"), indent=1)
    if traffic:
        json.dump({"tag": tag, "head": head, "source_hash": c["source_hash"], "kernels": traffic,
                   "note": "FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM) + WRITE_SIZE; separate --pmc passes; the counters sit on the "
                           "L2's fabric side and include Infinity-Cache hits"}, open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
    if named:
        json.dump({"tag": tag, "head": head, "source_hash": c["source_hash"], "kernels": named, "note": "SQ_INSTS_VALU / SQ_WAVES per launch"},
                  open(os.path.join(ROOT, "profiles", "valu_latest.json"), "w"), indent=1)
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "resource_usage.py"), "--md", os.path.join(DST, "resource_usage.md"),
                    "reg_kernel", "srif_", "mc_kernel", "chisq_", "split_kernel"], stdout=subprocess.DEVNULL)
    print("\n".join(md))


if __name__ == "__main__":
    main()
