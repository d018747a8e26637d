"""The core clock beside the time of a latency-bound kernel (Vanilla 12/6 on the split kernel varies 192-257 us
box to box; "it follows the core clock" had no reading next to it).  Runs the 262 144-filter Vanilla 12/6 step back to back for about a
second per round while a thread samples `rocm-smi --showclocks` (sclk / mclk of GPU 0); prints us per step and the clock samples per round.
usage (inside a gpurun command): python scripts/clock_probe.py [rounds]"""
import json
import re
import subprocess
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import gokalman_amd as ga
from gokalman_amd import _capi as k, synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N, n, p = 1 << 18, 12, 6
d = synth.linear_batch(N, n, p, 1, seed=synth.SEED + 77)
b = ga.FilterBatch.new_ldkf(k.VANILLA, d["x0"], d["P0"], d["F"], None, d["H"], d["Q"], d["R"])
y = torch.from_numpy(np.ascontiguousarray(d["y"][0].T)).cuda()
torch.cuda.synchronize()
stream = torch.cuda.ExternalStream(b.stream())


def clocks():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
    except Exception as e:   # noqa: BLE001
        return {"error": str(e)}
    got = {}
    for name in ("sclk", "mclk", "fclk", "socclk"):
        m = re.search(r"GPU\[0\]\s*:\s*%s clock level: \S+ \((\d+)Mhz\)" % name, out)
        if m:
            got[name] = int(m.group(1))
    return got or {"raw": out[-300:]}


for r in range(rounds):
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(clocks())
            time.sleep(0.15)
    th = threading.Thread(target=sampler)
    th.start()
    for _ in range(200):
        b.update_dev(y.data_ptr(), N)
    b.synchronize()
    K = 4000
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(K):
        b.update_dev(y.data_ptr(), N)
    e1.record(stream)
    b.synchronize()
    stop.set(); th.join()
    us = e0.elapsed_time(e1) / K * 1e3
    sclk = [s.get("sclk") for s in samples if "sclk" in s]
    print(json.dumps({"round": r, "kernel": "vanilla_split_kernel<double,12,6,0,4>", "filters": N, "us_per_step": us,
                      "sclk_mhz": {"min": min(sclk) if sclk else None, "max": max(sclk) if sclk else None, "samples": len(sclk)},
                      "last_sample": samples[-1] if samples else None}), flush=True)
    time.sleep(1.0)
