"""The counter files under profiles/ are only quoted by bench.py while they describe the kernel sources of this checkout
(gokalman_amd/roofline.py kernel_source_hash)."""
import json
import os
import shutil

from gokalman_amd import roofline as rl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_counter_documents_are_bound_to_the_kernel_sources(tmp_path):
    h = rl.kernel_source_hash(ROOT)
    assert len(h) == 16 and h == rl.kernel_source_hash(ROOT)
    assert rl.counters_current(ROOT, {"source_hash": h})[0]
    ok, note = rl.counters_current(ROOT, {"source_hash": "0" * 16})
    assert not ok and note["matches_sources"] is False and note["source_hash"] == h
    assert not rl.counters_current(ROOT, {})[0]   # the files of the earlier rounds carry a git head only
    # a one-byte change of a kernel source changes the hash
    copy = tmp_path / "repo"
    for sub in ("gokalman_amd/csrc", "include"):
        os.makedirs(copy / sub, exist_ok=True)
    for f in os.listdir(os.path.join(ROOT, "gokalman_amd", "csrc")):
        if f.endswith((".hip", ".h", ".inc")):
            shutil.copy(os.path.join(ROOT, "gokalman_amd", "csrc", f), copy / "gokalman_amd" / "csrc" / f)
    shutil.copy(os.path.join(ROOT, "include", "gokalman_amd.h"), copy / "include" / "gokalman_amd.h")
    assert rl.kernel_source_hash(str(copy)) == h
    with open(copy / "gokalman_amd" / "csrc" / "kb_vanilla_reg.h", "a") as fo:
        fo.write("\n")
    assert rl.kernel_source_hash(str(copy)) != h


def test_load_traffic_falls_back_when_the_sources_moved_on(tmp_path):
    os.makedirs(tmp_path / "profiles")
    doc = {"tag": "t", "source_hash": "0" * 16, "kernels": [{"kernel": "k<double>", "filters": 4, "hbm_bytes_per_launch": 400.0}]}
    json.dump(doc, open(tmp_path / "profiles" / "traffic_latest.json", "w"))
    # (the hash is always the hash of THIS checkout's sources: pass the real root for them, the temporary one for the file)
    real = rl.kernel_source_hash
    try:
        rl.kernel_source_hash = lambda root=None: real(ROOT)
        bpf, src = rl.load_traffic(str(tmp_path), "k<double>")
        assert bpf is None and src["matches_sources"] is False and "analytic" in src
        doc["source_hash"] = real(ROOT)
        json.dump(doc, open(tmp_path / "profiles" / "traffic_latest.json", "w"))
        bpf, src = rl.load_traffic(str(tmp_path), "k<double>")
        assert bpf == 100.0 and src["matches_sources"] is True
    finally:
        rl.kernel_source_hash = real
