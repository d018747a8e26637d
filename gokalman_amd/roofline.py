"""Byte conventions and roofline arithmetic shared by bench.py and scripts/bench_kinds.py.

Two byte counts exist per filter-step and the bench lines carry both:

  algorithmic -- SURVEY.md section 8d / BASELINE.md section 4: full (unpacked) matrices, state-only
                 outputs, w * (4 n^2 + p n + p^2 + 2 n + p).  What the contract calls `achieved`
                 when divided by the kernel time.  It OVERCOUNTS what the kernels move, because a
                 mat64.SymDense only carries its upper triangle (helper.go:65-84) and the engine
                 stores exactly that; a fraction built on it can exceed 1 and is therefore only
                 reported as `frac_algorithmic`.
  moved       -- the bytes a launch really moves per filter (packed working set, reads + writes),
                 i.e. what the rocprofv3 counters FETCH_SIZE x 2 + WRITE_SIZE report (profiles/).
                 `roofline.frac` is built on this one: measured counter bytes when a counter file
                 for the kernel exists, else this analytic count (they agree to < 1 %).

Peaks: HBM3E 8.0 TB/s spec, 6.29 TB/s achievable (float4 copy) -- MI355X_MICROARCH.md.
"""
import json
import os

HBM_PEAK_GBPS = 8000.0
HBM_ACHIEVABLE_GBPS = 6290.0
# fp64 vector peak: 256 CUs x 4 SIMDs x 16 FMA lanes / clk x 2.4 GHz x 2 flop = 78.6 TFLOP/s; a wave64 fp64
# FMA-class instruction therefore occupies its SIMD's issue port for 4 cycles, and so does an fp32 one issued by a
# wave that is alone on its SIMD (MI355X_MICROARCH.md, "vector-instruction ISSUE cost").
N_SIMD = 256 * 4
CLOCK_HZ = 2.4e9
ISSUE_CYCLES = 4.0


def tri(n):
    return n * (n + 1) // 2


def algorithmic_bytes(kind, n, p, w=8):
    """SURVEY.md 8d figure per filter-step."""
    if kind == "srif":      # b, R, Phi, Htilde, L, real, computed read; b, R written
        return w * (n + n * n + n * n + p * n + p * p + 2 * p + n + n * n)
    if kind == "hybrid":    # x, P, Phi, Htilde, R, real, computed read; x, P written
        return w * (n + n * n + n * n + p * n + p * p + 2 * p + n + n * n)
    return w * (4 * n * n + p * n + p * p + 2 * n + p)


def moved_bytes(kind, n, p, w=8):
    """Bytes the register kernels move per filter-step (packed symmetric / triangular storage)."""
    t, tp = tri(n), tri(p)
    if kind == "vanilla":       # x, P, F, H, Q, R, y -> x, P
        return w * (n + t + n * n + p * n + t + tp + p + n + t)
    if kind == "vanilla_full":  # + P-, K, innovation, yhat written
        return w * (n + t + n * n + p * n + t + tp + p + n + t + t + n * p + 2 * p)
    if kind == "vanilla_awgn":  # the Noiseless working set + the filter's 4-byte step lag (chol(Q) is formed in registers from Q)
        return w * (n + t + n * n + p * n + t + tp + p + n + t) + 4
    if kind == "squareroot":    # x, S, F, H, chol Q, chol R, y -> x, S
        return w * (n + t + n * n + p * n + t + tp + p + n + t)
    if kind == "information":   # i, I, F^-1, Q^-1 (packed), H, R^-1 (packed), y -> i, I
        return w * (n + t + n * n + t + p * n + tp + p + n + t)
    if kind == "hybrid":        # x, P, Phi, Htilde, R, real, computed -> x, P
        return w * (n + t + n * n + p * n + tp + 2 * p + n + t)
    if kind == "srif":          # b, R upper, Phi, Htilde, chol R, real, computed -> b, R upper (fused Update)
        return w * (n + t + n * n + p * n + tp + 2 * p + n + t)
    if kind == "srif_pair":     # two-lanes-per-filter Update: as "srif" plus the n / 2 stored zeros R[2 s + 1][2 s] the upper half reads with its rows
        return w * (n + t + n // 2 + n * n + p * n + tp + 2 * p + n + t)
    if kind == "srif_split":    # time kernel: b, R, Phi -> b, Rbar (full); meas kernel: b, Rbar, Htilde, L, y -> b, R upper... written full
        return w * ((n + n * n + n * n + n + n * n) + (n + n * n + p * n + tp + 2 * p + n + n * n))
    raise KeyError(kind)


def kernel_source_hash(root=None):
    """sha256 over what decides the machine code of the kernels: every file of gokalman_amd/csrc (*.hip, *.h), the C ABI header and
    the compiler flags of gokalman_amd/build.py.  The counter files under profiles/ carry the hash of the sources they were
    measured on; a bench line only quotes them while it still equals the hash of the sources it runs."""
    import glob
    import hashlib
    from . import build as kb_build
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "gokalman_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "gokalman_amd", "csrc", "*.h")) + glob.glob(os.path.join(root, "gokalman_amd", "csrc", "*.inc")) +
                   [os.path.join(root, "include", "gokalman_amd.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    h.update(repr((kb_build.ARCH, kb_build.FLAGS, sorted(kb_build.EXTRA.items()))).encode())
    return h.hexdigest()[:16]


def counters_current(root, doc):
    """(usable, note) for a counter document of profiles/: usable only if it was measured on these very kernel sources."""
    have = doc.get("source_hash")
    now = kernel_source_hash(root)
    if have == now:
        return True, {"source_hash": now, "matches_sources": True}
    return False, {"counter_file_source_hash": have, "source_hash": now, "matches_sources": False,
                   "note": "the kernel sources changed since the counters were collected: analytic byte / instruction counts are used instead"}


def load_traffic(root, kernel_substr):
    """Counter-measured bytes per launch for the kernel whose name contains `kernel_substr`, from profiles/traffic_latest.json --
    if that file was measured on the kernel sources of this checkout (kernel_source_hash); returns (bytes_per_filter, source
    dict), or (None, source dict saying why not) when the sources have changed, or (None, None) without a file."""
    path = os.path.join(root, "profiles", "traffic_latest.json")
    try:
        tj = json.load(open(path))
    except Exception:
        return None, None
    ok, note = counters_current(root, tj)
    if not ok:
        return None, dict(note, file="profiles/traffic_latest.json", profile_tag=tj.get("tag"), live=False,
                          analytic="packed working set, gokalman_amd/roofline.py moved_bytes()")
    entries = tj.get("kernels") or [tj]
    for e in entries:
        if kernel_substr in e.get("kernel", ""):
            per_filter = e["hbm_bytes_per_launch"] / float(e.get("filters", 1 << 20))
            src = dict(note, file="profiles/traffic_latest.json", profile_tag=e.get("tag"), head=e.get("head"), live=False,
                       counters="rocprofv3 --pmc FETCH_SIZE (x2, gfx950) and --pmc WRITE_SIZE, separate passes")
            return per_filter, src
    return None, None


def load_valu(root):
    """profiles/valu_latest.json (SQ_INSTS_VALU / SQ_WAVES per kernel) if it matches the kernel sources: (kernels dict, source dict);
    ({}, source dict with the reason) otherwise."""
    try:
        vj = json.load(open(os.path.join(root, "profiles", "valu_latest.json")))
    except Exception:
        return {}, None
    ok, note = counters_current(root, vj)
    src = dict(note, file="profiles/valu_latest.json", profile_tag=vj.get("tag"), counter="SQ_INSTS_VALU / SQ_WAVES", live=False)
    return (vj.get("kernels", {}) if ok else {}), src


def hbm_roofline(kernel_ms, filters, algo_bytes_per_filter, moved_bytes_per_filter, counter_bytes_per_filter=None,
                 traffic_source=None):
    """The `roofline` object of a bench line.  frac is physical (bytes moved / time / 8 TB/s) and <= 1."""
    s = kernel_ms * 1e-3
    phys = counter_bytes_per_filter if counter_bytes_per_filter is not None else moved_bytes_per_filter
    achieved = phys * filters / s / 1e9
    algo = algo_bytes_per_filter * filters / s / 1e9
    return {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
        "traffic": phys * filters,
        "traffic_source": traffic_source or {"live": False, "analytic": "packed working set, gokalman_amd/roofline.py moved_bytes()"},
        "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBPS, "achievable_GBps": HBM_ACHIEVABLE_GBPS,
        "achieved_algorithmic": algo, "frac_algorithmic": algo / HBM_PEAK_GBPS,
        "bytes_convention": {"algorithmic_bytes_per_filter_step": algo_bytes_per_filter,
                             "moved_bytes_per_filter_step": moved_bytes_per_filter,
                             "note": "algorithmic = SURVEY 8d full matrices; moved = packed upper triangles (what mat64.SymDense "
                                     "carries) actually read + written; frac uses the moved/counter bytes, so it is <= 1"},
        "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": algo_bytes_per_filter * filters,
    }


def valu_roofline(kernel_ms, tiles, valu_insts_per_tile, source):
    """Issue-rate roofline of a VALU-bound kernel: every wave64 VALU instruction of these kernels (fp64 FMA-class, or
    fp32 at one wave per SIMD) holds a SIMD's issue port for ISSUE_CYCLES cycles; floor = tiles x insts x 4 cycles /
    (1024 SIMDs x 2.4 GHz)."""
    floor_s = tiles * valu_insts_per_tile * ISSUE_CYCLES / (N_SIMD * CLOCK_HZ)
    return {"bound": "valu_issue", "floor_ms": floor_s * 1e3, "kernel_ms": kernel_ms, "frac": floor_s / (kernel_ms * 1e-3),
            "valu_insts_per_wave": valu_insts_per_tile, "issue_cycles_per_inst": ISSUE_CYCLES, "simds": N_SIMD,
            "clock_ghz": CLOCK_HZ / 1e9, "source": source}
