"""Builds libgokalman_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

`python -m gokalman_amd.build` or gokalman_amd.build.build().  Objects are compiled in
parallel (one hipcc per translation unit) and cached by source mtime under csrc/_obj/.
"""
import concurrent.futures
import glob
import os
import shutil
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libgokalman_amd.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function",
         "-ffp-contract=fast-honor-pragmas", "-fno-fast-math"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build the gfx950 kernels")


# per-file extra flags: the SLP vectoriser packs fp32 pairs in the fully unrolled SRIF panels and
# lengthens live ranges (more AGPR / scratch spills); it buys nothing there
# the two-lanes-per-filter SRIF kernel is ~10k instructions of straight-line code per variant: past LLVM's default budget for
# `#pragma unroll` (16k cost units) a loop silently stays rolled, its register arrays become scratch arrays
_PAIR = ["-fno-slp-vectorize", "-mllvm", "-pragma-unroll-threshold=200000"]
_UNROLL = ["-mllvm", "-pragma-unroll-threshold=200000"]
EXTRA = {"kb_srif_reg.hip": ["-fno-slp-vectorize"], "kb_srif_pair32.hip": _PAIR, "kb_srif_pair64.hip": _PAIR, "kb_srif_pair32b.hip": _PAIR, "kb_srif_pair32c.hip": _PAIR, "kb_srif_pair32d.hip": _PAIR, "kb_srif_pair32e.hip": _PAIR, "kb_srif_pair32f.hip": _PAIR, "kb_srif_pair32g.hip": _PAIR, 
         "kb_vanilla_split12.hip": _UNROLL, "kb_vanilla_split16.hip": _UNROLL, "kb_vanilla_split12p.hip": _UNROLL, "kb_vanilla_split12n.hip": _UNROLL, "kb_hybrid_split.hip": _UNROLL, "kb_hybrid_split8.hip": _UNROLL, "kb_vanilla_split16p.hip": _UNROLL, "kb_squareroot_split12.hip": _UNROLL, "kb_squareroot_split12p.hip": _UNROLL, "kb_squareroot_split16p.hip": _UNROLL, "kb_squareroot_split16.hip": _UNROLL, "kb_information_split12.hip": _UNROLL, "kb_information_split8.hip": _UNROLL, "kb_information_split12f.hip": _UNROLL,
         "kb_srif_split_a.hip": _UNROLL, "kb_srif_split_b.hip": _UNROLL, "kb_srif_split_c.hip": _UNROLL, "kb_srif_split_d.hip": _UNROLL, "kb_srif_split_e.hip": _UNROLL, "kb_srif_split_f32a.hip": _UNROLL, "kb_srif_split_f32b.hip": _UNROLL, "kb_srif_split_f32c.hip": _UNROLL, "kb_srif_split_f32d.hip": _UNROLL}


# file-name prefixes of the kernel families whose translation units MUST have an EXTRA entry
_NEEDS_EXTRA = ("kb_srif_pair", "kb_srif_split", "kb_vanilla_split", "kb_squareroot_split", "kb_information_split", "kb_hybrid_split")


def _read_deps(dep):
    """Prerequisites of a make-style dependency file (-MMD): the source and every header it includes."""
    try:
        txt = open(dep).read()
    except OSError:
        return None
    txt = txt.replace("\\\n", " ")
    return [t for t in txt.split(":", 1)[1].split() if t] if ":" in txt else None


def _compile(src, force):
    """One translation unit -> csrc/_obj/<name>.o, rebuilt when the source, a header IT includes (hipcc -MMD) or its
    command line changed (a header edit used to rebuild all 55 units: six minutes)."""
    base = os.path.basename(src)
    if base not in EXTRA and any(base.startswith(pfx) for pfx in _NEEDS_EXTRA):
        # a missed entry silently turns the register arrays of these kernels into scratch arrays (NOTES.md)
        raise RuntimeError("build.py: %s belongs to a kernel family that needs per-file flags (EXTRA) and has none" % base)
    obj = os.path.join(OBJ, base + ".o")
    dep, cmdf = obj + ".d", obj + ".cmd"
    cmd = [_hipcc()] + FLAGS + EXTRA.get(base, []) + os.environ.get("KB_EXTRA_DEFS", "").split() + ["-MMD", "-MF", dep, "-c", src, "-o", obj]
    if not force and os.path.exists(obj):
        deps = _read_deps(dep)
        try:
            same_cmd = open(cmdf).read() == " ".join(cmd)
        except OSError:
            same_cmd = False
        if deps and same_cmd and all(os.path.exists(d) and os.path.getmtime(d) < os.path.getmtime(obj) for d in deps):
            return obj
    t0 = time.perf_counter()
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, res.stderr[-4000:]))
    with open(cmdf, "w") as fh:
        fh.write(" ".join(cmd))
    _TIMES[base] = time.perf_counter() - t0
    return obj


_TIMES = {}   # seconds per translation unit compiled by this process (printed by --times)

# translation units that take the longest to compile (seconds on this image), for the scheduling order of a build from scratch
_SLOW = {"kb_srif_pair32g.hip": 82, "kb_srif_pair32c.hip": 66, "kb_srif_pair32e.hip": 63, "kb_srif_pair32f.hip": 59, "kb_srif_pair32b.hip": 48,
         "kb_srif_pair64.hip": 44, "kb_srif_split_e.hip": 44, "kb_srif_split_f32d.hip": 42, "kb_squareroot_split16p.hip": 41, "kb_srif_split_d.hip": 40,
         "kb_srif_split_f32c.hip": 35, "kb_srif_split_c.hip": 35, "kb_information_reg.hip": 34, "kb_srif_split_b.hip": 33, "kb_squareroot_split12p.hip": 33,
         "kb_vanilla_shared.hip": 30}


def _cost(src):
    base = os.path.basename(src)
    obj = os.path.join(OBJ, base + ".o")
    if base in _SLOW:
        return _SLOW[base] * 1e6
    return os.path.getsize(obj) if os.path.exists(obj) else os.path.getsize(src)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")), key=_cost, reverse=True)   # longest first: the tail of the build is not one late giant
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if (force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)):
        cmd = [_hipcc(), "-shared", "-fPIC", "-s", "--offload-arch=" + ARCH, "-o", LIB] + objs   # (-s: no host symbol table; the C ABI is in .dynsym, the kernels' names in the code objects)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("link failed:\n%s" % res.stderr[-4000:])
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    if "--times" in sys.argv:
        for name, sec in sorted(_TIMES.items(), key=lambda kv: -kv[1])[:16]:
            print("%6.1f s  %s" % (sec, name))
