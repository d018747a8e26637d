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
EXTRA = {"kb_srif_reg.hip": ["-fno-slp-vectorize"], "kb_srif_pair32.hip": _PAIR, "kb_srif_pair64.hip": _PAIR, "kb_srif_pair32b.hip": _PAIR, "kb_srif_pair32c.hip": _PAIR, "kb_srif_pair32d.hip": _PAIR, "kb_srif_pair32e.hip": _PAIR, "kb_srif_pair32f.hip": _PAIR, "kb_srif_pair32g.hip": _PAIR, "kb_srif_pair32h.hip": _PAIR, 
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
        if deps and same_cmd and all(os.path.exists(d) and os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
            return obj
    t0 = time.perf_counter()
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, res.stderr[-4000:]))
    with open(cmdf + ".tmp", "w") as fh:   # (written behind the compile, through a rename: an interrupted build leaves no .cmd that vouches for a stale .o)
        fh.write(" ".join(cmd))
    os.replace(cmdf + ".tmp", cmdf)
    _TIMES[base] = time.perf_counter() - t0
    return obj


_TIMES = {}   # seconds per translation unit compiled by this process (printed by --times)

# Scheduling order of a build from scratch: longest translation unit first.  The seconds come from the previous build of this tree
# (csrc/_obj/times.json, written below); without one, from the object / source sizes.
def _recorded_times():
    try:
        import json
        with open(os.path.join(OBJ, "times.json")) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return {}


def _cost(src, rec):
    base = os.path.basename(src)
    obj = os.path.join(OBJ, base + ".o")
    if base in rec:
        return rec[base] * 1e6
    return os.path.getsize(obj) if os.path.exists(obj) else os.path.getsize(src)


STAMP = LIB + ".srchash"   # hash of (kernel sources, headers, flags, KB_EXTRA_DEFS) the library was linked from


def _source_stamp():
    from . import roofline as rl
    return rl.kernel_source_hash(os.path.dirname(HERE)) + "|" + os.environ.get("KB_EXTRA_DEFS", "")


def build(force=False, verbose=False):
    # The shipped library next to unchanged sources (a checkout that stamps every file alike, a GPU box that got the .so but not
    # csrc/_obj/): nothing to do -- judged by CONTENT, not by modification times.
    stamp = _source_stamp()
    if not force and os.path.exists(LIB):
        try:
            if open(STAMP).read() == stamp:
                if verbose:
                    print("built", LIB, "(up to date: source hash)")
                return LIB
        except OSError:
            pass
    os.makedirs(OBJ, exist_ok=True)
    rec = _recorded_times()
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")), key=lambda s_: _cost(s_, rec), reverse=True)   # longest first: the tail of the build is not one late giant
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if (force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)):
        # -s: no host symbol table (the C ABI is in .dynsym, the kernels' names in the code objects); kept out of diagnostic builds
        # (KB_EXTRA_DEFS / KB_DEBUG_SYMBOLS set), where host backtraces matter
        strip = [] if (os.environ.get("KB_EXTRA_DEFS") or os.environ.get("KB_DEBUG_SYMBOLS")) else ["-s"]
        cmd = [_hipcc(), "-shared", "-fPIC"] + strip + ["--offload-arch=" + ARCH, "-o", LIB] + objs
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("link failed:\n%s" % res.stderr[-4000:])
    if _TIMES:
        import json
        rec.update(_TIMES)
        with open(os.path.join(OBJ, "times.json"), "w") as fh:
            json.dump(rec, fh)
    with open(STAMP + ".tmp", "w") as fh:
        fh.write(stamp)
    os.replace(STAMP + ".tmp", STAMP)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    if "--times" in sys.argv:
        for name, sec in sorted(_TIMES.items(), key=lambda kv: -kv[1])[:16]:
            print("%6.1f s  %s" % (sec, name))
