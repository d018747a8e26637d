"""Per-kernel resource usage of the built library, read from the code object's metadata (the numbers the hardware
launches with): VGPRs, AGPRs, SGPRs, scratch (private segment) and LDS (group segment) bytes, and the waves per SIMD the
unified 512-entry register file then allows (MI355X_MICROARCH.md, "Register files": allocation granule 8).
usage: python scripts/resource_usage.py [--md out.md] [filter-substring ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gokalman_amd", "libgokalman_amd.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels():
    """Every translation unit's fat binary (csrc/_obj/*.o, what libgokalman_amd.so links) -> gfx950 code object -> metadata."""
    import glob
    notes, sym = "", ""
    objs = sorted(glob.glob(os.path.join(ROOT, "gokalman_amd", "csrc", "_obj", "*.o")))
    if not objs:
        raise SystemExit("no objects under gokalman_amd/csrc/_obj: run python -m gokalman_amd.build first")
    with tempfile.TemporaryDirectory() as td:
        for o in objs:
            fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
            if subprocess.call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, o], stderr=subprocess.DEVNULL):
                continue   # a translation unit without device code
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle", "--input=" + fat,
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], stderr=subprocess.DEVNULL)
            notes += subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
            sym += subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "-s", "--wide", co], text=True)
    sizes = {}
    for line in sym.splitlines():
        f = line.split()
        if len(f) >= 8 and f[3] == "FUNC":
            sizes[f[7]] = int(f[2])
    out = []
    for blk in notes.split("- .agpr_count:")[1:]:
        blk = ".agpr_count:" + blk
        get = lambda key: re.search(r"\." + key + r":\s+(\S+)", blk)
        name = get("name").group(1)
        d = {"symbol": name, "code_bytes": sizes.get(name, 0)}
        for key in ("agpr_count", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size"):
            m = get(key)
            d[key] = int(m.group(1)) if m else 0
        out.append(d)
    demangled = subprocess.run(["c++filt"], input="\n".join(k["symbol"] for k in out), capture_output=True, text=True).stdout.splitlines()
    for k, dn in zip(out, demangled):
        k["name"] = dn.replace("kb::", "").replace("(kb::StepArgs)", "")
        alloc = -(-max(k["vgpr_count"], 1) // 8) * 8     # vgpr_count already includes the AGPRs (unified file)
        k["waves_per_simd"] = min(8, 512 // alloc)
    return out


def main():
    args = [a for a in sys.argv[1:]]
    md = None
    if "--md" in args:
        md = args[args.index("--md") + 1]
        del args[args.index("--md"):args.index("--md") + 2]
    rows = [k for k in kernels() if not args or any(a in k["name"] for a in args)]
    rows.sort(key=lambda k: k["name"])
    lines = ["| kernel | VGPR+AGPR (of which AGPR) | SGPR | scratch B/lane | LDS B/workgroup | threads/wg | waves/SIMD (registers) | code B |",
             "|---|---|---|---|---|---|---|---|"]
    for k in rows:
        lines.append("| `%s` | %d (%d) | %d | %d | %d | %d | %d | %d |" % (k["name"][:110], k["vgpr_count"], k["agpr_count"], k["sgpr_count"],
                     k["private_segment_fixed_size"], k["group_segment_fixed_size"], k["max_flat_workgroup_size"], k["waves_per_simd"], k["code_bytes"]))
    text = "\n".join(lines)
    if md:
        open(md, "w").write("# Kernel resource usage (code-object metadata of gokalman_amd/libgokalman_amd.so, gfx950)\n\n"
                            "`python scripts/resource_usage.py`; vgpr_count is the unified VGPR+AGPR allocation request, waves/SIMD = "
                            "min(8, floor(512 / ceil8(vgpr_count))).\n\n" + text + "\n")
    print(text)


if __name__ == "__main__":
    main()
