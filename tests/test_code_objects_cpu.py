"""Reads the gfx950 code objects of the built library (scripts/resource_usage.py: the metadata the hardware launches with) and fails when
a kernel documented as scratch-free spills, or loses the waves per SIMD DESIGN.md quotes.  A missed per-file flag
in gokalman_amd/build.py (EXTRA: the unroll threshold) turns register arrays into scratch arrays without a diagnostic, a few more live
values push an allocation over 256 registers: both show here, on the CPU, before any GPU minute is spent."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

# (regular expression on the demangled kernel name, max scratch bytes per lane, min waves per SIMD by registers, what it is)
RULES = [
    (r"^void vanilla_reg_kernel<double, 6, 3, 0, false, false, false, false, false, false>", 0, 2, "headline: Vanilla 6/3 per step"),
    (r"^void vanilla_reg_kernel<double, 6, 3, 0, true, false, false, false, false, false>", 0, 2, "Vanilla 6/3 FULL"),
    (r"^void vanilla_reg_kernel<double, 6, 3, 0, false, false, false, false, true, false>", 0, 2, "Vanilla 6/3 + AWGN per step"),
    (r"^void vanilla_reg_kernel<double, 6, 3, 0, false, false, true, false, false, false>", 0, 1, "Vanilla 6/3 time-fused"),
    (r"^void vanilla_reg_kernel<double, 6, 3, 0, false, false, true, false, true, false>", 32, 1, "Vanilla 6/3 + AWGN time-fused (28 B: the draws at the 512-register cap)"),
    (r"^void squareroot_reg_kernel<double, 6, 3, 0, false, false, false, false, false>", 0, 2, "config C: SquareRoot 6/3 per step"),
    (r"^void squareroot_reg_kernel<double, 6, 3, 0, false, false, false, false, true>", 0, 1, "SquareRoot 6/3 time-fused"),
    (r"^void information_reg_kernel<double, 6, 3, 0, false, false, false, false, false>", 0, 2, "Information 6/3"),
    (r"^void hybrid_reg_kernel<double, 6, 2, false, false", 0, 2, "config D(ii): Hybrid 6/2"),
    (r"^void hybrid_fused_kernel<double, 6, (1|2|3), (true|false)>", 0, 2, "config D(ii) time-fused (round 6)"),
    (r"^void srif_pair_kernel<float, 12, 6, false, (true|false), false>", 0, 2, "config E: SRIF 12/6 fp32"),
    (r"^void srif_pair_kernel<double, 12, 6, ", 0, 1, "SRIF 12/6 fp64, two lanes"),
    (r"^void srif_pair_fused_kernel<float, 12, 6>", 48, 2, "config E time-fused (round 6; 40 B: the loop-carried panel at the 256-register cap)"),
    (r"^void vanilla_split_kernel<double, 12, 6, 0, 4, false, (true|false), (true|false), false, false, 0, false>", 0, 2, "Vanilla 12/6 exact, four lanes"),
    (r"^void vanilla_split_kernel<double, 12, 6, 0, 4, false, (true|false), false, false, false, (1|2), false>", 32, 2, "Vanilla 12/6 exact + AWGN / BatchNoise (round 6; 16-24 B: the noise vectors on top of the exact kernel's 256 registers)"),
    (r"^void vanilla_split_kernel<double, 12, 8, 0, 4, false, ", 0, 2, "Vanilla 12/8 exact, four lanes, S^-1 once per filter"),
    (r"^void vanilla_split_kernel<double, 16, 8, 0, 8, false, ", 0, 2, "Vanilla 16/8 exact, eight lanes, S^-1 once per filter"),
    (r"^void vanilla_split_kernel<double, 12, 8, 2, 4, true, false, false, false, false, 0, false>", 32, 2, "Vanilla n <= 12, p = 7, 8 padded (round 4: one wave per SIMD, 340 registers)"),
    (r"^void vanilla_split_kernel<double, 16, 8, 2, 8, true, false, false, false, false, 0, false>", 16, 2, "Vanilla n <= 16, p = 7, 8 padded (round 4: 192 B)"),
    (r"^void squareroot_split_kernel<double, 12, 6, 0, 4, false, ", 0, 2, "SquareRoot 12/6 exact"),
    (r"^void information_split_kernel<double, 12, 6, 0, 4, false, ", 0, 2, "Information 12/6 exact"),
    (r"^void srif_split_kernel<double, \d+, (4|6), (4|8)>", 0, 2, "SRIF fp64 split, p <= 6 (every n)"),
    (r"^void srif_split_kernel<double, \d+, 8, (4|8)>", 16, 2, "SRIF fp64 split, p = 7, 8"),
    (r"^void srif_split_kernel<float, ([1-9]|11|15), (4|8), 4>", 0, 2, "SRIF fp32 split, odd n and n < 6"),
    (r"^void srif_split_kernel<float, (13|14|16), (4|8), 4>", 64, 2, "SRIF fp32 split at 13 / 14 / 16 states (16 filters per wave: the elimination rows fill the register file)"),
    (r"^void mc_kernel<double, 4, 2, ", 0, 2, "config D(i): Monte-Carlo statOD5044"),
]


@pytest.fixture(scope="module")
def kernels():
    import resource_usage
    return resource_usage.kernels()


def test_documented_kernels_do_not_spill_and_keep_their_occupancy(kernels):
    bad, seen = [], {}
    for k in kernels:
        for pat, max_scratch, min_waves, what in RULES:
            if re.search(pat, k["name"]):
                seen[pat] = seen.get(pat, 0) + 1
                if k["private_segment_fixed_size"] > max_scratch or k["waves_per_simd"] < min_waves:
                    bad.append((what, k["name"][:100], "scratch %d B (<= %d)" % (k["private_segment_fixed_size"], max_scratch),
                                "waves/SIMD %d (>= %d)" % (k["waves_per_simd"], min_waves), "VGPR+AGPR %d" % k["vgpr_count"]))
    missing = [what for pat, _, _, what in RULES if pat not in seen]
    assert not missing, "rules that match no kernel (renamed template parameters?): %s" % missing
    assert not bad, "\n".join(str(b) for b in bad)


def test_every_translation_unit_of_a_flagged_family_has_its_flags():
    """build.py refuses to compile a kb_*_split* / kb_srif_pair* unit without an EXTRA entry; the table and the tree agree."""
    import glob
    from gokalman_amd import build as kb_build
    srcs = [os.path.basename(p) for p in glob.glob(os.path.join(kb_build.CSRC, "*.hip"))]
    flagged = [s for s in srcs if any(s.startswith(p) for p in kb_build._NEEDS_EXTRA)]
    assert flagged and all(s in kb_build.EXTRA for s in flagged), [s for s in flagged if s not in kb_build.EXTRA]
    stale = [s for s in kb_build.EXTRA if s not in srcs]
    assert not stale, "EXTRA names files that no longer exist: %s" % stale


def _disassemble(obj_name):
    """{mangled kernel symbol: {mnemonic: count}} of one translation unit's gfx950 code object."""
    import collections
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    obj = os.path.join(ROOT, "gokalman_amd", "csrc", "_obj", obj_name)
    out, cur = {}, None
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.check_call([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj])
        subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--type=o", "--unbundle", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], stderr=subprocess.DEVNULL)
        txt = subprocess.check_output([os.path.join(llvm, "llvm-objdump"), "-d", "--no-show-raw-insn", co], text=True)
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = out.setdefault(m.group(1), collections.Counter())
        elif cur is not None and line.startswith("\t"):
            cur[line.split()[0]] += 1
    return out


def test_paired_lds_slots_are_read_with_16_byte_loads():
    """Round 5: the split kernels keep two consecutive elements per lane in LDS so that contiguous runs are read with ds_read_b128 -- the LDS array
    serves the ds_read2_b64 pairs the compiler forms from 8-byte slots at half the bytes per clock (profiles/NOTES.md).  The property lives in index
    expressions and an alignment attribute: losing either silently brings the pairs back.  Instruction counts of the exact 12/6 kernels."""
    want = {"kb_vanilla_split12.hip.o": ("_ZN2kb20vanilla_split_kernelIdLi12ELi6ELi0ELi4ELb0ELb0ELb0ELb0ELb0ELi0ELb0EEEvNS_8StepArgsE", 180, 120),
            "kb_squareroot_split12.hip.o": ("_ZN2kb23squareroot_split_kernelIdLi12ELi6ELi0ELi4ELb0ELb0ELb0ELb0EEEvNS_8StepArgsE", 200, 60),
            "kb_information_split12.hip.o": ("_ZN2kb24information_split_kernelIdLi12ELi6ELi0ELi4ELb0ELb0EEEvNS_8StepArgsE", 150, 120)}
    for obj, (sym, min_b128, max_read2) in want.items():
        ks = _disassemble(obj)
        assert sym in ks, (obj, [k for k in ks if "split_kernel" in k][:4])
        c = ks[sym]
        assert c["ds_read_b128"] >= min_b128 and c["ds_read2_b64"] <= max_read2, (obj, c["ds_read_b128"], c["ds_read2_b64"], c["ds_read_b64"])
