#!/usr/bin/env python3
"""Per-basic-block instruction census of one k_frames variant, from the compiler's assembly (hipcc -S, labels kept).

    tools/isa_census.py [--log2n 10] [--variant 'ILi10ELb0ELi8E'] [--blocks] [--min 40]

Builds the device assembly of csrc/sp_inst_frames.hip for the size (the library's own flags), cuts the kernel into basic blocks
(labels, `; %bb.N` markers, and every branch ends one), classifies every instruction, and prints
  * the loop nest as the compiler annotates it,
  * one row per block with its class counts (blocks below --min instructions are summed per loop),
  * totals per loop depth.
The classes follow the PMC classes of profiles/*_valu.json (f64 / cvt / trans / f32 / int) and split their remainder ("other") into what
it is made of: v_mov / v_accvgpr moves, v_cndmask, compares, lane ops (permlane swaps, DPP, readfirstlane), and the v_readlane /
v_writelane of SGPR spills - the instructions VERDICT r4 asked to itemise.  Which blocks run how often per frame is not in the ISA:
profiles/r05_cfg2_census.txt combines this table with the loop structure of sp_kernel_frames.h.
"""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CLASSES = ["f64", "cvt", "trans", "f32", "int", "mov", "cndmask", "cmp", "swap", "dpp_rfl", "sgpr_spill", "valu_other", "lds", "vmem", "salu",
           "s_nop", "waitcnt", "branch"]


def classify(op, text):
    if op.startswith("v_"):
        if op in ("v_readlane_b32", "v_writelane_b32"):
            return "sgpr_spill"
        if op.startswith("v_permlane") or op.startswith("v_swap"):
            return "swap"
        if op == "v_readfirstlane_b32" or "dpp" in text or "row_" in text:
            return "dpp_rfl"
        if op.startswith("v_cmp") or op.startswith("v_cmpx"):
            return "cmp"
        if op.startswith("v_cndmask"):
            return "cndmask"
        if op.startswith("v_mov") or op.startswith("v_accvgpr"):
            return "mov"
        if op.startswith("v_cvt") or op.startswith("v_rndne") or op.startswith("v_fract") or op.startswith("v_floor") or op.startswith("v_trunc"):
            return "cvt"
        if re.match(r"v_(log|exp|rcp|rsq|sqrt|sin|cos)_", op):
            return "trans"
        if op.endswith("_f64") or "_f64_" in op:
            return "f64"
        if op.endswith("_f32") or "_f32_" in op or op.endswith("_f16"):
            return "f32"
        if re.search(r"_(u32|i32|b32|u16|i16|b16|u24|i24|b64|u64|i64)(_|$)", op) or op.startswith("v_bfe") or op.startswith("v_bfi") or \
                op.startswith("v_perm") or op.startswith("v_lshl") or op.startswith("v_and") or op.startswith("v_or") or \
                op.startswith("v_xor") or op.startswith("v_mad") or op.startswith("v_mul") or op.startswith("v_add") or \
                op.startswith("v_sub") or op.startswith("v_alignb") or op.startswith("v_not") or op.startswith("v_lshr") or \
                op.startswith("v_ashr") or op.startswith("v_min") or op.startswith("v_max") or op.startswith("v_med3"):
            return "int"
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op == "s_nop":
        return "s_nop"
    if op == "s_waitcnt":
        return "waitcnt"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_barrier", "s_setpc")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "valu_other"


def build_asm(log2n, out):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function",
           "-DSP_INST_FRAMES_LOG2N=%d" % log2n, "--offload-arch=gfx950", "--cuda-device-only", "-S", "-o", out,
           os.path.join(ROOT, "spectroplot-js_amd", "csrc", "sp_inst_frames.hip")]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)


def kernel_lines(path, variant):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN4spk28k_frames" + variant) and l.rstrip().split(";")[0].strip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start + 1:end]


def blocks_of(lines):
    blocks = []            # [name, depth, header_of, [ (op, text) ]]
    cur = {"name": "entry", "loop": None, "depth": 0, "ins": [], "hdr": False}
    blocks.append(cur)

    def new(name):
        nonlocal cur
        if not cur["ins"] and cur["name"].startswith("+"):
            cur["name"] = name
            return
        cur = {"name": name, "loop": None, "depth": 0, "ins": [], "hdr": False}
        blocks.append(cur)

    for l in lines:
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            new(m.group(1))
        elif s.startswith("; %bb."):
            new(s.split()[1].rstrip(":"))
        if "Loop Header: Depth=" in s:
            cur["depth"] = int(re.search(r"Depth=(\d+)", s).group(1))
            cur["loop"] = cur["name"]
            cur["hdr"] = True
            continue
        m = re.search(r"in Loop: Header=(\S+) Depth=(\d+)", s)
        if m:
            cur["loop"] = ".L" + m.group(1) if not m.group(1).startswith(".L") else m.group(1)
            cur["depth"] = int(m.group(2))
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        code = s.split(";")[0].strip()
        if not code:
            continue
        op = code.split()[0]
        cur["ins"].append((op, code))
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm")):
            new("+" + str(len(blocks)))
    return [b for b in blocks if b["ins"]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=10)
    ap.add_argument("--variant", default=None, help="mangled template arguments, e.g. ILi10ELb0ELi8E (default: <log2n, false, 8>)")
    ap.add_argument("--asm", default=None, help="an existing .s file instead of compiling")
    ap.add_argument("--min", type=int, default=30, help="list blocks with at least this many instructions")
    ap.add_argument("--dump", default=None, help="write the kernel's assembly here")
    a = ap.parse_args()
    variant = a.variant or "ILi%dELb0ELi8E" % a.log2n
    asm = a.asm or "/tmp/sp_frames_%d.s" % a.log2n
    if not a.asm:
        build_asm(a.log2n, asm)
    lines = kernel_lines(asm, variant)
    if a.dump:
        open(a.dump, "w").write("\n".join(lines) + "\n")
    blocks = blocks_of(lines)
    hdr = "%-14s %-12s %2s %5s | " % ("block", "loop", "d", "ins") + " ".join("%5s" % c[:5] for c in CLASSES)
    print("kernel k_frames<%s>: %d instructions in %d blocks" % (variant, sum(len(b["ins"]) for b in blocks), len(blocks)))
    print(hdr)
    per_loop = collections.defaultdict(collections.Counter)
    small = collections.defaultdict(collections.Counter)
    for b in blocks:
        c = collections.Counter(classify(op, text) for op, text in b["ins"])
        key = (b["loop"] or "-", b["depth"])
        per_loop[key].update(c)
        per_loop[key]["_ins"] += len(b["ins"])
        if len(b["ins"]) >= a.min:
            print("%-14s %-12s %2d %5d | " % (b["name"], b["loop"] or "-", b["depth"], len(b["ins"])) + " ".join("%5d" % c[k] for k in CLASSES))
        else:
            small[key].update(c)
            small[key]["_ins"] += len(b["ins"])
            small[key]["_blocks"] += 1
    print("\nblocks below %d instructions, summed per loop:" % a.min)
    for key, c in sorted(small.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        print("%-14s %-12s %2d %5d | " % ("(%d small)" % c["_blocks"], key[0], key[1], c["_ins"]) + " ".join("%5d" % c[k] for k in CLASSES))
    print("\ntotals per loop (static):")
    for key, c in sorted(per_loop.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        print("%-14s %-12s %2d %5d | " % ("", key[0], key[1], c["_ins"]) + " ".join("%5d" % c[k] for k in CLASSES))
    return 0


if __name__ == "__main__":
    sys.exit(main())
