#!/usr/bin/env python3
"""Build check for the LDS reads the kernels issue from inline asm (sp_frame_parts.h: exchange() reads with single ds_read_b64 instructions
and waits for them later, in exchange_wait()).  The compiler does not know that the destination registers of such a read are not valid
until that wait, so nothing but this check stops it from copying, spilling or re-using one of them in between (the register allocation
of one toolchain version is not a guarantee).  This script disassembles every kernel of the given objects and follows the in-order LDS
counter: every LDS instruction is an entry in a queue (with the VGPRs it will write, if it returns data), `s_waitcnt lgkmcnt(N)` retires
all but the youngest N entries, and any instruction that reads or writes a VGPR an outstanding entry will still write is an error.
    tools/check_lds_reads.py spectroplot-js_amd/build/frames_*.o        (exit status 1 and a listing on a violation)
"""
import os
import re
import subprocess
import sys
import tempfile



def _llvm_bin():
    """The LLVM tools that belong to the compiler in use: next to HIPCC's clang (HIPCC / ROCM_PATH as the Makefile reads them)."""
    import shutil
    cands = []
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
    cands += [os.path.join(root, "lib", "llvm", "bin"), os.path.join(root, "llvm", "bin")]
    if os.environ.get("ROCM_PATH"):
        cands.append(os.path.join(os.environ["ROCM_PATH"], "lib", "llvm", "bin"))
    cands.append("/opt/rocm/lib/llvm/bin")
    for c in cands:
        if os.path.exists(os.path.join(c, "llvm-objdump")) and os.path.exists(os.path.join(c, "clang-offload-bundler")):
            return c
    return None


LLVM = _llvm_bin()
ARCH = os.environ.get("ARCH", "gfx950")          # as the Makefile's ARCH
RETURNS = re.compile(r"^(ds_read|ds_.*_rtn|ds_bpermute|ds_permute|ds_swizzle|ds_consume|ds_append|ds_ordered_count|ds_condxchg)")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_listing(lines, where=""):
    """lines: disassembly of one or more functions (llvm-objdump -d).  Returns a list of violation strings."""
    bad = []
    pending = []          # outstanding lgkm operations, oldest first: (set of VGPRs still to be written, text)
    func = ""
    for ln in lines:
        m = re.match(r"^[0-9a-f]+ <(.*)>:", ln)
        if m:
            func, pending = m.group(1), []
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*(//.*)?$", ln)
        if not m:
            continue
        op, args = m.group(1), m.group(2)
        if op == "s_waitcnt":
            w = re.search(r"lgkmcnt\((\d+)\)", args)
            if w:
                n = int(w.group(1))
                pending = pending[len(pending) - n:] if n else []
            continue
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            pending = []      # the next instruction in address order is not this one's successor
            continue
        used = vregs(args)
        for dests, text in pending:
            hit = used & dests
            if hit:
                bad.append("%s %s: `%s %s` touches v%s while `%s` is outstanding" % (where, func[:60], op, args, sorted(hit), text))
        if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load") or op in ("s_memtime", "s_memrealtime"):
            dests = set()
            if op.startswith("ds_") and RETURNS.match(op):
                dests = vregs(args.split(",")[0])
            pending.append((dests, op + " " + args))
    return bad


def disassemble(obj):
    with tempfile.TemporaryDirectory() as t:
        fb, co = os.path.join(t, "fb.bin"), os.path.join(t, "k.co")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fb])
        if not os.path.exists(fb) or os.path.getsize(fb) == 0:
            return []                                   # an object without device code (host-only translation unit)
        listed = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", "--input=" + fb], capture_output=True, text=True)
        if ("amdgcn-amd-amdhsa--" + ARCH) not in listed.stdout:
            return []                                   # no code object for this architecture in it
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fb,
                               "--targets=hipv4-amdgcn-amd-amdhsa--" + ARCH, "--output=" + co], stderr=subprocess.DEVNULL)
        return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", co], text=True).split("\n")


def main(objs):
    import shutil
    if LLVM is None or shutil.which("objcopy") is None:
        # (a toolchain without the disassembler: nothing can be checked, and a build must not fail for that)
        print("check_lds_reads: llvm-objdump / clang-offload-bundler / objcopy not found, nothing checked (set HIPCC or ROCM_PATH)")
        return 0
    bad, kernels, reads, empty = [], 0, 0, []
    for o in objs:
        lines = disassemble(o)
        k = sum(1 for ln in lines if re.match(r"^[0-9a-f]+ <.*>:", ln))
        kernels += k
        # a frame-loop object without a single kernel for ARCH (wrong --offload-arch, a bundle this tool cannot read) would pass with
        # nothing checked: that is a failure of the gate, not a pass (host-only objects are not frames_*.o)
        if k == 0 and os.path.basename(o).startswith("frames_"):
            empty.append(os.path.basename(o))
        reads += sum(1 for ln in lines if re.search(r"\sds_read_b64\s", ln))
        bad += check_listing(lines, os.path.basename(o))
    print("check_lds_reads: %d objects, %d kernels, %d ds_read_b64, %d violations" % (len(objs), kernels, reads, len(bad)))
    for b in bad[:40]:
        print("  " + b)
    if empty:
        print("check_lds_reads: no %s kernels found in %s: nothing was checked there" % (ARCH, ", ".join(empty)))
    return 1 if bad or empty else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
