"""Checks on the compiled gfx950 code objects (no GPU needed: hipcc cross-compiles, llvm-objdump disassembles).

The re-distributions of the frame loop read LDS with inline-asm ds_read_b64 instructions whose wait comes later, in a separate asm
statement (sp_frame_parts.h); the compiler does not know the destination registers are invalid in between.  tools/check_lds_reads.py
follows the in-order LDS counter through every kernel of the product and fails on any access to a register with an outstanding read."""
import glob
import importlib.util
import os

import pytest

from __graft_entry__ import ROOT, build


def _checker():
    spec = importlib.util.spec_from_file_location("check_lds_reads", os.path.join(ROOT, "tools", "check_lds_reads.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_checker_sees_a_register_touched_before_its_read_has_returned():
    c = _checker()
    ok = """
0000000000001000 <k>:
	ds_write_b64 v10, v[2:3]                                   // 000000001000: D89A0000 0000020A
	ds_read_b64 v[2:3], v11 offset:8                           // 000000001008: D8EC0008 0200000B
	ds_read_b64 v[4:5], v11 offset:16                          // 000000001010: D8EC0010 0400000B
	v_add_f64 v[6:7], v[8:9], v[8:9]                           // 000000001018: D2800006 00021108
	s_waitcnt lgkmcnt(1)                                       // 000000001020: BF8CC17F
	v_mul_f64 v[2:3], v[2:3], v[8:9]                           // 000000001024: D2810002 00021102
	s_waitcnt lgkmcnt(0)                                       // 00000000102C: BF8CC07F
	v_mul_f64 v[4:5], v[4:5], v[8:9]                           // 000000001030: D2810004 00021104
	s_endpgm                                                   // 000000001038: BF810000
""".split("\n")
    assert c.check_listing(ok) == []
    copied_early = [ln.replace("v_add_f64 v[6:7], v[8:9], v[8:9]", "v_mov_b32_e32 v20, v4") for ln in ok]        # a copy before the wait
    assert len(c.check_listing(copied_early)) == 1
    clobbered = [ln.replace("v_add_f64 v[6:7], v[8:9], v[8:9]", "v_add_f64 v[2:3], v[8:9], v[8:9]") for ln in ok]   # a write under the read
    assert len(c.check_listing(clobbered)) == 1
    short_wait = [ln.replace("s_waitcnt lgkmcnt(1)", "s_waitcnt lgkmcnt(2)") for ln in ok]                           # waits for nothing
    assert len(c.check_listing(short_wait)) == 1


def test_no_kernel_of_the_product_touches_a_register_with_an_outstanding_lds_read():
    objs = sorted(glob.glob(os.path.join(ROOT, "spectroplot-js_amd", "build", "*.o")))
    if len(objs) < 9:
        build()
        objs = sorted(glob.glob(os.path.join(ROOT, "spectroplot-js_amd", "build", "*.o")))
    objs = [o for o in objs if not o.endswith("sp_host.o")]
    assert len(objs) >= 9
    c = _checker()
    bad, reads = [], 0
    for o in objs:
        lines = c.disassemble(o)
        reads += sum(1 for ln in lines if " ds_read_b64 " in ln.replace("\t", " "))
        bad += c.check_listing(lines, os.path.basename(o))
    assert reads > 5000            # the asm reads are there (16 per component and re-distribution in every variant)
    assert not bad, bad[:10]


def test_the_prefetching_iq_variants_of_the_frame_loop_use_no_scratch_memory():
    """A kernel that touches scratch (spilled VGPRs, or the argument structure dropped to the stack because a lambda was not inlined: it
    happened in an experiment and cost 40 %) launches its waves slower and reads its arguments from memory.  Every variant the BASELINE
    configs run - interleaved I/Q, samples prefetched (PFB != 0) - must have a private segment of zero bytes and no spilled VGPR."""
    import re
    import subprocess
    import tempfile
    objs = sorted(glob.glob(os.path.join(ROOT, "spectroplot-js_amd", "build", "frames_*.o")))
    if len(objs) < 8:
        build()
        objs = sorted(glob.glob(os.path.join(ROOT, "spectroplot-js_amd", "build", "frames_*.o")))
    assert len(objs) == 8
    llvm = "/opt/rocm/lib/llvm/bin"
    seen = 0
    with tempfile.TemporaryDirectory() as t:
        for o in objs:
            subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", o, os.path.join(t, "fb.bin")])
            subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + os.path.join(t, "fb.bin"),
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + os.path.join(t, "k.co")],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            notes = subprocess.check_output([os.path.join(llvm, "llvm-readelf"), "--notes", os.path.join(t, "k.co")], text=True)
            for blk in notes.split(".name:")[1:]:
                m = re.match(r"\s*_ZN4spk28k_framesILi(\d+)ELb([01])ELi(\d+)E", blk)
                if not m or m.group(2) == "1" or m.group(3) == "0":
                    continue
                seen += 1
                priv = int(re.search(r"\.private_segment_fixed_size:\s*(\d+)", blk).group(1))
                spill = int(re.search(r"\.vgpr_spill_count:\s*(\d+)", blk).group(1))
                assert priv == 0 and spill == 0, (m.group(0), priv, spill)
    assert seen == 8 * 5           # eight sizes x five prefetch widths


def test_the_prologue_waits_for_its_tables_not_for_the_first_frames_samples():
    """n <= 1024, prefetching variants: the first frame's 16 sample loads are issued BEHIND the table loads and the wait in front of the
    LDS table stores counts them (s_waitcnt vmcnt(16)).  Vector-memory operations complete in order: a vmcnt(0) there - the samples
    requested ahead of the tables, or a request the compiler cannot count - makes the table wait a wait for HBM (+1.5 us per launch)."""
    import re
    objs = sorted(glob.glob(os.path.join(ROOT, "spectroplot-js_amd", "build", "frames_*.o")))
    if len(objs) < 8:
        build()
        objs = sorted(glob.glob(os.path.join(ROOT, "spectroplot-js_amd", "build", "frames_*.o")))
    c = _checker()
    seen = 0
    for o in objs:
        m = re.search(r"frames_(\d+)\.o$", o)
        if int(m.group(1)) > 10:
            continue
        kernel, waits = None, []
        for ln in c.disassemble(o):
            h = re.match(r"[0-9a-f]+ <_ZN4spk28k_framesILi(\d+)ELb([01])ELi(\d+)E", ln)
            if h:
                kernel, waits = h.groups(), []
                continue
            if kernel is None:
                continue
            t = ln.replace("\t", " ")
            w = re.search(r" s_waitcnt .*vmcnt\((\d+)\)", t)
            if w:
                waits.append(int(w.group(1)))
            if " s_barrier" in t:                      # the prologue's barrier: everything before it is the prologue
                if kernel[2] != "0" and kernel[1] == "0":
                    assert waits and waits[-1] == 16 and 0 not in waits, (kernel, waits)
                    seen += 1
                kernel = None
    assert seen >= 5 * 5          # five prefetch widths at n = 64 .. 1024


def test_the_instrumentation_patch_applies_to_the_kernel_sources():
    """tools/experiments/frames_instrumentation.patch (per-wave clocks for tools/stamps.py, cost-attribution switches, the round-5 tile
    prototype) lives outside the product sources and is applied to a copy by tools/build_variant.sh: a kernel change that moves its
    context must refresh it, or the next round finds its measuring tools broken."""
    import shutil
    import subprocess
    import tempfile
    if shutil.which("patch") is None:
        pytest.skip("no patch(1) here")
    with tempfile.TemporaryDirectory() as t:
        os.makedirs(os.path.join(t, "spectroplot-js_amd"))
        shutil.copytree(os.path.join(ROOT, "spectroplot-js_amd", "csrc"), os.path.join(t, "spectroplot-js_amd", "csrc"))
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(t, "include"))
        with open(os.path.join(ROOT, "tools", "experiments", "frames_instrumentation.patch")) as fh:
            r = subprocess.run(["patch", "-p1", "--dry-run", "-s"], cwd=t, stdin=fh, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-500:]


def _regs(operand):
    """VGPR numbers an operand names: 'v7' -> {7}, 'v[4:7]' -> {4..7}, anything else -> empty."""
    import re
    m = re.fullmatch(r"v(\d+)", operand)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def store_hazards(lines, where=""):
    """gfx940 / gfx950: a VMEM store of more than 64 bits reads its data registers late - a VALU instruction that overwrites them needs 2
    wait states behind the store (LLVM's GCNHazardRecognizer inserts them for stores it knows; k_frames issues its write-out stores as
    inline asm, which it cannot see: ADVICE r5).  Returns the stores followed within 2 wait states by a VALU write to their data."""
    import re
    ins = []
    for ln in lines:
        t = ln.replace("\t", " ").split("//")[0].strip()
        if not t or t.endswith(":") or re.match(r"[0-9a-f]+ <", t):
            continue
        ins.append(t)
    bad = []
    for i, t in enumerate(ins):
        m = re.match(r"global_store_dwordx[34] (\S+), (v\[\d+:\d+\])", t)
        if not m:
            continue
        data = _regs(m.group(2).rstrip(","))
        states, j = 0, i + 1
        while states < 2 and j < len(ins):
            u = ins[j]
            nop = re.match(r"s_nop (\d+)", u)
            if nop:
                states += int(nop.group(1)) + 1
                j += 1
                continue
            if u.startswith("v_"):
                ops = [o.strip().rstrip(",") for o in u.split(None, 1)[1].split(",")] if " " in u else []
                dests = ops[:2] if ("_swap" in u.split()[0]) else ops[:1]
                written = set().union(*[_regs(o) for o in dests]) if dests else set()
                if written & data:
                    bad.append("%s: `%s` then `%s`" % (where, t, u))
            states += 1
            j += 1
    return bad


def test_store_hazard_checker_sees_a_data_register_overwritten_too_early():
    ok = ["\tglobal_store_dwordx4 v1, v[4:7], s[2:3] nt   // 0000: X", "\tv_mov_b32_e32 v9, v4  // 0008", "\ts_nop 0  // 000c",
          "\tv_mov_b32_e32 v4, v9  // 0010"]
    assert store_hazards(ok) == []
    assert len(store_hazards([ok[0], "\tv_mov_b32_e32 v5, v9 // x"])) == 1
    assert len(store_hazards([ok[0], ok[1], "\tv_add_f64 v[6:7], v[8:9], v[8:9] // x"])) == 1
    assert store_hazards([ok[0], "\ts_nop 1 // x", "\tv_mov_b32_e32 v5, v9 // x"]) == []
    assert len(store_hazards([ok[0], "\tv_permlane32_swap_b32_e32 v20, v7 // x"])) == 1


def test_no_valu_write_lands_on_the_data_of_a_wide_store_within_two_wait_states():
    objs = sorted(glob.glob(os.path.join(ROOT, "spectroplot-js_amd", "build", "*.o")))
    if len(objs) < 9:
        build()
        objs = sorted(glob.glob(os.path.join(ROOT, "spectroplot-js_amd", "build", "*.o")))
    objs = [o for o in objs if not o.endswith("sp_host.o")]
    c = _checker()
    bad, stores = [], 0
    for o in objs:
        lines = c.disassemble(o)
        stores += sum(1 for ln in lines if "global_store_dwordx4" in ln)
        bad += store_hazards(lines, os.path.basename(o))
    assert stores > 500            # the write-out's 16-byte stores are there, in every variant
    assert not bad, bad[:10]
