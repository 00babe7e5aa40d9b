#!/usr/bin/env python3
"""bench.py — throughput of the I/Q STFT -> RGBA hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg1|cfg3|cfg4|cfg5] [--no-cpu-baseline] [--rotate K]

A "step" is one pass of the hot path (sp_plan_execute: ONE kernel, the frame loop, which also produces the side outputs) over one
capture of synthetic I/Q that is already resident in HBM, producing the RGBA image, both histograms, the gauges and the dBfs range
in HBM.  Default workload = BASELINE.json configs[1]: 16 MSample (2^24) cf32, n = 1024, Blackman-Harris, viridis,
gain 6, range 30, spectrogram layout, W = S/n = 16384 frames.  With N > 1 (launched by torch.distributed.run, one rank
per GPU) every rank renders its own contiguous time slice of an N-times longer capture (weak scaling, the reference's
own slice scheme, lib/spectroplot.js:1206-1228) and the per-slice histograms / dBfs range are combined on the device
(an RCCL all-gather of the records per batch of renders + sp_merge_replies) inside the timed region; the RGBA strips stay resident on their GPUs (see DESIGN.md "Multi-GPU") and the
gather of the strips to rank 0 is timed separately and reported as extra fields.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GEN = dict(seed=0x5EED0001, step=7321, gshift=11, amp=0.5, namp=0.02)   # trinoise, tests/siggen.py

CONFIGS = {
    # name: format, log2(samples per GPU), n, window, cmap, frames (None = S/n), description
    "cfg1": ("CU8", 20, 512, "hann", "cube1", None, "1 MSample cu8, N=512, Hann, Cube1"),
    "cfg2": ("CF32", 24, 1024, "blackmanHarris", "viridis", None, "16 MSample cf32, N=1024, Blackman-Harris, Viridis"),
    "cfg3": ("CS16", 28, 2048, "hann", "cube1", None, "256 MSample cs16, N=2048, Hann, Cube1"),
    "cfg4": ("CU8", 28, 1024, "blackmanHarris", "cube1", None, "256 MSample cu8 slice (1/8 of 2 GSample), N=1024, Blackman-Harris, Cube1"),
    # (not a BASELINE config: config 5's shape with 4-byte samples)
    "cfg5cs16": ("CS16", 26, 8192, "blackmanHarris", "cube1", (1 << 26) // 8192 * 8, "64 MSample cs16, N=8192, zoom x8, Blackman-Harris, Cube1"),
    "cfg5": ("CS12", 26, 8192, "blackmanHarris", "cube1", (1 << 26) // 8192 * 8, "64 MSample cs12, N=8192, zoom x8, Blackman-Harris, Cube1"),
    # not BASELINE configs: the 8-byte formats at the larger sizes (tuning runs of those kernel variants)
    "cf32_2048": ("CF32", 26, 2048, "blackmanHarris", "viridis", None, "64 MSample cf32, N=2048, Blackman-Harris, Viridis (not a BASELINE config)"),
    "cf32_4096": ("CF32", 26, 4096, "blackmanHarris", "viridis", None, "64 MSample cf32, N=4096, Blackman-Harris, Viridis (not a BASELINE config)"),
    # config 5 without its zoom (hop = n, every sample read once): calibrates the HBM read counter for the 3-byte loader
    "cs12_8192": ("CS12", 26, 8192, "blackmanHarris", "cube1", None, "64 MSample cs12, N=8192, no zoom (not a BASELINE config)"),
}
SAMPLE_WIDTH = {"CU8": 2, "CF32": 8, "CS16": 4, "CS12": 3}


def load_cmap(name):
    import numpy as np
    idx = json.load(open(os.path.join(ROOT, "tests", "golden", "cmaps.json")))
    blob = np.fromfile(os.path.join(ROOT, "tests", "golden", "cmaps.bin"), dtype=np.uint8)
    e = [x for x in idx if x["name"] == name + "_cmap"][0]
    lut = blob[e["offset"]:e["offset"] + 3 * e["length"]].reshape(-1, 3).copy()
    lut[0] = (0, 0, 0)            # the caller's end forcing, lib/spectroplot.js:1129-1130
    lut[-1] = (255, 255, 255)
    return lut


def host_cpu_model():
    """The host CPU's model string and socket count (BASELINE.md section 3 asks for the CPU the baseline ran on)."""
    try:
        names, sockets = [], set()
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                names.append(line.split(":", 1)[1].strip())
            elif line.startswith("physical id"):
                sockets.add(line.split(":", 1)[1].strip())
        if names:
            return "%s x %d socket(s)" % (names[0], max(len(sockets), 1))
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine() or "unknown"


def cpu_baseline(fmt, n, window, seconds=8.0):
    """Times the JavaScript oracle (bit-exact restatement of the reference worker, incl. its fft_nayuki radix-2 loop) under
    Node on this host on a bounded sample of the same workload, two legs: one thread, and one
    worker_thread per host core as the reference starts one Web Worker per navigator.hardwareConcurrency
    (lib/spectroplot.js:87).  `value` is the all-cores figure; the one-thread figure is reported beside it."""
    script = os.path.join(ROOT, "oracle", "js", "cpu_baseline.js")
    cores = os.cpu_count() or 1
    model = host_cpu_model()

    def leg(log2_s, threads):
        out = subprocess.run(["node", script, fmt, str(log2_s), str(n), window, str(seconds), str(threads)], capture_output=True, text=True,
                             timeout=600)
        return json.loads(out.stdout.strip().splitlines()[-1])

    try:
        one = leg(22, 1)
        res = {"value": one["frames_per_s"], "unit": "frames/s", "cores": 1, "kind": "port", "cpu_model": model, "node": one["node"],
               "sample": "oracle/js/worker_oracle.js under node %s on %s: 2^22 samples of the same synthetic %s signal, n=%d, %s, %d renders in "
                         "%.1f s on 1 thread of %d host cores" % (one["node"], model, fmt, n, window, one["reps"], one["seconds"], cores),
               "msamples_per_s": one["msamples_per_s"], "one_thread": {"frames_per_s": one["frames_per_s"], "msamples_per_s": one["msamples_per_s"]}}
    except Exception as e:  # the baseline is reported, never required
        return {"value": None, "unit": "frames/s", "cores": 1, "kind": "port", "cpu_model": model, "sample": "failed: %r" % (e,)}
    try:
        if cores > 1:
            allc = leg(20, cores)
            res.update({"value": allc["frames_per_s"], "cores": cores, "msamples_per_s": allc["msamples_per_s"],
                        "sample": res["sample"] + "; all-cores leg: %d worker_threads, each rendering 2^20 samples of the signal for %.1f s"
                                  % (cores, allc["seconds"])})
    except Exception as e:
        res["all_cores_failed"] = repr(e)
    return res


def e2e_dropin():
    """The message path a caller of the reference sees: postMessage -> onmessage through HipWorker (N-API, host buffers, PCIe both
    ways), tools/js_dropin_bench.js under Node.  Reported beside the HBM-resident figure, never as `value`."""
    try:
        out = subprocess.run(["node", os.path.join(ROOT, "tools", "js_dropin_bench.js"), "--json"], capture_output=True, text=True, timeout=600)
        rows = json.loads(out.stdout.strip().splitlines()[-1])
        pick = {r["name"]: r for r in rows}
        a, b = pick["config 2"], pick["config 2, request buffer page-locked"]
        return {"workload": "one config-2 worker message (16 MSample cf32 in, 64 MiB RGBA out) through HipWorker under Node",
                "ms_per_message": a["ms_per_message"], "msamples_per_s": a["msamples_per_s"], "first_message_ms": a.get("first_message_ms"),
                "ms_per_message_pinned_request": b["ms_per_message"], "msamples_per_s_pinned_request": b["msamples_per_s"],
                # mean over `messages` back-to-back messages (V8's collector pauses included) and their median
                "messages": a.get("messages"), "median_ms_per_message": a.get("median_ms_per_message"),
                "median_ms_per_message_pinned_request": b.get("median_ms_per_message"),
                "config1_ms_per_message": pick["config 1"]["ms_per_message"], "config1_js_worker_ms": pick["config 1"]["js_worker_ms"],
                # the longer direction of a config-2 message (128 MiB in, 64 MiB out, full duplex) against the host link's 63 GB/s
                "pcie_frac": (128 * 2**20) / (b["ms_per_message"] * 1e-3) / 63e9,
                "pcie_frac_note": "134 MB of samples per message / ms_per_message_pinned_request / 63 GB/s (PCIe Gen5 x16, one direction)",
                # the same capture at a screen-wide image (2048 frames, stride ~ 8 n: lib/worker.js:50, 70-75): only the frames' samples travel
                "sparse_ms_per_message": pick["sparse"]["ms_per_message"], "sparse_median_ms_per_message": pick["sparse"].get("median_ms_per_message"),
                "sparse_ms_per_message_pinned_request": pick["sparse, request buffer page-locked"]["ms_per_message"],
                "sparse_images_identical_to_js_worker": pick["sparse"]["images_identical"],
                "sparse_note": "16 MSample cf32, n = 1024, width 2048: the frames' own 16 MiB (+ the pitched rows' widening) cross the link "
                               "instead of the capture's 128 MiB; SPECTROPLOT_HIP_NO_PACKED_UPLOAD=1 restores the contiguous upload"}
    except Exception as e:
        return {"failed": repr(e)}


PROFILE_TAGS = ("r06", "r05", "r04", "r03")   # profiles/<tag>_<config>_{traffic,valu}.json: committed rocprofv3 --pmc summaries of this same command (newest first)


def rocprof_kernel_us(argv_config):
    """Average duration of the frame-loop kernel by rocprofv3 --kernel-trace --stats over a short child run of this same command (the
    figure the committed profiles/<tag>_<config>_kernel_stats.csv holds).  Started before this process touches the GPU.  None when
    rocprofv3 is not there or anything about the child run fails: the roofline then uses the un-corrected event-pair time."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    d = tempfile.mkdtemp(prefix="sp_kt_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
               "--steps", "400", "--warmup", "100", "--no-cpu-baseline", "--no-e2e", "--no-rocprof", "--no-extras"] + argv_config
        # (its own process group: on a timeout the profiler AND the python below it go, nothing keeps the GPU busy behind our back)
        child_out = os.path.join(d, "child_stdout.txt")
        with open(child_out, "w") as fh:
            child = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=fh, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                child.wait(timeout=600)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(child.pid, signal.SIGKILL)
                child.wait()
                return None
        # the child's own step time (same process as the kernel average: `kernel <= step` can be checked without crossing processes)
        child_ms = None
        try:
            for line in open(child_out):
                if line.startswith("{") and '"ms_per_step"' in line:
                    j = json.loads(line)
                    child_ms = j.get("ms_per_step_steady", j["ms_per_step"])   # (the launches the average is over are mostly steady ones)
        except Exception:
            child_ms = None
        best = None
        for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if any(k in row["Name"] for k in ("k_frames", "k_scratch_radix2")):
                    t = float(row["TotalDurationNs"])
                    if best is None or t > best[0]:
                        best = (t, float(row["AverageNs"]) / 1e3, int(row["Calls"]), row["Name"].split("(")[0])
        return None if best is None else {"avg_us": best[1], "calls": best[2], "kernel": best[3], "child_ms_per_step": child_ms}
    except Exception:
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def pmc_counter_mean(directory, counter):
    """Mean of one counter over the second half of the frame-loop kernel's dispatches in rocprofv3's counter_collection CSVs under
    `directory` (one row per dispatch and counter: Kernel_Name, Counter_Name, Counter_Value), None with fewer than 8 dispatches."""
    import csv
    import glob
    vals = []
    for f in sorted(glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] == counter and any(k in row["Kernel_Name"] for k in ("k_frames", "k_scratch_radix2")):
                    vals.append(float(row["Counter_Value"]))
    if len(vals) < 8:
        return None
    half = vals[len(vals) // 2:]
    return sum(half) / len(half)


def rocprof_hbm_bytes(argv_config):
    """HBM bytes per launch of the frame-loop kernel from the PMC counters, as MI355X_MICROARCH.md's HBM section prescribes: FETCH_SIZE and
    WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (they do not fit one pass; no trace option next to --pmc), both in KiB, mean over the
    second half of the kernel's launches of a short child run of this command; gfx950's FETCH_SIZE tallies the 128-byte requests of wide
    coalesced streaming reads at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte streaming stores.  None if rocprofv3 is
    missing or a pass fails (the committed figure of profiles/ is reported then, labelled as such)."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="sp_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "40", "--warmup", "1000", "--no-cpu-baseline", "--no-e2e", "--no-rocprof", "--no-extras"] + argv_config
            child = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                child.wait(timeout=300)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(child.pid, signal.SIGKILL)
                child.wait()
                return None
            got[counter] = pmc_counter_mean(d, counter)
            if got[counter] is None:
                return None
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"FETCH_SIZE_KB": got["FETCH_SIZE"], "WRITE_SIZE_KB": got["WRITE_SIZE"],
            "hbm_bytes_per_launch": (2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"]) * 1024.0}


def valu_roofline(config, kernel_us, frames):
    """The resource that binds the frame loop in fact: VALU issue.  Counted wave-instructions per launch (rocprofv3
    --pmc, committed under profiles/) times the measured issue cost of their class with two waves per SIMD (profiles/r05_op_cost.txt),
    over the chip's 1024 SIMDs at 2.4 GHz: the time the launch would take if no SIMD ever waited."""
    f, tag = None, None
    for tag in PROFILE_TAGS:
        f = os.path.join(ROOT, "profiles", "%s_%s_valu.json" % (tag, config))
        if os.path.exists(f):
            break
    else:
        return None
    c = json.load(open(f))["counters"]
    if not c.get("SQ_INSTS_VALU"):
        return None
    g = lambda k: c.get(k) or 0.0  # noqa: E731
    f64 = g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_FMA_F64")
    cvt, trans, i32 = g("SQ_INSTS_VALU_CVT"), g("SQ_INSTS_VALU_TRANS_F32"), g("SQ_INSTS_VALU_INT32") + g("SQ_INSTS_VALU_INT64")
    f32 = g("SQ_INSTS_VALU_FMA_F32") + g("SQ_INSTS_VALU_MUL_F32") + g("SQ_INSTS_VALU_ADD_F32")
    other = max(g("SQ_INSTS_VALU") - f64 - cvt - trans - i32 - f32, 0.0)
    # cycles per wave-instruction per SIMD with two waves per SIMD (profiles/r05_op_cost.txt).  Round 5 measured more instructions than
    # rounds 2-4 had: nearly EVERY VALU instruction costs 4.6-4.9 - shifts, v_add_lshl_u32, v_lshl_or_b32, v_med3, SDWA forms, conversions
    # and the f64 operations alike; only v_add_u32 / v_mul_f32 (2.5) and v_fma_f32 (3.5) are cheaper, v_log_f32 (8.8) and the permlane
    # swaps (8.1) dearer.  The integer and f32 classes are mixes of those (the PMC classes do not separate them); "other" holds the
    # swaps (64 of ~290 per 1024-point frame).
    cost = {"f64": 4.7, "cvt": 4.8, "trans_f32": 8.8, "int": 4.0, "f32": 3.6, "other": 5.0}
    cycles = f64 * cost["f64"] + cvt * cost["cvt"] + trans * cost["trans_f32"] + i32 * cost["int"] + f32 * cost["f32"] + other * cost["other"]
    floor_us = cycles / 1024.0 / 2400.0
    return {"bound": "f64 VALU issue", "f64_wave_insts": f64, "f64_wave_insts_per_frame_wave": f64 / max(frames, 1), "fused_f64_insts": g("SQ_INSTS_VALU_FMA_F64"),
            "valu_wave_insts": g("SQ_INSTS_VALU"), "cycles_per_inst": cost["f64"], "class_costs": cost,
            "class_counts": {"f64": f64, "cvt": cvt, "trans_f32": trans, "int": i32, "f32": f32, "other": other},
            "fused_f64_note": "v_fma_f64 belongs to the IEEE divisions of the side outputs' software log10 and gauge scaling (a few per group of "
                              "frames); the transform's multiplies and adds are all separate instructions",
            "floor_us": floor_us, "kernel_us": kernel_us, "frac": floor_us / kernel_us if kernel_us else None,
            "source": "profiles/%s_%s_valu.json (rocprofv3 --pmc of this command); issue costs: profiles/r05_op_cost.txt" % (tag, config)}


def launch_ranks(argv, gpus):
    """`python bench.py --gpus N` started plainly: one child rank per GPU through torch.distributed.run, before this process has
    touched the GPU; the children's output (rank 0 prints the JSON line) passes through."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=None,
                    help="untimed steps before the timed ones (default: the clock spin-up of the configuration + 200; `value` is ALWAYS "
                         "--steps steps right after --warmup steps of the fresh process, nothing in between)")
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", action="store_true", help="(default when N > 1) also time the RCCL gather of the RGBA strips to rank 0")
    ap.add_argument("--no-gather", action="store_true", help="skip the strip gather")
    ap.add_argument("--no-e2e", action="store_true", help="skip the Node drop-in leg")
    ap.add_argument("--no-rocprof", action="store_true", help="skip the rocprofv3 child run that times the dominant kernel (N = 1)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to exercise the N>1 path on one GPU)")
    ap.add_argument("--oversubscribe", action="store_true", help="diagnostic: map ranks onto the available GPUs modulo their count")
    ap.add_argument("--force-dist", action="store_true",
                    help="diagnostic: run the collectives of the N > 1 path with N = 1 too (one-rank RCCL group: exercises the nccl code path on a one-GPU box)")
    ap.add_argument("--waterfall", action="store_true", help="diagnostic: waterfall layout instead of spectrogram")
    ap.add_argument("--no-rgba", action="store_true", help="diagnostic: skip the image output (INVALID as a benchmark)")
    ap.add_argument("--merge-every", type=int, default=64,
                    help="N > 1: the side-output records of this many renders travel in ONE all-gather and are merged in ONE launch "
                         "(1 = one collective per render; default 64: what a collective costs the frame loops - its kernel needs CUs that "
                         "the frame loop fills completely - is paid once per batch)")
    ap.add_argument("--merge-per-render", action="store_true",
                    help="N > 1, diagnostic: one sp_merge_replies launch per render of a batch (rounds 1-5) instead of one sp_merge_replies_batch per batch")
    ap.add_argument("--kernel", default="auto", choices=["auto", "scratch", "frames"], help="A/B runs: force a device kernel")
    ap.add_argument("--rotate", type=int, default=3,
                    help="N = 1: also time the kernel over this many capture / image sets in rotation, a working set beyond the 256 MiB "
                         "Infinity Cache (reported as roofline.rotating; 0 or 1 = skip)")
    ap.add_argument("--rotate-all", action="store_true",
                    help="diagnostic: EVERY step of the run (spin-up, warm-up, timed region, the rocprofv3 child) cycles --rotate capture / "
                         "image sets instead of re-rendering one: the whole line is then measured on a working set beyond the Infinity Cache")
    ap.add_argument("--channel-mode", action="store_true", help="the reference's L/R split (channelMode: lib/fft_nayuki.js:103-119); not a BASELINE config")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child runs that measure the kernel's HBM traffic (N = 1)")
    ap.add_argument("--no-extras", action="store_true", help="skip the rotating-buffers and two-requests-in-flight legs (N = 1)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(sys.argv[1:], args.gpus))

    # dominant kernel's duration as rocprofv3 reports it: a short child run of this command, before this process touches the GPU
    prof = prof_rot = pmc = None
    if args.gpus == 1 and not args.no_rocprof and "WORLD_SIZE" not in os.environ:
        shape = ["--config", args.config, "--kernel", args.kernel] + (["--waterfall"] if args.waterfall else []) + (["--no-rgba"] if args.no_rgba else []) + (["--channel-mode"] if args.channel_mode else [])
        prof = rocprof_kernel_us(shape + (["--rotate-all", "--rotate", str(args.rotate)] if args.rotate_all else []))
        if args.rotate >= 2 and not args.rotate_all and not args.no_extras:
            # the same kernel over K capture / image sets in rotation, the whole child run: is the default figure HBM bandwidth? (below)
            prof_rot = rocprof_kernel_us(shape + ["--rotate-all", "--rotate", str(args.rotate)])
        if not args.no_pmc:
            pmc = rocprof_hbm_bytes(shape)                   # two more short child runs: the kernel's HBM traffic, live

    import numpy as np
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    import importlib
    sharding = importlib.import_module("spectroplot_js_amd.sharding")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: the two must agree" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    if args.oversubscribe:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:                   # --force-dist started plainly: a one-rank group of its own
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sock.getsockname()[1]))
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    fmt, lg, n, window, cmap, frames, desc = CONFIGS[args.config]
    S = 1 << lg
    sw = SAMPLE_WIDTH[fmt]
    W = frames if frames else S // n
    dev = torch.device("cuda", local_rank)

    ctx = pkg.Context(local_rank)
    # one explicit stream for torch fills, the library's kernels and the collectives (the null stream's handle is 0, which
    # sp_context_set_stream reads as "use your own stream")
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    win, weight = pkg.window(window, n)
    lut = load_cmap(cmap)
    plan = ctx.plan(fmt, n, win, 1.0 / weight, 6.0, 30.0, lut, channel_mode=args.channel_mode, waterfall=args.waterfall)
    if args.kernel != "auto":
        plan.force_kernel(args.kernel)

    # operands resident in HBM
    d_in = torch.empty(S * sw, dtype=torch.uint8, device=dev)
    ctx.synth_trinoise(d_in.data_ptr(), fmt, rank * S, S, GEN["seed"], GEN["step"], GEN["gshift"], GEN["amp"], GEN["namp"])
    rgba = torch.empty(4 * W * n, dtype=torch.uint8, device=dev)
    rgba_ptr = 0 if args.no_rgba else rgba.data_ptr()
    gauges = torch.empty(3 * W, dtype=torch.uint8, device=dev)
    # side outputs of one slice as one record [c_hist | cB_hist | dBfs_min, dBfs_max as f64 bits]: the library writes straight
    # into it.  The records of M = --merge-every renders form one batch; two batches in rotation: the all-gather of a batch (RCCL's
    # stream) overlaps the frame loops of the next one.  (The frame-loop kernel fills every CU - all of its VGPRs and 145 KiB of its LDS -
    # so a collective's kernel waits for a CU and then holds up workgroups of the following launch: one collective per 80 us render
    # would put that on every step; M renders per collective leave the exchange itself as it is and divide the interference by M.)
    L = len(lut)
    P = L + 1000 + 2
    M = max(1, args.merge_every) if dist is not None else 1

    def merge_on_device(by_rank, count, out):
        # the caller's merge of the slices' side outputs (lib/spectroplot.js:1229-1238), on device: one render's records, every rank's
        p = out.data_ptr()
        ctx.merge_replies(by_rank.data_ptr(), count, L, p, p + 8 * L, p + 8 * (L + 1000))

    # (spectroplot-js_amd/sharding.py RecordBatcher: the batching, the all-gather per batch and the merge one batch later; its shapes
    # and views are exercised on CPU tensors with gloo by tests/test_sharding_gloo.py)
    def merge_batch_on_device(gathered, ranks, renders, merged_all):
        # ... a whole batch of renders in ONE launch ([rank][render][record] as the all-gather left it -> [render][record])
        ctx.merge_replies_batch(gathered.data_ptr(), ranks, renders, L, merged_all.data_ptr())

    batcher = sharding.RecordBatcher(world, M, P, dev, merge_fn=merge_on_device, collectives=dist is not None,
                                     merge_batch_fn=None if args.merge_per_render else merge_batch_on_device)
    records = batcher.records

    rot_sets, rot_k = None, [0]
    if args.rotate_all and args.rotate >= 2:
        rot_sets = [(d_in, rgba)] + [(torch.empty_like(d_in), torch.empty_like(rgba)) for _ in range(args.rotate - 1)]
        for k in range(1, args.rotate):
            ctx.synth_trinoise(rot_sets[k][0].data_ptr(), fmt, rank * S, S, GEN["seed"], GEN["step"], GEN["gshift"], GEN["amp"], GEN["namp"])

    def run_slice(rec, j=0):
        p = rec.data_ptr() + 8 * P * j
        src, dst = d_in, rgba
        if rot_sets is not None:
            src, dst = rot_sets[rot_k[0] % len(rot_sets)]
            rot_k[0] += 1
        plan.execute(src.data_ptr(), S * sw, W, 0 if args.no_rgba else dst.data_ptr(), gauges.data_ptr(), gauges.data_ptr() + W,
                     gauges.data_ptr() + 2 * W, p, p + 8 * L, p + 8 * (L + 1000))

    def step():
        run_slice(batcher.next_record())
        batcher.rendered()

    def finish_pending():
        batcher.finish()

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # `value` = --steps steps right after --warmup steps of this fresh process: nothing else sits between the two.  An idle MI355X runs
    # its first few hundred launches ~5-15 % slower while the shader clock ramps, so the DEFAULT warm-up is that
    # spin-up (~0.3 s at config 2, scaled down for the larger configs) + 200; a caller who passes a short --warmup gets the figure that
    # warm-up gives (the driver's `--steps 20 --warmup 5` times the ramp) and the steady figure beside it as `value_steady`.
    spinup = max(50, int(3000 * 16384 / max(W, 16384) * 1024 / n) if n <= 1024 else 200)
    warmup_given = args.warmup is not None
    if not warmup_given:
        args.warmup = spinup + 200
    # the warm-up steps, the first ones of them on the clock (reported as `cold`, never the headline): the very first step of this
    # process (kernel code upload, cold caches, idle clocks) and up to 20 steps right after it
    done = 0
    cold = {"first_step_ms": None, "next_steps": 0, "ms_per_step_next": None}
    torch.cuda.synchronize()
    if args.warmup >= 1:
        tc = time.perf_counter()
        step()
        finish_pending()
        torch.cuda.synchronize()
        cold["first_step_ms"] = (time.perf_counter() - tc) * 1e3
        done = 1
    k = min(20, args.warmup - done)
    if k > 0:
        tc = time.perf_counter()
        for _ in range(k):
            step()
        finish_pending()
        torch.cuda.synchronize()
        cold.update({"next_steps": k, "ms_per_step_next": (time.perf_counter() - tc) * 1e3 / k})
        done += k
    for _ in range(args.warmup - done):
        step()
    finish_pending()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    finish_pending()                                          # the last render's merge belongs to the timed region
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the steady figure (an extra): where the warm-up was shorter than the spin-up, the same --steps again after the rest of it
    dt_steady, steady_after, steady_leg = dt, args.warmup, False
    if args.warmup < spinup:
        steady_leg = True
        for _ in range(spinup + 200 - args.warmup):
            step()
        finish_pending()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        finish_pending()
        sync()
        dt_steady = time.perf_counter() - t0
        steady_after = spinup + 200 + args.steps
        if dist is not None:
            t = torch.tensor([dt_steady], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_steady = float(t.item())
    final = batcher.final_record()
    hsum = int(final[:L].sum().item())                        # all slices after the merge

    # dominant kernel: live HIP-event timing of the frame-loop kernel on its stream, outside the timed region
    ctx.enable_timing(True)
    kms = []
    for _ in range(40):
        # a short back-to-back batch, as in the timed region; the events bracket the batch's last frame-loop kernel
        for _ in range(max(2, min(args.steps, 8))):
            run_slice(records[0])
        torch.cuda.synchronize()
        kms.append(ctx.last_kernel_ms())
    ctx.enable_timing(False)
    kernel_ms_events = float(np.mean(kms))
    # An event pair adds the dispatch latency of the packet it brackets (measured here on a one-wavefront no-op kernel);
    # rocprofv3's kernel duration (profiles/) does not contain it.  The roofline uses the kernel's own duration.
    event_overhead_ms = ctx.event_pair_overhead_ms()
    # roofline: the kernel's duration as rocprofv3 --kernel-trace reports it (the child run above); without rocprofv3, the event-pair
    # time as measured, un-corrected (it contains the dispatch latency: a lower bound on the fraction)
    kernel_ms = prof["avg_us"] * 1e-3 if prof else kernel_ms_events

    rotating = two_in_flight = None
    if world == 1 and dist is None and not args.no_extras:
        # (a) Is the default figure HBM bandwidth?  Config 2's working set (128 MiB in + 64 MiB out) fits the 256 MiB Infinity Cache and
        # every step re-renders the same buffers.  The same kernel over K capture / image sets in rotation touches K times that.
        K = args.rotate
        if K >= 2:
            try:
                ins = [d_in] + [torch.empty_like(d_in) for _ in range(K - 1)]
                outs = [rgba] + [torch.empty_like(rgba) for _ in range(K - 1)]
                for k in range(1, K):
                    ctx.synth_trinoise(ins[k].data_ptr(), fmt, k * S, S, GEN["seed"], GEN["step"], GEN["gshift"], GEN["amp"], GEN["namp"])
                rec = records[0]

                def run_set(k):
                    plan.execute(ins[k].data_ptr(), S * sw, W, 0 if args.no_rgba else outs[k].data_ptr(), gauges.data_ptr(), gauges.data_ptr() + W,
                                 gauges.data_ptr() + 2 * W, rec.data_ptr(), rec.data_ptr() + 8 * L, rec.data_ptr() + 8 * (L + 1000))
                nrot = max(40, min(args.steps, 400))
                for i in range(2 * K):
                    run_set(i % K)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(nrot):
                    run_set(i % K)
                torch.cuda.synchronize()
                ms_rot = (time.perf_counter() - t0) * 1e3 / nrot
                t0 = time.perf_counter()
                for i in range(nrot):
                    run_set(0)
                torch.cuda.synchronize()
                ms_one = (time.perf_counter() - t0) * 1e3 / nrot
                ctx.enable_timing(True)
                ev = []
                for i in range(40):
                    for j in range(4):
                        run_set((4 * i + j) % K)
                    torch.cuda.synchronize()
                    ev.append(ctx.last_kernel_ms())
                ctx.enable_timing(False)
                ev_rot = float(np.mean(ev))
                # the kernel's duration over the rotating sets: rocprofv3's average over a child run that rotates from its first step to its
                # last (the event pair of this process brackets single launches and is too coarse for a 5 % question)
                ratio = (prof_rot["avg_us"] / prof["avg_us"]) if (prof and prof_rot) else ev_rot / kernel_ms_events
                step_ratio = ms_rot / ms_one
                rotating = {"sets": K, "working_set_MiB": K * (S * sw + rgba.numel()) / 2**20, "ms_per_step_rotating": ms_rot,
                            "value_rotating": W / (ms_rot * 1e-3),
                            "ms_per_step_one_set_same_loop": ms_one, "kernel_ms_event_pair_rotating": ev_rot,
                            "kernel_ms_event_pair_one_set": kernel_ms_events, "ratio": ratio, "step_ratio": step_ratio,
                            "kernel_ms_rotating": prof_rot["avg_us"] * 1e-3 if prof_rot else kernel_ms * ratio,
                            "kernel_ms_rotating_source": ("rocprofv3 --kernel-trace --stats, child run with --rotate-all: average of %d launches" % prof_rot["calls"])
                                                         if prof_rot else "event pairs of this process (ratio to the one-set event pair)",
                            "verdict": ("kernel duration " + ("within 2 %" if abs(ratio - 1) <= 0.02 else "differs by more than 2 %")
                                        + " (ratio %.3f); back-to-back steps %.1f %% %s over the rotating sets" % (ratio, abs(step_ratio - 1) * 100,
                                                                                                             "slower" if step_ratio > 1 else "faster")
                                        + ("" if step_ratio <= 1.02 and ratio <= 1.02 else
                                           ": the default figure leans on the Infinity Cache (the one-set image's stores never have to reach HBM); "
                                           "`python bench.py --rotate-all` measures the whole line on the rotating sets"))}
                del ins, outs
            except RuntimeError as e:          # (out of device memory on a small card: reported, not fatal)
                rotating = {"failed": repr(e)}
        # (b) Two requests in flight on one GPU, as the reference's pool keeps several workers busy: two contexts on two streams, each
        # with its own plan and outputs.  A launch's workgroups occupy whole CUs, so the two do not share CUs, but the tail of one launch
        # (CUs that have finished) overlaps the start of the next.  An extra figure, never the headline.
        try:
            ctx2 = pkg.Context(local_rank)
            plan2 = ctx2.plan(fmt, n, win, 1.0 / weight, 6.0, 30.0, lut, channel_mode=args.channel_mode, waterfall=args.waterfall)
            if args.kernel != "auto":
                plan2.force_kernel(args.kernel)
            rgba2 = torch.empty_like(rgba)
            gauges2 = torch.empty_like(gauges)
            rec2 = torch.zeros(P, dtype=torch.int64, device=dev)
            ctx.set_stream(0)                      # each context on its own (non-blocking) stream

            def both():
                plan.execute(d_in.data_ptr(), S * sw, W, rgba_ptr, gauges.data_ptr(), gauges.data_ptr() + W, gauges.data_ptr() + 2 * W,
                             records[0].data_ptr(), records[0].data_ptr() + 8 * L, records[0].data_ptr() + 8 * (L + 1000))
                plan2.execute(d_in.data_ptr(), S * sw, W, 0 if args.no_rgba else rgba2.data_ptr(), gauges2.data_ptr(), gauges2.data_ptr() + W,
                              gauges2.data_ptr() + 2 * W, rec2.data_ptr(), rec2.data_ptr() + 8 * L, rec2.data_ptr() + 8 * (L + 1000))
            npair = max(20, min(args.steps, 400) // 2)
            for _ in range(10):
                both()
            ctx.synchronize()
            ctx2.synchronize()
            t0 = time.perf_counter()
            for _ in range(npair):
                both()
            ctx.synchronize()
            ctx2.synchronize()
            dt2 = time.perf_counter() - t0
            ok2 = bool(torch.equal(rec2[:L], records[0][:L])) and (args.no_rgba or bool(torch.equal(rgba2, rgba)))
            two_in_flight = {"value_two_in_flight": 2 * npair * W / dt2, "unit": "frames/s", "ms_per_request": dt2 / (2 * npair) * 1e3,
                             "requests": 2 * npair, "outputs_equal": ok2,
                             "note": "two contexts / streams / plans on one GPU rendering the same capture into separate outputs, back to back"}
            plan2.close()
            ctx2.close()
            ctx.set_stream(stream.cuda_stream)
        except Exception as e:
            two_in_flight = {"failed": repr(e)}

    gather_ms = place_ms = merged_ok = None
    if dist is not None and not args.no_gather:
        # the caller's putImageData of every slice (lib/spectroplot.js:1241-1244), HBM to HBM: the strips are gathered into ONE device
        # buffer on rank 0 (rank order) and placed in the merged image by sp_place_strips; both steps are timed, outside the timed region
        per = rgba.numel()
        sync()
        g0 = time.perf_counter()
        allstrips = sharding.gather_to_one_buffer(rgba, dst=0)
        sync()
        gather_ms = (time.perf_counter() - g0) * 1e3
        if rank == 0:
            image = torch.zeros(world * per, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            p0 = time.perf_counter()
            ctx.place_strips(image.data_ptr(), allstrips.data_ptr(), world, n, world * W, W, args.waterfall)
            torch.cuda.synchronize()
            place_ms = (time.perf_counter() - p0) * 1e3
            # check: rank 0 renders every slice itself (the generator is seekable) and composes the image with torch indexing
            want = torch.empty_like(image)
            wv = want.view(world * W, n, 4) if args.waterfall else want.view(n, world * W, 4)
            one = torch.empty_like(rgba)
            tmp_in = torch.empty_like(d_in)
            for r in range(world):
                ctx.synth_trinoise(tmp_in.data_ptr(), fmt, r * S, S, GEN["seed"], GEN["step"], GEN["gshift"], GEN["amp"], GEN["namp"])
                plan.execute(tmp_in.data_ptr(), S * sw, W, one.data_ptr(), gauges.data_ptr(), gauges.data_ptr() + W, gauges.data_ptr() + 2 * W,
                             records[0].data_ptr(), records[0].data_ptr() + 8 * L, records[0].data_ptr() + 8 * (L + 1000))
                torch.cuda.synchronize()
                if args.waterfall:
                    wv[world * W - W - r * W:world * W - r * W] = one.view(W, n, 4)
                else:
                    wv[:, r * W:(r + 1) * W] = one.view(n, W, 4)
            merged_ok = bool(torch.equal(image, want))
            del want, one, tmp_in, image, allstrips

    ranks_seen = None
    if dist is not None:
        # who took part: rank, device index and PCI bus id per rank, so that a scaling run shows N distinct GPUs
        props = torch.cuda.get_device_properties(local_rank)
        mine = {"rank": rank, "device": int(torch.cuda.current_device()), "name": props.name,
                "pci_bus_id": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", -1) & 0xff,
                                                  getattr(props, "pci_device_id", -1) & 0xff),
                "uuid": str(getattr(props, "uuid", ""))}
        seen = [None] * world
        dist.all_gather_object(seen, mine)
        ranks_seen = {"world_size": dist.get_world_size(), "backend": args.backend, "ranks": seen,
                      "distinct_devices": len({(r["pci_bus_id"], r["uuid"]) for r in seen})}

    stride_eff = min((S - n) / (W - 1), n)
    bytes_per_frame = sw * stride_eff + 4 * n + 3                 # unique input bytes + RGBA + 3 gauge bytes
    algo_bytes = bytes_per_frame * W
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
    frames_per_s = world * W * args.steps / dt

    traffic, traffic_source = None, None
    if pmc:
        traffic = pmc["hbm_bytes_per_launch"]
        traffic_source = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two child runs of this command in this run: %.1f KiB x 2 (gfx950: wide "
                          "streaming reads are tallied at half their bytes) + %.1f KiB" % (pmc["FETCH_SIZE_KB"], pmc["WRITE_SIZE_KB"]))
    for tag in (PROFILE_TAGS + ("r02",)) if traffic is None else ():
        tfile = os.path.join(ROOT, "profiles", "%s_%s_traffic.json" % (tag, args.config))
        if os.path.exists(tfile):   # HBM bytes per launch from separate rocprofv3 --pmc passes of this same command (committed, not live)
            traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
            traffic_source = "profiles/%s_%s_traffic.json (committed rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of this command; not measured in this run)" % (tag, args.config)
            break

    # The roofline is HBM: where the one-set loop leans on the Infinity Cache (config 2's 192 MiB fit its 256 MiB; the rotating leg's
    # verdict says so), the fraction to quote is the one over sets that stream through HBM; the one-set figure stays beside it.
    frac_one_set = achieved / 8000.0
    achieved_rot = (algo_bytes / (rotating["kernel_ms_rotating"] * 1e-3) / 1e9) if rotating and rotating.get("kernel_ms_rotating") else None
    # (only a SLOWER rotating leg switches the headline: a rotating leg that measures faster is noise, not a reason to quote the higher figure)
    leans = bool(achieved_rot) and (rotating["ratio"] > 1.02 or rotating["step_ratio"] > 1.02) and not rot_sets
    achieved_head = achieved_rot if leans else achieved
    if rank == 0:
        out = {
            "metric": "STFT frames/sec (N=1024 cf32) + IQ MSamples/s end-to-end to RGBA" if args.config == "cfg2"
                      else "STFT frames/sec (%s)" % desc,
            "value": frames_per_s, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "cold": cold,
            "value_is": "--steps steps right after --warmup steps of a fresh process (%s --warmup %d), nothing in between"
                        % ("the caller's" if warmup_given else "default", args.warmup),
            "value_steady": world * W * args.steps / dt_steady, "ms_per_step_steady": dt_steady / args.steps * 1e3,
            "value_steady_note": ("the same --steps steps again once the process had run %d steps (the clock spin-up of this configuration: "
                                  "%d steps + 200): the shader clock no longer ramps" % (steady_after, spinup)) if steady_leg
                                 else "the warm-up covered the clock spin-up of this configuration (%d steps): value_steady is value" % spinup,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc + ", gain 6, range 30, spectrogram layout, %d frames per GPU" % W + (", L/R split (channelMode)" if args.channel_mode else "")
                                   + (", %d capture / image sets in rotation" % len(rot_sets) if rot_sets else ""), "format": fmt, "n": n,
                       "samples_per_gpu": S, "frames_per_gpu": W, "window": window, "cmap": cmap,
                       "sharding": "contiguous time slice per GPU" if world > 1 else "single GPU",
                       "butterfly_graph": "the reference's radix-2 DIT graph (lib/fft_nayuki.js:54-96) in f64 without fused multiply-adds, executed "
                                          "as 16-point register passes with LDS / register re-distribution between them (not a Stockham "
                                          "re-factorisation: any other factorisation changes the rounding and with it RGBA bytes)",
                       "generator": "trinoise seed=0x%08X step=%d gshift=%d amp=%g namp=%g" % (GEN["seed"], GEN["step"], GEN["gshift"], GEN["amp"], GEN["namp"])},
            "msamples_per_s": frames_per_s * stride_eff / 1e6,
            "kernel": plan.kernel_name(),
            "roofline": {"bound": "hbm", "achieved": achieved_head, "peak": 8000.0, "unit": "GB/s", "frac": achieved_head / 8000.0, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "frac_basis": ("kernel duration over %d capture / image sets in rotation (%.0f MiB: streams through HBM); the one-set loop "
                                        "leans on the Infinity Cache" % (rotating["sets"], rotating["working_set_MiB"])) if leans
                                       else "kernel duration of the timed loop (its working set does not fit the Infinity Cache, or the rotating leg measured no difference)",
                         "frac_one_set": frac_one_set, "achieved_one_set": achieved,
                         "kernel_ms": kernel_ms,
                         "child_run_ms_per_step": prof.get("child_ms_per_step") if prof else None,
                         "kernel_ms_source": ("rocprofv3 --kernel-trace --stats, child run of this command: average of %d launches of %s"
                                              % (prof["calls"], prof["kernel"])) if prof else "HIP event pair around the kernel, un-corrected",
                         "kernel_ms_event_pair": kernel_ms_events, "event_pair_overhead_ms": event_overhead_ms,
                         "kernel_ms_event_pair_minus_overhead": max(kernel_ms_events - event_overhead_ms, 0.0),
                         "algorithmic_bytes_per_launch": algo_bytes, "bytes_per_frame": bytes_per_frame,
                         "frac_of_copy_ceiling_6290": achieved / 6290.0, "rotating": rotating,
                         "frac_rotating": (algo_bytes / (rotating["kernel_ms_rotating"] * 1e-3) / 1e9 / 8000.0)
                                          if rotating and rotating.get("kernel_ms_rotating") else None},
            "roofline_valu": valu_roofline(args.config, kernel_ms * 1e3, W) if world == 1 else None,
            "value_two_in_flight": two_in_flight.get("value_two_in_flight") if two_in_flight else None,
            "two_in_flight": two_in_flight,
            "checks": {"c_hist_sum": hsum, "expected": world * W * n},
            "renders_per_collective": M if dist is not None else None,
            "rccl_ranks_seen": ranks_seen,
        }
        if gather_ms is not None:
            out["rgba_gather_ms"] = gather_ms
            out["rgba_gather_GBps"] = (world - 1) * rgba.numel() / (gather_ms * 1e-3) / 1e9
            out["rgba_place_ms"] = place_ms
            out["rgba_place_GBps"] = 2 * world * rgba.numel() / (place_ms * 1e-3) / 1e9 if place_ms else None
            out["checks"]["merged_image_equals_single_slice_renders"] = merged_ok
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(fmt, n, window)
        if world == 1 and not args.no_e2e and args.config == "cfg2":
            out["e2e"] = e2e_dropin()
        print(json.dumps(out))
        if hsum != world * W * n or merged_ok is False:
            print("bench.py: colour histogram total %d (expected %d pixels), merged image check %r" % (hsum, world * W * n, merged_ok), file=sys.stderr)
            plan.close()
            sys.exit(1)
    plan.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
