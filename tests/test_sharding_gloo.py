"""The N > 1 path on CPU: two gloo ranks play two reference workers.  The per-rank renderer is injected — here the C oracle,
because no GPU exists in this container (the product's own renderer is HipWorker.render) — so what is tested is the
slicing, the collectives and the strip placement, against the reference's merged vectors for 2 workers."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from __graft_entry__ import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, case_names, out_dir, renderer="oracle"):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import goldenlib
    from __graft_entry__ import load_package
    from oracle import pyoracle
    pkg = load_package()
    from spectroplot_js_amd import sharding
    g = goldenlib.Golden()

    def oracle_worker(m):   # stands in for HipWorker.render on a GPU-less box (tests may use the oracle)
        lut = np.asarray(m["cmap"], dtype=np.uint8)
        r = pyoracle.render(m["format"], m["buffer"], m["n"], m["windowc"], m["block_norm"], m["gain"], m["range"], lut, m["width"],
                            m["channelMode"], m["waterfall"])
        return {"cB_hist": r["cB_hist"], "c_hist": r["c_hist"], "dBfs_min": r["dBfs_min"], "dBfs_max": r["dBfs_max"],
                "offset": m["offset"], "imageData": {"data": r["rgba"]}}

    worker = oracle_worker
    hip = None
    if renderer == "hip":   # the product's renderer: every rank drives its own HipWorker (ranks share GPU 0 on a one-GPU box)
        import torch
        hip = pkg.HipWorker(rank % max(1, torch.cuda.device_count()))
        worker = hip.render

    ok = True
    for name in case_names:
        c, e = g.cases[name], g.expected[name]
        data = g.input(c)
        win, weight = pyoracle.window(c["window"], c["n"])
        cmap = g.lut(c, force_ends=False).tolist()
        m = sharding.render_sharded(worker, data, c["format"], c["n"], c["width"], win, weight, cmap, c["gain"], c["range"],
                                    c["channelMode"], c["waterfall"], force_ends=c["force_ends"])
        ok &= [int(v) for v in m["c_hist"]] == e["merged"]["c_hist"]
        ok &= goldenlib.same_f64(m["dBfs_min"], e["merged"]["dBfs_min"]) and goldenlib.same_f64(m["dBfs_max"], e["merged"]["dBfs_max"])
        ok &= m["slice_width"] == e["merged"]["slice_width"]
        ok &= not goldenlib.check_reply({"rgba": m["reply"]["imageData"]["data"], "gauge_mins": b"", "gauge_maxs": b"", "gauge_amps": b"",
                                         "c_hist": m["reply"]["c_hist"], "cB_hist": m["reply"]["cB_hist"], "dBfs_min": m["reply"]["dBfs_min"],
                                         "dBfs_max": m["reply"]["dBfs_max"]},
                                        dict(e["slices"][rank], gauge_mins="", gauge_maxs="", gauge_amps=""))
        if rank == 0:
            ok &= goldenlib.sha256(m["data"]) == e["merged"]["rgba_sha256"]
        else:
            ok &= m["data"] is None
    if hip is not None:
        hip.terminate()
    with open(os.path.join(out_dir, "rank%d" % rank), "w") as f:
        f.write("ok" if ok else "FAIL")
    dist.destroy_process_group()


def test_two_ranks_reproduce_reference_two_worker_merge(tmp_path, golden):
    names = [n for n, c in golden.cases.items() if c.get("slices") == 2]
    assert len(names) >= 3
    mp.spawn(_rank_main, args=(2, _free_port(), names, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "rank0").read() == "ok"
    assert open(tmp_path / "rank1").read() == "ok"


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_ranks_with_the_product_renderer_reproduce_reference_merge(tmp_path, golden, world):
    """sharding.render_sharded with HipWorker.render as every rank's worker (the 8-GPU path of BASELINE config 4, with the ranks
    sharing the one GPU of this box; gloo carries the merge collectives): slices, side outputs and the merged image against the
    reference run with `world` workers."""
    names = [n for n, c in golden.cases.items() if c.get("slices") == world]
    assert names
    mp.spawn(_rank_main, args=(world, _free_port(), names, str(tmp_path), "hip"), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / ("rank%d" % r)).read() == "ok", r


def _device_rank_main(rank, world, port, case_names, out_dir, backend):
    """Every rank: its slice of the capture on the device -> sharding.render_sharded_device -> merged record / image on the device."""
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import goldenlib
    from __graft_entry__ import load_package
    from oracle import pyoracle
    pkg = load_package()
    from spectroplot_js_amd import sharding
    g = goldenlib.Golden()
    ctx = pkg.Context(dev.index)
    bad = []

    def chk(cond, what):
        if not cond:
            bad.append(what)

    for name in case_names:
        c, e = g.cases[name], g.expected[name]
        data = g.input(c)
        win, weight = pyoracle.window(c["window"], c["n"])
        lut = g.lut(c, force_ends=c["force_ends"])
        plan = ctx.plan(c["format"], c["n"], win, 1.0 / weight, c["gain"], c["range"], lut, c["channelMode"], c["waterfall"])
        b0, b1 = sharding.my_slice(data.size, pkg.parse_format(c["format"])[1], rank, world)
        d_slice = torch.from_numpy(np.ascontiguousarray(data[b0:b1])).to(dev)
        m = sharding.render_sharded_device(plan, d_slice, c["width"], waterfall=c["waterfall"])
        L = len(lut)
        rec = m["record"].cpu().numpy()
        chk([int(v) for v in rec[:L]] == e["merged"]["c_hist"], name + ": [int(v) for v in rec[:L]] == e['merged']['c_hist']")
        mm = rec[L + 1000:].view(np.float64)
        # the caller starts its merge from dBfs_min = 0, dBfs_max = -200 (spectroplot.js:1125-1126); every slice reply is already clamped so
        chk(goldenlib.same_f64(float(mm[0]), e["merged"]["dBfs_min"]) and goldenlib.same_f64(float(mm[1]), e["merged"]["dBfs_max"]), name + ": goldenlib.same_f64(float(mm[0]), e['merged']['dBfs")
        chk(m["slice_width"] == e["merged"]["slice_width"], name + ": m['slice_width'] == e['merged']['slice_width']")
        chk(goldenlib.sha256(m["strip"].cpu().numpy()) == e["slices"][rank]["rgba_sha256"], name + ": goldenlib.sha256(m['strip'].cpu().numpy()) == e['s")
        slw = m["slice_width"]
        gz = m["gauges"].cpu().numpy()
        for k, key in enumerate(("gauge_mins", "gauge_maxs", "gauge_amps")):
            chk(gz[k * slw:(k + 1) * slw].tobytes().hex() == e["slices"][rank][key], name + ": gz[k * slw:(k + 1) * slw].tobytes().hex() == e['sl")
        if rank == 0:
            chk(m["image"].is_cuda and goldenlib.sha256(m["image"].cpu().numpy()) == e["merged"]["rgba_sha256"], name + ": m['image'].is_cuda and goldenlib.sha256(m['image']")
        else:
            chk(m["image"] is None, name + ": m['image'] is None")
        plan.close()
    ctx.close()
    with open(os.path.join(out_dir, "rank%d" % rank), "w") as f:
        f.write("ok" if not bad else "FAIL " + "; ".join(bad[:12]))
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,backend", [(2, "gloo"), (8, "gloo"), (1, "nccl")])
def test_device_resident_sharded_render_reproduces_reference_merge(tmp_path, golden, world, backend):
    """sharding.render_sharded_device: slice bytes in HBM -> sp_plan_execute -> records all-gathered and merged on the device
    (sp_merge_replies) -> strips gathered and placed on the device (sp_place_strips), against the reference's merged vectors for `world`
    workers (cfg4_scaled_8slices among them).  2 and 8 ranks share this box's one GPU (gloo carries the collectives); the one-rank nccl
    group runs the same code with RCCL moving device tensors."""
    names = [n for n, c in golden.cases.items() if c.get("slices") == world]
    if world == 1:
        names = [n for n, c in golden.cases.items() if c.get("slices") == 1][:4] or []
    assert names, "no golden case with %d slices" % world
    mp.spawn(_device_rank_main, args=(world, _free_port(), names, str(tmp_path), backend), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / ("rank%d" % r)).read() == "ok", r


def _batcher_rank(rank, world, port, steps, per, out_dir):
    """bench.py's N > 1 bookkeeping on CPU tensors: sharding.RecordBatcher (batched all-gather of the records, merge one batch later)
    and sharding.gather_to_one_buffer (the strips into views of one buffer), with the device library replaced by torch arithmetic."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    load_package()
    from spectroplot_js_amd import sharding
    L, CB = 7, 1000
    P = L + CB + 2
    f64 = lambda t: t.view(torch.float64)  # noqa: E731

    def record_of(r, k):        # what rank r's render k "produces": counts that name (rank, render), a dBfs range as f64 bit patterns
        rec = torch.zeros(P, dtype=torch.int64)
        rec[:L] = torch.arange(L) + 100 * r + k
        rec[L:L + CB] = (torch.arange(CB) * (r + 1) + k) % 13
        f64(rec[L + CB:])[:] = torch.tensor([-1.5 * r - k, -150.0 + r + 2.0 * k], dtype=torch.float64)
        return rec

    merges = []

    def merge_fn(by_rank, count, out):      # sp_merge_replies in torch: sums, min / max from (0, -200)  (lib/spectroplot.js:1229-1238)
        assert by_rank.shape == (count * P,) and count == world and by_rank.is_contiguous()
        recs = by_rank.view(count, P)
        out[:L + CB] = recs[:, :L + CB].sum(0)
        mm = f64(recs[:, L + CB:].contiguous()).view(count, 2)
        f64(out[L + CB:])[:] = torch.stack([torch.minimum(mm[:, 0].min(), torch.tensor(0.0, dtype=torch.float64)),
                                            torch.maximum(mm[:, 1].max(), torch.tensor(-200.0, dtype=torch.float64))])
        merges.append(out.clone())

    def merge_batch_fn(gathered, ranks, renders, merged_all):   # sp_merge_replies_batch in torch: [rank][render][record] -> [render][record]
        assert gathered.shape == (ranks * renders * P,) and merged_all.shape == (renders * P,) and ranks == world
        g = gathered.view(ranks, renders, P)
        for j in range(renders):
            merge_fn(g[:, j, :].contiguous().view(-1), ranks, merged_all[j * P:(j + 1) * P])
            merges.pop()

    ok = True
    # ... the same batches reduced by ONE call per collective (what bench.py does on the device): the last render's merged record must
    # be the one final_record() names, partial last batch included
    for M in (1, 3, 16):
        b = sharding.RecordBatcher(world, M, P, "cpu", merge_fn=merge_fn, merge_batch_fn=merge_batch_fn)
        for k in range(steps):
            b.next_record().copy_(record_of(rank, k))
            b.rendered()
            if k % 7 == 6:
                b.finish()                               # (a caller may finish in the middle: partial batches, then a fresh one)
        b.finish()
        want = torch.zeros(P, dtype=torch.int64)
        k = steps - 1
        want[:L + CB] = sum(record_of(r, k)[:L + CB] for r in range(world))
        f64(want[L + CB:])[:] = torch.tensor([min(0.0, min(-1.5 * r - k for r in range(world))),
                                             max(-200.0, max(-150.0 + r + 2.0 * k for r in range(world)))], dtype=torch.float64)
        ok &= bool(torch.equal(b.final_record(), want))
    for M in (1, 3, 16):
        merges.clear()
        b = sharding.RecordBatcher(world, M, P, "cpu", merge_fn=merge_fn)
        ok &= b.records[0].shape == (M * P,) and b.gathered[0].shape == (world * M * P,)
        for k in range(steps):
            rec = b.next_record()
            ok &= rec.shape == (P,)
            rec.copy_(record_of(rank, k))
            b.rendered()
        b.finish()
        ok &= len(merges) == steps                      # every render merged once, in order, batches that do not divide the steps included
        for k, got in enumerate(merges):
            want = torch.zeros(P, dtype=torch.int64)
            merge_fn_expected = sum(record_of(r, k)[:L + CB] for r in range(world))
            want[:L + CB] = merge_fn_expected
            f64(want[L + CB:])[:] = torch.tensor([min(0.0, min(-1.5 * r - k for r in range(world))),
                                                 max(-200.0, max(-150.0 + r + 2.0 * k for r in range(world)))], dtype=torch.float64)
            ok &= bool(torch.equal(got, want))
        ok &= bool(torch.equal(b.final_record(), merges[-1]))
    # one rank without a process group keeps its own records
    solo = sharding.RecordBatcher(1, 1, P, "cpu", collectives=False)
    solo.next_record().copy_(record_of(rank, 5))
    solo.rendered()
    solo.finish()
    ok &= bool(torch.equal(solo.final_record(), record_of(rank, 5)))
    # strips: one buffer on the root, rank order
    strip = (torch.arange(per, dtype=torch.int64) * (rank + 3) % 251).to(torch.uint8)
    allstrips = sharding.gather_to_one_buffer(strip, dst=0)
    if rank == 0:
        ok &= allstrips.shape == (max(world * per, 16),)
        for r in range(world):
            ok &= bool(torch.equal(allstrips[r * per:(r + 1) * per], (torch.arange(per, dtype=torch.int64) * (r + 3) % 251).to(torch.uint8)))
    else:
        ok &= allstrips is None
    with open(os.path.join(out_dir, "rank%d" % rank), "w") as f:
        f.write("ok" if ok else "FAIL")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_record_batches_and_strip_gather_of_the_bench_on_cpu(tmp_path, world):
    """The bookkeeping bench.py --gpus N runs around its renders - records in batches of M per collective (M = 1, 3, 16 against 20
    renders: partial batches, two buffers in rotation), each render's records merged exactly once and in order; strips gathered into
    views of one buffer - with gloo on CPU tensors, so that a first multi-GPU run is not its first run."""
    port = _free_port()
    mp.spawn(_batcher_rank, args=(world, port, 20, 4 * 6 * 8, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "rank%d" % r)).read() == "ok"
