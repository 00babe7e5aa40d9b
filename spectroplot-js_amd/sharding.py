"""Multi-GPU time sharding of one capture: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The reference already shards a capture: `processData` cuts it into `renderWorkerCount` contiguous, equal sample ranges
(lib/samples.js:253-258), renders each range independently with its own frame stride (lib/worker.js:50) and merges the
replies (lib/spectroplot.js:1229-1244).  Here rank r plays worker r: it renders slice r on its GPU; the merge becomes
  * c_hist, cB_hist : all-reduce SUM             (spectroplot.js:1233-1238)
  * dBfs_min / max  : all-reduce MIN / MAX        (spectroplot.js:1230-1231)
  * RGBA strips     : gather to `dst`, placed like putImageData(strip, offset, 0) — column bands of the n x width image —
                      or, for the waterfall layout, row bands in reverse rank order (spectroplot.js:1241-1244)
There is no halo and no collective inside the frame loop: frames never straddle slices in the reference either, so parity
for a W-rank run is defined against the reference run with W workers.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import binding


def my_slice(nbytes, sample_width, rank, world):
    """Byte range [begin, end) of this rank's slice (lib/samples.js:253-258; remainder samples are dropped)."""
    return binding.slice_bounds(nbytes, sample_width, rank, world)


def slice_width(width, world):
    return width // world            # ~~(width / renderWorkerCount), spectroplot.js:1208


def merge_side_outputs(c_hist, cb_hist, dbfs_min, dbfs_max, device=None, group=None):
    """All-reduces the per-slice side outputs in place of the caller's merge loop; returns (c_hist, cB_hist, min, max).
    Host-side replies (the worker-message path): the counts travel through `device` when one is given (required for the nccl backend),
    else as CPU tensors (gloo).  The HBM-resident path is render_sharded_device below."""
    dev = device or "cpu"
    h = torch.cat([torch.as_tensor(np.asarray(c_hist, dtype=np.int64)), torch.as_tensor(np.asarray(cb_hist, dtype=np.int64))]).to(dev)
    mm = torch.tensor([-float(dbfs_min), float(dbfs_max)], dtype=torch.float64, device=dev)   # (-min, max): one MAX serves both
    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(mm, op=dist.ReduceOp.MAX, group=group)
    h = h.cpu().numpy()
    L = len(c_hist)
    # the caller starts its merge from dBfs_min = 0, dBfs_max = -200 (spectroplot.js:1125-1126)
    return h[:L], h[L:], min(0.0, -float(mm[0])), max(-200.0, float(mm[1]))


def gather_strips(strip, n, width, waterfall, dst=0, group=None):
    """Gathers every rank's RGBA strip (uint8 tensor, 4 * slice_width * n bytes) to `dst` and places them in the full
    n x width (spectrogram) or width x n (waterfall) image; columns beyond world * slice_width stay zero, as on the
    reference's canvas.  Returns the merged uint8 tensor on `dst`, None elsewhere."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    sw = slice_width(width, world)
    strip = strip.reshape(-1)
    bufs = [torch.empty_like(strip) for _ in range(world)] if rank == dst else None
    dist.gather(strip, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    if waterfall:
        img = torch.zeros((width, n, 4), dtype=torch.uint8, device=strip.device)
        for r, b in enumerate(bufs):
            y0 = width - sw - r * sw                               # putImageData(strip, 0, width - sliceWidth - offset)
            img[y0:y0 + sw] = b.reshape(sw, n, 4)
    else:
        img = torch.zeros((n, width, 4), dtype=torch.uint8, device=strip.device)
        for r, b in enumerate(bufs):
            img[:, r * sw:(r + 1) * sw] = b.reshape(n, sw, 4)       # putImageData(strip, offset, 0)
    return img.reshape(-1)


def gather_to_one_buffer(strip, dst=0, group=None):
    """Gathers every rank's strip (1-D uint8 tensor, the same length on all ranks) into ONE buffer on `dst`, rank order - the layout
    sp_place_strips takes: the receive list handed to dist.gather is a list of views of that buffer.  Returns it on `dst`, None
    elsewhere."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    per = strip.numel()
    allstrips, bufs = None, None
    if rank == dst:
        allstrips = torch.empty(max(world * per, 16), dtype=torch.uint8, device=strip.device)
        bufs = [allstrips[r * per:(r + 1) * per] for r in range(world)]
    dist.gather(strip, bufs, dst=dst, group=group)
    return allstrips


class RecordBatcher:
    """The side-output records of a stream of sliced renders, merged M renders per collective (what bench.py --gpus N times).

    Every render of this rank writes one record [c_hist | cB_hist | dBfs_min, dBfs_max] of `record_len` int64 into the batch buffer
    (`next_record()` names the place); after M renders ONE all_gather_into_tensor ships the batch (queued behind the batch's kernels,
    it runs while the next batch computes: two buffers in rotation) and the batch gathered before it is reduced render by render with
    `merge_fn(by_rank, count, out)` - `by_rank` the `count` ranks' records of one render end to end, `out` the merged record
    (sp_merge_replies on the device; the caller's merge, lib/spectroplot.js:1229-1238) - or, where the caller has one, with ONE call of
    `merge_batch_fn(gathered, ranks, renders, merged_all)` per batch (`gathered` = [rank][M][record] exactly as the all-gather left it,
    `merged_all` = [M][record]; only the first `renders` of a partial batch are meaningful): sp_merge_replies_batch, one launch instead
    of M launches and a re-ordering copy between the frame loops.  With `collectives=False` (one rank, no process group) the records are
    only kept."""

    def __init__(self, world, renders_per_collective, record_len, device, merge_fn=None, group=None, collectives=True, merge_batch_fn=None):
        self.world, self.M, self.P = int(world), max(1, int(renders_per_collective)), int(record_len)
        self.group, self.merge_fn, self.collectives = group, merge_fn, bool(collectives)
        self.merge_batch_fn = merge_batch_fn
        self.merged_all = torch.zeros(self.M * self.P, dtype=torch.int64, device=device) if merge_batch_fn is not None else None
        self.records = [torch.zeros(self.M * self.P, dtype=torch.int64, device=device) for _ in range(2)]
        self.gathered = [torch.zeros(self.world * self.M * self.P, dtype=torch.int64, device=device) for _ in range(2)] if collectives else None
        self.merged = torch.zeros(self.P, dtype=torch.int64, device=device)
        self.k, self.pending, self.last = 0, None, None

    def next_record(self):
        """The int64 view [record_len] the next render writes its record into."""
        slot, j = (self.k // self.M) & 1, self.k % self.M
        return self.records[slot][j * self.P:(j + 1) * self.P]

    def by_render(self, slot):
        """The gathered batch [rank][render][record] re-ordered to [render][rank][record]: one render's records end to end."""
        g = self.gathered[slot]
        return g.view(self.world, self.M, self.P).transpose(0, 1).contiguous() if self.M > 1 else g.view(1, self.world, self.P)

    def _merge(self, slot, count):
        if self.merge_batch_fn is not None:
            # (a partial batch's trailing records are stale on every rank alike: they are merged too and never looked at)
            self.merge_batch_fn(self.gathered[slot], self.world, self.M, self.merged_all)
            self.merged = self.merged_all[(count - 1) * self.P:count * self.P]
            return
        b = self.by_render(slot)
        for j in range(count):
            self.merge_fn(b[j].reshape(-1), self.world, self.merged)

    def _ship(self, slot, count):
        work = dist.all_gather_into_tensor(self.gathered[slot], self.records[slot], group=self.group, async_op=True)
        if self.pending is not None:          # the previous batch's gather has long finished: merge it
            pw, pslot, pcount = self.pending
            pw.wait()
            self._merge(pslot, pcount)
        self.pending = (work, slot, count)

    def rendered(self):
        """The render that writes next_record() has been queued."""
        slot, j = (self.k // self.M) & 1, self.k % self.M
        self.last = (slot, j)
        self.k += 1
        if self.collectives and j == self.M - 1:
            self._ship(slot, self.M)

    def finish(self):
        """Ships a partial batch and merges what is still in flight (every rank calls it at the same point)."""
        if not self.collectives:
            return
        if self.k % self.M:
            self._ship((self.k // self.M) & 1, self.k % self.M)
            self.k = (self.k // self.M + 1) * self.M
        if self.pending is not None:
            pw, pslot, pcount = self.pending
            pw.wait()
            self._merge(pslot, pcount)
            self.pending = None

    def final_record(self):
        """The merged record of the last render (after finish()), or this rank's own last record without collectives."""
        if self.collectives:
            return self.merged
        slot, j = self.last
        return self.records[slot][j * self.P:(j + 1) * self.P]


def render_sharded(render_fn, data, fmt, n, width, windowc, weight, cmap, gain, rng, channel_mode=False, waterfall=False,
                   force_ends=True, device=None, dst=0, group=None):
    """One capture over all ranks of the process group.  `render_fn(message) -> reply` is this rank's worker
    (HipWorker.render on its GPU).  `data` is the whole capture (numpy uint8) — each rank touches only its slice.
    Returns a dict with the merged side outputs on every rank and the merged image on `dst`."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    _, sw = binding.parse_format(fmt)
    cmap = [list(c) for c in cmap]
    if force_ends:
        cmap[0] = [0, 0, 0]
        cmap[-1] = [255, 255, 255]
    slw = slice_width(width, world)
    b0, b1 = my_slice(len(data), sw, rank, world)
    reply = render_fn({"block_norm": 1.0 / weight, "gain": gain, "range": rng, "cmap": cmap, "n": n, "windowc": windowc,
                       "width": slw, "offset": rank * slw, "buffer": data[b0:b1], "format": fmt, "channelMode": channel_mode,
                       "waterfall": waterfall})
    c_hist, cb_hist, dmin, dmax = merge_side_outputs(reply["c_hist"], reply["cB_hist"], reply["dBfs_min"], reply["dBfs_max"],
                                                     device=device, group=group)
    strip = torch.as_tensor(np.asarray(reply["imageData"]["data"], dtype=np.uint8))
    if device is not None:
        strip = strip.to(device)
    img = gather_strips(strip, n, width, waterfall, dst=dst, group=group)
    return {"data": None if img is None else img.cpu().numpy(), "c_hist": c_hist, "cB_hist": cb_hist, "dBfs_min": dmin,
            "dBfs_max": dmax, "slice_width": slw, "reply": reply}


def render_sharded_device(plan, d_slice, width, waterfall=False, dst=0, group=None, want_image=True):
    """One capture over all ranks, operands resident in HBM from the raw bytes to the merged image.

    `plan` is this rank's binding.Plan (same request on every rank), `d_slice` a uint8 CUDA tensor holding this rank's slice of the
    capture (bytes [my_slice(...)) of it), `width` the frames of the WHOLE image.  Rank r renders `slice_width(width, world)` frames with
    sp_plan_execute into a device strip and a device record [c_hist | cB_hist | dBfs range]; the records are all-gathered and reduced on
    the device (sp_merge_replies: the caller's merge, lib/spectroplot.js:1229-1238); the strips are gathered to `dst` and placed on the
    device (sp_place_strips: putImageData, :1241-1244).  This function never copies a strip, histogram or gauge to host memory: the
    collectives are handed device tensors (nccl = RCCL moves them over xGMI; gloo, used by the tests whose ranks share one GPU, stages
    them through the host inside the collective: its transport, not this path).

    Returns device tensors: {"image": uint8 [4 * width * n] on dst (None elsewhere), "record": int64 [L + 1000 + 2] merged side outputs
    (every rank), "strip", "gauges": uint8 [3 * slice_width] (this rank's gauge_mins | gauge_maxs | gauge_amps), "slice_width"}."""
    ctx, n, L = plan.ctx, plan.n, plan.lut_len
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = d_slice.device
    slw = slice_width(width, world)
    P = L + binding.SP_CB_HIST_SIZE + 2
    # one explicit stream for the library's kernels and the collectives (the null stream's handle is 0, which sp_context_set_stream
    # reads as "use your own stream": the render would then not be ordered before the collectives)
    stream = torch.cuda.current_stream(dev)
    if stream.cuda_stream == 0:
        stream = torch.cuda.Stream(device=dev)
        stream.wait_stream(torch.cuda.current_stream(dev))   # whatever produced d_slice there
    # The context's work so far (its own stream, or one a caller bound) must not overlap what follows: its workspaces rely on one
    # stream's order.  The binding is restored on the way out.
    previous = ctx.get_stream()
    if previous != stream.cuda_stream:
        ctx.synchronize()
    ctx.set_stream(stream.cuda_stream)
    try:
        with torch.cuda.stream(stream):
            return _render_sharded_device_on(stream, plan, d_slice, width, waterfall, dst, group, want_image, rank, world, slw, P)
    finally:
        ctx.set_stream(previous)


def _render_sharded_device_on(stream, plan, d_slice, width, waterfall, dst, group, want_image, rank, world, slw, P):
    ctx, n, L = plan.ctx, plan.n, plan.lut_len
    dev = d_slice.device
    d_slice.record_stream(stream)
    strip = torch.empty(max(4 * slw * n, 16), dtype=torch.uint8, device=dev)
    gauges = torch.empty(max(3 * slw, 16), dtype=torch.uint8, device=dev)
    record = torch.zeros(P, dtype=torch.int64, device=dev)
    p = record.data_ptr()
    plan.execute(d_slice.data_ptr(), d_slice.numel(), slw, strip.data_ptr(), gauges.data_ptr(), gauges.data_ptr() + slw,
                 gauges.data_ptr() + 2 * slw, p, p + 8 * L, p + 8 * (L + binding.SP_CB_HIST_SIZE))
    gathered = torch.empty(world * P, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(gathered, record, group=group)
    merged = torch.empty(P, dtype=torch.int64, device=dev)
    m = merged.data_ptr()
    ctx.merge_replies(gathered.data_ptr(), world, L, m, m + 8 * L, m + 8 * (L + binding.SP_CB_HIST_SIZE))
    image = None
    if want_image:
        per = 4 * slw * n
        allstrips = gather_to_one_buffer(strip[:per], dst=dst, group=group)          # one buffer on dst, rank order
        if rank == dst:
            image = torch.zeros(4 * width * n, dtype=torch.uint8, device=dev)        # un-rendered columns stay blank, as on the canvas
            if per:
                ctx.place_strips(image.data_ptr(), allstrips.data_ptr(), world, n, width, slw, waterfall)
    stream.synchronize()
    return {"image": image, "record": merged, "strip": strip[:4 * slw * n], "gauges": gauges[:3 * slw], "slice_width": slw}
