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
    """All-reduces the per-slice side outputs in place of the caller's merge loop; returns (c_hist, cB_hist, min, max)."""
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
