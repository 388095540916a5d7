"""Multi-GPU plumbing: images are independent, so a batch is cut into contiguous blocks, one per rank, and the
fixed-size per-strip records are gathered once at the end.  No collective on the data path.

Works with any torch.distributed backend ("nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""
from __future__ import annotations

import numpy as np


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous block [lo, hi) of rank `rank`: item i belongs to rank i*world//n (blocks differ by at most one item)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    lo = (n_items * rank) // world
    hi = (n_items * (rank + 1)) // world
    return lo, hi


def pack_strip_records(strips: np.ndarray, peaks: np.ndarray, counts: np.ndarray, image0: int = 0) -> np.ndarray:
    """The fixed-size per-strip result record of every strip of a block of images: global image index, the LfaStrip
    (box, area, label) and the LfaPeaks of its profile.  strips / peaks: [n_images][max_strips] structured arrays as the C
    ABI fills them, counts: strips per image; image0: index of the block's first image in the whole batch."""
    dt = np.dtype([("image", "<i4"), ("strip", strips.dtype), ("peaks", peaks.dtype)])
    rec = np.zeros(int(counts.sum()), dtype=dt)
    k = 0
    for i in range(len(counts)):
        c = int(counts[i])
        rec["image"][k:k + c] = image0 + i
        rec["strip"][k:k + c] = strips[i, :c]
        rec["peaks"][k:k + c] = peaks[i, :c]
        k += c
    return rec


def gather_strip_records(local_counts: np.ndarray, local_records: np.ndarray, n_items: int, device=None):
    """All-gather per-image strip counts and the packed records (any structured dtype) so that every rank ends with
    the results of all n_items images in image order.  local_counts: int32 [hi-lo]; local_records: [sum(counts)]."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_range(n_items, rank, world)
    assert len(local_counts) == hi - lo and int(local_counts.sum()) == len(local_records)
    if world == 1:
        return local_counts.copy(), local_records.copy()
    dev = device if device is not None else torch.device("cpu")
    itemsize = local_records.dtype.itemsize
    # 1) counts: fixed-size blocks padded to the largest shard
    blk = max(shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world))
    cbuf = torch.zeros(blk, dtype=torch.int32, device=dev)
    cbuf[:hi - lo] = torch.from_numpy(np.ascontiguousarray(local_counts, dtype=np.int32)).to(dev)
    call = [torch.zeros(blk, dtype=torch.int32, device=dev) for _ in range(world)]
    dist.all_gather(call, cbuf)
    counts = []
    for r in range(world):
        a, b = shard_range(n_items, r, world)
        counts.append(call[r][:b - a].cpu().numpy())
    totals = [int(c.sum()) for c in counts]
    # 2) records: byte blocks padded to the largest total
    mx = max(max(totals), 1) * itemsize
    rbuf = torch.zeros(mx, dtype=torch.uint8, device=dev)
    raw = np.ascontiguousarray(local_records).view(np.uint8).reshape(-1)
    if raw.size:
        rbuf[:raw.size] = torch.from_numpy(raw.copy()).to(dev)
    rall = [torch.zeros(mx, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(rall, rbuf)
    recs = [rall[r][:totals[r] * itemsize].cpu().numpy().view(local_records.dtype) for r in range(world)]
    return np.concatenate(counts), np.concatenate(recs)
