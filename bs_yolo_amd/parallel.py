"""Multi-GPU sharding of the hot path: one process per GPU, images (or SAHI tiles) are independent units.

The reference has no multi-device inference path (select_device picks one device; DDP is training-only,
engine/trainer.py:217-224), so this is new: contiguous batch split + one all-gather of the fixed-size detection
tensor and of the per-image counts (RCCL over xGMI on GPUs, gloo on CPU for tests).  Payload per rank is
B_local * max_det * (6+nm) * 4 bytes (460 KB at B_local=64) -- latency-bound, no reduce anywhere.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous split of n_items over `world` ranks, sizes differ by at most one (70 tiles / 8 -> 9,9,9,9,9,9,8,8)."""
    q, r = divmod(n_items, world)
    out, start = [], 0
    for k in range(world):
        size = q + (1 if k < r else 0)
        out.append((start, start + size))
        start += size
    return out


def gather_detections(det: torch.Tensor, counts: torch.Tensor, n_items: Optional[int] = None, group=None):
    """det (B_local, max_det, row) fp32, counts (B_local,) int32 on every rank -> (det_all (n_items, max_det, row),
    counts_all (n_items,)) on every rank, in global item order.  Ranks may hold shards that differ by one item
    (shard_bounds); shorter shards are padded for the collective and the padding dropped afterwards."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return det, counts
    world = dist.get_world_size(group)
    b_local = det.shape[0]
    if n_items is None:
        n_items = b_local * world
    bounds = shard_bounds(n_items, world)
    b_max = max(e - s for s, e in bounds)
    assert b_local == bounds[dist.get_rank(group)][1] - bounds[dist.get_rank(group)][0], "shard size mismatch"
    if b_local < b_max:
        det = torch.cat((det, det.new_zeros((b_max - b_local,) + det.shape[1:])))
        counts = torch.cat((counts, counts.new_zeros(b_max - b_local)))
    det_all = det.new_empty((world * b_max,) + det.shape[1:])
    cnt_all = counts.new_empty(world * b_max)
    if det.is_cuda:
        dist.all_gather_into_tensor(det_all, det.contiguous(), group=group)
        dist.all_gather_into_tensor(cnt_all, counts.contiguous(), group=group)
    else:  # gloo
        dl = list(det_all.chunk(world))
        cl = list(cnt_all.chunk(world))
        dist.all_gather(dl, det.contiguous(), group=group)
        dist.all_gather(cl, counts.contiguous(), group=group)
        det_all, cnt_all = torch.cat(dl), torch.cat(cl)
    if all(e - s == b_max for s, e in bounds):
        return det_all, cnt_all
    keep = torch.cat([torch.arange(k * b_max, k * b_max + (e - s)) for k, (s, e) in enumerate(bounds)]).to(det.device)
    return det_all[keep], cnt_all[keep]
