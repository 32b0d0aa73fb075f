"""Multi-GPU sharding of the path: one process per GPU, torch.distributed (RCCL on ROCm, gloo on
CPU) as plumbing.

Images are hash-partitioned over ranks (image i belongs to rank i mod world); after local
hashing ONE all-gather shares the 64-bit hash table; the tile triangle of the
pair scan is then dealt round-robin to ranks (ke_hamming_scan part_index/part_count), each
rank returns its edges to the host, and rank 0 (or every rank) merges them for clustering.
No other collective exists on the path.
"""
from __future__ import annotations

from typing import Optional

import numpy as np


def owned_indices(n_items: int, rank: int, world: int) -> np.ndarray:
    """Positions of the corpus this rank hashes: i = rank (mod world)."""
    return np.arange(rank, n_items, world, dtype=np.int64)


def interleave_gathered(parts: list, n_items: int) -> np.ndarray:
    """Undo the mod-world partition: parts[r][k] is item r + k*world."""
    world = len(parts)
    out = np.empty(n_items, dtype=parts[0].dtype)
    for r, p in enumerate(parts):
        out[r::world] = p[: len(range(r, n_items, world))]
    return out


def allgather_hashes(local, n_items: int, *, group=None):
    """All-gather of the per-rank hash shards -> the full table in corpus order.

    ``local``: torch tensor (int64 view of the u64 hashes) on the rank's device, padded or not;
    every rank must pass ceil(n_items / world) elements.  Returns a torch tensor of n_items
    int64 on the same device.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    per = (n_items + world - 1) // world
    if local.numel() != per:
        padded = torch.zeros(per, dtype=local.dtype, device=local.device)
        padded[: local.numel()] = local
        local = padded
    gathered = torch.empty(world * per, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, local.contiguous(), group=group)
    # gathered[r*per + k] is item r + k*world: transpose (world, per) -> (per, world) restores corpus order
    return gathered.view(world, per).t().contiguous().view(-1)[:n_items]


def gather_edges(edges: np.ndarray, *, group=None, dst: Optional[int] = None) -> Optional[np.ndarray]:
    """Collect every rank's edge array on ``dst`` (None = all ranks) through the host-side
    object collective; edge volume is O(#near-duplicates), so this is not a data-path step."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if dst is None:
        parts = [None] * world
        dist.all_gather_object(parts, edges, group=group)
    else:
        parts = [None] * world if dist.get_rank(group) == dst else None
        dist.gather_object(edges, parts, dst=dst, group=group)
        if parts is None:
            return None
    return np.concatenate(parts) if parts else edges
