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


def allgather_edge_buffers(edges_u8, count: int, *, group=None):
    """Device-side merge of the per-rank edge lists: all-gather of the counts, then of the edge
    buffers cut to the largest count.  ``edges_u8``: this rank's uint8 tensor holding ``count``
    24-byte ke_edge records at its start.  Returns (host ndarray of all edges as raw bytes viewed
    per record by the caller, per-rank counts)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    mine = torch.tensor([int(count)], dtype=torch.int64, device=edges_u8.device)
    counts_t = torch.empty(world, dtype=torch.int64, device=edges_u8.device)
    dist.all_gather_into_tensor(counts_t, mine, group=group)
    counts = counts_t.cpu().tolist()
    width = max(max(counts), 1) * 24
    if edges_u8.numel() < width:           # a rank with a short buffer pads up to the common width
        padded = torch.zeros(width, dtype=torch.uint8, device=edges_u8.device)
        padded[: edges_u8.numel()] = edges_u8
        edges_u8 = padded
    gathered = torch.empty(world * width, dtype=torch.uint8, device=edges_u8.device)
    dist.all_gather_into_tensor(gathered, edges_u8[:width].contiguous(), group=group)
    host = gathered.cpu().numpy().reshape(world, width)
    merged = np.concatenate([host[r, : counts[r] * 24] for r in range(world)])
    return merged, counts
