"""Multi-GPU sharding of the path: one process per GPU, torch.distributed (RCCL on ROCm, gloo on
CPU) as plumbing.

Images are hash-partitioned over ranks (image i belongs to rank i mod world); after local
hashing ONE all-gather shares the 64-bit hash table; the tile triangle of the
pair scan is then dealt round-robin to ranks (ke_hamming_scan part_index/part_count), each
rank returns its edges to the host, and rank 0 (or every rank) merges them for clustering.
With the SSIM refine stage on, the merged candidate pairs are dealt round-robin to the ranks and one more
small all-gather returns the scores (``ssim_refine_sharded``).
"""
from __future__ import annotations

from typing import Optional

import numpy as np


class RcclExchange:
    """The two exchange steps of the sharded path through the C ABI (``ke_allgather_hashes`` / ``ke_allgather_edges``,
    include/keyes.h) on a communicator the library makes itself: rank 0 draws the unique id, ``torch.distributed`` (any
    backend -- it is only the rendezvous) hands it round, every rank joins with ``ke_comm_create``.  Without a process
    group this is a one-rank communicator (the RCCL calls still run)."""

    def __init__(self, ctx, *, group=None) -> None:
        import torch.distributed as dist

        from . import _native

        self.ctx = ctx
        live = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if live else 1
        self.rank = dist.get_rank(group) if live else 0
        box = [_native.comm_unique_id() if self.rank == 0 else None]
        if live and self.world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        self.comm = ctx.comm_create(box[0], self.world, self.rank)

    def hashes(self, local_ptr: int, n_total: int, table_ptr: int) -> None:
        """``local_ptr``: this rank's ceil(n_total / world) hashes (device); ``table_ptr``: n_total hashes, corpus order."""
        self.ctx.allgather_hashes(self.comm, self.world, local_ptr, n_total, table_ptr)

    def edges(self, edges_ptr: int, n_local: int):
        """-> (every rank's edges as one host array, rank by rank; per-rank counts)."""
        return self.ctx.allgather_edges(self.comm, self.world, edges_ptr, n_local)

    def close(self) -> None:
        if self.comm:
            self.ctx.comm_destroy(self.comm)
            self.comm = 0


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


_edge_slots_hint: dict = {}     # per process group: edge records every rank sends along with its count
_edge_buffers: dict = {}        # (group, device, width, world) -> (send record, gathered records), reused across calls


def allgather_edge_buffers(edges_u8, count: int, *, group=None):
    """Device-side merge of the per-rank edge lists, normally ONE collective: every rank sends a fixed-width
    record ``[count:int64 | first K edges]`` and the counts are read from the gathered headers.  K follows the
    largest count seen so far (x1.25, rounded up to a power of two); only when some rank holds more than K edges
    is a second all-gather of the full buffers needed.  ``edges_u8``: this rank's uint8 tensor holding ``count``
    24-byte ke_edge records at its start.  Returns (host ndarray of all edges as raw bytes viewed per record by
    the caller, per-rank counts)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    dev = edges_u8.device
    key = id(group) if group is not None else 0
    slots = _edge_slots_hint.get(key, 1024)
    width = 8 + slots * 24
    # the record and the gathered buffer live across calls: no allocation or fill kernel on the step's critical path
    bufs = _edge_buffers.get((key, str(dev), width, world))
    if bufs is None:
        bufs = (torch.zeros(width, dtype=torch.uint8, device=dev), torch.empty(world * width, dtype=torch.uint8, device=dev))
        _edge_buffers.clear()
        _edge_buffers[(key, str(dev), width, world)] = bufs
    send, gathered = bufs
    send[:8] = torch.tensor([int(count)], dtype=torch.int64).view(torch.uint8).to(dev)
    head = min(int(count), slots) * 24
    if head:
        send[8:8 + head] = edges_u8[:head]
    dist.all_gather_into_tensor(gathered, send, group=group)
    host = gathered.cpu().numpy().reshape(world, width)
    counts = [int(v) for v in np.ascontiguousarray(host[:, :8]).view(np.int64)[:, 0]]
    top = max(max(counts), 1)
    if top > slots:                         # rare: some list did not fit the record -> gather the full buffers
        full = top * 24
        if edges_u8.numel() < full:         # a rank with a short buffer pads up to the common width
            padded = torch.zeros(full, dtype=torch.uint8, device=dev)
            padded[: edges_u8.numel()] = edges_u8
            edges_u8 = padded
        gathered = torch.empty(world * full, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(gathered, edges_u8[:full].contiguous(), group=group)
        body = gathered.cpu().numpy().reshape(world, full)
    else:
        body = host[:, 8:]
    want = 1024
    while want < top * 5 // 4:
        want *= 2
    _edge_slots_hint[key] = max(slots, want)
    merged = np.concatenate([body[r, : counts[r] * 24] for r in range(world)])
    return merged, counts


def ssim_refine_sharded(ctx, edges: np.ndarray, fetch_images, width: int, height: int, channels: int = 3, *, group=None):
    """SSIM of every candidate pair, the pairs dealt round-robin over the ranks (SSIM stage).

    ``edges``: the merged edge records, identical on every rank (fields a, b = corpus positions).
    ``fetch_images(ids) -> int``: device pointer to the images of the sorted unique positions ``ids``, packed back to
    back -- the evaluating rank must hold both images of a pair; for the synthetic corpus they are regenerated from
    (seed, position), for real files this is a fetch.  Returns float64 SSIM per edge, the same array on every rank.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = len(edges)
    if n == 0:                                  # nothing to refine: no zero-length collective
        return np.empty(0, np.float64)
    mine = np.arange(rank, n, world)
    per = (n + world - 1) // world
    local = np.full(per, np.nan)
    if len(mine):
        a, b = edges["a"][mine].astype(np.int64), edges["b"][mine].astype(np.int64)
        ids = np.unique(np.concatenate([a, b]))
        pixels = fetch_images(ids)
        local[: len(mine)] = ctx.ssim_pairs_uniform(pixels, len(ids), width, height, channels, np.searchsorted(ids, a),
                                                    np.searchsorted(ids, b))
    if world == 1:
        return local[:n]
    dev = torch.device("cuda", ctx.device) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    gathered = torch.empty(world * per, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(gathered, torch.from_numpy(local).to(dev), group=group)
    # gathered[r*per + k] is edge r + k*world
    return gathered.cpu().numpy().reshape(world, per).T.reshape(-1)[:n]
