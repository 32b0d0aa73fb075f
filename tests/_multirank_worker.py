"""Worker of tests/test_gpu_multirank.py: RANK-th of WORLD processes sharing cuda:0.  Collectives run over gloo
on host copies (RCCL cannot put two ranks on one device); everything else is the real N>1 path of bench.py:
hash-partitioned hashing, hash-table all-gather, sharded scan, edge merge, labels."""
import os
import sys

root, port, rank, world, n, side = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
sys.path.insert(0, root)
import numpy as np
import torch
import torch.distributed as dist

from kobato_eyes_amd import _native
from kobato_eyes_amd.distributed import allgather_edge_buffers, allgather_hashes, owned_indices, ssim_refine_sharded

dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
ctx = _native.Context(0)
seed = 20260604
mine = owned_indices(n, rank, world)
img_bytes = side * side * 3
px = ctx.malloc(len(mine) * img_bytes)
ctx.synth_rgb_indexed(seed, mine, side, side, px)
local = np.zeros((n + world - 1) // world, np.uint64)
ph, _ = ctx.hash_uniform(px, len(mine), side, side, 3, want_dhash=False)
local[: len(mine)] = ph
table = allgather_hashes(torch.from_numpy(local.view(np.int64)), n).numpy().view(np.uint64)
edges, counters = ctx.hamming_scan(table, n, threshold=8, part_index=rank, part_count=world)
buf = torch.from_numpy(np.ascontiguousarray(edges).view(np.uint8).copy()) if len(edges) else torch.zeros(24, dtype=torch.uint8)
merged, counts = allgather_edge_buffers(buf, len(edges))
all_edges = merged.view(_native.EDGE_DTYPE)
labels = _native.cluster_labels(all_edges, n)
# every rank checks the merged result against a single-process run of the same corpus
full_px = ctx.malloc(n * img_bytes)
ctx.synth_rgb(seed, 0, n, side, side, out=full_px)
ref_hash, _ = ctx.hash_uniform(full_px, n, side, side, 3, want_dhash=False)
assert np.array_equal(table, ref_hash), "all-gathered table differs from the single-process hashes"
ref_edges, ref_counters = ctx.hamming_scan(ref_hash, n, threshold=8)
key = lambda e: sorted(map(tuple, e[["a", "b", "h", "bands"]].tolist()))
assert key(all_edges) == key(ref_edges), "merged shard edges differ from the single-process scan"
assert sum(counts) == len(ref_edges)
pairs = torch.tensor([int(counters[0])], dtype=torch.int64)
dist.all_reduce(pairs)
assert int(pairs.item()) == n * (n - 1) // 2, "shards do not tile the pair space exactly once"
assert np.array_equal(labels, _native.cluster_labels(ref_edges, n))
# SSIM stage: the merged pairs dealt round-robin, each rank regenerating the images of its pairs from (seed, position)
scratch = {"ptr": 0}
def fetch(ids):
    if scratch["ptr"]:
        ctx.free(scratch["ptr"])
    scratch["ptr"] = ctx.malloc(len(ids) * img_bytes)
    ctx.synth_rgb_indexed(seed, ids, side, side, scratch["ptr"])
    return scratch["ptr"]
ssim = ssim_refine_sharded(ctx, all_edges, fetch, side, side, 3)
ref_ssim = ctx.ssim_pairs_uniform(full_px, n, side, side, 3, all_edges["a"], all_edges["b"])
assert len(ssim) == len(all_edges) and np.array_equal(ssim, ref_ssim), "sharded SSIM differs from the single-process scores"
assert np.array_equal(_native.cluster_labels(all_edges[ssim >= 0.9], n), _native.cluster_labels(all_edges[ref_ssim >= 0.9], n))
if scratch["ptr"]:
    ctx.free(scratch["ptr"])
ctx.free(px); ctx.free(full_px)
dist.barrier()
dist.destroy_process_group()
print(f"rank {rank}/{world} ok: {len(edges)} local edges, {len(all_edges)} merged")
