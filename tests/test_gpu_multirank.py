"""The N>1 path with real kernels: 2 and 3 processes share the one GPU of the test box (hash shards ->
all-gather -> sharded scan -> edge merge -> labels) and must reproduce the single-process result exactly.
Collectives run over gloo here; the RCCL calls themselves are rehearsed at world size 1 by
`torch.distributed.run ... bench.py` and run for real by the driver's scaling bench."""
from __future__ import annotations

import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_pipeline_matches_single_process(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    worker = os.path.join(ROOT, "tests", "_multirank_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, worker, ROOT, str(port), str(r), str(world), "3001", "256"], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]


def test_rccl_entry_points_one_rank_and_the_reorder_step():
    """The C-ABI exchange (ke_comm_* / ke_allgather_hashes / ke_allgather_edges) on a real RCCL communicator.  A one-GPU
    box can only host a one-rank communicator (RCCL refuses two ranks on one device), so this rehearses the calls, the
    record-growth protocol of the edge gather and the error paths; the reorder kernel is checked on its own for 2, 3 and 8
    ranks against the host restatement of the partition (kobato_eyes_amd.distributed.interleave_gathered)."""
    import ctypes as C

    import numpy as np

    from kobato_eyes_amd import _native
    from kobato_eyes_amd.distributed import RcclExchange, interleave_gathered, owned_indices

    ctx = _native.get_context(0)
    ex = RcclExchange(ctx)                                     # no process group: unique id -> ke_comm_create(world 1)
    try:
        n = 10_007
        table = (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(0x5555)
        d_in, d_out = ctx.malloc(n * 8), ctx.malloc(n * 8)
        try:
            ctx.memcpy(d_in, table, n * 8)
            ex.hashes(d_in, n, d_out)
            ctx.synchronize()
            back = np.empty(n, np.uint64)
            ctx.memcpy(back, d_out, n * 8)
            assert np.array_equal(back, table)
            for world in (2, 3, 8):                            # reorder of a gathered buffer, as N ranks would produce it
                per = (n + world - 1) // world
                parts = [np.concatenate([table[owned_indices(n, r, world)], np.zeros(per, np.uint64)])[:per] for r in range(world)]
                assert np.array_equal(interleave_gathered(parts, n), table)
                g = ctx.malloc(world * per * 8)
                try:
                    ctx.memcpy(g, np.concatenate(parts), world * per * 8)
                    ctx.interleave_shards(g, world, n, d_out)
                    ctx.memcpy(back, d_out, n * 8)
                finally:
                    ctx.free(g)
                assert np.array_equal(back, table), world
        finally:
            ctx.free(d_in)
            ctx.free(d_out)
        # edge gather: fewer edges than the record carries, then more (second, wider round), then the grown record
        for count in (0, 5, 1024, 5000, 5000, 7):
            edges = np.zeros(max(count, 1), _native.EDGE_DTYPE)
            edges["a"] = np.arange(len(edges)); edges["b"] = edges["a"] + 3; edges["h"] = 2; edges["bands"] = 1
            d = ctx.malloc(edges.nbytes)
            try:
                ctx.memcpy(d, edges, edges.nbytes)
                merged, counts = ex.edges(d, count)
            finally:
                ctx.free(d)
            assert counts.tolist() == [count] and np.array_equal(merged, edges[:count])
        # capacity protocol of the raw entry: the total comes back, only what fits is written
        edges = np.zeros(300, _native.EDGE_DTYPE); edges["a"] = np.arange(300)
        d = ctx.malloc(edges.nbytes)
        try:
            ctx.memcpy(d, edges, edges.nbytes)
            small = np.full(100, -1, dtype=np.int64).view(np.uint8)[:100 * 8]
            out = np.zeros(100, _native.EDGE_DTYPE)
            total = C.c_int64(0)
            rc = ctx._lib.ke_allgather_edges(ctx._h, ex.comm, 1, d, 300, out.ctypes.data, 100, C.byref(total), None)
            assert rc == 0 and total.value == 300 and out["a"].tolist() == list(range(100))
            host_side = np.zeros(4, _native.EDGE_DTYPE)
            rc = ctx._lib.ke_allgather_edges(ctx._h, ex.comm, 1, host_side.ctypes.data, 4, out.ctypes.data, 100, C.byref(total), None)
            assert rc == -1                                    # local edges must be device memory
        finally:
            ctx.free(d)
    finally:
        ex.close()


def test_bench_self_launch_one_rank_over_rccl():
    """`python bench.py --gpus 1 --self-launch`: the parent starts one rank under torch.distributed.run before touching the GPU;
    the rank runs the whole step with the library's RCCL exchange (one-rank communicator) and the parent relays ONE line."""
    import json

    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--self-launch", "--steps", "2", "--warmup", "1",
                          "--images", "6000", "--side", "256", "--no-cpu-baseline", "--no-h2d", "--no-decode"],
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["self_launched"] is True and out["n_gpus"] == 1 and out["n_ranks_seen"] == 1
    assert out["config"]["exchange"].startswith(("ke_allgather_hashes", "torch.distributed"))
    assert set(out["phase_ms"]) == {"hash", "allgather_hashes", "scan+count", "edge_merge", "labels"}
    assert out["edges"] > 0 and out["value"] > 0
