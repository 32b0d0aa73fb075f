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
