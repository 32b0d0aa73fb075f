#!/usr/bin/env python3
"""Run ONE hash case repeatedly (for rocprofv3 passes): python benchmarks/one_case.py W H N [reps] [dhash]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kobato_eyes_amd import _native
w, h, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dh = len(sys.argv) > 5 and sys.argv[5] == "dhash"
ctx = _native.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
dev = torch.device("cuda", 0)
px = torch.empty(n * w * h * 3, dtype=torch.uint8, device=dev)
ctx.synth_rgb(20260604, 0, n, w, h, out=px.data_ptr())
ph = torch.empty(n, dtype=torch.int64, device=dev)
dd = torch.empty(n, dtype=torch.int64, device=dev)
ms = []
for _ in range(reps):
    ctx.hash_uniform(px.data_ptr(), n, w, h, 3, phash_out=ph.data_ptr(), dhash_out=dd.data_ptr() if dh else None, want_dhash=dh)
    ms.append(ctx.last_kernel_ms(0))
print(w, h, n, "median ms", sorted(ms)[len(ms) // 2], "GB/s", n * 3 * w * h / sorted(ms)[len(ms) // 2] / 1e6)
