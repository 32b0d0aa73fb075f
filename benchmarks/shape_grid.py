#!/usr/bin/env python3
"""Hash throughput for every (width, height) of the mixed-resolution config, 2.4 GB of synthetic RGB per shape:
python benchmarks/shape_grid.py [dhash]"""
import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kobato_eyes_amd import _native
ctx = _native.Context(0)
S = [256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096]
dh = len(sys.argv) > 1
buf = torch.empty(2_500_000_000, dtype=torch.uint8, device="cuda")
ctx.synth_rgb(1, 0, 2_500_000_000 // (512 * 512 * 3), 512, 512, out=buf.data_ptr())
print("rows: width, cols: height; TB/s", "(both hashes)" if dh else "(pHash)")
print("      " + " ".join(f"{h:6d}" for h in S))
tot = 0
for w in S:
    row = []
    for h in S:
        n = max(1, 2_400_000_000 // (w * h * 3))
        ph = torch.empty(n, dtype=torch.int64, device="cuda"); dd = torch.empty(n, dtype=torch.int64, device="cuda")
        ms = []
        for _ in range(3):
            ctx.hash_uniform(buf.data_ptr(), n, w, h, 3, phash_out=ph.data_ptr(), dhash_out=dd.data_ptr() if dh else None, want_dhash=dh)
            ms.append(ctx.last_kernel_ms(0))
        t = sorted(ms)[1]; tot += t
        row.append(n * w * h * 3 / t / 1e9)
    print(f"{w:5d} " + " ".join(f"{v:6.2f}" for v in row), flush=True)
print("sum of medians ms", round(tot, 2))
