#!/usr/bin/env python3
"""BASELINE configs[4] in a sample that fits one GPU: N mixed-resolution images (independent width/height from
{256..4096}, weights 1/side), device resident, hashed through ke_hash_images (grouped by shape: single-pass kernels up
to 2048 pixels wide -- per band of rows for tall images -- and the strip kernel beyond).  One JSON line."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SIDES = np.array([256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=30000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--dhash", action="store_true")
    args = ap.parse_args()
    import ctypes as C

    import torch

    from kobato_eyes_amd import _native

    ctx = _native.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(20260604)
    p = (1.0 / SIDES) / (1.0 / SIDES).sum()
    w = rng.choice(SIDES, args.images, p=p).astype(np.int32)
    h = rng.choice(SIDES, args.images, p=p).astype(np.int32)
    sizes = w.astype(np.int64) * h * 3
    offsets = np.zeros(args.images, np.uint64)
    offsets[1:] = np.cumsum(sizes[:-1]).astype(np.uint64)
    total = int(sizes.sum())
    px = torch.empty(total, dtype=torch.uint8, device="cuda")
    # fill per shape group with the on-device generator (contents do not matter for throughput)
    order = np.lexsort((h, w))
    for i in order.tolist():
        ctx.synth_rgb(20260604, i % 1000, 1, int(w[i]), int(h[i]), out=px.data_ptr() + int(offsets[i]))
    torch.cuda.synchronize()
    ph = torch.empty(args.images, dtype=torch.int64, device="cuda")
    dh = torch.empty(args.images, dtype=torch.int64, device="cuda")
    status = np.empty(args.images, np.int32)
    times = []
    for _ in range(args.reps + 1):
        t0 = time.perf_counter()
        rc = ctx._lib.ke_hash_images(ctx._h, px.data_ptr(), offsets.ctypes.data, w.ctypes.data, h.ctypes.data, 3, args.images,
                                     ph.data_ptr(), dh.data_ptr() if args.dhash else None, status.ctypes.data)
        ctx._check(rc, "ke_hash_images")
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    t = float(np.median(times[1:]))
    print(json.dumps({"case": "mixed_resolution", "images": args.images, "distinct_shapes": int(len(set(zip(w.tolist(), h.tolist())))),
                      "bytes": total, "dhash": args.dhash, "wall_ms": t * 1e3, "images_per_s": args.images / t, "gbs": total / t / 1e9,
                      "frac_hbm": total / t / 8e12}))


if __name__ == "__main__":
    main()
