#!/usr/bin/env python3
"""A long-running process calling the batch hasher again and again over collections of mixed formats and sizes (the application
scans folder after folder): does the process's memory, the device's, or /dev/shm grow from call to call?
    python benchmarks/soak_fastsig.py [--rounds 30 --files 3000]
Prints one JSON line per five rounds: host RSS, free device memory (torch is only the messenger), files under /dev/shm."""
from __future__ import annotations

import argparse
import io
import json
import os
import shutil
import sqlite3
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rss_mb() -> float:
    with open("/proc/self/status") as fh:
        for line in fh:
            if line.startswith("VmRSS:"):
                return int(line.split()[1]) / 1024
    return 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=30)
    ap.add_argument("--files", type=int, default=3000)
    ap.add_argument("--fixed", action="store_true", help="the same images and sizes in every round (what grows then is a leak, not a high-water mark)")
    args = ap.parse_args()
    import torch
    from PIL import Image

    from kobato_eyes_amd import _native, fastsig, refine_parallel

    ctx = _native.get_context(0)
    rng = np.random.default_rng(3)
    root = tempfile.mkdtemp(prefix="ke_soak_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        db = os.path.join(root, "sig.db")
        with sqlite3.connect(db) as conn:
            conn.execute("CREATE TABLE signatures (file_id INTEGER PRIMARY KEY, phash_u64 INTEGER NOT NULL, dhash_u64 INTEGER NOT NULL)")
        kinds = [("jpg", "JPEG", {"quality": 85}), ("png", "PNG", {}), ("bmp", "BMP", {}), ("gif", "GIF", {}), ("webp", "WEBP", {"quality": 80, "method": 0}),
                 ("tif", "TIFF", {})]
        t0 = time.perf_counter()
        for r in range(args.rounds):
            folder = os.path.join(root, f"r{r}")
            os.makedirs(folder)
            if args.fixed:
                rng = np.random.default_rng(3)
            sizes = [(int(rng.integers(64, 900)), int(rng.integers(64, 700))) for _ in range(12)]
            px = [ctx.synth_rgb(1000 + (0 if args.fixed else r), k, 1, w, h)[0] for k, (w, h) in enumerate(sizes)]
            blobs = []
            for k, a in enumerate(px):
                ext, fmt, kw = kinds[k % len(kinds)]
                b = io.BytesIO()
                Image.fromarray(a).save(b, fmt, **kw)
                blobs.append((ext, b.getvalue()))
            blobs.append(("jpg", blobs[0][1][: len(blobs[0][1]) // 2]))          # a truncated file
            blobs.append(("png", b"not an image"))
            items = []
            for i in range(args.files):
                ext, data = blobs[int(rng.integers(0, len(blobs)))]
                p = os.path.join(folder, f"f{i:05d}.{ext}")
                with open(p, "wb") as fh:
                    fh.write(data)
                items.append((r * 1000000 + i, p))
            rows = fastsig.fast_fill_missing_signatures(db, items)
            thumbs = refine_parallel._thumbnails_decoded_on_gpu([p for _, p in items[:200]], 128, 0)
            shutil.rmtree(folder)
            if r % 5 == 4 or r == args.rounds - 1:
                free_b, total_b = torch.cuda.mem_get_info(0)
                shm = [f for f in os.listdir("/dev/shm") if f.startswith("ke_")] if os.path.isdir("/dev/shm") else []
                print(json.dumps({"round": r + 1, "rows": len(rows), "thumbnails": len(thumbs), "rss_mb": round(rss_mb(), 1),
                                  "device_free_gb": round(free_b / 2**30, 3), "shm_entries": len(shm), "elapsed_s": round(time.perf_counter() - t0, 1)}), flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
