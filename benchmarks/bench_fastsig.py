#!/usr/bin/env python3
"""The batch-hasher seam end to end (`fast_fill_missing_signatures`, the drop-in for src/core/fastsig.py:102-126): N image files
on disk -> signatures rows in SQLite.  Two runs of the same call: JPEG / PNG files decoded on the GPU (the default), and with
`KE_GPU_JPEG=0 KE_GPU_PNG=0 KE_GPU_BMP=0` (Pillow on the thread pool, pixels through the pinned staging buffers -- the route every other
format takes); both hash on the GPU.
    python benchmarks/bench_fastsig.py [--images 16384 --format jpeg|png|mixed --content corpus|drawing]
One JSON line."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=16384)
    ap.add_argument("--side", type=int, default=512)
    ap.add_argument("--format", choices=["jpeg", "png", "mixed", "bmp", "gif", "webp", "tiff", "collection"], default="jpeg",
                    help="webp / tiff: formats outside the GPU decoders -- the whole batch takes the Pillow route (decoder processes); "
                         "collection: 70 %% JPEG, 20 %% PNG, 4 %% BMP, 3 %% WebP, 3 %% TIFF in one call (the Pillow share runs beside the GPU share)")
    ap.add_argument("--distinct", type=int, default=128, help="distinct images behind the files")
    ap.add_argument("--content", choices=["corpus", "drawing"], default="corpus")
    ap.add_argument("--pillow-sample", type=int, default=4096, help="files of the batch given to the Pillow-route run")
    args = ap.parse_args()
    from PIL import Image

    from kobato_eyes_amd import _native, fastsig

    ctx = _native.get_context(0)
    distinct, s = args.distinct, args.side
    px = ctx.synth_rgb(20260604, 0, distinct, s, s)
    if args.content == "drawing":
        rng = np.random.default_rng(11)
        cells = rng.integers(0, 256, (distinct, s // 16 + 1, s // 16 + 1, 3), dtype=np.uint8)
        px = np.repeat(np.repeat(cells, 16, 1), 16, 2)[:, :s, :s].copy()
        px[:, ::48, :, :] = 0
        px[:, :, ::64, :] = 0
    encoded = {"jpeg": [], "png": [], "bmp": [], "gif": [], "webp": [], "tiff": []}
    wanted = ("jpeg", "png") if args.format == "mixed" else ("jpeg", "png", "bmp", "webp", "tiff") if args.format == "collection" else (args.format,)
    mix = ["jpeg"] * 70 + ["png"] * 20 + ["bmp"] * 4 + ["webp"] * 3 + ["tiff"] * 3
    np.random.default_rng(5).shuffle(mix)
    for k in range(distinct):
        for fmt, kw in (("jpeg", {"quality": 85, "subsampling": 2}), ("png", {}), ("bmp", {}), ("gif", {}), ("webp", {"quality": 85, "method": 0}), ("tiff", {})):
            if fmt not in wanted:
                continue
            b = io.BytesIO()
            Image.fromarray(px[k]).save(b, fmt.upper(), **kw)
            encoded[fmt].append(b.getvalue())
    root = tempfile.mkdtemp(prefix="ke_fastsig_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        items, nbytes = [], 0
        for i in range(args.images):
            fmt = ("jpeg", "png")[i & 1] if args.format == "mixed" else mix[i % 100] if args.format == "collection" else args.format
            path = os.path.join(root, f"f{i:07d}.{'jpg' if fmt == 'jpeg' else fmt}")
            data = encoded[fmt][i % distinct]
            with open(path, "wb") as fh:
                fh.write(data)
            nbytes += len(data)
            items.append((i + 1, path))
        db = os.path.join(root, "sig.db")
        with sqlite3.connect(db) as conn:
            conn.execute("CREATE TABLE signatures (file_id INTEGER PRIMARY KEY, phash_u64 INTEGER NOT NULL, dhash_u64 INTEGER NOT NULL)")
        # first call: includes growing the context's page-locked packing buffer and its device decode buffer to the batch size
        # (hipHostMalloc / hipMalloc of gigabytes: 0.3-0.7 s, once per process); the second call is what a large library sees
        t0 = time.perf_counter()
        fastsig.fast_fill_missing_signatures(db, items, apply_to_db=False)
        t_first = time.perf_counter() - t0
        t0 = time.perf_counter()
        rows = fastsig.fast_fill_missing_signatures(db, items)
        t_gpu = time.perf_counter() - t0
        assert len(rows) == args.images
        with sqlite3.connect(db) as conn:
            assert conn.execute("SELECT COUNT(*) FROM signatures").fetchone()[0] == args.images
        sample = items[: args.pillow_sample]
        os.environ["KE_GPU_JPEG"] = os.environ["KE_GPU_PNG"] = os.environ["KE_GPU_BMP"] = os.environ["KE_GPU_GIF"] = os.environ["KE_GPU_TIFF"] = "0"
        fastsig.fast_fill_missing_signatures(db, sample[:256], apply_to_db=False)
        t0 = time.perf_counter()
        rows_cpu = fastsig.fast_fill_missing_signatures(db, sample, apply_to_db=False)
        t_pil = time.perf_counter() - t0
        assert rows_cpu == rows[: len(sample)]                     # the same hashes by either route
        print(json.dumps({"case": "fast_fill_missing_signatures", "format": args.format, "content": args.content, "images": args.images,
                          "side": s, "file_mb": nbytes / 1e6, "first_call_wall_ms": t_first * 1e3, "gpu_decode_wall_ms": t_gpu * 1e3, "gpu_decode_images_per_s": args.images / t_gpu,
                          "pillow_route_images": len(sample), "pillow_route_wall_ms": t_pil * 1e3,
                          "pillow_route_images_per_s": len(sample) / t_pil, "decode_threads": max(1, fastsig._usable_cpus() - 1),
                          "same_rows": True}))
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
