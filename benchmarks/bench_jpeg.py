#!/usr/bin/env python3
"""GPU JPEG decode (+ hash) throughput beside Pillow on the host cores: N baseline JPEG files of side x side pixels (256 distinct
synthetic-corpus images encoded by Pillow at quality 85, 4:2:0, repeated to N), one ke_jpeg_decode call per batch.
    python benchmarks/bench_jpeg.py [--images 16384 --side 512 --batch 16384]
One JSON line: decode kernels (HIP events), decode + hash wall time (compressed bytes start in host memory), Pillow decode on
the usable cores."""
from __future__ import annotations

import argparse
import io
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=16384)
    ap.add_argument("--side", type=int, default=512)
    ap.add_argument("--quality", type=int, default=85)
    ap.add_argument("--subsampling", type=int, default=2)
    ap.add_argument("--format", choices=["jpeg", "png", "gif", "bmp"], default="jpeg")
    ap.add_argument("--distinct", type=int, default=4096,
                    help="distinct images encoded (the batch cycles through them; few copies of each, so that the lanes of a wave hold different images)")
    ap.add_argument("--progressive", action="store_true", help="JPEG: progressive files (libjpeg's default scan script)")
    ap.add_argument("--interlaced", action="store_true", help="PNG: Adam7 files (Pillow has no writer for them: built here, Sub filter on every row)")
    ap.add_argument("--content", choices=["corpus", "drawing"], default="corpus",
                    help="corpus: the synthetic corpus images (textured, photograph-like: PNG stays near half its raw size); "
                         "drawing: flat 16-px cells, a few ramps and thin outlines (illustration-like: PNG shrinks 20-50x)")
    args = ap.parse_args()
    from PIL import Image

    from kobato_eyes_amd import _native

    ctx = _native.Context(0)
    distinct = min(args.distinct, args.images)
    px = ctx.synth_rgb(20260604, 0, distinct, args.side, args.side)
    if args.content == "drawing":
        rng = np.random.default_rng(11)
        s = args.side
        cells = rng.integers(0, 256, (distinct, s // 16 + 1, s // 16 + 1, 3), dtype=np.uint8)
        px = np.repeat(np.repeat(cells, 16, 1), 16, 2)[:, :s, :s].copy()
        ramp = (np.arange(s) * 255 // max(s - 1, 1)).astype(np.uint8)
        px[:, s // 3: s // 2, :, 1] = ramp[None, None, :]                 # a band with a horizontal ramp in one channel
        px[:, ::48, :, :] = 0                                             # outlines
        px[:, :, ::64, :] = 0
    def adam7(a):
        import struct
        import zlib

        raw = bytearray()
        for (x0, y0, dx, dy) in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = a[y0::dy, x0::dx]
            if sub.shape[0] == 0 or sub.shape[1] == 0:
                continue
            rows = sub.reshape(sub.shape[0], -1).astype(np.int16)
            f = rows.copy()
            f[:, 3:] -= rows[:, :-3]
            raw += np.concatenate([np.ones((rows.shape[0], 1), np.uint8), (f & 255).astype(np.uint8)], 1).tobytes()

        def ch(t, d):
            return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))

        h, w = a.shape[:2]
        return b"\x89PNG\r\n\x1a\n" + ch(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 1)) + ch(b"IDAT", zlib.compress(bytes(raw), 6)) + ch(b"IEND", b"")

    def encode(k):
        b = io.BytesIO()
        if args.format == "png" and args.interlaced:
            return adam7(px[k])
        if args.format in ("gif", "bmp"):
            Image.fromarray(px[k]).save(b, args.format.upper())
        elif args.format == "png":
            Image.fromarray(px[k]).save(b, "PNG")
        else:
            Image.fromarray(px[k]).save(b, "JPEG", quality=args.quality, subsampling=args.subsampling, progressive=args.progressive)
        return b.getvalue()

    with ThreadPoolExecutor(16) as ex:
        files = list(ex.map(encode, range(distinct)))
    blobs = [files[k % distinct] for k in range(args.images)]
    comp_bytes = sum(len(b) for b in blobs)
    ctx.jpeg_hash(blobs[:512], kind=args.format)                 # warm-up: allocations, tables
    t_wall, t_kern = [], []
    for _ in range(3):
        t0, k0 = time.perf_counter(), ctx.decode_kernel_ms
        ph, dh, st = ctx.jpeg_hash(blobs, want_dhash=False, kind=args.format)
        t_wall.append(time.perf_counter() - t0)
        t_kern.append(ctx.decode_kernel_ms - k0)                 # every decode call of the batch (one beyond the byte limits is halved)
    assert (st == 0).all()
    # Pillow on the host cores
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        cores = os.cpu_count() if quota == "max" else max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        cores = os.cpu_count() or 1
    sample = blobs[: max(256, 64 * cores)]

    def dec(b):
        with Image.open(io.BytesIO(b)) as im:
            im.load()
        return 1

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(dec, sample, chunksize=8))
    t_cpu = time.perf_counter() - t0
    wall, kern = float(np.median(t_wall)), float(np.median(t_kern))
    print(json.dumps({"case": args.format + ("_progressive" if args.progressive else "") + ("_adam7" if args.interlaced else "") + "_decode", "content": args.content, "distinct": distinct, "images": args.images, "side": args.side, "quality": args.quality,
                      "subsampling": ["4:4:4", "4:2:2", "4:2:0"][args.subsampling], "compressed_mb": comp_bytes / 1e6,
                      "decode_kernels_ms": kern, "decode_images_per_s": args.images / (kern * 1e-3),
                      "decode_plus_hash_wall_ms": wall * 1e3, "decode_plus_hash_images_per_s": args.images / wall,
                      "pillow_threads": cores, "pillow_images_per_s": len(sample) / t_cpu}))


if __name__ == "__main__":
    main()
