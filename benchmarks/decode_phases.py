#!/usr/bin/env python3
"""Where a decode-inclusive call spends its wall clock: pack -> probe -> ke_jpeg_decode (host parse, H2D, kernels) -> ke_hash_images.
One JSON line per batch size (JPEG 512x512 q85 4:2:0, 4096 distinct files)."""
import io, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from concurrent.futures import ThreadPoolExecutor
from PIL import Image
from kobato_eyes_amd import _native
import ctypes as C

ctx = _native.get_context(0)
side, distinct = 512, 4096
src = ctx.synth_rgb(20260604, 0, distinct, side, side)
def enc(k):
    b = io.BytesIO(); Image.fromarray(src[k]).save(b, "JPEG", quality=85, subsampling=2); return b.getvalue()
with ThreadPoolExecutor(16) as ex:
    files = list(ex.map(enc, range(distinct)))
del src
for n in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "16384,65536").split(",")]:
    blobs = [files[k % distinct] for k in range(n)]
    ctx.jpeg_hash(blobs, want_dhash=True)
    lib, h = ctx._lib, ctx._h
    for rep in range(2):
        t = {}
        t0 = time.perf_counter()
        with ctx._lock:
            flat, offsets, sizes = ctx._pack_blobs_pinned(blobs)
            t["pack"] = time.perf_counter() - t0; t1 = time.perf_counter()
            w, hh, c, st = (np.zeros(n, np.int32) for _ in range(4))
            lib.ke_jpeg_probe(_native._addr(flat), _native._addr(offsets), _native._addr(sizes), n, _native._addr(w), _native._addr(hh), _native._addr(c), _native._addr(st))
            t["probe"] = time.perf_counter() - t1; t1 = time.perf_counter()
            nbytes = np.where(st == 0, w.astype(np.int64) * hh * c, 0); padded = (nbytes + 15) & ~np.int64(15)
            out_off = np.zeros(n, np.uint64); out_off[1:] = np.cumsum(padded[:-1]).astype(np.uint64)
            dev = ctx._decoded_ptr
            t["layout"] = time.perf_counter() - t1; t1 = time.perf_counter()
            ctx._check(lib.ke_jpeg_decode(h, _native._addr(flat), _native._addr(offsets), _native._addr(sizes), n, dev, _native._addr(out_off), _native._addr(st)), "dec")
            t["decode_call"] = time.perf_counter() - t1; t["decode_kernels_ms"] = ctx.last_kernel_ms(4); t1 = time.perf_counter()
            p = np.zeros(n, np.uint64); d = np.zeros(n, np.uint64); status = np.zeros(n, np.int32)
            ws, hs = np.ascontiguousarray(w), np.ascontiguousarray(hh)
            ctx._check(lib.ke_hash_images(h, dev, _native._addr(out_off), _native._addr(ws), _native._addr(hs), 3, n, _native._addr(p), _native._addr(d), _native._addr(status)), "hash")
            t["hash_call"] = time.perf_counter() - t1; t["hash_kernel_ms"] = ctx.last_kernel_ms(0)
        t["total"] = time.perf_counter() - t0
    print(json.dumps({"n": n, "compressed_mb": float(sizes.sum()) / 1e6, **{k: round(v * (1 if k.endswith("_ms") else 1e3), 2) for k, v in t.items()}, "unit": "ms"}), flush=True)
