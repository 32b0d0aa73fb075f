#!/usr/bin/env python3
"""Randomised sweep of the hashing dispatch against the CPU oracle (every kernel family: single-pass exact / run-time
row length / wide rows / RGBA, banded, generic): python tests/fuzz_shapes.py [cases] [seed] (test infrastructure: it checks the GPU library against oracle/).  Exits non-zero on
the first mismatch.  KE_FUSED_MIN_IMAGES=1 so that a few images reach the single-pass kernels."""
import os
import sys

os.environ.setdefault("KE_FUSED_MIN_IMAGES", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from kobato_eyes_amd import _native
from oracle import oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = _native.Context(0)
bad = 0
for k in range(cases):
    kind = rng.integers(0, 6)
    if kind == 0:
        w = int(rng.integers(17, 193)) * 4          # multiple of 4 up to 768
    elif kind == 1:
        w = int(rng.integers(177, 513)) * 4         # 708..2048
    elif kind == 2:
        w = int(rng.choice([256, 384, 512, 640, 768, 1024, 1536, 2048]))
    elif kind == 3:
        w = int(rng.integers(1, 2600))              # anything
    elif kind == 4:
        w = int(rng.integers(513, 1100)) * 4        # beyond the single-pass kernels
    else:
        w = int(rng.integers(16, 64)) * 4
    h = int(rng.choice([rng.integers(1, 40), rng.integers(16, 700), rng.integers(700, 2200), rng.integers(2200, 4400)]))
    if w * h > 6_000_000:
        h = max(1, 6_000_000 // w)
    ch = int(rng.choice([1, 3, 3, 3, 4]))
    n = int(rng.integers(1, 4))
    px = rng.integers(0, 256, (n, h, w) if ch == 1 else (n, h, w, ch), dtype=np.uint8)
    if rng.integers(0, 3) == 0:
        px[0] = (rng.integers(0, 2, px[0].shape) * 255).astype(np.uint8)
    both = bool(rng.integers(0, 2))
    got_p, got_d = ctx.hash_uniform(px, n, w, h, ch, want_dhash=both)
    for j in range(n):
        ep, ed = O.hash_image(px[j])[:2]
        if int(got_p[j]) != ep or (both and int(got_d[j]) != ed):
            bad += 1
            print("MISMATCH", dict(w=w, h=h, ch=ch, n=n, both=both, image=j), flush=True)
    if (k + 1) % 50 == 0:
        print(k + 1, "cases,", bad, "mismatches", flush=True)
# ragged batches: images of many shapes in one ke_hash_images call (offsets, per-shape groups, output scatter)
for k in range(max(1, cases // 50)):
    imgs = []
    for _ in range(int(rng.integers(5, 40))):
        w = int(rng.choice([rng.integers(1, 300), rng.integers(17, 513) * 4, rng.integers(513, 1100) * 4, 512, 640, 1024]))
        h = int(rng.choice([rng.integers(1, 64), rng.integers(16, 900), rng.integers(900, 2400)]))
        if w * h > 3_000_000:
            h = max(1, 3_000_000 // w)
        imgs.append(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
        if rng.integers(0, 4) == 0:
            imgs.append(imgs[-1].copy())          # a second image of the same shape: groups of more than one
    got_p, got_d, st = ctx.hash_images(imgs)
    for j, im in enumerate(imgs):
        ep, ed = O.hash_image(im)[:2]
        if st[j] != 0 or int(got_p[j]) != ep or int(got_d[j]) != ed:
            bad += 1
            print("MISMATCH ragged", k, j, im.shape, flush=True)
print("done:", cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
