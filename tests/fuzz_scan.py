#!/usr/bin/env python3
"""Randomised sweep of ke_hamming_scan against the oracle's reference-shaped banded scan (test infrastructure):
random table sizes around the 16- and 1024-hash tile edges, thresholds 0..64, band shapes, duplicate ids, size-ratio
filter, bucket cap, shard counts.  python tests/fuzz_scan.py [cases] [seed]; exits non-zero on the first mismatch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from kobato_eyes_amd import _native
from oracle import oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = _native.Context(0)
key = lambda e, cols=("a", "b", "h", "bands"): sorted(map(tuple, e[list(cols)].tolist()))
bad = 0
for k in range(cases):
    n = int(rng.choice([rng.integers(1, 40), rng.integers(1000, 1050), rng.integers(2040, 2060), rng.integers(40, 3500), 1024, 2048, 16, 17]))
    base = rng.integers(0, 2**63, max(1, n // 6 + 1), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, max(1, n // 6 + 1), dtype=np.uint64)
    h = base[rng.integers(0, len(base), n)].copy()
    flips = rng.integers(0, 14, n)
    for i in range(n):                                   # near-duplicates of a few bases: dense candidate structure
        for b in rng.integers(0, 64, flips[i]):
            h[i] ^= np.uint64(1) << np.uint64(b)
    t = int(rng.choice([0, 1, 4, 8, 8, 12, 20, 64]))
    bb, bc = [(16, 4), (8, 8), (32, 2), (4, 16), (1, 64), (64, 1), (13, 4), (21, 3)][int(rng.integers(0, 8))]
    ids = rng.integers(0, max(2, n // 2), n).astype(np.int64) if rng.integers(0, 3) == 0 else None
    sizes = rng.integers(0, 5000, n).astype(np.int64) if rng.integers(0, 2) == 0 else None
    ratio = float(rng.choice([0.0, 0.5, 0.9])) if sizes is not None else 0.0
    cap = int(rng.choice([0, 0, 3, 50])) if bb <= 24 else 0
    # the kernel reports positions and drops pairs of equal file id (src/dup/scanner.py:266); folding equal-id pairs into one
    # id-keyed edge is host logic (scanner._edges_from_raw, covered by the drop-in tests), so the truth here is positional
    exp, exp_c = O.scan_banded(h, ids=None, sizes=sizes, threshold=t, band_bits=bb, band_count=bc, size_ratio=ratio, bucket_pair_cap=cap)
    if ids is not None:
        exp = exp[ids[exp["a"]] != ids[exp["b"]]]
    parts = int(rng.choice([1, 1, 2, 3, 8]))
    got = [ctx.hamming_scan(h, n, ids=ids, sizes=sizes, threshold=t, band_bits=bb, band_count=bc, size_ratio=ratio, bucket_pair_cap=cap,
                            part_index=p, part_count=parts, capacity=int(rng.choice([64, 1 << 16]))) for p in range(parts)]
    edges = np.concatenate([g[0] for g in got])
    pairs = sum(int(g[1][0]) for g in got)
    cols = ("a", "b", "h", "bands") if bc <= 31 else ("a", "b", "h")      # the bands mask is an int32: bands 31.. share bit 31
    if key(edges, cols) != key(exp, cols) or pairs != n * (n - 1) // 2:
        bad += 1
        print("MISMATCH", dict(n=n, t=t, bb=bb, bc=bc, ids=ids is not None, sizes=sizes is not None, ratio=ratio, cap=cap, parts=parts,
                               got=len(edges), exp=len(exp), pairs=pairs), flush=True)
    if (k + 1) % 25 == 0:
        print(k + 1, "cases,", bad, "mismatches", flush=True)
print("done:", cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
