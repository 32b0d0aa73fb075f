#!/usr/bin/env python3
"""Scan kernel alone at several table sizes, back to back (HIP-event kernel times of ke_hamming_scan on device-resident
synthetic hash tables): where the per-tile rate of the large tables is lost at N = 100 000.
    python benchmarks/scan_sizes.py [N ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from kobato_eyes_amd import _native  # noqa: E402

SEED = 20260604
ctx = _native.Context(0)
sizes = [int(a) for a in sys.argv[1:]] or [20_000, 50_000, 100_000, 200_000, 400_000, 1_000_000]
for n in sizes:
    d = ctx.malloc(n * 8)
    ctx.synth_hashes(SEED, n, out=d)
    ms = []
    for _ in range(12):
        edges, counters = ctx.hamming_scan(d, n, threshold=8)
        ms.append(ctx.last_kernel_ms(1))
    ctx.free(d)
    med = float(np.median(ms[2:]))
    pairs = n * (n - 1) // 2
    print(json.dumps({"n": n, "scan_ms_median": med, "scan_ms_min": float(min(ms)), "edges": int(len(edges)),
                      "tpairs_per_s": pairs / (med * 1e-3) / 1e12, "frac_of_fp4_peak": pairs * 256 / (med * 1e-3) / 1e16}), flush=True)
