#!/usr/bin/env python3
"""Wall clock of the scanner seam -- kobato_eyes_amd.DuplicateScanner.build_clusters, Python objects in, clusters out -- beside
the reference's own figure for the same call (1.52 s at N = 100 000 uniform hashes, one core).

    python benchmarks/bench_scanner.py [--sizes 100000,1000000] [--reps 3]

One JSON line per (N, configuration): total seconds (median of --reps after one warm-up) and where they go: the three
np.fromiter passes over the DuplicateFile objects, ke_hamming_scan (upload + kernels + edges back), the edge dictionary with
the reference's funnel counters, cluster assembly (union-find + keeper + ordering)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="100000,1000000")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--funnel", action="store_true", help="also read the funnel counters (scanner.last_counters) inside the timed call")
    args = ap.parse_args()
    import kobato_eyes_amd as K
    from kobato_eyes_amd import _native, scanner as S

    ctx = _native.get_context(0)
    for n in (int(v) for v in args.sizes.split(",")):
        hashes = ctx.synth_hashes(20260604, n)
        files = [K.DuplicateFile(file_id=i + 1, path=Path(f"img_{i:07d}.png"), size=1000 + (i % 7), width=512, height=512,
                                 phash=h, embedding=None) for i, h in enumerate(hashes.tolist())]
        for label, cfg_kw, cap in (("T=8", {"hamming_threshold": 8}, None),
                                   ("T=8 size_ratio=0.9985", {"hamming_threshold": 8, "size_ratio": 0.9985}, None),
                                   ("T=8 KE_DUP_BUCKET_PAIR_CAP=100", {"hamming_threshold": 8}, 100)):
            if cap is None:
                os.environ.pop("KE_DUP_BUCKET_PAIR_CAP", None)
            else:
                os.environ["KE_DUP_BUCKET_PAIR_CAP"] = str(cap)
            scanner = K.DuplicateScanner(K.DuplicateScanConfig(**cfg_kw))
            split = {}

            def timed(obj, name, key):
                inner = getattr(obj, name)

                def wrapper(*a, **kw):
                    t0 = time.perf_counter()
                    try:
                        return inner(*a, **kw)
                    finally:
                        split[key] = split.get(key, 0.0) + time.perf_counter() - t0
                return inner, wrapper

            totals, splits, clusters = [], [], None
            for rep in range(args.reps + 1):
                split.clear()
                saved = []
                for obj, name, key in ((ctx, "hamming_scan", "ke_hamming_scan"), (scanner, "_edges_from_raw", "edge_dict"),
                                       (S, "assemble_clusters", "assemble")):
                    inner, wrapper = timed(obj, name, key)
                    saved.append((obj, name, inner))
                    setattr(obj, name, wrapper)
                t0 = time.perf_counter()
                clusters = scanner.build_clusters(files)
                if args.funnel:
                    _ = scanner.last_counters
                total = time.perf_counter() - t0
                for obj, name, inner in saved:
                    if obj is S:
                        setattr(obj, name, inner)
                    else:
                        delattr(obj, name)                       # instance attribute shadowing the method
                if rep:
                    totals.append(total)
                    splits.append(dict(split, arrays=total - sum(split.values())))
            k = int(np.argsort(totals)[len(totals) // 2])
            print(json.dumps({"case": "build_clusters", "n": n, "config": label, "seconds": totals[k],
                              "split": {a: round(b, 4) for a, b in splits[k].items()}, "clusters": len(clusters),
                              "counters": scanner.last_counters, "funnel_in_timed_call": bool(args.funnel),
                              "reference_seconds_at_100k": 1.52 if n == 100000 else None}), flush=True)
        os.environ.pop("KE_DUP_BUCKET_PAIR_CAP", None)
        del files


if __name__ == "__main__":
    main()
