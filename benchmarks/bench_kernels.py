#!/usr/bin/env python3
"""Per-kernel measurements beside bench.py's headline (one JSON line per case):
hash kernels at several shapes (fused and generic paths, with/without dHash), the scan at
N = 100k / 1M (BASELINE configs 2-4), SSIM pairs at 512x512 (config 4), and the PCIe-inclusive
rate of the host-buffer path.  Timings are HIP-event kernel times from ke_last_kernel_ms (median
of --reps) unless a case says wall."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SEED = 20260604


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--cases", default="hash,scan,ssim,pcie")
    args = ap.parse_args()
    import torch

    from kobato_eyes_amd import _native

    ctx = _native.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    dev = torch.device("cuda", 0)
    cases = set(args.cases.split(","))

    def emit(**kw):
        print(json.dumps(kw), flush=True)

    if "hash" in cases:
        for (w, h, n, dh) in [(512, 512, 20000, False), (512, 512, 20000, True), (256, 256, 80000, False), (256, 256, 80000, True),
                              (384, 384, 30000, False), (512, 1024, 10000, False), (1024, 1024, 2000, False),
                              (640, 480, 4000, False), (2048, 2048, 300, False), (4096, 4096, 60, False)]:
            px = torch.empty(n * w * h * 3, dtype=torch.uint8, device=dev)
            ctx.synth_rgb(SEED, 0, n, w, h, out=px.data_ptr())
            ph = torch.empty(n, dtype=torch.int64, device=dev)
            dhash = torch.empty(n, dtype=torch.int64, device=dev)
            ms = []
            for _ in range(args.reps):
                ctx.hash_uniform(px.data_ptr(), n, w, h, 3, phash_out=ph.data_ptr(), dhash_out=dhash.data_ptr() if dh else None,
                                 want_dhash=dh)
                ms.append(ctx.last_kernel_ms(0))
            t = float(np.median(ms)) * 1e-3
            emit(case="hash", w=w, h=h, n=n, dhash=dh, ms=t * 1e3, images_per_s=n / t, gbs=n * (3 * w * h + 8) / t / 1e9,
                 frac_hbm=n * (3 * w * h + 8) / t / 8e12)
            del px
    if "scan" in cases:
        for n in (100_000, 1_000_000):
            hs = torch.empty(n, dtype=torch.int64, device=dev)
            ctx.synth_hashes(SEED, n, out=hs.data_ptr())
            ms, ne = [], 0
            for _ in range(args.reps if n <= 100_000 else 3):
                e, c = ctx.hamming_scan(hs.data_ptr(), n, threshold=8, capacity=4 * n)
                ms.append(ctx.last_kernel_ms(1))
                ne = len(e)
            t = float(np.median(ms)) * 1e-3
            pairs = n * (n - 1) // 2
            emit(case="scan", n=n, ms=t * 1e3, edges=ne, gpairs_per_s=pairs / t / 1e9, hbm_convention_gbs=pairs * 16 / t / 1e9,
                 frac_hbm_convention=pairs * 16 / t / 8e12, mfma_fp4_tflops=pairs * 256 / t / 1e12, frac_mfma_fp4=pairs * 256 / t / 1e16)
    if "ssim" in cases:
        for (w, h, n_img, n_pairs) in [(512, 512, 4000, 20000), (256, 256, 8000, 40000)]:
            px = torch.empty(n_img * w * h * 3, dtype=torch.uint8, device=dev)
            ctx.synth_rgb(SEED, 0, n_img, w, h, out=px.data_ptr())
            rng = np.random.default_rng(0)
            pa, pb = rng.integers(0, n_img, n_pairs), rng.integers(0, n_img, n_pairs)
            ms = []
            for _ in range(args.reps):
                ctx.ssim_pairs_uniform(px.data_ptr(), n_img, w, h, 3, pa, pb)
                ms.append(ctx.last_kernel_ms(2))
            t = float(np.median(ms)) * 1e-3
            emit(case="ssim", w=w, h=h, pairs=n_pairs, ms=t * 1e3, pairs_per_s=n_pairs / t, gbs=n_pairs * (6 * w * h + 8) / t / 1e9,
                 frac_hbm=n_pairs * (6 * w * h + 8) / t / 8e12)
            del px
    if "pcie" in cases:
        n, w, h = 2000, 512, 512
        host = ctx.synth_rgb(SEED, 0, n, w, h)       # pageable host memory
        ctx.hash_uniform(host, n, w, h, 3, want_dhash=False)
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.hash_uniform(host, n, w, h, 3, want_dhash=False)
        t = (time.perf_counter() - t0) / 3
        emit(case="pcie_inclusive_hash", w=w, h=h, n=n, wall_ms=t * 1e3, images_per_s=n / t, gbs=n * 3 * w * h / t / 1e9,
             note="host numpy buffer -> ke_hash_uniform (staged H2D copy + kernel + D2H of hashes), wall clock")


if __name__ == "__main__":
    main()
