#!/usr/bin/env python3
"""GPU decoders of the small formats against their CPU builds (which the CPU suite holds against Pillow) under damage, at scale:
    python tests/fuzz_formats_gpu.py [variants per file, default 150] [seed]
For GIF, BMP and TIFF: every pool file damaged `variants` times (the generators of tests/test_<fmt>_cpu.py), decoded in one call
per format on the GPU and one by one by the CPU build; status and pixels must agree file by file.  Prints one line per format;
exit code 1 on any difference.  Not part of the suites (tens of thousands of files); tests/test_gpu_jpeg.py runs a sample."""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import _bmp_cases as B
import _gif_cases as G
import _tiff_cases as T
import test_bmp_cpu as TB
import test_gif_cpu as TG
import test_tiff_cpu as TT


def main():
    variants = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from kobato_eyes_amd import _native

    ctx = _native.get_context(0)
    bad = 0
    for fmt, cases, mod in (("gif", list(G.supported(full=True)) + list(G.handmade()), TG), ("bmp", list(B.supported()) + list(B.handmade(full=True)), TB),
                            ("tiff", list(T.supported()) + list(T.handmade()), TT)):
        pool = [c for c in cases if c[2] is not None and c[2].size <= 30000]
        rng = np.random.default_rng(seed)
        files = list(mod.damaged(rng, pool, variants))
        lib = mod._lib()
        out, status = getattr(ctx, f"{fmt}_decode")([d for _, d in files])
        taken = wrong = 0
        for (name, data), px, st in zip(files, out, status):
            cst, cpx = mod._decode(lib, data)
            same = int(st) == int(cst) and ((px is None and (cpx is None or cst != 0)) or (px is not None and cpx is not None and px.shape == cpx.shape and np.array_equal(px, cpx)))
            taken += int(st == 0)
            if not same:
                wrong += 1
                if wrong <= 5:
                    print(f"  {fmt}: {name}: GPU status {st}, CPU build {cst}")
        print(f"{fmt}: {len(files)} damaged files, {taken} taken, {wrong} differences from the CPU build")
        bad += wrong
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
