#!/usr/bin/env python3
"""Progressive JPEG files, clean and damaged (test infrastructure): the kernels (mask-driven refinement scans, blocks in zigzag
order) against the host build of the reference-shaped decoder (oracle/libkeyes_jpeg_cpu.so) -- the same status for every file and
the same pixels for every file taken.  18 000 damaged variants of 600 files per run.
    python tests/fuzz_jpeg_progressive_parity.py      (needs the GPU; exits non-zero on a mismatch)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fuzz_jpeg_damage as F, _jpeg_cases as J
from kobato_eyes_amd import _native
ctx = _native.get_context(0)
cpu = F.cpu_decoder()
pool = [c for c in J.supported(full=True) if "progressive" in c[0] or "_p1" in c[0]]
print("progressive files", len(pool), flush=True)
# clean files first: GPU pixels == CPU pixels == Pillow
out, st = ctx.jpeg_decode([c[1] for c in pool])
assert (np.asarray(st) == 0).all() and all(np.array_equal(a, c[2]) for a, c in zip(out, pool))
rng = np.random.default_rng(2024)
bad = tot = taken = 0
for rnd in range(3):
    sub = [pool[i] for i in rng.choice(len(pool), 60, replace=False)]
    for kind in ("body", "header"):
        batch = list(F.damaged(rng, sub, 50, kind))
        blobs = [b for _, _, b in batch]
        out, st = ctx.jpeg_decode(blobs)
        cout, cst = cpu(blobs)
        for (name, what, _), a, s, ca, cs in zip(batch, out, np.asarray(st).tolist(), cout, cst):
            tot += 1
            if s != cs or (s == 0 and not np.array_equal(a, ca)):
                bad += 1
                if bad < 10: print("MISMATCH", name, what, "gpu", s, "cpu", cs, flush=True)
            taken += s == 0
print("cases", tot, "taken", taken, "bad", bad)
sys.exit(1 if bad else 0)
