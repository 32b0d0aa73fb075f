#!/usr/bin/env python3
"""cProfile of fast_fill_missing_signatures over N JPEG files in /dev/shm (second call): where the interpreter's share goes."""
import cProfile, io, os, pstats, shutil, sqlite3, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image
from kobato_eyes_amd import _native, fastsig
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
ctx = _native.get_context(0)
px = ctx.synth_rgb(20260604, 0, 128, 512, 512)
enc = []
for k in range(128):
    b = io.BytesIO(); Image.fromarray(px[k]).save(b, "JPEG", quality=85, subsampling=2); enc.append(b.getvalue())
root = tempfile.mkdtemp(prefix="ke_prof_", dir="/dev/shm")
try:
    items = []
    for i in range(n):
        p = os.path.join(root, f"f{i:07d}.jpg")
        with open(p, "wb") as fh: fh.write(enc[i % 128])
        items.append((i + 1, p))
    db = os.path.join(root, "sig.db")
    with sqlite3.connect(db) as conn:
        conn.execute("CREATE TABLE signatures (file_id INTEGER PRIMARY KEY, phash_u64 INTEGER NOT NULL, dhash_u64 INTEGER NOT NULL)")
    fastsig.fast_fill_missing_signatures(db, items, apply_to_db=False)
    t0 = time.perf_counter(); fastsig.fast_fill_missing_signatures(db, items); print("plain wall", time.perf_counter() - t0)
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter(); fastsig.fast_fill_missing_signatures(db, items); wall = time.perf_counter() - t0
    pr.disable()
    print("profiled wall", wall)
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:6000])
finally:
    shutil.rmtree(root, ignore_errors=True)
