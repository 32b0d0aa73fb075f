"""Helpers that load the committed golden vectors (tests/golden/, produced by make_golden.py)."""
from __future__ import annotations

import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sig_cases():
    """Yield (name, pixels, tile32, tile98, phash, dhash, margin) for every signature vector."""
    from oracle import oracle as O

    g = np.load(os.path.join(GOLDEN, "sig_golden.npz"))
    for k, name in enumerate(g["names"]):
        name = str(name)
        if str(g["kind"][k]) == "synth":
            i, w, h = (int(v) for v in g["params"][k])
            px = O.synth_rgb(i, w, h)
        else:
            px = g["px_" + name]
        yield name, px, g["tile32"][k], g["tile98"][k], int(g["phash"][k]), int(g["dhash"][k]), float(g["margin"][k]), str(g["sha256"][k])


def scan_scenarios():
    with open(os.path.join(GOLDEN, "scan_golden.json")) as fh:
        data = json.load(fh)
    out = {}
    for name, sc in data.items():
        cols = sc["files"]
        n = len(cols["file_id"])
        files = [{k: (int(cols[k][i]) if k == "phash" else cols[k][i]) for k in cols} for i in range(n)]
        out[name] = dict(sc, files=files)
    return out


def rows_golden():
    with open(os.path.join(GOLDEN, "rows_golden.json")) as fh:
        data = json.load(fh)

    def dec(r):
        out = {}
        for k, v in r.items():
            if isinstance(v, dict) and "__bytes__" in v:
                v = bytes.fromhex(v["__bytes__"])
            elif isinstance(v, str) and k in ("sig", "phash_u64") and v.lstrip("-").isdigit():
                v = int(v)
            out[k] = v
        return out

    return [dec(r) for r in data["rows"]], data["expected"], [dec(r) for r in data["bad_rows"]]


def ssim_cases():
    g = np.load(os.path.join(GOLDEN, "ssim_golden.npz"))
    for k, name in enumerate(g["names"]):
        name = str(name)
        yield name, g["a_" + name], g["b_" + name], float(g["ssim"][k])


def files_to_arrays(files):
    """(hashes u64, ids i64, sizes i64 with None->0) in list order."""
    hashes = np.array([f["phash"] & 0xFFFFFFFFFFFFFFFF for f in files], dtype=np.uint64)
    ids = np.array([f["file_id"] for f in files], dtype=np.int64)
    sizes = np.array([(f["size"] or 0) for f in files], dtype=np.int64)
    return hashes, ids, sizes


def refine_parallel_golden():
    with open(os.path.join(GOLDEN, "refine_parallel_golden.json")) as fh:
        return json.load(fh)


def refine_corpus():
    """The same (name, pixels) list make_golden.py wrote to PNG files for the reference."""
    from oracle import oracle as O

    items = []
    for i in (7, 19, 17, 29, 15, 39, 0, 1):
        items.append((f"v{i:03d}_256", O.synth_rgb(i, 256, 256)))
    for (i, w, h) in [(2, 300, 451), (3, 512, 512), (4, 64, 48), (5, 1000, 37), (6, 33, 200), (8, 16, 16)]:
        items.append((f"s{i:03d}_{w}x{h}", O.synth_rgb(i, w, h)))
    rng = np.random.default_rng(21)
    items.append(("gray_L", rng.integers(0, 256, (90, 120), dtype=np.uint8)))
    items.append(("rgba", rng.integers(0, 256, (77, 91, 4), dtype=np.uint8)))
    shifted = O.synth_rgb(7, 256, 256).astype(np.int16)
    shifted[:, :, :] += 2
    items.append(("v007_plus2", np.clip(shifted, 0, 255).astype(np.uint8)))
    return items
