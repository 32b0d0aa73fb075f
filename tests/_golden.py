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


def fit_cases():
    """ImageOps.fit + BICUBIC vectors: yields (name, px_a, px_b, (w, h), fitted_a, fitted_b, ssim)."""
    from oracle import oracle as O

    g = np.load(os.path.join(GOLDEN, "fit_golden.npz"))
    for k, name in enumerate(g["names"]):
        name = str(name)
        ia, wa, ha = (int(v) for v in g["pa_" + name])
        ib, wb, hb = (int(v) for v in g["pb_" + name])
        yield (name, O.synth_rgb(ia, wa, ha), O.synth_rgb(ib, wb, hb), tuple(int(v) for v in g["size"][k]),
               g["fa_" + name], g["fb_" + name], float(g["ssim"][k]))


def fit_extra_cases():
    """(luma input, (out_w, out_h), expected) for explicit target sizes, including upscaling."""
    g = np.load(os.path.join(GOLDEN, "fit_golden.npz"))
    for k in range(int(g["n_extra"])):
        yield g[f"xin_{k}"], tuple(int(v) for v in g[f"xsize_{k}"]), g[f"xout_{k}"]


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


def write_image_io_files(td):
    """Files exercising the loader in front of the refine stage; returns [(name, path, kwargs)]."""
    from PIL import Image

    rng = np.random.default_rng(31)
    td = str(td)
    cases = []

    def add(name, img, fmt, kwargs=None, **save):
        path = os.path.join(td, f"{name}.{fmt.lower()}")
        img.save(path, format=fmt, **save)
        cases.append((name, path, kwargs or {}))

    rgb = Image.fromarray(rng.integers(0, 256, (60, 80, 3), dtype=np.uint8))
    add("rgb_png", rgb, "PNG")
    add("rgba_png", Image.fromarray(rng.integers(0, 256, (50, 70, 4), dtype=np.uint8)), "PNG")
    add("la_png", Image.fromarray(rng.integers(0, 256, (40, 30, 2), dtype=np.uint8), mode="LA"), "PNG")
    add("l_png", Image.fromarray(rng.integers(0, 256, (33, 44), dtype=np.uint8)), "PNG")
    pal = rgb.convert("P", palette=Image.Palette.ADAPTIVE, colors=16)
    add("p_png", pal, "PNG")
    add("p_transparent_png", pal, "PNG", transparency=3)
    add("i16_png", Image.fromarray((rng.integers(0, 65536, (20, 25))).astype(np.uint16)), "PNG")
    add("cmyk_jpg", rgb.convert("CMYK"), "JPEG", quality=90)
    exif = Image.Exif()
    exif[0x0112] = 6                                       # orientation: rotate 270 on load
    add("exif_rot_jpg", rgb, "JPEG", quality=95, exif=exif.tobytes())
    big = Image.fromarray(np.repeat(np.repeat(rng.integers(0, 256, (50, 64, 3), dtype=np.uint8), 100, axis=0), 100, axis=1)[:4500, :6000])
    add("big_png", big, "PNG", compress_level=1)
    add("big_jpg", big, "JPEG", quality=85)                # draft mode decodes it reduced
    add("big_jpg_max1000", big, "JPEG", {"max_side": 1000}, quality=85)
    add("hard_skip", rgb, "PNG", {"hard_skip_pixels": 1000})
    add("bomb_skip", rgb, "PNG", {"bomb_pixel_cap": 1000, "skip_on_bomb": True})
    add("bomb_pass", rgb, "PNG", {"bomb_pixel_cap": 1000})
    add("bomb_warn_only", rgb, "PNG", {"bomb_pixel_cap": 3000})
    add("keep_mode", Image.fromarray(rng.integers(0, 256, (50, 70, 4), dtype=np.uint8)), "PNG", {"rgb": False})
    broken = os.path.join(td, "broken.png")
    with open(broken, "wb") as fh:
        fh.write(b"not an image")
    cases.append(("broken", broken, {}))
    trunc_src = os.path.join(td, "rgb_png.png")
    trunc = os.path.join(td, "truncated.png")
    with open(trunc_src, "rb") as fh:
        data = fh.read()
    with open(trunc, "wb") as fh:
        fh.write(data[: len(data) * 2 // 3])
    cases.append(("truncated", trunc, {}))
    cases.append(("missing", os.path.join(td, "does_not_exist.png"), {}))
    return cases


def image_io_golden():
    with open(os.path.join(GOLDEN, "image_io_golden.json")) as fh:
        return json.load(fh)


def degenerate_tiles():
    """(name, pixels, phash, dhash, margin): flat and two-level images, where every AC term of the DCT is exactly zero or
    one of a few large values.  The values are THIS build's policy (folded fp64 DCT: exact zeros for flat and mirror-symmetric
    inputs, strict `>` against the float32 mean), fixed here so it cannot drift: real OpenCV is not available to pin them,
    and for such images its float32 DCT leaves rounding noise where these are exact zeros ("Mixing
    signatures")."""
    def flat(v, h, w):
        return np.full((h, w, 3), v, np.uint8)

    def two(level_a, level_b, mask):
        out = np.full(mask.shape + (3,), level_a, np.uint8)
        out[mask] = level_b
        return out

    yy, xx = np.indices((64, 64))
    table = [
        ("flat0", flat(0, 64, 64), 0x0000000000000000, 0x0000000000000000, 0.0),
        ("flat37", flat(37, 48, 80), 0x8000000000000000, 0x0000000000000000, 0.0),
        ("flat255", flat(255, 512, 512), 0x8000000000000000, 0x0000000000000000, 0.0),
        ("split_lr", two(0, 255, xx >= 32), 0xBBFFFFFFFFFFFFFF, 0x5A5A5A5A5A5A5A5A, 42.24908447265625),
        ("split_tb", two(0, 255, yy >= 32), 0xFF7FFFFFFF7FFFFF, 0x0000000000000000, 42.249080657958984),
        ("quad", two(0, 255, (yy < 32) == (xx < 32)), 0x8044001100440011, 0x242400245A185A5A, 27.564197540283203),
        ("split_lr_1level", two(200, 201, xx >= 32), 0xBBFFFFFFFFFFFFFF, 0x0808080808080808, 0.16512662172317505),
        ("checker32", two(0, 255, ((yy // 32 + xx // 32) % 2) == 1), 0xFFBBFFEEFFBBFFEE, 0x5A5A185A24002424, 27.564197540283203),
    ]
    return table


# ---- the seams end to end (reference-run fixtures: make_golden.py --only-seams)
def config0_golden():
    """BASELINE configs[0] through the reference: {"rows": iter_files_for_dup-shaped dicts without hashes, "phash_s64",
    "dhash_s64", "edges", "counters", "clusters"}."""
    with open(os.path.join(GOLDEN, "config0_golden.json")) as fh:
        g = json.load(fh)
    n, side = g["n"], g["side"]
    g["rows"] = [{"file_id": i + 1, "path": f"img_{i:07d}.png", "size": 1000 + (i % 7), "width": side, "height": side}
                 for i in range(n)]
    g["phash_s64"] = [int(v) for v in g["phash_s64"]]
    g["dhash_s64"] = [int(v) for v in g["dhash_s64"]]
    return g


def scan100k_golden():
    with open(os.path.join(GOLDEN, "scan100k_golden.json")) as fh:
        return json.load(fh)


def scan_listing_digests(edges, clusters):
    """sha256 of the canonical listings, as make_golden.make_scan100k writes them: edges = sorted (a, b, h) with a < b;
    clusters = [[keeper_id, [[file_id, best_hamming], ...]], ...] in the scanner's order."""
    import hashlib

    e = hashlib.sha256(json.dumps(sorted([min(a, b), max(a, b), h] for a, b, h in edges), separators=(",", ":")).encode()).hexdigest()
    c = hashlib.sha256(json.dumps(clusters, separators=(",", ":")).encode()).hexdigest()
    return e, c


def worker_golden():
    """[(name, file bytes | None, file_id, expected row (file_id, phash_s64, dhash_s64) | None, opened_as)] in task order: the
    corpus of make_golden.worker_corpus with what the reference's _compute_worker returned for each file; bytes None = a
    path that is a directory / does not exist."""
    with open(os.path.join(GOLDEN, "worker_golden.json")) as fh:
        rows = json.load(fh)["rows"]
    z = np.load(os.path.join(GOLDEN, "worker_corpus.npz"))
    blobs = {str(n): z[f"f{k}"].tobytes() for k, n in enumerate(z["names"])}
    out = []
    for name, rec in rows.items():
        row = rec["row"]
        out.append((name, blobs.get(name), rec["file_id"], None if row is None else (row[0], int(row[1]), int(row[2])), rec["opened_as"]))
    out.sort(key=lambda t: t[2])
    return out


def write_worker_corpus(td):
    """The corpus as files under td -> [(file_id, path)], expected rows in that order (dropped files left out)."""
    tasks, expected = [], []
    for name, data, fid, row, _ in worker_golden():
        p = os.path.join(str(td), name)
        if data is not None:
            with open(p, "wb") as fh:
                fh.write(data)
        elif name.startswith("a_directory"):
            os.mkdir(p)
        tasks.append((fid, p))
        if row is not None:
            expected.append(row)
    return tasks, expected


def refine_turned_golden():
    """(json, {name: file bytes}): the reference's shipped refine stage run on files the loader has to normalise first
    (JPEG files of every EXIF orientation, RGBA / gray + alpha / 16-bit / Adam7 PNG, BMP, GIF, TIFF; make_golden.make_refine_turned)."""
    with open(os.path.join(GOLDEN, "refine_turned_golden.json")) as fh:
        g = json.load(fh)
    z = np.load(os.path.join(GOLDEN, "refine_turned_corpus.npz"))
    return g, {str(n): z[f"f{k}"].tobytes() for k, n in enumerate(z["names"])}
