#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in the build
container (it cannot travel to the GPU box; only these data files do).

    python tests/golden/make_golden.py        # needs /root/reference

What runs: the reference's own sig.phash (phash/dhash/_to_grayscale, src/sig/phash.py)
and dup.scanner (src/dup/scanner.py) imported from /root/reference/src.  `cv2.dct` is the
SciPy stand-in in _standins/cv2.py (OpenCV absent -> pHash values are labelled
dct_backend=scipy).  dup.refine cannot be imported (cv2 + skimage absent), so the SSIM
vectors come from a scipy.ndimage restatement of skimage's published algorithm and are
labelled as such.
"""
from __future__ import annotations

import hashlib
import json
import logging
import os
import sys
from pathlib import Path

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, "_standins"))
sys.path.insert(0, "/root/reference/src")
sys.path.insert(0, ROOT)

from PIL import Image  # noqa: E402

import sig.phash as ref_sig  # noqa: E402
from dup.scanner import DuplicateFile, DuplicateScanConfig, DuplicateScanner  # noqa: E402
import dup.scanner as ref_scanner  # noqa: E402

from oracle import oracle as O  # noqa: E402  (only for the synthetic generators = inputs)

U64 = (1 << 64) - 1


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ref_hashes(img: Image.Image):
    t32 = np.asarray(ref_sig._to_grayscale(img, (32, 32))).astype(np.uint8)
    t98 = np.asarray(ref_sig._to_grayscale(img, (9, 8))).astype(np.uint8)
    ph = ref_sig.phash(img) & U64
    dh = ref_sig.dhash(img) & U64
    # tie margin of the stand-in DCT, for the record
    import cv2
    flat = cv2.dct(t32.astype(np.float32))[:8, :8].flatten()
    margin = float(np.abs(flat - flat[1:].mean()).min())
    return t32, t98, ph, dh, margin


def make_sig():
    cases = []   # (name, kind, params)
    pixels = {}  # stored pixel arrays for 'stored' kind
    # reference test images: tests/core/test_image_signature.py:24-27
    for seed in list(range(20)) + [123]:
        rng = np.random.default_rng(seed)
        arr = (rng.random((64, 64, 3)) * 255).astype("uint8")
        cases.append((f"ref_random64_seed{seed}", "stored", arr))
    # synthetic corpus images (DESIGN.md "Synthetic data"), BASELINE configs 1, 2 and 5 shapes
    for i in range(24):
        cases.append((f"synth256_{i}", "synth", (i, 256, 256)))
    for i in [0, 1, 9, 19, 29, 99]:
        cases.append((f"synth512_{i}", "synth", (i, 512, 512)))
    for (i, w, h) in [(3, 300, 451), (4, 1000, 37), (5, 33, 31), (6, 16, 16), (7, 32, 32), (8, 384, 768),
                      (9, 1024, 256), (10, 2, 500), (11, 5, 500), (12, 640, 480), (13, 9, 8), (14, 31, 33),
                      (15, 2048, 64), (16, 100, 1536), (17, 7, 7), (18, 1, 1), (19, 36, 36), (20, 510, 514)]:
        cases.append((f"synth{w}x{h}_{i}", "synth", (i, w, h)))
    # flats, ramps, symmetric: the tie cases
    for v in (0, 128, 255):
        cases.append((f"flat{v}", "stored", np.full((64, 64, 3), v, np.uint8)))
    yy, xx = np.indices((96, 128))
    cases.append(("ramp_x", "stored", np.repeat(((xx * 2) % 256).astype(np.uint8)[:, :, None], 3, 2)))
    cases.append(("ramp_y", "stored", np.repeat(((yy * 2) % 256).astype(np.uint8)[:, :, None], 3, 2)))
    cases.append(("checker8", "stored", np.repeat(((((yy // 8) + (xx // 8)) % 2) * 255).astype(np.uint8)[:, :, None], 3, 2)))
    sym = np.random.default_rng(7).integers(0, 256, (64, 32, 3), dtype=np.uint8)
    cases.append(("mirror_lr", "stored", np.concatenate([sym, sym[:, ::-1]], axis=1)))
    # other PIL modes reach the path too: L and RGBA (alpha ignored by convert("L"))
    cases.append(("mode_L", "stored", np.random.default_rng(8).integers(0, 256, (80, 100), dtype=np.uint8)))
    cases.append(("mode_RGBA", "stored", np.random.default_rng(9).integers(0, 256, (70, 90, 4), dtype=np.uint8)))

    out = {"names": [], "kind": [], "params": [], "sha256": [], "tile32": [], "tile98": [], "phash": [], "dhash": [],
           "margin": []}
    for name, kind, payload in cases:
        if kind == "synth":
            i, w, h = payload
            arr = O.synth_rgb(i, w, h)
            params = [i, w, h]
        else:
            arr = payload
            params = [0, arr.shape[1], arr.shape[0]]
            pixels["px_" + name] = arr
        img = Image.fromarray(arr)  # mode L / RGB / RGBA by shape
        t32, t98, ph, dh, margin = ref_hashes(img)
        out["names"].append(name); out["kind"].append(kind); out["params"].append(params); out["sha256"].append(sha(arr))
        out["tile32"].append(t32); out["tile98"].append(t98); out["phash"].append(ph); out["dhash"].append(dh)
        out["margin"].append(margin)
    np.savez_compressed(
        os.path.join(HERE, "sig_golden.npz"),
        names=np.array(out["names"]), kind=np.array(out["kind"]), params=np.array(out["params"], np.int64),
        sha256=np.array(out["sha256"]), tile32=np.array(out["tile32"], np.uint8), tile98=np.array(out["tile98"], np.uint8),
        phash=np.array(out["phash"], np.uint64), dhash=np.array(out["dhash"], np.uint64),
        margin=np.array(out["margin"], np.float64), dct_backend=np.array("scipy.fft.dctn(type=2,norm=ortho) float32"),
        pillow_version=np.array(Image.__version__), **pixels)
    print("sig cases:", len(cases), "min margin", min(out["margin"]))


# ------------------------------------------------------------------ scanner
class _Capture:
    """Grab build_clusters' local `edges` dict and funnel counters when it returns."""

    def __init__(self):
        self.edges = None
        self.counters = None

    def __call__(self, frame, event, arg):
        if event == "return" and frame.f_code.co_name == "build_clusters":
            loc = frame.f_locals
            if "edges" in loc:
                self.edges = [(e.file_id_a, e.file_id_b, e.hamming) for e in loc["edges"].values()]
                self.counters = [loc.get("pair_total", 0), loc.get("pair_after_size", 0), loc.get("pair_after_ham", 0),
                                 loc.get("pair_after_cos", 0)]


def run_scanner(files, cfg_kwargs, cap=None):
    if cap is None:
        os.environ.pop("KE_DUP_BUCKET_PAIR_CAP", None)
    else:
        os.environ["KE_DUP_BUCKET_PAIR_CAP"] = str(cap)
    dfs = [DuplicateFile(file_id=f["file_id"], path=Path(f["path"]), size=f["size"], width=f["width"],
                         height=f["height"], phash=f["phash"], embedding=None) for f in files]
    cap_obj = _Capture()
    sys.setprofile(cap_obj)
    try:
        clusters = DuplicateScanner(DuplicateScanConfig(**cfg_kwargs)).build_clusters(dfs)
    finally:
        sys.setprofile(None)
        os.environ.pop("KE_DUP_BUCKET_PAIR_CAP", None)
    return {
        "config": cfg_kwargs, "bucket_pair_cap": cap,
        "files": {k: [(str(f[k]) if k == "phash" else f[k]) for f in files]
                  for k in ("file_id", "path", "size", "width", "height", "phash")},
        "edges": sorted([[min(a, b), max(a, b), h] for a, b, h in (cap_obj.edges or [])]),
        "counters": cap_obj.counters,
        "clusters": [{"keeper_id": c.keeper_id, "entries": [[e.file.file_id, e.best_hamming] for e in c.files]} for c in clusters],
    }


def synth_files(hashes, ext_cycle=("png", "jpg", "webp", "jpeg", "bmp", "tif")):
    files = []
    for i, hv in enumerate(hashes):
        files.append({"file_id": i + 1, "path": f"dir{i % 3}/img_{i:07d}.{ext_cycle[i % len(ext_cycle)]}",
                      "size": 1000 + (i % 7), "width": 512 - (i % 5), "height": 512, "phash": int(hv)})
    return files


def make_scan():
    scen = {}
    # tests/dup/test_scanner.py:31-69
    base = 0xFFFF_FFFF_0000_0000
    scen["ref_test_keeper"] = run_scanner(
        [{"file_id": 1, "path": "a.jpg", "size": 1000, "width": 640, "height": 480, "phash": base},
         {"file_id": 2, "path": "b.png", "size": 2000, "width": 640, "height": 480, "phash": base ^ 1},
         {"file_id": 3, "path": "c.jpg", "size": 1500, "width": 800, "height": 600, "phash": base ^ 2}],
        {"hamming_threshold": 4})
    # tests/dup/test_scanner.py:72-117 without embeddings (from_row never populates them, scanner.py:110-117)
    b2 = 0xAAAA_AAAA_AAAA_AAAA
    scen["ref_test_ratio"] = run_scanner(
        [{"file_id": 1, "path": "small.jpg", "size": 100, "width": 100, "height": 100, "phash": b2},
         {"file_id": 2, "path": "large.jpg", "size": 1000, "width": 100, "height": 100, "phash": b2},
         {"file_id": 3, "path": "cosine_a.jpg", "size": 800, "width": 200, "height": 200, "phash": b2 ^ 1},
         {"file_id": 4, "path": "cosine_b.jpg", "size": 820, "width": 200, "height": 200, "phash": b2 ^ 2},
         {"file_id": 5, "path": "cosine_bad.jpg", "size": 830, "width": 200, "height": 200, "phash": b2 ^ 3}],
        {"hamming_threshold": 4, "size_ratio": 0.5})
    # tests/dup/test_scanner.py:153-163 (T=0 identical hashes)
    scen["ref_test_t0"] = run_scanner(
        [{"file_id": 1, "path": "a.jpg", "size": 100, "width": 10, "height": 10, "phash": 0x1234_5678_0000_0000},
         {"file_id": 2, "path": "b.jpg", "size": 100, "width": 10, "height": 10, "phash": 0x1234_5678_0000_0000}],
        {"hamming_threshold": 0})
    # banding != all-pairs: SURVEY finding 3 (N=1000 planted corpus, T=8)
    h1000 = O.synth_hashes(1000)
    scen["synth1000_t8"] = run_scanner(synth_files(h1000), {"hamming_threshold": 8})
    scen["synth1000_t10"] = run_scanner(synth_files(h1000), {"hamming_threshold": 10})
    scen["synth1000_t3"] = run_scanner(synth_files(h1000), {"hamming_threshold": 3})
    scen["synth1000_t8_ratio"] = run_scanner(synth_files(h1000), {"hamming_threshold": 8, "size_ratio": 0.9985})
    scen["synth1000_t8_bands8x8"] = run_scanner(synth_files(h1000), {"hamming_threshold": 8, "band_bits": 8, "band_count": 8})
    scen["synth1000_t8_bands32x2"] = run_scanner(synth_files(h1000), {"hamming_threshold": 8, "band_bits": 32, "band_count": 2})
    scen["synth1000_t8_bands12x3"] = run_scanner(synth_files(h1000), {"hamming_threshold": 8, "band_bits": 12, "band_count": 3})
    scen["no_bucket_ge2"] = run_scanner(synth_files(h1000[:200]), {"hamming_threshold": 64})
    # bucket cap: many files share the low band -> that bucket is skipped, others still link
    rng = np.random.default_rng(5)
    hot = [(int(rng.integers(0, 1 << 48)) << 16) | 0xBEEF for _ in range(40)]
    for k in range(0, 40, 2):
        hot[k + 1] = hot[k] ^ (1 << 20)  # near-dup partner sharing 3 bands
    scen["bucket_cap"] = run_scanner(synth_files(hot + [int(v) for v in h1000[:100]]), {"hamming_threshold": 8}, cap=100)
    scen["bucket_cap_off"] = run_scanner(synth_files(hot + [int(v) for v in h1000[:100]]), {"hamming_threshold": 8})
    # duplicate file ids + signed hash inputs (readers re-mask, scanner.py:81,229).  Sources of the
    # planted tail live in [0,900); give some planted copies the id of an unrelated file and of
    # their own source, so "same id -> skip" (:266) and "first writer wins" (:287-290) both fire.
    dup_files = synth_files([int(v) for v in h1000])
    t8_edges = scen["synth1000_t8"]["edges"]
    (a0, b0, _), (a1, b1, _), (a2, b2_, _) = t8_edges[0], t8_edges[1], t8_edges[2]
    dup_files[b0 - 1]["file_id"] = a0            # edge endpoints share an id -> pair skipped
    dup_files[5]["file_id"] = a1                 # unrelated hash carries the id of an edge endpoint
    dup_files[b2_ - 1]["file_id"] = dup_files[7]["file_id"]  # edge endpoint renamed to another file's id
    for f in dup_files[::3]:
        if f["phash"] >= 1 << 63:
            f["phash"] -= 1 << 64
    scen["dup_ids_signed"] = run_scanner(dup_files, {"hamming_threshold": 10})
    scen["t64_tail"] = run_scanner(synth_files([int(v) for v in h1000[850:1000]]), {"hamming_threshold": 64})
    # sizes None / zero pass the ratio filter (scanner.py:362-366)
    nz = synth_files([int(v) for v in h1000[900:1000]])
    for k, f in enumerate(nz):
        if k % 4 == 0:
            f["size"] = None
        if k % 4 == 1:
            f["size"] = 0
        if k % 4 == 2:
            f["size"] = 10 + k
    scen["sizes_none_zero"] = run_scanner(nz, {"hamming_threshold": 10, "size_ratio": 0.8})
    scen["empty"] = run_scanner([], {"hamming_threshold": 8})
    scen["single"] = run_scanner(synth_files([123]), {"hamming_threshold": 8})
    with open(os.path.join(HERE, "scan_golden.json"), "w") as fh:
        json.dump(scen, fh, separators=(",", ":"))
    for k, v in scen.items():
        print(f"scan {k}: files={len(v['files']['file_id'])} edges={len(v['edges'])} clusters={len(v['clusters'])} counters={v['counters']}")


# ------------------------------------------------------------------ from_row (a7)
def make_rows():
    rows = [
        {"file_id": 10, "path": "blob.png", "size": 12, "width": 3, "height": 4, "phash_bytes": (123).to_bytes(8, "big")},
        {"id": 11, "file_path": "hex.png", "size": 12, "width": 3, "height": 4, "phash_hex": "ff"},
        {"file_id": 1, "path": "x.jpg", "phash_u64": -1},
        {"file_id": 2, "path": "x.JPG", "phash": "0x10", "size": 5.7, "width": "7", "height": None},
        {"file_id": 3, "path": "y.webp", "phash64": "12345678901234567890"},
        {"file_id": 4, "path": "z.tif", "signature": " 0b101 "},
        {"file_id": 5, "path": "q", "sig": (1 << 70) + 5},
        {"path": "noid.gif", "phash_u64": 7},
        {"file_id": 6, "path": "bytes9.png", "phash_bytes": bytes(range(1, 10))},
        {"file_id": 7, "path": "u.apng", "phash_u64": None, "phash": 99},
    ]
    bad = [
        {"file_id": 1, "path": "broken.jpg", "phash_hex": "not-a-hex-value"},
        {"file_id": 2, "path": "none.jpg"},
        {"file_id": 3, "path": "empty.jpg", "phash": "   "},
    ]
    exp = []
    for r in rows:
        f = DuplicateFile.from_row(r)
        exp.append({"file_id": f.file_id, "path": f.path.as_posix(), "size": f.size, "width": f.width,
                    "height": f.height, "phash": str(f.phash), "resolution": f.resolution,
                    "extension_priority": f.extension_priority})
    for r in bad:
        try:
            DuplicateFile.from_row(r)
            raise SystemExit("expected ValueError")
        except ValueError as e:
            assert "missing perceptual hash" in str(e)

    def enc(r):
        return {k: ({"__bytes__": v.hex()} if isinstance(v, bytes) else (str(v) if isinstance(v, int) and abs(v) > 1 << 53 else v))
                for k, v in r.items()}

    with open(os.path.join(HERE, "rows_golden.json"), "w") as fh:
        json.dump({"rows": [enc(r) for r in rows], "expected": exp, "bad_rows": [enc(r) for r in bad]}, fh, indent=1)
    print("rows:", len(rows), "bad:", len(bad))


# ------------------------------------------------------------------ ssim (restatement, NOT the reference)
def skimage_ssim_restated(a_u8: np.ndarray, b_u8: np.ndarray) -> float:
    """skimage.metrics.structural_similarity(a/255 f32, b/255 f32, data_range=1.0), as called at
    src/dup/refine.py:50-52: win 7, uniform_filter, K1=.01, K2=.03, sample covariance, crop 3."""
    from scipy.ndimage import uniform_filter
    x = np.asarray(a_u8, np.float32) / np.float32(255.0)
    y = np.asarray(b_u8, np.float32) / np.float32(255.0)
    ux, uy = uniform_filter(x, size=7), uniform_filter(y, size=7)
    uxx, uyy, uxy = uniform_filter(x * x, size=7), uniform_filter(y * y, size=7), uniform_filter(x * y, size=7)
    cn = np.float32(49.0 / 48.0)
    vx, vy, vxy = cn * (uxx - ux * ux), cn * (uyy - uy * uy), cn * (uxy - ux * uy)
    C1, C2 = np.float32(0.01 ** 2), np.float32(0.03 ** 2)
    A1, A2, B1, B2 = 2 * ux * uy + C1, 2 * vxy + C2, ux ** 2 + uy ** 2 + C1, vx + vy + C2
    S = (A1 * A2) / (B1 * B2)
    return float(S[3:-3, 3:-3].mean(dtype=np.float64))


def make_ssim():
    pairs = []
    from PIL import ImageEnhance
    # tests/dup/test_refine.py:24-34 : solid (200,10,10) 64x64 vs brightness x1.02
    a = Image.new("RGB", (64, 64), color=(200, 10, 10))
    b = ImageEnhance.Brightness(a).enhance(1.02)
    pairs.append(("ref_solid_bright", np.asarray(a.convert("L")), np.asarray(b.convert("L"))))
    # tests/dup/test_refine.py:37-46 : green vs blue
    pairs.append(("ref_green_blue", np.asarray(Image.new("RGB", (64, 64), (0, 255, 0)).convert("L")),
                  np.asarray(Image.new("RGB", (64, 64), (0, 0, 255)).convert("L"))))
    for (i, j, w, h) in [(O.synth_info(19)[0], 19, 256, 256), (O.synth_info(29)[0], 29, 512, 512), (0, 1, 256, 256), (19, 29, 128, 96), (2, 2, 64, 64), (3, 4, 7, 7), (5, 6, 40, 9)]:
        la = np.asarray(Image.fromarray(O.synth_rgb(i, w, h)).convert("L"))
        lb = np.asarray(Image.fromarray(O.synth_rgb(j, w, h)).convert("L"))
        pairs.append((f"synth_{i}_{j}_{w}x{h}", la, lb))
    rng = np.random.default_rng(11)
    n1 = rng.integers(0, 256, (50, 70), dtype=np.uint8)
    n2 = np.clip(n1.astype(int) + rng.integers(-6, 7, n1.shape), 0, 255).astype(np.uint8)
    pairs.append(("noise_pm6", n1, n2))
    store = {"names": np.array([p[0] for p in pairs]), "ssim": np.array([skimage_ssim_restated(p[1], p[2]) for p in pairs]),
             "source": np.array("scipy.ndimage.uniform_filter restatement of skimage 0.25 structural_similarity (skimage absent)")}
    for name, la, lb in pairs:
        store["a_" + name] = la
        store["b_" + name] = lb
    np.savez_compressed(os.path.join(HERE, "ssim_golden.npz"), **store)
    print("ssim:", dict(zip(store["names"].tolist(), store["ssim"].round(6).tolist())))


# ------------------------------------------------------------------ ImageOps.fit + BICUBIC of the SSIM step
def make_fit():
    """What src/dup/refine.py:45-49 does to two images of different size before SSIM: both go through
    ImageOps.fit(image.convert("L"), (min w, min h), BICUBIC).  Pillow is installed here, so these are the
    reference's own pixels; the SSIM values beside them come from the SciPy restatement above."""
    from PIL import ImageOps
    # (synthetic image index, its w x h) pairs; sizes cover: wider / taller source, both axes resampled, one axis
    # cropped only (integer box = plain crop), identity, odd sizes, Pillow's tall-image rule, tiny images
    pairs = [((19, 160, 120), (O.synth_info(19)[0], 96, 96)), ((29, 200, 150), (O.synth_info(29)[0], 120, 150)),
             ((3, 131, 77), (4, 64, 99)), ((5, 128, 128), (6, 128, 128)), ((7, 100, 80), (8, 80, 80)),
             ((9, 90, 60), (10, 45, 30)), ((11, 33, 31), (12, 29, 37)), ((13, 3, 420), (14, 3, 200)),
             ((15, 257, 64), (16, 64, 257))]
    store = {"names": [], "ssim": [], "size": []}
    for (ia, wa, ha), (ib, wb, hb) in pairs:
        a, b = Image.fromarray(O.synth_rgb(ia, wa, ha)), Image.fromarray(O.synth_rgb(ib, wb, hb))
        size = (min(a.width, b.width), min(a.height, b.height))
        fa = np.asarray(ImageOps.fit(a.convert("L"), size, Image.Resampling.BICUBIC))
        fb = np.asarray(ImageOps.fit(b.convert("L"), size, Image.Resampling.BICUBIC))
        name = f"{ia}_{wa}x{ha}__{ib}_{wb}x{hb}"
        store["names"].append(name)
        store["size"].append(size)
        store["ssim"].append(skimage_ssim_restated(fa, fb) if min(size) >= 7 else np.nan)
        store["fa_" + name], store["fb_" + name] = fa, fb
        store["pa_" + name], store["pb_" + name] = np.array([ia, wa, ha]), np.array([ib, wb, hb])
    # upscaling and explicit target sizes (ImageOps.fit is also the thumbnail helper of Pillow)
    rng = np.random.default_rng(23)
    extra = []
    for (w, h, ow, oh) in [(40, 30, 64, 64), (64, 48, 100, 20), (50, 50, 50, 49), (31, 17, 31, 17), (120, 9, 60, 9), (2, 2, 5, 3)]:
        px = rng.integers(0, 256, (h, w), dtype=np.uint8)
        extra.append((px, (ow, oh), np.asarray(ImageOps.fit(Image.fromarray(px, "L"), (ow, oh), Image.Resampling.BICUBIC))))
    store["n_extra"] = np.array(len(extra))
    for k, (px, size, out) in enumerate(extra):
        store[f"xin_{k}"], store[f"xsize_{k}"], store[f"xout_{k}"] = px, np.array(size), out
    store["names"], store["ssim"], store["size"] = np.array(store["names"]), np.array(store["ssim"]), np.array(store["size"])
    store["source"] = np.array(f"Pillow {Image.__version__} ImageOps.fit(convert('L'), size, BICUBIC); ssim = SciPy restatement")
    np.savez_compressed(os.path.join(HERE, "fit_golden.npz"), **store)
    print("fit:", dict(zip(store["names"].tolist(), np.round(store["ssim"], 6).tolist())))


# ------------------------------------------------------------------ shipped refine stage
def refine_corpus():
    """(name, pixel array) list; PNG round trips are lossless, so tests rebuild the very same files."""
    items = []
    for i in (7, 19, 17, 29, 15, 39, 0, 1):                     # bases and their planted variants
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


def make_refine_parallel():
    import tempfile
    from dataclasses import dataclass

    import ui.dup_refine_parallel as R

    @dataclass
    class F:
        file_id: int
        path: Path

    @dataclass
    class E:
        file: F

    @dataclass
    class Cl:
        files: list
        keeper_id: int

    out = {"cases": {}, "clusters": []}
    with tempfile.TemporaryDirectory() as td:
        paths = {}
        for name, arr in refine_corpus():
            p = Path(td) / f"{name}.png"
            Image.fromarray(arr).save(p)
            paths[name] = p
        thumbs = {}
        for name, p in paths.items():
            thumbs[name] = R._load_small_gray(p, 128)
            out["cases"][name] = {
                "ahash": {f"{g}x{t}": format(R.tile_ahash_bits(p, grid=g, tile=t), "x") for g, t in ((4, 8), (2, 16), (8, 4), (3, 5))},
                "thumb128_sha256": sha(thumbs[name]),
                "thumb64_sha256": sha(R._load_small_gray(p, 64)),
            }
        names = list(paths)
        out["mae"] = [[a, b, R._mae01(thumbs[a], thumbs[b])] for a, b in
                      [("v007_256", "v019_256"), ("v017_256", "v029_256"), ("v007_256", "v007_plus2"), ("v000_256", "v001_256"),
                       ("gray_L", "rgba"), ("s003_512x512", "v007_256")]]
        ids = {n: k + 1 for k, n in enumerate(names)}
        groups = [(["v007_256", "v019_256", "v007_plus2", "v000_256"], "v007_256"), (["v017_256", "v029_256"], "v029_256"),
                  (["v015_256", "v039_256", "s003_512x512"], "v015_256"), (["gray_L", "rgba"], "gray_L"),
                  (["v000_256", "v001_256"], "v001_256")]
        clusters = [Cl([E(F(ids[n], paths[n])) for n in members], ids[keeper]) for members, keeper in groups]
        out["cluster_inputs"] = [{"members": m, "keeper": k} for m, k in groups]
        for max_bits in (32, 200, 8):
            res = R.refine_by_tilehash_parallel(clusters, grid=4, tile=8, max_bits=max_bits, io_workers=2)
            out["clusters"].append({"stage": "tilehash", "max_bits": max_bits,
                                    "result": [[cl.keeper_id, [e.file.file_id for e in cl.files]] for cl in res]})
        for thr in (0.004, 0.006, 0.05):
            res = R.refine_by_pixels_parallel(clusters, mae_thr=thr, thumb_size=128, workers=1)
            out["clusters"].append({"stage": "pixels", "mae_thr": thr,
                                    "result": sorted([cl.keeper_id, [e.file.file_id for e in cl.files]] for cl in res)})
        out["ids"] = ids
    with open(os.path.join(HERE, "refine_parallel_golden.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("refine_parallel:", len(out["cases"]), "files;", [(c["stage"], len(c["result"])) for c in out["clusters"]], "mae", [round(m[2], 5) for m in out["mae"]])


# ------------------------------------------------------------------ cluster maintenance
def make_cluster_update():
    import ui.dup_cluster_update as U
    import ui.dup_tree_state as T

    files = synth_files([int(v) for v in O.synth_hashes(1000)])
    dfs = [DuplicateFile(file_id=f["file_id"], path=Path(f["path"]), size=f["size"], width=f["width"], height=f["height"],
                         phash=f["phash"], embedding=None) for f in files]
    clusters = DuplicateScanner(DuplicateScanConfig(hamming_threshold=12, band_bits=8, band_count=8)).build_clusters(dfs)
    enc = lambda cs: [[c.keeper_id, [[e.file.file_id, e.best_hamming] for e in c.files]] for c in cs]
    out = {"scan": {"hamming_threshold": 12, "band_bits": 8, "band_count": 8}, "clusters": enc(clusters), "removals": []}
    for removed in ([c.keeper_id for c in clusters[::2]], [f["file_id"] for f in files[::3]], [], [f["file_id"] for f in files]):
        out["removals"].append({"removed": removed, "result": enc(U.rebuild_clusters_after_removal(clusters, set(removed)))})
    out["hamming_score"] = [T.cluster_hamming_score(c) for c in clusters]
    out["default_checked"] = [[e.file.file_id for e in T.default_checked_entries(c)] for c in clusters]
    with open(os.path.join(HERE, "cluster_update_golden.json"), "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print("cluster_update:", len(clusters), "clusters;", [len(r["result"]) for r in out["removals"]])


# ------------------------------------------------------------------ decode normalisation
def image_io_cases(td):
    """Write the test files into directory td; returns [(name, path, kwargs)].  Shared with tests/_golden.py."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _golden import write_image_io_files
    return write_image_io_files(td)


def make_image_io():
    import tempfile

    from utils.image_io import safe_load_image

    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name, path, kwargs in image_io_cases(td):
            import warnings
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    img = safe_load_image(path, **kwargs)
            except Exception as exc:          # e.g. Pillow >= 10 raises DecompressionBombError already in Image.open
                out[name] = {"raises": type(exc).__name__}
                continue
            out[name] = None if img is None else {"mode": img.mode, "size": list(img.size), "sha256": hashlib.sha256(img.tobytes()).hexdigest()}
    with open(os.path.join(HERE, "image_io_golden.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("image_io:", {k: (v and (v.get("size") or v.get("raises"))) for k, v in out.items()})


# ------------------------------------------------------------------ the seams, end to end
def config0_rows(n=1000, side=256):
    """Row dicts in the shape of db.repository.iter_files_for_dup for BASELINE configs[0] (file_id = i + 1,
    size = 1000 + (i mod 7), path = img_{i:07d}.png); shared with tests/_golden.py through the fixture itself."""
    return [{"file_id": i + 1, "path": f"img_{i:07d}.png", "size": 1000 + (i % 7), "width": side, "height": side} for i in range(n)]


def make_config0():
    """BASELINE configs[0] whole: 1 000 synthetic 256 x 256 RGB images -> the reference's phash / dhash (src/sig/phash.py:33-57,
    SciPy DCT stand-in) -> DuplicateFile.from_row -> DuplicateScanner(hamming_threshold=8).build_clusters
    (src/dup/scanner.py:211-356) -> hashes, edges, funnel counters, clusters, keepers."""
    from core.fastsig import _to_signed64

    n, side = 1000, 256
    rows = config0_rows(n, side)
    ph, dh = [], []
    for i in range(n):
        img = Image.fromarray(O.synth_rgb(i, side, side))
        ph.append(_to_signed64(ref_sig.phash(img)))
        dh.append(_to_signed64(ref_sig.dhash(img)))
    for r, p in zip(rows, ph):
        r["phash_u64"] = p                       # what the signatures table holds: the signed wrap
    files = [DuplicateFile.from_row(r) for r in rows]
    cap_obj = _Capture()
    sys.setprofile(cap_obj)
    try:
        clusters = DuplicateScanner(DuplicateScanConfig(hamming_threshold=8)).build_clusters(files)
    finally:
        sys.setprofile(None)
    out = {
        "n": n, "side": side, "config": {"hamming_threshold": 8}, "dct_backend": "scipy.fft.dctn(type=2,norm=ortho) float32",
        "phash_s64": [str(v) for v in ph], "dhash_s64": [str(v) for v in dh],
        "edges": sorted([min(a, b), max(a, b), h] for a, b, h in cap_obj.edges),
        "counters": cap_obj.counters,
        "clusters": [{"keeper_id": c.keeper_id, "entries": [[e.file.file_id, e.best_hamming] for e in c.files]} for c in clusters],
    }
    with open(os.path.join(HERE, "config0_golden.json"), "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print(f"config0: {n} images, {len(out['edges'])} edges, {len(clusters)} clusters, counters {out['counters']}")


def make_scan100k():
    """The reference's scanner on a BASELINE-size table (100 000 synthetic hashes, T = 8; the probe of BASELINE.md section 2):
    too many edges to store, so digests of the canonical listings + the counts."""
    n = 100_000
    hashes = O.synth_hashes(n)
    files = [DuplicateFile(file_id=i + 1, path=Path(f"img_{i:07d}.png"), size=1000 + (i % 7), width=512, height=512,
                           phash=int(hashes[i]), embedding=None) for i in range(n)]
    out = {"n": n, "hash_generator": "oracle.synth_hashes", "runs": []}
    for kwargs in ({"hamming_threshold": 8}, {"hamming_threshold": 8, "size_ratio": 0.9985}):
        cap_obj = _Capture()
        sys.setprofile(cap_obj)
        try:
            clusters = DuplicateScanner(DuplicateScanConfig(**kwargs)).build_clusters(files)
        finally:
            sys.setprofile(None)
        edges = sorted((min(a, b), max(a, b), h) for a, b, h in cap_obj.edges)
        listing = [[c.keeper_id, [[e.file.file_id, e.best_hamming] for e in c.files]] for c in clusters]
        out["runs"].append({
            "config": kwargs, "n_edges": len(edges), "n_clusters": len(clusters), "counters": cap_obj.counters,
            "edges_sha256": hashlib.sha256(json.dumps(edges, separators=(",", ":")).encode()).hexdigest(),
            "clusters_sha256": hashlib.sha256(json.dumps(listing, separators=(",", ":")).encode()).hexdigest(),
            "first_clusters": listing[:5]})
        print(f"scan100k {kwargs}: {len(edges)} edges, {len(clusters)} clusters, counters {cap_obj.counters}")
    with open(os.path.join(HERE, "scan100k_golden.json"), "w") as fh:
        json.dump(out, fh, separators=(",", ":"))


def _png_chunks(w, h, depth, ctype, interlace, payload_rows: bytes, *, idat_split=None, between=None) -> bytes:
    """A PNG container around already filtered scanlines (filter byte + packed samples per row / per Adam7 pass row)."""
    import struct
    import zlib

    def ch(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))

    z = zlib.compress(payload_rows, 6)
    out = b"\x89PNG\r\n\x1a\n" + ch(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace))
    if idat_split:
        cut = max(1, len(z) // idat_split)
        parts = [z[o:o + cut] for o in range(0, len(z), cut)]
    else:
        parts = [z]
    for k, part in enumerate(parts):
        out += ch(b"IDAT", part)
        if between is not None and k == 0 and len(parts) > 1:
            out += ch(*between)
    return out + ch(b"IEND", b"")


def _adam7_rows(arr: np.ndarray) -> bytes:
    """8- or 16-bit samples (H x W x C, big-endian bytes already) -> the seven passes' rows, filter 0."""
    h, w = arr.shape[:2]
    out = b""
    for (x0, y0, dx, dy) in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        sub = arr[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        for row in sub:
            out += b"\x00" + np.ascontiguousarray(row).tobytes()
    return out


def worker_corpus():
    """[(file name, bytes)] -- the files the batch-hasher seam meets: every format family the reference ranks as a keeper
    (src/dup/scanner.py:16-28) in the modes Pillow opens them in, damaged and mis-named files included.  Written ONCE here and
    stored as bytes in the fixture, so the test does not depend on any encoder."""
    import io

    rng = np.random.default_rng(41)
    files = []

    def photo(w, h, i=3):
        return O.synth_rgb(i, w, h)

    def save(name, img, fmt, **kw):
        b = io.BytesIO()
        img.save(b, fmt, **kw)
        files.append((name, b.getvalue()))
        return b.getvalue()

    rgb = Image.fromarray(photo(96, 80))
    rgb2 = Image.fromarray(photo(200, 120, 19))
    odd = Image.fromarray(photo(97, 61, 7))
    gray = rgb.convert("L")
    # ---- JPEG family (suffix routes them to the GPU decoder in the build)
    save("base_420.jpg", rgb, "JPEG", quality=85)
    save("base_444_q95.jpeg", rgb2, "JPEG", quality=95, subsampling=0)
    save("base_422.JPG", odd, "JPEG", quality=75, subsampling=1)
    save("base_gray.jpg", gray, "JPEG", quality=80)
    save("base_optimized.jpe", rgb2, "JPEG", quality=60, optimize=True)
    save("progressive.jpg", rgb2, "JPEG", quality=85, progressive=True)
    save("progressive_gray.jpg", gray, "JPEG", quality=85, progressive=True)
    save("progressive_444.jfif", odd, "JPEG", quality=92, progressive=True, subsampling=0)
    save("cmyk.jpg", rgb.convert("CMYK"), "JPEG", quality=90)
    exif = Image.Exif()
    exif[0x0112] = 6
    save("exif_rot6.jpg", rgb2, "JPEG", quality=90, exif=exif.tobytes())       # the worker does NOT transpose (fastsig.py:31-34)
    exif[0x0112] = 3
    save("exif_rot3_progressive.jpg", odd, "JPEG", quality=88, exif=exif.tobytes(), progressive=True)
    try:
        save("restart_rows.jpg", rgb2, "JPEG", quality=85, restart_marker_rows=1)
        save("restart_blocks.jpg", rgb2, "JPEG", quality=85, subsampling=0, restart_marker_blocks=3)
    except TypeError:
        pass
    save("tiny_8x8.jpg", Image.fromarray(photo(8, 8)), "JPEG", quality=90)
    save("tiny_1x1.jpg", Image.fromarray(photo(1, 1)), "JPEG", quality=90)
    save("thin_3x300.jpg", Image.fromarray(photo(3, 300)), "JPEG", quality=90)
    save("vga.jpg", Image.fromarray(photo(640, 480, 29)), "JPEG", quality=85)
    save("q100_noise.jpg", Image.fromarray(rng.integers(0, 256, (70, 90, 3), dtype=np.uint8)), "JPEG", quality=100, subsampling=0)
    save("rgb_coded.jpg", rgb, "JPEG", quality=90, keep_rgb=True)            # RGB-coded: no colour transform (Adobe marker)
    # ---- PNG family
    save("rgb.png", rgb, "PNG")
    save("rgb_level0.png", rgb2, "PNG", compress_level=0)
    save("rgb_level9_opt.PNG", odd, "PNG", optimize=True)
    save("rgba.png", Image.fromarray(rng.integers(0, 256, (70, 90, 4), dtype=np.uint8)), "PNG")
    save("rgba_photo_halfalpha.png", Image.fromarray(np.dstack([photo(96, 80), np.tile(np.arange(96, dtype=np.uint8) * 2, (80, 1))])), "PNG")
    save("la.png", Image.fromarray(np.dstack([np.asarray(gray), np.tile(np.arange(96, dtype=np.uint8) * 2, (80, 1))]), "LA"), "PNG")
    save("l.png", gray, "PNG")
    pal = rgb.convert("P", palette=Image.Palette.ADAPTIVE, colors=256)
    save("p256.png", pal, "PNG")
    save("p256_trns.png", pal, "PNG", transparency=3)
    save("p16.png", rgb.convert("P", palette=Image.Palette.ADAPTIVE, colors=16), "PNG")
    save("p4.png", rgb.convert("P", palette=Image.Palette.ADAPTIVE, colors=4), "PNG")
    save("p2.png", rgb.convert("P", palette=Image.Palette.ADAPTIVE, colors=2), "PNG")
    save("p16_trns_bytes.png", rgb.convert("P", palette=Image.Palette.ADAPTIVE, colors=16), "PNG", transparency=bytes([0, 128, 255, 7]))
    save("bilevel.png", gray.convert("1"), "PNG")
    save("i16.png", Image.fromarray((np.asarray(gray).astype(np.uint16) * 257) ^ 0x55), "PNG")
    save("apng_2frames.png", rgb, "PNG", save_all=True, append_images=[rgb2.resize(rgb.size)])
    # hand-made: what Pillow's writer never produces
    g8 = np.asarray(gray)
    for depth in (2, 4):
        per = 8 // depth
        vals = (g8 >> (8 - depth)).astype(np.uint8)
        pad = np.zeros((vals.shape[0], (vals.shape[1] + per - 1) // per * per), np.uint8)
        pad[:, :vals.shape[1]] = vals
        packed = np.zeros((vals.shape[0], pad.shape[1] // per), np.uint8)
        for q in range(per):
            packed |= pad[:, q::per] << (8 - depth * (q + 1))
        rows = b"".join(b"\x00" + r.tobytes() for r in packed)
        files.append((f"gray{depth}bit.png", _png_chunks(vals.shape[1], vals.shape[0], depth, 0, 0, rows)))
    a3 = photo(75, 53, 9)
    files.append(("adam7_rgb.png", _png_chunks(75, 53, 8, 2, 1, _adam7_rows(a3))))
    files.append(("adam7_gray.png", _png_chunks(96, 80, 8, 0, 1, _adam7_rows(g8[:, :, None]))))
    a4 = np.dstack([photo(75, 53, 9), np.full((53, 75), 200, np.uint8)])
    files.append(("adam7_rgba.png", _png_chunks(75, 53, 8, 6, 1, _adam7_rows(a4))))
    r16 = (photo(64, 48, 11).astype(np.uint16) * 257 + 13).astype(">u2")
    rows16 = b"".join(b"\x00" + np.ascontiguousarray(r).tobytes() for r in r16)
    files.append(("rgb16.png", _png_chunks(64, 48, 16, 2, 0, rows16)))
    la16 = np.dstack([np.asarray(gray)[:48, :64].astype(np.uint16) * 257, np.full((48, 64), 40000, np.uint16)]).astype(">u2")
    files.append(("la16.png", _png_chunks(64, 48, 16, 4, 0, b"".join(b"\x00" + np.ascontiguousarray(r).tobytes() for r in la16))))
    rows8 = b"".join(b"\x00" + np.ascontiguousarray(r).tobytes() for r in photo(64, 48, 11))
    files.append(("idat_split5.png", _png_chunks(64, 48, 8, 2, 0, rows8, idat_split=5)))
    files.append(("idat_text_idat.png", _png_chunks(64, 48, 8, 2, 0, rows8, idat_split=2, between=(b"tEXt", b"k\x00v"))))
    # ---- the other keeper formats (src/dup/scanner.py:16-28): Pillow decodes them in the build as in the reference
    save("rgb24.bmp", rgb2, "BMP")
    save("pal8.bmp", pal, "BMP")
    save("gray8.bmp", gray, "BMP")
    save("rgba32.bmp", Image.fromarray(a4), "BMP")
    save("still.gif", pal, "GIF")
    save("anim.gif", pal, "GIF", save_all=True, append_images=[rgb2.resize(rgb.size).convert("P")], duration=50)
    save("gif_transparent.gif", pal, "GIF", transparency=5)
    save("rgb.tif", rgb2, "TIFF")
    save("rgb_lzw.tiff", odd, "TIFF", compression="tiff_lzw")
    save("gray16.tif", Image.fromarray((np.asarray(gray).astype(np.uint16) * 200)), "TIFF")
    save("float32.tif", Image.fromarray((np.asarray(gray).astype(np.float32) * 1.5 - 20)), "TIFF")
    save("int32.tif", Image.fromarray((np.asarray(gray).astype(np.int32) * 3 - 100)), "TIFF")
    save("cmyk.tif", rgb.convert("CMYK"), "TIFF")
    save("bilevel.tif", gray.convert("1"), "TIFF")
    save("lossy.webp", rgb2, "WEBP", quality=80)
    save("lossless.webp", odd, "WEBP", lossless=True)
    save("alpha.webp", Image.fromarray(a4), "WEBP", quality=90)
    save("raw.ppm", rgb, "PPM")
    save("raw.pgm", gray, "PPM")
    save("rgb.tga", rgb, "TGA")
    save("icon.ico", rgb.resize((64, 64)), "ICO")
    save("wavelet.jp2", rgb2, "JPEG2000")
    save("ycbcr.tif", rgb.convert("YCbCr"), "TIFF")
    # ---- mis-named, damaged, not there
    png_bytes = dict(files)["rgb.png"]
    jpg_bytes = dict(files)["base_420.jpg"]
    prog_bytes = dict(files)["progressive.jpg"]
    files.append(("png_named.jpg", png_bytes))
    files.append(("jpeg_named.png", jpg_bytes))
    files.append(("webp_named.png", dict(files)["lossy.webp"]))
    files.append(("truncated_60.jpg", jpg_bytes[: len(jpg_bytes) * 6 // 10]))
    files.append(("truncated_last2.jpg", jpg_bytes[:-2]))
    files.append(("truncated_prog.jpg", prog_bytes[: len(prog_bytes) // 2]))
    files.append(("truncated.png", png_bytes[: len(png_bytes) * 2 // 3]))
    files.append(("truncated_iend.png", png_bytes[:-12]))
    files.append(("trailing_garbage.jpg", jpg_bytes + b"\x00garbage after EOI" * 7))
    files.append(("trailing_garbage.png", png_bytes + b"tail"))
    files.append(("empty.jpg", b""))
    files.append(("empty.png", b""))
    files.append(("garbage.jpg", bytes(rng.integers(0, 256, 600, dtype=np.uint8))))
    files.append(("text.png", b"not an image at all\n"))
    bad_crc = bytearray(png_bytes)
    bad_crc[29] ^= 0xFF                                    # the IHDR chunk's CRC
    files.append(("bad_ihdr_crc.png", bytes(bad_crc)))
    flipped = bytearray(jpg_bytes)
    flipped[len(flipped) // 2] ^= 0x5A
    files.append(("bitflip_mid.jpg", bytes(flipped)))
    return [(n, b) for n, b in files if b is not None]


def make_worker():
    """The corpus above through the reference's worker, core.fastsig._compute_worker (src/core/fastsig.py:24-37): per file
    (file_id, phash_s64, dhash_s64) or None (dropped).  A missing path and a directory are part of the task list."""
    import tempfile
    import warnings

    from core.fastsig import _compute_worker

    corpus = worker_corpus()
    names = [n for n, _ in corpus]
    assert len(set(names)) == len(names)
    store = {"names": np.array(names)}
    for k, (_, data) in enumerate(corpus):
        store[f"f{k}"] = np.frombuffer(data, np.uint8)
    rows = {}
    with tempfile.TemporaryDirectory() as td:
        tasks = []
        for k, (name, data) in enumerate(corpus):
            p = os.path.join(td, name)
            with open(p, "wb") as fh:
                fh.write(data)
            tasks.append((k + 1, p, name))
        os.mkdir(os.path.join(td, "a_directory.jpg"))
        tasks.append((len(corpus) + 1, os.path.join(td, "a_directory.jpg"), "a_directory.jpg"))
        tasks.append((len(corpus) + 2, os.path.join(td, "does_not_exist.png"), "does_not_exist.png"))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for fid, p, name in tasks:
                out = _compute_worker((fid, p))
                info = None
                try:
                    with Image.open(p) as im:
                        info = [im.format, im.mode, list(im.size)]
                except Exception:
                    pass
                rows[name] = {"file_id": fid, "row": None if out is None else [out[0], str(out[1]), str(out[2])], "opened_as": info}
    np.savez_compressed(os.path.join(HERE, "worker_corpus.npz"), **store)
    with open(os.path.join(HERE, "worker_golden.json"), "w") as fh:
        json.dump({"source": "core.fastsig._compute_worker run on /root/reference with the SciPy cv2.dct stand-in",
                   "pillow_version": Image.__version__, "rows": rows}, fh, indent=1)
    kept = sum(1 for v in rows.values() if v["row"] is not None)
    print(f"worker: {len(rows)} tasks, {kept} hashed, dropped:", [n for n, v in rows.items() if v["row"] is None])
    print("  total bytes", sum(len(b) for _, b in corpus))


def make_refine_turned():
    """The shipped refine stage on files the loader has to normalise first (src/ui/dup_refine_parallel.py:59-83, 203-207:
    Image.open + ImageOps.exif_transpose + convert("L") + BILINEAR resize): JPEG files carrying every EXIF orientation, RGBA and
    gray + alpha PNG files -- written once, stored as bytes, through the reference's own tile_ahash_bits / _load_small_gray /
    _mae01 and both refine_by_* functions."""
    import io
    import tempfile
    from dataclasses import dataclass

    import ui.dup_refine_parallel as R

    @dataclass
    class F:
        file_id: int
        path: Path

    @dataclass
    class E:
        file: F

    @dataclass
    class Cl:
        files: list
        keeper_id: int

    rng = np.random.default_rng(77)
    base = O.synth_rgb(2001, 200, 152)
    near = np.clip(base.astype(np.int16) + rng.integers(-3, 4, base.shape), 0, 255).astype(np.uint8)
    other = O.synth_rgb(2002, 200, 152)
    files = []

    def put(name, img, fmt, **kw):
        b = io.BytesIO()
        img.save(b, fmt, **kw)
        files.append((name, b.getvalue()))

    for o in range(1, 9):
        ex = Image.Exif()
        ex[0x0112] = o
        # stored so that the loader's turn brings every one of them back to the same upright picture (o = 1 is upright)
        src = {1: near, 2: near[:, ::-1], 3: near[::-1, ::-1], 4: near[::-1], 5: near.transpose(1, 0, 2), 6: near.transpose(1, 0, 2)[:, ::-1],
               7: near[::-1, ::-1].transpose(1, 0, 2), 8: near.transpose(1, 0, 2)[::-1]}[o]
        put(f"turned{o}.jpg", Image.fromarray(np.ascontiguousarray(src)), "JPEG", quality=93, exif=ex.tobytes(), progressive=bool(o % 2))
    put("upright.jpg", Image.fromarray(base), "JPEG", quality=93)
    put("other.jpg", Image.fromarray(other), "JPEG", quality=90)
    alpha = rng.integers(0, 256, base.shape[:2], dtype=np.uint8)
    put("rgba.png", Image.fromarray(np.dstack([near, alpha]), "RGBA"), "PNG")
    put("la.png", Image.fromarray(np.dstack([np.asarray(Image.fromarray(near).convert("L")), alpha]), "LA"), "PNG")
    # 16-bit and Adam7 PNG files (no writer for them in Pillow: filter type 0 rows in a hand-made container): samples with the
    # picture in the high byte and noise in the low one; 16-bit grayscale dark enough that only some samples pass 255
    low = rng.integers(0, 256, base.shape, dtype=np.uint8)

    def be16(hi, lo):
        return np.stack([hi, lo], -1).reshape(hi.shape[0], hi.shape[1], -1)          # H x W x 2C bytes, big-endian samples

    def rows0(a):
        return b"".join(b"\x00" + np.ascontiguousarray(r).tobytes() for r in a)

    h_, w_ = base.shape[:2]
    files.append(("rgb16.png", _png_chunks(w_, h_, 16, 2, 0, rows0(be16(near, low)))))
    g8 = np.asarray(Image.fromarray(near).convert("L"))
    files.append(("gray16.png", _png_chunks(w_, h_, 16, 0, 0, rows0(be16((g8 > 200).astype(np.uint8)[:, :, None], g8[:, :, None])))))
    files.append(("la16.png", _png_chunks(w_, h_, 16, 4, 0, rows0(be16(np.dstack([g8, alpha]), low[:, :, :2])))))
    files.append(("rgba16_adam7.png", _png_chunks(w_, h_, 16, 6, 1, _adam7_rows(be16(np.dstack([near, alpha]), np.dstack([low, alpha]))))))
    files.append(("rgb_adam7.png", _png_chunks(w_, h_, 8, 2, 1, _adam7_rows(near))))
    # BMP as Pillow's writer makes it: 24-bit, 32-bit (read back as RGB), 8-bit palette, 8-bit gray
    put("rgb.bmp", Image.fromarray(near), "BMP")
    put("rgbx.bmp", Image.fromarray(np.dstack([near, alpha]), "RGBA"), "BMP")
    put("pal.bmp", Image.fromarray(near).quantize(200), "BMP")
    put("gray.bmp", Image.fromarray(g8), "BMP")
    # GIF: the first frame is what the stage sees (palette -> luma), of a still and of an animation
    put("still.gif", Image.fromarray(near).quantize(256), "GIF")
    put("anim.gif", Image.fromarray(near).quantize(64), "GIF", save_all=True, append_images=[Image.fromarray(other).quantize(64)], duration=60, loop=0)
    put("gray.gif", Image.fromarray(g8), "GIF", interlace=False)
    # TIFF without compression (Pillow's own raw decoder): RGB, RGBA (unassociated alpha), gray, palette
    put("rgb.tif", Image.fromarray(near), "TIFF")
    put("rgba.tiff", Image.fromarray(np.dstack([near, alpha]), "RGBA"), "TIFF")
    put("gray.tif", Image.fromarray(g8), "TIFF")
    put("pal.tif", Image.fromarray(near).quantize(200), "TIFF")
    out = {"names": [n for n, _ in files], "cases": {}, "clusters": []}
    store = {"names": np.array(out["names"])}
    for k, (_, data) in enumerate(files):
        store[f"f{k}"] = np.frombuffer(data, np.uint8)
    with tempfile.TemporaryDirectory() as td:
        paths = {}
        for name, data in files:
            paths[name] = Path(td) / name
            paths[name].write_bytes(data)
        thumbs = {}
        for name, p in paths.items():
            thumbs[name] = R._load_small_gray(p, 128)
            out["cases"][name] = {"ahash": {f"{g}x{t}": format(R.tile_ahash_bits(p, grid=g, tile=t), "x") for g, t in ((4, 8), (8, 4))},
                                  "thumb128_sha256": sha(thumbs[name]), "thumb32_sha256": sha(R._load_small_gray(p, 32))}
        out["mae"] = [[a, b, R._mae01(thumbs[a], thumbs[b])] for a, b in
                      [("upright.jpg", "turned1.jpg"), ("upright.jpg", "turned6.jpg"), ("turned3.jpg", "turned8.jpg"), ("upright.jpg", "rgba.png"),
                       ("rgba.png", "la.png"), ("upright.jpg", "other.jpg"), ("rgba.png", "rgb16.png"), ("la.png", "la16.png"),
                       ("rgba.png", "rgba16_adam7.png"), ("rgb_adam7.png", "rgb16.png"), ("la.png", "gray16.png"), ("rgba.png", "rgb.bmp"),
                       ("rgb.bmp", "rgbx.bmp"), ("rgb.bmp", "pal.bmp"), ("la.png", "gray.bmp"), ("pal.bmp", "still.gif"),
                       ("still.gif", "anim.gif"), ("gray.bmp", "gray.gif"), ("rgb.bmp", "rgb.tif"), ("rgba.png", "rgba.tiff"),
                       ("gray.gif", "gray.tif"), ("pal.bmp", "pal.tif")]]
        ids = {n: k + 1 for k, n in enumerate(paths)}
        groups = [(["upright.jpg"] + [f"turned{o}.jpg" for o in range(1, 9)], "upright.jpg"), (["rgba.png", "la.png", "other.jpg"], "rgba.png"),
                  (["turned5.jpg", "other.jpg", "turned2.jpg"], "turned5.jpg"),
                  (["rgb16.png", "rgb_adam7.png", "rgba16_adam7.png", "la16.png", "gray16.png", "other.jpg"], "rgb16.png"),
                  (["rgb.bmp", "rgbx.bmp", "pal.bmp", "gray.bmp", "upright.jpg", "other.jpg"], "rgb.bmp"),
                  (["still.gif", "anim.gif", "gray.gif", "rgb.bmp", "other.jpg"], "still.gif"),
                  (["rgb.tif", "rgba.tiff", "gray.tif", "pal.tif", "upright.jpg", "other.jpg"], "rgb.tif")]
        clusters = [Cl([E(F(ids[n], paths[n])) for n in members], ids[keeper]) for members, keeper in groups]
        out["cluster_inputs"] = [{"members": m, "keeper": k} for m, k in groups]
        for max_bits in (4, 400, 1024):
            res = R.refine_by_tilehash_parallel(clusters, grid=4, tile=8, max_bits=max_bits, io_workers=2)
            out["clusters"].append({"stage": "tilehash", "max_bits": max_bits, "result": [[cl.keeper_id, [e.file.file_id for e in cl.files]] for cl in res]})
        for thr in (0.004, 0.02, 0.2):
            res = R.refine_by_pixels_parallel(clusters, mae_thr=thr, thumb_size=128, workers=1)
            out["clusters"].append({"stage": "pixels", "mae_thr": thr, "result": sorted([cl.keeper_id, [e.file.file_id for e in cl.files]] for cl in res)})
        out["ids"] = ids
    np.savez_compressed(os.path.join(HERE, "refine_turned_corpus.npz"), **store)
    with open(os.path.join(HERE, "refine_turned_golden.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("refine_turned:", len(files), "files;", [(c["stage"], len(c["result"])) for c in out["clusters"]], "mae", [round(m[2], 5) for m in out["mae"]])


if __name__ == "__main__":
    logging.basicConfig(level=logging.WARNING)
    if "--only-refine-turned" in sys.argv:
        make_refine_turned()
        raise SystemExit(0)
    if "--only-seams" in sys.argv:
        make_config0()
        make_worker()
        make_scan100k()
        raise SystemExit(0)
    if "--only-cluster-update" in sys.argv:
        make_cluster_update()
        raise SystemExit(0)
    if "--only-image-io" in sys.argv:
        make_image_io()
        raise SystemExit(0)
    if "--only-fit" in sys.argv:
        make_fit()
        raise SystemExit(0)
    if "--only-refine-parallel" in sys.argv:
        make_refine_parallel()
        raise SystemExit(0)
    make_sig()
    make_scan()
    make_rows()
    make_ssim()
    make_fit()
    make_refine_parallel()
    make_image_io()
    make_cluster_update()
    make_config0()
    make_worker()
    make_scan100k()
    make_refine_turned()
