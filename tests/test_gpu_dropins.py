"""GPU tests of the reference-shaped seams with real files and a real SQLite table:
fast_fill_missing_signatures (src/core/fastsig.py:102-126), ensure_signatures
(src/core/signature.py:31-62), refine_pair / ClusterBuilder (src/dup/refine.py:71-117,
src/dup/cluster.py:22-70), find_duplicates / run_duplicate_scan (call sequence of
src/ui/dup_workers.py:148-239).  They read like the reference's own tests
(tests/core/test_fastsig.py, tests/core/test_image_signature.py, tests/dup/test_refine.py)."""
from __future__ import annotations

import os

import sqlite3
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
Image = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def K():
    import kobato_eyes_amd

    kobato_eyes_amd._native.get_context(0)
    return kobato_eyes_amd


def _make_conn(path=":memory:"):
    conn = sqlite3.connect(path)
    conn.row_factory = sqlite3.Row
    conn.execute("CREATE TABLE IF NOT EXISTS signatures (file_id INTEGER PRIMARY KEY, phash_u64 INTEGER NOT NULL, dhash_u64 INTEGER NOT NULL)")
    return conn


def _corpus(tmp_path: Path, n=12, side=96):
    """PNG files (lossless, so the decoded pixels are known): synthetic corpus images 10..10+n."""
    items, arrays = [], {}
    for k in range(n):
        arr = O.synth_rgb(10 + k, side + 8 * (k % 3), side)
        p = tmp_path / f"img_{k:03d}.png"
        Image.fromarray(arr).save(p)
        items.append((k + 1, str(p)))
        arrays[k + 1] = arr
    return items, arrays


def test_fast_fill_missing_signatures_end_to_end(K, tmp_path):
    items, arrays = _corpus(tmp_path)
    items.insert(3, (77, str(tmp_path / "does_not_exist.png")))
    (tmp_path / "broken.png").write_bytes(b"not an image")
    items.insert(5, (78, str(tmp_path / "broken.png")))
    db = tmp_path / "sig.db"
    _make_conn(db).close()
    seen = []
    out = K.fast_fill_missing_signatures(str(db), items, max_workers=4, chunksize=5, progress=lambda d, t: seen.append((d, t)))
    assert [fid for fid, _, _ in out] == [fid for fid, _ in items if fid < 70]      # input order, failures dropped
    assert seen[-1] == (len(items), len(items))
    for fid, ph, dh in out:
        ep, ed = O.hash_image(arrays[fid])
        assert (ph, dh) == (O.to_signed64(ep), O.to_signed64(ed))
        assert -(1 << 63) <= ph < (1 << 63)
    rows = _make_conn(db).execute("SELECT file_id, phash_u64, dhash_u64 FROM signatures ORDER BY file_id").fetchall()
    assert [tuple(r) for r in rows] == sorted(out)
    # signed store -> unsigned round trip through the scanner's row parser (tests/core/test_image_signature.py:48-55)
    f = K.DuplicateFile.from_row({"file_id": rows[0]["file_id"], "path": "x.png", "phash_u64": rows[0]["phash_u64"]})
    assert f.phash == rows[0]["phash_u64"] & ((1 << 64) - 1)


def test_fast_fill_spills_past_a_small_staging_buffer(K, tmp_path, monkeypatch):
    """KE_STAGE_BYTES smaller than a chunk's pixels: what does not fit the pinned buffer is hashed on its own; same rows."""
    items, arrays = _corpus(tmp_path)
    monkeypatch.setenv("KE_STAGE_BYTES", str(200 * 1024))       # two or three of the corpus images per buffer
    out = K.compute_signatures_mp(items, max_workers=3, chunksize=6)
    assert [fid for fid, _, _ in out] == [fid for fid, _ in items]
    for fid, ph, dh in out:
        ep, ed = O.hash_image(arrays[fid])
        assert (ph, dh) == (O.to_signed64(ep), O.to_signed64(ed))


def test_fast_fill_on_decoder_processes(K, tmp_path, monkeypatch):
    """The Pillow route on a spawn-context process pool writing into shared, page-locked staging buffers
    (ke_stage_create_shared): same rows as the thread route and as the oracle -- also when a worker's region of the buffer
    is too small for its files (decoded by the parent instead) and when files are missing or broken."""
    items, arrays = _corpus(tmp_path)
    items.insert(3, (77, str(tmp_path / "does_not_exist.png")))
    (tmp_path / "broken.bmp").write_bytes(b"not an image")
    items.insert(5, (78, str(tmp_path / "broken.bmp")))
    monkeypatch.setenv("KE_GPU_PNG", "0")                        # every file takes the Pillow route
    monkeypatch.setenv("KE_GPU_JPEG", "0")
    monkeypatch.setenv("KE_DECODE_PROCESSES", "0")
    by_threads = K.compute_signatures_mp(items, max_workers=3, chunksize=6)
    monkeypatch.setenv("KE_DECODE_PROCESSES", "1")
    monkeypatch.setenv("KE_DECODE_PROCESS_MIN", "1")
    import kobato_eyes_amd.fastsig as fs

    fs._stop_pools()
    by_processes = K.compute_signatures_mp(items, max_workers=3, chunksize=6)
    assert 3 in fs._pools                                        # the pool was made, i.e. the process route ran
    monkeypatch.setenv("KE_STAGE_BYTES", str(300 * 1024))       # 100 KB per worker: most files do not fit their region
    squeezed = K.compute_signatures_mp(items, max_workers=3, chunksize=6)
    assert by_processes == by_threads and squeezed == by_threads
    monkeypatch.setenv("KE_STAGE_BYTES", str(20 * 1024))        # no image fits a staging buffer at all: the workers report a
    tiny = K.compute_signatures_mp(items, max_workers=3, chunksize=6)    # spill, the thread route takes the files, each hashed alone
    assert tiny == by_threads
    assert [fid for fid, _, _ in by_threads] == [fid for fid, _ in items if fid < 70]
    for fid, ph, dh in by_processes:
        ep, ed = O.hash_image(arrays[fid])
        assert (ph, dh) == (O.to_signed64(ep), O.to_signed64(ed))


def test_ensure_signatures(K):
    conn = _make_conn()
    for file_id in range(1, 21):
        rng = np.random.default_rng(file_id)
        img = Image.fromarray((rng.random((64, 64, 3)) * 255).astype("uint8"))   # tests/core/test_image_signature.py:24-27
        assert K.ensure_signatures(conn, file_id, image=img) is True
        p, d = K.compute_signatures_from_image(img)
        row = conn.execute("SELECT phash_u64, dhash_u64 FROM signatures WHERE file_id=?", (file_id,)).fetchone()
        assert (row["phash_u64"], row["dhash_u64"]) == (p, d)
        ep, ed = O.hash_image(np.asarray(img))
        assert (p, d) == (O.to_signed64(ep), O.to_signed64(ed))
    assert conn.execute("SELECT COUNT(*) FROM signatures").fetchone()[0] == 20
    assert K.ensure_signatures(conn, 1) is True                      # row exists -> kept
    assert K.ensure_signatures(conn, 99) is False                    # nothing to compute from
    assert K.ensure_signatures(conn, 99, path="/nonexistent/file.png") is False


def test_refine_pair_like_the_reference_tests(K, tmp_path):
    from PIL import ImageEnhance

    a, b = tmp_path / "a.png", tmp_path / "b.png"
    Image.new("RGB", (64, 64), color=(200, 10, 10)).save(a)
    ImageEnhance.Brightness(Image.open(a).convert("RGB")).enhance(1.02).save(b)
    r = K.refine_pair(1, 2, a, b)                                   # tests/dup/test_refine.py:24-34
    assert isinstance(r, K.RefinedMatch) and r.is_duplicate and r.ssim > 0.95 and r.reason == "ssim>=0.9"
    assert abs(r.ssim - O.ssim_luma(np.asarray(Image.open(a).convert("L")), np.asarray(Image.open(b).convert("L")))) <= 1e-6
    g, bl = tmp_path / "g.png", tmp_path / "bl.png"
    Image.new("RGB", (64, 64), (0, 255, 0)).save(g)
    Image.new("RGB", (64, 64), (0, 0, 255)).save(bl)
    r = K.refine_pair(1, 3, g, bl, thresholds=K.RefinementThresholds(ssim=0.95, orb=0.5))   # :37-46
    assert not r.is_duplicate and r.reason == "below thresholds" and r.orb_ratio is None
    (tmp_path / "broken.png").write_bytes(b"not an image")
    assert K.refine_pair(1, 2, a, tmp_path / "broken.png") is None   # :49-55
    tiny = tmp_path / "tiny.png"
    Image.new("RGB", (5, 5)).save(tiny)
    r = K.refine_pair(1, 2, tiny, tiny)                              # SSIM raises for < 7 px -> "ssim unavailable"
    assert r.ssim is None and not r.is_duplicate and r.reason == "ssim unavailable"
    # two sizes of the same picture (the usual near-duplicate): both go through ImageOps.fit + BICUBIC on the GPU
    big, small = tmp_path / "big.png", tmp_path / "small.png"
    src = Image.fromarray(O.synth_rgb(5, 320, 240))
    src.save(big)
    src.resize((200, 160), Image.Resampling.LANCZOS).save(small)
    r = K.refine_pair(7, 8, big, small, thresholds=K.RefinementThresholds(ssim=0.5))
    exp = O.ssim_fit(np.asarray(Image.open(big).convert("RGB")), np.asarray(Image.open(small).convert("RGB")))
    assert abs(r.ssim - exp) <= 1e-6 and r.is_duplicate == (exp >= 0.5)
    from PIL import ImageOps
    fa = np.asarray(ImageOps.fit(Image.open(big).convert("L"), (200, 160), Image.Resampling.BICUBIC))
    fb = np.asarray(ImageOps.fit(Image.open(small).convert("L"), (200, 160), Image.Resampling.BICUBIC))
    assert abs(r.ssim - O.ssim_luma(fa, fb)) <= 1e-6                 # the reference's own preprocessing, Pillow-made


def test_find_duplicates_and_headless_scan(K, tmp_path):
    # corpus with planted variants: images 19, 29, 39 are variants of earlier bases
    idx = sorted({19, 29, 39, O.synth_info(19)[0], O.synth_info(29)[0], O.synth_info(39)[0], 0, 1, 2, 3})
    rows, arrays = [], {}
    for i in idx:
        arr = O.synth_rgb(i, 256, 256)
        p = tmp_path / f"c_{i:03d}.png"
        Image.fromarray(arr).save(p)
        rows.append({"file_id": i + 1, "path": str(p), "size": p.stat().st_size, "width": 256, "height": 256, "phash_u64": None})
        arrays[i + 1] = arr
    stages = []
    clusters = K.run_duplicate_scan(rows, config=K.DuplicateScanConfig(hamming_threshold=8),
                                    progress=lambda s, d, t: stages.append(s))
    assert [s for s in dict.fromkeys(stages)] == ["Loading files", "Computing signatures", "Building groups", "Clustering duplicates"]
    # expected clusters from the oracle end to end
    hashes = np.array([O.hash_image(arrays[r["file_id"]])[0] for r in rows], np.uint64)
    ids = np.array([r["file_id"] for r in rows], np.int64)
    e, _ = O.scan_banded(hashes, ids, threshold=8)
    exp = O.assemble_clusters([dict(r) for r in rows], [(int(ids[x["a"]]), int(ids[x["b"]]), int(x["h"])) for x in e])
    got = [(c.keeper_id, [(en.file.file_id, en.best_hamming) for en in c.files]) for c in clusters]
    assert got == exp and len(got) >= 2
    # find_duplicates alias on rows that already carry hashes, with and without the SSIM re-check
    for r, hv in zip(rows, hashes.tolist()):
        r["phash_u64"] = O.to_signed64(hv)
    plain = K.find_duplicates(rows + [{"file_id": 999, "path": "bad.png"}], hamming_threshold=8)
    assert [(c.keeper_id, [(en.file.file_id, en.best_hamming) for en in c.files]) for c in plain] == exp
    strict = K.find_duplicates(rows, hamming_threshold=8, ssim_threshold=0.999)
    loose = K.find_duplicates(rows, hamming_threshold=8, ssim_threshold=0.5)
    assert strict == [] and len(loose) == len(plain)


def test_refine_pairs_batches_the_seam(K, tmp_path):
    """refine_pairs == the per-pair loop of refine_pair (same RefinedMatch list, None for unreadable files, "ssim
    unavailable" for images under 7 px), with every file decoded once and a handful of launches instead of three per pair."""
    rng = np.random.default_rng(5)
    files = {}
    for k, (w, h) in enumerate([(256, 256)] * 10 + [(320, 240)] * 4 + [(200, 160), (5, 5), (6, 300)]):
        base = O.synth_rgb(1000 + 10 * (k // 2), w, h)
        if k % 2:
            base = np.clip(base.astype(np.int16) + rng.integers(-3, 4, base.shape), 0, 255).astype(np.uint8)
        p = tmp_path / f"r{k:02d}.png"
        Image.fromarray(base).save(p)
        files[k] = p
    (tmp_path / "broken.png").write_bytes(b"nope")
    files[99] = tmp_path / "broken.png"
    pairs = [(a, b, files[a], files[b]) for a, b in
             [(0, 1), (2, 3), (4, 5), (6, 7), (8, 9), (0, 2), (1, 9), (10, 11), (12, 13), (10, 14), (0, 14), (0, 10), (15, 15), (16, 0),
              (0, 99), (99, 1), (3, 4), (5, 6), (7, 8), (2, 9), (1, 3), (0, 9)]]
    th = K.RefinementThresholds(ssim=0.9)
    one_by_one = [K.refine_pair(a, b, pa, pb, thresholds=th) for a, b, pa, pb in pairs]
    stats = {}
    batched = K.refine_pairs(pairs, thresholds=th, stats=stats)
    assert batched == one_by_one
    assert sum(m is None for m in batched) == 2 and sum(m is not None and m.reason == "ssim unavailable" for m in batched) == 2
    assert any(m is not None and m.is_duplicate for m in batched) and any(m is not None and m.reason == "below thresholds" for m in batched)
    launches = stats["fit_launches"] + stats["ssim_launches"]
    assert stats["decodes"] == 18 and launches <= 12, stats    # the per-pair loop: 44 decodes, 2 fits + 1 SSIM per readable pair = 60 launches
    same_size = [p for p in pairs if p[0] < 10 and p[1] < 10]  # the shape of a scan over one camera's files: one size
    stats = {}
    assert K.refine_pairs(same_size, thresholds=th, stats=stats) == [m for m, p in zip(one_by_one, pairs) if p in same_size]
    assert stats["fit_launches"] + stats["ssim_launches"] == 2 and 3 * len(same_size) >= 10 * 2, stats   # 36 launches -> 2
    # a decode budget of one file per run: many runs, same answers
    assert K.refine_pairs(pairs, thresholds=th, max_decoded_bytes=1) == one_by_one


def test_refine_pairs_decodes_on_the_gpu_only_what_the_loader_leaves_alone(K, tmp_path):
    """refine_pairs takes JPEG / PNG files through the GPU decoders when the reference's defensive loader
    (src/utils/image_io.py:60-138) would hand over Image.open's pixels unchanged -- RGB, no EXIF orientation, no side over
    4096 -- and through that loader otherwise (rotated by EXIF, alpha over white, gray, palette with transparency, shrunk):
    the answers are refine_pair's either way."""
    from PIL import Image

    rng = np.random.default_rng(12)
    base = O.synth_rgb(2000, 320, 240)
    noisy = np.clip(base.astype(np.int16) + rng.integers(-4, 5, base.shape), 0, 255).astype(np.uint8)
    files = {}

    def put(k, name, img, **kw):
        files[k] = tmp_path / name
        img.save(files[k], **kw)

    put(0, "a.jpg", Image.fromarray(base), quality=92)
    put(1, "b.jpg", Image.fromarray(noisy), quality=85, progressive=True)
    put(2, "c.png", Image.fromarray(noisy))
    exif = Image.Exif()
    exif[0x0112] = 6                                            # stored rotated: the loader turns it, Image.open does not
    put(3, "rotated.jpg", Image.fromarray(base), quality=92, exif=exif.tobytes())
    rgba = np.dstack([base, np.full(base.shape[:2], 128, np.uint8)])
    put(4, "alpha.png", Image.fromarray(rgba, "RGBA"))
    put(5, "gray.jpg", Image.fromarray(base[:, :, 1]), quality=90)
    pal = Image.fromarray(base).convert("P")
    put(6, "palette_trns.png", pal, transparency=3)
    put(7, "palette.png", pal)
    wide = np.repeat(base[:40], 13, 1)[:, :4100]
    put(8, "wide.jpg", Image.fromarray(wide), quality=80)
    exif1 = Image.Exif()
    exif1[0x0112] = 1
    put(9, "upright.jpg", Image.fromarray(noisy), quality=88, exif=exif1.tobytes())
    # every orientation the tag can ask for (turned on the device: ke_normalise_rgb), alpha of every strength incl. 0 and 255
    for o in (2, 3, 4, 5, 7, 8):
        ex = Image.Exif()
        ex[0x0112] = o
        put(10 + o, f"turned{o}.jpg", Image.fromarray(noisy if o % 2 else base), quality=90, exif=ex.tobytes())
    alpha = rng.integers(0, 256, base.shape[:2], dtype=np.uint8)
    alpha[:8] = 0
    alpha[8:16] = 255
    put(20, "alpha_any.png", Image.fromarray(np.dstack([noisy, alpha]), "RGBA"))
    # the kinds added in round 3: BMP (RGB as it is, an alpha layout over white, a palette file through the loader), 16-bit and
    # Adam7 PNG (no writer in Pillow: hand-made containers)
    import _bmp_cases as B
    import _png_cases as P

    put(21, "d.bmp", Image.fromarray(noisy))
    files[22] = tmp_path / "alpha.bmp"
    files[22].write_bytes(B.bmp(320, 240, 32, B._rows(np.dstack([noisy[:, :, ::-1], alpha])), hs=124, comp=3, masks=(0xFF0000, 0xFF00, 0xFF, 0xFF000000)))
    put(23, "pal.bmp", pal)
    wide16 = np.stack([noisy, rng.integers(0, 256, noisy.shape, dtype=np.uint8)], -1).reshape(240, 320, 6)
    files[24] = tmp_path / "rgb16.png"
    files[24].write_bytes(P._container2(P._filtered(np.ascontiguousarray(wide16.reshape(240, -1)), 6, rng.integers(0, 5, 240)), 320, 240, 2, 16, 0))
    files[25] = tmp_path / "rgba_adam7.png"
    files[25].write_bytes(P._container2(P._adam7_stream(np.dstack([noisy, alpha]), 8, rng), 320, 240, 6, 8, 1))
    # a side over 4096: the loader's thumbnail((4096, 4096), LANCZOS) on the device (ke_thumbnail_rgb), after the turn where
    # there is one; beyond twice that size the loader itself (draft mode, reducing_gap)
    large = np.asarray(Image.fromarray(base).resize((5000, 3400), Image.Resampling.BICUBIC))
    put(26, "large.jpg", Image.fromarray(large), quality=80)
    ex6 = Image.Exif()
    ex6[0x0112] = 6
    put(27, "large_turned.jpg", Image.fromarray(np.ascontiguousarray(large.transpose(1, 0, 2)[:, ::-1])), quality=80, exif=ex6.tobytes())
    put(28, "large.png", Image.fromarray(np.repeat(np.repeat(noisy, 14, 0), 14, 1)[:3000, :4400]), compress_level=1)
    put(29, "huge.jpg", Image.fromarray(np.repeat(np.repeat(base, 26, 0), 26, 1)[:6000, :8300]), quality=60)
    put(30, "e.tif", Image.fromarray(noisy))                                       # uncompressed: the GPU route; RGBA over white on the device
    put(31, "alpha.tiff", Image.fromarray(np.dstack([noisy, alpha]), "RGBA"))
    put(32, "lzw.tif", Image.fromarray(noisy), compression="tiff_lzw")             # libtiff's in Pillow: the loader
    pairs = [(a, b, files[a], files[b]) for a, b in [(0, 1), (0, 2), (1, 2), (0, 3), (3, 2), (0, 4), (4, 2), (0, 5), (5, 1), (6, 0), (7, 2),
                                                      (6, 7), (8, 0), (8, 8), (9, 0), (9, 2), (12, 0), (13, 1), (14, 0), (15, 3), (17, 3),
                                                      (18, 15), (20, 4), (20, 2), (21, 2), (22, 20), (23, 7), (24, 2), (25, 20), (21, 0),
                                                      (26, 0), (27, 26), (28, 26), (29, 26), (30, 2), (31, 20), (32, 30)]]
    th = K.RefinementThresholds(ssim=0.8)
    one_by_one = [K.refine_pair(a, b, pa, pb, thresholds=th) for a, b, pa, pb in pairs]
    stats = {}
    assert K.refine_pairs(pairs, thresholds=th, stats=stats) == one_by_one
    # a.jpg, b.jpg, c.png, upright.jpg, d.bmp, rgb16.png as they are; rotated.jpg, the six turned*.jpg, alpha.png, alpha_any.png,
    # alpha.bmp and rgba_adam7.png normalised on the device; the 4100-pixel-wide file and the three large ones shrunk on the
    # device; gray, palette (PNG and BMP) and the 8300-pixel-wide file go through the loader
    assert stats["decodes"] == 29 and stats["gpu_decodes"] == 23 and stats["gpu_normalised"] == 12 and stats["gpu_shrunk"] == 4, stats
    by = {(a, b): m for (a, b, _, _), m in zip(pairs, one_by_one)}
    assert all(by[k].ssim > 0.999999 for k in ((21, 2), (24, 2), (22, 20), (25, 20), (30, 2), (31, 20), (32, 30)))    # same pixels, other container
    assert any(m.is_duplicate for m in one_by_one) and any(not m.is_duplicate for m in one_by_one)
    os.environ["KE_GPU_REFINE_DECODE"] = "0"
    try:
        stats = {}
        assert K.refine_pairs(pairs, thresholds=th, stats=stats) == one_by_one and stats["gpu_decodes"] == 0
    finally:
        del os.environ["KE_GPU_REFINE_DECODE"]


def test_shipped_refine_stage_kernels_and_dropins(K, tmp_path):
    """ke_resize_luma_uniform (BILINEAR) + ke_tile_ahash + ke_sad_pairs and the ui.dup_refine_parallel drop-ins
    against what the reference itself produced for the same files (tests/golden/refine_parallel_golden.json)."""
    import hashlib
    from dataclasses import dataclass

    import _golden as G
    from kobato_eyes_amd import refine_parallel as RP

    g = G.refine_parallel_golden()
    ctx = K._native.get_context(0)
    paths, thumbs = {}, {}
    for name, px in G.refine_corpus():
        p = tmp_path / f"{name}.png"
        Image.fromarray(px).save(p)
        paths[name] = p
        exp = g["cases"][name]
        for key, hexbits in exp["ahash"].items():
            grid, tile = (int(v) for v in key.split("x"))
            assert format(K.tile_ahash_bits(p, grid=grid, tile=tile), "x") == hexbits, (name, key)
        thumbs[name] = RP._load_small_gray(p, 128)
        assert hashlib.sha256(thumbs[name].tobytes()).hexdigest() == exp["thumb128_sha256"], name
        assert hashlib.sha256(RP._load_small_gray(p, 64).tobytes()).hexdigest() == exp["thumb64_sha256"], name
        h, w = px.shape[:2]
        ch = 1 if px.ndim == 2 else px.shape[2]
        lan = ctx.resize_luma_uniform(px[None], 1, w, h, ch, 32, 32, filter=0)[0]      # LANCZOS through the same entry point
        assert np.array_equal(lan, O.hash_image(px, want_tiles=True)[2]), name
    for a, b, mae in g["mae"]:
        assert RP._mae01(thumbs[a], thumbs[b]) == mae
    # the same thumbnails with the files decoded on the GPU (what refine_by_tilehash_parallel does for .jpg / .png), JPEG files
    # stored rotated included: == Image.open + exif_transpose + resize
    turned = tmp_path / "turned.jpg"
    exif = Image.Exif()
    exif[0x0112] = 8
    Image.fromarray(next(px for _, px in G.refine_corpus() if px.ndim == 3 and px.shape[2] == 3)).save(turned, quality=90, exif=exif.tobytes())
    plain = tmp_path / "plain.jpg"
    Image.fromarray(next(px for _, px in G.refine_corpus() if px.ndim == 3 and px.shape[2] == 3)).save(plain, quality=90)
    more = []
    for o in (2, 3, 4, 5, 6, 7):                     # every other orientation, on images of two shapes
        ex = Image.Exif()
        ex[0x0112] = o
        src = [px for _, px in G.refine_corpus() if px.ndim == 3 and px.shape[2] == 3][o % 3]
        more.append(tmp_path / f"turned{o}.jpg")
        Image.fromarray(src).save(more[-1], quality=90, exif=ex.tobytes())
    on_gpu = RP._thumbnails_decoded_on_gpu(list(paths.values()) + [turned, plain] + more, 32, 0)
    assert set(on_gpu) == set(paths.values()) | {plain, turned} | set(more)        # turned on the device (ke_normalise_rgb)
    for p, t in on_gpu.items():
        assert np.array_equal(t, RP._thumbnails([RP._decode(p)], 32, 0)[0]), p

    @dataclass
    class F:
        file_id: int
        path: object

    @dataclass
    class E:
        file: F

    @dataclass
    class Cl:
        files: list
        keeper_id: int

    ids = g["ids"]
    clusters = [Cl([E(F(ids[n], paths[n])) for n in c["members"]], ids[c["keeper"]]) for c in g["cluster_inputs"]]
    ticks = []
    for case in g["clusters"]:
        if case["stage"] == "tilehash":
            res = K.refine_by_tilehash_parallel(clusters, grid=4, tile=8, max_bits=case["max_bits"], io_workers=2,
                                                tick=lambda d, t, phase: ticks.append((phase, d, t)))
            got = [[cl.keeper_id, [e.file.file_id for e in cl.files]] for cl in res]
            assert got == case["result"], case
            assert all(isinstance(cl, Cl) for cl in res)
        else:
            res = K.refine_by_pixels_parallel(clusters, mae_thr=case["mae_thr"], thumb_size=128, workers=2)
            assert sorted([cl.keeper_id, [e.file.file_id for e in cl.files]] for cl in res) == case["result"], case
    assert (1, 12, 12) in ticks and (2, 5, 5) in ticks              # progress at the end of each phase
    assert K.refine_by_tilehash_parallel(clusters, is_cancelled=lambda: True) == []
    broken = tmp_path / "broken.png"
    broken.write_bytes(b"nope")
    bad = [Cl([E(F(1, paths["v007_256"])), E(F(2, broken)), E(F(3, paths["v019_256"]))], 1)]
    res = K.refine_by_tilehash_parallel(bad, max_bits=200)          # unreadable member skipped, cluster survives
    assert [[cl.keeper_id, [e.file.file_id for e in cl.files]] for cl in res] == [[1, [1, 3]]]
    res = K.refine_by_pixels_parallel(bad, mae_thr=0.05)
    assert [[cl.keeper_id, [e.file.file_id for e in cl.files]] for cl in res] == [[1, [1, 3]]]


def test_cli_scan_dups_over_sqlite(K, tmp_path):
    """Headless scan of a database laid out like the reference's (files + signatures), CSV in the reference's layout."""
    import csv

    from kobato_eyes_amd import cli

    db = tmp_path / "kobato.db"
    conn = sqlite3.connect(db)
    conn.execute("CREATE TABLE files (id INTEGER PRIMARY KEY AUTOINCREMENT, path TEXT NOT NULL UNIQUE, size INTEGER, "
                 "is_present INTEGER NOT NULL DEFAULT 1, width INTEGER, height INTEGER)")
    conn.execute("CREATE TABLE signatures (file_id INTEGER PRIMARY KEY, phash_u64 INTEGER NOT NULL, dhash_u64 INTEGER NOT NULL)")
    idx = sorted({19, 29, O.synth_info(19)[0], O.synth_info(29)[0], 0, 1, 2})
    rows, arrays = [], {}
    for i in idx:
        arr = O.synth_rgb(i, 256, 256)
        p = tmp_path / f"f_{i:03d}.png"
        Image.fromarray(arr).save(p)
        cur = conn.execute("INSERT INTO files (path, size, width, height) VALUES (?, ?, 256, 256)", (str(p), p.stat().st_size))
        rows.append({"file_id": cur.lastrowid, "path": str(p), "size": p.stat().st_size, "width": 256, "height": 256})
        arrays[cur.lastrowid] = arr
    conn.execute("INSERT INTO files (path, size, is_present) VALUES (?, 1, 0)", (str(tmp_path / "gone.png"),))   # not present
    conn.execute("INSERT INTO files (path, size) VALUES (?, 1)", (str(tmp_path / "missing.png"),))               # unreadable -> skipped
    # one signature is already stored (resume): it must be used, not recomputed
    first = rows[0]["file_id"]
    ep, ed = O.hash_image(arrays[first])
    conn.execute("INSERT INTO signatures VALUES (?, ?, ?)", (first, O.to_signed64(ep), O.to_signed64(ed)))
    conn.commit()
    conn.close()
    out_csv = tmp_path / "dups.csv"
    assert cli.main(["scan-dups", "--db", str(db), "--hamming", "8", "--csv", str(out_csv)]) == 0
    hashes = np.array([O.hash_image(arrays[r["file_id"]])[0] for r in rows], np.uint64)
    ids = np.array([r["file_id"] for r in rows], np.int64)
    e, _ = O.scan_banded(hashes, ids, threshold=8)
    exp = O.assemble_clusters(rows, [(int(ids[x["a"]]), int(ids[x["b"]]), int(x["h"])) for x in e])
    with open(out_csv, newline="") as fh:
        got = list(csv.reader(fh))
    assert got[0] == ["group", "file_id", "path", "size", "width", "height", "keeper", "hamming"]
    by_id = {r["file_id"]: r for r in rows}
    want = []
    for gi, (keeper, entries) in enumerate(exp, 1):
        for fid, best in entries:
            r = by_id[fid]
            want.append([str(gi), str(fid), Path(r["path"]).as_posix(), str(r["size"]), "256", "256", "1" if fid == keeper else "0",
                         "" if best is None else str(best)])
    assert got[1:] == want and len(exp) >= 1
    stored = sqlite3.connect(db).execute("SELECT COUNT(*) FROM signatures").fetchone()[0]
    assert stored == len(rows)                                       # every readable present file now has a row


def test_fast_fill_can_be_cancelled_half_way_and_called_again(K, tmp_path, monkeypatch):
    """A cancel callback stops the run between batches / chunks (src/core/fastsig.py:86-90: the partial list comes back, in
    input order); read-ahead buffers, staging buffers and the Pillow share's thread are given back: the next call works."""
    items, arrays = _corpus(tmp_path, n=12)
    more = []
    for k in range(300):                                             # JPEG, PNG and BMP copies: every route has work
        src = arrays[1 + k % 12]
        p = tmp_path / f"more_{k:04d}.{('jpg', 'png', 'bmp')[k % 3]}"
        Image.fromarray(src).save(p)
        more.append((1000 + k, str(p)))
    monkeypatch.setenv("KE_GPU_BATCH_CANCELLABLE", "64")             # several batches
    full = K.compute_signatures_mp(more, max_workers=3, chunksize=8)
    assert [r[0] for r in full] == [fid for fid, _ in more]
    calls = {"n": 0}

    def cancel():
        calls["n"] += 1
        return calls["n"] > 150

    part = K.compute_signatures_mp(more, max_workers=3, chunksize=8, cancel_fn=cancel)
    assert 0 < len(part) < len(full) and part == full[:len(part)]
    assert K.compute_signatures_mp(more, max_workers=3, chunksize=8, cancel_fn=lambda: True) == []
    assert K.compute_signatures_mp(more, max_workers=3, chunksize=8) == full          # nothing was left behind


def test_shipped_refine_stage_on_turned_and_transparent_files_equals_the_reference(K, tmp_path):
    """JPEG files of every EXIF orientation, RGBA and gray + alpha PNG through tile_ahash_bits / _load_small_gray / _mae01 /
    refine_by_tilehash_parallel / refine_by_pixels_parallel == what the reference's own ui.dup_refine_parallel returned for
    the same bytes (tests/golden/refine_turned_golden.json) -- by the GPU decoders with the turn applied on the device
    (ke_normalise_rgb), and with KE_GPU_REFINE_DECODE=0 (Pillow)."""
    import hashlib
    from dataclasses import dataclass

    import _golden as G
    from kobato_eyes_amd import refine_parallel as RP

    g, blobs = G.refine_turned_golden()
    paths = {}
    for name, data in blobs.items():
        paths[name] = tmp_path / name
        paths[name].write_bytes(data)

    @dataclass
    class F:
        file_id: int
        path: object

    @dataclass
    class E:
        file: F

    @dataclass
    class Cl:
        files: list
        keeper_id: int

    ids = g["ids"]
    clusters = [Cl([E(F(ids[n], paths[n])) for n in c["members"]], ids[c["keeper"]]) for c in g["cluster_inputs"]]
    for route in ("gpu", "pillow"):
        if route == "pillow":
            os.environ["KE_GPU_REFINE_DECODE"] = "0"
        try:
            if route == "gpu":
                on_gpu = RP._thumbnails_decoded_on_gpu(list(paths.values()), 128, 0)
                assert set(on_gpu) == set(paths.values())                      # every one of these files stays on the GPU route
                for name, p in paths.items():
                    assert hashlib.sha256(on_gpu[p].tobytes()).hexdigest() == g["cases"][name]["thumb128_sha256"], name
                on_gpu32 = RP._thumbnails_decoded_on_gpu(list(paths.values()), 32, 0)
                for name, p in paths.items():
                    assert hashlib.sha256(on_gpu32[p].tobytes()).hexdigest() == g["cases"][name]["thumb32_sha256"], name
            for name, p in paths.items():
                for key, hexbits in g["cases"][name]["ahash"].items():
                    grid, tile = (int(v) for v in key.split("x"))
                    assert format(K.tile_ahash_bits(p, grid=grid, tile=tile), "x") == hexbits, (route, name, key)
                assert hashlib.sha256(RP._load_small_gray(p, 128).tobytes()).hexdigest() == g["cases"][name]["thumb128_sha256"], (route, name)
            for a, b, mae in g["mae"]:
                assert RP._mae01(RP._load_small_gray(paths[a], 128), RP._load_small_gray(paths[b], 128)) == mae
            for c in g["clusters"]:
                if c["stage"] == "tilehash":
                    res = K.refine_by_tilehash_parallel(clusters, grid=4, tile=8, max_bits=c["max_bits"], io_workers=2)
                    assert [[cl.keeper_id, [e.file.file_id for e in cl.files]] for cl in res] == c["result"], (route, c["max_bits"])
                else:
                    res = K.refine_by_pixels_parallel(clusters, mae_thr=c["mae_thr"], thumb_size=128, workers=1)
                    assert sorted([cl.keeper_id, [e.file.file_id for e in cl.files]] for cl in res) == c["result"], (route, c["mae_thr"])
        finally:
            os.environ.pop("KE_GPU_REFINE_DECODE", None)
