"""CPU tests: the oracle (oracle/keyes_oracle.c) against the golden vectors produced by the
reference itself (tests/golden/make_golden.py) and against Pillow/NumPy where installed."""
from __future__ import annotations

import hashlib

import numpy as np
import pytest

import _golden as G
from oracle import oracle as O


def test_sig_golden_tiles_dhash_phash():
    n = 0
    for name, px, t32, t98, ph, dh, margin, sha in G.sig_cases():
        assert hashlib.sha256(np.ascontiguousarray(px).tobytes()).hexdigest() == sha, f"input drifted: {name}"
        got_ph, got_dh, g32, g98, _ = O.hash_image(px, want_tiles=True)
        assert np.array_equal(g32, t32), name   # reference _to_grayscale(32,32), src/sig/phash.py:21-26
        assert np.array_equal(g98, t98), name   # reference _to_grayscale(9,8)
        assert got_dh == dh, name               # reference dhash, src/sig/phash.py:49-57 (integer, fully pinned)
        # pHash: reference phash() with the SciPy stand-in for cv2.dct (parity unpinned vs OpenCV)
        assert got_ph == ph, f"{name}: margin={margin}"
        n += 1
    assert n >= 70


def test_resample_matches_installed_pillow():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(42)
    for (w, h) in [(256, 256), (300, 451), (1000, 37), (16, 16), (33, 31), (7, 9), (1, 1), (2, 500), (4, 500),
                   (5, 500), (10, 3000), (640, 480), (31, 32), (32, 40), (2048, 64)]:
        arr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        im = Image.fromarray(arr)
        L = np.asarray(im.convert("L"))
        assert np.array_equal(O.luma(arr), L)
        for (ow, oh) in [(32, 32), (9, 8)]:
            ref = np.asarray(im.convert("L").resize((ow, oh), Image.Resampling.LANCZOS))
            assert np.array_equal(O.resample(L, ow, oh), ref), (w, h, ow, oh)


def test_numpy_float32_mean_order():
    """The oracle's restatement of `flat[1:].mean()` (src/sig/phash.py:41) is NumPy's own order."""
    rng = np.random.default_rng(3)
    for _ in range(300):
        tile = rng.integers(0, 256, (32, 32), dtype=np.uint8)
        c = O.dct8x8(tile).astype(np.float32).flatten()
        mean = c[1:].mean()
        bits = 0
        for b in (c > mean):
            bits = (bits << 1) | int(b)
        assert bits == O.phash_from_tile(tile)[0]


def test_dct_matches_scipy_and_flat_is_exact():
    dctn = pytest.importorskip("scipy.fft").dctn
    rng = np.random.default_rng(4)
    tile = rng.integers(0, 256, (32, 32), dtype=np.uint8)
    ref = dctn(tile.astype(np.float64), type=2, norm="ortho")[:8, :8]
    assert np.allclose(O.dct8x8(tile), ref, rtol=0, atol=1e-9)
    flat = O.dct8x8(np.full((32, 32), 77, np.uint8))
    assert flat[0, 0] == 77 * 32 and np.count_nonzero(flat) == 1   # every AC term exactly zero
    assert O.phash_from_tile(np.full((32, 32), 77, np.uint8))[0] == 1 << 63


def test_signed_wrap_and_hamming():
    # tests/core/test_image_signature.py:58-68, tests/core/test_fastsig.py (wrap incl. (1<<64)+7 -> 7)
    for v, e in [(0, 0), ((1 << 64) - 1, -1), (1 << 63, -(1 << 63)), ((1 << 63) - 1, (1 << 63) - 1), ((1 << 64) + 7, 7)]:
        assert O.to_signed64(v) == e
    assert O.hamming64(-1, 0) == 64 and O.hamming64(0xF0, 0x0F) == 8


@pytest.mark.parametrize("name", sorted(G.scan_scenarios()))
def test_scan_and_clusters_match_reference(name):
    sc = G.scan_scenarios()[name]
    files, cfg = sc["files"], sc["config"]
    if not files:
        assert sc["edges"] == [] and sc["clusters"] == []
        return
    hashes, ids, sizes = G.files_to_arrays(files)
    edges, counters = O.scan_banded(hashes, ids, sizes, threshold=cfg["hamming_threshold"],
                                    band_bits=cfg.get("band_bits", 16), band_count=cfg.get("band_count", 4),
                                    size_ratio=cfg.get("size_ratio"), bucket_pair_cap=sc["bucket_pair_cap"])
    got = sorted([int(min(ids[e["a"]], ids[e["b"]])), int(max(ids[e["a"]], ids[e["b"]])), int(e["h"])] for e in edges)
    assert got == sc["edges"]
    if sc["counters"] is not None:
        assert [int(c) for c in counters] == sc["counters"][:3]   # funnel log, src/dup/scanner.py:292-299
    clusters = O.assemble_clusters(files, got)
    assert clusters == [(c["keeper_id"], [tuple(e) for e in c["entries"]]) for c in sc["clusters"]]


def test_banded_equals_closed_form():
    """edge(i,j) <=> popc<=T and some band lane of x^y is zero."""
    h = O.synth_hashes(3000)
    for (t, bb, bc) in [(8, 16, 4), (10, 16, 4), (12, 8, 8), (6, 32, 2), (64, 16, 4)]:
        a, _ = O.scan_banded(h, threshold=t, band_bits=bb, band_count=bc)
        b = O.scan_bruteforce(h, threshold=t, band_bits=bb, band_count=bc)
        assert sorted(map(tuple, a[["a", "b", "h", "bands"]].tolist())) == sorted(map(tuple, b[["a", "b", "h", "bands"]].tolist()))
    # banding is a strict subset of all-pairs at T=8 (SURVEY finding 3)
    h1k = O.synth_hashes(1000)
    banded = O.scan_bruteforce(h1k, threshold=8, band_bits=16, band_count=4)
    x = h1k[:, None] ^ h1k[None, :]
    pc = np.zeros(x.shape, np.int64)
    for b in range(64):
        pc += ((x >> np.uint64(b)) & np.uint64(1)).astype(np.int64)
    assert len(banded) == 73 and int(np.triu(pc <= 8, 1).sum()) == 86


def test_ssim_matches_restated_skimage():
    for name, a, b, exp in G.ssim_cases():
        assert abs(O.ssim_luma(a, b) - exp) <= 1e-6, name
    # tests/dup/test_refine.py:24-46 inequalities
    cases = {n: O.ssim_luma(a, b) for n, a, b, _ in G.ssim_cases()}
    assert cases["ref_solid_bright"] > 0.95
    assert cases["ref_green_blue"] < 0.95


def test_fit_bicubic_matches_pillow_golden():
    """src/dup/refine.py:45-49: ImageOps.fit(convert("L"), (min w, min h), BICUBIC) -- the golden tiles are
    Pillow's own output for these inputs; SSIM beside them is the SciPy restatement (unpinned vs skimage)."""
    n = 0
    for name, px_a, px_b, (w, h), fa, fb, ssim in G.fit_cases():
        assert np.array_equal(O.fit_luma(O.luma(px_a), w, h), fa), name
        assert np.array_equal(O.fit_luma(O.luma(px_b), w, h), fb), name
        if min(w, h) >= 7:
            assert abs(O.ssim_fit(px_a, px_b) - ssim) <= 1e-6, name
        n += 1
    for px, (ow, oh), exp in G.fit_extra_cases():
        assert np.array_equal(O.fit_luma(px, ow, oh), exp), (px.shape, ow, oh)
        n += 1
    assert n >= 15


def test_fit_bicubic_matches_installed_pillow():
    Image = pytest.importorskip("PIL.Image")
    from PIL import ImageOps
    rng = np.random.default_rng(7)
    for _ in range(60):
        w, h = int(rng.integers(1, 300)), int(rng.integers(1, 300))
        ow, oh = int(rng.integers(1, 200)), int(rng.integers(1, 200))
        L = rng.integers(0, 256, (h, w), dtype=np.uint8)
        ref = np.asarray(ImageOps.fit(Image.fromarray(L, "L"), (ow, oh), Image.Resampling.BICUBIC))
        assert np.array_equal(O.fit_luma(L, ow, oh), ref), (w, h, ow, oh)


def test_cluster_builder_matches_reference_test():
    # tests/dup/test_cluster.py:9-23
    out = O.cluster_builder([(1, 2, True), (2, 3, True), (4, 5, True), (3, 5, False)])
    assert [c[1] for c in out] == [[1, 2, 3], [4, 5]]
    assert [c[0] for c in out] == [1, 4]


def test_synth_generator_is_stable():
    img = O.synth_rgb(19, 64, 48)
    assert img.shape == (48, 64, 3)
    base, delta, variant = O.synth_info(19)
    assert variant and base < 19 and base % 10 != 9 and -3 <= delta <= 3
    assert not O.synth_info(18)[2]
    h = O.synth_hashes(1000)
    assert len(np.unique(h[:900])) == 900


def test_shipped_refine_stage_matches_reference():
    """Oracle restatement of ui.dup_refine_parallel (tile aHash, 128x128 BILINEAR thumbnails, MAE) against the
    values the reference produced on the same images (tests/golden/refine_parallel_golden.json)."""
    g = G.refine_parallel_golden()
    thumbs = {}
    for name, px in G.refine_corpus():
        exp = g["cases"][name]
        for key, hexbits in exp["ahash"].items():
            grid, tile = (int(v) for v in key.split("x"))
            assert format(O.tile_ahash_bits(px, grid, tile), "x") == hexbits, (name, key)
        thumbs[name] = O.small_gray(px, 128)
        assert hashlib.sha256(thumbs[name].tobytes()).hexdigest() == exp["thumb128_sha256"], name
        assert hashlib.sha256(O.small_gray(px, 64).tobytes()).hexdigest() == exp["thumb64_sha256"], name
    for a, b, mae in g["mae"]:
        assert O.mae01(thumbs[a], thumbs[b]) == mae


def test_bilinear_resample_matches_installed_pillow():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(8)
    for (w, h) in [(512, 512), (300, 451), (16, 16), (1000, 37), (2, 500), (127, 129), (4096, 64)]:
        arr = rng.integers(0, 256, (h, w), dtype=np.uint8)
        for (ow, oh) in [(32, 32), (128, 128), (16, 24)]:
            ref = np.asarray(Image.fromarray(arr).resize((ow, oh), Image.Resampling.BILINEAR))
            assert np.array_equal(O.resample_filter(arr, ow, oh, 1), ref), (w, h, ow, oh)


def test_degenerate_tiles_are_pinned():
    """Flat and two-level images: this build's tie policy, fixed (see tests/_golden.py:degenerate_tiles)."""
    for name, px, ph, dh, margin in G.degenerate_tiles():
        got_p, got_d, _, _, got_m = O.hash_image(px, want_tiles=True)
        assert (got_p, got_d) == (ph, dh), name
        assert np.float32(got_m) == np.float32(margin), name


def test_loader_normalisation_matches_installed_pillow(tmp_path):
    """oracle.normalise_rgb (what ke_normalise_rgb is held against) == ImageOps.exif_transpose of a file carrying each
    orientation, and == Image.alpha_composite over white + convert("RGB") for every (channel value, alpha) pair."""
    Image = pytest.importorskip("PIL.Image")
    from PIL import ImageOps

    rng = np.random.default_rng(17)
    a = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    for o in range(1, 9):
        ex = Image.Exif()
        ex[0x0112] = o
        p = tmp_path / f"o{o}.png"                              # PNG: lossless, and Pillow reads its eXIf chunk
        Image.fromarray(a).save(p, exif=ex.tobytes())
        with Image.open(p) as im:
            ref = np.asarray(ImageOps.exif_transpose(im))
        assert np.array_equal(O.normalise_rgb(a, o), ref), o
    cc, aa = np.meshgrid(np.arange(256), np.arange(256))
    px = np.stack([cc, 255 - cc, (cc * 7) % 256, aa], -1).astype(np.uint8)
    bg = Image.new("RGBA", (256, 256), "WHITE")
    bg.alpha_composite(Image.fromarray(px, "RGBA"))
    assert np.array_equal(O.normalise_rgb(px), np.asarray(bg.convert("RGB")))
