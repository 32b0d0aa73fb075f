"""GPU JPEG decoder (ke_jpeg_probe / ke_jpeg_decode, csrc/ke_jpeg.hip) against the installed Pillow: the pixels of
`Image.open(file)`, bit for bit, for every file the decoder takes -- one mixed batch (sizes, samplings, grayscale, restart
markers together) -- the right per-file refusal for the rest, and the hashes of the decode -> hash route without the pixels
leaving the GPU against the oracle on Pillow's pixels."""
from __future__ import annotations

import os

import numpy as np
import pytest

import _jpeg_cases as J
import _png_cases as P
import _bmp_cases as B
import _gif_cases as GF
import _tiff_cases as TF
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from kobato_eyes_amd import _native

    return _native.get_context(0)


def test_decode_matches_pillow_in_one_mixed_batch(ctx):
    cases = list(J.supported(full=True))
    refused = list(J.refused())
    blobs = [c[1] for c in cases] + [r[1] if r[1] else b"\0" for r in refused]
    out, status = ctx.jpeg_decode(blobs)
    for k, (name, _, ref) in enumerate(cases):
        assert status[k] == 0, name
        assert out[k].shape == ref.shape and np.array_equal(out[k], ref), name
    for k, (name, _, expected) in enumerate(refused, len(cases)):
        assert status[k] == expected and out[k] is None, name
    assert len(cases) > 400
    w, h, c, st = ctx.jpeg_probe(blobs[:3])
    assert (w[0], h[0], c[0], st[0]) == (cases[0][2].shape[1], cases[0][2].shape[0], 3, 0)


def test_decode_and_hash_without_leaving_the_gpu(ctx):
    cases = [c for c in J.supported() if min(c[2].shape[:2]) >= 16][:120]
    refused = list(J.refused())[:2]
    blobs = [c[1] for c in cases] + [r[1] for r in refused]
    ph, dh, status = ctx.jpeg_hash(blobs)
    for k, (name, _, ref) in enumerate(cases):
        assert status[k] == 0, name
        assert (int(ph[k]), int(dh[k])) == O.hash_image(ref), name
    assert status[len(cases):].tolist() == [r[2] for r in refused]


def test_large_batch_of_equal_files_and_damage(ctx):
    """4 096 files in one call (64 waves of the entropy kernel), every 97th damaged in its entropy data: the damaged ones are
    reported, the rest are untouched by their neighbours' failure."""
    import io

    from PIL import Image

    rng = np.random.default_rng(5)
    a = np.clip(np.repeat(np.repeat(rng.integers(0, 256, (16, 16, 3)), 16, 0), 16, 1) + rng.integers(-5, 6, (256, 256, 3)), 0, 255).astype(np.uint8)
    b = io.BytesIO()
    Image.fromarray(a).save(b, "JPEG", quality=88)
    good = b.getvalue()
    ref = np.asarray(Image.open(io.BytesIO(good)))
    cut = good[: len(good) // 2] + b"\xff\xd9"                 # entropy data ends early
    blobs = [cut if k % 97 == 5 else good for k in range(4096)]
    ph, dh, status = ctx.jpeg_hash(blobs)
    exp = O.hash_image(ref)
    for k in range(4096):
        if k % 97 == 5:
            assert status[k] != 0
        else:
            assert status[k] == 0 and (int(ph[k]), int(dh[k])) == exp, k


def test_fast_fill_takes_the_gpu_route_for_jpeg_files(tmp_path, monkeypatch):
    """A folder of baseline, progressive, grayscale and damaged JPEGs plus PNGs through fast_fill_missing_signatures: rows equal
    the Pillow route's (KE_GPU_JPEG=0) and the oracle on Pillow's pixels, in input order, failures dropped."""
    import io

    from PIL import Image

    import kobato_eyes_amd as K

    rng = np.random.default_rng(9)
    items, expect = [], {}
    for k in range(40):
        base = np.repeat(np.repeat(rng.integers(0, 256, (12, 16, 3)), 16, 0), 16, 1).astype(np.int16)
        arr = np.clip(base + rng.integers(-4, 5, base.shape), 0, 255).astype(np.uint8)
        kind = k % 5
        name = tmp_path / (f"f{k:02d}.png" if kind == 4 else f"f{k:02d}.jpg" if kind != 2 else f"f{k:02d}.JPEG")
        if kind == 4:
            Image.fromarray(arr).save(name)
        elif kind == 0:
            Image.fromarray(arr).save(name, "JPEG", quality=90, subsampling=k % 3)
        elif kind == 1:
            Image.fromarray(arr).save(name, "JPEG", quality=80, progressive=True)
        elif kind == 2:
            Image.fromarray(arr[:, :, 0]).save(name, "JPEG", quality=85)
        else:
            b = io.BytesIO()
            Image.fromarray(arr).save(b, "JPEG", quality=85)
            name.write_bytes(b.getvalue()[:300])                   # damaged: Pillow raises -> the file is dropped
        items.append((k + 1, str(name)))
        try:
            with Image.open(name) as im:
                expect[k + 1] = O.hash_image(np.asarray(im.convert("RGB") if im.mode not in ("L", "RGB") else im))
        except OSError:
            pass
    items.insert(7, (900, str(tmp_path / "gone.jpg")))             # no such file: dropped by either route
    (tmp_path / "empty.png").write_bytes(b"")
    items.insert(19, (901, str(tmp_path / "empty.png")))
    rows = K.compute_signatures_mp(items, max_workers=4, chunksize=16)
    monkeypatch.setenv("KE_GPU_BATCH", "5")                        # GPU batches of 16 (never below the chunk size), then of 7
    assert K.compute_signatures_mp(items, max_workers=4, chunksize=16) == rows
    assert K.compute_signatures_mp(items, max_workers=4, chunksize=7) == rows
    # the batches above were read ahead of their decode (Context.read_files_ahead); without, and with a run given up half way
    from kobato_eyes_amd import _native, fastsig

    context = _native.get_context(0)
    assert context._ahead[0][0] and not any(buf[2] for buf in context._ahead)
    monkeypatch.setenv("KE_READ_AHEAD", "0")
    assert K.compute_signatures_mp(items, max_workers=4, chunksize=7) == rows
    monkeypatch.delenv("KE_READ_AHEAD")
    batches = fastsig._Pipeline(items, 4, 7, 0).run_batches()
    next(batches)
    batches.close()                                                # the batch being read ahead gives its buffer back
    assert not any(buf[2] for buf in context._ahead)
    assert K.compute_signatures_mp(items, max_workers=4, chunksize=7) == rows
    monkeypatch.setenv("KE_GPU_JPEG", "0")
    monkeypatch.setenv("KE_GPU_PNG", "0")
    rows_pillow = K.compute_signatures_mp(items, max_workers=4, chunksize=16)
    assert rows == rows_pillow
    assert [r[0] for r in rows] == sorted(expect)
    for fid, ph, dh in rows:
        assert (ph, dh) == tuple(O.to_signed64(v) for v in expect[fid]), fid


def test_png_decode_matches_pillow_in_one_mixed_batch(ctx):
    """ke_png_decode: files of every kind (8-bit L / RGB / RGBA / LA, palette and sub-byte, Adam7, 16-bit) and every compression
    level in one call -- stored, fixed and dynamic deflate blocks, all five filters -- equal to what Pillow opens them to;
    damaged files and files beyond the decoder's limits reported per file."""
    cases = list(P.supported(full=True)) + list(P.handmade(full=True)) + list(P.mapped(full=True)) + list(P.interlaced(full=True)) + \
        list(P.wide(full=True)) + [c for c in P.animated(full=True) if c[2] is not None]
    refused = list(P.refused()) + [(c[0], c[1], 1) for c in P.animated() if c[2] is None]
    blobs = [c[1] for c in cases] + [r[1] for r in refused]
    out, status = ctx.png_decode(blobs)
    for k, (name, _, ref) in enumerate(cases):
        assert status[k] == 0, name
        assert out[k].shape == ref.shape and np.array_equal(out[k], ref), name
    for k, (name, _, expected) in enumerate(refused, len(cases)):
        assert status[k] == expected and out[k] is None, name
    assert len(cases) > 300
    ph, dh, st = ctx.png_hash([c[1] for c in cases if min(c[2].shape[:2]) >= 16][:60])
    refs = [c[2] for c in cases if min(c[2].shape[:2]) >= 16][:60]
    for k, ref in enumerate(refs):
        assert st[k] == 0 and (int(ph[k]), int(dh[k])) == O.hash_image(ref), k


def test_bmp_unpack_matches_pillow_in_one_mixed_batch(ctx):
    """ke_bmp_decode: 24-bit, 32-bit (BGRX and every BITFIELDS layout Pillow lists), 8-bit palette and gray files, bottom-up
    and top-down, every header size, data offsets with gaps -- equal to what the reference's hashes see of them through
    Pillow; RLE / 1 / 4 / 16-bit / OS/2 files and truncated ones reported per file; header damage never decoded differently."""
    import test_bmp_cpu as TB

    cases = [c for c in list(B.supported(full=True)) + list(B.handmade(full=True))]
    refused = list(B.refused())
    blobs = [c[1] for c in cases] + [r[1] for r in refused]
    out, status = ctx.bmp_decode(blobs)
    taken = 0
    for k, (name, _, ref) in enumerate(cases):
        if ref is None:
            assert status[k] != 0 and out[k] is None, name
            continue
        assert status[k] == 0, name
        assert out[k].shape == ref.shape and np.array_equal(out[k], ref), name
        taken += 1
    for k, (name, _, expected) in enumerate(refused, len(cases)):
        assert status[k] == expected and out[k] is None, name
    assert taken > 150
    big = [c for c in cases if c[2] is not None and min(c[2].shape[:2]) >= 16]
    ph, dh, st = ctx.bmp_hash([c[1] for c in big])
    for k, c in enumerate(big):
        assert st[k] == 0 and (int(ph[k]), int(dh[k])) == O.hash_image(c[2]), c[0]
    rng = np.random.default_rng(32)
    pool = [c for c in cases if c[2] is not None and c[2].shape[0] <= 80]
    damaged = list(TB.damaged(rng, pool, 12))
    out, status = ctx.bmp_decode([d for _, d in damaged])
    decoded = 0
    for (name, data), px, st in zip(damaged, out, status):
        if st == 0:
            decoded += 1
            ref = TB.strict_pillow(data)
            assert ref is not None and ref.shape == px.shape and np.array_equal(ref, px), name
    assert decoded > 100


def test_gif_first_frame_matches_pillow_in_one_mixed_batch(ctx, tmp_path):
    """ke_gif_decode: Pillow-written GIFs (palette, gray, interlaced, optimised, transparent, animated) and hand-made streams
    (clear codes at any distance or never, long runs, 3-byte sub-blocks, local / short / no palettes, nothing behind the last
    pixel) in one call -- the luma the reference's hashes see of the first frame; frames that do not cover the screen, early
    end codes and truncated data reported per file; damaged files never decoded differently from Pillow; .gif files take the
    route in the batch hasher."""
    import test_gif_cpu as TG

    cases = list(GF.supported(full=True)) + list(GF.handmade(full=True))
    refused = list(GF.refused())
    out, status = ctx.gif_decode([c[1] for c in cases] + [r[1] for r in refused])
    taken = 0
    for k, (name, _, ref) in enumerate(cases):
        if ref is None or name.startswith(GF.LEFT_TO_PILLOW):
            assert status[k] != 0 and out[k] is None, name
            continue
        assert status[k] == 0, name
        assert out[k].shape == ref.shape and np.array_equal(out[k], ref), name
        taken += 1
    for k, (name, _, expected) in enumerate(refused, len(cases)):
        assert status[k] == expected and out[k] is None, name
    assert taken > 300
    big = [c for c in cases if c[2] is not None and not c[0].startswith(GF.LEFT_TO_PILLOW) and min(c[2].shape) >= 16]
    ph, dh, st = ctx.gif_hash([c[1] for c in big])
    for k, c in enumerate(big):
        assert st[k] == 0 and (int(ph[k]), int(dh[k])) == O.hash_image(c[2]), c[0]
    rng = np.random.default_rng(34)
    pool = [c for c in cases if c[2] is not None and c[2].size <= 8000]
    damaged = list(TG.damaged(rng, pool, 20))
    out, status = ctx.gif_decode([d for _, d in damaged])
    decoded = 0
    for (name, data), px, st in zip(damaged, out, status):
        if st == 0:
            decoded += 1
            ref = TG.strict_pillow(data)
            assert ref is not None and ref.shape == px.shape and np.array_equal(ref, px), name
    assert decoded > 200
    # the batch hasher: .gif files on the GPU route == the Pillow route, row for row
    from kobato_eyes_amd import fastsig as K

    items = []
    for k, c in enumerate(big[:40] + [c for c in cases if c[0].startswith(GF.LEFT_TO_PILLOW)][:3]):
        p = tmp_path / f"{k:03d}.gif"
        p.write_bytes(c[1])
        items.append((500 + k, str(p)))
    rows = K.compute_signatures_mp(items, max_workers=4, chunksize=16)
    os.environ["KE_GPU_GIF"] = "0"
    try:
        assert rows == K.compute_signatures_mp(items, max_workers=4, chunksize=16) and len(rows) >= 40
    finally:
        del os.environ["KE_GPU_GIF"]


def test_tiff_unpack_matches_pillow_in_one_mixed_batch(ctx, tmp_path):
    """ke_tiff_decode: uncompressed 8-bit TIFFs -- Pillow-written gray / RGB / RGBA / palette files and hand-made directories
    (both byte orders, strips of 1 / 3 / all rows, SHORT and LONG fields, WhiteIsZero, an unspecified fourth sample, duplicate and
    unknown-type entries) -- equal to what the reference's hashes see of them through Pillow; compressed / planar / turned /
    premultiplied files and truncated strips reported per file; damaged directories never decoded differently; .tif files take
    the route in the batch hasher."""
    import test_tiff_cpu as TT

    cases = list(TF.supported(full=True)) + list(TF.handmade(full=True))
    refused = [r for r in TF.refused() if r[2] is not None]
    out, status = ctx.tiff_decode([c[1] for c in cases] + [r[1] for r in refused])
    taken = 0
    for k, (name, _, ref) in enumerate(cases):
        if ref is None or name.startswith(TF.LEFT_TO_PILLOW):
            assert status[k] != 0 and out[k] is None, name
            continue
        assert status[k] == 0, name
        assert out[k].shape == ref.shape and np.array_equal(out[k], ref), name
        taken += 1
    for k, (name, _, expected) in enumerate(refused, len(cases)):
        assert status[k] == expected and out[k] is None, name
    assert taken > 120
    big = [c for c in cases if c[2] is not None and not c[0].startswith(TF.LEFT_TO_PILLOW) and min(c[2].shape[:2]) >= 16]
    ph, dh, st = ctx.tiff_hash([c[1] for c in big])
    for k, c in enumerate(big):
        assert st[k] == 0 and (int(ph[k]), int(dh[k])) == O.hash_image(c[2]), c[0]
    rng = np.random.default_rng(36)
    pool = [c for c in cases if c[2] is not None and c[2].shape[0] <= 80]
    damaged = list(TT.damaged(rng, pool, 15))
    out, status = ctx.tiff_decode([d for _, d in damaged])
    decoded = 0
    for (name, data), px, st in zip(damaged, out, status):
        if st == 0:
            decoded += 1
            ref = TT.strict_pillow(data)
            assert ref is not None and ref.shape == px.shape and np.array_equal(ref, px), name
    assert decoded > 150
    from kobato_eyes_amd import fastsig as K

    items = []
    for k, c in enumerate(big[:40] + [c for c in cases if c[0].startswith(TF.LEFT_TO_PILLOW)][:5]):
        p = tmp_path / f"{k:03d}.{'tif' if k % 2 else 'tiff'}"
        p.write_bytes(c[1])
        items.append((700 + k, str(p)))
    rows = K.compute_signatures_mp(items, max_workers=4, chunksize=16)
    os.environ["KE_GPU_TIFF"] = "0"
    try:
        assert rows == K.compute_signatures_mp(items, max_workers=4, chunksize=16) and len(rows) >= 40
    finally:
        del os.environ["KE_GPU_TIFF"]


def test_png_damage_is_reported_not_decoded(ctx):
    """Hundreds of damaged variants of valid files in one call (bytes flipped inside the image data, truncated streams,
    lengths and filter bytes overwritten): the kernels must come back with a status for each -- and a file they do decode
    must be one Pillow decodes to the same pixels."""
    import io

    from PIL import Image

    rng = np.random.default_rng(21)
    good = [c for c in P.supported() if c[2].shape[0] >= 64][:9] + list(P.handmade())[:6] + \
        [c for c in P.interlaced() if c[2].shape[0] >= 53 and "_c2_" in c[0] or "_c0_96" in c[0] or "_c6_257" in c[0]]
    blobs, refs = [], []
    for name, data, ref in good:
        idat = data.index(b"IDAT") + 4
        for v in range(40):
            d = bytearray(data)
            kind = v % 4
            if kind == 0:                                  # one flipped bit somewhere in the image data
                pos = int(rng.integers(idat, len(d) - 16))
                d[pos] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:                                # a run of random bytes
                pos = int(rng.integers(idat, len(d) - 40))
                d[pos:pos + 24] = rng.integers(0, 256, 24, dtype=np.uint8).tobytes()
            elif kind == 2:                                # the head of the zlib stream / first block header
                pos = idat + int(rng.integers(0, 12))
                d[pos] = int(rng.integers(0, 256))
            else:                                          # data zeroed from some point on (chunk structure intact)
                pos = int(rng.integers(idat, len(d) - 16))
                d[pos:len(d) - 16] = bytes(len(d) - 16 - pos)
            blobs.append(bytes(d))
            refs.append(name)
    out, status = ctx.png_decode(blobs)
    decoded = 0
    for k, blob in enumerate(blobs):
        assert status[k] in (0, 2), (refs[k], k)
        if status[k] == 0:
            decoded += 1
            with Image.open(io.BytesIO(blob)) as im:
                assert np.array_equal(np.asarray(im), out[k]), (refs[k], k)
        else:
            assert out[k] is None
    assert decoded < len(blobs) // 4                       # a flip in unused padding bits or a stored block's data can survive
    # the context is still good for a clean batch afterwards
    clean, st = ctx.png_decode([g[1] for g in good])
    assert (st == 0).all() and all(np.array_equal(a, g[2]) for a, g in zip(clean, good))


def test_decode_batches_beyond_the_byte_limits_are_halved(ctx, tmp_path):
    """A call whose compressed bytes or decoded pixels exceed the context's limits is split in halves until each part fits;
    the hashes are those of the unsplit call -- from buffers and from files on disk."""
    cases = [c for c in J.supported()][:40]
    blobs = [c[1] for c in cases]
    ph, dh, st = ctx.jpeg_hash(blobs)
    paths = []
    for k, b in enumerate(blobs):
        p = tmp_path / f"{k}.jpg"
        p.write_bytes(b)
        paths.append(str(p))
    saved = ctx.pack_limit, ctx.decode_limit
    try:
        ctx.release_decode_buffers()
        ctx.pack_limit = max(len(b) for b in blobs) * 3
        for got in (ctx.jpeg_hash(blobs), ctx.hash_files(paths)):
            assert np.array_equal(got[0], ph) and np.array_equal(got[1], dh) and np.array_equal(got[2], st)
        ctx.pack_limit = saved[0]
        ctx.decode_limit = 200_000
        for got in (ctx.jpeg_hash(blobs), ctx.hash_files(paths)):
            assert np.array_equal(got[0], ph) and np.array_equal(got[1], dh) and np.array_equal(got[2], st)
        pixels, st2 = ctx.jpeg_decode(blobs)
        assert np.array_equal(st2, st) and all((a is None) == (s != 0) for a, s in zip(pixels, st))
    finally:
        ctx.pack_limit, ctx.decode_limit = saved
        ctx.release_decode_buffers()


def test_jpeg_damage_is_survived(ctx, monkeypatch):
    """Hundreds of damaged variants of sequential and progressive files in one call (flipped bits and random runs inside the
    entropy-coded data, tables and scan headers overwritten, truncations): the kernels come back with a status for each and
    the context decodes a clean batch afterwards; the damaged files it does take (a flipped bit often leaves a valid stream) carry
    Pillow's pixels."""
    import io

    from PIL import Image

    rng = np.random.default_rng(33)
    good = [c for c in J.supported() if c[2].shape[0] >= 64 and c[2].shape[1] >= 64][:24]
    assert any("progressive" in c[0] for c in good) and any("progressive" not in c[0] for c in good)
    blobs = []
    for name, data, _ in good:
        sos = data.index(b"\xff\xda")
        for v in range(36):
            d = bytearray(data)
            kind = v % 4
            if kind == 0:
                pos = int(rng.integers(sos, len(d) - 2))
                d[pos] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                pos = int(rng.integers(sos, max(sos + 1, len(d) - 40)))
                d[pos:pos + 24] = rng.integers(0, 256, 24, dtype=np.uint8).tobytes()
            elif kind == 2:                                # anywhere in the file, headers and tables included
                pos = int(rng.integers(2, len(d) - 2))
                d[pos] = int(rng.integers(0, 256))
            else:
                d = d[: int(rng.integers(sos, len(d)))]
            blobs.append(bytes(d))
    out, status = ctx.jpeg_decode(blobs)
    assert set(np.unique(status).tolist()) <= {0, 1, 2}
    assert all((a is None) == (s != 0) for a, s in zip(out, status))
    from PIL import ImageFile

    monkeypatch.setattr(ImageFile, "LOAD_TRUNCATED_IMAGES", False)      # safe_load_image (as the reference's) leaves it set
    same = 0
    for blob, a in zip(blobs, out):                  # a damaged file the decoder takes has the pixels Pillow makes of it
        if a is None:
            continue
        try:
            im = Image.open(io.BytesIO(blob))
            im.load()
        except Exception:                            # Pillow gives up on some the decoder reads to the end
            continue
        ref = np.asarray(im)
        assert ref.shape == a.shape and np.array_equal(ref, a)
        same += 1
    assert same > 100
    clean, st = ctx.jpeg_decode([g[1] for g in good])
    assert (st == 0).all() and all(np.array_equal(a, g[2]) for a, g in zip(clean, good))


def test_random_files_in_one_batch(ctx):
    """300 random JPEGs from Pillow's writer and 250 progressive ones with random scan scripts (what other encoders produce),
    300 random PNGs from Pillow's writer and 400 hand-made ones (interlacing, 16 bits, sub-byte gray, short
    palettes, any filter in any row) -- every kind the decoders take, random sizes, qualities, scripts, compression levels -- in
    one call each: Pillow's pixels, file by file."""
    for cases, decode in ((list(J.random_cases(300, 41)) + list(J.scripted(250, 44)), ctx.jpeg_decode),
                          (list(P.random_cases(300, 42)) + list(P.random_handmade(400, 43)), ctx.png_decode)):
        out, status = decode([c[1] for c in cases])
        for k, (name, _, ref) in enumerate(cases):
            assert status[k] == 0, name
            assert out[k].shape == ref.shape and np.array_equal(out[k], ref), name


def test_files_with_blocks_beyond_the_16_bit_bound_are_handed_back(ctx):
    """ke_idct_islow's bound (csrc/ke_jpeg_core.h): files whose quantisation tables are overwritten with larger and larger
    steps.  Below the bound the pixels are Pillow's; beyond it libjpeg's C arithmetic and Pillow's SIMD build differ, and the
    decoder answers KE_JPEG_UNSUPPORTED (the CPU restatement the same, file by file) -- never other pixels than Pillow's."""
    import io

    from PIL import Image

    import test_jpeg_cpu as T

    L = T._lib()
    good = [c for c in J.supported() if c[2].shape[0] >= 64 and c[2].shape[1] >= 64 and "q100" not in c[0]][:12]
    good += [c for c in J.supported() if "q100" in c[0] and c[2].shape[0] >= 64][:4]
    assert any("progressive" in c[0] for c in good) and any("gray" in c[0] for c in good)
    steps = (255, 64, 33, 16, 8, 4, 2, 1)
    blobs = [J.with_quantisation_tables(data, v) for _, data, _ in good for v in steps]
    out, status = ctx.jpeg_decode(blobs)
    taken = {v: 0 for v in steps}
    for k, blob in enumerate(blobs):
        st, ref = T._decode(L, blob)
        assert status[k] == st and st in (0, 1), k
        if st == 0:
            assert np.array_equal(out[k], ref) and np.array_equal(out[k], np.asarray(Image.open(io.BytesIO(blob)))), k
            taken[steps[k % len(steps)]] += 1
    assert taken[255] == 0 and taken[64] == 0 and taken[1] == len(good) and 0 < taken[16] < len(good)
    clean, st = ctx.jpeg_decode([g[1] for g in good])                    # the status is per file and per call
    assert (st == 0).all() and all(np.array_equal(a, g[2]) for a, g in zip(clean, good))


def test_damaged_files_the_kernels_take_are_decoded_as_pillow_does(ctx):
    """tests/fuzz_jpeg_damage.py through the kernels: two batches of 1 600 damaged files (headers / entropy data); every file the
    decoder takes has Pillow's pixels, and the statuses are those of the CPU build of the same headers."""
    import fuzz_jpeg_damage as F

    seen = {}

    def decode(blobs):
        out, status = ctx.jpeg_decode(blobs)
        _, cpu_status = F.cpu_decoder()(blobs)
        seen[len(seen)] = (np.asarray(status).tolist(), list(cpu_status))
        return out, status

    cases, taken, wrong = F.check(decode, 40, 11)
    assert not wrong, wrong[:5]
    assert cases == 4800 and taken > 800
    assert all(gpu == cpu for gpu, cpu in seen.values())
    # PNG files the same way (chunk and zlib checksums refuse nearly all of them)
    cases, taken, wrong = F.check(ctx.png_decode, 40, 12, fmt="png")
    assert not wrong, wrong[:5]
    assert cases == 3200 and taken >= 1


def test_decompression_bombs_are_left_to_pillow(ctx, tmp_path, monkeypatch):
    """`Image.open` raises DecompressionBombError beyond twice Image.MAX_IMAGE_PIXELS, so the batch hasher drops such a file;
    the GPU decoders hand them back (status 1, nothing written) and the seam's rows are the Pillow route's.  The cap is
    lowered to 2 000 pixels here: 64 x 64 files are 'bombs', 17 x 33 ones are not."""
    from PIL import Image

    import kobato_eyes_amd as K

    jpegs, pngs = list(J.supported()), list(P.supported())            # (made with Pillow: before the cap is lowered)
    monkeypatch.setattr(Image, "MAX_IMAGE_PIXELS", 2000)
    for cases, decode, suffix in ((jpegs, ctx.jpeg_decode, "jpg"), (pngs, ctx.png_decode, "png")):
        small = [c for c in cases if c[2].shape[0] * c[2].shape[1] <= 4000][:12]
        large = [c for c in cases if c[2].shape[0] * c[2].shape[1] > 4000][:12]
        assert len(small) >= 4 and len(large) >= 4
        mixed = [c for pair in zip(small, large) for c in pair]
        out, status = decode([c[1] for c in mixed])
        for k, (name, _, ref) in enumerate(mixed):
            if k % 2 == 0:
                assert status[k] == 0 and np.array_equal(out[k], ref), name
            else:
                assert status[k] != 0 and out[k] is None, name
        items = []
        for k, (name, data, _) in enumerate(mixed):
            path = tmp_path / f"{suffix}{k:02d}.{suffix}"
            path.write_bytes(data)
            items.append((k + 1, str(path)))
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore", Image.DecompressionBombWarning)
            monkeypatch.setenv("KE_DECODE_PROCESSES", "0")         # the lowered cap is this process's: Pillow on threads
            rows = K.compute_signatures_mp(items, max_workers=2, chunksize=8)
            monkeypatch.setenv("KE_GPU_JPEG", "0")
            monkeypatch.setenv("KE_GPU_PNG", "0")
            assert K.compute_signatures_mp(items, max_workers=2, chunksize=8) == rows
            monkeypatch.delenv("KE_GPU_JPEG")
            monkeypatch.delenv("KE_GPU_PNG")
        assert [r[0] for r in rows] == [k + 1 for k in range(len(mixed)) if k % 2 == 0]


def test_png_batch_with_more_than_4_gb_of_compressed_data(ctx):
    """One launch over 12 288 copies of a 384 KB PNG (4.7 GB of zlib streams: their offsets are 64-bit): every file decodes --
    each stream's Adler-32 is checked on the device -- and hashes like Pillow's pixels of the file."""
    import io

    from PIL import Image

    rng = np.random.default_rng(77)
    arr = rng.integers(0, 256, (512, 512, 3), dtype=np.uint8)
    arr[:, 100:400] = arr[:, 99:100]                                  # some long copies among the literals
    b = io.BytesIO()
    Image.fromarray(arr).save(b, "PNG", compress_level=1)
    blob = b.getvalue()
    n = (4_700_000_000 + len(blob) - 1) // len(blob)
    ph, dh, st = ctx.jpeg_hash([blob] * n, kind="png")
    assert (st == 0).all()
    want = O.hash_image(np.asarray(Image.open(io.BytesIO(blob))))
    assert (ph == np.uint64(want[0])).all() and (dh == np.uint64(want[1])).all()
    ctx.release_decode_buffers()
