"""JPEG files for the decoder tests, made with the installed Pillow (libjpeg-turbo): every variant the GPU decoder takes
(sizes around the MCU edges, 4:4:4 / 4:2:2 / 4:2:0, grayscale, qualities, optimised Huffman tables, restart markers) and the
ones it must hand back to Pillow (progressive, CMYK, RGB-coded, truncated)."""
from __future__ import annotations

import io

import numpy as np
from PIL import Image, ImageFile

ImageFile.MAXBLOCK = 1 << 26          # Pillow's encoder buffer: noise at quality 100 does not fit the default


def _image(rng, w, h, kind):
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == 0:
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if kind == 1:
        return np.stack([xx * 255 // max(w - 1, 1), yy * 255 // max(h - 1, 1), (xx + yy) * 255 // max(w + h - 2, 1)], -1).astype(np.uint8)
    base = rng.integers(0, 256, (h // 16 + 1, w // 16 + 1, 3), dtype=np.uint8)
    a = np.repeat(np.repeat(base, 16, 0), 16, 1)[:h, :w]
    return np.clip(a.astype(np.int16) + rng.integers(-6, 7, a.shape), 0, 255).astype(np.uint8)


def _save(arr, **kw) -> bytes:
    b = io.BytesIO()
    Image.fromarray(arr).save(b, "JPEG", **kw)
    return b.getvalue()


def supported(full: bool = False):
    """Yields (name, file bytes, pixels as Pillow decodes them)."""
    rng = np.random.default_rng(1)
    sizes = [(1, 1), (2, 2), (1, 9), (9, 1), (5, 3), (8, 8), (9, 9), (17, 33), (64, 64), (101, 77), (255, 257), (512, 512), (640, 480)]
    if full:
        sizes += [(1000, 31), (33, 1000), (1024, 768), (3, 3), (4, 4), (16, 16), (15, 17)]
    for (w, h) in sizes:
        for kind in (0, 1, 2):
            a = _image(rng, w, h, kind)
            for sub in (0, 1, 2):
                for q, extra in ((30, {}), (75, {"optimize": True}), (95, {}), (100, {})) if full or kind == 2 else ((85, {}),):
                    data = _save(a, quality=q, subsampling=sub, **extra)
                    yield f"{w}x{h}_k{kind}_s{sub}_q{q}", data, np.asarray(Image.open(io.BytesIO(data)))
            data = _save(a[:, :, 1], quality=80)
            yield f"{w}x{h}_k{kind}_gray", data, np.asarray(Image.open(io.BytesIO(data)))
            # progressive files (libjpeg's default script: spectral selection + successive approximation, optimised tables)
            for sub in (0, 1, 2):
                for q in (30, 85, 100) if full or kind == 2 else (85,):
                    data = _save(a, quality=q, subsampling=sub, progressive=True)
                    yield f"{w}x{h}_k{kind}_s{sub}_q{q}_progressive", data, np.asarray(Image.open(io.BytesIO(data)))
            data = _save(a[:, :, 1], quality=80, progressive=True)
            yield f"{w}x{h}_k{kind}_gray_progressive", data, np.asarray(Image.open(io.BytesIO(data)))
    a = _image(rng, 200, 120, 2)
    for kw in ({"restart_marker_blocks": 1}, {"restart_marker_blocks": 7}, {"restart_marker_rows": 1}, {"restart_marker_rows": 3}):
        for sub in (0, 2):
            for prog in (False, True):
                try:
                    data = _save(a, quality=85, subsampling=sub, progressive=prog, **kw)
                except TypeError:                      # a Pillow without the restart options
                    continue
                yield f"restart_{list(kw)[0]}_{list(kw.values())[0]}_s{sub}_p{int(prog)}", data, np.asarray(Image.open(io.BytesIO(data)))
    # 4:4:0 (what jpegtran makes of a 4:2:2 file it turns): a 4:2:2 file with the luma sampling byte of its frame header set
    # to 1x2 -- the scan is the same bit stream (two luma blocks, Cb, Cr per MCU), laid out as two block rows instead of two
    # columns; the picture is scrambled, the arithmetic (block order, h1v2 triangle upsampling) is what is compared
    # (sizes whose MCU count is the same either way: the stream must neither run dry nor have blocks left over)
    for (w, h) in ((128, 128), (61, 57), (200, 195), (16, 16), (33, 40), (7, 5), (2, 2)):
        assert -(-w // 16) * -(-h // 8) == -(-w // 8) * -(-h // 16)
        for prog in (False, True):
            b = _image(rng, w, h, 2)
            data = with_luma_sampling(_save(b, quality=88, subsampling=1, progressive=prog), 0x12)
            yield f"{w}x{h}_440_p{int(prog)}", data, np.asarray(Image.open(io.BytesIO(data)))
    # a DQT between the scans of a progressive file is legal; libjpeg keeps the table that was in effect at a component's first
    # scan (jdinput.c latch_quant_tables), so the redefinition changes nothing for components already seen
    for sub in (0, 2):
        data = with_dqt_between_scans(_save(a, quality=85, subsampling=sub, progressive=True), 1)
        yield f"progressive_dqt_between_scans_s{sub}", data, np.asarray(Image.open(io.BytesIO(data)))


def with_luma_sampling(data: bytes, factors: int) -> bytes:
    """The file with the sampling byte of its first frame component set to `factors` (0x12 = 1 x 2)."""
    d = bytearray(data)
    sof = max(data.find(b"\xff\xc0"), data.find(b"\xff\xc2"))
    d[sof + 11] = factors
    return bytes(d)


def with_dqt_between_scans(data: bytes, value: int) -> bytes:
    """A progressive file with a DQT segment (tables 0 and 1, every step = `value`) put in front of its second scan."""
    first = data.index(b"\xff\xda")
    pos = first + 2 + int.from_bytes(data[first + 2:first + 4], "big")
    while not (data[pos] == 0xFF and data[pos + 1] not in (0x00, 0xFF) and not 0xD0 <= data[pos + 1] <= 0xD7):
        pos += 1                                          # end of the first scan's entropy data
    dqt = b"\xff\xdb" + (2 + 2 * 65).to_bytes(2, "big") + b"\x00" + bytes([value]) * 64 + b"\x01" + bytes([value]) * 64
    return data[:pos] + dqt + data[pos:]


def with_quantisation_tables(data: bytes, value: int) -> bytes:
    """The file with every entry of its (8-bit) quantisation tables set to `value`."""
    d = bytearray(data)
    pos = 2
    while d[pos + 1] != 0xDA:
        length = int.from_bytes(d[pos + 2:pos + 4], "big")
        if d[pos + 1] == 0xDB:
            q = pos + 4
            while q < pos + 2 + length:
                assert d[q] >> 4 == 0
                d[q + 1:q + 65] = bytes([value]) * 64
                q += 65
        pos += 2 + length
    return bytes(d)


def refused():
    """Yields (name, file bytes, expected status): 1 = left to Pillow, 2 = damaged."""
    rng = np.random.default_rng(2)
    a = _image(rng, 96, 64, 2)
    prog = _save(a, quality=85, progressive=True)
    yield "progressive_truncated", prog[: len(prog) * 2 // 3], 2
    cut = prog.rindex(b"\xff\xda")                       # the last scan (a refinement) dropped: libjpeg would smooth the blocks
    yield "progressive_last_scan_missing", prog[:cut] + b"\xff\xd9", 1
    yield "cmyk", (lambda b: (Image.fromarray(a).convert("CMYK").save(b, "JPEG"), b.getvalue())[1])(io.BytesIO()), 1
    try:
        yield "rgb_coded", _save(a, quality=90, keep_rgb=True), 1
    except TypeError:
        pass
    good = _save(a, quality=85)
    yield "truncated", good[: len(good) * 2 // 3], 2
    yield "not_a_jpeg", b"\\x89PNG\\r\\n\\x1a\\n" + bytes(64), 2
    yield "empty", b"", 2


def random_cases(n: int, seed: int = 0):
    """n random Huffman JPEGs as Pillow writes them (sizes up to 200 x 150, gray / 4:4:4 / 4:2:2 / 4:2:0, any quality,
    sequential or progressive, optimised or not): (name, file bytes, Pillow's pixels)."""
    rng = np.random.default_rng(seed)
    for k in range(n):
        w, h = int(rng.integers(1, 201)), int(rng.integers(1, 151))
        a = _image(rng, w, h, int(rng.integers(0, 3)))
        gray = bool(rng.integers(0, 4) == 0)
        kw = {"quality": int(rng.integers(1, 101)), "progressive": bool(rng.integers(0, 2)), "optimize": bool(rng.integers(0, 2))}
        if not gray:
            kw["subsampling"] = int(rng.integers(0, 3))
        data = _save(a[:, :, 1] if gray else a, **kw)
        yield f"random{k}_{w}x{h}_{kw}", data, np.asarray(Image.open(io.BytesIO(data)))


def scripted(n: int, seed: int = 0):
    """n progressive files with random legal scan scripts (tests/_jpeg_prog_encoder.py: what Pillow's writer never produces and
    other encoders -- mozjpeg -- do): (name, file bytes, pixels as Pillow decodes them)."""
    import _jpeg_prog_encoder as E

    rng = np.random.default_rng(seed)
    for k in range(n):
        w, h = int(rng.integers(1, 90)), int(rng.integers(1, 70))
        gray = bool(rng.integers(0, 5) == 0)
        sampling = ("444", "422", "420", "440")[int(rng.integers(0, 4))]
        data, script = E.random_file(rng, w, h, sampling, gray)
        with Image.open(io.BytesIO(data)) as im:
            ref = np.asarray(im)
        yield f"scripted{k}_{'gray' if gray else sampling}_{w}x{h}_{len(script)}scans", data, ref
