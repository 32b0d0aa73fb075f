"""PNG files for the decoder tests, made with the installed Pillow: what the GPU decoder takes (8-bit L / RGB / RGBA, every
compression level incl. stored blocks and optimised encoding, sizes from 1x1; palette, sub-byte, gray+alpha, Adam7 and 16-bit files)
and what it hands back (truncated, damaged, beyond its limits)."""
from __future__ import annotations

import io
import struct
import zlib

import numpy as np
from PIL import Image


def _save(arr, **kw) -> bytes:
    b = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(arr)).save(b, "PNG", **kw)
    return b.getvalue()


def supported(full: bool = False):
    """Yields (name, file bytes, pixels as Pillow decodes them)."""
    rng = np.random.default_rng(3)
    sizes = [(1, 1), (2, 3), (7, 5), (64, 64), (101, 77), (300, 200), (512, 512)] + ([(1000, 31), (33, 1000), (1024, 768)] if full else [])
    for (w, h) in sizes:
        for kind in range(4):
            if kind == 0:
                a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
            elif kind == 1:
                yy, xx = np.mgrid[0:h, 0:w]
                a = np.stack([xx * 255 // max(w - 1, 1), yy * 255 // max(h - 1, 1), (xx + yy) % 256, (xx * yy) % 256], -1).astype(np.uint8)
            elif kind == 2:
                base = rng.integers(0, 256, (h // 16 + 1, w // 16 + 1, 4), dtype=np.uint8)
                a = np.repeat(np.repeat(base, 16, 0), 16, 1)[:h, :w]
            else:
                a = np.full((h, w, 4), 200, np.uint8)
            for mode, arr in (("RGB", a[:, :, :3]), ("RGBA", a), ("L", a[:, :, 0])):
                for kw in ({}, {"compress_level": 0}, {"compress_level": 9}, {"optimize": True}) if full or kind == 2 else ({},):
                    data = _save(arr, **kw)
                    yield f"{w}x{h}_k{kind}_{mode}_{kw}", data, np.asarray(Image.open(io.BytesIO(data)))


def mapped(full: bool = False):
    """Yields (name, file bytes, luma as the reference's hashes see it: Image.open(...).convert("L")): palette files of every
    depth Pillow writes (2 / 4 / 16 / 256 colours -> 1 / 2 / 4 / 8 bits), with and without transparency, mode "1", and
    hand-packed 2- and 4-bit grayscale."""
    rng = np.random.default_rng(8)
    sizes = [(1, 1), (5, 3), (13, 9), (64, 64), (101, 77), (300, 200)] + ([(1000, 31), (33, 1000), (517, 389)] if full else [])
    for (w, h) in sizes:
        for ncol in (2, 4, 16, 256):
            pal = rng.integers(0, 256, ncol * 3, dtype=np.uint8)
            kinds = (0, 1) if full or (w, h) == (101, 77) else (0,)
            for kind in kinds:
                idx = rng.integers(0, ncol, (h, w), dtype=np.uint8) if kind == 0 else \
                    np.repeat(np.repeat(rng.integers(0, ncol, (h // 8 + 1, w // 8 + 1), dtype=np.uint8), 8, 0), 8, 1)[:h, :w]
                im = Image.fromarray(np.ascontiguousarray(idx), "P")
                im.putpalette(pal.tolist())
                for kw in ({}, {"transparency": 1}):
                    b = io.BytesIO()
                    im.save(b, "PNG", **kw)
                    data = b.getvalue()
                    yield f"P{ncol}_{w}x{h}_k{kind}_{kw}", data, np.asarray(Image.open(io.BytesIO(data)).convert("L"))
        b = io.BytesIO()
        Image.fromarray(rng.integers(0, 2, (h, w), dtype=np.uint8) * 255).convert("1").save(b, "PNG")
        data = b.getvalue()
        yield f"bilevel_{w}x{h}", data, np.asarray(Image.open(io.BytesIO(data)).convert("L"))
        la = np.stack([rng.integers(0, 256, (h, w), dtype=np.uint8), rng.integers(0, 256, (h, w), dtype=np.uint8)], -1)
        if (w, h) == (101, 77):
            la[:, :, 0] = np.repeat(np.repeat(rng.integers(0, 256, (h // 8 + 1, w // 8 + 1), dtype=np.uint8), 8, 0), 8, 1)[:h, :w]
        for kw in ({}, {"compress_level": 9}) if full or (w, h) == (64, 64) else ({},):
            b = io.BytesIO()                                         # gray + alpha: the reference hashes convert("L") = the gray band
            Image.fromarray(la, "LA").save(b, "PNG", **kw)
            data = b.getvalue()
            yield f"LA_{w}x{h}_{kw}", data, np.asarray(Image.open(io.BytesIO(data)).convert("L"))
        for depth in (2, 4):                                         # grayscale below 8 bits: rows packed by hand
            per = 8 // depth
            vals = rng.integers(0, 1 << depth, (h, w), dtype=np.uint8)
            padded = np.zeros((h, (w + per - 1) // per * per), np.uint8)
            padded[:, :w] = vals
            packed = np.zeros((h, padded.shape[1] // per), np.uint8)
            for q in range(per):
                packed |= padded[:, q::per] << (8 - depth * (q + 1))
            rows = np.concatenate([rng.integers(0, 5, (h, 1), dtype=np.uint8), packed], 1)
            # the filter byte says how the row is stored; undo nothing here: store rows as filter 0 to keep the values as packed
            rows[:, 0] = 0
            data = _container(rows.tobytes(), w, h, 0, 6, 0, 1 << 30, depth=depth)
            yield f"gray{depth}_{w}x{h}", data, np.asarray(Image.open(io.BytesIO(data)).convert("L"))


def _container(raw: bytes, w: int, h: int, ctype: int, level: int, strategy: int, chunk: int, depth: int = 8) -> bytes:
    """A PNG around filtered scanlines given as they are, with control over what Pillow's writer never varies: the zlib
    strategy and the size of the IDAT chunks."""
    co = zlib.compressobj(level, zlib.DEFLATED, 15, 9, strategy)
    z = co.compress(raw) + co.flush()

    def ch(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))

    out = b"\x89PNG\r\n\x1a\n" + ch(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
    for o in range(0, len(z), chunk):
        out += ch(b"IDAT", z[o:o + chunk])
    return out + ch(b"IEND", b"")


def handmade(full: bool = False):
    """Yields (name, file bytes, pixels as Pillow decodes them): streams Pillow's own writer does not produce -- code lengths
    up to 15 bits (a geometric byte histogram), Huffman-only / RLE / fixed-code / stored blocks, IDAT data cut into chunks
    of a few bytes, every filter type on random rows."""
    rng = np.random.default_rng(5)
    sizes = [(257, 129), (640, 480)] if full else [(257, 129)]
    for (w, h) in sizes:
        for ctype, ch_ in ((0, 1), (2, 3), (6, 4)):
            vals = np.minimum(rng.geometric(0.35, size=(h, w * ch_)) - 1, 255).astype(np.uint8)
            rows = np.concatenate([rng.integers(0, 5, (h, 1), dtype=np.uint8), vals], 1)
            for level, strategy, chunk in ((6, 0, 1 << 30), (9, zlib.Z_FILTERED, 4096), (1, zlib.Z_HUFFMAN_ONLY, 100), (6, zlib.Z_RLE, 7),
                                           (6, zlib.Z_FIXED, 1 << 30), (0, 0, 5000)):
                data = _container(rows.tobytes(), w, h, ctype, level, strategy, chunk)
                yield f"handmade_{w}x{h}_c{ctype}_l{level}_s{strategy}_k{chunk}", data, np.asarray(Image.open(io.BytesIO(data)))
    # copies of every short distance, overlapping their own output, in long dependent chains: row y repeats a random pattern
    # of (y % 24) + 1 bytes; then rows that repeat earlier rows at distances of a few hundred to a few thousand bytes
    for (w, h, ctype, ch_) in ((1200, 96, 0, 1), (401, 96, 2, 3), (16384, 3, 6, 4), (3, 4000, 2, 3)):
        rb = w * ch_
        body = np.empty((h, rb), np.uint8)
        for y in range(h):
            pat = rng.integers(0, 256, (y % 24) + 1, dtype=np.uint8)
            body[y] = np.resize(pat, rb)
        body[h // 2:] = body[: h - h // 2]
        rows = np.concatenate([np.zeros((h, 1), np.uint8), body], 1)
        for level in (6, 9):
            data = _container(rows.tobytes(), w, h, ctype, level, 0, 1 << 30)
            yield f"periodic_{w}x{h}_c{ctype}_l{level}", data, np.asarray(Image.open(io.BytesIO(data)))


_ADAM7 = ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2))


def _filtered(rows: np.ndarray, unit: int, types) -> bytes:
    """The forward PNG filters (specification, "Filter algorithms") on rows of packed bytes, one type per row."""
    h, rb = rows.shape
    out = bytearray()
    prev = np.zeros(rb, np.int32)
    for y in range(h):
        cur = rows[y].astype(np.int32)
        a = np.zeros(rb, np.int32)
        c = np.zeros(rb, np.int32)
        a[unit:] = cur[:rb - unit] if rb > unit else 0
        c[unit:] = prev[:rb - unit] if rb > unit else 0
        t = int(types[y])
        if t == 0:
            pred = 0
        elif t == 1:
            pred = a
        elif t == 2:
            pred = prev
        elif t == 3:
            pred = (a + prev) >> 1
        else:
            pa, pb, pc = np.abs(prev - c), np.abs(a - c), np.abs(a + prev - 2 * c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        out += bytes([t]) + ((cur - pred) & 255).astype(np.uint8).tobytes()
        prev = cur
    return bytes(out)


def _adam7_stream(samples: np.ndarray, depth: int, rng, passes=_ADAM7) -> bytes:
    """samples: H x W x C (8-bit) or H x W x 1 values below 2**depth -> the seven passes' filtered rows, random filter types
    (passes=((0, 0, 1, 1),): the rows of a file without interlacing)."""
    h, w, ch = samples.shape
    out = b""
    for (x0, y0, dx, dy) in passes:
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        if depth == 8:
            rows, unit = sub.reshape(sub.shape[0], -1), ch
        else:
            bits = ((sub[:, :, 0, None] >> np.arange(depth - 1, -1, -1)) & 1).reshape(sub.shape[0], -1).astype(np.uint8)
            rows, unit = np.packbits(bits, axis=1), 1
        out += _filtered(np.ascontiguousarray(rows), unit, rng.integers(0, 5, sub.shape[0]))
    return out


def _container2(raw: bytes, w: int, h: int, ctype: int, depth: int, interlace: int, plte: bytes = None, trns: bytes = None,
                level: int = 6, chunk: int = 1 << 30) -> bytes:
    def ch(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))

    z = zlib.compress(raw, level)
    out = b"\x89PNG\r\n\x1a\n" + ch(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace))
    if plte is not None:
        out += ch(b"PLTE", plte)
    if trns is not None:
        out += ch(b"tRNS", trns)
    for o in range(0, len(z), chunk):
        out += ch(b"IDAT", z[o:o + chunk])
    return out + ch(b"IEND", b"")


def interlaced(full: bool = False):
    """Yields (name, file bytes, what the reference's hashes see): Adam7 files (no writer in Pillow -- built here, every filter
    type in every pass) of every kind the decoder takes: 8-bit L / RGB / RGBA / LA, palette and grayscale of 1 / 2 / 4 / 8 bits;
    sizes where some passes are empty (1x1 .. 4x4), where rows end inside a byte, and ordinary ones."""
    rng = np.random.default_rng(12)
    sizes = [(1, 1), (2, 1), (1, 2), (3, 2), (4, 4), (5, 5), (8, 8), (9, 17), (75, 53), (96, 80), (257, 129)] + ([(640, 333), (33, 1000), (1030, 40)] if full else [])
    for (w, h) in sizes:
        smooth = (w * h) > 5000
        for ctype, chans in ((0, 1), (2, 3), (6, 4), (4, 2)):
            if smooth:
                yy, xx = np.mgrid[0:h, 0:w]
                a = np.stack([(xx * 2 + yy) % 256, (yy * 3) % 256, (xx + yy // 2) % 256, (xx * yy // 7) % 256], -1).astype(np.uint8)[:, :, :chans]
                a = (a + rng.integers(0, 4, a.shape)).astype(np.uint8)
            else:
                a = rng.integers(0, 256, (h, w, chans), dtype=np.uint8)
            data = _container2(_adam7_stream(a, 8, rng), w, h, ctype, 8, 1, chunk=(1 << 30) if ctype != 2 else 37)
            with Image.open(io.BytesIO(data)) as im:
                ref = np.asarray(im.convert("L") if im.mode == "LA" else im)
            yield f"adam7_c{ctype}_{w}x{h}", data, ref
        for depth in (1, 2, 4, 8):
            vals = rng.integers(0, 1 << depth, (h, w, 1), dtype=np.uint8)
            for ctype in ((0, 3) if depth < 8 else (3,)):
                plte = rng.integers(0, 256, 3 * (1 << depth), dtype=np.uint8).tobytes() if ctype == 3 else None
                trns = bytes([0, 128]) if ctype == 3 and w % 2 else None
                data = _container2(_adam7_stream(vals, depth, rng), w, h, ctype, depth, 1, plte=plte, trns=trns)
                with Image.open(io.BytesIO(data)) as im:
                    ref = np.asarray(im.convert("L"))
                yield f"adam7_c{ctype}_d{depth}_{w}x{h}", data, ref


def wide(full: bool = False):
    """Yields (name, file bytes, what Pillow opens the file to -- for grayscale what convert("L") makes of its mode "I;16"):
    16-bit files of every colour type, with and without interlacing, every filter type (unit: 2 bytes per sample), samples
    with independent high and low bytes."""
    rng = np.random.default_rng(15)
    sizes = [(1, 1), (2, 3), (5, 4), (9, 17), (64, 48), (131, 67)] + ([(640, 333), (3, 900), (1030, 40)] if full else [])
    for (w, h) in sizes:
        for ctype, chans in ((0, 1), (2, 3), (4, 2), (6, 4)):
            a = rng.integers(0, 256, (h, w, 2 * chans), dtype=np.uint8)        # big-endian samples as bytes
            if ctype == 0:
                a[:, :, 0] = np.where(rng.random((h, w)) < 0.7, 0, a[:, :, 0])  # mostly below 256: convert("L") clips the rest
            if w * h > 3000:                                                    # something deflate finds matches in
                a[h // 4: h // 2] = a[h // 4]
            for lace in (0, 1):
                if lace:
                    raw = _adam7_stream(a, 8, rng)
                else:
                    raw = _filtered(np.ascontiguousarray(a.reshape(h, -1)), 2 * chans, rng.integers(0, 5, h))
                data = _container2(raw, w, h, ctype, 16, lace, chunk=(1 << 30) if ctype != 6 else 61)
                with Image.open(io.BytesIO(data)) as im:
                    ref = np.asarray(im.convert("L") if im.mode == "I;16" else im)
                yield f"wide_c{ctype}_{w}x{h}_i{lace}", data, ref


def animated(full: bool = False):
    """Yields (name, file bytes, what the reference's hashes see: frame 0 as Image.open shows it) for animated PNG files written
    by Pillow (the first frame is the IDAT image, or a separate default image is), and hand-edited ones: the file cut right
    behind frame 0's data, and what the decoder leaves to Pillow (expected = None: a first frame smaller than the image,
    a frame count of zero, a second acTL)."""
    rng = np.random.default_rng(21)
    sizes = [(7, 5), (64, 48), (101, 77)] + ([(300, 200)] if full else [])
    for (w, h) in sizes:
        frames = [rng.integers(0, 256, (h, w, 4), dtype=np.uint8) for _ in range(3)]
        frames[1][: h // 2] = frames[0][: h // 2]
        for mode in ("RGB", "RGBA", "L", "P"):
            ims = [Image.fromarray(f).convert("RGB").quantize(64) if mode == "P" else Image.fromarray(f).convert(mode) for f in frames]
            for default_image in (False, True):
                b = io.BytesIO()
                ims[0].save(b, "PNG", save_all=True, append_images=ims[1:], default_image=default_image, duration=80, loop=0)
                data = b.getvalue()
                with Image.open(io.BytesIO(data)) as im:
                    assert im.n_frames > 1
                    ref = np.asarray(im.convert("L") if im.mode in ("P", "1") else im)
                yield f"apng_{mode}_{w}x{h}_default{int(default_image)}", data, ref
                if mode == "RGB":
                    # nothing behind frame 0's data: Pillow stops reading at the next frame anyway
                    cut = data.index(b"fcTL", data.index(b"IDAT")) - 4
                    yield f"apng_cut_behind_frame0_{w}x{h}_default{int(default_image)}", data[:cut], ref
                    yield f"apng_cut_inside_next_fctl_{w}x{h}_default{int(default_image)}", data[:cut + 14], ref
    # left to Pillow
    b = io.BytesIO()
    ims = [Image.fromarray(rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)) for _ in range(2)]
    ims[0].save(b, "PNG", save_all=True, append_images=ims[1:])
    data = b.getvalue()

    def rechunk(at, body):
        n = struct.unpack(">I", data[at - 4:at])[0]
        t = data[at:at + 4]
        return data[:at + 4] + body + struct.pack(">I", zlib.crc32(t + body)) + data[at + 8 + n:]

    a = data.index(b"acTL")
    f = data.index(b"fcTL")
    fc = data[f + 4:f + 30]
    yield "apng_zero_frames", rechunk(a, struct.pack(">II", 0, 0)), None
    yield "apng_first_frame_smaller", rechunk(f, fc[:4] + struct.pack(">IIII", 30, 40, 0, 0) + fc[20:]), None
    yield "apng_first_frame_offset", rechunk(f, fc[:4] + struct.pack(">IIII", 49, 40, 1, 0) + fc[20:]), None
    yield "apng_sequence_from_1", rechunk(f, struct.pack(">I", 1) + fc[4:]), None
    actl = data[a - 4:a + 16]
    yield "apng_two_actl", data[:a - 4] + actl + actl + data[a + 16:], None


def refused():
    """Yields (name, file bytes, expected status): 1 = left to Pillow, 2 = damaged."""
    rng = np.random.default_rng(4)
    a = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    b = io.BytesIO()
    Image.fromarray(a).convert("P").save(b, "PNG")
    pal = b.getvalue()
    cut = pal.index(b"PLTE") - 4
    n = struct.unpack(">I", pal[cut:cut + 4])[0]
    yield "palette_missing", pal[:cut] + pal[cut + 12 + n:], 2
    rng.integers(0, 65535, (20, 30))                                               # (keeps the stream of random numbers of earlier rounds)
    yield "16bit_beyond_the_row_limit", _container2(b"".join(b"\x00" + bytes(2 * 9000) for _ in range(2)), 9000, 2, 0, 16, 0), 1
    yield "16bit_palette", _container2(b"\x00" + bytes(8), 4, 1, 3, 16, 0, plte=bytes(6)), 1
    good = _save(rng.integers(0, 256, (40, 50, 3), dtype=np.uint8))
    yield "bad_adler", good[:-17] + bytes([good[-17] ^ 0x40]) + good[-16:], 2     # last byte of the zlib trailer (IDAT's CRC and the 12 bytes of IEND follow)
    yield "truncated", good[: len(good) // 2], 2
    yield "bit_flip_in_idat", good[:100] + bytes([good[100] ^ 1]) + good[101:], 2
    yield "not_a_png", b"GIF89a" + bytes(64), 2
    # IDAT, another chunk, IDAT: Pillow ends the stream at the first non-IDAT chunk ("image file is truncated"), the
    # reference's worker drops the file (src/core/fastsig.py:36-37)
    rows = np.concatenate([np.zeros((40, 1), np.uint8), rng.integers(0, 256, (40, 150), dtype=np.uint8)], 1).tobytes()
    whole = _container(rows, 50, 40, 2, 6, 0, 1 << 30)
    z_at = whole.index(b"IDAT") - 4
    zlen = struct.unpack(">I", whole[z_at:z_at + 4])[0]
    z = whole[z_at + 8:z_at + 8 + zlen]

    def ch(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))

    yield "idat_text_idat", whole[:z_at] + ch(b"IDAT", z[: zlen // 2]) + ch(b"tEXt", b"k\x00v") + ch(b"IDAT", z[zlen // 2:]) + ch(b"IEND", b""), 2


def random_handmade(n: int, seed: int = 0):
    """n random files of every colour type x bit depth x interlacing the decoder takes, every filter type, sizes up to
    120 x 90, IDAT data in chunks of random size: (name, file bytes, what the reference's hashes see)."""
    rng = np.random.default_rng(seed)
    kinds = [(0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (2, 8), (2, 16), (3, 1), (3, 2), (3, 4), (3, 8), (4, 8), (4, 16), (6, 8), (6, 16)]
    for k in range(n):
        w, h = int(rng.integers(1, 121)), int(rng.integers(1, 91))
        ctype, depth = kinds[int(rng.integers(0, len(kinds)))]
        lace = int(rng.integers(0, 2))
        chans = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
        if depth >= 8:
            a = rng.integers(0, 256, (h, w, chans * depth // 8), dtype=np.uint8)
            if rng.integers(0, 2):                                  # smooth content: matches, long filter chains
                a = (np.cumsum(rng.integers(0, 3, a.shape), 1) % 256).astype(np.uint8)
            if ctype == 0 and depth == 16:
                a[:, :, 0] = np.where(rng.random((h, w)) < 0.7, 0, a[:, :, 0])
            raw = _adam7_stream(a, 8, rng, _ADAM7 if lace else ((0, 0, 1, 1),))
        else:
            raw = _adam7_stream(rng.integers(0, 1 << depth, (h, w, 1), dtype=np.uint8), depth, rng, _ADAM7 if lace else ((0, 0, 1, 1),))
        plte = rng.integers(0, 256, 3 * int(rng.integers(1, (1 << depth) + 1)), dtype=np.uint8).tobytes() if ctype == 3 else None
        data = _container2(raw, w, h, ctype, depth, lace, plte=plte, level=int(rng.integers(0, 10)), chunk=int(rng.choice([7, 100, 8192, 1 << 30])))
        with Image.open(io.BytesIO(data)) as im:
            ref = np.asarray(im.convert("L") if im.mode in ("P", "1", "LA", "I;16") or (im.mode == "L" and depth < 8) else im)
        yield f"handmade{k}_c{ctype}_d{depth}_i{lace}_{w}x{h}", data, ref


def random_cases(n: int, seed: int = 0):
    """n random files of the kinds the decoder takes (sizes up to 200 x 150, every mode, compression level and texture):
    (name, file bytes, what the reference's hashes see)."""
    rng = np.random.default_rng(seed)
    for k in range(n):
        w, h = int(rng.integers(1, 201)), int(rng.integers(1, 151))
        mode = ("L", "RGB", "RGBA", "P", "1")[int(rng.integers(0, 5))]
        texture = int(rng.integers(0, 3))
        if texture == 0:
            a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        elif texture == 1:
            a = np.repeat(np.repeat(rng.integers(0, 256, (h // 8 + 1, w // 8 + 1, 4), dtype=np.uint8), 8, 0), 8, 1)[:h, :w]
        else:
            yy, xx = np.mgrid[0:h, 0:w]
            a = np.stack([xx * 3 % 256, yy * 5 % 256, (xx + yy) % 256, (xx * yy) % 256], -1).astype(np.uint8)
        if mode == "P":
            im = Image.fromarray(np.ascontiguousarray(a[:, :, :3])).quantize(int(rng.integers(2, 257)))
        elif mode == "1":
            im = Image.fromarray(np.ascontiguousarray(a[:, :, 0])).convert("1")
        else:
            im = Image.fromarray(np.ascontiguousarray({"L": a[:, :, 0], "RGB": a[:, :, :3], "RGBA": a}[mode]))
        b = io.BytesIO()
        im.save(b, "PNG", compress_level=int(rng.integers(0, 10)), optimize=bool(rng.integers(0, 2)))
        data = b.getvalue()
        with Image.open(io.BytesIO(data)) as back:
            ref = np.asarray(back.convert("L") if back.mode in ("P", "1") else back)
        yield f"random{k}_{mode}_{w}x{h}", data, ref
