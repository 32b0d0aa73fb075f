"""GIF files for the decoder tests: what Pillow's own writer produces (palette and gray images, interlaced or not, animated,
with transparency, comments, optimised palettes) and hand-made streams for what it never varies -- clear codes at any
distance or never (a full dictionary), long runs (a code that is the entry being built), tiny sub-blocks, local palettes,
short palettes, nothing behind the last pixel -- plus what the decoder hands back."""
from __future__ import annotations

import io
import struct

import numpy as np
from PIL import Image


def _pillow(data: bytes):
    """What the reference's hashes see of the file: Image.open (the first frame), convert("L")."""
    with Image.open(io.BytesIO(data)) as im:
        im.load()
        return np.asarray(im.convert("L"))


def lzw(indices, bits: int, *, clear_every: int = 0, never_clear: bool = False, end: bool = True, block: int = 255) -> bytes:
    """Index stream -> min-code-size byte + sub-blocks (+ terminator).  clear_every: a clear code after that many codes;
    never_clear: keep a full dictionary (codes stay 12 bits wide) instead of starting over."""
    clear, stop = 1 << bits, (1 << bits) + 1
    out_bits = []

    def put(code, size):
        out_bits.append((code, size))

    table, emitted, size = {}, 0, bits + 1

    def emit(code):
        nonlocal emitted, size
        put(code, size)
        emitted += 1
        if emitted >= 2:
            e = clear + 2 + (emitted - 2)
            if e < 4096 and e == (1 << size) - 1 and size < 12:
                size += 1

    def reset():
        nonlocal table, emitted, size
        put(clear, size)
        table, emitted, size = {}, 0, bits + 1

    reset()
    prefix = None
    for px in indices:
        px = int(px)
        if prefix is None:
            prefix = px
            continue
        key = (prefix, px)
        if key in table:
            prefix = table[key]
            continue
        emit(prefix)
        index = clear + 2 + (emitted - 1)
        if index < 4096:
            table[key] = index
        elif not never_clear:
            reset()
        if clear_every and emitted >= clear_every:
            reset()
        prefix = px
    if prefix is not None:
        emit(prefix)
    if end:
        put(stop, size)
    acc = n = 0
    raw = bytearray()
    for code, width in out_bits:
        acc |= code << n
        n += width
        while n >= 8:
            raw.append(acc & 255)
            acc >>= 8
            n -= 8
    if n:
        raw.append(acc & 255)
    body = bytearray([bits])
    for o in range(0, len(raw), block):
        part = raw[o:o + block]
        body += bytes([len(part)]) + part
    return bytes(body) + b"\x00"


def gif(w, h, data: bytes, *, palette=None, local=None, interlace=False, frame=None, ext=b"", trailer=True, version=b"GIF89a") -> bytes:
    """A GIF around LZW data given as it is.  palette / local: N x 3 arrays (N a power of two); frame: (x0, y0, w, h) of the image
    descriptor (the screen by default)."""
    def table(p):
        return np.asarray(p, np.uint8).tobytes()

    def size_bits(p):
        return int(np.log2(len(p))) - 1

    flags = (0x80 | size_bits(palette)) if palette is not None else 0
    out = version + struct.pack("<HHBBB", w, h, flags, 0, 0)
    if palette is not None:
        out += table(palette)
    out += ext
    x0, y0, fw, fh = frame if frame else (0, 0, w, h)
    lf = (0x40 if interlace else 0) | ((0x80 | size_bits(local)) if local is not None else 0)
    out += b"," + struct.pack("<HHHHB", x0, y0, fw, fh, lf)
    if local is not None:
        out += table(local)
    return out + data + (b";" if trailer else b"")


def _rows_interlaced(a: np.ndarray) -> np.ndarray:
    h = a.shape[0]
    order = list(range(0, h, 8)) + list(range(4, h, 8)) + list(range(2, h, 4)) + list(range(1, h, 2))
    return a[order]


def supported(full: bool = False):
    """Yields (name, file bytes, expected luma)."""
    rng = np.random.default_rng(13)
    sizes = [(1, 1), (2, 3), (7, 5), (16, 16), (17, 33), (64, 64), (101, 77), (300, 200)] + ([(512, 512), (33, 1000), (1000, 31)] if full else [])
    for (w, h) in sizes:
        yy, xx = np.mgrid[0:h, 0:w]
        photo = np.stack([(xx * 3 + yy) % 256, (yy * 5) % 256, (xx * yy // 3) % 256], -1).astype(np.uint8) ^ rng.integers(0, 16, (h, w, 3), dtype=np.uint8)
        flat = np.repeat(np.repeat(rng.integers(0, 256, (h // 8 + 1, w // 8 + 1, 3), dtype=np.uint8), 8, 0), 8, 1)[:h, :w]
        for kind, a in (("photo", photo), ("flat", flat)):
            for mode in ("P", "L", "P16"):
                im = Image.fromarray(a)
                im = im.convert("L") if mode == "L" else im.quantize(16 if mode == "P16" else 256)
                for kw in ({}, {"interlace": False}, {"optimize": True}, {"transparency": 3}, {"comment": b"made for the tests"}):
                    if not full and kw and not (kind == "flat" and mode == "P"):
                        continue
                    b = io.BytesIO()
                    im.save(b, "GIF", **kw)
                    yield f"pillow_{kind}_{mode}_{w}x{h}_{kw}", b.getvalue(), _pillow(b.getvalue())
        b = io.BytesIO()
        Image.fromarray(photo).quantize(64).save(b, "GIF", save_all=True, append_images=[Image.fromarray(flat).quantize(32)], duration=40, loop=0)
        yield f"pillow_animated_{w}x{h}", b.getvalue(), _pillow(b.getvalue())


def handmade(full: bool = False):
    """Yields (name, file bytes, expected luma or None where Pillow itself refuses the file)."""
    rng = np.random.default_rng(14)
    for (w, h) in [(9, 7), (64, 40), (200, 150)] + ([(640, 480)] if full else []):
        pal256 = rng.integers(0, 256, (256, 3), dtype=np.uint8)
        pal16 = rng.integers(0, 256, (16, 3), dtype=np.uint8)
        gray = np.repeat(np.arange(256, dtype=np.uint8)[:, None], 3, 1)
        noise = rng.integers(0, 256, (h, w), dtype=np.uint8)
        runs = np.repeat(rng.integers(0, 256, (h, w // 16 + 1), dtype=np.uint8), 16, 1)[:, :w]
        low = rng.integers(0, 16, (h, w), dtype=np.uint8)
        cases = {
            "noise": gif(w, h, lzw(noise.ravel(), 8), palette=pal256),
            "noise_never_clear": gif(w, h, lzw(noise.ravel(), 8, never_clear=True), palette=pal256),
            "noise_clear_every_5": gif(w, h, lzw(noise.ravel(), 8, clear_every=5), palette=pal256),
            "noise_clear_every_1": gif(w, h, lzw(noise.ravel(), 8, clear_every=1), palette=pal256),
            "runs": gif(w, h, lzw(runs.ravel(), 8), palette=pal256),
            "runs_never_clear": gif(w, h, lzw(runs.ravel(), 8, never_clear=True), palette=pal256),
            "runs_blocks_of_3": gif(w, h, lzw(runs.ravel(), 8, block=3), palette=pal256),
            "one_colour": gif(w, h, lzw(np.full(w * h, 7), 8), palette=pal256),
            "one_colour_never_clear": gif(w, h, lzw(np.full(w * h, 7), 8, never_clear=True), palette=pal256),
            "low4": gif(w, h, lzw(low.ravel(), 4), palette=pal16),
            "low4_wide_codes": gif(w, h, lzw(low.ravel(), 8), palette=pal16),                # indices below 16, code size 8
            "beyond_palette": gif(w, h, lzw(noise.ravel(), 8), palette=pal16),               # indices beyond a 16-entry palette
            "two_bits": gif(w, h, lzw((low & 3).ravel(), 2), palette=pal16[:4]),
            "interlaced": gif(w, h, lzw(_rows_interlaced(noise).ravel(), 8), palette=pal256, interlace=True),
            "interlaced_runs": gif(w, h, lzw(_rows_interlaced(runs).ravel(), 8, never_clear=True), palette=pal256, interlace=True),
            "local_palette": gif(w, h, lzw(noise.ravel(), 8), palette=pal16, local=pal256),
            "local_only": gif(w, h, lzw(noise.ravel(), 8), local=pal256),
            "no_palette": gif(w, h, lzw(noise.ravel(), 8)),
            "gray_ramp": gif(w, h, lzw(noise.ravel(), 8), palette=gray),
            "gray_ramp_local": gif(w, h, lzw(noise.ravel(), 8), palette=pal256, local=gray),
            "no_end_code": gif(w, h, lzw(noise.ravel(), 8, end=False), palette=pal256),
            "no_trailer": gif(w, h, lzw(noise.ravel(), 8), palette=pal256, trailer=False),
            "nothing_behind_the_last_pixel": gif(w, h, lzw(noise.ravel(), 8, end=False)[:-1], palette=pal256, trailer=False),
            "more_pixels_than_the_frame": gif(w, h, lzw(np.concatenate([noise.ravel(), noise.ravel()[: w * 3]]), 8), palette=pal256),
            "gif87a": gif(w, h, lzw(noise.ravel(), 8), palette=pal256, version=b"GIF87a"),
            "control_and_comment": gif(w, h, lzw(noise.ravel(), 8), palette=pal256,
                                       ext=b"!\xf9\x04\x01\x0a\x00\x05\x00" + b"!\xfe\x05hello\x03abc\x00" + b"!\xff\x0bNETSCAPE2.0\x03\x01\x00\x00\x00"),
            # what the decoder leaves to Pillow
            "frame_inside_the_screen": gif(w + 4, h + 2, lzw(noise.ravel(), 8), palette=pal256, frame=(2, 1, w, h)),
            "frame_beyond_the_screen": gif(w - 1, h, lzw(noise.ravel(), 8), palette=pal256, frame=(0, 0, w, h)),
            "end_code_early": gif(w, h, lzw(noise.ravel()[: w * h // 2], 8), palette=pal256),
        }
        for name, data in cases.items():
            try:
                ref = _pillow(data)
            except (OSError, EOFError, SyntaxError, ValueError):
                ref = None
            yield f"{name}_{w}x{h}", data, ref


LEFT_TO_PILLOW = ("frame_inside_the_screen", "frame_beyond_the_screen", "end_code_early")


def refused():
    """Yields (name, file bytes, expected status): 1 = left to Pillow, 2 = damaged (Pillow raises)."""
    rng = np.random.default_rng(15)
    w, h = 40, 30
    pal = rng.integers(0, 256, (256, 3), dtype=np.uint8)
    noise = rng.integers(0, 256, w * h, dtype=np.uint8)
    good = gif(w, h, lzw(noise, 8), palette=pal)
    yield "not_gif", b"GIF88a" + good[6:], 2
    yield "truncated_half", good[: len(good) // 2], 2
    yield "truncated_in_the_last_block", good[:-40], 2
    yield "code_size_9", gif(w, h, b"\x09" + lzw(noise, 8)[1:], palette=pal), 1
    yield "code_size_1", gif(w, h, b"\x01" + lzw(noise & 1, 2)[1:], palette=pal), 1
    yield "no_image", good[: 13 + 768] + b";", 1
    yield "unknown_tag", good[: 13 + 768] + b"\x55" + good[13 + 768:], 1
    yield "zero_width", gif(0, h, lzw(noise, 8), palette=pal), 1
