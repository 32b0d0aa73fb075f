"""BMP files for the unpacker tests: what Pillow's own writer produces (24-bit, 32-bit BGRX, 8-bit palette and gray) and
hand-made headers for everything the parser reads that the writer never varies (top-down rows, every header size, the
BITFIELDS layouts, short palettes, data offsets with gaps / zero / pointing at the palette), plus what it hands back."""
from __future__ import annotations

import io
import struct

import numpy as np
from PIL import Image


def _pillow(data: bytes):
    """What the reference's hashes see of the file: Image.open, palette / gray files through convert("L")."""
    with Image.open(io.BytesIO(data)) as im:
        im.load()
        return np.asarray(im.convert("L") if im.mode in ("P", "L", "1") else im)


def bmp(w, h, bits, rows: bytes, *, hs=40, comp=0, colors=0, palette=b"", masks=(), offset=None, topdown=False, gap=0) -> bytes:
    """A BMP around stored rows given as they are.  masks: BITFIELDS -- inside the header for hs >= 52, behind it for hs == 40."""
    head = struct.pack("<iiHHIIiiII", w, -h if topdown else h, 1, bits, comp, len(rows), 2835, 2835, colors, 0)
    extra = b""
    if hs >= 52:
        m = list(masks) + [0] * (4 - len(masks))
        extra = struct.pack("<III", *m[:3]) + (struct.pack("<I", m[3]) if hs >= 56 else b"")
        extra += bytes(hs - 40 - len(extra))
    after = struct.pack("<III", *masks[:3]) if (hs == 40 and comp == 3) else b""
    body = struct.pack("<I", hs) + head + extra + after + palette + bytes(gap)
    off = 14 + len(body) if offset is None else offset
    return b"BM" + struct.pack("<IHHI", 14 + len(body) + len(rows), 0, 0, off) + body + rows


def _rows(a: np.ndarray, topdown=False) -> bytes:
    """H x W x bytes-per-pixel -> stored rows (bottom-up unless topdown), each padded to four bytes."""
    h = a.shape[0]
    flat = a.reshape(h, -1)
    pad = (-flat.shape[1]) % 4
    flat = np.concatenate([flat, np.zeros((h, pad), np.uint8)], 1)
    return (flat if topdown else flat[::-1]).tobytes()


def supported(full: bool = False):
    """Yields (name, file bytes, expected pixels)."""
    rng = np.random.default_rng(6)
    sizes = [(1, 1), (2, 3), (7, 5), (64, 64), (101, 77), (300, 200)] + ([(1000, 31), (33, 1000), (1024, 768)] if full else [])
    for (w, h) in sizes:
        a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        for mode in ("RGB", "RGBA", "L", "P"):
            im = Image.fromarray(a).convert("RGB").quantize(200) if mode == "P" else Image.fromarray(a).convert(mode)
            b = io.BytesIO()
            im.save(b, "BMP")
            yield f"pillow_{mode}_{w}x{h}", b.getvalue(), _pillow(b.getvalue())


def handmade(full: bool = False):
    """Yields (name, file bytes, expected pixels -- None where Pillow itself refuses the combination)."""
    rng = np.random.default_rng(7)
    for (w, h) in [(5, 4), (33, 17)] + ([(257, 129)] if full else []):
        px3 = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        px4 = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        idx = rng.integers(0, 256, (h, w, 1), dtype=np.uint8)
        cases = {
            "topdown24": bmp(w, h, 24, _rows(px3, True), topdown=True),
            "topdown32": bmp(w, h, 32, _rows(px4, True), topdown=True),
            "offset0_24": bmp(w, h, 24, _rows(px3), offset=0),
            "gap24": bmp(w, h, 24, _rows(px3), gap=13),
        }
        for hs in (52, 56, 64, 108, 124):
            cases[f"hs{hs}_24"] = bmp(w, h, 24, _rows(px3), hs=hs)
            cases[f"hs{hs}_32"] = bmp(w, h, 32, _rows(px4), hs=hs)
        for k, m in enumerate(((0xFF0000, 0xFF00, 0xFF, 0), (0xFF000000, 0xFF0000, 0xFF00, 0), (0xFF000000, 0xFF00, 0xFF, 0),
                               (0xFF000000, 0xFF0000, 0xFF00, 0xFF), (0xFF, 0xFF00, 0xFF0000, 0xFF000000), (0xFF0000, 0xFF00, 0xFF, 0xFF000000),
                               (0xFF000000, 0xFF00, 0xFF, 0xFF0000), (0, 0, 0, 0))):
            for hs in (40, 52, 56, 108, 124):
                cases[f"bitfields32_m{k}_hs{hs}"] = bmp(w, h, 32, _rows(px4), hs=hs, comp=3, masks=m)
        cases["bitfields24"] = bmp(w, h, 24, _rows(px3), comp=3, masks=(0xFF0000, 0xFF00, 0xFF))
        cases["bitfields24_hs108"] = bmp(w, h, 24, _rows(px3), hs=108, comp=3, masks=(0xFF0000, 0xFF00, 0xFF, 0))
        pal = rng.integers(0, 256, 1024, dtype=np.uint8).tobytes()
        gray = b"".join(bytes([v, v, v, 0]) for v in range(256))
        cases["pal256_colors0"] = bmp(w, h, 8, _rows(idx), palette=pal)
        cases["pal256"] = bmp(w, h, 8, _rows(idx), palette=pal, colors=256)
        cases["pal16_indices_beyond"] = bmp(w, h, 8, _rows(idx), palette=pal[:64], colors=16)
        cases["pal16_within"] = bmp(w, h, 8, _rows(idx % 16), palette=pal[:64], colors=16)
        cases["pal_gray"] = bmp(w, h, 8, _rows(idx), palette=gray, colors=256)
        cases["pal_gray_topdown"] = bmp(w, h, 8, _rows(idx, True), palette=gray, topdown=True)
        cases["pal_offset_at_palette"] = bmp(w, h, 8, _rows(idx), palette=pal, offset=14 + 40)      # Pillow steps over the palette itself
        cases["pal_offset0"] = bmp(w, h, 8, _rows(idx), palette=pal, offset=0)
        cases["pal_gap"] = bmp(w, h, 8, _rows(idx), palette=pal, gap=9)
        cases["pal_hs124"] = bmp(w, h, 8, _rows(idx), palette=pal, hs=124)
        for name, data in cases.items():
            try:
                ref = _pillow(data)
            except OSError:                   # a layout the plugin does not list (e.g. an alpha mask a 40-byte header cannot carry)
                ref = None
            yield f"{name}_{w}x{h}", data, ref


def refused():
    """Yields (name, file bytes, expected status): 1 = left to Pillow, 2 = damaged (Pillow raises)."""
    rng = np.random.default_rng(9)
    w, h = 12, 9
    px3 = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    good = bmp(w, h, 24, _rows(px3))
    yield "truncated_by_one", good[:-1], 2
    yield "truncated_half", good[: len(good) // 2], 2
    yield "not_bmp", b"BA" + good[2:], 2
    yield "offset_beyond_file", bmp(w, h, 24, _rows(px3), offset=100000), 2
    core = b"BM" + struct.pack("<IHHI", 26 + len(_rows(px3)), 0, 0, 26) + struct.pack("<IHHHH", 12, w, h, 1, 24) + _rows(px3)
    yield "os2_header", core, 1
    yield "rle8", bmp(w, h, 8, bytes(40), comp=1, palette=bytes(1024)), 1
    yield "bits16", bmp(w, h, 16, bytes(((w * 16 + 31) // 32 * 4) * h)), 1
    yield "bits4", bmp(w, h, 4, bytes(((w * 4 + 31) // 32 * 4) * h), palette=bytes(64)), 1
    yield "bits1", bmp(w, h, 1, bytes(4 * h), palette=bytes(8)), 1
    yield "two_colours_8bit", bmp(w, h, 8, bytes(12 * h), palette=bytes(8), colors=2), 1
    yield "odd_masks", bmp(w, h, 32, bytes(4 * w * h), comp=3, masks=(0xFF00, 0xFF0000, 0xFF, 0), hs=56), 1
    yield "header_size_41", good[:14] + struct.pack("<I", 41) + good[18:], 1
    yield "zero_width", bmp(0, h, 24, b""), 1
