"""PNG files for the decoder tests, made with the installed Pillow: what the GPU decoder takes (8-bit L / RGB / RGBA, every
compression level incl. stored blocks and optimised encoding, sizes from 1x1) and what it hands back (palette, gray+alpha,
16-bit, 1-bit, interlaced, truncated, damaged)."""
from __future__ import annotations

import io

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


def refused():
    """Yields (name, file bytes, expected status): 1 = left to Pillow, 2 = damaged."""
    rng = np.random.default_rng(4)
    a = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    for mode in ("P", "LA", "1"):
        b = io.BytesIO()
        Image.fromarray(a).convert(mode).save(b, "PNG")
        yield f"mode_{mode}", b.getvalue(), 1
    b = io.BytesIO()
    Image.fromarray(rng.integers(0, 65535, (20, 30)).astype(np.uint16)).save(b, "PNG")
    yield "16bit", b.getvalue(), 1
    good = _save(rng.integers(0, 256, (40, 50, 3), dtype=np.uint8))
    yield "truncated", good[: len(good) // 2], 2
    yield "bit_flip_in_idat", good[:100] + bytes([good[100] ^ 1]) + good[101:], 2
    yield "not_a_png", b"GIF89a" + bytes(64), 2
