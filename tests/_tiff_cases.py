"""TIFF files for the unpacker tests: what Pillow's own writer produces without compression (gray, RGB, RGBA, palette) and
hand-made directories for what it never varies -- both byte orders, many strips, SHORT and LONG fields, values at offsets,
WhiteIsZero, an unspecified fourth sample, a single BitsPerSample value for all samples, "every strip covers the image",
duplicate and unknown-type entries -- plus what the unpacker hands back."""
from __future__ import annotations

import io
import struct

import numpy as np
from PIL import Image


def _pillow(data: bytes):
    """What the reference's hashes see of the file: Image.open (the first directory), palette files through convert("L")."""
    with Image.open(io.BytesIO(data)) as im:
        im.load()
        return np.asarray(im.convert("L") if im.mode in ("P", "1") else im)


def tiff(entries, blobs=(), *, order="<", magic=42, pad_front=0, next_ifd=0) -> bytes:
    """entries: [(tag, type, count, values or bytes)] -- values that do not fit four bytes are placed behind the directory;
    blobs: [(name, bytes)] placed first (pixel data), their offsets available to entries as ("@", name)."""
    e = order
    body = bytearray(b"\0" * pad_front)
    where = {}
    for name, data in blobs:
        where[name] = 8 + len(body)
        body += data
        if len(body) % 2:
            body += b"\0"
    ifd = 8 + len(body)
    fmt = {1: "B", 2: "c", 3: "H", 4: "I", 5: "II", 6: "b", 7: "B", 8: "h", 9: "i", 11: "f", 12: "d", 16: "Q"}
    packed = []
    for tag, typ, count, values in entries:
        if isinstance(values, (bytes, bytearray)):
            raw = bytes(values)
        else:
            vals = [where[v[1]] if isinstance(v, tuple) and v[0] == "@" else v for v in values]
            if typ == 5:
                raw = b"".join(struct.pack(e + "II", *v) for v in vals)
            else:
                raw = b"".join(struct.pack(e + fmt.get(typ, "B"), v) for v in vals)
        packed.append((tag, typ, count, raw))
    extra = bytearray()
    at = ifd + 2 + 12 * len(packed) + 4
    out = bytearray()
    for tag, typ, count, raw in packed:
        if len(raw) <= 4:
            field = raw + b"\0" * (4 - len(raw))
        else:
            field = struct.pack(e + "I", at + len(extra))
            extra += raw + (b"\0" if len(raw) % 2 else b"")
        out += struct.pack(e + "HHI", tag, typ, count) + field
    head = (b"II" if e == "<" else b"MM") + struct.pack(e + "HI", magic, ifd)
    return head + bytes(body) + struct.pack(e + "H", len(packed)) + bytes(out) + struct.pack(e + "I", next_ifd) + bytes(extra)


def plain(a: np.ndarray, *, order="<", photo=None, rows=None, extra=None, long_fields=False, bits_single=False, resolution=True, more=(), strips_last_only=False):
    """An uncompressed chunky 8-bit TIFF of a (H x W or H x W x C array) in strips of `rows` rows."""
    h, w = a.shape[:2]
    spp = 1 if a.ndim == 2 else a.shape[2]
    rows = rows or h
    data = a.tobytes()
    stride = w * spp
    blobs, offs = [], []
    for s, y in enumerate(range(0, h, rows)):
        blobs.append((f"s{s}", data[y * stride:(y + rows) * stride]))
        offs.append(("@", f"s{s}"))
    if strips_last_only:                                   # several offsets although one strip covers the image: the last counts
        blobs = [("junk", b"\x55" * 40), ("s0", data)]
        offs = [("@", "junk"), ("@", "s0")]
    t = 4 if long_fields else 3
    photo = (2 if spp >= 3 else 1) if photo is None else photo
    entries = [(256, t, 1, [w]), (257, t, 1, [h]),
               (258, 3, 1 if bits_single else spp, [8] * (1 if bits_single else spp)),
               (259, 3, 1, [1]), (262, 3, 1, [photo]), (273, 4, len(offs), offs), (277, 3, 1, [spp]), (278, t, 1, [rows]),
               (279, 4, len(offs), [len(b[1]) for b in blobs[-len(offs):]])]
    if resolution:
        entries += [(282, 5, 1, [(72, 1)]), (283, 5, 1, [(72, 1)]), (296, 3, 1, [2])]
    if extra is not None:
        entries.append((338, 3, 1, [extra]))
    entries += list(more)
    entries.sort(key=lambda x: x[0])
    return tiff(entries, blobs, order=order)


def supported(full: bool = False):
    """Yields (name, file bytes, expected pixels)."""
    rng = np.random.default_rng(16)
    sizes = [(1, 1), (2, 3), (7, 5), (64, 64), (101, 77), (300, 200)] + ([(1000, 31), (33, 1000)] if full else [])
    for (w, h) in sizes:
        a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        for mode in ("RGB", "RGBA", "L", "P"):
            im = Image.fromarray(a).convert("RGB").quantize(200) if mode == "P" else Image.fromarray(a).convert(mode)
            b = io.BytesIO()
            im.save(b, "TIFF")
            yield f"pillow_{mode}_{w}x{h}", b.getvalue(), _pillow(b.getvalue())


def handmade(full: bool = False):
    """Yields (name, file bytes, expected pixels -- None where Pillow itself refuses the combination)."""
    rng = np.random.default_rng(17)
    for (w, h) in [(5, 4), (33, 17)] + ([(257, 129)] if full else []):
        g = rng.integers(0, 256, (h, w), dtype=np.uint8)
        c3 = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        c4 = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        cmap = rng.integers(0, 65536, 768).tolist()
        cases = {}
        for order in "<>":
            o = "II" if order == "<" else "MM"
            cases[f"gray_{o}"] = plain(g, order=order)
            cases[f"gray_white_is_zero_{o}"] = plain(g, order=order, photo=0)
            cases[f"rgb_{o}"] = plain(c3, order=order)
            cases[f"rgb_strips_of_3_{o}"] = plain(c3, order=order, rows=3)
            cases[f"rgb_strips_of_1_{o}"] = plain(c3, order=order, rows=1)
            cases[f"rgb_long_fields_{o}"] = plain(c3, order=order, long_fields=True, rows=2)
            cases[f"rgba_unassociated_{o}"] = plain(c4, order=order, extra=2)
            cases[f"rgba_no_extrasamples_{o}"] = plain(c4, order=order)
            cases[f"rgbx_{o}"] = plain(c4, order=order, extra=0)
            cases[f"rgba_premultiplied_{o}"] = plain(c4, order=order, extra=1)                  # left to Pillow
            cases[f"palette_{o}"] = plain(g, order=order, photo=3, more=[(320, 3, 768, cmap)])
            cases[f"rgb_one_bits_value_{o}"] = plain(c3, order=order, bits_single=True)
            cases[f"rgb_no_resolution_{o}"] = plain(c3, order=order, resolution=False)
            cases[f"rgb_rows_beyond_height_{o}"] = plain(c3, order=order, rows=h + 7)
            cases[f"rgb_last_offset_counts_{o}"] = plain(c3, order=order, strips_last_only=True)
            cases[f"gray_orientation_1_{o}"] = plain(g, order=order, more=[(274, 3, 1, [1])])
            cases[f"gray_orientation_6_{o}"] = plain(g, order=order, more=[(274, 3, 1, [6])])            # Pillow turns it: left to Pillow
            cases[f"gray_sampleformat_{o}"] = plain(g, order=order, more=[(339, 3, 1, [1])])
            cases[f"rgb_sampleformat_111_{o}"] = plain(c3, order=order, more=[(339, 3, 3, [1, 1, 1])])
            cases[f"gray_unknown_type_entry_{o}"] = plain(g, order=order, more=[(65000, 14, 1, [7])])
            cases[f"gray_software_{o}"] = plain(g, order=order, more=[(305, 2, 12, b"made by hand")])
            cases[f"gray_duplicate_width_{o}"] = plain(g, order=order, more=[(256, 3, 1, [w])])
            cases[f"gray_xmp_{o}"] = plain(g, order=order, more=[(700, 1, 30, b'<x tiff:Orientation="6"></x>  ')])     # left to Pillow
            cases[f"gray_planar2_{o}"] = plain(g, order=order, more=[(284, 3, 1, [2])])
            cases[f"gray_lzw_tag_{o}"] = plain(g, order=order, more=[(259, 3, 1, [5])])
        for name, data in cases.items():
            try:
                ref = _pillow(data)
            except Exception:
                ref = None
            yield f"{name}_{w}x{h}", data, ref


LEFT_TO_PILLOW = ("rgba_premultiplied", "gray_orientation_6", "gray_xmp", "gray_planar2", "gray_lzw_tag")


def refused():
    """Yields (name, file bytes, expected status): 1 = left to Pillow, 2 = damaged (Pillow raises)."""
    rng = np.random.default_rng(18)
    c3 = rng.integers(0, 256, (9, 12, 3), dtype=np.uint8)
    good = plain(c3, rows=4)
    yield "not_tiff", b"IJ" + good[2:], 2
    yield "bad_magic", good[:2] + b"\x2b\x01" + good[4:], 2
    yield "bigtiff", good[:2] + b"\x2b\x00" + good[4:], 1
    yield "no_directory", good[:4] + struct.pack("<I", len(good) + 10), 1
    def with_strip_at(off):
        return tiff([(256, 3, 1, [12]), (257, 3, 1, [9]), (258, 3, 3, [8, 8, 8]), (259, 3, 1, [1]), (262, 3, 1, [2]), (273, 4, 1, [off]), (277, 3, 1, [3]),
                     (278, 3, 1, [9])], [("px", c3.tobytes())])

    whole = with_strip_at(8)
    yield "strip_ends_behind_the_file", with_strip_at(len(whole) - 300), 2        # 9 rows of 36 bytes from 300 bytes before the end
    yield "strip_begins_behind_the_file", with_strip_at(len(whole) + 50), 2
    yield "bits_16", plain(c3.astype(np.uint8), more=[(258, 3, 3, [16, 16, 16])]), 1
    yield "tiles", plain(c3, more=[(324, 4, 1, [8])]), 1
    yield "exif_ifd", plain(c3, more=[(34665, 4, 1, [8])]), 1
