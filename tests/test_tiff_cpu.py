"""The product's TIFF directory parsing (csrc/ke_tiff_parse.h, the header ke_tiff.hip compiles) built for the host
(oracle/libkeyes_tiff_cpu.so) against the installed Pillow: pixel-exact on every file the unpacker takes, a refusal for the
rest, and under random damage never a file taken that Pillow refuses or decodes differently."""
from __future__ import annotations

import ctypes as C
import os
import warnings

import numpy as np
from PIL import ImageFile

import _tiff_cases as T
from oracle import oracle as O


def _lib():
    path = os.path.join(os.path.dirname(O.__file__), "libkeyes_tiff_cpu.so")
    if not os.path.exists(path):
        O.build(force=True)
    L = C.CDLL(path)
    L.ko_tiff_probe.argtypes = [C.c_void_p, C.c_uint64] + [C.POINTER(C.c_int32)] * 3
    L.ko_tiff_decode.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    return L


def _decode(L, data: bytes):
    buf = np.frombuffer(data + b"\0", np.uint8)
    w, h, ch = C.c_int32(), C.c_int32(), C.c_int32()
    st = L.ko_tiff_probe(buf.ctypes.data, len(data), C.byref(w), C.byref(h), C.byref(ch))
    if st:
        return st, None
    out = np.empty((h.value, w.value, ch.value) if ch.value > 1 else (h.value, w.value), np.uint8)
    return L.ko_tiff_decode(buf.ctypes.data, len(data), out.ctypes.data), out


def strict_pillow(data: bytes):
    saved, ImageFile.LOAD_TRUNCATED_IMAGES = ImageFile.LOAD_TRUNCATED_IMAGES, False
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return T._pillow(data)
    except Exception:
        return None
    finally:
        ImageFile.LOAD_TRUNCATED_IMAGES = saved


def test_unpacking_matches_pillow():
    L = _lib()
    n = 0
    for name, data, ref in list(T.supported()) + list(T.handmade()):
        st, out = _decode(L, data)
        if ref is None or name.startswith(T.LEFT_TO_PILLOW):
            assert st != 0, name
            continue
        assert st == 0, name
        assert out.shape == ref.shape and np.array_equal(out, ref), name
        n += 1
    assert n > 90


def test_files_outside_the_unpacker_are_refused():
    L = _lib()
    for name, data, expected in T.refused():
        st, _ = _decode(L, data)
        if expected is None:
            assert st != 0 and strict_pillow(data) is None, name
            continue
        assert st == expected, name
        if expected == 2:
            assert strict_pillow(data) is None, name


def damaged(rng, pool, variants):
    for name, data, _ in pool:
        for v in range(variants):
            d = bytearray(data)
            how = v % 5
            ifd = int.from_bytes(data[4:8], "little" if data[:2] == b"II" else "big")
            if how == 0:                                     # a byte of the directory (or of what follows it) replaced
                pos = int(rng.integers(min(ifd, len(d) - 1), len(d)))
                d[pos] = int(rng.integers(0, 256))
            elif how == 1:                                   # a field set to a value that means something
                k = int(rng.integers(0, 12))
                pos = min(ifd + 2 + 12 * k + int(rng.choice([0, 2, 4, 8])), len(d) - 2)
                val = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 8, 16, 256, 257, 258, 259, 262, 273, 274, 277, 278, 284, 338, 339, 700]))
                d[pos:pos + 2] = val.to_bytes(2, "little" if data[:2] == b"II" else "big")
            elif how == 2:
                d = d[: int(rng.integers(8, len(d)))]
            elif how == 3:                                   # a header byte
                pos = int(rng.integers(0, 8))
                d[pos] = int(rng.integers(0, 256))
            else:
                pos = int(rng.integers(8, len(d)))
                d[pos] ^= 1 << int(rng.integers(0, 8))
            yield f"{name}/{v}", bytes(d)


def test_damaged_directories_are_never_decoded_differently_from_pillow():
    L = _lib()
    rng = np.random.default_rng(35)
    pool = [c for c in list(T.supported()) + list(T.handmade()) if c[2] is not None and c[2].shape[0] <= 80]
    taken = cases = 0
    for name, data in damaged(rng, pool, 25):
        cases += 1
        st, out = _decode(L, data)
        if st != 0:
            continue
        taken += 1
        ref = strict_pillow(data)
        assert ref is not None, name
        assert ref.shape == out.shape and np.array_equal(ref, out), name
    assert cases > 2500 and taken > 300
