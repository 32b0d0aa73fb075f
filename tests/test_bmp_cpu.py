"""The product's BMP header parsing (csrc/ke_bmp_parse.h, the header ke_bmp.hip compiles) built for the host
(oracle/libkeyes_bmp_cpu.so) against the installed Pillow: pixel-exact on every file the unpacker takes, a refusal for the
rest, and under random damage to the header never a file taken that Pillow refuses or decodes differently."""
from __future__ import annotations

import ctypes as C
import io
import os

import numpy as np
from PIL import Image, ImageFile

import _bmp_cases as B
from oracle import oracle as O


def _lib():
    path = os.path.join(os.path.dirname(O.__file__), "libkeyes_bmp_cpu.so")
    if not os.path.exists(path):
        O.build(force=True)
    L = C.CDLL(path)
    L.ko_bmp_probe.argtypes = [C.c_void_p, C.c_uint64] + [C.POINTER(C.c_int32)] * 3
    L.ko_bmp_decode.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    return L


def _decode(L, data: bytes):
    buf = np.frombuffer(data + b"\0", np.uint8)
    w, h, ch = C.c_int32(), C.c_int32(), C.c_int32()
    st = L.ko_bmp_probe(buf.ctypes.data, len(data), C.byref(w), C.byref(h), C.byref(ch))
    if st:
        return st, None
    out = np.empty((h.value, w.value, ch.value) if ch.value > 1 else (h.value, w.value), np.uint8)
    return L.ko_bmp_decode(buf.ctypes.data, len(data), out.ctypes.data), out


def strict_pillow(data: bytes):
    saved, ImageFile.LOAD_TRUNCATED_IMAGES = ImageFile.LOAD_TRUNCATED_IMAGES, False
    try:
        return B._pillow(data)
    except Exception:
        return None
    finally:
        ImageFile.LOAD_TRUNCATED_IMAGES = saved


def test_unpacking_matches_pillow():
    L = _lib()
    n = 0
    for name, data, ref in list(B.supported()) + list(B.handmade()):
        st, out = _decode(L, data)
        if ref is None:
            assert st != 0, name
            continue
        assert st == 0, name
        assert out.shape == ref.shape and np.array_equal(out, ref), name
        n += 1
    assert n > 100


def test_files_outside_the_unpacker_are_refused():
    L = _lib()
    for name, data, expected in B.refused():
        st, _ = _decode(L, data)
        assert st == expected, name
        if expected == 2:
            assert strict_pillow(data) is None, name


def damaged(rng, pool, variants):
    for name, data, _ in pool:
        for v in range(variants):
            d = bytearray(data)
            how = v % 4
            if how == 0:                                     # a header byte replaced
                pos = int(rng.integers(2, min(len(d), 140)))
                d[pos] = int(rng.integers(0, 256))
            elif how == 1:                                   # a header field set to a value that means something
                pos = int(rng.choice([10, 14, 18, 22, 25, 26, 28, 30, 46]))
                val = int(rng.choice([0, 1, 2, 3, 8, 12, 16, 24, 32, 40, 52, 54, 56, 108, 124, 255, 256, 1078]))
                d[pos:pos + 2] = val.to_bytes(2, "little")
            elif how == 2:
                d = d[: int(rng.integers(14, len(d)))]
            else:
                pos = int(rng.integers(14, min(len(d), 140)))
                del d[pos:pos + int(rng.integers(1, 5))]
            yield f"{name}/{v}", bytes(d)


def test_damaged_headers_are_never_decoded_differently_from_pillow():
    L = _lib()
    rng = np.random.default_rng(31)
    pool = [c for c in list(B.supported()) + list(B.handmade()) if c[2] is not None and c[2].shape[0] <= 80]
    taken = cases = 0
    for name, data in damaged(rng, pool, 24):
        cases += 1
        st, out = _decode(L, data)
        if st != 0:
            continue
        taken += 1
        ref = strict_pillow(data)
        assert ref is not None, name
        assert ref.shape == out.shape and np.array_equal(ref, out), name
    assert cases > 3000 and taken > 100
