"""The product's GIF container walk and LZW arithmetic (csrc/ke_gif_core.h, the header ke_gif.hip compiles) built for the host
(oracle/libkeyes_gif_cpu.so) against the installed Pillow: the first frame's luma, pixel-exact, for every file the decoder takes;
a refusal for the rest; under random damage never a file taken that Pillow refuses or shows differently."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
from PIL import ImageFile

import _gif_cases as G
from oracle import oracle as O


def _lib():
    path = os.path.join(os.path.dirname(O.__file__), "libkeyes_gif_cpu.so")
    if not os.path.exists(path):
        O.build(force=True)
    L = C.CDLL(path)
    L.ko_gif_probe.argtypes = [C.c_void_p, C.c_uint64] + [C.POINTER(C.c_int32)] * 3
    L.ko_gif_decode.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    return L


def _decode(L, data: bytes):
    buf = np.frombuffer(data + b"\0", np.uint8)
    w, h, ch = C.c_int32(), C.c_int32(), C.c_int32()
    st = L.ko_gif_probe(buf.ctypes.data, len(data), C.byref(w), C.byref(h), C.byref(ch))
    if st:
        return st, None
    out = np.empty((h.value, w.value), np.uint8)
    st = L.ko_gif_decode(buf.ctypes.data, len(data), out.ctypes.data)
    return st, (out if st == 0 else None)


def strict_pillow(data: bytes):
    saved, ImageFile.LOAD_TRUNCATED_IMAGES = ImageFile.LOAD_TRUNCATED_IMAGES, False
    try:
        return G._pillow(data)
    except Exception:
        return None
    finally:
        ImageFile.LOAD_TRUNCATED_IMAGES = saved


def test_first_frame_matches_pillow():
    L = _lib()
    n = 0
    for name, data, ref in list(G.supported()) + list(G.handmade()):
        st, out = _decode(L, data)
        if ref is None or name.startswith(G.LEFT_TO_PILLOW):
            assert st != 0, name
            continue
        assert st == 0, name
        assert out.shape == ref.shape and np.array_equal(out, ref), name
        n += 1
    assert n > 120


def test_files_outside_the_decoder_are_refused():
    L = _lib()
    for name, data, expected in G.refused():
        st, _ = _decode(L, data)
        assert st == expected, name
        if expected == 2:
            assert strict_pillow(data) is None, name


def damaged(rng, pool, variants):
    for name, data, _ in pool:
        for v in range(variants):
            d = bytearray(data)
            how = v % 5
            if how == 0:
                pos = int(rng.integers(6, len(d)))
                d[pos] = int(rng.integers(0, 256))
            elif how == 1:
                pos = int(rng.integers(6, len(d)))
                d[pos] ^= 1 << int(rng.integers(0, 8))
            elif how == 2:
                d = d[: int(rng.integers(13, len(d)))]
            elif how == 3:
                pos = int(rng.integers(6, len(d)))
                del d[pos:pos + int(rng.integers(1, 5))]
            else:
                pos = int(rng.integers(6, len(d)))
                d[pos:pos] = rng.integers(0, 256, int(rng.integers(1, 5)), dtype=np.uint8).tobytes()
            yield f"{name}/{v}", bytes(d)


def test_damaged_files_are_never_decoded_differently_from_pillow():
    L = _lib()
    rng = np.random.default_rng(33)
    pool = [c for c in list(G.supported()) + list(G.handmade()) if c[2] is not None and c[2].size <= 8000]
    taken = cases = 0
    for name, data in damaged(rng, pool, 30):
        cases += 1
        st, out = _decode(L, data)
        if st != 0:
            continue
        taken += 1
        ref = strict_pillow(data)
        assert ref is not None, name
        assert ref.shape == out.shape and np.array_equal(ref, out), name
    assert cases > 2000 and taken > 200
