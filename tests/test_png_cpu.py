"""The product's PNG arithmetic (csrc/ke_png_core.h + ke_png_parse.h, the headers the HIP kernels compile) built for the host
(oracle/libkeyes_png_cpu.so) against the installed Pillow: pixel-exact on every file the decoder takes, the right refusal for
the rest.  The GPU kernels are then held against Pillow directly (tests/test_gpu_jpeg.py)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

import _png_cases as P
from oracle import oracle as O


def _lib():
    path = os.path.join(os.path.dirname(O.__file__), "libkeyes_png_cpu.so")
    if not os.path.exists(path):
        O.build(force=True)
    L = C.CDLL(path)
    L.ko_png_probe.argtypes = [C.c_void_p, C.c_uint64] + [C.POINTER(C.c_int32)] * 3
    L.ko_png_decode.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    return L


def _decode(L, data: bytes):
    buf = np.frombuffer(data + b"\0", np.uint8)
    w, h, ch = C.c_int32(), C.c_int32(), C.c_int32()
    st = L.ko_png_probe(buf.ctypes.data, len(data), C.byref(w), C.byref(h), C.byref(ch))
    if st:
        return st, None
    out = np.empty((h.value, w.value, ch.value) if ch.value > 1 else (h.value, w.value), np.uint8)
    return L.ko_png_decode(buf.ctypes.data, len(data), out.ctypes.data), out


def test_decoder_arithmetic_matches_pillow():
    L = _lib()
    n = 0
    for name, data, ref in list(P.supported()) + list(P.handmade()) + list(P.mapped()):
        st, out = _decode(L, data)
        assert st == 0, name
        assert out.shape == ref.shape and np.array_equal(out, ref), name
        n += 1
    assert n > 100


def test_files_outside_the_decoder_are_refused():
    L = _lib()
    for name, data, expected in P.refused():
        st, _ = _decode(L, data)
        assert st == expected, name
