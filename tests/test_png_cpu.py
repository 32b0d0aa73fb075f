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
    for name, data, ref in list(P.supported()) + list(P.handmade()) + list(P.mapped()) + list(P.interlaced()) + list(P.wide()):
        st, out = _decode(L, data)
        assert st == 0, name
        assert out.shape == ref.shape and np.array_equal(out, ref), name
        n += 1
    assert n > 100


def test_damaged_files_the_decoder_takes_are_decoded_as_pillow_does():
    """tests/fuzz_jpeg_damage.py --png, a bounded sample: bytes changed, flipped, deleted, inserted anywhere behind the
    signature, truncation.  Nearly everything is refused (chunk checksums, the zlib checksum); what is taken has Pillow's
    pixels, and Pillow takes it too."""
    import fuzz_jpeg_damage as F

    cases, taken, wrong = F.check(F.cpu_decoder("png"), 25, 3, files=40, fmt="png")
    assert not wrong, wrong[:5]
    assert cases == 2000 and taken >= 1


def test_files_outside_the_decoder_are_refused():
    L = _lib()
    for name, data, expected in P.refused():
        st, _ = _decode(L, data)
        assert st == expected, name


def test_random_files_decode_as_pillow_does():
    """Property: whatever 8-bit image Pillow writes as a non-interlaced PNG -- any size up to 80 x 60, L / RGB / RGBA / P /
    bilevel, any compression level, optimised or not, with transparency for palette files -- the decoder arithmetic yields what
    the reference's hashes see (the pixels, or convert("L") of them for the mapped kinds)."""
    import io

    from hypothesis import given, settings
    from hypothesis import strategies as st
    from PIL import Image

    L = _lib()

    @settings(max_examples=120, deadline=None, derandomize=True)
    @given(st.integers(1, 80), st.integers(1, 60), st.sampled_from(["L", "RGB", "RGBA", "P", "1"]), st.integers(0, 9), st.booleans(),
           st.integers(0, 3), st.integers(0, 2 ** 32 - 1))
    def check(w, h, mode, level, optimize, texture, seed):
        rng = np.random.default_rng(seed)
        if texture == 0:
            a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        elif texture == 1:
            a = np.repeat(np.repeat(rng.integers(0, 256, (h // 8 + 1, w // 8 + 1, 4), dtype=np.uint8), 8, 0), 8, 1)[:h, :w]
        elif texture == 2:
            a = np.broadcast_to(rng.integers(0, 256, (1, 1, 4), dtype=np.uint8), (h, w, 4)).copy()
        else:
            yy, xx = np.mgrid[0:h, 0:w]
            a = np.stack([xx * 3 % 256, yy * 5 % 256, (xx + yy) % 256, (xx * yy) % 256], -1).astype(np.uint8)
        if mode == "P":
            im = Image.fromarray(a[:, :, :3]).quantize(int(rng.integers(2, 257)))
        elif mode == "1":
            im = Image.fromarray(a[:, :, 0]).convert("1")
        else:
            im = Image.fromarray({"L": a[:, :, 0], "RGB": a[:, :, :3], "RGBA": a}[mode])
        kw = {"compress_level": level, "optimize": optimize}
        if mode == "P" and seed & 1:
            kw["transparency"] = int(seed >> 8) % 2
        b = io.BytesIO()
        im.save(b, "PNG", **kw)
        data = b.getvalue()
        with Image.open(io.BytesIO(data)) as back:
            ref = np.asarray(back.convert("L") if back.mode in ("P", "1") else back)
        status, out = _decode(L, data)
        assert status == 0 and out.shape == ref.shape and np.array_equal(out, ref)

    check()


def test_random_handmade_files_decode_as_pillow_does():
    """What Pillow's writer never produces -- interlacing, 16 bits, sub-byte grayscale, short palettes, every filter type in any
    row, IDAT data in chunks of a few bytes -- in random combinations (tests/_png_cases.random_handmade)."""
    L = _lib()
    n = 0
    for name, data, ref in P.random_handmade(400, 17):
        st, out = _decode(L, data)
        assert st == 0, name
        assert out.shape == ref.shape and np.array_equal(out, ref), name
        n += 1
    assert n == 400


def test_animated_files_yield_frame_0():
    """Animated PNG: what Image.open shows is frame 0 -- the IDAT image; files in forms the parser does not mirror are refused
    (and whatever it takes is what Pillow shows, also when Pillow itself refuses the hand-edited file: then it must not be taken)."""
    import io

    from PIL import Image

    L = _lib()
    n = 0
    for name, data, ref in P.animated():
        st, out = _decode(L, data)
        if ref is None:
            assert st != 0, name
            continue
        assert st == 0, name
        assert out.shape == ref.shape and np.array_equal(out, ref), name
        n += 1
    assert n >= 30
