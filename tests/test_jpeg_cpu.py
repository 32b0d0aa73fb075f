"""The product's JPEG arithmetic (csrc/ke_jpeg_core.h + ke_jpeg_parse.h, the headers the HIP kernels compile) built for the
host (oracle/libkeyes_jpeg_cpu.so) against the installed Pillow: pixel-exact on every file the decoder takes, and the right
refusal for the rest.  The GPU kernels are then held against Pillow directly (tests/test_gpu_jpeg.py)."""
from __future__ import annotations

import ctypes as C
import io
import os

import numpy as np
from PIL import Image

import _jpeg_cases as J
from oracle import oracle as O


def _lib():
    path = os.path.join(os.path.dirname(O.__file__), "libkeyes_jpeg_cpu.so")
    if not os.path.exists(path):
        O.build(force=True)                                 # make -B builds both libraries of oracle/
    L = C.CDLL(path)
    L.ko_jpeg_probe.argtypes = [C.c_void_p, C.c_uint64] + [C.POINTER(C.c_int32)] * 3
    L.ko_jpeg_decode.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    return L


def _decode(L, data: bytes):
    buf = np.frombuffer(data + b"\0", np.uint8)          # never a zero-length array
    w, h, ch = C.c_int32(), C.c_int32(), C.c_int32()
    st = L.ko_jpeg_probe(buf.ctypes.data, len(data), C.byref(w), C.byref(h), C.byref(ch))
    if st:
        return st, None
    out = np.empty((h.value, w.value, 3) if ch.value == 3 else (h.value, w.value), np.uint8)
    return L.ko_jpeg_decode(buf.ctypes.data, len(data), out.ctypes.data), out


def test_decoder_arithmetic_matches_pillow():
    L = _lib()
    n = 0
    for name, data, ref in J.supported():
        st, out = _decode(L, data)
        assert st == 0, name
        assert out.shape == ref.shape and np.array_equal(out, ref), name
        n += 1
    assert n > 150


def test_blocks_beyond_the_idct_bound_hand_the_file_back():
    """ke_idct_islow's bound: with the quantisation tables overwritten by larger and larger steps the decode is Pillow's until
    a block leaves 16 bits, and KE_JPEG_UNSUPPORTED from there (libjpeg's C form and Pillow's SIMD build differ beyond)."""
    L = _lib()
    good = [c for c in J.supported() if c[2].shape[0] >= 64 and "gray" not in c[0]][:6] + [c for c in J.supported() if "q100" in c[0] and c[2].shape[0] >= 64][:2]
    taken = {}
    for _, data, _ in good:
        for v in (255, 16, 4, 1):
            blob = J.with_quantisation_tables(data, v)
            st, out = _decode(L, blob)
            assert st in (0, 1)
            if st == 0:
                assert np.array_equal(out, np.asarray(Image.open(io.BytesIO(blob))))
            taken[v] = taken.get(v, 0) + (st == 0)
    assert taken[255] == 0 and taken[1] == len(good) and 0 < taken[16] < len(good)


def test_damaged_files_the_decoder_takes_are_decoded_as_pillow_does():
    """tests/fuzz_jpeg_damage.py, a bounded sample: one header byte changed / the entropy data flipped, overwritten, cut, with
    bytes deleted or inserted.  A file the decoder still takes has Pillow's pixels (reserved markers, bad segment lengths, bad
    sampling factors, unusable Huffman tables, renumbered or missing restart markers, data that runs dry are all refused: libjpeg
    stops on the former and patches the latter up by heuristics of its own)."""
    import fuzz_jpeg_damage as F

    cases, taken, wrong = F.check(F.cpu_decoder(), 12, 7, files=30)
    assert not wrong, wrong[:5]
    assert cases == 1080 and taken > 150


def test_files_outside_the_decoder_are_refused():
    L = _lib()
    for name, data, expected in J.refused():
        st, _ = _decode(L, data)
        assert st == expected, name


def test_quad_upsampler_equals_the_per_sample_one():
    """ke_upsample4 (the colour kernel's) == ke_upsample_at (the one held against Pillow above) for every component width 1..21,
    height 1..9 and sampling 1x1 / 2x1 / 2x2."""
    assert _lib().ko_jpeg_upsample_selftest() == 0


def test_random_files_decode_as_pillow_does():
    """Property: whatever Pillow (libjpeg-turbo) writes as a Huffman JPEG -- any size up to 90 x 70, grayscale or colour at
    4:4:4 / 4:2:2 / 4:2:0, any quality, sequential or progressive, optimised tables or not, restart intervals -- the decoder
    arithmetic yields Pillow's pixels."""
    import io

    from hypothesis import given, settings
    from hypothesis import strategies as st
    from PIL import Image

    L = _lib()

    @settings(max_examples=120, deadline=None, derandomize=True)
    @given(st.integers(1, 90), st.integers(1, 70), st.booleans(), st.integers(0, 2), st.integers(1, 100), st.booleans(), st.booleans(),
           st.integers(0, 9), st.integers(0, 2), st.integers(0, 2 ** 32 - 1))
    def check(w, h, gray, sub, quality, progressive, optimize, restart, texture, seed):
        rng = np.random.default_rng(seed)
        if texture == 0:
            a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        elif texture == 1:
            a = np.repeat(np.repeat(rng.integers(0, 256, (h // 8 + 1, w // 8 + 1, 3), dtype=np.uint8), 8, 0), 8, 1)[:h, :w]
        else:
            yy, xx = np.mgrid[0:h, 0:w]
            a = np.stack([xx * 3 % 256, yy * 5 % 256, (xx + yy) % 256], -1).astype(np.uint8)
        kw = {"quality": quality, "progressive": progressive, "optimize": optimize}
        if not gray:
            kw["subsampling"] = sub
        if restart:
            kw["restart_marker_blocks"] = restart
        b = io.BytesIO()
        try:
            Image.fromarray(a[:, :, 0] if gray else a).save(b, "JPEG", **kw)
        except TypeError:                                   # a Pillow without the restart options
            kw.pop("restart_marker_blocks")
            Image.fromarray(a[:, :, 0] if gray else a).save(b, "JPEG", **kw)
        data = b.getvalue()
        ref = np.asarray(Image.open(io.BytesIO(data)))
        status, out = _decode(L, data)
        assert status == 0 and out.shape == ref.shape and np.array_equal(out, ref)

    check()


def test_progressive_files_with_any_scan_script_decode_as_pillow_does():
    """Pillow's writer only knows libjpeg's default progression; other encoders (mozjpeg: what most image hosts serve) cut and
    order their scans differently.  120 files with random legal scripts (tests/_jpeg_prog_encoder.py: DC scans for one, some or
    all components, AC bands cut anywhere, successive approximation from any bit, scans shuffled as far as T.81 allows)."""
    L = _lib()
    n = 0
    for name, data, ref in J.scripted(120, 5):
        st, out = _decode(L, data)
        assert st == 0, name
        assert out.shape == ref.shape and np.array_equal(out, ref), name
        n += 1
    assert n == 120
