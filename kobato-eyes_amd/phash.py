"""Perceptual hashes on the MI355X: drop-in for the reference's ``sig.phash``.

Mirrors src/sig/phash.py: ``phash(image)``, ``dhash(image)``, ``hamming64(a, b)`` with the
same argument meaning and return convention (Python ints wrapped to signed 64-bit, bits MSB
first).  The arithmetic runs in libkeyes_hip.so (csrc/ke_hash.hip); there is no CPU path.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np

from . import _native

_MASK64 = (1 << 64) - 1


def _to_signed(value: int) -> int:
    """Two's-complement wrap of an unsigned 64-bit value (src/sig/phash.py:29-30)."""
    value = int(value) & _MASK64
    return value - (1 << 64) if value >> 63 else value


def image_to_array(image) -> np.ndarray:
    """PIL image (or ndarray) -> contiguous u8 array HxW (L), HxWx3 (RGB) or HxWx4 (RGBX).

    ``convert("L")`` at src/sig/phash.py:25 reads only R,G,B of 4-byte modes, so RGBA/RGBX go
    to the device untouched; every other mode is first brought to "L" by Pillow itself (a
    decode-side normalisation, not part of the hashed arithmetic).
    """
    if isinstance(image, np.ndarray):
        arr = image
    else:
        mode = getattr(image, "mode", None)
        if mode not in ("L", "RGB", "RGBA", "RGBX"):
            image = image.convert("L")
        arr = np.asarray(image)
    if arr.dtype != np.uint8 or arr.ndim not in (2, 3) or (arr.ndim == 3 and arr.shape[2] not in (1, 3, 4)):
        raise ValueError(f"unsupported pixel array {arr.dtype} {arr.shape}")
    if arr.ndim == 3 and arr.shape[2] == 1:
        arr = arr[:, :, 0]
    return np.ascontiguousarray(arr)


def _hash_one(image, want_p: bool, want_d: bool, device: int):
    arr = image_to_array(image)
    h, w = arr.shape[:2]
    if h == 0 or w == 0:
        raise ValueError("cannot hash an empty image")
    ch = 1 if arr.ndim == 2 else arr.shape[2]
    ctx = _native.get_context(device)
    return ctx.hash_uniform(arr, 1, w, h, ch, want_phash=want_p, want_dhash=want_d)


def phash(image, *, device: int = 0) -> int:
    """64-bit pHash: 32x32 LANCZOS luma, DCT-II, 8x8 corner against the mean of its 63 AC terms
    (src/sig/phash.py:33-46).  Raises RuntimeError when the HIP library/GPU is unavailable."""
    ph, _ = _hash_one(image, True, False, device)
    return _to_signed(int(ph[0]))


def dhash(image, *, device: int = 0) -> int:
    """64-bit dHash: 9x8 LANCZOS luma, horizontal gradient sign (src/sig/phash.py:49-57)."""
    _, dh = _hash_one(image, False, True, device)
    return _to_signed(int(dh[0]))


def phash_dhash(image, *, device: int = 0) -> tuple[int, int]:
    """Both hashes from one upload (what src/core/fastsig.py:31-34 computes per file)."""
    ph, dh = _hash_one(image, True, True, device)
    return _to_signed(int(ph[0])), _to_signed(int(dh[0]))


def hash_batch(images: Sequence, *, want_dhash: bool = True, device: int = 0):
    """Ragged batch of PIL images / arrays -> (phash u64[n], dhash u64[n] | None, ok bool[n]).
    Images are grouped by channel count; each group is one ke_hash_images call."""
    arrays = [image_to_array(im) for im in images]
    n = len(arrays)
    ph = np.zeros(n, np.uint64)
    dh = np.zeros(n, np.uint64) if want_dhash else None
    ok = np.zeros(n, bool)
    ctx = _native.get_context(device)
    by_ch: dict[int, list[int]] = {}
    for i, a in enumerate(arrays):
        if a.shape[0] > 0 and a.shape[1] > 0:
            by_ch.setdefault(1 if a.ndim == 2 else a.shape[2], []).append(i)
    for ch, idx in by_ch.items():
        p, d, status = ctx.hash_images([arrays[i] for i in idx], want_dhash=want_dhash)
        ph[idx] = p
        if want_dhash:
            dh[idx] = d
        ok[idx] = status == 0
    return ph, dh, ok


def hamming64(a: int, b: int) -> int:
    """Hamming distance of two 64-bit hashes, signed or unsigned (src/sig/phash.py:60-63)."""
    return ((int(a) ^ int(b)) & _MASK64).bit_count() if hasattr(int, "bit_count") else bin((int(a) ^ int(b)) & _MASK64).count("1")


__all__ = ["phash", "dhash", "phash_dhash", "hash_batch", "hamming64"]
