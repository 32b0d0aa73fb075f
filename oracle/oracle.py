"""ctypes front-end of the CPU oracle (oracle/keyes_oracle.c) plus the pure-Python
restatement of the reference's cluster assembly.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.

Reference lines restated here (paths relative to /root/reference):
  cluster assembly + keeper + ordering : src/dup/scanner.py:304-356, 402-415
  ClusterBuilder                       : src/dup/cluster.py:22-70
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import PurePath
from typing import Iterable, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkeyes_oracle.so")


class OracleEdge(C.Structure):
    _fields_ = [("a", C.c_int64), ("b", C.c_int64), ("h", C.c_int32), ("bands", C.c_int32)]


EDGE_DTYPE = np.dtype([("a", "<i8"), ("b", "<i8"), ("h", "<i4"), ("bands", "<i4")])


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
        os.path.join(_HERE, "keyes_oracle.c")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        u8p, u64p, i64p, i32p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint64, C.c_int64, C.c_int32))
        L.ko_luma.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.ko_luma.restype = None
        L.ko_lanczos_coeffs.argtypes = [C.c_int, C.c_int, C.POINTER(i32p), C.POINTER(i32p)]
        L.ko_lanczos_coeffs.restype = C.c_int
        L.ko_free.argtypes = [C.c_void_p]
        L.ko_resample_lanczos.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ko_resample_lanczos.restype = C.c_int
        L.ko_resample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ko_resample.restype = C.c_int
        L.ko_fit_luma.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ko_fit_luma.restype = C.c_int
        L.ko_fit_box.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ko_fit_box.restype = None
        L.ko_tile_ahash.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.ko_tile_ahash.restype = None
        L.ko_dct8x8.argtypes = [C.c_void_p, C.c_void_p]
        L.ko_phash_from_tile.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.ko_phash_from_tile.restype = C.c_uint64
        L.ko_dhash_from_tile.argtypes = [C.c_void_p]
        L.ko_dhash_from_tile.restype = C.c_uint64
        L.ko_hash_image.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, u64p, u64p, C.c_void_p, C.c_void_p,
                                    C.POINTER(C.c_float)]
        L.ko_hash_image.restype = C.c_int
        L.ko_hash_batch.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ko_hash_batch.restype = C.c_int
        L.ko_synth_rgb.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        L.ko_synth_rgb.restype = None
        L.ko_synth_info.argtypes = [C.c_uint64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.ko_synth_info.restype = None
        L.ko_synth_info2.argtypes = [C.c_uint64, C.c_int64, C.POINTER(C.c_int64)] + [C.POINTER(C.c_int32)] * 4
        L.ko_synth_info2.restype = None
        L.ko_synth_hashes.argtypes = [C.c_uint64, C.c_int64, C.c_void_p]
        L.ko_synth_hashes.restype = None
        L.ko_scan_banded.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                     C.c_double, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
        L.ko_scan_banded.restype = C.c_int64
        L.ko_scan_bruteforce.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64]
        L.ko_scan_bruteforce.restype = C.c_int64
        L.ko_ssim_luma.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.ko_ssim_luma.restype = C.c_double
        _lib = L
    return _lib


def _ptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# --------------------------------------------------------------------------- sig
def luma(px: np.ndarray) -> np.ndarray:
    px = np.ascontiguousarray(px, dtype=np.uint8)
    ch = 1 if px.ndim == 2 else px.shape[2]
    out = np.empty(px.shape[:2], dtype=np.uint8)
    lib().ko_luma(_ptr(px), px.shape[0] * px.shape[1], ch, _ptr(out))
    return out


def lanczos_coeffs(in_size: int, out_size: int):
    """(bounds[out,2], kk[out,ksize]) exactly as Pillow's 8bpc resampler quantises them."""
    bp, kp = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
    ks = lib().ko_lanczos_coeffs(in_size, out_size, C.byref(bp), C.byref(kp))
    if ks < 0:
        raise MemoryError
    bounds = np.ctypeslib.as_array(bp, shape=(out_size, 2)).copy()
    kk = np.ctypeslib.as_array(kp, shape=(out_size, ks)).copy()
    lib().ko_free(bp)
    lib().ko_free(kp)
    return bounds, kk


def resample(L: np.ndarray, ow: int, oh: int) -> np.ndarray:
    L = np.ascontiguousarray(L, dtype=np.uint8)
    out = np.empty((oh, ow), dtype=np.uint8)
    rc = lib().ko_resample_lanczos(_ptr(L), L.shape[1], L.shape[0], ow, oh, _ptr(out))
    if rc:
        raise ValueError(f"ko_resample_lanczos rc={rc}")
    return out


def resample_filter(L: np.ndarray, ow: int, oh: int, filter: int) -> np.ndarray:
    """filter 0 = LANCZOS, 1 = BILINEAR (Pillow's 8-bit resampler either way)."""
    L = np.ascontiguousarray(L, dtype=np.uint8)
    out = np.empty((oh, ow), dtype=np.uint8)
    rc = lib().ko_resample(_ptr(L), L.shape[1], L.shape[0], ow, oh, filter, _ptr(out))
    if rc:
        raise ValueError(f"ko_resample rc={rc}")
    return out


def fit_luma(L: np.ndarray, ow: int, oh: int, filter: int = 2) -> np.ndarray:
    """ImageOps.fit(L-image, (ow, oh), BICUBIC) -- centre crop to the target aspect, then Pillow's resize with
    that box (src/dup/refine.py:48-49).  filter 2 = BICUBIC."""
    L = np.ascontiguousarray(L, dtype=np.uint8)
    out = np.empty((oh, ow), dtype=np.uint8)
    rc = lib().ko_fit_luma(_ptr(L), L.shape[1], L.shape[0], ow, oh, filter, _ptr(out))
    if rc:
        raise ValueError(f"ko_fit_luma rc={rc}")
    return out


def fit_box(w: int, h: int, ow: int, oh: int) -> np.ndarray:
    box = np.empty(4, dtype=np.float32)
    lib().ko_fit_box(w, h, ow, oh, _ptr(box))
    return box


def ssim_fit(px_a: np.ndarray, px_b: np.ndarray) -> float:
    """src/dup/refine.py:44-52 on decoded pixels of any two sizes."""
    la = luma(px_a) if px_a.ndim == 3 else px_a
    lb = luma(px_b) if px_b.ndim == 3 else px_b
    w, h = min(la.shape[1], lb.shape[1]), min(la.shape[0], lb.shape[0])
    return ssim_luma(fit_luma(la, w, h), fit_luma(lb, w, h))


def tile_ahash_bits(px: np.ndarray, grid: int = 4, tile: int = 8) -> int:
    """src/ui/dup_refine_parallel.py:59-83 on decoded pixels (HxW, HxWx3 or HxWx4)."""
    side = grid * tile
    thumb = resample_filter(luma(px), side, side, 1)
    words = np.zeros((side * side + 63) // 64, np.uint64)
    lib().ko_tile_ahash(_ptr(thumb), grid, tile, _ptr(words))
    return int.from_bytes(words.tobytes(), "little")


def small_gray(px: np.ndarray, size: int = 128) -> np.ndarray:
    """src/ui/dup_refine_parallel.py:203-207."""
    return resample_filter(luma(px), size, size, 1)


def mae01(a: np.ndarray, b: np.ndarray) -> float:
    """src/ui/dup_refine_parallel.py:208-210."""
    return float(np.mean(np.abs(a.astype(np.int16) - b.astype(np.int16))) / 255.0)


def dct8x8(tile32: np.ndarray) -> np.ndarray:
    tile32 = np.ascontiguousarray(tile32, dtype=np.uint8)
    out = np.empty((8, 8), dtype=np.float64)
    lib().ko_dct8x8(_ptr(tile32), _ptr(out))
    return out


def phash_from_tile(tile32: np.ndarray) -> tuple[int, float]:
    tile32 = np.ascontiguousarray(tile32, dtype=np.uint8)
    m = C.c_float()
    v = lib().ko_phash_from_tile(_ptr(tile32), C.byref(m))
    return int(v), float(m.value)


def dhash_from_tile(tile98: np.ndarray) -> int:
    tile98 = np.ascontiguousarray(tile98, dtype=np.uint8)
    return int(lib().ko_dhash_from_tile(_ptr(tile98)))


def hash_image(px: np.ndarray, want_tiles: bool = False):
    """px: HxW (L), HxWx3 (RGB) or HxWx4 (RGBX).  Returns (phash_u64, dhash_u64[, tile32, tile98, margin])."""
    px = np.ascontiguousarray(px, dtype=np.uint8)
    h, w = px.shape[:2]
    ch = 1 if px.ndim == 2 else px.shape[2]
    ph, dh, mg = C.c_uint64(), C.c_uint64(), C.c_float()
    t32 = np.empty((32, 32), np.uint8)
    t98 = np.empty((8, 9), np.uint8)
    rc = lib().ko_hash_image(_ptr(px), w, h, ch, C.byref(ph), C.byref(dh), _ptr(t32), _ptr(t98), C.byref(mg))
    if rc:
        raise ValueError(f"ko_hash_image rc={rc}")
    if want_tiles:
        return int(ph.value), int(dh.value), t32, t98, float(mg.value)
    return int(ph.value), int(dh.value)


def hash_batch(px: np.ndarray, want_dhash: bool = True):
    """px: N x H x W x C contiguous u8."""
    px = np.ascontiguousarray(px, dtype=np.uint8)
    n, h, w = px.shape[:3]
    ch = 1 if px.ndim == 3 else px.shape[3]
    ph = np.empty(n, np.uint64)
    dh = np.empty(n, np.uint64) if want_dhash else None
    rc = lib().ko_hash_batch(_ptr(px), n, w, h, ch, _ptr(ph), _ptr(dh))
    if rc:
        raise ValueError(f"ko_hash_batch rc={rc}")
    return ph, dh


def hamming64(a: int, b: int) -> int:
    """src/sig/phash.py:60-63."""
    return bin((int(a) ^ int(b)) & 0xFFFFFFFFFFFFFFFF).count("1")


def to_signed64(x: int) -> int:
    """src/core/fastsig.py:19-21."""
    v = int(x) & 0xFFFFFFFFFFFFFFFF
    return v - (1 << 64) if v >= (1 << 63) else v


# --------------------------------------------------------------------------- synthetic data
SEED = 20260604


def synth_rgb(index: int, w: int, h: int, seed: int = SEED) -> np.ndarray:
    out = np.empty((h, w, 3), np.uint8)
    lib().ko_synth_rgb(seed, index, w, h, _ptr(out))
    return out


def synth_info(index: int, seed: int = SEED) -> tuple[int, int, bool]:
    """(base index, brightness delta, is_variant) of corpus image `index`."""
    b, d, v = C.c_int64(), C.c_int32(), C.c_int32()
    lib().ko_synth_info(seed, index, C.byref(b), C.byref(d), C.byref(v))
    return int(b.value), int(d.value), bool(v.value)


def synth_info2(index: int, seed: int = SEED) -> tuple[int, int, bool, bool, int]:
    """(base, delta, is_variant, low_noise, replaced cells per 32): the variant class of corpus image `index`."""
    b, d, v, ln, q = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    lib().ko_synth_info2(seed, index, C.byref(b), C.byref(d), C.byref(v), C.byref(ln), C.byref(q))
    return int(b.value), int(d.value), bool(v.value), bool(ln.value), int(q.value)


def synth_rgb_batch(first: int, n: int, w: int, h: int, seed: int = SEED) -> np.ndarray:
    out = np.empty((n, h, w, 3), np.uint8)
    for k in range(n):
        lib().ko_synth_rgb(seed, first + k, w, h, _ptr(out[k]))
    return out


def synth_hashes(n: int, seed: int = SEED) -> np.ndarray:
    out = np.empty(n, np.uint64)
    lib().ko_synth_hashes(seed, n, _ptr(out))
    return out


# --------------------------------------------------------------------------- scan
def scan_banded(hashes, ids=None, sizes=None, threshold=8, band_bits=16, band_count=4, size_ratio=None,
                bucket_pair_cap=None):
    """Reference-shaped candidate generation.  Returns (edges[EDGE_DTYPE] sorted by id pair, counters[3])."""
    hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
    n = len(hashes)
    ids_a = None if ids is None else np.ascontiguousarray(ids, dtype=np.int64)
    sizes_a = None if sizes is None else np.ascontiguousarray(sizes, dtype=np.int64)
    counters = np.zeros(3, np.uint64)
    ratio = float(size_ratio) if size_ratio else 0.0
    cap = int(bucket_pair_cap) if bucket_pair_cap else 0
    capacity = 1 << 16
    while True:
        edges = np.zeros(capacity, EDGE_DTYPE)
        ne = lib().ko_scan_banded(_ptr(hashes), _ptr(ids_a), _ptr(sizes_a), n, threshold, band_bits, band_count,
                                  ratio, cap, _ptr(edges), capacity, _ptr(counters))
        if ne < 0:
            raise ValueError(f"ko_scan_banded rc={ne}")
        if ne <= capacity:
            return edges[:ne], counters
        capacity = int(ne)


def scan_bruteforce(hashes, threshold=8, band_bits=16, band_count=4):
    hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
    capacity = 1 << 16
    while True:
        edges = np.zeros(capacity, EDGE_DTYPE)
        ne = lib().ko_scan_bruteforce(_ptr(hashes), len(hashes), threshold, band_bits, band_count, _ptr(edges), capacity)
        if ne <= capacity:
            return edges[:ne]
        capacity = int(ne)


# --------------------------------------------------------------------------- clusters (pure Python, small inputs)
_EXT_PRIORITY = {"png": 4, "apng": 4, "webp": 3, "tiff": 2, "tif": 2, "bmp": 1, "gif": 1}


def _ext_priority(path: str) -> int:
    return _EXT_PRIORITY.get(PurePath(path).suffix.lower().lstrip("."), 0)


def assemble_clusters(files: Sequence[dict], edges: Iterable[tuple[int, int, int]]):
    """files: dicts with file_id, path, size, width, height.  edges: (file_id_a, file_id_b, h).

    Returns [(keeper_id, [(file_id, best_hamming), ...]), ...] in the reference's order
    (src/dup/scanner.py:304-356).
    """
    by_id = {}
    for f in files:
        by_id[f["file_id"]] = f  # last one wins, as the dict comprehension at :305
    parent: dict[int, int] = {}

    def find(x):
        parent.setdefault(x, x)
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    best: dict[int, int] = {}
    for a, b, h in edges:
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[ra] = rb
        for fid in (a, b):
            if fid not in best or h < best[fid]:
                best[fid] = h
    groups: dict[int, list[int]] = {}
    for fid in parent:
        groups.setdefault(find(fid), []).append(fid)

    def sz(f):
        return f.get("size") or 0

    def res(f):
        return (f.get("width") or 0) * (f.get("height") or 0)

    out = []
    for members in groups.values():
        if len(members) < 2:
            continue
        entries = [by_id[m] for m in sorted(members) if m in by_id]
        if len(entries) < 2:
            continue
        keeper = min(
            entries,
            key=lambda f: (-sz(f), -res(f), -_ext_priority(f["path"]), PurePath(f["path"]).suffix.lower(),
                           PurePath(f["path"]).name.lower(), f["file_id"]),
        )["file_id"]
        entries.sort(
            key=lambda f: (0 if f["file_id"] == keeper else 1, -sz(f), -res(f), -_ext_priority(f["path"]),
                           PurePath(f["path"]).name.lower(), f["file_id"])
        )
        out.append((keeper, entries))
    out.sort(key=lambda ke: (-max(sz(f) for f in ke[1]), PurePath(ke[1][0]["path"]).as_posix().lower()))
    return [(k, [(f["file_id"], best.get(f["file_id"])) for f in es]) for k, es in out]


def cluster_builder(matches: Iterable[tuple[int, int, bool]]):
    """src/dup/cluster.py:22-70 on (file_id_a, file_id_b, is_duplicate) triples ->
    [(representative, members, [match indices])] sorted by representative."""
    ml = [(i, a, b) for i, (a, b, d) in enumerate(matches) if d]
    parent: dict[int, int] = {}

    def find(x):
        parent.setdefault(x, x)
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for _, a, b in ml:
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[max(ra, rb)] = min(ra, rb)
    groups: dict[int, list[int]] = {}
    for node in list(parent):
        groups.setdefault(find(node), []).append(node)
    by_root: dict[int, list[int]] = {}
    for i, a, _ in ml:
        by_root.setdefault(find(a), []).append(i)
    out = [(min(m), sorted(m), by_root.get(r, [])) for r, m in groups.items()]
    out.sort(key=lambda c: c[0])
    return out


# --------------------------------------------------------------------------- ssim
def ssim_luma(a: np.ndarray, b: np.ndarray) -> float:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    assert a.shape == b.shape and a.ndim == 2
    return float(lib().ko_ssim_luma(_ptr(a), _ptr(b), a.shape[1], a.shape[0]))


# --------------------------------------------------------------------------- loader normalisation (test infrastructure)
def normalise_rgb(px: np.ndarray, orientation: int = 1) -> np.ndarray:
    """What the reference's defensive loader makes of decoded pixels (src/utils/image_io.py:116-131, 137-151): RGBA composited
    over white as Image.alpha_composite + convert("RGB") do (Pillow's AlphaComposite.c in integers), then turned as
    ImageOps.exif_transpose turns an image whose EXIF orientation is `orientation` (1..8).  px: HxWx3 or HxWx4 u8."""
    px = np.asarray(px, np.uint8)
    if px.shape[2] == 4:
        sa = px[..., 3:4].astype(np.int64)
        c = px[..., :3].astype(np.int64)
        t = c * (sa * 128) + 255 * (255 * 128 - sa * 128) + (0x80 << 7)
        px = np.where(sa == 0, 255, (((t >> 8) + t) >> 8) >> 7).astype(np.uint8)
    if orientation == 2:
        px = px[:, ::-1]
    elif orientation == 3:
        px = px[::-1, ::-1]
    elif orientation == 4:
        px = px[::-1]
    elif orientation == 5:
        px = px.transpose(1, 0, 2)
    elif orientation == 6:
        px = px[::-1].transpose(1, 0, 2)
    elif orientation == 7:
        px = px[::-1, ::-1].transpose(1, 0, 2)
    elif orientation == 8:
        px = px[:, ::-1].transpose(1, 0, 2)
    return np.ascontiguousarray(px)
