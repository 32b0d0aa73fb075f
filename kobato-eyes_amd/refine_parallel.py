"""The refine stage the application actually runs after a scan, on the MI355X: drop-in for the
reference's ``ui.dup_refine_parallel`` (src/ui/dup_refine_parallel.py).

Same functions, arguments, progress cadence and failure handling:
``tile_ahash_bits`` / ``tile_hamming`` (:59-88), ``refine_by_tilehash_parallel`` (:113-200),
``refine_by_pixels_parallel`` (:215-313).  Files are decoded by Pillow on a thread pool (EXIF
transpose included, as the reference does); the luma + BILINEAR thumbnails, the tile-mean bits and
the absolute-difference sums run in libkeyes_hip.so (ke_resize_luma_uniform, ke_tile_ahash,
ke_sad_pairs), batched per image shape.
"""
from __future__ import annotations

import importlib
import logging
import os
from collections import Counter
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Callable, Optional, Sequence

import numpy as np

from . import _native

_phash = importlib.import_module(".phash", __package__)
log = logging.getLogger("ui.dup_refine")


def _rebuild_cluster_like(cluster, files):
    return type(cluster)(files=list(files), keeper_id=cluster.keeper_id)


def _norm_path(p) -> Path:
    try:
        return Path(p).resolve(strict=False)
    except Exception:
        return Path(os.path.normcase(os.path.abspath(str(p))))


def _format_failure_summary(counts: Counter, samples: dict) -> str:
    parts = []
    for err, count in counts.items():
        sample = samples.get(err)
        parts.append(f"{count}×{err}" if sample is None else f"{count}×{err} (例: {sample})")
    return "; ".join(parts)


def _decode(path) -> np.ndarray:
    """Image.open + EXIF transpose -> pixel array (src/ui/dup_refine_parallel.py:67-70, 204-205)."""
    from PIL import Image, ImageOps

    with Image.open(path) as opened:
        return _phash.image_to_array(ImageOps.exif_transpose(opened))


def _thumbnails(arrays: Sequence[np.ndarray], side: int, device: int) -> list:
    """BILINEAR luma thumbnails of side x side for decoded arrays of any shapes, batched per shape."""
    ctx = _native.get_context(device)
    out: list = [None] * len(arrays)
    groups: dict[tuple, list[int]] = {}
    for i, a in enumerate(arrays):
        groups.setdefault(a.shape, []).append(i)
    for shape, idx in groups.items():
        h, w = shape[:2]
        ch = 1 if len(shape) == 2 else shape[2]
        stack = np.stack([arrays[i] for i in idx])
        thumbs = ctx.resize_luma_uniform(stack, len(idx), w, h, ch, side, side, filter=1)
        for k, i in enumerate(idx):
            out[i] = thumbs[k]
    return out


_GPU_SUFFIXES = {"jpeg": (".jpg", ".jpeg", ".jpe", ".jfif"), "png": (".png", ".apng"), "bmp": (".bmp",), "gif": (".gif",), "tiff": (".tif", ".tiff")}


def _thumbnails_decoded_on_gpu(paths: Sequence[Path], side: int, device: int) -> dict:
    """{path: side x side BILINEAR luma thumbnail} for the JPEG / PNG / BMP / GIF / TIFF files whose pixels ``_decode`` would return exactly as
    the GPU decoders do -- every kind they take, unless the file carries an EXIF orientation to apply (ke_*_caveats).  The
    files are read, decoded and shrunk without their pixels ever being in host memory; files left out (other formats,
    refused, damaged, turned) are for ``_decode``.  ``KE_GPU_REFINE_DECODE=0`` turns the route off."""
    out: dict = {}
    if os.environ.get("KE_GPU_REFINE_DECODE", "1") == "0":
        return out
    ctx = _native.get_context(device)
    not_laid = np.uint64(0xFFFFFFFFFFFFFFFF)
    for kind, suffixes in _GPU_SUFFIXES.items():
        if os.environ.get({"jpeg": "KE_GPU_JPEG", "png": "KE_GPU_PNG", "bmp": "KE_GPU_BMP", "gif": "KE_GPU_GIF", "tiff": "KE_GPU_TIFF"}[kind], "1") == "0":
            continue
        mine = [p for p in paths if str(p).lower().endswith(suffixes)]
        at = 0
        while at < len(mine):
            stop, estimate = at, 0                                   # a few GB of decoded pixels per call
            while stop < len(mine) and stop - at < 8192 and (estimate < (4 << 30) or stop == at):
                try:
                    estimate += 48 * os.path.getsize(mine[stop])
                except OSError:
                    pass
                stop += 1
            part = mine[at:stop]
            at = stop
            try:
                dev, off, w, h, c, st, flags = ctx.decode_files_owned([str(p) for p in part], kind, by_shape=True)
            except (RuntimeError, ValueError):
                continue                                             # the Pillow route decides about these files
            if not dev:
                continue
            try:
                laid = np.nonzero(off != not_laid)[0]
                laid = laid[np.argsort(off[laid], kind="stable")]   # layout order: one run per (width, height, channels)
                shapes = np.stack([w[laid], h[laid], c[laid]], 1)
                cuts = np.nonzero((shapes[1:] != shapes[:-1]).any(1))[0] + 1
                for run in np.split(np.arange(len(laid)), cuts):
                    idx = laid[run]
                    ww, hh, cc = int(w[idx[0]]), int(h[idx[0]]), int(c[idx[0]])
                    thumbs = ctx.resize_luma_uniform(dev + int(off[idx[0]]), len(idx), ww, hh, cc, side, side, filter=1)
                    for k, i in enumerate(idx.tolist()):
                        if st[i] == 0 and not flags[i] & 1:
                            out[part[i]] = thumbs[k]
                # files to turn first (src/ui/dup_refine_parallel.py:67-70: ImageOps.exif_transpose before the resize -- every
                # camera writes the tag): turned on the device into a buffer of their own, then shrunk group by group
                orient = (flags >> 8) & 15
                turn = np.nonzero((st == 0) & (c == 3) & ((flags & 1) == 1) & (orient >= 2) & (orient <= 8) & (off != not_laid))[0]
                if len(turn):
                    dev2, off2, w2, h2 = ctx.normalise_rgb(dev, off[turn], w[turn], h[turn], c[turn], orient[turn], by_shape=True)
                    try:
                        order = np.argsort(off2, kind="stable")
                        shapes2 = np.stack([w2[order], h2[order]], 1)
                        cuts2 = np.nonzero((shapes2[1:] != shapes2[:-1]).any(1))[0] + 1
                        for run in np.split(order, cuts2):
                            thumbs = ctx.resize_luma_uniform(dev2 + int(off2[run[0]]), len(run), int(w2[run[0]]), int(h2[run[0]]), 3, side, side, filter=1)
                            for k, j in enumerate(run.tolist()):
                                out[part[int(turn[j])]] = thumbs[k]
                    finally:
                        ctx.free(dev2)
            except (RuntimeError, ValueError):
                pass
            finally:
                ctx.free(dev)
    return out


def tile_ahash_from_arrays(arrays: Sequence[np.ndarray], grid: int = 4, tile: int = 8, *, device: int = 0) -> list:
    """Tile aHash of decoded images -> Python ints (little-endian packing of the reference)."""
    if not arrays:
        return []
    side = grid * tile
    thumbs = np.stack(_thumbnails(arrays, side, device))
    words = _native.get_context(device).tile_ahash(thumbs, len(arrays), grid, tile)
    return [int.from_bytes(row.tobytes(), "little") for row in words]


def tile_ahash_bits(path, grid: int = 4, tile: int = 8, *, device: int = 0) -> int:
    return tile_ahash_from_arrays([_decode(path)], grid, tile, device=device)[0]


def tile_hamming(a_bits: int, b_bits: int) -> int:
    return bin(a_bits ^ b_bits).count("1")


def _decode_all(paths: Sequence[Path], workers: int):
    """[(path, array | Exception)] in input order."""
    def work(p):
        try:
            return p, _decode(p)
        except Exception as exc:
            return p, exc

    with ThreadPoolExecutor(max_workers=max(1, workers)) as ex:
        return list(ex.map(work, paths))


def refine_by_tilehash_parallel(clusters: Sequence, grid: int = 4, tile: int = 8, max_bits: int = 32,
                                io_workers: Optional[int] = None, tick: Optional[Callable] = None,
                                is_cancelled: Optional[Callable[[], bool]] = None, *, device: int = 0) -> list:
    if is_cancelled and is_cancelled():
        return []
    # --- phase 1: signatures of every distinct file
    all_paths = [_norm_path(e.file.path) for cl in clusters for e in cl.files]
    uniq_paths = sorted(set(all_paths), key=lambda p: (p.anchor, str(p.parent)))
    total1 = len(uniq_paths)
    if io_workers is None:
        io_workers = int(os.environ.get("KE_TILEHASH_THREADS", "0")) or min(8, (os.cpu_count() or 4) * 2)
    log.info("TileHash phase1: %d files, threads=%d", total1, io_workers)
    cache: dict[Path, int] = {}
    failure_counts: Counter = Counter()
    failure_samples: dict = {}
    done = 0
    # files are decoded a few at a time and the run is cut as soon as the decoded pixels pass 512 MB (the reference holds at
    # most io_workers decoded images at once and shrinks each to 32x32 immediately; 256 twelve-megapixel files at once
    # would be 9 GB of host memory)
    budget, step = 512 << 20, max(8, 2 * io_workers)
    side = grid * tile

    def count(n_files: int) -> None:
        nonlocal done
        before, done = done, done + n_files
        if tick:
            for mark in range((before // 64 + 1) * 64, done + 1, 64):
                tick(mark, total1, phase=1)
            if done == total1 and done % 64:
                tick(done, total1, phase=1)

    # JPEG / PNG files first, thousands at a time, decoded and shrunk on the GPU; what that route leaves goes through Pillow
    todo = []
    for lo in range(0, total1, 4096):
        if is_cancelled and is_cancelled():
            return []
        batch = uniq_paths[lo:lo + 4096]
        thumbs = _thumbnails_decoded_on_gpu(batch, side, device)
        if thumbs:
            keys = list(thumbs)
            try:
                words = _native.get_context(device).tile_ahash(np.stack([thumbs[p] for p in keys]), len(keys), grid, tile)
                cache.update({p: int.from_bytes(row.tobytes(), "little") for p, row in zip(keys, words)})
                count(len(keys))
            except (RuntimeError, ValueError):
                thumbs = {}
        todo.extend(p for p in batch if p not in thumbs)
    uniq_paths, total_rest = todo, len(todo)
    start = 0
    while start < total_rest:
        if is_cancelled and is_cancelled():
            return []
        decoded, held = [], 0
        while start < total_rest and held < budget and len(decoded) < 256:
            part = _decode_all(uniq_paths[start:start + step], io_workers)
            start += len(part)
            decoded.extend(part)
            held += sum(a.nbytes for _, a in part if not isinstance(a, Exception))
        good = [(p, a) for p, a in decoded if not isinstance(a, Exception)]
        if good:
            try:
                sigs = tile_ahash_from_arrays([a for _, a in good], grid, tile, device=device)
                cache.update({p: s for (p, _), s in zip(good, sigs)})
            except (RuntimeError, ValueError) as exc:
                decoded = [(p, exc) for p, _ in decoded]
        for p, a in decoded:
            if isinstance(a, Exception):
                key = f"{type(a).__name__}: {a}"
                failure_counts[key] += 1
                failure_samples.setdefault(key, p)
        count(len(decoded))
    if failure_counts:
        log.warning("TileHash phase1 skipped %d file(s) due to errors: %s", sum(failure_counts.values()),
                    _format_failure_summary(failure_counts, failure_samples))
    # --- phase 2: keep the members within max_bits of their keeper
    out = []
    total2 = len(clusters)
    for i, cl in enumerate(clusters, 1):
        if is_cancelled and is_cancelled():
            return []
        keep = next((e for e in cl.files if e.file.file_id == cl.keeper_id), None)
        base = cache.get(_norm_path(keep.file.path)) if keep else None
        if base is not None:
            oks = []
            for e in cl.files:
                sig = cache.get(_norm_path(e.file.path))
                if sig is not None and tile_hamming(base, sig) <= max_bits:
                    oks.append(e)
            if len(oks) >= 2:
                out.append(_rebuild_cluster_like(cl, oks))
        if tick and (i % 16 == 0 or i == total2):
            tick(i, total2, phase=2)
    return out


def _load_small_gray(path, size: int = 128, *, device: int = 0) -> np.ndarray:
    return _thumbnails([_decode(path)], size, device)[0]


def _mae01(a: np.ndarray, b: np.ndarray, *, device: int = 0) -> float:
    """0..1 normalised mean absolute error of two equally sized u8 arrays (sum on the GPU)."""
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    sad = _native.get_context(device).sad_pairs(np.stack([a.reshape(-1), b.reshape(-1)]), 2, a.size, [0], [1])
    return float(int(sad[0]) / a.size / 255.0)


def refine_by_pixels_parallel(clusters: Sequence, mae_thr: float = 0.006, thumb_size: int = 128,
                              workers: Optional[int] = None, tick: Optional[Callable[[int, int], None]] = None,
                              is_cancelled: Optional[Callable[[], bool]] = None, *, device: int = 0) -> list:
    total = len(clusters)
    worker_count = workers if workers is not None else min(8, (os.cpu_count() or 4))
    keeper_failures: Counter = Counter()
    keeper_samples: dict = {}
    entry_failures: Counter = Counter()
    entry_samples: dict = {}
    out = []
    ctx = _native.get_context(device)
    pixels = thumb_size * thumb_size
    # the JPEG / PNG members of all clusters first, thousands at a time: read, decoded and shrunk on the GPU (the per-cluster
    # loop below then only decodes what that route left out)
    ready: dict = {}
    distinct = list(dict.fromkeys(_norm_path(e.file.path) for cl in clusters for e in cl.files))
    for lo in range(0, len(distinct), 4096):
        if is_cancelled and is_cancelled():
            return []
        ready.update(_thumbnails_decoded_on_gpu(distinct[lo:lo + 4096], thumb_size, device))
    for done, cl in enumerate(clusters, 1):
        if is_cancelled and is_cancelled():
            return []
        keep = next((e for e in cl.files if e.file.file_id == cl.keeper_id), None)
        if keep is not None:
            paths = [keep.file.path] + [e.file.path for e in cl.files]
            have = [ready.get(_norm_path(p)) for p in paths]
            rest = dict(_decode_all([p for p, t in zip(paths, have) if t is None], worker_count))
            decoded = [(p, t if t is not None else rest[p]) for p, t in zip(paths, have)]      # thumbnail, array or Exception
            if isinstance(decoded[0][1], Exception):
                key = f"{type(decoded[0][1]).__name__}: {decoded[0][1]}"
                keeper_failures[key] += 1
                keeper_samples.setdefault(key, keep.file.path)
            else:
                good = [k for k, (_, a) in enumerate(decoded) if not isinstance(a, Exception)]
                for k, (p, a) in enumerate(decoded[1:], 1):
                    if isinstance(a, Exception):
                        key = f"{type(a).__name__}: {a}"
                        entry_failures[key] += 1
                        entry_samples.setdefault(key, p)
                try:
                    todo = [k for k in good if have[k] is None]
                    made = dict(zip(todo, _thumbnails([decoded[k][1] for k in todo], thumb_size, device))) if todo else {}
                    thumbs = np.stack([have[k] if have[k] is not None else made[k] for k in good])
                    slot = {k: j for j, k in enumerate(good)}
                    members = [k for k in good if k >= 1]
                    sad = ctx.sad_pairs(thumbs.reshape(len(good), -1), len(good), pixels, [slot[k] for k in members],
                                        [0] * len(members))
                except (RuntimeError, ValueError) as exc:          # a device error costs this cluster, not the run
                    key = f"{type(exc).__name__}: {exc}"
                    keeper_failures[key] += 1
                    keeper_samples.setdefault(key, keep.file.path)
                else:
                    oks = [cl.files[k - 1] for k, s in zip(members, sad.tolist())
                           if float(float(int(s)) / pixels / 255.0) <= mae_thr]
                    if len(oks) >= 2:
                        out.append(_rebuild_cluster_like(cl, oks))
        if tick and (done % 16 == 0 or done == total):
            tick(done, total)
    if keeper_failures:
        log.warning("Pixel MAE skipped %d cluster(s) due to keeper load errors: %s", sum(keeper_failures.values()),
                    _format_failure_summary(keeper_failures, keeper_samples))
    if entry_failures:
        log.warning("Pixel MAE excluded %d file(s) due to image load errors: %s", sum(entry_failures.values()),
                    _format_failure_summary(entry_failures, entry_samples))
    return out


__all__ = ["tile_ahash_bits", "tile_hamming", "tile_ahash_from_arrays", "refine_by_tilehash_parallel",
           "refine_by_pixels_parallel"]
