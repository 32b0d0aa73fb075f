"""SSIM refinement on the MI355X: drop-in for the reference's ``dup.refine``.

Mirrors src/dup/refine.py: ``refine_pair``, ``RefinementThresholds``, ``RefinedMatch``.  The
SSIM value comes from csrc/ke_ssim.hip (skimage's 7x7 uniform-window SSIM, float32).  ORB
(src/dup/refine.py:55-68) is outside this build's scope: ``orb_ratio`` is
always None and the decision is ``ssim >= thresholds.ssim`` alone.
"""
from __future__ import annotations

import importlib
import logging
import os
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _native
from .image_io import MAX_SIDE, load_rgb

_phash = importlib.import_module(".phash", __package__)   # the package also exports a function named phash

logger = logging.getLogger(__name__)


@dataclass(frozen=True)
class RefinementThresholds:
    ssim: float = 0.9
    orb: float = 0.15


@dataclass(frozen=True)
class RefinedMatch:
    file_id_a: int
    file_id_b: int
    ssim: Optional[float]
    orb_ratio: Optional[float]
    is_duplicate: bool
    reason: str


def _common_size(img_a, img_b) -> tuple[int, int]:
    """(min width, min height) of the two images (src/dup/refine.py:45-47)."""
    size = (min(img_a.width, img_b.width), min(img_a.height, img_b.height))
    if size[0] == 0 or size[1] == 0:
        size = (max(img_a.width, img_b.width), max(img_a.height, img_b.height))
    return size


def compute_ssim(img_a, img_b, *, device: int = 0) -> float:
    """SSIM of two PIL images of any two sizes (src/dup/refine.py:44-52).

    Both are brought to the common size the way the reference does -- ``ImageOps.fit(image.convert("L"), size,
    BICUBIC)``: centre crop to the aspect ratio, Pillow's fixed-point bicubic resize -- by ``ke_fit_luma_uniform``;
    the two luma planes stay in device memory for the SSIM kernel.
    """
    ctx = _native.get_context(device)
    w, h = _common_size(img_a, img_b)
    if w < 7 or h < 7:
        raise ValueError("win_size exceeds image extent")  # what skimage raises for tiny images
    planes = ctx.malloc(2 * w * h)
    try:
        for k, image in enumerate((img_a, img_b)):
            arr = _phash.image_to_array(image)
            channels = 1 if arr.ndim == 2 else arr.shape[2]
            ctx.fit_luma_uniform(arr, 1, arr.shape[1], arr.shape[0], channels, w, h, _native.FILTER_BICUBIC,
                                 out=planes + k * w * h)
        return float(ctx.ssim_pairs_uniform(planes, 2, w, h, 1, [0], [1])[0])
    finally:
        ctx.free(planes)


def ssim_pairs(images: np.ndarray, pair_a: Sequence[int], pair_b: Sequence[int], *, device: int = 0) -> np.ndarray:
    """Batched form: images (n,h,w[,c]) u8 of one size, SSIM for every (pair_a[k], pair_b[k])."""
    images = np.ascontiguousarray(images, dtype=np.uint8)
    n, h, w = images.shape[:3]
    ch = 1 if images.ndim == 3 else images.shape[3]
    return _native.get_context(device).ssim_pairs_uniform(images, n, w, h, ch, pair_a, pair_b)


def _decide(file_id_a: int, file_id_b: int, ssim_value: Optional[float], errors: list, cfg: RefinementThresholds) -> RefinedMatch:
    """The decision and ``reason`` text of src/dup/refine.py:100-117 (ORB left out)."""
    reasons: list[str] = []
    if ssim_value is not None and ssim_value >= cfg.ssim:
        reasons.append(f"ssim>={cfg.ssim}")
    reason = ", ".join(reasons or errors) if (reasons or errors) else "below thresholds"
    return RefinedMatch(file_id_a, file_id_b, ssim_value, None, bool(reasons), reason)


def refine_pairs(pairs: Sequence[tuple], *, thresholds: Optional[RefinementThresholds] = None, device: int = 0,
                 io_workers: int = 8, max_decoded_bytes: int = 1 << 30, stats: Optional[dict] = None) -> list:
    """``refine_pair`` for a list of ``(file_id_a, file_id_b, path_a, path_b)``: the same ``RefinedMatch`` (or ``None``)
    per pair, in input order -- but every file is decoded once however many pairs share it, and the kernels see groups
    instead of single pairs: one ``ke_fit_luma_uniform`` launch per (source size, channels, common size) and one
    ``ke_ssim_pairs_uniform`` launch per common size.  The reference's loop decodes both files of every pair again and
    runs the metric pair by pair (src/dup/refine.py:82-98).

    Pairs are worked off in runs whose decoded files stay below ``max_decoded_bytes``.  ``stats`` (optional dict)
    receives the launch and decode counts.
    """
    from concurrent.futures import ThreadPoolExecutor

    cfg = thresholds or RefinementThresholds()
    ctx = _native.get_context(device)
    out: list = [None] * len(pairs)
    count = {"decodes": 0, "gpu_decodes": 0, "fit_launches": 0, "ssim_launches": 0, "pairs": len(pairs)}

    gpu_kinds = {}
    if os.environ.get("KE_GPU_REFINE_DECODE", "1") != "0":
        if os.environ.get("KE_GPU_JPEG", "1") != "0":
            gpu_kinds["jpeg"] = (".jpg", ".jpeg", ".jpe", ".jfif")
        if os.environ.get("KE_GPU_PNG", "1") != "0":
            gpu_kinds["png"] = (".png", ".apng")
        if os.environ.get("KE_GPU_BMP", "1") != "0":
            gpu_kinds["bmp"] = (".bmp",)
        if os.environ.get("KE_GPU_TIFF", "1") != "0":
            gpu_kinds["tiff"] = (".tif", ".tiff")

    def decode_on_gpu(need: list, placed: dict, buffers: list) -> None:
        """JPEG / PNG / BMP / TIFF files whose pixels the reference's loader would hand over exactly as Image.open yields them -- RGB, no
        EXIF orientation to apply, nothing to shrink (src/utils/image_io.py:107-138 are all no-ops then) -- are decoded on the
        GPU and stay there: placed[path] = (device address, width, height).  Everything else is left for Pillow."""
        for kind, suffixes in gpu_kinds.items():
            paths = [p for p in need if p.lower().endswith(suffixes)]
            if not paths:
                continue
            dev, off, w, h, c, st, flags = ctx.decode_files_owned(paths, kind)
            if not dev:
                continue
            buffers.append(dev)
            fits = (st == 0) & (np.maximum(w, h) <= MAX_SIDE)          # larger ones are shrunk by the loader (draft mode + LANCZOS)
            ok = fits & (c == 3) & ((flags & 3) == 0)
            for p, good, o, ww, hh in zip(paths, ok.tolist(), off.tolist(), w.tolist(), h.tolist()):
                if good:
                    placed[p] = (dev + int(o), ww, hh)
            count["gpu_decodes"] += int(ok.sum())
            # what the loader does to the rest (src/utils/image_io.py:116-131) happens on the device as well: the EXIF
            # orientation of a JPEG file applied (every camera writes one), an RGBA PNG composited over white
            orient = (flags >> 8) & 15
            turn = fits & (c == 3) & ((flags & 3) == 1) & (orient >= 2) & (orient <= 8) if kind == "jpeg" else np.zeros(len(paths), bool)
            over = fits & (c == 4) & ((flags & 3) == 0) if kind in ("png", "bmp", "tiff") else np.zeros(len(paths), bool)
            fix = np.nonzero(turn | over)[0]
            if len(fix):
                dev2, off2, w2, h2 = ctx.normalise_rgb(dev, off[fix], w[fix], h[fix], c[fix], np.where(turn[fix], orient[fix], 1))
                buffers.append(dev2)
                for k, o2, ww, hh in zip(fix.tolist(), off2.tolist(), w2.tolist(), h2.tolist()):
                    placed[paths[k]] = (dev2 + int(o2), ww, hh)
                count["gpu_decodes"] += len(fix)
                count["gpu_normalised"] = count.get("gpu_normalised", 0) + len(fix)
            # a side over MAX_SIDE: the loader's img.thumbnail((MAX_SIDE, MAX_SIDE), LANCZOS) (src/utils/image_io.py:122-124), after
            # the turn.  Below twice that size neither JPEG draft mode nor thumbnail's reducing_gap changes what is resampled
            # (both act from a factor of two on); larger files and those with an alpha channel stay with the loader.
            longest = np.maximum(w, h)
            big = (st == 0) & (c == 3) & (longest > MAX_SIDE) & (longest < 2 * MAX_SIDE - 256)
            plain = big & ((flags & 3) == 0)
            turned = big & ((flags & 3) == 1) & (orient >= 2) & (orient <= 8) if kind == "jpeg" else np.zeros(len(paths), bool)
            for k in np.nonzero(plain | turned)[0].tolist():
                addr, ww, hh, held = dev + int(off[k]), int(w[k]), int(h[k]), None
                try:
                    if turned[k]:
                        held, o2, w2, h2 = ctx.normalise_rgb(dev, off[k:k + 1], w[k:k + 1], h[k:k + 1], c[k:k + 1], orient[k:k + 1])
                        addr, ww, hh = held + int(o2[0]), int(w2[0]), int(h2[0])
                    small, sw, sh = ctx.thumbnail_rgb(addr, ww, hh, MAX_SIDE)
                finally:
                    if held is not None:
                        ctx.free(held)
                buffers.append(small)
                placed[paths[k]] = (small, sw, sh)
                count["gpu_decodes"] += 1
                count["gpu_shrunk"] = count.get("gpu_shrunk", 0) + 1

    def run(chunk: list, placed: dict, arrays: dict, buffers: list) -> None:
        """One ``ke_ssim_pairs`` call for the pairs of this run (the library groups them: one fit launch per (source size,
        common size), one SSIM launch per common size).  Images decoded by Pillow are uploaded next to the ones decoded on
        the GPU."""
        host = [(p, a) for p, a in arrays.items() if a is not None]
        if host:
            sizes = [(a.size + 15) & ~15 for _, a in host]
            flat = np.zeros(sum(sizes), np.uint8)
            at = 0
            dev = ctx.malloc(len(flat) + 64)
            buffers.append(dev)
            for (p, a), sz in zip(host, sizes):
                flat[at:at + a.size] = a.reshape(-1)
                placed[p] = (dev + at, a.shape[1], a.shape[0])
                at += sz
            ctx.memcpy(dev, flat, len(flat))
        live = []
        for k in chunk:
            fid_a, fid_b, pa, pb = pairs[k]
            if str(pa) not in placed or str(pb) not in placed:
                out[k] = None                                      # unreadable file: src/dup/refine.py:82-85
            else:
                live.append(k)
        if not live:
            return
        paths = list(placed)
        index = {p: i for i, p in enumerate(paths)}
        scores, status = ctx.ssim_pairs_on_device([placed[p][0] for p in paths], [placed[p][1] for p in paths], [placed[p][2] for p in paths],
                                                  3, [index[str(pairs[k][2])] for k in live], [index[str(pairs[k][3])] for k in live])
        common, fits = set(), set()
        for k, sc, st in zip(live, scores.tolist(), status.tolist()):
            fid_a, fid_b, pa, pb = pairs[k]
            if st != 0:                                            # skimage raises for images smaller than its window
                logger.warning("SSIM refinement failed for %s and %s: win_size exceeds image extent", pa, pb)
                out[k] = _decide(fid_a, fid_b, None, ["ssim unavailable"], cfg)
                continue
            out[k] = _decide(fid_a, fid_b, float(sc), [], cfg)
            (_, wa, ha), (_, wb, hb) = placed[str(pa)], placed[str(pb)]
            size = (min(wa, wb), min(ha, hb))
            common.add(size)
            fits.update({((ha, wa), size), ((hb, wb), size)})
        count["fit_launches"] += len(fits)
        count["ssim_launches"] += len(common)

    def load(p: str):
        img = load_rgb(p)
        if img is None:
            return None
        arr = _phash.image_to_array(img)
        return arr if arr.ndim == 3 and arr.shape[2] == 3 else None

    with ThreadPoolExecutor(max_workers=max(1, io_workers)) as pool:
        start = 0
        while start < len(pairs):
            need, size_est, stop = [], 0, start                    # grow the run until the decode budget is reached
            seen: set = set()
            while stop < len(pairs) and (size_est < max_decoded_bytes or stop == start):
                for p in (str(pairs[stop][2]), str(pairs[stop][3])):
                    if p not in seen:
                        seen.add(p)
                        need.append(p)
                        try:
                            size_est += 48 * os.path.getsize(p)    # rough decoded size of a compressed file; only paces the runs
                        except OSError:
                            pass
                stop += 1
            placed: dict = {}
            buffers: list = []
            try:
                decode_on_gpu(need, placed, buffers)
                rest = [p for p in need if p not in placed]
                arrays = dict(zip(rest, pool.map(load, rest)))
                count["decodes"] += len(need)
                run(list(range(start, stop)), placed, arrays, buffers)
            finally:
                for dev in buffers:
                    ctx.free(dev)
            start = stop
    if stats is not None:
        stats.update(count)
    return out


def refine_pair(file_id_a: int, file_id_b: int, path_a, path_b, *, thresholds: Optional[RefinementThresholds] = None,
                device: int = 0) -> Optional[RefinedMatch]:
    image_a, image_b = load_rgb(path_a), load_rgb(path_b)
    if image_a is None or image_b is None:
        return None
    cfg = thresholds or RefinementThresholds()
    ssim_value: Optional[float] = None
    errors: list[str] = []
    try:
        ssim_value = compute_ssim(image_a, image_b, device=device)
    except Exception as exc:
        logger.warning("SSIM refinement failed for %s and %s: %s", path_a, path_b, exc)
        errors.append("ssim unavailable")
    return _decide(file_id_a, file_id_b, ssim_value, errors, cfg)


__all__ = ["RefinementThresholds", "RefinedMatch", "refine_pair", "refine_pairs", "compute_ssim", "ssim_pairs"]
