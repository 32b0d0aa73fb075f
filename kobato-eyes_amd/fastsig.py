"""Batch signature fill on the MI355X: drop-in for the reference's ``core.fastsig``.

Public names and contracts follow src/core/fastsig.py -- ``fast_fill_missing_signatures`` (:102-126),
``compute_signatures_mp`` (:65-99), ``bulk_upsert_signatures`` (:48-62), ``_to_signed64`` (:19-21): results come back in
input order with failed files left out (:36-37), progress fires every 200 files and at the end (:93), a true
``cancel_fn`` stops the run and hands back what is finished (:86-90), values are stored as signed 64-bit, and the
optional "unsafe fast" PRAGMAs are the reference's (:40-45).

The machinery is different: where the reference fans single files out to a spawn-context process pool, this is a
two-stage pipeline -- Pillow decodes chunk k+1 on a thread pool (it releases the GIL) straight into a page-locked
staging buffer while chunk k is copied to the GPU and hashed (``ke_stage_*``, include/keyes.h).
"""
from __future__ import annotations

import atexit
import importlib
import mmap
import multiprocessing
import os
import sqlite3
import threading
from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor
from concurrent.futures.process import BrokenProcessPool
from itertools import chain
from pathlib import Path
from typing import Callable, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from . import _native

_phash = importlib.import_module(".phash", __package__)   # the package also exports a function named phash

U64MASK = (1 << 64) - 1
PROGRESS_STRIDE = 200
FAST_PRAGMAS = ("journal_mode=WAL", "synchronous=OFF", "temp_store=MEMORY", "mmap_size=30000000000")
UPSERT_SQL = ("INSERT INTO signatures (file_id, phash_u64, dhash_u64) VALUES (?, ?, ?) "
              "ON CONFLICT(file_id) DO UPDATE SET phash_u64 = excluded.phash_u64, dhash_u64 = excluded.dhash_u64")

Task = Tuple[int, str]
Row = Tuple[int, int, int]


def _usable_cpus() -> int:
    """CPUs this process may actually use: the affinity mask and the cgroup's quota, not the machine's core count (the
    reference's `os.cpu_count() - 1` default starts 255 decode threads on a 16-CPU share of a large host)."""
    n = os.cpu_count() or 4
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _to_signed64(x: int) -> int:
    v = int(x) & U64MASK
    return v - (1 << 64) if v >> 63 else v


_strict_lock = threading.Lock()
_strict_state = {"active": 0, "saved": False}


class _StrictPillow:
    """While any decode of this module is under way Pillow's process-wide ``ImageFile.LOAD_TRUNCATED_IMAGES`` is off; the last
    one out puts back what it found (unless somebody else has switched it on in the meantime: then it stays on).

    The reference's worker is a freshly spawned process (src/core/fastsig.py:81-85): the switch is at its default there, so a
    truncated file raises and is dropped (:36-37).  Here the decoders may be threads of the calling process, where
    ``safe_load_image`` (src/utils/image_io.py:60-138, and this package's restatement) leaves the switch ON for good after its
    first call -- a truncated file would then be padded and hashed.  The decodes themselves run side by side; only the two
    counter updates take the lock."""

    def __enter__(self):
        from PIL import ImageFile

        with _strict_lock:
            if _strict_state["active"] == 0:
                _strict_state["saved"] = ImageFile.LOAD_TRUNCATED_IMAGES
            ImageFile.LOAD_TRUNCATED_IMAGES = False
            _strict_state["active"] += 1

    def __exit__(self, *exc):
        from PIL import ImageFile

        with _strict_lock:
            _strict_state["active"] -= 1
            if _strict_state["active"] == 0 and not ImageFile.LOAD_TRUNCATED_IMAGES:
                ImageFile.LOAD_TRUNCATED_IMAGES = _strict_state["saved"]
        return False


def _read_pixels(path_text: str, room: Optional[int] = None):
    """Pixels of one file, or None for anything that cannot be opened (speed first: no reason kept).  With ``room``: a file
    whose pixels would not fit is not decoded at all -- its decoded size (an int) comes back instead (the header says it)."""
    try:
        from PIL import Image

        path = Path(path_text)
        if not path.is_file():
            return None
        with _StrictPillow(), Image.open(path) as im:
            if room is not None:
                need = _decoded_bytes(im)
                if need > room:
                    return int(need)
            return _phash.image_to_array(im)
    except Exception:
        return None


JPEG_SUFFIXES = (".jpg", ".jpeg", ".jpe", ".jfif")
PNG_SUFFIXES = (".png", ".apng")
BMP_SUFFIXES = (".bmp",)
GIF_SUFFIXES = (".gif",)
TIFF_SUFFIXES = (".tif", ".tiff")
GPU_KINDS = ("jpeg", "png", "bmp", "gif", "tiff")           # the order in which a batch's files lie in the read-ahead buffer


def _read_bytes(path_text: str):
    """The file as it is (JPEG route: the GPU decodes it), or None."""
    try:
        with open(path_text, "rb") as fh:
            return fh.read()
    except OSError:
        return None


# ---- decoder processes.  Pillow's per-file work in Python (opening, plugin dispatch, conversions) holds the interpreter lock,
# so decode THREADS stop scaling at two or three cores; the reference runs a spawn-context process pool for that reason
# (src/core/fastsig.py:83-85).  Here the processes only decode: they write the pixels straight into the staging buffers, which
# for this purpose are files in /dev/shm mapped by both sides and page-locked by the library (ke_stage_create_shared), and hand
# back where each image went.  The parent never touches the pixels.
_worker_maps: dict = {}


def _decode_into_shared(job):
    """Worker: (path of the mapped buffer, region start, region length, [file paths]) -> per file (offset, w, h, channels),
    "spill" when the region is full, None when the file cannot be read."""
    shm_path, start, length, paths = job
    view = _worker_maps.get(shm_path)
    if view is None:
        fd = os.open(shm_path, os.O_RDWR)
        try:
            view = np.frombuffer(mmap.mmap(fd, 0), np.uint8)
        finally:
            os.close(fd)
        _worker_maps[shm_path] = view
    out, cursor, end = [], start, start + length
    for p in paths:
        off = (cursor + 15) & ~15
        arr = _read_pixels(p, room=end - off)
        if arr is None or (not isinstance(arr, int) and arr.size == 0):
            out.append(None)
            continue
        if isinstance(arr, int):                      # does not fit what is left of the region: not decoded, its size handed back
            out.append(("spill", arr))
            continue
        view[off:off + arr.size] = arr.reshape(-1)
        cursor = off + int(arr.size)
        h, w = arr.shape[:2]
        out.append((off, w, h, 1 if arr.ndim == 2 else arr.shape[2]))
    return out


def _decoded_bytes(image) -> int:
    """Bytes image_to_array will produce for an opened (not yet decoded) Pillow image: L / RGB / RGBA / RGBX keep their bands,
    every other mode goes through convert("L")."""
    w, h = image.size
    return w * h * (len(image.mode) if image.mode in ("RGB", "RGBA", "RGBX") else 1)


def _probe_decoded_bytes(paths: Sequence[str]) -> int:
    """Largest decoded size among a few files, from their headers alone (0 when none opens)."""
    from PIL import Image

    largest = 0
    for p in paths:
        try:
            with Image.open(p) as im:
                largest = max(largest, _decoded_bytes(im))
        except Exception:
            pass
    return largest


_pools: dict = {}


def _process_pool(workers: int) -> ProcessPoolExecutor:
    """One spawn-context pool per worker count, kept for the life of the process (starting 15 interpreters costs about a
    second; a library scan calls the batch hasher many times)."""
    pool = _pools.get(workers)
    if pool is None:
        pool = _pools[workers] = ProcessPoolExecutor(max_workers=workers, mp_context=multiprocessing.get_context("spawn"))
    return pool


@atexit.register
def _stop_pools() -> None:
    for pool in _pools.values():
        pool.shutdown(wait=False, cancel_futures=True)
    _pools.clear()


class _SharedBuffers:
    """Two page-aligned buffers that other processes can map: files in /dev/shm, unlinked when the object goes."""

    def __init__(self, nbytes: int, count: int = 2) -> None:
        import fcntl

        self.paths, self.maps, self._fds = [], [], []
        # what a process that died left behind.  A live owner holds an flock on its files for as long as it runs, so a file
        # whose lock can be taken has no owner -- whatever PID namespace that owner lived in (containers that share /dev/shm
        # do not share /proc, so "no such PID here" proves nothing).
        for stale in os.listdir("/dev/shm"):
            if not stale.startswith("ke_stage_"):
                continue
            full = os.path.join("/dev/shm", stale)
            try:
                fd = os.open(full, os.O_RDWR)
            except OSError:
                continue
            try:
                fcntl.flock(fd, fcntl.LOCK_EX | fcntl.LOCK_NB)
                os.unlink(full)
            except OSError:
                pass
            finally:
                os.close(fd)
        for k in range(count):
            path = f"/dev/shm/ke_stage_{os.getpid()}_{id(self):x}_{k}"
            fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
            try:
                fcntl.flock(fd, fcntl.LOCK_EX | fcntl.LOCK_NB)
                os.ftruncate(fd, nbytes)
                self.maps.append(mmap.mmap(fd, nbytes))
            except BaseException:
                os.close(fd)
                os.unlink(path)
                raise
            self._fds.append(fd)                             # stays open: the lock lives as long as this descriptor
            self.paths.append(path)
        self.addresses = [np.frombuffer(m, np.uint8).ctypes.data for m in self.maps]

    def close(self) -> None:
        for path in self.paths:
            try:
                os.unlink(path)
            except OSError:
                pass
        self.paths = []
        for fd in self._fds:
            try:
                os.close(fd)
            except OSError:
                pass
        self._fds = []

    def __del__(self) -> None:  # pragma: no cover
        self.close()


class _GpuStage:
    """The context's pinned staging buffers (``ke_stage_*``) behind the four calls the pipeline needs; tests of the host
    logic put a stand-in here (``_make_stage``)."""

    def __init__(self, device: int, stage_bytes: int, max_images: int) -> None:
        self.device = device
        self.ctx = _native.get_context(device)                  # GPU decode calls, read-ahead buffers
        # the staging route lives on a context of its own (own stream, lock, scratch): the decoder processes' buffers are
        # copied and hashed while a GPU decode call -- seconds for large PNG streams -- holds the main one
        self.sctx = _native.get_context(device, "staging")
        stage_bytes = (int(stage_bytes) + 4095) & ~4095
        shared = os.environ.get("KE_DECODE_PROCESSES", "1") != "0" and os.path.isdir("/dev/shm")
        geom = (stage_bytes, max_images, shared)
        if getattr(self.sctx, "_stage_geom", None) != geom:
            old = getattr(self.sctx, "_stage_shared", None)    # stays mapped until the library has let go of it
            buffers = None
            if shared:
                try:
                    buffers = _SharedBuffers(stage_bytes)
                    self.sctx.stage_create_shared(buffers.addresses, stage_bytes, max_images)
                except (OSError, RuntimeError, ValueError):
                    if buffers is not None:
                        buffers.close()
                    buffers, shared = None, False
            if not shared:
                self.sctx.stage_create(stage_bytes, max_images, 2)
            self.sctx._stage_shared = buffers
            self.sctx._stage_geom = (stage_bytes, max_images, shared)
            if old is not None:
                old.close()
        buffers = getattr(self.sctx, "_stage_shared", None)
        self.shared_paths = buffers.paths if buffers is not None else None     # slot k <-> shared_paths[k]
        self.stage_bytes = stage_bytes

    def acquire(self):
        return self.sctx.stage_acquire()

    def submit(self, slot, offsets, widths, heights, channels):
        return self.sctx.stage_submit_hash(slot, offsets, widths, heights, channels, want_dhash=True)

    def wait(self, slot: int) -> None:
        self.sctx.stage_wait(slot)

    def jpeg_hash(self, blobs, kind: str = "jpeg"):
        """(phash, dhash, status) of JPEG / PNG files decoded on the GPU (``ke_jpeg_decode`` / ``ke_png_decode`` ->
        ``ke_hash_images``); status != 0: the decoder leaves the file to Pillow."""
        return self.ctx.jpeg_hash(blobs, want_dhash=True, kind=kind)

    def hash_files(self, paths, kind: str = "jpeg"):
        """jpeg_hash for files on disk: the library reads them (host threads, page-locked memory), no bytes objects."""
        return self.ctx.hash_files(paths, want_dhash=True, kind=kind)

    def read_ahead(self, paths, spans=()):
        """The files read into one of the context's read-ahead buffers, headers parsed (any thread; None if none is free)."""
        return self.ctx.read_files_ahead(paths, spans)

    def hash_ahead(self, held, lo: int, hi: int, kind: str = "jpeg", skip=None):
        """hash_files for files lo..hi of what read_ahead returned; ``skip``: a mask of files to leave alone (status 1)."""
        return self.ctx.jpeg_hash(None, want_dhash=True, kind=kind, ahead=(held, lo, hi), skip=skip)

    def hash_one(self, arr):
        """(phash, dhash) or None for an image that did not fit a staging buffer."""
        ph, dh, ok = _phash.hash_batch([arr], want_dhash=True, device=self.device)
        return (int(ph[0]), int(dh[0])) if ok[0] else None


_make_stage = _GpuStage


class _Pipeline:
    """decode (threads) -> pinned staging buffer -> H2D + hash (GPU), without a host-side copy in between; JPEG and PNG files
    decoded on the GPU instead.

    Files are taken ``KE_GPU_BATCH`` (default 32768) at a time (a batch whose compressed bytes exceed ``KE_PACK_LIMIT_BYTES`` or
    whose pixels exceed ``KE_DECODE_LIMIT_BYTES`` is halved by the context until it fits).  Within such a batch:

    * JPEG and PNG files (by suffix) skip Pillow altogether: the library's host threads read the next batch's files into one
      of the context's two page-locked read-ahead buffers (``Context.read_files_ahead``; ``KE_READ_AHEAD=0``: inside the decode
      call instead) while this one is on the GPU, and ONE ``ke_jpeg_decode`` / ``ke_png_decode`` call per kind decodes them all,
      pixel-identical to ``Image.open`` for the files the decoders take; the hash kernels run on the decoded pixels where they lie.  The batch is large because those decoders are one thread per image: a wave of 64 files takes as long as
      thousands of waves side by side.  ``KE_GPU_JPEG=0`` / ``KE_GPU_PNG=0`` turn the routes off.
    * every other file, and what the GPU decoders refuse (progressive, CMYK, palette, 16-bit, damaged ...), is decoded by
      Pillow on the thread pool ``chunk`` files at a time: the threads of chunk k write their pixels straight into one of the
      context's two page-locked staging buffers (``ke_stage_acquire``; a bump allocator hands out 16-byte aligned regions);
      ``ke_stage_submit_hash`` then enqueues the copy and the kernels and returns, so chunk k+1 is decoded into the other
      buffer while chunk k crosses PCIe and is hashed.  An image that does not fit what is left of a buffer is hashed on its
      own through ``ke_hash_images``.
    """

    def __init__(self, tasks: Sequence[Task], workers: int, chunk: int, device: int,
                 cancel_fn: Optional[Callable[[], bool]] = None) -> None:
        self.tasks, self.chunk, self.device = tasks, max(1, int(chunk)), device
        self.batch = max(self.chunk, int(os.environ.get("KE_GPU_BATCH", "32768")))
        self.cancel_fn = cancel_fn
        self.stopped = False
        if cancel_fn is not None:
            # somebody may want to stop: the reference asks after every file (src/core/fastsig.py:86-90).  Here the question is
            # put between the chunks / rounds of the Pillow route and before every GPU decode call, and a batch is small
            # enough that even a batch of slow files is abandoned within a chunk's time
            self.batch = max(self.chunk, min(self.batch, int(os.environ.get("KE_GPU_BATCH_CANCELLABLE", "4096"))))
        self.workers = max(1, workers)
        self.pool = ThreadPoolExecutor(max_workers=self.workers)
        self.stage = _make_stage(device, int(os.environ.get("KE_STAGE_BYTES", str(256 << 20))), max(self.chunk, 4096))
        self.process_min = int(os.environ.get("KE_DECODE_PROCESS_MIN", "256"))
        self.bytes_per_image = 1 << 20                     # running estimate of a decoded image, for sizing the workers' jobs
        # per-file bookkeeping is array work: a library scan is hundreds of thousands of files, and a microsecond of
        # interpreter per file and step was a third of the seam's wall clock
        self.paths = [str(t[1]) for t in tasks]
        self.fids = np.fromiter((int(t[0]) for t in tasks), np.int64, len(tasks))

    def cancelled(self) -> bool:
        if not self.stopped and self.cancel_fn is not None:
            try:
                self.stopped = bool(self.cancel_fn())
            except Exception:
                self.stopped = False
        return self.stopped

    def _decode_into(self, path_text: str, view, alloc: dict):
        """One file -> (offset, width, height, channels) inside the staging buffer, ("spill", array) when it does not
        fit, None when it cannot be read."""
        arr = _read_pixels(path_text)
        if arr is None or arr.size == 0:
            return None
        nbytes = int(arr.size)
        with alloc["lock"]:
            off = (alloc["cursor"] + 15) & ~15
            if off + nbytes > len(view):
                return ("spill", arr)
            alloc["cursor"] = off + nbytes
        view[off:off + nbytes] = arr.reshape(-1)                  # the only copy between Pillow and the GPU (GIL released)
        h, w = arr.shape[:2]
        return (off, w, h, 1 if arr.ndim == 2 else arr.shape[2])

    # ---- the GPU decoders' share of a batch
    def _start_reads(self, start: int) -> dict:
        """The JPEG / PNG / BMP / GIF / TIFF files of the batch that begins at ``start``, on their way into memory while the batch before is
        on the GPU.  Runs on a pool thread (the classification is a pass of the interpreter over the batch, the reading is the
        library's): {"jpeg" / "png" / "bmp" / "gif" / "tiff": positions (arrays, ascending), "blobs": position -> future of the file's bytes for
        a stage without ``hash_files``, "ahead": the context's FilesAhead holding [JPEG | PNG | BMP | GIF | TIFF files] or None}."""
        gpu_jpeg = os.environ.get("KE_GPU_JPEG", "1") != "0"
        gpu_png = os.environ.get("KE_GPU_PNG", "1") != "0"
        gpu_bmp = os.environ.get("KE_GPU_BMP", "1") != "0"
        gpu_gif = os.environ.get("KE_GPU_GIF", "1") != "0"
        gpu_tiff = os.environ.get("KE_GPU_TIFF", "1") != "0"
        by_path = hasattr(self.stage, "hash_files")          # the library reads the files itself, into page-locked memory
        stop = min(start + self.batch, len(self.tasks))
        tails = [p[-5:].lower() for p in self.paths[start:stop]]
        kind = np.fromiter((1 if t.endswith(JPEG_SUFFIXES) else 2 if t.endswith(PNG_SUFFIXES) else 3 if t.endswith(BMP_SUFFIXES) else 4 if t.endswith(GIF_SUFFIXES) else 5 if t.endswith(TIFF_SUFFIXES) else 0
                            for t in tails), np.int8, stop - start)
        jpeg = start + np.nonzero(kind == 1)[0] if gpu_jpeg else np.zeros(0, np.int64)
        png = start + np.nonzero(kind == 2)[0] if gpu_png else np.zeros(0, np.int64)
        bmp = start + np.nonzero(kind == 3)[0] if gpu_bmp else np.zeros(0, np.int64)
        gif = start + np.nonzero(kind == 4)[0] if gpu_gif else np.zeros(0, np.int64)
        tiff = start + np.nonzero(kind == 5)[0] if gpu_tiff else np.zeros(0, np.int64)
        reads = {"jpeg": jpeg, "png": png, "bmp": bmp, "gif": gif, "tiff": tiff, "blobs": {}, "ahead": None}
        order = np.concatenate([reads[k] for k in GPU_KINDS]).tolist()
        if not by_path:
            reads["blobs"] = {int(k): self.pool.submit(_read_bytes, self.paths[k]) for k in order}
        elif order and hasattr(self.stage, "read_ahead") and os.environ.get("KE_READ_AHEAD", "1") != "0":
            ends = np.cumsum([len(reads[k]) for k in GPU_KINDS]).tolist()
            try:
                reads["ahead"] = self.stage.read_ahead([self.paths[k] for k in order],
                                                       tuple((k, e - len(reads[k]), e) for k, e in zip(GPU_KINDS, ends)))
            except Exception:
                reads["ahead"] = None                      # the decode call reads the files itself
        return reads

    def _png_for_pillow(self, held, lo: int, hi: int, kind: str = "png"):
        """Which PNG (or GIF) files of a batch the decoder processes should take instead of the GPU (a mask over files lo..hi of
        the read-ahead buffer, or None = the GPU takes them all).

        ``ke_png_inflate`` is one lane per stream and a deflate stream is sequential, so a batch takes as long as its longest
        stream whatever its size -- about 0.45 us per symbol, i.e. 0.9 us per compressed byte of a textured image -- while a
        decoder process gets through about 250 MB of decoded pixels per second.  2 048 textured 2048 x 2048 files: 5.1 s on
        the GPU, 6.2 s on 16 host threads; 512 of them: 5.7 s against 1.5 s.  With the sizes known (the files are in memory,
        their headers parsed) the k largest files go to the processes, k minimising the larger of (longest stream left for the
        GPU) and (decoded bytes moved to the processes) in those terms -- the two shares are worked off side by side.  KE_PNG_GPU_US_PER_BYTE / KE_PILLOW_MB_PER_S set the two rates;
        KE_PNG_GPU_US_PER_BYTE=0 sends every PNG file to the GPU, as before.  The LZW walk of ``ke_gif_codes`` has the same shape (one lane per
        stream, ~0.65 us per code = 0.75 us per compressed byte: a 512 x 512 frame is 40 ms whatever the batch; KE_GIF_GPU_US_PER_BYTE)."""
        gpu_rate = float(os.environ.get("KE_PNG_GPU_US_PER_BYTE", "0.9") if kind == "png" else os.environ.get("KE_GIF_GPU_US_PER_BYTE", "0.75")) * 1e-6
        if gpu_rate <= 0.0:
            return None
        known = getattr(held, "probed", {}).get((kind, lo, hi))
        if known is None:
            return None
        w, h, c, st = known
        sizes = np.asarray(held.sizes[lo:hi], np.float64)
        decoded = np.where(st == 0, w.astype(np.float64) * h * c, 0.0)
        n = len(sizes)
        if n == 0:
            return None
        cpu_rate = 1.0 / (max(1, self.workers) * float(os.environ.get("KE_PILLOW_MB_PER_S", "250")) * 1e6)
        order = np.argsort(-np.where(st == 0, sizes, 0.0), kind="stable")
        longest_left = np.concatenate([np.where(st == 0, sizes, 0.0)[order] * gpu_rate + 4e-3, [0.0]])   # + the launches' fixed cost
        moved = np.concatenate([[0.0], np.cumsum(decoded[order]) * cpu_rate])
        if self.workers not in _pools:                          # starting the decoder processes costs about a second and a half, once
            moved[1:] += float(os.environ.get("KE_DECODE_POOL_START_S", "1.5"))
        k = int(np.argmin(np.maximum(longest_left, moved)))     # the two shares run side by side (run_batches)
        if k == 0:
            return None
        mask = np.zeros(n, bool)
        mask[order[:k]] = True
        return mask

    def _decode_on_gpu(self, reads: dict, start: int, ph: np.ndarray, dh: np.ndarray, ok: np.ndarray, skips=None) -> list:
        """Fills ph / dh / ok (indexed by position - start) for the files the GPU decoders take; returns the positions they
        left to Pillow (``skips``: {kind: mask} of the PNG / GIF files the caller has given to the decoder processes already --
        not decoded here, not returned)."""
        skips = skips or {}
        by_path = hasattr(self.stage, "hash_files")
        held = reads["ahead"]
        refused: list = []
        first = 0
        try:
            for kind in GPU_KINDS:
                positions = reads[kind]
                if len(positions) == 0:
                    continue
                lo, first = first, first + len(positions)
                if self.cancelled():
                    continue
                if not by_path:                            # a stage that takes bytes: files that cannot be read stay with Pillow
                    got = [(k, reads["blobs"][k].result()) for k in positions.tolist()]
                    positions = np.array([k for k, b in got if b is not None], np.int64)
                    blobs = [b for _, b in got if b is not None]
                    if not blobs:
                        continue
                try:
                    if held is not None:                   # this kind's files are lo .. first of the buffer
                        skip = skips.get(kind)
                        if skip is not None and skip.all():
                            continue
                        p, d, st = self.stage.hash_ahead(held, lo, first, kind) if skip is None else \
                            self.stage.hash_ahead(held, lo, first, kind, skip=skip)
                    elif by_path:
                        p, d, st = self.stage.hash_files([self.paths[k] for k in positions.tolist()], kind)
                    else:
                        p, d, st = self.stage.jpeg_hash(blobs, kind)
                except (RuntimeError, ValueError, MemoryError):      # e.g. no room on the device for this batch: Pillow decodes it
                    refused.extend(positions.tolist())
                    continue
                good = np.asarray(st) == 0
                at = positions[good] - start
                ph[at] = np.asarray(p, np.uint64).view(np.int64)[good]
                dh[at] = np.asarray(d, np.uint64).view(np.int64)[good]
                ok[at] = True
                left = ~good if skips.get(kind) is None else (~good & ~skips[kind])
                refused.extend(positions[left].tolist())   # outside the GPU decoder: Pillow decodes it, as the reference does
        finally:
            if held is not None:
                held.release()
        return refused

    # ---- the Pillow share: chunks through the two staging buffers
    def _start(self, positions: Sequence[int]):
        slot, view = self.stage.acquire()
        alloc = {"lock": threading.Lock(), "cursor": 0}
        futures = [self.pool.submit(self._decode_into, self.tasks[k][1], view, alloc) for k in positions]
        return slot, futures

    def _submit(self, slot: int, futures):
        decoded = [fut.result() for fut in futures]
        staged = [(k, d) for k, d in enumerate(decoded) if d is not None and d[0] != "spill"]
        spills = [(k, d[1]) for k, d in enumerate(decoded) if d is not None and d[0] == "spill"]
        handle = None
        if staged:
            handle = self.stage.submit(slot, [d[0] for _, d in staged], [d[1] for _, d in staged],
                                       [d[2] for _, d in staged], [d[3] for _, d in staged])
        return [k for k, _ in staged], handle, spills

    def _collect(self, slot: int, positions: Sequence[int], submitted, out: dict) -> None:
        staged_pos, handle, spills = submitted
        if handle is not None:
            self.stage.wait(slot)
            for k, p, d, st in zip(staged_pos, handle["phash"].tolist(), handle["dhash"].tolist(), handle["status"].tolist()):
                if st == 0:
                    out[positions[k]] = (_to_signed64(p), _to_signed64(d))
        for k, arr in spills:                                    # larger than a staging buffer's free space
            sig = self.stage.hash_one(arr)
            if sig is not None:
                out[positions[k]] = (_to_signed64(sig[0]), _to_signed64(sig[1]))

    def _decode_with_processes(self, todo: Sequence[int], out: dict) -> None:
        """The Pillow share on decoder PROCESSES: every round hands each worker a job of a few files and a region of the
        shared staging buffer to write them into; the buffer of round k is copied and hashed while round k+1 decodes."""
        pool = _process_pool(self.workers)
        # Camera-sized files of formats outside the GPU decoders (WebP, TIFF, BMP ...): the staging buffers are sized so that
        # every worker can hold two images of the size the first files announce in their headers (up to 2 GB per buffer); with
        # the default 256 MB a 12-megapixel batch kept three of fifteen decoders busy.
        if self.bytes_per_image == 1 << 20 and todo:
            seen = _probe_decoded_bytes([self.paths[k] for k in todo[:16]])
            if seen:
                self.bytes_per_image = seen
                want = min(2 << 30, 2 * seen * self.workers)
                if want > self.stage.stage_bytes and "KE_STAGE_BYTES" not in os.environ:
                    try:
                        self.stage.wait(-1)
                        self.stage = _make_stage(self.device, want, max(self.chunk, 4096))
                    except Exception:
                        pass
        if not getattr(self.stage, "shared_paths", None):
            return self._decode_with_threads(todo, out)
        shared_paths = self.stage.shared_paths
        queue = list(todo)
        previous, pending, at = None, [], 0
        oversize: list = []                                       # larger than a whole staging buffer: hashed on their own
        try:
            while at < len(queue) and not self.cancelled():
                slot, _ = self.stage.acquire()
                # a worker's region holds two images of the size seen so far (one, when the buffer has no room for that):
                # camera-sized files mean fewer workers per round with a larger share of the buffer each
                per_image = max(self.bytes_per_image, 1)
                active = int(max(1, min(self.workers, self.stage.stage_bytes // (2 * per_image))))
                if active < self.workers:
                    active = int(max(active, min(self.workers, self.stage.stage_bytes // (per_image + 4096))))
                region = (self.stage.stage_bytes // active) & ~4095
                per_job = int(min(64, max(1, region // (2 * per_image))))
                pending = []
                for r in range(active):
                    part = queue[at:at + per_job]
                    if not part:
                        break
                    at += len(part)
                    job = (shared_paths[slot], r * region, region, [self.paths[k] for k in part])
                    pending.append((part, pool.submit(_decode_into_shared, job)))
                if previous is not None:
                    self._collect(previous[1], previous[0], previous[2], out)
                    previous = None
                positions, placed, nbytes, again = [], [], 0, []
                for part, fut in pending:
                    for k, res in zip(part, fut.result()):
                        if res is None:
                            continue
                        if res[0] == "spill":                     # not decoded: its header says how much room it needs
                            if res[1] + 4096 > self.stage.stage_bytes:
                                oversize.append(k)
                            else:
                                again.append(k)
                                self.bytes_per_image = max(self.bytes_per_image, int(res[1]))
                            continue
                        placed.append((len(positions), res))
                        positions.append(k)
                        nbytes += res[1] * res[2] * res[3]
                pending = []
                if placed and not again:
                    self.bytes_per_image = max(1, nbytes // len(placed))
                queue[at:at] = again                              # next round, with regions that hold them
                handle = None
                if placed:
                    handle = self.stage.submit(slot, [d[0] for _, d in placed], [d[1] for _, d in placed],
                                               [d[2] for _, d in placed], [d[3] for _, d in placed])
                previous = (positions, slot, ([i for i, _ in placed], handle, []))
            if previous is not None:
                self._collect(previous[1], previous[0], previous[2], out)
                previous = None
        finally:
            for _, fut in pending:                                # nobody may still be writing into a staging buffer
                try:
                    fut.result()
                except Exception:
                    pass
        if oversize:
            self._decode_with_threads(sorted(oversize), out)

    def _decode_with_pillow(self, todo: Sequence[int], out: dict) -> None:
        if self.workers > 1 and len(todo) >= self.process_min and getattr(self.stage, "shared_paths", None):
            try:
                return self._decode_with_processes(todo, out)
            except (BrokenProcessPool, OSError):                  # e.g. a __main__ the children cannot import, a staging file
                _pools.pop(self.workers, None)                    # that went away under a worker: threads after all
                self.process_min = 1 << 62
                todo = [k for k in todo if k not in out]
        self._decode_with_threads(todo, out)

    def _decode_with_threads(self, todo: Sequence[int], out: dict) -> None:
        previous = None                                           # (positions, slot, submitted) of the chunk on the GPU
        for c0 in range(0, len(todo), self.chunk):
            if self.cancelled():
                break
            positions = todo[c0:c0 + self.chunk]
            slot, futures = self._start(positions)                # decode of this chunk runs on the pool from here on
            if previous is not None:                              # ... while the previous one is copied and hashed
                self._collect(previous[1], previous[0], previous[2], out)
            previous = (positions, slot, self._submit(slot, futures))
        if previous is not None:
            self._collect(previous[1], previous[0], previous[2], out)

    def run(self) -> Iterator[Tuple[int, Optional[Tuple[int, int]]]]:
        """Yields (file_id, hashes | None) in task order."""
        for fids, ph, dh, ok in self.run_batches():
            for fid, p, d, good in zip(fids.tolist(), ph.tolist(), dh.tolist(), ok.tolist()):
                yield fid, ((p, d) if good else None)

    @staticmethod
    def _drop_reads(future) -> None:
        """A batch read ahead and never decoded gives its buffer back."""
        if future is None:
            return
        try:
            reads = future.result()
        except Exception:
            return
        if reads and reads.get("ahead") is not None:
            reads["ahead"].release()

    def run_batches(self) -> Iterator[Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]]:
        """Yields (file ids, pHash, dHash, done) of one batch after the other, in task order: int64 arrays (the hashes as the
        signed values the table stores) and a mask of the files that were hashed."""
        ahead_pool = ThreadPoolExecutor(max_workers=1)            # one batch is prepared while the one before it is on the GPU
        next_reads = None
        try:
            next_reads = ahead_pool.submit(self._start_reads, 0)
            for start in range(0, len(self.tasks), self.batch):
                stop = min(start + self.batch, len(self.tasks))
                reads_future, next_reads = next_reads, None
                if self.cancelled():
                    self._drop_reads(reads_future)
                    return
                reads = reads_future.result()
                if stop < len(self.tasks):
                    next_reads = ahead_pool.submit(self._start_reads, stop)
                ph, dh = np.zeros(stop - start, np.int64), np.zeros(stop - start, np.int64)
                ok = np.zeros(stop - start, bool)
                taken = np.zeros(stop - start, bool)
                for k in GPU_KINDS:
                    taken[reads[k] - start] = True
                skips = {}
                if reads["ahead"] is not None:
                    ends = np.cumsum([len(reads[k]) for k in GPU_KINDS]).tolist()
                    for kind, e in zip(GPU_KINDS, ends):
                        if kind in ("png", "gif") and len(reads[kind]):
                            mask = self._png_for_pillow(reads["ahead"], e - len(reads[kind]), e, kind)
                            if mask is not None:
                                skips[kind] = mask
                todo = (start + np.nonzero(~taken)[0]).tolist()
                for kind, mask in skips.items():
                    todo += reads[kind][mask].tolist()
                # the Pillow share (other formats, the PNG files given to the processes) beside the GPU decoders' share: its
                # staging route has a context of its own, so neither waits for the other's calls
                out: dict = {}
                failure: list = []
                side = None
                if todo:
                    def pillow_share(positions=sorted(todo)):
                        try:
                            self._decode_with_pillow(positions, out)
                        except BaseException as exc:              # re-raised on the pipeline's thread
                            failure.append(exc)
                    side = threading.Thread(target=pillow_share, name="ke-pillow-share")
                    side.start()
                try:
                    refused = self._decode_on_gpu(reads, start, ph, dh, ok, skips)
                finally:
                    if side is not None:
                        side.join()
                if failure:
                    raise failure[0]
                if refused and not self.stopped:
                    self._decode_with_pillow(sorted(refused), out)
                for k, (p, d) in out.items():
                    ph[k - start], dh[k - start], ok[k - start] = p, d, True
                if self.stopped:                                  # abandoned half way: the caller returns what earlier batches gave
                    return
                yield self.fids[start:stop], ph, dh, ok
        finally:
            self._drop_reads(next_reads)                          # abandoned half way: the batch read ahead gives its buffer back
            ahead_pool.shutdown(wait=True, cancel_futures=True)
            self.pool.shutdown(wait=True, cancel_futures=True)    # no thread may still be writing into a staging buffer
            try:
                self.stage.wait(-1)
            except Exception:
                pass


class _SignedRows(list):
    """Rows whose hashes are Python ints in the signed 64-bit range (what the pipeline yields): nothing to convert on upsert."""


def compute_signatures_mp(tasks: List[Task], *, max_workers: Optional[int] = None, chunksize: int = 64,
                          progress: Optional[Callable[[int, int], None]] = None,
                          cancel_fn: Optional[Callable[[], bool]] = None, device: int = 0) -> List[Row]:
    """[(file_id, path)] -> [(file_id, phash_s64, dhash_s64)], input order, failures omitted."""
    total = len(tasks)
    rows: List[Row] = _SignedRows()
    if total == 0:
        return rows
    workers = max_workers or max(1, _usable_cpus() - 1)
    def report(seen: int) -> None:
        try:
            progress(seen, total)
        except Exception:
            pass

    pipeline = _Pipeline(tasks, workers, chunksize, device, cancel_fn)
    if cancel_fn is None:                           # nobody to ask between files: whole batches at a time
        seen = 0
        for fids, ph, dh, ok in pipeline.run_batches():
            rows.extend(zip(fids[ok].tolist(), ph[ok].tolist(), dh[ok].tolist()))
            if progress is not None:
                before, seen = seen, seen + len(fids)
                for mark in range((before // PROGRESS_STRIDE + 1) * PROGRESS_STRIDE, seen + 1, PROGRESS_STRIDE):
                    report(mark)
                if seen == total and seen % PROGRESS_STRIDE:
                    report(seen)
            else:
                seen += len(fids)
        return rows
    for seen, (fid, sig) in enumerate(pipeline.run(), 1):
        if cancel_fn():
            break                                   # the generator's finally clause cancels what is queued
        if sig is not None:
            rows.append((fid, sig[0], sig[1]))
        if progress is not None and (seen % PROGRESS_STRIDE == 0 or seen == total):
            report(seen)
    return rows


def bulk_upsert_signatures(conn: sqlite3.Connection, rows: Iterable[Row]) -> int:
    """One executemany upsert into signatures(file_id, phash_u64, dhash_u64); returns the row count."""
    if isinstance(rows, _SignedRows):               # compute_signatures_mp's own rows: ints in the signed range already
        payload = rows
    else:
        payload = [(int(fid), _to_signed64(ph), _to_signed64(dh)) for fid, ph, dh in rows]
    if not payload:
        return 0
    # One transaction, as in the reference (src/core/fastsig.py:129-142).  A statement with several thousand rows of VALUES takes
    # about half the time of executemany over the same rows (one prepare / step per 10 000 rows instead of a bind-step-reset
    # per row); a library built with the old limit of 999 variables says so, and executemany does the whole list.
    head, tail = UPSERT_SQL.split("VALUES (?, ?, ?)")
    per = 10000
    with conn:
        try:
            done = 0
            for first in range(0, len(payload), per):
                part = payload[first:first + per]
                done += conn.execute(head + "VALUES " + ",".join(["(?,?,?)"] * len(part)) + tail, list(chain.from_iterable(part))).rowcount or 0
            return done
        except sqlite3.OperationalError as exc:
            if "variables" not in str(exc):
                raise
            conn.rollback()
        return conn.executemany(UPSERT_SQL, payload).rowcount or 0


def fast_fill_missing_signatures(db_path: str, items: List[Task], *, max_workers: Optional[int] = None, chunksize: int = 64,
                                 progress: Optional[Callable[[int, int], None]] = None, apply_to_db: bool = True,
                                 unsafe_fast: bool = True, cancel_fn: Optional[Callable[[], bool]] = None,
                                 device: int = 0) -> List[Row]:
    """Hash the files that have no signature yet and (optionally) upsert them; returns the computed rows."""
    rows = compute_signatures_mp(items, max_workers=max_workers, chunksize=chunksize, progress=progress, cancel_fn=cancel_fn,
                                 device=device)
    if rows and apply_to_db:
        with sqlite3.connect(db_path) as conn:
            if unsafe_fast:
                for pragma in FAST_PRAGMAS:
                    conn.execute(f"PRAGMA {pragma}")
            bulk_upsert_signatures(conn, rows)
    return rows
