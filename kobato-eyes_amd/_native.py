"""ctypes binding of libkeyes_hip.so (include/keyes.h).

The product path has NO CPU fallback: if the shared library is missing, or no gfx950 device
is visible, every entry point raises ``RuntimeError`` -- the same way the reference's
``phash`` raises when OpenCV is missing (src/sig/phash.py:35-36).
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional, Sequence

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KE_LIBKEYES") or os.path.join(_PKG_DIR, "libkeyes_hip.so")   # KE_LIBKEYES: a build variant (benchmarks)

KE_OK = 0
EDGE_DTYPE = np.dtype([("a", "<i8"), ("b", "<i8"), ("h", "<i4"), ("bands", "<i4")])

# every symbol include/keyes.h declares (tests check the library exports all of them)
FILTER_LANCZOS, FILTER_BILINEAR, FILTER_BICUBIC = 0, 1, 2   # include/keyes.h KE_FILTER_*

EXPORTS = (
    "ke_abi_version", "ke_create", "ke_create_error", "ke_destroy", "ke_last_error", "ke_set_stream",
    "ke_get_stream", "ke_synchronize", "ke_device_info", "ke_malloc", "ke_free", "ke_memcpy",
    "ke_hash_images", "ke_hash_uniform", "ke_hash_images_ex", "ke_hash_uniform_ex", "ke_luma_tiles_uniform", "ke_hamming_scan", "ke_band_pairs_after_size",
    "ke_band_pairs_after_size",
    "ke_stage_create", "ke_stage_create_shared", "ke_stage_destroy", "ke_stage_acquire", "ke_stage_submit_hash", "ke_stage_wait",
    "ke_comm_unique_id", "ke_comm_create", "ke_comm_destroy", "ke_allgather_u64", "ke_allgather_hashes", "ke_allgather_edges",
    "ke_interleave_shards", "ke_host_alloc", "ke_host_free", "ke_host_pack", "ke_host_read_files", "ke_jpeg_probe", "ke_jpeg_decode", "ke_png_probe", "ke_png_decode", "ke_jpeg_caveats", "ke_png_caveats", "ke_bmp_probe", "ke_bmp_decode", "ke_bmp_caveats", "ke_gif_probe", "ke_gif_decode", "ke_gif_caveats", "ke_tiff_probe", "ke_tiff_decode", "ke_tiff_caveats", "ke_normalise_rgb", "ke_thumbnail_rgb", "ke_cluster_labels", "ke_ssim_pairs_uniform", "ke_ssim_pairs", "ke_ssim_set_mode", "ke_resize_luma_uniform", "ke_fit_luma_uniform", "ke_tile_ahash", "ke_sad_pairs", "ke_synth_rgb",
    "ke_synth_rgb_indexed",
    "ke_synth_hashes", "ke_last_kernel_ms",
)

_lib: Optional[C.CDLL] = None
_lib_lock = threading.Lock()


class _BatchTooLarge(Exception):
    """A decode batch whose compressed or decoded bytes exceed the context's limits: the caller halves it."""


class NativeUnavailable(RuntimeError):
    """libkeyes_hip.so (or a gfx950 device) is not available."""


def _share_the_hip_runtime_with_torch() -> None:
    """PyTorch's ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 / librccl.  One process must run on ONE HIP
    runtime: if this library bound /opt/rocm's copy and PyTorch (or its RCCL) were loaded afterwards, the second copy would
    see no initialised device.  So when PyTorch is installed its runtime is mapped first (without importing torch); the
    loader then resolves libkeyes_hip.so's dependency to it by soname.  KE_SYSTEM_ROCM=1 keeps the system copy."""
    import importlib.util
    import sys

    if "torch" in sys.modules or os.environ.get("KE_SYSTEM_ROCM"):
        return                      # torch already brought its runtime (same effect) / the caller wants /opt/rocm's
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library() -> C.CDLL:
    """Load libkeyes_hip.so and declare the prototypes.  Does not touch the GPU."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        _share_the_hip_runtime_with_torch()
        if not os.path.exists(LIB_PATH):
            raise NativeUnavailable(
                f"{LIB_PATH} is missing: build it with kobato-eyes_amd/build.sh (hipcc, gfx950). "
                "There is no CPU fallback for the hash/scan path."
            )
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as exc:  # e.g. libamdhip64 not found
            raise NativeUnavailable(f"cannot load {LIB_PATH}: {exc}") from exc
        vp, i32, i64, u64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double
        lib.ke_abi_version.restype = C.c_int
        lib.ke_create.argtypes = [C.c_int]
        lib.ke_create.restype = vp
        lib.ke_create_error.restype = C.c_char_p
        lib.ke_destroy.argtypes = [vp]
        lib.ke_destroy.restype = None
        lib.ke_last_error.argtypes = [vp]
        lib.ke_last_error.restype = C.c_char_p
        lib.ke_set_stream.argtypes = [vp, vp]
        lib.ke_get_stream.argtypes = [vp]
        lib.ke_get_stream.restype = vp
        lib.ke_synchronize.argtypes = [vp]
        lib.ke_device_info.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(i32), C.POINTER(i64)]
        lib.ke_malloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
        lib.ke_free.argtypes = [vp, vp]
        lib.ke_memcpy.argtypes = [vp, vp, vp, C.c_size_t]
        lib.ke_hash_images.argtypes = [vp, vp, vp, vp, vp, i32, i64, vp, vp, vp]
        lib.ke_hash_uniform.argtypes = [vp, vp, i64, i32, i32, i32, vp, vp]
        lib.ke_hash_images_ex.argtypes = [vp, vp, vp, vp, vp, i32, i64, vp, vp, vp, vp]
        lib.ke_hash_uniform_ex.argtypes = [vp, vp, i64, i32, i32, i32, vp, vp, vp]
        lib.ke_luma_tiles_uniform.argtypes = [vp, vp, i64, i32, i32, i32, vp, vp]
        lib.ke_stage_create.argtypes = [vp, C.c_size_t, i64, i32]
        lib.ke_stage_create_shared.argtypes = [vp, vp, C.c_size_t, i64, i32]
        lib.ke_stage_destroy.argtypes = [vp]
        lib.ke_stage_acquire.argtypes = [vp, C.POINTER(i32), C.POINTER(vp), C.POINTER(C.c_size_t)]
        lib.ke_stage_submit_hash.argtypes = [vp, i32, vp, vp, vp, vp, i64, vp, vp, vp, vp]
        lib.ke_stage_wait.argtypes = [vp, i32]
        lib.ke_comm_unique_id.argtypes = [vp]
        lib.ke_comm_create.argtypes = [vp, vp, i32, i32, C.POINTER(vp)]
        lib.ke_comm_destroy.argtypes = [vp, vp]
        lib.ke_allgather_u64.argtypes = [vp, vp, i32, vp, i64, vp]
        lib.ke_allgather_hashes.argtypes = [vp, vp, i32, vp, i64, vp]
        lib.ke_allgather_edges.argtypes = [vp, vp, i32, vp, i64, vp, i64, C.POINTER(i64), vp]
        lib.ke_interleave_shards.argtypes = [vp, vp, i32, i64, vp]
        lib.ke_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
        lib.ke_host_free.argtypes = [vp, vp]
        lib.ke_host_pack.argtypes = [vp, vp, vp, vp, i64]
        lib.ke_host_read_files.argtypes = [vp, i64, vp, C.c_uint64, vp, vp, C.POINTER(C.c_uint64)]
        lib.ke_jpeg_probe.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp]
        lib.ke_jpeg_decode.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp]
        lib.ke_png_probe.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp]
        lib.ke_png_decode.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp]
        lib.ke_jpeg_caveats.argtypes = [vp, vp, vp, i64, vp]
        lib.ke_png_caveats.argtypes = [vp, vp, vp, i64, vp]
        lib.ke_bmp_probe.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp]
        lib.ke_bmp_decode.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp]
        lib.ke_bmp_caveats.argtypes = [vp, vp, vp, i64, vp]
        lib.ke_gif_probe.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp]
        lib.ke_gif_decode.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp]
        lib.ke_gif_caveats.argtypes = [vp, vp, vp, i64, vp]
        lib.ke_tiff_probe.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp]
        lib.ke_tiff_decode.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp]
        lib.ke_tiff_caveats.argtypes = [vp, vp, vp, i64, vp]
        lib.ke_normalise_rgb.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, vp, vp]
        lib.ke_thumbnail_rgb.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
        lib.ke_band_pairs_after_size.argtypes = [vp, vp, vp, i64, i32, i32, dbl, i64, vp]
        lib.ke_hamming_scan.argtypes = [vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, dbl, i64, vp, i64,
                                        C.POINTER(i64), vp]
        lib.ke_cluster_labels.argtypes = [vp, i64, i64, vp]
        lib.ke_ssim_pairs_uniform.argtypes = [vp, vp, i64, i32, i32, i32, vp, vp, i64, vp]
        lib.ke_ssim_set_mode.argtypes = [vp, i32]
        lib.ke_ssim_pairs.argtypes = [vp, vp, vp, vp, vp, i32, i64, vp, vp, i64, vp, vp]
        lib.ke_resize_luma_uniform.argtypes = [vp, vp, i64, i32, i32, i32, i32, i32, i32, vp]
        lib.ke_fit_luma_uniform.argtypes = [vp, vp, i64, i32, i32, i32, i32, i32, i32, vp]
        lib.ke_tile_ahash.argtypes = [vp, vp, i64, i32, i32, vp]
        lib.ke_sad_pairs.argtypes = [vp, vp, i64, i64, vp, vp, i64, vp]
        lib.ke_synth_rgb.argtypes = [vp, u64, i64, i64, i32, i32, vp]
        lib.ke_synth_rgb_indexed.argtypes = [vp, u64, vp, i64, i32, i32, vp]
        lib.ke_synth_hashes.argtypes = [vp, u64, i64, vp]
        lib.ke_last_kernel_ms.argtypes = [vp, i32]
        lib.ke_last_kernel_ms.restype = dbl
        for name in ("ke_set_stream", "ke_synchronize", "ke_device_info", "ke_malloc", "ke_free", "ke_memcpy",
                     "ke_hash_images", "ke_hash_uniform", "ke_hash_images_ex", "ke_hash_uniform_ex", "ke_luma_tiles_uniform", "ke_hamming_scan", "ke_band_pairs_after_size",
                     "ke_stage_create", "ke_stage_create_shared", "ke_stage_destroy", "ke_stage_acquire", "ke_stage_submit_hash", "ke_stage_wait",
                     "ke_comm_unique_id", "ke_comm_create", "ke_comm_destroy", "ke_allgather_u64", "ke_allgather_hashes", "ke_allgather_edges",
                     "ke_interleave_shards", "ke_host_alloc", "ke_host_free", "ke_host_pack", "ke_host_read_files", "ke_jpeg_probe", "ke_jpeg_decode", "ke_png_probe", "ke_png_decode", "ke_jpeg_caveats", "ke_png_caveats", "ke_bmp_probe", "ke_bmp_decode", "ke_bmp_caveats", "ke_gif_probe", "ke_gif_decode", "ke_gif_caveats", "ke_tiff_probe", "ke_tiff_decode", "ke_tiff_caveats", "ke_normalise_rgb", "ke_thumbnail_rgb", "ke_cluster_labels", "ke_ssim_pairs_uniform", "ke_ssim_pairs", "ke_ssim_set_mode", "ke_resize_luma_uniform", "ke_fit_luma_uniform", "ke_tile_ahash",
                     "ke_sad_pairs", "ke_synth_rgb", "ke_synth_rgb_indexed", "ke_synth_hashes"):
            getattr(lib, name).restype = C.c_int
        _lib = lib
        return lib


def _addr(buf) -> Optional[int]:
    """Address of a host ndarray, a raw device pointer (int) or None."""
    if buf is None:
        return None
    if isinstance(buf, np.ndarray):
        return buf.ctypes.data
    if isinstance(buf, int):
        return buf
    if hasattr(buf, "data_ptr"):  # torch tensor (host or device): plumbing only
        return int(buf.data_ptr())
    raise TypeError(f"unsupported buffer type {type(buf)!r}")


def _leave_bombs_to_pillow(w, h, st, sizes) -> None:
    """Every decode here stands in for an ``Image.open``, which raises DecompressionBombError on an image of more than twice
    ``Image.MAX_IMAGE_PIXELS`` pixels: such files are marked unsupported (1) and hidden from the decoder (size 0: it reports
    them as not decodable and writes nothing), so that the caller's Pillow route raises as the reference's does.  The live
    value counts (safe_load_image changes it while it runs); None = no cap."""
    try:
        from PIL import Image
    except ModuleNotFoundError:  # pragma: no cover
        return
    cap = Image.MAX_IMAGE_PIXELS
    if cap is not None:
        bombs = (st == 0) & (w.astype(np.int64) * h > 2 * int(cap))
        st[bombs] = 1
        sizes[bombs] = 0


class FilesAhead:
    """The files of a batch in one of a context's read-ahead buffers (Context.read_files_ahead): file i is
    ``flat[offsets[i]:offsets[i] + sizes[i]]`` (size 0 = unreadable).  ``release()`` hands the buffer back."""

    def __init__(self, ctx, slot: int, flat, offsets, sizes) -> None:
        self._ctx, self._slot, self.flat, self.offsets, self.sizes = ctx, slot, flat, offsets, sizes
        self.probed = {}                  # (kind, lo, hi) -> (widths, heights, channels, status) where the reader parsed headers too

    def __len__(self) -> int:
        return len(self.sizes)

    def release(self) -> None:
        if self._ctx is not None:
            self._ctx._ahead[self._slot][2] = False
            self._ctx, self.flat = None, None


class Context:
    """One ke_ctx: a device, a stream, scratch buffers.  Not re-entrant."""

    def __init__(self, device: int = 0) -> None:
        self._lib = load_library()
        self._h = self._lib.ke_create(int(device))
        if not self._h:
            msg = self._lib.ke_create_error().decode("utf-8", "replace")
            raise NativeUnavailable(f"ke_create({device}) failed: {msg}")
        self.device = int(device)
        self._lock = threading.RLock()
        self._pack_ptr, self._pack_cap = 0, 0
        self._decoded_ptr, self._decoded_cap = 0, 0
        self._ahead = [[0, 0, False], [0, 0, False]]         # read-ahead buffers: [page-locked ptr, capacity, taken]
        self._ahead_lock = threading.Lock()
        self.decode_kernel_ms = 0.0        # kernel time of every ke_jpeg_decode / ke_png_decode call so far (a batch beyond the limits is several)
        # a decode call beyond these is split in halves: compressed bytes in page-locked memory, decoded pixels on the device
        self.pack_limit = int(os.environ.get("KE_PACK_LIMIT_BYTES", str(16 << 30)))
        self.decode_limit = int(os.environ.get("KE_DECODE_LIMIT_BYTES", str(48 << 30)))

    # -- lifetime ---------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            if getattr(self, "_pack_ptr", 0):
                self._lib.ke_host_free(self._h, self._pack_ptr)
                self._pack_ptr, self._pack_cap = 0, 0
            if getattr(self, "_decoded_ptr", 0):
                self._lib.ke_free(self._h, self._decoded_ptr)
                self._decoded_ptr, self._decoded_cap = 0, 0
            for buf in getattr(self, "_ahead", []):
                if buf[0]:
                    self._lib.ke_host_free(self._h, buf[0])
                    buf[0], buf[1] = 0, 0
            self._lib.ke_destroy(self._h)
            self._h = None

    def __del__(self) -> None:  # pragma: no cover - interpreter shutdown order
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str) -> None:
        if rc != KE_OK:
            msg = self._lib.ke_last_error(self._h).decode("utf-8", "replace")
            if rc == -1:
                raise ValueError(f"{what}: {msg}")
            raise RuntimeError(f"{what} failed (rc={rc}): {msg}")

    # -- plumbing ---------------------------------------------------------------------------
    def set_stream(self, hip_stream: Optional[int]) -> None:
        self._check(self._lib.ke_set_stream(self._h, hip_stream), "ke_set_stream")

    def synchronize(self) -> None:
        self._check(self._lib.ke_synchronize(self._h), "ke_synchronize")

    def device_info(self) -> dict:
        name = C.create_string_buffer(256)
        cus, mem = C.c_int32(), C.c_int64()
        self._check(self._lib.ke_device_info(self._h, name, 256, C.byref(cus), C.byref(mem)), "ke_device_info")
        return {"name": name.value.decode(), "compute_units": cus.value, "total_mem": mem.value}

    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._check(self._lib.ke_malloc(self._h, nbytes, C.byref(p)), "ke_malloc")
        return int(p.value)

    def free(self, ptr: int) -> None:
        self._check(self._lib.ke_free(self._h, ptr), "ke_free")

    def memcpy(self, dst, src, nbytes: int) -> None:
        self._check(self._lib.ke_memcpy(self._h, _addr(dst), _addr(src), nbytes), "ke_memcpy")

    def last_kernel_ms(self, kind: int) -> float:
        return float(self._lib.ke_last_kernel_ms(self._h, kind))

    # -- hashing ----------------------------------------------------------------------------
    def hash_uniform(self, pixels, n: int, width: int, height: int, channels: int, *, want_phash=True,
                     want_dhash=True, phash_out=None, dhash_out=None, margin_out=None):
        """pixels: host ndarray (n,h,w[,c]) u8 or a device pointer.  Returns (phash u64[n] | None, dhash | None)
        as host arrays unless explicit output buffers (host arrays or device pointers) are given.
        margin_out: float32[n] host array or device pointer that receives the pHash tie margins (ke_hash_uniform_ex)."""
        if isinstance(pixels, np.ndarray):
            pixels = np.ascontiguousarray(pixels, dtype=np.uint8)
            if pixels.size != n * width * height * channels:
                raise ValueError("pixel buffer does not match n*h*w*c")
        ph = phash_out if phash_out is not None else (np.empty(n, np.uint64) if want_phash else None)
        dh = dhash_out if dhash_out is not None else (np.empty(n, np.uint64) if want_dhash else None)
        with self._lock:
            if margin_out is not None:
                self._check(self._lib.ke_hash_uniform_ex(self._h, _addr(pixels), n, width, height, channels, _addr(ph),
                                                         _addr(dh), _addr(margin_out)), "ke_hash_uniform_ex")
            else:
                self._check(self._lib.ke_hash_uniform(self._h, _addr(pixels), n, width, height, channels, _addr(ph),
                                                      _addr(dh)), "ke_hash_uniform")
        return ph, dh

    def luma_tiles_uniform(self, pixels, n: int, width: int, height: int, channels: int, *, want32=True, want98=True):
        if isinstance(pixels, np.ndarray):
            pixels = np.ascontiguousarray(pixels, dtype=np.uint8)
        t32 = np.empty((n, 32, 32), np.uint8) if want32 else None
        t98 = np.empty((n, 8, 9), np.uint8) if want98 else None
        with self._lock:
            self._check(self._lib.ke_luma_tiles_uniform(self._h, _addr(pixels), n, width, height, channels, _addr(t32),
                                                        _addr(t98)), "ke_luma_tiles_uniform")
        return t32, t98

    def hash_images(self, images: Sequence[np.ndarray], *, want_dhash=True, want_margin=False):
        """Ragged batch of host images sharing one channel count.  Returns (phash, dhash|None, status), plus the
        float32 tie margins as a fourth item with ``want_margin``."""
        n = len(images)
        if n == 0:
            return np.empty(0, np.uint64), (np.empty(0, np.uint64) if want_dhash else None), np.empty(0, np.int32)
        chans = {1 if im.ndim == 2 else im.shape[2] for im in images}
        if len(chans) != 1:
            raise ValueError("all images of one call must share the channel count")
        ch = chans.pop()
        widths = np.array([im.shape[1] for im in images], np.int32)
        heights = np.array([im.shape[0] for im in images], np.int32)
        sizes = widths.astype(np.int64) * heights * ch
        padded = (sizes + 15) & ~np.int64(15)          # every image starts on a 16-byte boundary (see keyes.h)
        offsets = np.zeros(n, np.uint64)
        offsets[1:] = np.cumsum(padded[:-1]).astype(np.uint64)
        flat = np.zeros(int(padded.sum()), np.uint8)
        for im, off, sz in zip(images, offsets, sizes):
            flat[int(off):int(off) + int(sz)] = np.ascontiguousarray(im, dtype=np.uint8).reshape(-1)
        ph = np.empty(n, np.uint64)
        dh = np.empty(n, np.uint64) if want_dhash else None
        status = np.empty(n, np.int32)
        margin = np.empty(n, np.float32) if want_margin else None
        with self._lock:
            self._check(self._lib.ke_hash_images_ex(self._h, _addr(flat), _addr(offsets), _addr(widths), _addr(heights), ch, n,
                                                    _addr(ph), _addr(dh), _addr(status), _addr(margin)), "ke_hash_images_ex")
        return (ph, dh, status, margin) if want_margin else (ph, dh, status)

    # -- pinned staging ---------------------------------------------------------------------
    def stage_create(self, bytes_per_buffer: int, max_images: int, n_buffers: int = 2) -> None:
        self._stage_geom = None                              # whoever cached "my buffers are in place" (fastsig) must look again
        self._check(self._lib.ke_stage_create(self._h, int(bytes_per_buffer), int(max_images), int(n_buffers)), "ke_stage_create")

    def stage_create_shared(self, addresses, bytes_per_buffer: int, max_images: int) -> None:
        """Staging on the caller's page-aligned buffers (shared memory that decoder processes write into)."""
        ptrs = (C.c_void_p * len(addresses))(*[int(a) for a in addresses])
        self._stage_geom = None
        self._check(self._lib.ke_stage_create_shared(self._h, ptrs, int(bytes_per_buffer), int(max_images), len(addresses)),
                    "ke_stage_create_shared")

    def stage_destroy(self) -> None:
        self._stage_geom = None
        self._check(self._lib.ke_stage_destroy(self._h), "ke_stage_destroy")

    def stage_acquire(self):
        """Next pinned buffer in rotation: (slot, uint8 ndarray view of the whole buffer).  Blocks until the buffer's
        previous batch is done (its results land in the arrays given at its submit)."""
        slot, ptr, size = C.c_int32(), C.c_void_p(), C.c_size_t()
        with self._lock:
            self._check(self._lib.ke_stage_acquire(self._h, C.byref(slot), C.byref(ptr), C.byref(size)), "ke_stage_acquire")
        view = np.ctypeslib.as_array((C.c_uint8 * size.value).from_address(ptr.value))
        return int(slot.value), view

    def stage_submit_hash(self, slot: int, offsets, widths, heights, channels, *, want_dhash=True, want_margin=False):
        """Enqueue copy + hash of the images described; returns a dict of host arrays (phash, dhash, status, margin) that
        are valid after ``stage_wait(slot)`` (status: immediately).  The arrays must be kept alive until then."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        widths = np.ascontiguousarray(widths, dtype=np.int32)
        heights = np.ascontiguousarray(heights, dtype=np.int32)
        channels = np.ascontiguousarray(channels, dtype=np.int32)
        n = len(offsets)
        out = {"phash": np.zeros(n, np.uint64), "dhash": np.zeros(n, np.uint64) if want_dhash else None,
               "status": np.zeros(n, np.int32), "margin": np.zeros(n, np.float32) if want_margin else None,
               "_keep": (offsets, widths, heights, channels)}
        with self._lock:
            self._check(self._lib.ke_stage_submit_hash(self._h, slot, _addr(offsets), _addr(widths), _addr(heights), _addr(channels), n,
                                                       _addr(out["phash"]), _addr(out["dhash"]), _addr(out["status"]),
                                                       _addr(out["margin"])), "ke_stage_submit_hash")
        return out

    def stage_wait(self, slot: int = -1) -> None:
        with self._lock:
            self._check(self._lib.ke_stage_wait(self._h, int(slot)), "ke_stage_wait")

    # -- RCCL exchange ----------------------------------------------------------------------
    def comm_create(self, unique_id: bytes, world: int, rank: int) -> int:
        """ncclCommInitRank on this context's device; ``unique_id``: the 128 bytes of ``comm_unique_id()`` of rank 0."""
        comm = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self._check(self._lib.ke_comm_create(self._h, buf, world, rank, C.byref(comm)), "ke_comm_create")
        return int(comm.value)

    def comm_destroy(self, comm: int) -> None:
        self._check(self._lib.ke_comm_destroy(self._h, comm), "ke_comm_destroy")

    def allgather_hashes(self, comm: int, world: int, local, n_total: int, table) -> None:
        """local: device pointer to ceil(n_total/world) u64; table: device pointer to n_total u64 (corpus order)."""
        self._check(self._lib.ke_allgather_hashes(self._h, comm, world, _addr(local), n_total, _addr(table)), "ke_allgather_hashes")

    def interleave_shards(self, gathered, world: int, n_total: int, table) -> None:
        self._check(self._lib.ke_interleave_shards(self._h, _addr(gathered), world, n_total, _addr(table)), "ke_interleave_shards")

    def allgather_edges(self, comm: int, world: int, local, n_local: int):
        """local: device pointer to this rank's ke_edge records.  Returns (all edges of all ranks as a host array, counts)."""
        counts = np.zeros(world, np.int64)
        total = C.c_int64(0)
        cap = max(1024, 2 * int(n_local) * world)
        while True:
            merged = np.empty(cap, EDGE_DTYPE)
            self._check(self._lib.ke_allgather_edges(self._h, comm, world, _addr(local), n_local, _addr(merged), cap, C.byref(total),
                                                     _addr(counts)), "ke_allgather_edges")
            if total.value <= cap:
                return merged[: total.value], counts
            cap = int(total.value)

    # -- JPEG decode on the GPU ---------------------------------------------------------------
    @staticmethod
    def _pack_blobs(blobs):
        sizes = np.fromiter((len(b) for b in blobs), np.uint64, len(blobs))
        offsets = np.zeros(len(blobs), np.uint64)
        offsets[1:] = np.cumsum(sizes[:-1])
        flat = np.frombuffer(b"".join(blobs) + bytes(64), np.uint8)     # one C-level copy; the decoder takes any alignment
        return flat, offsets, sizes

    def _grow_pack(self, total: int) -> None:
        if total > self._pack_cap:
            if self._pack_ptr:
                self._check(self._lib.ke_host_free(self._h, self._pack_ptr), "ke_host_free")
                self._pack_ptr, self._pack_cap = 0, 0
            cap = max(total + total // 4, 1 << 24)
            p = C.c_void_p()
            self._check(self._lib.ke_host_alloc(self._h, cap, C.byref(p)), "ke_host_alloc")
            self._pack_ptr, self._pack_cap = int(p.value), cap

    def _read_files_pinned(self, paths, bounded: bool = True):
        """The files themselves, read by the library's host threads straight into the page-locked buffer (no bytes objects, no
        interpreter loop over the files); unreadable files get size 0.  Call with the lock held."""
        n = len(paths)
        names = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
        offsets, sizes = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        needed = C.c_uint64(0)
        rc = self._lib.ke_host_read_files(names, n, self._pack_ptr, self._pack_cap, _addr(offsets), _addr(sizes), C.byref(needed))
        if rc == -4:                                       # KE_ENOMEM: the buffer is too small for this batch
            if bounded and int(needed.value) > self.pack_limit and n > 1:
                raise _BatchTooLarge
            self._grow_pack(int(needed.value))
            rc = self._lib.ke_host_read_files(names, n, self._pack_ptr, self._pack_cap, _addr(offsets), _addr(sizes), C.byref(needed))
        if rc != KE_OK:
            raise ValueError("ke_host_read_files: bad arguments")
        total = int(needed.value)
        flat = np.ctypeslib.as_array((C.c_uint8 * total).from_address(self._pack_ptr))
        return flat, offsets, sizes

    def read_files_ahead(self, paths, spans=()):
        """Read files into one of the context's two page-locked read-ahead buffers -- from any thread, while another call of the
        context is decoding the previous batch on the GPU (nothing here touches the stream or the context's lock).  Returns a
        FilesAhead to pass to ``jpeg_hash(..., ahead=(it, lo, hi))`` and to ``release()`` afterwards, or None when both buffers
        are taken or the files exceed ``pack_limit`` (the caller then lets the decode call read them itself).  ``spans`` =
        [(kind, lo, hi)]: the headers of files lo..hi are parsed here as well (``ke_<kind>_probe``), off the decoding thread."""
        paths = list(paths)
        n = len(paths)
        if n == 0:
            return None
        with self._ahead_lock:
            slot = next((k for k, buf in enumerate(self._ahead) if not buf[2]), None)
            if slot is None:
                return None
            self._ahead[slot][2] = True
        buf = self._ahead[slot]
        try:
            names = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
            offsets, sizes = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
            needed = C.c_uint64(0)
            rc = self._lib.ke_host_read_files(names, n, buf[0], buf[1], _addr(offsets), _addr(sizes), C.byref(needed))
            if rc == -4:                                   # KE_ENOMEM: the buffer is too small for this batch
                if int(needed.value) > self.pack_limit and n > 1:
                    raise _BatchTooLarge
                if buf[0]:
                    self._check(self._lib.ke_host_free(self._h, buf[0]), "ke_host_free")
                    buf[0], buf[1] = 0, 0
                cap = max(int(needed.value) + int(needed.value) // 4, 1 << 24)
                p = C.c_void_p()
                self._check(self._lib.ke_host_alloc(self._h, cap, C.byref(p)), "ke_host_alloc")
                buf[0], buf[1] = int(p.value), cap
                rc = self._lib.ke_host_read_files(names, n, buf[0], buf[1], _addr(offsets), _addr(sizes), C.byref(needed))
            if rc != KE_OK:
                raise ValueError("ke_host_read_files: bad arguments")
        except _BatchTooLarge:
            buf[2] = False
            return None
        except BaseException:
            buf[2] = False
            raise
        flat = np.ctypeslib.as_array((C.c_uint8 * int(needed.value)).from_address(buf[0]))
        held = FilesAhead(self, slot, flat, offsets, sizes)
        for kind, lo, hi in spans:
            if hi > lo:
                w, h, c, st = (np.zeros(hi - lo, np.int32) for _ in range(4))
                o, z = np.ascontiguousarray(offsets[lo:hi]), np.ascontiguousarray(sizes[lo:hi])
                if getattr(self._lib, f"ke_{kind}_probe")(_addr(flat), _addr(o), _addr(z), hi - lo, _addr(w), _addr(h), _addr(c), _addr(st)) == KE_OK:
                    held.probed[(kind, lo, hi)] = (w, h, c, st)
        return held

    def _pack_blobs_pinned(self, blobs):
        """The files back to back in the context's page-locked buffer (grown on demand, reused from call to call): the copy
        to the device then runs at link speed.  Call with the lock held; the buffer is busy until the decode has returned."""
        sizes = np.fromiter((len(b) for b in blobs), np.uint64, len(blobs))
        offsets = np.zeros(len(blobs), np.uint64)
        offsets[1:] = np.cumsum(sizes[:-1])
        total = int(sizes.sum()) + 64
        if total > self.pack_limit and len(blobs) > 1:
            raise _BatchTooLarge
        self._grow_pack(total)
        base = self._pack_ptr
        flat = np.ctypeslib.as_array((C.c_uint8 * total).from_address(base))
        n = len(blobs)
        srcs = (C.c_char_p * n)(*blobs)                    # the buffers of the bytes objects themselves, no copies
        if self._lib.ke_host_pack(base, srcs, _addr(offsets), _addr(sizes), n) != KE_OK:
            raise ValueError("ke_host_pack: bad arguments")
        C.memset(base + total - 64, 0, 64)
        return flat, offsets, sizes

    def jpeg_probe(self, blobs, kind: str = "jpeg"):
        """(widths, heights, channels, status) of JPEG (or, kind="png", PNG) files given as bytes; status 0 = the GPU
        decoder takes the file."""
        flat, offsets, sizes = self._pack_blobs(blobs)
        n = len(blobs)
        w, h, c, st = (np.zeros(n, np.int32) for _ in range(4))
        rc = getattr(self._lib, f"ke_{kind}_probe")(_addr(flat), _addr(offsets), _addr(sizes), n, _addr(w), _addr(h), _addr(c), _addr(st))
        if rc != KE_OK:
            raise ValueError(f"ke_{kind}_probe: bad arguments")
        return w, h, c, st

    def png_probe(self, blobs):
        return self.jpeg_probe(blobs, "png")

    def png_decode(self, blobs):
        """Pixels of PNG files decoded on the GPU (HxW, HxWx3 or HxWx4), None where the decoder refused the file."""
        return self.jpeg_decode(blobs, "png")

    def png_hash(self, blobs, *, want_dhash=True):
        return self.jpeg_hash(blobs, want_dhash=want_dhash, kind="png")

    def bmp_probe(self, blobs):
        return self.jpeg_probe(blobs, "bmp")

    def bmp_decode(self, blobs):
        """Pixels of uncompressed BMP files unpacked on the GPU (HxW luma of a palette file, HxWx3 or HxWx4), None where the
        unpacker refused the file."""
        return self.jpeg_decode(blobs, "bmp")

    def bmp_hash(self, blobs, *, want_dhash=True):
        return self.jpeg_hash(blobs, want_dhash=want_dhash, kind="bmp")

    def gif_probe(self, blobs):
        return self.jpeg_probe(blobs, "gif")

    def gif_decode(self, blobs):
        """Luma (HxW) of the first frame of GIF files decoded on the GPU -- what ``Image.open(f).convert("L")`` yields --, None
        where the decoder refused the file."""
        return self.jpeg_decode(blobs, "gif")

    def gif_hash(self, blobs, *, want_dhash=True):
        return self.jpeg_hash(blobs, want_dhash=want_dhash, kind="gif")

    def tiff_probe(self, blobs):
        return self.jpeg_probe(blobs, "tiff")

    def tiff_decode(self, blobs):
        """Pixels of uncompressed 8-bit TIFF files unpacked on the GPU (HxW gray or luma of a palette file, HxWx3, HxWx4), None
        where the unpacker refused the file."""
        return self.jpeg_decode(blobs, "tiff")

    def tiff_hash(self, blobs, *, want_dhash=True):
        return self.jpeg_hash(blobs, want_dhash=want_dhash, kind="tiff")

    def _jpeg_to_device(self, blobs, kind: str = "jpeg", *, paths=None, ahead=None, skip=None):
        """Decode what the GPU decoder takes into the context's decode buffer (device memory, grown on demand and kept:
        allocating tens of GB per call costs up to a second): (device ptr or 0, byte offsets, widths, heights, channels,
        status).  Call with the lock held and keep it until the pixels have been used."""
        n = ahead[2] - ahead[1] if ahead is not None else len(blobs) if paths is None else len(paths)
        w, h, c, st = (np.zeros(n, np.int32) for _ in range(4))
        with self._lock:
            if ahead is not None:                        # files lo..hi of a batch some thread has read already
                held, lo, hi = ahead
                flat, offsets, sizes = held.flat, np.ascontiguousarray(held.offsets[lo:hi]), np.array(held.sizes[lo:hi], np.uint64)
            else:
                flat, offsets, sizes = self._pack_blobs_pinned(blobs) if paths is None else self._read_files_pinned(paths)
            known = ahead[0].probed.get((kind, ahead[1], ahead[2])) if ahead is not None else None
            if known is not None:
                w, h, c, st = (a.copy() for a in known)
            else:
                rc = getattr(self._lib, f"ke_{kind}_probe")(_addr(flat), _addr(offsets), _addr(sizes), n, _addr(w), _addr(h), _addr(c), _addr(st))
                if rc != KE_OK:
                    raise ValueError(f"ke_{kind}_probe: bad arguments")
            _leave_bombs_to_pillow(w, h, st, sizes)
            if skip is not None:                         # files the caller keeps for another decoder: as if refused
                skip = np.asarray(skip, bool)
                st[skip & (st == 0)] = 1
                sizes[skip] = 0
            nbytes = np.where(st == 0, w.astype(np.int64) * h * c, 0)
            padded = (nbytes + 15) & ~np.int64(15)
            out_off = np.zeros(n, np.uint64)
            out_off[1:] = np.cumsum(padded[:-1]).astype(np.uint64)
            total = int(padded.sum())
            if total == 0:
                return 0, out_off, w, h, c, st
            if total > self.decode_limit and n > 1:
                raise _BatchTooLarge
            if total + 64 > self._decoded_cap:
                if self._decoded_ptr:
                    self.free(self._decoded_ptr)
                    self._decoded_ptr, self._decoded_cap = 0, 0
                cap = total + total // 8 + 64
                self._decoded_ptr, self._decoded_cap = self.malloc(cap), cap
            dev = self._decoded_ptr
            self._check(getattr(self._lib, f"ke_{kind}_decode")(self._h, _addr(flat), _addr(offsets), _addr(sizes), n, dev,
                                                                _addr(out_off), _addr(st)), f"ke_{kind}_decode")
            self.decode_kernel_ms += self.last_kernel_ms(4)
        return dev, out_off, w, h, c, st

    def decode_files_owned(self, paths, kind: str = "jpeg", *, by_shape: bool = False):
        """Files on disk decoded into a device buffer of their own (the caller frees it with ``free``): (device ptr or 0, byte
        offsets, widths, heights, channels, status, caveat flags).  ``flags`` are ke_jpeg_caveats / ke_png_caveats' bits: what
        the reference's defensive loader would do to the file beyond Image.open (EXIF orientation, transparency).
        ``by_shape``: images of one (width, height, channels) lie back to back without padding, so that each such group
        can go to the uniform-batch kernels as it is (np.unique over (w, h, c) of the decodable files gives the groups)."""
        paths = list(paths)
        n = len(paths)
        w, h, c, st, flags = (np.zeros(n, np.int32) for _ in range(5))
        out_off = np.zeros(n, np.uint64)
        if n == 0:
            return 0, out_off, w, h, c, st, flags
        with self._lock:
            flat, offsets, sizes = self._read_files_pinned(paths, bounded=False)     # the caller paces its batches
            probe = getattr(self._lib, f"ke_{kind}_probe")
            if probe(_addr(flat), _addr(offsets), _addr(sizes), n, _addr(w), _addr(h), _addr(c), _addr(st)) != KE_OK:
                raise ValueError(f"ke_{kind}_probe: bad arguments")
            if getattr(self._lib, f"ke_{kind}_caveats")(_addr(flat), _addr(offsets), _addr(sizes), n, _addr(flags)) != KE_OK:
                raise ValueError(f"ke_{kind}_caveats: bad arguments")
            _leave_bombs_to_pillow(w, h, st, sizes)
            nbytes = np.where(st == 0, w.astype(np.int64) * h * c, 0)
            if by_shape:
                order = np.lexsort((w, h, c, st != 0))              # decodable files first, grouped by shape
                sorted_bytes = nbytes[order]
                key = np.stack([w[order], h[order], c[order]], 1)
                new_group = np.ones(n, bool)
                new_group[1:] = (key[1:] != key[:-1]).any(1)
                starts = np.zeros(n, np.int64)
                at = 0
                for k in range(n):                                  # groups start on 16 bytes, images inside follow tightly
                    if new_group[k]:
                        at = (at + 15) & ~15
                    starts[k] = at
                    at += int(sorted_bytes[k])
                out_off[order] = starts.astype(np.uint64)
                out_off[nbytes == 0] = np.uint64(0xFFFFFFFFFFFFFFFF)   # not laid out (refused by the probe)
                total = at
            else:
                padded = (nbytes + 15) & ~np.int64(15)
                out_off[1:] = np.cumsum(padded[:-1]).astype(np.uint64)
                total = int(padded.sum())
            if total == 0:
                return 0, out_off, w, h, c, st, flags
            dev = self.malloc(total + 64)
            try:
                self._check(getattr(self._lib, f"ke_{kind}_decode")(self._h, _addr(flat), _addr(offsets), _addr(sizes), n, dev,
                                                                    _addr(out_off), _addr(st)), f"ke_{kind}_decode")
            except Exception:
                self.free(dev)
                raise
        return dev, out_off, w, h, c, st, flags

    def normalise_rgb(self, src: int, src_offsets, widths, heights, channels, orientations, *, by_shape: bool = False):
        """Images on the device (decoded files) -> a device buffer of their own (the caller frees it) holding them as the
        reference's loader would hand them over: turned by their EXIF orientation, RGBA composited over white
        (ke_normalise_rgb).  Returns (device ptr, byte offsets, widths, heights) of the RGB results, in input order;
        ``by_shape``: results of one (width, height) lie back to back without padding (a group starts on 16 bytes), so that
        each group can go to the uniform-batch kernels as it is."""
        so = np.ascontiguousarray(src_offsets, dtype=np.uint64)
        w, h = np.ascontiguousarray(widths, dtype=np.int32), np.ascontiguousarray(heights, dtype=np.int32)
        c, o = np.ascontiguousarray(channels, dtype=np.int32), np.ascontiguousarray(orientations, dtype=np.int32)
        n = len(so)
        turned = o >= 5
        ow, oh = np.where(turned, h, w).astype(np.int32), np.where(turned, w, h).astype(np.int32)
        nbytes = ow.astype(np.int64) * oh * 3
        do = np.zeros(n, np.uint64)
        if by_shape:
            order = np.lexsort((oh, ow))
            at, prev = 0, None
            for k in order.tolist():
                shape = (int(ow[k]), int(oh[k]))
                if shape != prev:
                    at, prev = (at + 15) & ~15, shape
                do[k] = at
                at += int(nbytes[k])
            total = at
        else:
            padded = (nbytes + 15) & ~np.int64(15)
            do[1:] = np.cumsum(padded[:-1]).astype(np.uint64)
            total = int(padded.sum())
        dev = self.malloc(total + 64)
        try:
            with self._lock:
                self._check(self._lib.ke_normalise_rgb(self._h, src, _addr(so), _addr(w), _addr(h), _addr(c), _addr(o), n, dev, _addr(do)),
                            "ke_normalise_rgb")
        except Exception:
            self.free(dev)
            raise
        return dev, do, ow, oh

    def thumbnail_rgb(self, src: int, width: int, height: int, box: int, filter: int = 0):
        """An RGB image on the device -> what ``Image.thumbnail((box, box), LANCZOS)`` makes of it, in a device buffer of its own
        (the caller frees it): (device ptr, width, height).  The size is Image.thumbnail's (aspect preserved, PIL/Image.py
        ``preserve_aspect_ratio``); ke_thumbnail_rgb resamples each band as Pillow does."""
        import math

        def round_aspect(number, key):
            return max(min(math.floor(number), math.ceil(number), key=key), 1)

        x = y = box
        aspect = width / height
        if x / y >= aspect:
            x = round_aspect(y * aspect, key=lambda n: abs(aspect - n / y))
        else:
            y = round_aspect(x / aspect, key=lambda n: 0 if n == 0 else abs(aspect - x / n))
        dev = self.malloc(x * y * 3 + 64)
        try:
            with self._lock:
                self._check(self._lib.ke_thumbnail_rgb(self._h, src, width, height, x, y, filter, dev), "ke_thumbnail_rgb")
        except Exception:
            self.free(dev)
            raise
        return dev, x, y

    def release_decode_buffers(self) -> None:
        """Give back the page-locked packing buffer and the device decode buffer (they are kept between calls otherwise)."""
        with self._lock:
            if self._pack_ptr:
                self._check(self._lib.ke_host_free(self._h, self._pack_ptr), "ke_host_free")
                self._pack_ptr, self._pack_cap = 0, 0
            if self._decoded_ptr:
                self.free(self._decoded_ptr)
                self._decoded_ptr, self._decoded_cap = 0, 0
            with self._ahead_lock:
                for buf in self._ahead:
                    if buf[0] and not buf[2]:
                        self._check(self._lib.ke_host_free(self._h, buf[0]), "ke_host_free")
                        buf[0], buf[1] = 0, 0

    def jpeg_decode(self, blobs, kind: str = "jpeg"):
        """Pixels of JPEG files decoded on the GPU: list of ndarrays (HxW or HxWx3, what np.asarray(Image.open(f)) gives) with
        None where the decoder refused the file (status != 0); also returns the statuses."""
        out = [None] * len(blobs)
        with self._lock:
            try:
                dev, out_off, w, h, c, st = self._jpeg_to_device(blobs, kind)
            except _BatchTooLarge:
                half = len(blobs) // 2
                a, sa = self.jpeg_decode(blobs[:half], kind)
                b, sb = self.jpeg_decode(blobs[half:], kind)
                return a + b, np.concatenate([sa, sb])
            for i in range(len(blobs)):
                if st[i] == 0:
                    arr = np.empty((h[i], w[i], c[i]) if c[i] > 1 else (h[i], w[i]), np.uint8)
                    self.memcpy(arr, dev + int(out_off[i]), arr.nbytes)
                    out[i] = arr
        return out, st

    def hash_files(self, paths, *, want_dhash=True, kind: str = "jpeg"):
        """jpeg_hash for files on disk: read (host threads, page-locked buffer), decoded and hashed on the GPU."""
        return self.jpeg_hash(None, want_dhash=want_dhash, kind=kind, paths=list(paths))

    def jpeg_hash(self, blobs, *, want_dhash=True, kind: str = "jpeg", paths=None, ahead=None, skip=None):
        """pHash / dHash of JPEG files, decoded and hashed without the pixels leaving the GPU.  Returns (phash u64[n],
        dhash u64[n] | None, status int32[n]); status != 0 = not handled here (decode the file with Pillow).  The files come
        as bytes (``blobs``), as ``paths`` the library reads, or as ``ahead = (FilesAhead, lo, hi)``: files lo..hi of a batch
        read beforehand."""
        n = ahead[2] - ahead[1] if ahead is not None else len(blobs) if paths is None else len(paths)
        ph = np.zeros(n, np.uint64)
        dh = np.zeros(n, np.uint64) if want_dhash else None
        if n == 0:
            return ph, dh, np.zeros(0, np.int32)
        with self._lock:
            try:
                dev, out_off, w, h, c, st = self._jpeg_to_device(blobs, kind, paths=paths, ahead=ahead, skip=skip)
            except _BatchTooLarge:
                half = n // 2
                parts = [self.jpeg_hash(None if blobs is None else blobs[lo:hi], want_dhash=want_dhash, kind=kind,
                                        paths=None if paths is None else paths[lo:hi],
                                        ahead=None if ahead is None else (ahead[0], ahead[1] + lo, ahead[1] + hi),
                                        skip=None if skip is None else np.asarray(skip, bool)[lo:hi])
                         for lo, hi in ((0, half), (half, n))]
                return (np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]) if want_dhash else None,
                        np.concatenate([p[2] for p in parts]))
            for ch in (1, 3, 4):
                idx = np.nonzero((st == 0) & (c == ch))[0]
                if len(idx) == 0:
                    continue
                p = np.zeros(len(idx), np.uint64)
                d = np.zeros(len(idx), np.uint64) if want_dhash else None
                status = np.zeros(len(idx), np.int32)
                offs = np.ascontiguousarray(out_off[idx])
                ws, hs = np.ascontiguousarray(w[idx]), np.ascontiguousarray(h[idx])
                with self._lock:
                    self._check(self._lib.ke_hash_images(self._h, dev, _addr(offs), _addr(ws), _addr(hs), ch, len(idx), _addr(p), _addr(d),
                                                         _addr(status)), "ke_hash_images")
                ph[idx] = p
                if want_dhash:
                    dh[idx] = d
                st[idx[status != 0]] = 2
        return ph, dh, st

    # -- scan -------------------------------------------------------------------------------
    def hamming_scan(self, hashes, n: int, *, ids=None, sizes=None, threshold=8, band_bits=16, band_count=4,
                     size_ratio: float = 0.0, bucket_pair_cap: int = 0, part_index=0, part_count=1,
                     capacity: Optional[int] = None):
        """Returns (edges[EDGE_DTYPE] on the host, counters u64[4]).  hashes/ids/sizes: host arrays or device ptrs."""
        def host(a, dt):
            return np.ascontiguousarray(a, dtype=dt) if isinstance(a, (np.ndarray, list, tuple)) else a

        hashes, ids, sizes = host(hashes, np.uint64), host(ids, np.int64), host(sizes, np.int64)
        cap = int(capacity) if capacity else max(1 << 16, 2 * int(n))
        counters = np.zeros(4, np.uint64)
        n_edges = C.c_int64(0)
        while True:
            edges = np.empty(cap, EDGE_DTYPE)
            with self._lock:
                self._check(self._lib.ke_hamming_scan(self._h, _addr(hashes), _addr(ids), _addr(sizes), n, part_index,
                                                      part_count, threshold, band_bits, band_count, float(size_ratio),
                                                      int(bucket_pair_cap), _addr(edges), cap, C.byref(n_edges),
                                                      _addr(counters)), "ke_hamming_scan")
            if n_edges.value <= cap:
                return edges[: n_edges.value], counters
            cap = int(n_edges.value)  # overflow protocol: retry with the reported size

    def band_pairs_after_size(self, hashes, sizes, n: int, *, band_bits=16, band_count=4, size_ratio: float, bucket_pair_cap: int = 0) -> int:
        """The reference's "size=" funnel counter before pairs of equal file id are taken out (ke_band_pairs_after_size)."""
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64) if isinstance(hashes, np.ndarray) else hashes
        sizes = np.ascontiguousarray(sizes, dtype=np.int64) if isinstance(sizes, np.ndarray) else sizes
        out = np.zeros(1, np.uint64)
        with self._lock:
            self._check(self._lib.ke_band_pairs_after_size(self._h, _addr(hashes), _addr(sizes), n, band_bits, band_count,
                                                           float(size_ratio), int(bucket_pair_cap), _addr(out)), "ke_band_pairs_after_size")
        return int(out[0])

    def cluster_labels(self, edges: np.ndarray, n_nodes: int) -> np.ndarray:
        edges = np.ascontiguousarray(edges, dtype=EDGE_DTYPE)
        labels = np.empty(n_nodes, np.int64)
        rc = self._lib.ke_cluster_labels(_addr(edges), len(edges), n_nodes, _addr(labels))
        if rc != KE_OK:
            raise ValueError("ke_cluster_labels: edge endpoint outside [0, n_nodes)")
        return labels

    # -- ssim -------------------------------------------------------------------------------
    def ssim_set_mode(self, exact: bool) -> None:
        """False (default): integer-sum kernel, within 1e-5 of skimage's float32 arithmetic; True: the kernel that
        reproduces every rounding (bit-identical to the oracle, 3-4x slower)."""
        self._check(self._lib.ke_ssim_set_mode(self._h, 1 if exact else 0), "ke_ssim_set_mode")

    def ssim_pairs_uniform(self, images, n_images: int, width: int, height: int, channels: int, pair_a, pair_b):
        if isinstance(images, np.ndarray):
            images = np.ascontiguousarray(images, dtype=np.uint8)
        pa = np.ascontiguousarray(pair_a, dtype=np.int64)
        pb = np.ascontiguousarray(pair_b, dtype=np.int64)
        out = np.empty(len(pa), np.float64)
        with self._lock:
            self._check(self._lib.ke_ssim_pairs_uniform(self._h, _addr(images), n_images, width, height, channels, _addr(pa),
                                                        _addr(pb), len(pa), _addr(out)), "ke_ssim_pairs_uniform")
        return out

    def ssim_pairs(self, images: Sequence[np.ndarray], pair_a, pair_b):
        """src/dup/refine.py:44-52 for pairs of images of any sizes (one channel count): common size, ImageOps.fit + BICUBIC
        of both, SSIM.  Returns (float64 scores, NaN where the status is not 0; int32 statuses KE_PAIR_*)."""
        n = len(images)
        chans = {1 if im.ndim == 2 else im.shape[2] for im in images}
        if len(chans) != 1:
            raise ValueError("all images of one call must share the channel count")
        ch = chans.pop()
        widths = np.array([im.shape[1] for im in images], np.int32)
        heights = np.array([im.shape[0] for im in images], np.int32)
        sizes = widths.astype(np.int64) * heights * ch
        padded = (sizes + 15) & ~np.int64(15)
        offsets = np.zeros(n, np.uint64)
        offsets[1:] = np.cumsum(padded[:-1]).astype(np.uint64)
        flat = np.zeros(int(padded.sum()), np.uint8)
        for im, off, sz in zip(images, offsets, sizes):
            flat[int(off):int(off) + int(sz)] = np.ascontiguousarray(im, dtype=np.uint8).reshape(-1)
        pa = np.ascontiguousarray(pair_a, dtype=np.int64)
        pb = np.ascontiguousarray(pair_b, dtype=np.int64)
        out = np.empty(len(pa), np.float64)
        status = np.empty(len(pa), np.int32)
        with self._lock:
            self._check(self._lib.ke_ssim_pairs(self._h, _addr(flat), _addr(offsets), _addr(widths), _addr(heights), ch, n, _addr(pa),
                                                _addr(pb), len(pa), _addr(out), _addr(status)), "ke_ssim_pairs")
        return out, status

    def ssim_pairs_on_device(self, pointers, widths, heights, channels: int, pair_a, pair_b):
        """ssim_pairs for images that already lie in device memory, each at its own address (decoded there, or uploaded)."""
        ptrs = np.asarray(pointers, np.uint64)
        base = int(ptrs.min())
        offsets = np.ascontiguousarray(ptrs - np.uint64(base))
        widths = np.ascontiguousarray(widths, np.int32)
        heights = np.ascontiguousarray(heights, np.int32)
        pa = np.ascontiguousarray(pair_a, dtype=np.int64)
        pb = np.ascontiguousarray(pair_b, dtype=np.int64)
        out = np.empty(len(pa), np.float64)
        status = np.empty(len(pa), np.int32)
        with self._lock:
            self._check(self._lib.ke_ssim_pairs(self._h, base, _addr(offsets), _addr(widths), _addr(heights), channels, len(ptrs), _addr(pa),
                                                _addr(pb), len(pa), _addr(out), _addr(status)), "ke_ssim_pairs")
        return out, status

    # -- shipped refine stage -------------------------------------------------------------------
    def resize_luma_uniform(self, pixels, n: int, width: int, height: int, channels: int, out_w: int, out_h: int,
                            filter: int = 1, out=None):
        """n images -> (n, out_h, out_w) u8 luma thumbnails; filter 0 = LANCZOS, 1 = BILINEAR."""
        if isinstance(pixels, np.ndarray):
            pixels = np.ascontiguousarray(pixels, dtype=np.uint8)
        if out is None:
            out = np.empty((n, out_h, out_w), np.uint8)
        with self._lock:
            self._check(self._lib.ke_resize_luma_uniform(self._h, _addr(pixels), n, width, height, channels, out_w, out_h,
                                                         filter, _addr(out)), "ke_resize_luma_uniform")
        return out

    def fit_luma_uniform(self, pixels, n: int, width: int, height: int, channels: int, out_w: int, out_h: int,
                         filter: int = 2, out=None):
        """n images -> (n, out_h, out_w) u8: ImageOps.fit(image.convert("L"), (out_w, out_h), filter) of Pillow
        (centre crop to the aspect ratio + resize); filter 2 = BICUBIC as src/dup/refine.py:48 uses."""
        if isinstance(pixels, np.ndarray):
            pixels = np.ascontiguousarray(pixels, dtype=np.uint8)
        if out is None:
            out = np.empty((n, out_h, out_w), np.uint8)
        with self._lock:
            self._check(self._lib.ke_fit_luma_uniform(self._h, _addr(pixels), n, width, height, channels, out_w, out_h,
                                                      filter, _addr(out)), "ke_fit_luma_uniform")
        return out

    def tile_ahash(self, tiles, n: int, grid: int, tile: int):
        """(n, side, side) thumbnails -> (n, words) u64, little-endian bit order of tile_ahash_bits."""
        if isinstance(tiles, np.ndarray):
            tiles = np.ascontiguousarray(tiles, dtype=np.uint8)
        side = grid * tile
        words = (side * side + 63) // 64
        out = np.empty((n, words), np.uint64)
        with self._lock:
            self._check(self._lib.ke_tile_ahash(self._h, _addr(tiles), n, grid, tile, _addr(out)), "ke_tile_ahash")
        return out

    def sad_pairs(self, thumbs, n_thumbs: int, pixels: int, pair_a, pair_b):
        if isinstance(thumbs, np.ndarray):
            thumbs = np.ascontiguousarray(thumbs, dtype=np.uint8)
        pa = np.ascontiguousarray(pair_a, dtype=np.int64)
        pb = np.ascontiguousarray(pair_b, dtype=np.int64)
        out = np.empty(len(pa), np.uint64)
        with self._lock:
            self._check(self._lib.ke_sad_pairs(self._h, _addr(thumbs), n_thumbs, pixels, _addr(pa), _addr(pb), len(pa),
                                               _addr(out)), "ke_sad_pairs")
        return out

    # -- synthetic corpus ---------------------------------------------------------------------
    def synth_rgb(self, seed: int, first: int, n: int, width: int, height: int, out=None):
        if out is None:
            out = np.empty((n, height, width, 3), np.uint8)
        with self._lock:
            self._check(self._lib.ke_synth_rgb(self._h, seed, first, n, width, height, _addr(out)), "ke_synth_rgb")
        return out

    def synth_rgb_indexed(self, seed: int, indices, width: int, height: int, out: int) -> None:
        """Corpus images indices[k] -> device memory at ``out`` (k-th image at out + k*w*h*3)."""
        idx = np.ascontiguousarray(indices, dtype=np.int64)
        with self._lock:
            self._check(self._lib.ke_synth_rgb_indexed(self._h, seed, _addr(idx), len(idx), width, height, out), "ke_synth_rgb_indexed")

    def synth_hashes(self, seed: int, n: int, out=None):
        if out is None:
            out = np.empty(n, np.uint64)
        with self._lock:
            self._check(self._lib.ke_synth_hashes(self._h, seed, n, _addr(out)), "ke_synth_hashes")
        return out


_default: dict = {}
_default_lock = threading.Lock()


def get_context(device: int = 0, role: str = "") -> Context:
    """Process-wide context per device (created on first use).  ``role``: a second context of the same device with its own
    stream, lock and scratch (the batch hasher keeps its pinned staging route on one, so that decoder processes are served
    while a long GPU decode call holds the main context)."""
    key = device if not role else (device, role)
    with _default_lock:
        ctx = _default.get(key)
        if ctx is None:
            ctx = _default[key] = Context(device)
        return ctx


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the library (rank 0 calls it and hands the 128 bytes to the other ranks)."""
    lib = load_library()
    buf = (C.c_uint8 * 128)()
    rc = lib.ke_comm_unique_id(buf)
    if rc != KE_OK:
        raise RuntimeError(f"ke_comm_unique_id failed (rc={rc}): {lib.ke_create_error().decode('utf-8', 'replace')}")
    return bytes(buf)


def cluster_labels(edges: np.ndarray, n_nodes: int) -> np.ndarray:
    """Host-only entry point (no device needed): connected-component labels."""
    lib = load_library()
    edges = np.ascontiguousarray(edges, dtype=EDGE_DTYPE)
    labels = np.empty(n_nodes, np.int64)
    rc = lib.ke_cluster_labels(_addr(edges), len(edges), n_nodes, _addr(labels))
    if rc != KE_OK:
        raise ValueError("ke_cluster_labels: edge endpoint outside [0, n_nodes)")
    return labels
