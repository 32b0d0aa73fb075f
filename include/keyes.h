/*
 * keyes.h -- C ABI of libkeyes_hip.so, the MI355X (gfx950) replacement for the
 * src/sig + src/dup hot path of kobato-eyes.
 *
 * The reference is pure Python and has no FFI of its own; each entry point below cites the
 * reference code (paths relative to the reference checkout) whose work it takes over, and
 * INTEGRATION.md shows the ctypes stub a maintainer would add on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success or a negative KE_E* code; ke_last_error(ctx)
 *     returns a human-readable message for the last failure on that context;
 *   - no exceptions cross the boundary, no torch types appear in any signature;
 *   - data pointers may be HOST or DEVICE memory (decided with hipPointerGetAttributes); a pageable host buffer is
 *     copied to a device scratch in chunks of up to 1 GB with hipMemcpyAsync and hashed chunk by chunk (the call blocks);
 *     producers that can write their pixels where they are told -- decoders -- use the pinned staging buffers of
 *     ke_stage_* instead, whose copies overlap the kernels of the previous batch;
 *   - all work of a context is ordered on ONE HIP stream (ke_set_stream / ke_get_stream);
 *     calls return after the work has been ENQUEUED when every pointer is device memory,
 *     and after it has COMPLETED when any output pointer is host memory or the function
 *     returns a count (ke_hamming_scan);
 *   - a context is not re-entrant; use one per thread/device.  The library never falls
 *     back to the CPU: without a gfx950 device ke_create fails.
 */
#ifndef KEYES_H
#define KEYES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KE_ABI_VERSION 1

enum {
    KE_OK = 0,
    KE_EINVAL = -1,     /* bad argument */
    KE_ENODEV = -2,     /* no usable HIP device */
    KE_EHIP = -3,       /* a HIP runtime call failed (see ke_last_error) */
    KE_ENOMEM = -4,
    KE_EUNSUPPORTED = -5
};

/* per-image status written by ke_hash_images (reference: a failed image is dropped,
 * src/core/fastsig.py:36-37) */
enum { KE_IMG_OK = 0, KE_IMG_BAD_SHAPE = 1 };

typedef struct ke_ctx ke_ctx;

/* One candidate edge of the Hamming scan.  a < b are POSITIONS in the hash table handed to
 * ke_hamming_scan (the reference keys edges by file id, src/dup/scanner.py:287-290; the host
 * maps positions to ids).  bands: bit k set <=> band k of x_a ^ x_b is zero (and its bucket is
 * under the pair cap), i.e. the pair meets in bucket (k, value) of src/dup/scanner.py:227-233. */
typedef struct {
    int64_t a;
    int64_t b;
    int32_t h;      /* hamming64(x_a, x_b), src/sig/phash.py:60-63 */
    int32_t bands;
} ke_edge;

/* ---- context ------------------------------------------------------------------------- */
int         ke_abi_version(void);
ke_ctx     *ke_create(int device_id);               /* NULL on failure; see ke_create_error() */
const char *ke_create_error(void);
void        ke_destroy(ke_ctx *ctx);
const char *ke_last_error(ke_ctx *ctx);
int         ke_set_stream(ke_ctx *ctx, void *hip_stream);   /* NULL = the context's own stream */
void       *ke_get_stream(ke_ctx *ctx);
int         ke_synchronize(ke_ctx *ctx);
int         ke_device_info(ke_ctx *ctx, char *name, size_t name_len, int32_t *compute_units,
                           int64_t *total_mem_bytes);

/* device memory helpers for hosts that do not bring their own allocator */
int ke_malloc(ke_ctx *ctx, size_t bytes, void **dev_ptr_out);
int ke_free(ke_ctx *ctx, void *dev_ptr);
/* page-locked host memory (hipHostMalloc): compressed files packed here cross PCIe at link speed and asynchronously, from
 * pageable memory the runtime stages them through its own bounce buffer first.  ke_host_free waits for the context's stream. */
int ke_host_alloc(ke_ctx *ctx, size_t bytes, void **host_ptr_out);
int ke_host_free(ke_ctx *ctx, void *host_ptr);
/* n separate host buffers (srcs[i], sizes[i] bytes) copied to dst + offsets[i] on the host's threads: how a list of files read
 * one by one becomes the single `files` range the decoders take, without an interpreter-level loop over the files. */
int ke_host_pack(uint8_t *dst, const uint8_t *const *srcs, const uint64_t *offsets, const uint64_t *sizes, int64_t n);
/* The files at paths[0..n) read back to back into dst (capacity bytes, normally page-locked) on the host's threads: what
 * `Image.open(path)` of the reference's worker (src/core/fastsig.py:31) does for the bytes, for a whole batch at once.
 * offsets[i] / sizes[i] locate file i; a file that cannot be read has size 0 (the decoders report it as damaged, the caller
 * drops it as the reference drops a file that raises).  *needed_out = bytes the batch needs; when that exceeds capacity nothing
 * is read and KE_ENOMEM comes back, so that the caller can grow the buffer and call again. */
int ke_host_read_files(const char *const *paths, int64_t n, uint8_t *dst, uint64_t capacity, uint64_t *offsets, uint64_t *sizes,
                       uint64_t *needed_out);
int ke_memcpy(ke_ctx *ctx, void *dst, const void *src, size_t bytes);   /* any direction, synchronous */

/* ---- hashing: replaces sig.phash.phash / dhash (src/sig/phash.py:21-57) as driven by
 * core.fastsig._compute_worker (src/core/fastsig.py:24-37) -------------------------------
 * n interleaved 8-bit images.  channels: 1 (mode "L"), 3 (RGB) or 4 (RGBX/RGBA; the fourth
 * byte is ignored, as Pillow's convert("L") ignores it).  Image i starts at
 * pixels + offsets[i] (offsets == NULL: images are packed back to back) and is
 * heights[i] rows of widths[i]*channels bytes, no row padding.  Any byte offset is accepted; the fastest kernels
 * are used for size groups whose images all start on a 4-byte boundary (pad offsets to a multiple of 4 -- the Python
 * host pads to 16 -- when packing a device-resident batch).
 * phash_out / dhash_out: unsigned 64-bit hashes (either may be NULL).  The signed wrap of
 * src/sig/phash.py:29-30 / src/core/fastsig.py:19-21 is the host's reinterpretation.
 * status_out (nullable): KE_IMG_* per image; hashes of failed images are 0.
 *
 * Tuning knob (environment, read at every call): KE_FUSED_MIN_IMAGES = smallest group of equally sized images that is
 * hashed with one workgroup per image; smaller groups are cut into bands of rows so that a few large images still fill
 * the GPU.  Unset: chosen from the image size (about 200 images for 0.4-8 MB images, 512 above).  Results do not depend
 * on it. */
int ke_hash_images(ke_ctx *ctx, const uint8_t *pixels, const uint64_t *offsets, const int32_t *widths,
                   const int32_t *heights, int32_t channels, int64_t n, uint64_t *phash_out,
                   uint64_t *dhash_out, int32_t *status_out);

/* Same, n equally sized images packed back to back (the BASELINE configs). */
int ke_hash_uniform(ke_ctx *ctx, const uint8_t *pixels, int64_t n, int32_t width, int32_t height,
                    int32_t channels, uint64_t *phash_out, uint64_t *dhash_out);

/* The same two calls with one more output: margin_out[i] (float32, host or device, nullable) = min over the 64 DCT
 * coefficients of |coef - mean(AC)| for image i, i.e. the distance of the closest bit decision `coef > mean`
 * (src/sig/phash.py:41-42) from a tie.  pHash bits can only differ between DCT implementations (this library's folded
 * fp64 DCT vs OpenCV's float32 cv2.dct, src/sig/phash.py:38) where this margin is of the order of the implementations'
 * rounding differences (~1e-4 and below); callers count such images to know their exposure.  Needs phash_out. */
int ke_hash_uniform_ex(ke_ctx *ctx, const uint8_t *pixels, int64_t n, int32_t width, int32_t height,
                       int32_t channels, uint64_t *phash_out, uint64_t *dhash_out, float *margin_out);
int ke_hash_images_ex(ke_ctx *ctx, const uint8_t *pixels, const uint64_t *offsets, const int32_t *widths,
                      const int32_t *heights, int32_t channels, int64_t n, uint64_t *phash_out,
                      uint64_t *dhash_out, int32_t *status_out, float *margin_out);

/* ---- pinned staging: the batch boundary of core.fastsig (src/core/fastsig.py:24-37, 65-99: decode in workers, hash,
 * collect in order) as a two-stage pipeline -- north-star step (1), "decodes batches of images to a pinned staging buffer".
 *
 * ke_stage_create allocates n_buffers (1..4) page-locked host buffers of bytes_per_buffer with device twins, result
 * blocks for up to max_images images per batch, and a copy stream.  Per batch:
 *   ke_stage_acquire      -> the next buffer in rotation (blocks only until THAT buffer's previous batch has finished and
 *                            hands its results to the arrays named at its submit); decoders write pixels into it;
 *   ke_stage_submit_hash  -> image i is heights[i] rows of widths[i]*channels[i] bytes at host_ptr + offsets[i] (channels
 *                            per image: 1, 3 or 4; keep offsets multiples of 4, better 16).  Enqueues the H2D copy on the
 *                            copy stream and the hash kernels behind it on the context's stream, then RETURNS: the copy of
 *                            batch k+1 overlaps the kernels of batch k, the host is free to decode into the other buffer.
 *                            phash_out / dhash_out / margin_out (host, nullable) are written when the slot is waited for or
 *                            re-acquired; status_out (KE_IMG_*) is written immediately;
 *   ke_stage_wait         -> blocks until the slot's results are in the caller's arrays (slot -1: every slot, oldest first).
 * Hashes are those of ke_hash_images on the same pixels. */
int ke_stage_create(ke_ctx *ctx, size_t bytes_per_buffer, int64_t max_images, int32_t n_buffers);
/* The same staging on buffers the caller owns: n_buffers page-aligned regions of bytes_per_buffer bytes (a multiple of 256)
 * -- e.g. POSIX shared memory that decoder processes attach to, which is how a Python host gets past its interpreter lock the
 * way the reference does (a process pool, src/core/fastsig.py:83-85).  The library page-locks them (hipHostRegister) until
 * ke_stage_destroy / the next create; the memory must stay mapped that long. */
int ke_stage_create_shared(ke_ctx *ctx, void *const *buffers, size_t bytes_per_buffer, int64_t max_images, int32_t n_buffers);
int ke_stage_destroy(ke_ctx *ctx);
int ke_stage_acquire(ke_ctx *ctx, int32_t *slot_out, uint8_t **host_ptr_out, size_t *bytes_out);
int ke_stage_submit_hash(ke_ctx *ctx, int32_t slot, const uint64_t *offsets, const int32_t *widths, const int32_t *heights,
                         const int32_t *channels, int64_t n, uint64_t *phash_out, uint64_t *dhash_out, int32_t *status_out,
                         float *margin_out);
int ke_stage_wait(ke_ctx *ctx, int32_t slot);

/* ---- JPEG decode on the GPU: `Image.open(path)` + pixel access of the reference's batch hasher (src/core/fastsig.py:31-34;
 * the decode half of north-star step 1) for Huffman-coded JPEGs with 8-bit samples -- sequential files with one interleaved
 * scan and progressive files (spectral selection + successive approximation, any scan script) -- grayscale ("L") or YCbCr at
 * 4:4:4 / 4:2:2 / 4:2:0 / 4:4:0 ("RGB").  The pixels are libjpeg's as Pillow drives it (islow IDCT, fancy upsampling, jdcolor's
 * fixed-point YCbCr -> RGB), bit for bit; everything else (arithmetic coding, 12-bit, CMYK/YCCK, RGB-coded, other samplings,
 * sequential files in several scans, progressive files whose first AC coefficients are not refined to the last bit -- libjpeg
 * smooths those -- and files with a block whose IDCT leaves 16 bits, which no encoder writes and on which libjpeg's C and SIMD
 * routines differ: csrc/ke_jpeg_core.h, ke_idct_islow) is reported KE_JPEG_UNSUPPORTED per file and stays with Pillow,
 * truncated or damaged entropy data KE_JPEG_CORRUPT (Pillow raises on those).  EXIF orientation is not applied -- nor does Image.open.
 *
 * ke_jpeg_probe  : host only.  files + offsets[i] .. + sizes[i] = file i.  widths/heights/channels (1 or 3)/status per file,
 *                  so that the caller can lay out the pixel buffer.
 * ke_jpeg_decode : files in HOST memory (pageable or the pinned buffer of ke_stage_acquire: the compressed bytes are what
 *                  crosses PCIe), pixels_out in DEVICE memory: image i is written packed at pixels_out + out_offsets[i]
 *                  (heights[i] * widths[i] * channels[i] bytes) when status_out[i] == KE_JPEG_OK and left untouched
 *                  otherwise.  One thread per image decodes the entropy-coded segment, then one thread per 8x8 block runs the
 *                  IDCT and one per 4 pixels upsampling + colour: throughput comes from the batch (thousands of files per
 *                  call).  Blocks until the statuses are back. */
enum { KE_JPEG_OK_ = 0, KE_JPEG_UNSUPPORTED_ = 1, KE_JPEG_CORRUPT_ = 2 };   /* = KE_JPEG_OK / _UNSUPPORTED / _CORRUPT of the library */
int ke_jpeg_probe(const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n, int32_t *widths,
                  int32_t *heights, int32_t *channels, int32_t *status_out);
int ke_jpeg_decode(ke_ctx *ctx, const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n,
                   uint8_t *pixels_out, const uint64_t *out_offsets, int32_t *status_out);

/* ---- PNG decode on the GPU: the same step for PNG files, Adam7-interlaced or not: 8-bit grayscale ("L"), RGB and RGBA pixels as
 * they are; palette files (1 / 2 / 4 / 8 bits) and grayscale of 1 / 2 / 4 bits as the luma `convert("L")` makes of them --
 * what the reference's hashes see for such a file (src/sig/phash.py:25) -- with channels = 1.  Lossless, hence the pixels of
 * Image.open by construction (zlib/deflate, the five scanline filters).  One thread per image walks the deflate stream
 * (literals to their place, LZ77 copies recorded), one wave per image then makes the copies and one wave per image undoes the
 * filters and checks the stream's Adler-32; throughput comes from the batch.  Chunk CRCs are verified where Pillow verifies
 * them (every chunk but IDAT).  An interlaced file is seven reduced images one after another in the stream: the same kernels,
 * the unfilter pass run once per reduced image and its pixels scattered to their places.  8-bit gray + alpha files decode to
 * their gray samples, what convert("L") makes of mode "LA".  Of an animated PNG frame 0 is decoded -- the IDAT image, what Image.open
 * shows (acTL once, at most one fcTL in front of IDAT and that one for the whole image; other forms: status 1).  16-bit files decode to the 8-bit pixels Pillow opens them to: the
 * samples' high bytes for RGB / RGBA, RGBA (L, L, L, A) for gray + alpha, and for grayscale (Pillow: mode "I;16") the value
 * clipped to 255 that convert("L") / convert("RGB") make of it.  Rows wider than 16384 pixels (8192 at 16 bits):
 * KE_JPEG_UNSUPPORTED_ (1) per file, damaged ones KE_JPEG_CORRUPT_ (2).  Arguments and conventions as ke_jpeg_probe /
 * ke_jpeg_decode; channels is 1, 3 or 4. */
int ke_png_probe(const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n, int32_t *widths,
                 int32_t *heights, int32_t *channels, int32_t *status_out);
int ke_png_decode(ke_ctx *ctx, const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n,
                  uint8_t *pixels_out, const uint64_t *out_offsets, int32_t *status_out);

/* ---- BMP files unpacked on the GPU: the same step for the uncompressed files Pillow's BmpImagePlugin opens with its "raw"
 * decoder (src/dup/scanner.py:16-28 ranks the format among the keepers): 24-bit BGR and 32-bit BGRX as RGB, the 32-bit
 * BITFIELDS layouts the plugin lists (byte permutations; RGBA where one of them is alpha), 8-bit palette files as the luma
 * `convert("L")` makes of them -- what the reference's hashes see (src/sig/phash.py:25) -- with channels = 1.  The header is
 * read as BmpImageFile._bitmap reads it (defaults for the colour count and the data offset, padded bottom-up rows unless the
 * height's top byte is 0xFF); the files go to the device as they are and one kernel writes packed top-down rows.  OS/2
 * headers, RLE, 1 / 4 / 16 bits, other masks: KE_JPEG_UNSUPPORTED_ (1) per file; pixel data that ends early:
 * KE_JPEG_CORRUPT_ (2) (Pillow: "image file is truncated").  Arguments and conventions as ke_jpeg_probe / ke_jpeg_decode;
 * channels is 1, 3 or 4.  ke_bmp_caveats reports no flags (the format has no orientation tag; alpha shows as channels = 4). */
int ke_bmp_probe(const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n, int32_t *widths,
                 int32_t *heights, int32_t *channels, int32_t *status_out);
int ke_bmp_decode(ke_ctx *ctx, const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n,
                  uint8_t *pixels_out, const uint64_t *out_offsets, int32_t *status_out);
int ke_bmp_caveats(const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n, int32_t *flags_out);

/* ---- GIF files on the GPU: the same step for the first frame of a GIF file -- what Image.open shows, also of an animation --
 * as the luma `convert("L")` makes of it (src/sig/phash.py:25: what the reference's hashes see), channels = 1: palette index ->
 * luma of the local or global palette entry (no palette or an identity gray ramp: the index).  The container is walked as
 * GifImagePlugin walks it, the LZW stream read as Pillow's GifDecode.c reads it (whole sub-blocks only; over when the last
 * pixel is written); one thread per image walks the codes (a string is a copy from earlier output, recorded), one wave per
 * image makes the copies, a last kernel maps indices to luma and puts interlaced rows in place.  A first frame that does not
 * cover the logical screen, code sizes outside 2..8, an end code or the end of the block list before the last pixel, frames
 * of more than 2^23 pixels: KE_JPEG_UNSUPPORTED_ (1) per file (Pillow decides); data that ends early: KE_JPEG_CORRUPT_ (2).
 * Arguments and conventions as ke_jpeg_probe / ke_jpeg_decode.  For the hashing seams only: the reference's SSIM loader turns a
 * palette file into RGB, not luma.  ke_gif_caveats reports no flags. */
int ke_gif_probe(const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n, int32_t *widths,
                 int32_t *heights, int32_t *channels, int32_t *status_out);
int ke_gif_decode(ke_ctx *ctx, const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n,
                  uint8_t *pixels_out, const uint64_t *out_offsets, int32_t *status_out);
int ke_gif_caveats(const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n, int32_t *flags_out);

/* ---- TIFF files unpacked on the GPU: the same step for the uncompressed 8-bit files Pillow's TiffImagePlugin opens with its own
 * "raw" decoder (compressed TIFF is libtiff's there and stays with Pillow): gray (BlackIsZero; WhiteIsZero inverted), RGB, RGB
 * with unassociated alpha (RGBA) or an unspecified fourth sample (dropped), palette files as the luma `convert("L")` makes of
 * them (src/sig/phash.py:25), channels = 1.  The first directory is read as ImageFileDirectory_v2.load reads it (both byte
 * orders), the layout derived as TiffImageFile._setup derives it (strips of RowsPerStrip rows); the files go to the device as
 * they are and one kernel gathers the strips into packed rows.  A whitelist: other compression / planar layout / fill order /
 * bit depths / sample formats, tiles, an orientation other than 1, EXIF / GPS directories or an XMP packet (Pillow applies an
 * orientation found there at load time), a strip count that does not match the rows: KE_JPEG_UNSUPPORTED_ (1) per file; a
 * strip that ends behind the file: KE_JPEG_CORRUPT_ (2).  Arguments and conventions as ke_jpeg_probe / ke_jpeg_decode. */
int ke_tiff_probe(const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n, int32_t *widths,
                  int32_t *heights, int32_t *channels, int32_t *status_out);
int ke_tiff_decode(ke_ctx *ctx, const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n,
                   uint8_t *pixels_out, const uint64_t *out_offsets, int32_t *status_out);
int ke_tiff_caveats(const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n, int32_t *flags_out);

/* What `Image.open` alone does not tell about a file but the reference's defensive loader acts on (src/utils/image_io.py:60-138:
 * EXIF orientation applied, alpha composited over white): per file a set of KE_CAVEAT_* bits, so that a caller who wants that
 * loader's pixels sends flagged files through it and only the rest through ke_jpeg_decode / ke_png_decode.  ORIENTATION: the
 * file carries an EXIF orientation of 2..8 (JPEG APP1; PNG eXIf or a "Raw profile type ..." text chunk -- or EXIF data that
 * cannot be followed); TRANSPARENCY: a PNG tRNS chunk.  Host only. */
enum { KE_CAVEAT_ORIENTATION = 1, KE_CAVEAT_TRANSPARENCY = 2 };
/* ke_jpeg_caveats also reports the orientation itself in bits 8..11 (1..8) when the tag could be followed: (flags >> 8) & 15. */
int ke_jpeg_caveats(const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n, int32_t *flags_out);
int ke_png_caveats(const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n, int32_t *flags_out);

/* The loader's normalisation itself, for pixels that are on the device already (decoded by ke_jpeg_decode / ke_png_decode):
 * image k (widths[k] x heights[k], channels[k] = 3 or 4, at src + src_offsets[k]) -> RGB at dst + dst_offsets[k], turned as
 * ImageOps.exif_transpose turns it for orientations[k] in 1..8 (src/utils/image_io.py:116-120, src/ui/dup_refine_parallel.py:67-70;
 * 5..8 swap width and height) and, for 4 channels, composited over white as Image.alpha_composite + convert("RGB") do
 * (src/utils/image_io.py:137-151) -- Pillow's integer arithmetic, bit for bit.  src, dst: device; the arrays: host.  Blocks. */
int ke_normalise_rgb(ke_ctx *ctx, const uint8_t *src, const uint64_t *src_offsets, const int32_t *widths, const int32_t *heights,
                     const int32_t *channels, const int32_t *orientations, int64_t n, uint8_t *dst, const uint64_t *dst_offsets);

/* The loader's shrink for an image with a side over 4096 (src/utils/image_io.py:122-124: img.thumbnail((4096, 4096), LANCZOS)),
 * likewise on the device: RGB (width x height at src) -> RGB (out_w x out_h at dst), each band through the resampler that serves
 * the hashes (Pillow's two-pass integer resample, bit-exact); filter as ke_resize_luma_uniform.  The caller computes
 * (out_w, out_h) as Image.thumbnail does (aspect preserved).  src and dst are device memory. */
int ke_thumbnail_rgb(ke_ctx *ctx, const uint8_t *src, int32_t width, int32_t height, int32_t out_w, int32_t out_h, int32_t filter,
                     uint8_t *dst);

/* Debug/parity hook: the resampled luma tiles the hashes are computed from
 * (reference sig.phash._to_grayscale, src/sig/phash.py:21-26).  tile32_out: n*1024 bytes
 * ([y][x]); tile98_out: n*72 bytes (8 rows x 9 columns); either may be NULL. */
int ke_luma_tiles_uniform(ke_ctx *ctx, const uint8_t *pixels, int64_t n, int32_t width, int32_t height,
                          int32_t channels, uint8_t *tile32_out, uint8_t *tile98_out);

/* ---- candidate scan: replaces the bucket + pair loop of DuplicateScanner.build_clusters
 * (src/dup/scanner.py:227-299) with an all-pairs tiled popcount scan whose predicate is the
 * closed form of that loop:
 *   edge(i,j), i<j  <=>  ids[i] != ids[j]  and  popcount(x_i ^ x_j) <= threshold
 *                        and some band k < band_count has ((x_i ^ x_j) >> k*band_bits) & mask == 0
 *                            (and, with bucket_pair_cap > 0, that bucket holds <= cap pairs)
 *                        and size_ok(sizes[i], sizes[j], size_ratio)   (src/dup/scanner.py:358-370)
 * ids / sizes may be NULL (ids default to positions; no size filter).  size_ratio <= 0 = off.
 * The pair space is dealt tile-by-tile to `part_count` shards; this call evaluates shard
 * `part_index` (one shard per GPU; 0/1 = everything).
 * edges_out: capacity entries (host or device); *n_edges_out = edges found, which may exceed
 * capacity -- the caller then retries with a larger buffer.  Edge order is unspecified.
 * counters_out (nullable, 4 x u64, host): [0] pairs i < j evaluated by this shard (counted by the kernel, one atomic per
 * tile, and checked against the host's closed form: the call fails if they differ),
 * [1] sum over emitted edges of the number of shared bands (the reference's "ham=" funnel
 * counter, src/dup/scanner.py:292-299), [2] edges emitted, [3] bucket pairs of the WHOLE table (the same on every
 * shard): sum over bands and band values of C(bucket size, 2) for the buckets the reference walks (size >= 2, under
 * the pair cap) -- its "pairs total=" counter before pairs of equal file id are taken out (src/dup/scanner.py:258-270;
 * the host subtracts those, they need the ids grouped).  With no size filter "size=" equals it. */
int ke_hamming_scan(ke_ctx *ctx, const uint64_t *hashes, const int64_t *ids, const int64_t *sizes, int64_t n,
                    int32_t part_index, int32_t part_count, int32_t threshold, int32_t band_bits,
                    int32_t band_count, double size_ratio, int64_t bucket_pair_cap, ke_edge *edges_out,
                    int64_t capacity, int64_t *n_edges_out, uint64_t *counters_out);

/* The reference's "size=" funnel counter (src/dup/scanner.py:268-270 with _passes_size_ratio, :358-370): over every band
 * and band value whose bucket the reference walks (>= 2 members, under the pair cap), the member pairs that pass the size
 * filter -- pairs of equal file id included (the host takes those out, as for counters_out[3] of ke_hamming_scan).  A log
 * figure, not part of the candidate set: asked for separately so that a scan nobody watches does not pay for it.
 * size_ratio must be > 0.  hashes / sizes: host or device.  Blocks. */
int ke_band_pairs_after_size(ke_ctx *ctx, const uint64_t *hashes, const int64_t *sizes, int64_t n, int32_t band_bits,
                             int32_t band_count, double size_ratio, int64_t bucket_pair_cap, uint64_t *count_out);

/* Connected components of the candidate graph (DisjointSet, src/dup/scanner.py:176-200,
 * 304-318; ClusterBuilder union-find, src/dup/cluster.py:30-46).  HOST function, no device
 * work: label_out[v] = smallest node of v's component for v < n_nodes. */
int ke_cluster_labels(const ke_edge *edges, int64_t n_edges, int64_t n_nodes, int64_t *label_out);

/* ---- multi-GPU exchange on RCCL (xGMI): one process per GPU, image i on rank i mod world.  The reference has no
 * counterpart (it is a single process, src/core/fastsig.py:81-85 is its only parallelism); these are the two exchange
 * steps of the sharded path: the hash table before the scan, the edge lists after it.  RCCL (librccl.so.1) is bound at
 * first use; a process that already carries one -- PyTorch's -- is joined.  `comm` is an ncclComm_t as void*: made by
 * ke_comm_create from a 128-byte unique id (ke_comm_unique_id on rank 0, distributed by the host), or any communicator
 * of the calling process whose ranks are the ranks of this job.  All of it is ordered on the context's stream.
 *
 * ke_allgather_u64   : plain ncclAllGather, n_local entries per rank, rank order (device pointers).
 * ke_allgather_hashes: every rank passes its ceil(n_total / world) local hashes (image rank + k * world at slot k, padded);
 *                      table[i] = hash of image i for i < n_total on every rank: ONE all-gather + a reorder kernel.
 * ke_allgather_edges : per-rank edge lists of ke_hamming_scan (device, n_local valid records) -> all edges of all ranks in
 *                      a host array, rank by rank; *n_total_out may exceed capacity (then only whole records that fit are
 *                      written; retry larger).  counts_out (nullable, world entries): edges per rank.  ONE all-gather of
 *                      fixed-width records [count | first K edges] and one device-to-host copy; K follows the largest list
 *                      seen on this context, a second gather happens only when a list outgrows it.  Blocks. */
int ke_comm_unique_id(uint8_t *id_out /* 128 bytes */);
int ke_comm_create(ke_ctx *ctx, const uint8_t *unique_id, int32_t world, int32_t rank, void **comm_out);
int ke_comm_destroy(ke_ctx *ctx, void *comm);
int ke_allgather_u64(ke_ctx *ctx, void *comm, int32_t world, const uint64_t *local, int64_t n_local, uint64_t *gathered);
int ke_allgather_hashes(ke_ctx *ctx, void *comm, int32_t world, const uint64_t *local, int64_t n_total, uint64_t *table);
/* the reorder step on its own, for hosts that gather with another library: gathered = world shards of ceil(n_total / world)
 * entries in rank order (device), table[i] = gathered[(i mod world) * per + i / world] */
int ke_interleave_shards(ke_ctx *ctx, const uint64_t *gathered, int32_t world, int64_t n_total, uint64_t *table);
int ke_allgather_edges(ke_ctx *ctx, void *comm, int32_t world, const ke_edge *local, int64_t n_local, ke_edge *merged_out,
                       int64_t capacity, int64_t *n_total_out, int64_t *counts_out);

/* ---- SSIM refine: replaces dup.refine._compute_ssim (src/dup/refine.py:44-52 ->
 * skimage.metrics.structural_similarity, 7x7 uniform window, float32, data_range 1) for
 * pairs of equally sized images.  images: n_images interleaved images of width x height x
 * channels (1, 3 or 4; luma is taken exactly as convert("L") does), packed back to back.
 * pair_a/pair_b index into them.  ssim_out[k] is NaN when width or height < 7. */
/* Two kernels compute it.  KE_SSIM_FAST (default): every 7x7 window sum is an exact integer (luma 0..255) carried in
 * float32, variances formed without cancellation error -- equal to skimage's float32 arithmetic up to what skimage's own
 * intermediates round away (measured |delta| < 1e-5; the bar of BASELINE.json is 1e-4); images of fewer than 4096
 * windows ((w-6)*(h-6)) always take the exact kernel, where a single window's rounding is not averaged out.  KE_SSIM_EXACT: every float32 /
 * float64 rounding of scipy.ndimage.uniform_filter and of skimage's elementwise steps reproduced one by one, bit-identical
 * to the oracle, 3-4x slower. */
enum { KE_SSIM_FAST = 0, KE_SSIM_EXACT = 1 };
int ke_ssim_set_mode(ke_ctx *ctx, int32_t mode);
int ke_ssim_pairs_uniform(ke_ctx *ctx, const uint8_t *images, int64_t n_images, int32_t width,
                          int32_t height, int32_t channels, const int64_t *pair_a, const int64_t *pair_b,
                          int64_t n_pairs, double *ssim_out);

/* The whole of dup.refine._compute_ssim (src/dup/refine.py:44-52) for pairs of images of ANY sizes: per pair the common
 * size (min width, min height) (src/dup/refine.py:45-47), ImageOps.fit(convert("L"), size, BICUBIC) of both
 * (ke_fit_luma_uniform's arithmetic), then the SSIM above.  Image i: heights[i] rows of widths[i]*channels bytes at
 * pixels + offsets[i] (offsets NULL: packed back to back); pixels host or device, everything else host arrays.
 * Work is grouped: one fit launch per (source size, common size), one SSIM launch per common size.
 * ssim_out[k]: the score, NaN where status_out[k] != KE_PAIR_OK.  status_out (nullable): KE_PAIR_OK, KE_PAIR_TOO_SMALL (the
 * common size is under 7 pixels: skimage raises "win_size exceeds image extent"), KE_PAIR_BAD_IMAGE (an index outside
 * [0, n_images) or an image with a non-positive size). */
enum { KE_PAIR_OK = 0, KE_PAIR_TOO_SMALL = 1, KE_PAIR_BAD_IMAGE = 2 };
int ke_ssim_pairs(ke_ctx *ctx, const uint8_t *pixels, const uint64_t *offsets, const int32_t *widths, const int32_t *heights,
                  int32_t channels, int64_t n_images, const int64_t *pair_a, const int64_t *pair_b, int64_t n_pairs,
                  double *ssim_out, int32_t *status_out);

/* ---- shipped refine stage ("next" row of SURVEY 8f): replaces the per-file work of
 * ui.dup_refine_parallel -- tile_ahash_bits (src/ui/dup_refine_parallel.py:59-83), _load_small_gray
 * (:203-207) and _mae01 (:208-210).
 *
 * ke_resize_luma_uniform: n equally sized images -> n luma thumbnails of out_h rows x out_w bytes, exactly
 * image.convert("L").resize((out_w, out_h), filter) of Pillow; filter 0 = LANCZOS (src/sig/phash.py:24),
 * 1 = BILINEAR (src/ui/dup_refine_parallel.py:70, :204), 2 = BICUBIC (src/dup/refine.py:48).
 * ke_tile_ahash: thumbnails of side grid*tile -> ceil((grid*tile)^2 / 64) little-endian u64 words per image;
 * bit i (order gy, gx, ty, tx) = pixel > mean of its tile.
 * ke_sad_pairs: sum |a - b| over two thumbnails of `pixels` bytes; MAE = sad / pixels / 255 on the host. */
enum { KE_FILTER_LANCZOS = 0, KE_FILTER_BILINEAR = 1, KE_FILTER_BICUBIC = 2 };
int ke_resize_luma_uniform(ke_ctx *ctx, const uint8_t *pixels, int64_t n, int32_t width, int32_t height,
                           int32_t channels, int32_t out_w, int32_t out_h, int32_t filter, uint8_t *tiles_out);
/* ke_fit_luma_uniform: ImageOps.fit(image.convert("L"), (out_w, out_h), filter) of Pillow for n equally sized
 * images -- centre crop to the target aspect ratio (default bleed 0, centering 0.5/0.5; the crop box is formed
 * in Python's float arithmetic and handed to the resampler in single precision, as Pillow does), then
 * Image.resize(size, filter, box=crop).  This is how src/dup/refine.py:44-49 brings two images of different
 * size to the common (min width, min height) before SSIM; filter 2 = BICUBIC is the reference's choice.
 * A target equal to the input size is a plain luma conversion. */
int ke_fit_luma_uniform(ke_ctx *ctx, const uint8_t *pixels, int64_t n, int32_t width, int32_t height,
                        int32_t channels, int32_t out_w, int32_t out_h, int32_t filter, uint8_t *tiles_out);
int ke_tile_ahash(ke_ctx *ctx, const uint8_t *tiles, int64_t n, int32_t grid, int32_t tile, uint64_t *bits_out);
int ke_sad_pairs(ke_ctx *ctx, const uint8_t *thumbs, int64_t n_thumbs, int64_t pixels, const int64_t *pair_a,
                 const int64_t *pair_b, int64_t n_pairs, uint64_t *sad_out);

/* ---- synthetic corpus (BASELINE configs; DESIGN.md "Synthetic data") --------------------
 * Writes images [first_index, first_index+n) of the counter-based corpus as packed RGB into
 * device or host memory; ke_synth_hashes writes the scan-only hash table. */
int ke_synth_rgb(ke_ctx *ctx, uint64_t seed, int64_t first_index, int64_t n, int32_t width, int32_t height,
                 uint8_t *rgb_out);
/* The same generator for an arbitrary list of corpus positions (host or device array): image k of the output is
 * corpus image indices[k].  A rank that has to evaluate a pair whose images live on another rank regenerates them
 * this way (SSIM stage); rgb_out must be device memory. */
int ke_synth_rgb_indexed(ke_ctx *ctx, uint64_t seed, const int64_t *indices, int64_t n, int32_t width, int32_t height,
                         uint8_t *rgb_out);
int ke_synth_hashes(ke_ctx *ctx, uint64_t seed, int64_t n, uint64_t *hashes_out);

/* ---- timing hook for bench.py: wall time of the kernels enqueued by the LAST call of the
 * named kind on this context, measured with hipEvents on the context's stream.
 * kind: 0 = hash kernel(s), 1 = scan kernel, 2 = ssim kernel, 3 = synth kernel, 4 = JPEG / PNG decode kernels.
 * Returns milliseconds, or a negative value if nothing was recorded.  Blocks until done. */
double ke_last_kernel_ms(ke_ctx *ctx, int32_t kind);

#ifdef __cplusplus
}
#endif
#endif /* KEYES_H */
