/*
 * keyes_oracle.c -- CPU restatement of kobato-eyes' src/sig + src/dup hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under kobato-eyes_amd/ may import, link or
 * call this file; it is the checker for tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  All paths cited are relative to /root/reference.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - luma + LANCZOS resample : pinned bit-exact against Pillow (installed here) and
 *     against reference sig.phash._to_grayscale (src/sig/phash.py:21-26) via
 *     tests/golden/sig_golden.npz.
 *   - dHash                   : pinned bit-exact (reference src/sig/phash.py:49-57 ran here).
 *   - pHash                   : DCT is third-party (OpenCV 4.12, not installed, not in the
 *     reference tree) -> "parity unpinned" vs real cv2.dct; pinned against the reference
 *     phash() run here with a SciPy orthonormal DCT-II stand-in for cv2.dct.
 *   - banded Hamming scan      : pinned bit-exact against reference dup.scanner
 *     (src/dup/scanner.py:211-299) run here, tests/golden/scan_golden.json.
 *   - SSIM                     : third-party (scikit-image 0.25.2, not installed) -> "parity
 *     unpinned" beyond tests/dup/test_refine.py:24-46; cross-checked against a
 *     scipy.ndimage.uniform_filter restatement of the published algorithm.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "dct_table.h"

#define KO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* a1: image.convert("L")  -- Pillow rgb2l, called from src/sig/phash.py:25    */
/* ------------------------------------------------------------------------- */
KO_API void ko_luma(const uint8_t *px, int64_t n_pixels, int channels, uint8_t *out) {
    if (channels == 1) {
        memcpy(out, px, (size_t)n_pixels);
        return;
    }
    for (int64_t i = 0; i < n_pixels; i++) {
        const uint8_t *p = px + i * channels;
        out[i] = (uint8_t)((19595u * p[0] + 38470u * p[1] + 7471u * p[2] + 0x8000u) >> 16);
    }
}

/* ------------------------------------------------------------------------- */
/* a1: .resize(size, LANCZOS) -- Pillow ImagingResample (8 bits per channel),  */
/* called from src/sig/phash.py:25.  Coefficients are 22-bit fixed point.      */
/* ------------------------------------------------------------------------- */
#define KO_PRECISION_BITS 22

static double ko_sinc(double x) {
    if (x == 0.0) return 1.0;
    x *= M_PI;
    return sin(x) / x;
}

static double ko_lanczos3(double x) {
    if (-3.0 <= x && x < 3.0) return ko_sinc(x) * ko_sinc(x / 3.0);
    return 0.0;
}

/* Pillow BILINEAR ("triangle", support 1): the resize used by the shipped refine stage,
 * src/ui/dup_refine_parallel.py:70 (32x32 tile-aHash thumbnail) and :204 (128x128 MAE thumbnail). */
static double ko_triangle(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}

/* Pillow BICUBIC (Keys cubic convolution, a = -0.5, support 2): the resize inside ImageOps.fit at
 * src/dup/refine.py:48-49 (SSIM of two images of different size). */
static double ko_bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

static double ko_filter(int filter, double x) {
    return filter == 1 ? ko_triangle(x) : filter == 2 ? ko_bicubic(x) : ko_lanczos3(x);
}

/* Fills bounds[2*out_size] = (first tap, tap count) and kk[out_size*ksize] for the source interval
 * [in0, in1) of an axis of in_size samples (Pillow's `box`; single precision, as the C entry point
 * receives it).  Returns ksize, or <0 on allocation failure.  Caller frees *bounds_p, *kk_p. */
static int ko_axis_coeffs(int in_size, float in0, float in1, int out_size, int filter, int32_t **bounds_p, int32_t **kk_p);

KO_API int ko_lanczos_coeffs(int in_size, int out_size, int32_t **bounds_p, int32_t **kk_p) {
    return ko_axis_coeffs(in_size, 0.0f, (float)in_size, out_size, 0, bounds_p, kk_p);
}

/* filter: 0 = LANCZOS, 1 = BILINEAR, 2 = BICUBIC */
static int ko_axis_coeffs(int in_size, float in0, float in1, int out_size, int filter, int32_t **bounds_p, int32_t **kk_p) {
    double scale = (double)(in1 - in0) / (double)out_size;
    double filterscale = scale < 1.0 ? 1.0 : scale;
    double support = (filter == 1 ? 1.0 : filter == 2 ? 2.0 : 3.0) * filterscale;
    int ksize = (int)ceil(support) * 2 + 1;
    int32_t *bounds = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)out_size);
    int32_t *kk = (int32_t *)calloc((size_t)out_size * (size_t)ksize, sizeof(int32_t));
    double *w = (double *)malloc(sizeof(double) * (size_t)ksize);
    if (!bounds || !kk || !w) {
        free(bounds); free(kk); free(w);
        return -1;
    }
    double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; xx++) {
        double center = in0 + (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; x++) {
            w[x] = ko_filter(filter, (x + xmin - center + 0.5) * ss);
            ww += w[x];
        }
        int32_t *k = kk + (size_t)xx * ksize;
        for (int x = 0; x < xmax; x++) {
            double v = (ww != 0.0) ? w[x] / ww : w[x];
            k[x] = v < 0 ? (int32_t)(-0.5 + v * (double)(1 << KO_PRECISION_BITS))
                         : (int32_t)(0.5 + v * (double)(1 << KO_PRECISION_BITS));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    free(w);
    *bounds_p = bounds;
    *kk_p = kk;
    return ksize;
}

KO_API void ko_free(void *p) { free(p); }

static inline uint8_t ko_clip8(int32_t v) {
    v >>= KO_PRECISION_BITS; /* arithmetic shift, as Pillow's clip8 lookup index */
    return v < 0 ? 0 : (v > 255 ? 255 : (uint8_t)v);
}

/* L (h x w) -> out (oh x ow).  Horizontal pass first (only the rows the vertical
 * pass reads), u8 clip after each pass; a pass whose size is unchanged is skipped. */
static int ko_resample_2pass(const uint8_t *L, int w, int h, int ow, int oh, uint8_t *out, int filter, const float *box);
KO_API int ko_resample(const uint8_t *L, int w, int h, int ow, int oh, int filter, uint8_t *out);
KO_API int ko_resample_box(const uint8_t *L, int w, int h, int ow, int oh, int filter, const float *box, uint8_t *out);

KO_API int ko_resample_lanczos(const uint8_t *L, int w, int h, int ow, int oh, uint8_t *out) {
    return ko_resample(L, w, h, ow, oh, 0, out);
}

KO_API int ko_resample(const uint8_t *L, int w, int h, int ow, int oh, int filter, uint8_t *out) {
    const float box[4] = {0.0f, 0.0f, (float)w, (float)h};
    return ko_resample_box(L, w, h, ow, oh, filter, box, out);
}

/* Image.resize(size, filter, box) on an "L" image (PIL/Image.py + ImagingResample). */
KO_API int ko_resample_box(const uint8_t *L, int w, int h, int ow, int oh, int filter, const float *box, uint8_t *out) {
    if (w <= 0 || h <= 0 || ow <= 0 || oh <= 0) return -2;
    if (box[0] < 0 || box[1] < 0 || box[2] > w || box[3] > h || box[2] - box[0] < 0 || box[3] - box[1] < 0) return -2;
    /* Pillow's Image.resize (PIL/Image.py, observed in 12.2.0): very tall, narrow images
     * shrink vertically first, as two separate single-axis resizes. */
    if ((int64_t)h > (int64_t)w * 100 && oh < h) {
        uint8_t *mid = (uint8_t *)malloc((size_t)w * (size_t)oh);
        if (!mid) return -1;
        const float b1[4] = {0.0f, box[1], (float)w, box[3]}, b2[4] = {box[0], 0.0f, box[2], (float)oh};
        int rc = ko_resample_2pass(L, w, h, w, oh, mid, filter, b1);
        if (rc == 0) rc = ko_resample_2pass(mid, w, oh, ow, oh, out, filter, b2);
        free(mid);
        return rc;
    }
    return ko_resample_2pass(L, w, h, ow, oh, out, filter, box);
}

/* ImageOps.fit(image, (ow, oh), method) with the default bleed = 0 and centering = (0.5, 0.5): the crop box
 * in Python's float arithmetic (PIL/ImageOps.py), then Image.resize(size, method, box=crop), which is a copy
 * when nothing changes.  src/dup/refine.py:48-49. */
KO_API void ko_fit_box(int w, int h, int ow, int oh, float *box) {
    const double live_ratio = (double)w / (double)h, out_ratio = (double)ow / (double)oh;
    double cw, chh;
    if (live_ratio == out_ratio) { cw = w; chh = h; }
    else if (live_ratio >= out_ratio) { cw = out_ratio * h; chh = h; }
    else { cw = w; chh = w / out_ratio; }
    const double left = (w - cw) * 0.5, top = (h - chh) * 0.5;
    box[0] = (float)left; box[1] = (float)top; box[2] = (float)(left + cw); box[3] = (float)(top + chh);
}

KO_API int ko_fit_luma(const uint8_t *L, int w, int h, int ow, int oh, int filter, uint8_t *out) {
    float box[4];
    ko_fit_box(w, h, ow, oh, box);
    return ko_resample_box(L, w, h, ow, oh, filter, box, out);
}

static int ko_resample_2pass(const uint8_t *L, int w, int h, int ow, int oh, uint8_t *out, int filter, const float *box) {
    int need_h = (ow != w) || box[0] != 0.0f || box[2] != (float)ow;
    int need_v = (oh != h) || box[1] != 0.0f || box[3] != (float)oh;
    int32_t *bh = NULL, *kh = NULL, *bv = NULL, *kv = NULL;
    int ksh = 0, ksv = 0;
    if (need_h && (ksh = ko_axis_coeffs(w, box[0], box[2], ow, filter, &bh, &kh)) < 0) return -1;
    if (need_v && (ksv = ko_axis_coeffs(h, box[1], box[3], oh, filter, &bv, &kv)) < 0) { free(bh); free(kh); return -1; }
    const uint8_t *src = L;
    uint8_t *tmp = NULL;
    int src_w = w;
    if (need_h) {
        int y0 = 0, y1 = h;
        if (need_v) { y0 = bv[0]; y1 = bv[2 * (oh - 1)] + bv[2 * (oh - 1) + 1]; }
        uint8_t *dst = need_v ? (tmp = (uint8_t *)malloc((size_t)ow * (size_t)h)) : out;
        if (!dst) { free(bh); free(kh); free(bv); free(kv); return -1; }
        for (int y = y0; y < y1; y++) {
            const uint8_t *row = L + (size_t)y * w;
            for (int xx = 0; xx < ow; xx++) {
                int xmin = bh[2 * xx], cnt = bh[2 * xx + 1];
                const int32_t *k = kh + (size_t)xx * ksh;
                int32_t acc = 1 << (KO_PRECISION_BITS - 1);
                for (int x = 0; x < cnt; x++) acc += (int32_t)row[xmin + x] * k[x];
                dst[(size_t)y * ow + xx] = ko_clip8(acc);
            }
        }
        src = dst;
        src_w = ow;
    }
    if (need_v) {
        for (int yy = 0; yy < oh; yy++) {
            int ymin = bv[2 * yy], cnt = bv[2 * yy + 1];
            const int32_t *k = kv + (size_t)yy * ksv;
            for (int xx = 0; xx < src_w; xx++) {
                int32_t acc = 1 << (KO_PRECISION_BITS - 1);
                for (int y = 0; y < cnt; y++) acc += (int32_t)src[(size_t)(ymin + y) * src_w + xx] * k[y];
                out[(size_t)yy * src_w + xx] = ko_clip8(acc);
            }
        }
    } else if (!need_h) {
        memcpy(out, L, (size_t)w * (size_t)h);
    }
    free(tmp); free(bh); free(kh); free(bv); free(kv);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* a2: phash -- src/sig/phash.py:33-46.                                        */
/* DCT: folded orthonormal DCT-II in fp64, fixed fma order (DESIGN.md "K2").   */
/* Only dct[:8,:8] is produced (src/sig/phash.py:39).                          */
/* ------------------------------------------------------------------------- */
/* target_clones: the fma clone uses the hardware instruction, the default clone calls
 * libm's exactly-rounded fma(); both give identical bits. */
__attribute__((target_clones("fma", "default")))
static void ko_dct32_first8(const double x[32], double X[8]) {
    double d[16], u[16], e[8], v[8], f[4], w[4];
    for (int n = 0; n < 16; n++) { u[n] = x[n] + x[31 - n]; d[n] = x[n] - x[31 - n]; }
    for (int n = 0; n < 8; n++) { v[n] = u[n] + u[15 - n]; e[n] = u[n] - u[15 - n]; }
    for (int n = 0; n < 4; n++) { w[n] = v[n] + v[7 - n]; f[n] = v[n] - v[7 - n]; }
    X[0] = (w[0] + w[1]) + (w[2] + w[3]);
    for (int q = 0; q < 4; q++) {
        double acc = 0.0;
        for (int n = 0; n < 16; n++) acc = fma(d[n], KE_C32[q][n], acc);
        X[2 * q + 1] = acc;
    }
    for (int q = 0; q < 2; q++) {
        double acc = 0.0;
        for (int n = 0; n < 8; n++) acc = fma(e[n], KE_C16[q][n], acc);
        X[4 * q + 2] = acc;
    }
    {
        double acc = 0.0;
        for (int n = 0; n < 4; n++) acc = fma(f[n], KE_C8[n], acc);
        X[4] = acc;
    }
}

/* tile: 32x32 u8 row-major [y][x].  coef: 64 doubles, [ky][kx], ky = vertical frequency. */
KO_API void ko_dct8x8(const uint8_t *tile, double *coef) {
    double T[32][8]; /* [y][kx] */
    for (int y = 0; y < 32; y++) {
        double x[32];
        for (int n = 0; n < 32; n++) x[n] = (double)tile[y * 32 + n];
        ko_dct32_first8(x, T[y]);
    }
    for (int kx = 0; kx < 8; kx++) {
        double col[32], X[8];
        for (int y = 0; y < 32; y++) col[y] = T[y][kx];
        ko_dct32_first8(col, X);
        for (int ky = 0; ky < 8; ky++) {
            double s = (ky == 0 && kx == 0) ? KE_SCALE00 : ((ky == 0 || kx == 0) ? KE_SCALE0K : KE_SCALEKK);
            coef[ky * 8 + kx] = X[ky] * s;
        }
    }
}

/* numpy float32 add.reduce over 63 contiguous items (pairwise_sum, n<=128 branch):
 * eight running lanes over the first 56, a balanced combine, then the tail. */
static float ko_np_sum63_f32(const float *a) {
    volatile float r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    for (int i = 8; i < 56; i += 8)
        for (int j = 0; j < 8; j++) r[j] = r[j] + a[i + j];
    volatile float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (int i = 56; i < 63; i++) res = res + a[i];
    return res;
}

/* Returns the unsigned 64-bit pHash; *margin_out (nullable) = min |coef - mean|. */
KO_API uint64_t ko_phash_from_tile(const uint8_t *tile32, float *margin_out) {
    double c64[64];
    float c[64];
    ko_dct8x8(tile32, c64);
    for (int i = 0; i < 64; i++) c[i] = (float)c64[i]; /* cv2.dct returns float32 */
    float mean = ko_np_sum63_f32(c + 1) / 63.0f;       /* flat[1:].mean(), src/sig/phash.py:41 */
    uint64_t bits = 0;
    float margin = INFINITY;
    for (int i = 0; i < 64; i++) {
        bits = (bits << 1) | (uint64_t)(c[i] > mean);  /* :42-45, MSB first */
        float m = fabsf(c[i] - mean);
        if (m < margin) margin = m;
    }
    if (margin_out) *margin_out = margin;
    return bits;
}

/* a3: dhash -- src/sig/phash.py:49-57; tile is 8 rows x 9 columns. */
KO_API uint64_t ko_dhash_from_tile(const uint8_t *tile9x8) {
    uint64_t bits = 0;
    for (int r = 0; r < 8; r++)
        for (int c = 0; c < 8; c++)
            bits = (bits << 1) | (uint64_t)(tile9x8[r * 9 + c + 1] > tile9x8[r * 9 + c]);
    return bits;
}

/* Whole a1..a3 chain for one interleaved image (channels = 1, 3 or 4; for 4 the
 * fourth byte is ignored exactly as Pillow's rgb2l ignores it).
 * tile32_out / tile98_out are optional debug outputs (1024 / 72 bytes). */
KO_API int ko_hash_image(const uint8_t *px, int w, int h, int channels, uint64_t *phash_out,
                         uint64_t *dhash_out, uint8_t *tile32_out, uint8_t *tile98_out, float *margin_out) {
    if (w <= 0 || h <= 0 || (channels != 1 && channels != 3 && channels != 4)) return -2;
    uint8_t *L = (uint8_t *)malloc((size_t)w * (size_t)h);
    if (!L) return -1;
    ko_luma(px, (int64_t)w * h, channels, L);
    uint8_t t32[1024], t98[72];
    int rc = 0;
    if (phash_out || tile32_out) {
        rc = ko_resample_lanczos(L, w, h, 32, 32, t32);
        if (rc == 0 && phash_out) *phash_out = ko_phash_from_tile(t32, margin_out);
        if (rc == 0 && tile32_out) memcpy(tile32_out, t32, 1024);
    }
    if (rc == 0 && (dhash_out || tile98_out)) {
        rc = ko_resample_lanczos(L, w, h, 9, 8, t98);
        if (rc == 0 && dhash_out) *dhash_out = ko_dhash_from_tile(t98);
        if (rc == 0 && tile98_out) memcpy(tile98_out, t98, 72);
    }
    free(L);
    return rc;
}

/* Batch of n equally sized images, contiguous. */
KO_API int ko_hash_batch(const uint8_t *px, int64_t n, int w, int h, int channels, uint64_t *phash_out,
                         uint64_t *dhash_out) {
    size_t stride = (size_t)w * h * channels;
    for (int64_t i = 0; i < n; i++) {
        int rc = ko_hash_image(px + i * stride, w, h, channels, phash_out ? phash_out + i : NULL,
                               dhash_out ? dhash_out + i : NULL, NULL, NULL, NULL);
        if (rc) return rc;
    }
    return 0;
}

/* Shipped refine stage, src/ui/dup_refine_parallel.py:59-83: side = grid*tile luma thumbnail -> one bit
 * per pixel (pixel > float64 mean of its tile), order (gy, gx, ty, tx), np.packbits(bitorder="little").
 * words_out: ceil(side*side/64) little-endian u64 words. */
KO_API void ko_tile_ahash(const uint8_t *thumb, int grid, int tile, uint64_t *words_out) {
    int side = grid * tile, nbits = side * side, i = 0;
    for (int k = 0; k < (nbits + 63) / 64; k++) words_out[k] = 0;
    for (int gy = 0; gy < grid; gy++)
        for (int gx = 0; gx < grid; gx++) {
            double sum = 0.0;
            for (int ty = 0; ty < tile; ty++)
                for (int tx = 0; tx < tile; tx++) sum += thumb[(gy * tile + ty) * side + gx * tile + tx];
            double mean = sum / (double)(tile * tile);
            for (int ty = 0; ty < tile; ty++)
                for (int tx = 0; tx < tile; tx++, i++)
                    if ((double)thumb[(gy * tile + ty) * side + gx * tile + tx] > mean) words_out[i >> 6] |= 1ull << (i & 63);
        }
}

/* ------------------------------------------------------------------------- */
/* Synthetic corpus (DESIGN.md "Synthetic data"); integer-only, counter based.  */
/* ------------------------------------------------------------------------- */
static inline uint64_t ko_splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
#define KO_P1 0xD6E8FEB86659FD93ull
#define KO_P2 0xCA5A826395121157ull
#define KO_P3 0x9E6C63D0676A9A99ull

/* Identity of corpus image i.  Ordinary images are their own base.  Every index = 9 (mod 10) from 19 on is a
 * VARIANT of an earlier base: brightness shift delta in [-3, 3] and a fraction cell_q/32 of the 16x16 cells
 * replaced.  Two variant classes:
 *   ordinary  : fresh per-pixel noise (noise_index = i), cell_q = 1  -> SSIM vs the base 0.886-0.945;
 *   low-noise : from index 1000 on, variants with bit 52 of r set REUSE THE BASE'S per-pixel noise
 *               (noise_index = base) and replace cell_q = 1..4 of 32 cells -> SSIM vs the base spreads over
 *               roughly 0.93-0.99, so the reference's ssim_threshold = 0.95 cuts inside the class (BASELINE
 *               configs[3]).  The first 1000 images (BASELINE configs[0]) are as in round 1. */
#define KO_SYNTH_LOWNOISE_FROM 1000
static inline void ko_synth_identity2(uint64_t seed, int64_t i, int64_t *base, int *delta, int *is_variant,
                                      int64_t *noise_index, int *cell_q) {
    uint64_t r = ko_splitmix64(seed ^ (0x51ED270B0E3A6F5Dull + (uint64_t)i * KO_P1));
    *is_variant = (i >= 10) && (i % 10 == 9);
    *noise_index = i;
    *cell_q = 1;
    if (*is_variant) {
        int64_t c = (int64_t)((r >> 8) % (uint64_t)i);
        if (c % 10 == 9) c -= 1;
        *base = c;
        *delta = (int)((r >> 44) % 7) - 3;
        if (i >= KO_SYNTH_LOWNOISE_FROM && ((r >> 52) & 1)) {
            *noise_index = c;
            *cell_q = 1 + (int)((r >> 53) & 3);
        }
    } else {
        *base = i;
        *delta = 0;
    }
}

/* (base index, brightness delta, is_variant) of corpus image `index`. */
KO_API void ko_synth_info(uint64_t seed, int64_t index, int64_t *base, int32_t *delta, int32_t *is_variant) {
    int d, v, q; int64_t ni;
    ko_synth_identity2(seed, index, base, &d, &v, &ni, &q);
    *delta = d; *is_variant = v;
}

/* the same plus the variant class: low_noise = 1 when the variant reuses its base's noise, cell_q = replaced cells per 32 */
KO_API void ko_synth_info2(uint64_t seed, int64_t index, int64_t *base, int32_t *delta, int32_t *is_variant,
                           int32_t *low_noise, int32_t *cell_q) {
    int d, v, q; int64_t ni;
    ko_synth_identity2(seed, index, base, &d, &v, &ni, &q);
    *delta = d; *is_variant = v; *low_noise = (ni != index); *cell_q = q;
}

KO_API void ko_synth_rgb(uint64_t seed, int64_t index, int w, int h, uint8_t *out) {
    int64_t base, noise_index; int delta, variant, cell_q;
    ko_synth_identity2(seed, index, &base, &delta, &variant, &noise_index, &cell_q);
    for (int y = 0; y < h; y++) {
        int cy = (int)(((int64_t)y * 16) / h);
        for (int x = 0; x < w; x++) {
            int cx = (int)(((int64_t)x * 16) / w);
            uint64_t cell = ko_splitmix64((seed + 1) ^ ((uint64_t)base * KO_P1 + (uint64_t)cy * KO_P2 + (uint64_t)cx * KO_P3));
            if (variant) {
                uint64_t m = ko_splitmix64((seed + 2) ^ ((uint64_t)index * KO_P1 + (uint64_t)cy * KO_P2 + (uint64_t)cx * KO_P3));
                if ((int)(m & 31) < cell_q) cell = m >> 8;
            }
            uint64_t u = ko_splitmix64(seed ^ ((uint64_t)noise_index * KO_P1 + (uint64_t)y * KO_P2 + (uint64_t)x * KO_P3));
            uint8_t *p = out + ((size_t)y * w + x) * 3;
            for (int c = 0; c < 3; c++) {
                int v = (int)((cell >> (8 * c)) & 0xFF) + (int)((u >> (20 * c)) & 7) - 4 + delta;
                p[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
    }
}

/* Synthetic 64-bit hash table for scan-only runs: the first n - n/10 entries are
 * uniform, the last n/10 are copies of a uniform earlier entry with 0..10 distinct
 * bits flipped. */
KO_API void ko_synth_hashes(uint64_t seed, int64_t n, uint64_t *out) {
    int64_t n_plain = n - n / 10;
    for (int64_t i = 0; i < n_plain; i++) out[i] = ko_splitmix64(seed + (uint64_t)i);
    for (int64_t i = n_plain; i < n; i++) {
        uint64_t r = ko_splitmix64(seed ^ (0x7F4A7C159E3779B9ull + (uint64_t)i * KO_P2));
        uint64_t v = ko_splitmix64(seed + (r % (uint64_t)n_plain));
        int k = (int)((r >> 40) % 11);
        uint64_t flipped = 0, s = r;
        while (k > 0) {
            s = ko_splitmix64(s);
            uint64_t bit = 1ull << (s & 63);
            if (flipped & bit) continue;
            flipped |= bit;
            k--;
        }
        out[i] = v ^ flipped;
    }
}

/* ------------------------------------------------------------------------- */
/* a4 + a9: candidate generation -- src/dup/scanner.py:227-299 restated with the */
/* reference's own shape: band buckets in insertion order, in-bucket pair loop.  */
/* ------------------------------------------------------------------------- */
typedef struct {
    int64_t a, b;   /* positions in the input list, a < b */
    int32_t h;      /* hamming64, src/sig/phash.py:60-63 */
    int32_t bands;  /* bit b set <=> the pair met in an (uncapped) bucket of band b */
} ko_edge;

typedef struct { uint64_t key; int64_t idx; } ko_kv;

static int ko_kv_cmp(const void *pa, const void *pb) {
    const ko_kv *a = (const ko_kv *)pa, *b = (const ko_kv *)pb;
    if (a->key != b->key) return a->key < b->key ? -1 : 1;
    return a->idx < b->idx ? -1 : (a->idx > b->idx);
}

typedef struct { int64_t rank; int64_t begin, end; } ko_bucket;

static int ko_bucket_cmp(const void *pa, const void *pb) {
    const ko_bucket *a = (const ko_bucket *)pa, *b = (const ko_bucket *)pb;
    return a->rank < b->rank ? -1 : (a->rank > b->rank);
}

typedef struct { int64_t ka, kb; int64_t seq; ko_edge e; } ko_hit;

static int ko_hit_cmp(const void *pa, const void *pb) {
    const ko_hit *a = (const ko_hit *)pa, *b = (const ko_hit *)pb;
    if (a->ka != b->ka) return a->ka < b->ka ? -1 : 1;
    if (a->kb != b->kb) return a->kb < b->kb ? -1 : 1;
    return a->seq < b->seq ? -1 : (a->seq > b->seq);
}

static int ko_size_ok(int64_t sa, int64_t sb, double ratio) {
    /* _passes_size_ratio, src/dup/scanner.py:358-370 (true division, fp64 compare) */
    if (!(ratio > 0)) return 1;
    if (sa <= 0 || sb <= 0) return 1;
    int64_t small = sa < sb ? sa : sb, large = sa < sb ? sb : sa;
    return ((double)small / (double)large) >= ratio;
}

/* Returns the number of distinct edges (keyed by sorted file-id pair, first writer
 * wins -- src/dup/scanner.py:287-290), or <0 on error.  Up to `capacity` edges are
 * written sorted by (id_a,id_b) with a,b = list positions of the first writer.
 * counters[0..2] = pair_total, after_size, after_ham exactly as logged at :292-299
 * (pairs that share several bands are counted once per shared bucket). */
KO_API int64_t ko_scan_banded(const uint64_t *hashes, const int64_t *ids, const int64_t *sizes, int64_t n,
                              int threshold, int band_bits, int band_count, double size_ratio,
                              int64_t bucket_pair_cap, ko_edge *edges_out, int64_t capacity,
                              uint64_t *counters) {
    if (band_bits <= 0 || band_count <= 0 || band_bits * band_count > 64) return -2;
    uint64_t mask = band_bits >= 64 ? ~0ull : ((1ull << band_bits) - 1);
    uint64_t pair_total = 0, after_size = 0, after_ham = 0;
    ko_kv *kv = (ko_kv *)malloc(sizeof(ko_kv) * (size_t)(n > 0 ? n : 1));
    ko_bucket *buckets = (ko_bucket *)malloc(sizeof(ko_bucket) * (size_t)(n > 0 ? n : 1) * (size_t)band_count);
    int64_t *members = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1) * (size_t)band_count);
    size_t hit_cap = 1024, n_hits = 0;
    ko_hit *hits = (ko_hit *)malloc(sizeof(ko_hit) * hit_cap);
    if (!kv || !buckets || !members || !hits) { free(kv); free(buckets); free(members); free(hits); return -1; }
    int64_t n_buckets = 0, n_members = 0;
    for (int band = 0; band < band_count; band++) {
        for (int64_t i = 0; i < n; i++) {
            kv[i].key = (hashes[i] >> (band * band_bits)) & mask;
            kv[i].idx = i;
        }
        qsort(kv, (size_t)n, sizeof(ko_kv), ko_kv_cmp);
        for (int64_t s = 0; s < n;) {
            int64_t e = s + 1;
            while (e < n && kv[e].key == kv[s].key) e++;
            /* dict insertion order: buckets appear while scanning idx ascending, band ascending */
            buckets[n_buckets].rank = kv[s].idx * band_count + band;
            buckets[n_buckets].begin = n_members;
            for (int64_t t = s; t < e; t++) members[n_members++] = kv[t].idx;
            buckets[n_buckets].end = n_members;
            n_buckets++;
            s = e;
        }
    }
    qsort(buckets, (size_t)n_buckets, sizeof(ko_bucket), ko_bucket_cmp);
    int64_t seq = 0;
    for (int64_t bi = 0; bi < n_buckets; bi++) {
        int64_t len = buckets[bi].end - buckets[bi].begin;
        if (len < 2) continue;
        if (bucket_pair_cap > 0 && (len * (len - 1)) / 2 > bucket_pair_cap) continue;
        int band = (int)(buckets[bi].rank % band_count);
        const int64_t *m = members + buckets[bi].begin;
        for (int64_t i = 0; i + 1 < len; i++) {
            for (int64_t j = i + 1; j < len; j++) {
                int64_t a = m[i], b = m[j];
                int64_t ida = ids ? ids[a] : a, idb = ids ? ids[b] : b;
                if (ida == idb) continue;
                pair_total++;
                if (sizes && !ko_size_ok(sizes[a], sizes[b], size_ratio)) continue;
                after_size++;
                int hd = __builtin_popcountll(hashes[a] ^ hashes[b]);
                if (hd > threshold) continue;
                after_ham++;
                if (n_hits == hit_cap) {
                    hit_cap *= 2;
                    ko_hit *nh = (ko_hit *)realloc(hits, sizeof(ko_hit) * hit_cap);
                    if (!nh) { free(kv); free(buckets); free(members); free(hits); return -1; }
                    hits = nh;
                }
                ko_hit *hh = &hits[n_hits++];
                hh->ka = ida < idb ? ida : idb;
                hh->kb = ida < idb ? idb : ida;
                hh->seq = seq++;
                hh->e.a = a; hh->e.b = b; hh->e.h = hd; hh->e.bands = 1 << band;
            }
        }
    }
    qsort(hits, n_hits, sizeof(ko_hit), ko_hit_cmp);
    int64_t n_edges = 0;
    for (size_t s = 0; s < n_hits;) {
        size_t e = s + 1;
        int32_t bands = hits[s].e.bands;
        while (e < n_hits && hits[e].ka == hits[s].ka && hits[e].kb == hits[s].kb) { bands |= hits[e].e.bands; e++; }
        if (n_edges < capacity && edges_out) {
            edges_out[n_edges] = hits[s].e;
            edges_out[n_edges].bands = bands;
        }
        n_edges++;
        s = e;
    }
    if (counters) { counters[0] = pair_total; counters[1] = after_size; counters[2] = after_ham; }
    free(kv); free(buckets); free(members); free(hits);
    return n_edges;
}

/* Independent closed form: brute force over i<j with the band-lane test.
 * Used to cross-check ko_scan_banded; O(n^2), small n only.  Unique ids, no cap. */
KO_API int64_t ko_scan_bruteforce(const uint64_t *hashes, int64_t n, int threshold, int band_bits, int band_count,
                                  ko_edge *edges_out, int64_t capacity) {
    uint64_t mask = band_bits >= 64 ? ~0ull : ((1ull << band_bits) - 1);
    int64_t n_edges = 0;
    for (int64_t i = 0; i < n; i++)
        for (int64_t j = i + 1; j < n; j++) {
            uint64_t d = hashes[i] ^ hashes[j];
            int hd = __builtin_popcountll(d);
            if (hd > threshold) continue;
            int32_t bands = 0;
            for (int b = 0; b < band_count; b++)
                if (((d >> (b * band_bits)) & mask) == 0) bands |= 1 << b;
            if (!bands) continue;
            if (n_edges < capacity && edges_out) {
                edges_out[n_edges].a = i; edges_out[n_edges].b = j;
                edges_out[n_edges].h = hd; edges_out[n_edges].bands = bands;
            }
            n_edges++;
        }
    return n_edges;
}

/* ------------------------------------------------------------------------- */
/* a11: SSIM -- src/dup/refine.py:44-52 -> skimage.metrics.structural_similarity */
/* (float32 images, data_range=1.0, win 7, uniform filter, sample covariance).   */
/* scipy.ndimage.uniform_filter semantics: axis 0 then axis 1, each pass sums in */
/* fp64 and stores float32.  Only the interior (3 px crop) enters the mean, so    */
/* the boundary mode never matters.                                              */
/* ------------------------------------------------------------------------- */
static void ko_box7_f32(const float *in, int w, int h, float *tmp, float *out) {
    /* vertical pass, rows 3..h-4 */
    for (int y = 3; y < h - 3; y++)
        for (int x = 0; x < w; x++) {
            double s = 0.0;
            for (int d = -3; d <= 3; d++) s += (double)in[(size_t)(y + d) * w + x];
            tmp[(size_t)y * w + x] = (float)(s / 7.0);
        }
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            double s = 0.0;
            for (int d = -3; d <= 3; d++) s += (double)tmp[(size_t)y * w + x + d];
            out[(size_t)y * w + x] = (float)(s / 7.0);
        }
}

/* a, b: luma u8 images of identical size (w x h, both >= 7).  Returns SSIM. */
KO_API double ko_ssim_luma(const uint8_t *a, const uint8_t *b, int w, int h) {
    if (w < 7 || h < 7) return NAN;
    size_t n = (size_t)w * h;
    float *buf = (float *)malloc(sizeof(float) * n * 9);
    if (!buf) return NAN;
    float *x = buf, *y = buf + n, *p = buf + 2 * n, *tmp = buf + 3 * n;
    float *ux = buf + 4 * n, *uy = buf + 5 * n, *uxx = buf + 6 * n, *uyy = buf + 7 * n, *uxy = buf + 8 * n;
    for (size_t i = 0; i < n; i++) { x[i] = (float)a[i] / 255.0f; y[i] = (float)b[i] / 255.0f; }
    ko_box7_f32(x, w, h, tmp, ux);
    ko_box7_f32(y, w, h, tmp, uy);
    for (size_t i = 0; i < n; i++) p[i] = x[i] * x[i];
    ko_box7_f32(p, w, h, tmp, uxx);
    for (size_t i = 0; i < n; i++) p[i] = y[i] * y[i];
    ko_box7_f32(p, w, h, tmp, uyy);
    for (size_t i = 0; i < n; i++) p[i] = x[i] * y[i];
    ko_box7_f32(p, w, h, tmp, uxy);
    const float cov_norm = (float)(49.0 / 48.0);
    const float C1 = (float)(0.01 * 0.01), C2 = (float)(0.03 * 0.03);
    double total = 0.0;
    for (int yy = 3; yy < h - 3; yy++)
        for (int xx = 3; xx < w - 3; xx++) {
            size_t i = (size_t)yy * w + xx;
            volatile float mxx = ux[i] * ux[i], myy = uy[i] * uy[i], mxy = ux[i] * uy[i];
            volatile float vx = cov_norm * (uxx[i] - mxx);
            volatile float vy = cov_norm * (uyy[i] - myy);
            volatile float vxy = cov_norm * (uxy[i] - mxy);
            volatile float A1 = 2.0f * mxy + C1, A2 = 2.0f * vxy + C2;
            volatile float B1 = (mxx + myy) + C1, B2 = (vx + vy) + C2;
            volatile float num = A1 * A2, den = B1 * B2;
            float S = num / den;
            total += (double)S;
        }
    free(buf);
    return total / ((double)(w - 6) * (double)(h - 6));
}
