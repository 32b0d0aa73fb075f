// ke_hash.hip -- pHash / dHash kernels for gfx950.
//
// Replaces, per image: sig.phash._to_grayscale (Pillow convert("L") + LANCZOS resize,
// src/sig/phash.py:21-26), phash (cv2.dct + mean threshold, :33-46) and dhash (:49-57).
//
//   K1  luma + horizontal LANCZOS  -- integer, bit-exact with Pillow
//   K1' vertical LANCZOS            -- integer, bit-exact with Pillow
//   K2  8x8 corner of the 32x32 DCT in fp64 (folded, fixed fma order), float32 mean in
//       NumPy's pairwise order, 64 compare bits MSB first
//   K3  dHash from the 9x8 tile
//
// Paths, same arithmetic in all of them:
//   * single-pass kernels, one workgroup per image, the whole chain in one launch: the image is
//     streamed once from HBM (12 or 16 B/lane coalesced loads), luma goes to LDS as signed bytes,
//     the 22-bit tap weights are split into three signed byte planes; the horizontal taps run on
//     v_mfma_i32_16x16x64_i8 (exact int32), the (32x smaller) vertical taps on v_dot4_i32_i8.
//     HBM-bound by design: C*W*H bytes in, 8 (16 with dHash) out.
//       ke_phash_fused_mx    RGB / RGBX rows up to 768 pixels, 32-row tiles, 256 threads
//       ke_phash_fused_wide  RGB rows of 708..2048 pixels, 16-row tiles, 512 threads
//   * banded path (ke_hband + ke_vtile + ke_tiles_to_hashes): every other shape, odd widths,
//     1-byte pixels, and small groups of large images (an image is cut into bands of rows so that a
//     few images still fill the GPU); also the resize behind ke_resize_luma_uniform / ke_fit_luma_uniform
//   * generic passes (ke_resample_pass): last resort (Pillow's vertical-first rule, tiny images).
#include <cstdlib>

#include "ke_internal.h"
#include "dct_table.h"

namespace {

constexpr int kPrecisionBits = 22;

__device__ __forceinline__ int clip8_fixed(int acc) {
    int v = acc >> kPrecisionBits;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__device__ __forceinline__ int luma_of(const uint8_t *p, int ch) {
    if (ch == 1) return p[0];
    return (int)((19595u * p[0] + 38470u * p[1] + 7471u * p[2] + 0x8000u) >> 16);
}

// ---------------------------------------------------------------------------------------
// Generic single-axis resample pass.  AXIS 0: dst[y][o] over x taps (dst is h x out_size);
// AXIS 1: dst[o][x] over y taps (dst is out_size x w).  Source may be interleaved (ch 3/4:
// luma taken on the fly) or single channel.
// ---------------------------------------------------------------------------------------
template <int AXIS>
__global__ __launch_bounds__(256) void ke_resample_pass(const uint8_t *__restrict__ src,
                                                         const uint64_t *__restrict__ offsets, uint64_t stride,
                                                         int ch, int w, int h, const int32_t *__restrict__ bounds,
                                                         const int32_t *__restrict__ kk, int ksize, int out_size,
                                                         uint8_t *__restrict__ dst, int blocks_per_image) {
    const int64_t img = blockIdx.x / blocks_per_image;
    const int blk = blockIdx.x % blocks_per_image;
    const int ow = AXIS == 0 ? out_size : w;
    const int oh = AXIS == 0 ? h : out_size;
    const int e = blk * 256 + threadIdx.x;
    if (e >= ow * oh) return;
    const int oy = e / ow, ox = e % ow;
    const uint8_t *base = src + (offsets ? offsets[img] : (uint64_t)img * stride);
    const int o = AXIS == 0 ? ox : oy;
    const int first = bounds[2 * o], cnt = bounds[2 * o + 1];
    const int32_t *k = kk + (size_t)o * ksize;
    int acc = 1 << (kPrecisionBits - 1);
    if (AXIS == 0) {
        const uint8_t *row = base + ((size_t)oy * w + first) * ch;
        for (int t = 0; t < cnt; ++t) acc += luma_of(row + (size_t)t * ch, ch) * k[t];
    } else {
        const uint8_t *col = base + ((size_t)first * w + ox) * ch;
        for (int t = 0; t < cnt; ++t) acc += luma_of(col + (size_t)t * w * ch, ch) * k[t];
    }
    dst[(size_t)img * ow * oh + e] = (uint8_t)clip8_fixed(acc);
}

// ---------------------------------------------------------------------------------------
// K2: folded DCT.  One output of the length-32 orthonormal DCT-II (unnormalised), k in 0..7.
//   even k recurse on the folded sums, odd k are a short cosine dot product on the folded
//   differences; every sum is a single add, every dot product an fma chain in ascending n,
//   so flat and mirror-symmetric inputs give exact zeros.
// ---------------------------------------------------------------------------------------
__constant__ double c_C32[4][16];
__constant__ double c_C16[2][8];
__constant__ double c_C8[4];

__device__ double dct32_one(const double *x, int k) {
    if (k & 1) {
        const int q = k >> 1;
        double acc = 0.0;
#pragma unroll
        for (int n = 0; n < 16; ++n) acc = fma(x[n] - x[31 - n], c_C32[q][n], acc);
        return acc;
    }
    double u[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) u[n] = x[n] + x[31 - n];
    if (k & 2) {
        const int q = k >> 2;  // k = 2 -> 0, k = 6 -> 1
        double acc = 0.0;
#pragma unroll
        for (int n = 0; n < 8; ++n) acc = fma(u[n] - u[15 - n], c_C16[q][n], acc);
        return acc;
    }
    double v[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) v[n] = u[n] + u[15 - n];
    if (k == 4) {
        double acc = 0.0;
#pragma unroll
        for (int n = 0; n < 4; ++n) acc = fma(v[n] - v[7 - n], c_C8[n], acc);
        return acc;
    }
    const double w0 = v[0] + v[7], w1 = v[1] + v[6], w2 = v[2] + v[5], w3 = v[3] + v[4];
    return (w0 + w1) + (w2 + w3);
}

// tile: 32x32 u8 in LDS; Td: 256 doubles in LDS; cf: 64 floats in LDS.  All 256 threads of
// the workgroup call this; the hash is returned in thread 0 (undefined elsewhere).
// margin (nullable): min over the 64 coefficients of |coef - mean| in float32 -- how close the image is to a tie of the
// bit decision `coef > mean` (src/sig/phash.py:41-42), the one place where another DCT implementation could differ.
__device__ uint64_t tile32_to_phash(const uint8_t *tile, double *Td, float *cf, int tid, float *margin = nullptr) {
    {
        const int y = tid >> 3, kx = tid & 7;
        double x[32];
#pragma unroll
        for (int n = 0; n < 32; ++n) x[n] = (double)tile[y * 32 + n];
        Td[y * 8 + kx] = dct32_one(x, kx);
    }
    __syncthreads();
    if (tid < 64) {
        const int ky = tid >> 3, kx = tid & 7;
        double x[32];
#pragma unroll
        for (int n = 0; n < 32; ++n) x[n] = Td[n * 8 + kx];
        const double s = (ky == 0 && kx == 0) ? KE_SCALE00 : ((ky == 0 || kx == 0) ? KE_SCALE0K : KE_SCALEKK);
        cf[tid] = (float)(dct32_one(x, ky) * s);  // cv2.dct hands back float32
    }
    __syncthreads();
    uint64_t hash = 0;
    if (tid < 64) {
        // flat[1:].mean() in float32, NumPy pairwise order for 63 items (src/sig/phash.py:41):
        // eight running lanes over the first 56, balanced combine, then the 7-item tail.
        const float *a = cf + 1;
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = a[j];
#pragma unroll
        for (int i = 8; i < 56; i += 8)
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = r[j] + a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
        for (int i = 56; i < 63; ++i) res = res + a[i];
        const float mean = res / 63.0f;
        const unsigned long long m = __ballot(cf[tid] > mean);  // lane i <-> flat[i]
        hash = __brevll(m);                                     // flat[0] is the MSB (:43-45)
        if (margin) {                                           // workgroup-uniform
            float mg = fabsf(cf[tid] - mean);
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) mg = fminf(mg, __shfl_xor(mg, s));
            if (tid == 0) *margin = mg;
        }
    }
    return hash;
}

__global__ __launch_bounds__(256) void ke_tiles_to_hashes(const uint8_t *__restrict__ tile32,
                                                           const uint8_t *__restrict__ tile98,
                                                           const int64_t *__restrict__ out_idx,
                                                           uint64_t *__restrict__ phash, uint64_t *__restrict__ dhash,
                                                           float *__restrict__ margin) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[1024];
    __shared__ double s_T[256];
    __shared__ float s_cf[64];
    const int tid = threadIdx.x;
    const int64_t img = blockIdx.x;
    const int64_t slot = out_idx ? out_idx[img] : img;
    if (phash) {
        reinterpret_cast<uint32_t *>(s_tile)[tid] = reinterpret_cast<const uint32_t *>(tile32 + (size_t)img * 1024)[tid];
        __syncthreads();
        const uint64_t hv = tile32_to_phash(s_tile, s_T, s_cf, tid, margin ? margin + slot : nullptr);
        if (tid == 0) phash[slot] = hv;
    }
    if (dhash && tid < 64) {
        // src/sig/phash.py:49-57: 8 rows x 9 columns, bit = right > left, row-major, MSB first
        const uint8_t *t = tile98 + (size_t)img * 72;
        const int r = tid >> 3, c = tid & 7;
        const unsigned long long m = __ballot(t[r * 9 + c + 1] > t[r * 9 + c]);
        if (tid == 0) dhash[slot] = __brevll(m);
    }
}

// ---------------------------------------------------------------------------------------
// Pieces shared by the fused kernel and the banded path: luma of four packed RGB pixels, the byte-plane
// recombination, the kernel arguments and the part of the chain behind the row loop.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t luma4_biased(uint32_t d0, uint32_t d1, uint32_t d2) {
    // 19595 = 76*256+139, 38470 = 150*256+70, 7471 = 29*256+47: two u8 dot products per pixel.
    // The accumulator start folds in the +0x8000 rounding and the -128 bias of the signed
    // luma byte, which ends up in bits 16..23.
    constexpr uint32_t CHI = 0x001D964Cu, CLO = 0x002F468Bu, START = 0x8000u - 0x800000u;
    const uint32_t p0 = d0;
    const uint32_t p1 = __builtin_amdgcn_alignbyte(d1, d0, 3);
    const uint32_t p2 = __builtin_amdgcn_alignbyte(d2, d1, 2);
    // the fourth pixel sits in bytes 1..3 of d2: shift the weights, not the data
    const uint32_t s0 = (__builtin_amdgcn_udot4(p0, CHI, 0u, false) << 8) + __builtin_amdgcn_udot4(p0, CLO, START, false);
    const uint32_t s1 = (__builtin_amdgcn_udot4(p1, CHI, 0u, false) << 8) + __builtin_amdgcn_udot4(p1, CLO, START, false);
    const uint32_t s2 = (__builtin_amdgcn_udot4(p2, CHI, 0u, false) << 8) + __builtin_amdgcn_udot4(p2, CLO, START, false);
    const uint32_t s3 = (__builtin_amdgcn_udot4(d2, CHI << 8, 0u, false) << 8) + __builtin_amdgcn_udot4(d2, CLO << 8, START, false);
    // byte 2 of each sum -> one dword: two v_perm_b32 (selector bytes pick from {s_odd[3:0] = 4..7, s_even[3:0] = 0..3})
    // and one v_bfi_b32 instead of shift/and/or chains (the VALU issue rate bounds this kernel)
    const uint32_t lo = __builtin_amdgcn_perm(s1, s0, 0x0C0C0602u);   // [s0.b2, s1.b2, 0, 0]
    const uint32_t hi = __builtin_amdgcn_perm(s3, s2, 0x06020C0Cu);   // [0, 0, s2.b2, s3.b2]
    return lo | hi;
}

// Four consecutive pixels of one row -> four signed luma bytes.  d: the 4*C source bytes.
template <int C>
__device__ __forceinline__ uint32_t luma4_generic(const uint32_t *d) {
    if (C == 3) return luma4_biased(d[0], d[1], d[2]);
    if (C == 1) return d[0] ^ 0x80808080u;
    constexpr uint32_t CHI = 0x001D964Cu, CLO = 0x002F468Bu, START = 0x8000u - 0x800000u;
    uint32_t s[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)   // RGBX: byte 3 meets a zero weight, as Pillow's rgb2l ignores it
        s[k] = (__builtin_amdgcn_udot4(d[k], CHI, 0u, false) << 8) + __builtin_amdgcn_udot4(d[k], CLO, START, false);
    return __builtin_amdgcn_perm(s[1], s[0], 0x0C0C0602u) | __builtin_amdgcn_perm(s[3], s[2], 0x06020C0Cu);   // byte 2 of each sum
}

__device__ __forceinline__ int combine_planes(int d0, int d1, int d2, int bias) {
    return (int)((uint32_t)d0 + ((uint32_t)d1 << 8) + ((uint32_t)d2 << 16) + (uint32_t)bias);
}

struct KeFusedArgs {
    const uint8_t *pixels;
    const uint64_t *offsets;
    uint64_t stride;
    const int64_t *out_idx;
    int h;
    // horizontal axis (32 outputs): matrix-core operands (KeMxTable::frag), first tap column of the two
    // 16-output tiles, rounding/bias term per output
    const int32_t *mx_frag;
    int mx_base0, mx_base1;
    const int32_t *h_bias;
    const int32_t *v_packed, *v_start, *v_bias;          // vertical axis: 32 outputs, ndwv window dwords
    int ndwv;
    int qw, qw_inv, lp;   // GEN instantiations: quads per row (ceil(W / 4)), ceil(2^32 / qw), LDS pitch of a luma row
    int w, row_bytes;     // UNAL instantiations (W not a multiple of 4): row length in pixels and bytes
    int lt_half;    // LDS bytes of ONE luma tile buffer (two are allocated; the tail reuses them as scratch)
    int lt_bytes;   // = 2 * lt_half
    int hp;         // pitch of one HT column (bytes, multiple of 8)
    uint64_t *phash;
    float *margin;        // nullable: tie margin per image (same slot as phash)
    uint8_t *tile32_out;  // nullable debug output
    // dHash side (DH instantiations): operands of the 9-output horizontal axis, 8 output rows x 3 chunks vertically
    const int32_t *mxd_frag, *hd_bias;
    const int32_t *vd_cpacked, *vd_cstart, *vd_bias;
    int ndwcv;      // chunk dwords of the vertical dHash axis
    int hpd;        // pitch of one dHash HT column (bytes, multiple of 8)
    int x_off;      // LDS offset of the exchange buffer of the two half-row waves
    // band mode (bands > 0): a workgroup turns rows [band * band_rows, +band_rows) of its image into transposed columns and
    // writes them to the global scratch hs[img][32][hs_hp] (hsd[img][9][hsd_hp] for the dHash axis); ke_vtile finishes
    int bands, band_rows;
    uint8_t *hs, *hsd;
    int hs_hp, hsd_hp;
    uint64_t *dhash;
    uint8_t *tile98_out;
};

// UNAL instantiations: a row whose length is not a multiple of 4 ends in a partial quad whose load also takes the first
// pixels of the next row (harmless: those columns only meet zero taps).  The partial quad of the image's LAST row would
// read past the image, so its load is clamped and its 1..3 pixels are patched into the LDS tile here, byte by byte.
template <int C>
__device__ __forceinline__ void patch_last_quad(const uint8_t *image_last_row, int w, uint8_t *lds_row) {
    for (int x = w & ~3; x < w; ++x) lds_row[x] = (uint8_t)(luma_of(image_last_row + (size_t)x * C, C) ^ 0x80);
}

// Band mode of the single-pass kernels: the band's transposed columns (LDS, pitch hp) -> global scratch, whole dwords
// along y (band starts and pitches are multiples of 4; the scratch rows are padded).
__device__ __forceinline__ void band_writeout(const uint8_t *HT, int hp, int ncols, int rows, uint8_t *dst, int dst_hp, int tid,
                                              int nthreads) {
    const int wpc = (rows + 3) / 4;
    for (int e = tid; e < ncols * wpc; e += nthreads) {
        const int col = e / wpc, k = e % wpc;
        *reinterpret_cast<uint32_t *>(dst + (size_t)col * dst_hp + 4 * k) = *reinterpret_cast<const uint32_t *>(HT + col * hp + 4 * k);
    }
}

// K1' (vertical taps out of the transposed horizontal results HT), K2 (DCT corner, mean, bits) and, for the
// NDWD > 0 instantiations, the dHash vertical taps + K3: the part of the fused kernels behind the row loop.
// Lt is free by then and is reused as scratch (>= 3328 bytes).
template <int NDWD>
__device__ __forceinline__ void fused_tail(const KeFusedArgs &a, uint8_t *Lt, const uint8_t *HT, const uint8_t *HTd,
                                           const int tid, const int64_t img) {
    // ---- K1': vertical taps: thread = (output row yy, columns og + 8m)
    uint8_t *T32 = Lt;
    double *Td = reinterpret_cast<double *>(Lt + 1024);
    float *cf = reinterpret_cast<float *>(Lt + 1024 + 2048);
    {
        const int yy = tid & 31, og = tid >> 5;
        const int rotv = (yy >> 4) & 1;
        const int vst = a.v_start[yy];
        const int ndwv = a.ndwv;
        int acc[4][3];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m][0] = acc[m][1] = acc[m][2] = 0;
        for (int j = 0; j < ndwv; j += 2) {
            const int jj = j ^ (2 * rotv);
            const int32_t *cp = a.v_packed + ((size_t)yy * ndwv + jj) * 3;
            const int c00 = cp[0], c01 = cp[1], c02 = cp[2], c10 = cp[3], c11 = cp[4], c12 = cp[5];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const uint2 v = *reinterpret_cast<const uint2 *>(
                    __builtin_assume_aligned(HT + (size_t)(og + 8 * m) * a.hp + vst + 4 * jj, 8));
                acc[m][0] = __builtin_amdgcn_sdot4((int)v.x, c00, acc[m][0], false);
                acc[m][1] = __builtin_amdgcn_sdot4((int)v.x, c01, acc[m][1], false);
                acc[m][2] = __builtin_amdgcn_sdot4((int)v.x, c02, acc[m][2], false);
                acc[m][0] = __builtin_amdgcn_sdot4((int)v.y, c10, acc[m][0], false);
                acc[m][1] = __builtin_amdgcn_sdot4((int)v.y, c11, acc[m][1], false);
                acc[m][2] = __builtin_amdgcn_sdot4((int)v.y, c12, acc[m][2], false);
            }
        }
        const int vbias = a.v_bias[yy];
#pragma unroll
        for (int m = 0; m < 4; ++m)
            T32[yy * 32 + og + 8 * m] = (uint8_t)clip8_fixed(combine_planes(acc[m][0], acc[m][1], acc[m][2], vbias));
    }
    __syncthreads();
    if (a.tile32_out)
        reinterpret_cast<uint32_t *>(a.tile32_out + (size_t)img * 1024)[tid] = reinterpret_cast<const uint32_t *>(T32)[tid];
    // ---- K2: DCT corner, mean, bits
    const int64_t slot = a.out_idx ? a.out_idx[img] : img;
    const uint64_t hv = tile32_to_phash(T32, Td, cf, tid, (a.margin && a.phash) ? a.margin + slot : nullptr);
    if (tid == 0 && a.phash) a.phash[slot] = hv;
    if (NDWD > 0) {
        // ---- K1' + K3 for dHash: 8 rows x 9 columns, each output = 3 chunk sums (threads 0..215)
        __syncthreads();
        int *part = reinterpret_cast<int *>(Lt);                 // 216 ints
        uint8_t *T98 = Lt + 1024;                                // 72 bytes
        if (tid < 216) {
            const int out = tid / 3, chunk = tid % 3;
            const int yy = out / 9, oc = out % 9;
            const int vc = yy * 3 + chunk;
            const int vst = a.vd_cstart[vc];
            const int nd = a.ndwcv;
            int d0 = 0, d1 = 0, d2 = 0;
            for (int j = 0; j < nd; j += 2) {
                const int32_t *cp = a.vd_cpacked + ((size_t)vc * nd + j) * 3;
                const uint2 v = *reinterpret_cast<const uint2 *>(__builtin_assume_aligned(HTd + (size_t)oc * a.hpd + vst + 4 * j, 8));
                d0 = __builtin_amdgcn_sdot4((int)v.x, cp[0], d0, false);
                d1 = __builtin_amdgcn_sdot4((int)v.x, cp[1], d1, false);
                d2 = __builtin_amdgcn_sdot4((int)v.x, cp[2], d2, false);
                d0 = __builtin_amdgcn_sdot4((int)v.y, cp[3], d0, false);
                d1 = __builtin_amdgcn_sdot4((int)v.y, cp[4], d1, false);
                d2 = __builtin_amdgcn_sdot4((int)v.y, cp[5], d2, false);
            }
            part[tid] = combine_planes(d0, d1, d2, 0);
        }
        __syncthreads();
        if (tid < 72) {
            const int yy = tid / 9;
            const uint32_t sum = (uint32_t)part[3 * tid] + (uint32_t)part[3 * tid + 1] + (uint32_t)part[3 * tid + 2] + (uint32_t)a.vd_bias[yy];
            T98[tid] = (uint8_t)clip8_fixed((int)sum);
        }
        __syncthreads();
        if (a.tile98_out && tid < 72) a.tile98_out[(size_t)img * 72 + tid] = T98[tid];
        if (tid < 64) {
            const int r = tid >> 3, c = tid & 7;
            const unsigned long long m = __ballot(T98[r * 9 + c + 1] > T98[r * 9 + c]);   // src/sig/phash.py:52
            if (tid == 0 && a.dhash) a.dhash[a.out_idx ? a.out_idx[img] : img] = __brevll(m);
        }
    }
}

int upload_dct_tables(ke_ctx *ctx) {
    if (ctx->dct_tables_ready) return KE_OK;
    KE_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_C32), KE_C32, sizeof(KE_C32)));
    KE_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_C16), KE_C16, sizeof(KE_C16)));
    KE_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_C8), KE_C8, sizeof(KE_C8)));
    ctx->dct_tables_ready = true;
    return KE_OK;
}

// ---------------------------------------------------------------------------------------
// Fused pHash (+ dHash) kernel.  Template: W = 64*W64 pixels per row (RGB, 3 bytes/pixel), KS = operand
// steps of the horizontal taps, DH = also produce dHash.  One workgroup (256 threads) per image; rows are
// streamed in tiles of 32 and the luma tile is double-buffered in LDS as signed bytes (L - 128).
//
// K1b on the matrix cores: the horizontal resample of a tile is the banded product
// H[row][o] = sum_x luma[row][x] * k[o][x], which v_mfma_i32_16x16x64_i8 accumulates exactly in int32 --
// A = 16 rows x 64 luma bytes straight out of LDS (one ds_read_b128 per lane), B = one byte plane of the
// taps of 16 outputs (KeMxTable, resident in registers); the three planes of the 22-bit weights are combined
// afterwards, k = b0 + 256 b1 + 65536 b2.  Wave w owns rows 16(w>>1).. of the tile and outputs 16(w&1)..;
// its result registers hold four consecutive rows of one output column, i.e. one dword of the transposed
// HT the vertical pass reads.  With the taps on the VALU (v_dot4_i32_i8, 144 per lane and 16 rows) the kernel
// was co-limited by VALU issue at 6.5 TB/s; one byte plane less in an experiment already reached the
// streaming-read ceiling (benchmarks/micro/hbm_read.hip: 7.0-7.1 TB/s), which is where this version sits.
// LDS rows are padded by 16 bytes so the 16 rows of an A operand fall into distinct banks.
//
// Software pipeline: while the products of tile t run out of Lt[t&1], the 12-byte loads of tile t+1 are in
// flight; they are converted and written to Lt[(t+1)&1] at the end of the iteration.  One barrier per tile;
// the raw registers live inside one iteration only (a loop-carried register set makes hipcc wait for the
// loads right after issuing them), and two sched_barriers keep the loads at the top and their first use at
// the bottom.  The last tile is peeled: it has nothing to prefetch (a guarded prefetch would make hipcc wait
// for the loads right after issuing them, an unguarded one would fetch 1/ntiles more bytes from HBM).
// ---------------------------------------------------------------------------------------
constexpr int kRTM = 32;
typedef int ke_v4i __attribute__((ext_vector_type(4)));

// dHash leg (DH): the 9-output axis is one more 16-column operand tile whose taps span the whole row,
// ceil(W/64) steps.  The two waves of a row block take one half of those steps each (KD per wave, columns
// [0, 64 KD) and [64 KD, 128 KD)); the jt = 1 wave hands its plane-combined partial sums to its partner
// through a 1 KB LDS slot, and the partner adds, clips and writes the transposed dHash column one tile
// later (after the tile's barrier).
// GEN: the row length is a run-time value, any multiple of 4 in (64*(W64-1), 64*W64]; W64 and KS are then upper
// bounds (operand steps past an output tile's window hold zero taps; what they read of the next row is ignored).
// C: bytes per pixel, 3 (RGB), 4 (RGBX/RGBA: the fourth byte is ignored, as convert("L") ignores it) or 1 ("L").
template <int W64, int KS, bool DH, bool GEN, int C, bool UNAL>
__global__ __launch_bounds__(256, 2) void ke_phash_fused_mx(const KeFusedArgs a) {
    static_assert(!UNAL || GEN, "rows that are not a multiple of 4 pixels need the run-time row length");
    constexpr uint32_t QB = 4u * C;              // bytes of a 4-pixel quad
    constexpr int W = 64 * W64;
    constexpr int QPT = kRTM * (W / 4) / 256;    // (upper bound of the) quads per thread per 32-row tile
    const int QW = GEN ? a.qw : W / 4;           // 12-byte quads per row
    const int QUADS_PER_TILE = kRTM * QW;
    const int LP = GEN ? a.lp : W + 16;
    constexpr int KD = (W64 + 1) / 2;            // dHash steps per wave
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *Lt = smem;
    uint8_t *HT = smem + a.lt_bytes;
    uint8_t *HTd = HT + 32 * a.hp;
    uint8_t *X = smem + a.x_off;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int64_t img = a.bands ? blockIdx.x / a.bands : blockIdx.x;
    const int y_begin = a.bands ? (int)(blockIdx.x % a.bands) * a.band_rows : 0;
    const uint32_t row_bytes = UNAL ? (uint32_t)a.row_bytes : (uint32_t)QW * QB;
    const uint8_t *src = a.pixels + (a.offsets ? a.offsets[img] : (uint64_t)img * a.stride) + (size_t)y_begin * row_bytes;
    const int h = a.bands ? min(a.band_rows, a.h - y_begin) : a.h;          // rows this workgroup turns into columns
    const uint32_t image_bytes = (uint32_t)(a.h - y_begin) * row_bytes;     // bytes from src to the end of the image

    // this wave's slice of the product: rows 16*mt.., outputs 16*jt..
    const int mt = wv >> 1, jt = wv & 1;
    ke_v4i bf[KS][3];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p)
            bf[s][p] = reinterpret_cast<const ke_v4i *>(a.mx_frag)[((jt * KS + s) * 3 + p) * 64 + lane];
    constexpr int KDR = DH ? KD : 1;
    ke_v4i bd[KDR][3];
    int dbias = 0;
    if (DH) {
#pragma unroll
        for (int d = 0; d < KDR; ++d)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                bd[d][p] = reinterpret_cast<const ke_v4i *>(a.mxd_frag)[((jt * KD + d) * 3 + p) * 64 + lane];
        dbias = a.hd_bias[(lane & 15) < 9 ? (lane & 15) : 8];
    }
    const int ocol = 16 * jt + (lane & 15);
    const int hbias = a.h_bias[ocol];
    const int a_off = (16 * mt + (lane & 15)) * LP + (jt ? a.mx_base1 : a.mx_base0) + 16 * (lane >> 4);
    const int d_off = (16 * mt + (lane & 15)) * LP + 64 * KD * jt + 16 * (lane >> 4);
    uint8_t *ht_dst = HT + (size_t)ocol * a.hp + 16 * mt + 4 * (lane >> 4);
    uint8_t *htd_dst = HTd + (size_t)(lane & 15) * a.hpd + 16 * mt + 4 * (lane >> 4);
    ke_v4i *x_slot = reinterpret_cast<ke_v4i *>(X + mt * 1024 + lane * 16);     // + 2048 for odd tiles

    uint32_t raw[QPT][C];
    auto load_tile = [&](int t) {
        const uint32_t tile_off = UNAL ? (uint32_t)t * (kRTM * row_bytes) : (uint32_t)t * ((uint32_t)QUADS_PER_TILE * QB);
        const uint8_t *tile_ptr = src + tile_off;                 // wave-uniform
        // last loadable quad, relative to the tile; slots past the tile's own quads (run-time row lengths leave some)
        // re-read its last quad instead of pulling the next tile's bytes a second time
        const uint32_t lim = UNAL ? image_bytes - QB - tile_off : min(image_bytes - QB - tile_off, (uint32_t)(QUADS_PER_TILE - 1) * QB);
#pragma unroll
        for (int q = 0; q < QPT; ++q) {
            // Unconditional loads (a guarded load makes hipcc wait vmcnt(0) after each one): quads past the image
            // end re-read the last quad; those rows only ever meet zero tap weights.  Streamed once: non-temporal.
            uint32_t vo = (uint32_t)(q * 256 + tid) * QB;
            if (UNAL) {                                           // rows are not quad-aligned: address by (row, quad)
                const uint32_t i = min((uint32_t)(q * 256 + tid), (uint32_t)(QUADS_PER_TILE - 1));
                const uint32_t row = __umulhi(i, (uint32_t)a.qw_inv);
                vo = row * row_bytes + (i - row * (uint32_t)QW) * QB;
            }
            const uint32_t off = vo < lim ? vo : lim;
            const uint32_t *p = reinterpret_cast<const uint32_t *>(tile_ptr + off);
#pragma unroll
            for (int k = 0; k < C; ++k) raw[q][k] = __builtin_nontemporal_load(p + k);
        }
    };
    auto store_luma = [&](uint8_t *dst) {
#pragma unroll
        for (int q = 0; q < QPT; ++q) {
            const int i = q * 256 + tid;
            if (GEN) {
                const int row = (int)__umulhi((uint32_t)i, (uint32_t)a.qw_inv);      // i / QW, exact for i < 2^16
                if (i < QUADS_PER_TILE)
                    *reinterpret_cast<uint32_t *>(dst + row * LP + (i - row * QW) * 4) = luma4_generic<C>(raw[q]);
            } else {
                *reinterpret_cast<uint32_t *>(dst + (i / (W / 4)) * LP + (i % (W / 4)) * 4) = luma4_generic<C>(raw[q]);
            }
        }
    };
    auto pack_rows = [](const ke_v4i &v0, const ke_v4i &v1, const ke_v4i &v2, int bias) {
        uint32_t packed = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) packed |= (uint32_t)clip8_fixed(combine_planes(v0[i], v1[i], v2[i], bias)) << (8 * i);
        return packed ^ 0x80808080u;    // signed bytes for the vertical pass
    };
    // products of one tile; returns this wave's dHash partial (plane-combined, no bias)
    auto products = [&](int t, const uint8_t *cur) -> ke_v4i {
        ke_v4i acc[3], dacc[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) acc[p] = dacc[p] = ke_v4i{0, 0, 0, 0};
        const uint8_t *ap = cur + a_off;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const ke_v4i av = *reinterpret_cast<const ke_v4i *>(__builtin_assume_aligned(ap + 64 * s, 16));
#pragma unroll
            for (int p = 0; p < 3; ++p) acc[p] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bf[s][p], acc[p], 0, 0, 0);
        }
        if (DH) {
            const uint8_t *dp = cur + d_off;
#pragma unroll
            for (int d = 0; d < KDR; ++d) {
                const ke_v4i av = *reinterpret_cast<const ke_v4i *>(__builtin_assume_aligned(dp + 64 * d, 16));
#pragma unroll
                for (int p = 0; p < 3; ++p) dacc[p] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bd[d][p], dacc[p], 0, 0, 0);
            }
        }
        *reinterpret_cast<uint32_t *>(ht_dst + t * kRTM) = pack_rows(acc[0], acc[1], acc[2], hbias);
        ke_v4i part;
#pragma unroll
        for (int i = 0; i < 4; ++i) part[i] = combine_planes(dacc[0][i], dacc[1][i], dacc[2][i], 0);
        return part;
    };
    ke_v4i carry = {0, 0, 0, 0};
    auto finish_dhash = [&](int t) {     // jt = 0 waves: tile t's dHash column from both halves
        const ke_v4i other = x_slot[(t & 1) * 128];
        const ke_v4i zero = {0, 0, 0, 0};
        const ke_v4i sum = carry + other;
        const uint32_t packed = pack_rows(sum, zero, zero, dbias);
        if ((lane & 15) < 9) *reinterpret_cast<uint32_t *>(htd_dst + t * kRTM) = packed;
    };
    auto hpass = [&](int t, const uint8_t *cur) {
        if (!DH) {
            products(t, cur);
        } else if (jt) {
            x_slot[(t & 1) * 128] = products(t, cur);
        } else {
            if (t > 0) finish_dhash(t - 1);
            carry = products(t, cur);
        }
    };

    const int ntiles = (h + kRTM - 1) / kRTM;
    load_tile(0);
    store_luma(Lt);
    __syncthreads();
    for (int t = 0; t + 1 < ntiles; ++t) {
        load_tile(t + 1);
        __builtin_amdgcn_sched_barrier(0);
        hpass(t, Lt + (t & 1) * a.lt_half);
        __builtin_amdgcn_sched_barrier(0);
        store_luma(Lt + ((t + 1) & 1) * a.lt_half);
        __syncthreads();
    }
    if (UNAL) {
        if (tid == 0 && y_begin + h == a.h)
            patch_last_quad<C>(src + (size_t)(h - 1) * row_bytes, a.w, Lt + ((ntiles - 1) & 1) * a.lt_half + ((h - 1) - (ntiles - 1) * kRTM) * LP);
        __syncthreads();
    }
    hpass(ntiles - 1, Lt + ((ntiles - 1) & 1) * a.lt_half);
    __syncthreads();
    if (DH) {
        if (!jt) finish_dhash(ntiles - 1);
        __syncthreads();
    }
    if (a.bands) {
        band_writeout(HT, a.hp, 32, h, a.hs + (size_t)img * 32 * a.hs_hp + y_begin, a.hs_hp, tid, 256);
        if (DH) band_writeout(HTd, a.hpd, 9, h, a.hsd + (size_t)img * 9 * a.hsd_hp + y_begin, a.hsd_hp, tid, 256);
        return;
    }
    fused_tail<(DH ? 1 : 0)>(a, Lt, HT, HTd, tid, img);
}

// Band mode of the single-pass kernels (see KeFusedArgs::bands): how a group is cut, and where the columns go.
struct KeBandPlan {
    int bands, band_rows;
    uint8_t *hs, *hsd;
    int hs_hp, hsd_hp;
};

template <int W64, int KS, bool DH, bool GEN = false, int C = 3, bool UNAL = false>
int launch_fused_mx(ke_ctx *ctx, const KeHashGroup &g, const KeAxisCoeffs *ch, const KeAxisCoeffs *cv, uint64_t *d_phash,
                    uint8_t *d_tile32, uint64_t *d_dhash, uint8_t *d_tile98, const KeBandPlan *plan = nullptr) {
    const int W = g.w;
    if (GEN ? (W > 64 * W64 || W <= 64 * (W64 - 1) || (W % 4 != 0) != UNAL) : W != 64 * W64) return KE_EUNSUPPORTED;
    const KeMxTable *mx = ke_get_mx(ctx, ch, KS);     // at least KS steps per tile (zero-padded)
    if (!mx) return ke_fail(ctx, KE_EHIP, "coefficient upload failed");
    if (mx->tiles != 2 || mx->ks != KS) return KE_EUNSUPPORTED;
    KeFusedArgs a;
    std::memset(&a, 0, sizeof a);
    a.pixels = g.pixels; a.offsets = g.offsets; a.stride = g.stride; a.out_idx = g.out_idx; a.h = g.h;
    a.h_bias = ch->d_bias;
    a.v_packed = cv->d_packed; a.v_start = cv->d_start; a.v_bias = cv->d_bias;
    a.ndwv = cv->ndw;
    a.mx_frag = mx->d_frag; a.mx_base0 = mx->base[0]; a.mx_base1 = mx->base[1];
    const int rows_padded = plan ? plan->band_rows : ((g.h + kRTM - 1) / kRTM) * kRTM;
    if (plan) { a.bands = plan->bands; a.band_rows = plan->band_rows; a.hs = plan->hs; a.hsd = plan->hsd; a.hs_hp = plan->hs_hp; a.hsd_hp = plan->hsd_hp; }
    // one tile buffer: 32 padded rows + the part of the last row's operand window that overhangs the row
    constexpr int KD = (W64 + 1) / 2;
    const int overhang = std::max(0, std::max(std::max(mx->base[0], mx->base[1]) + 64 * KS, DH ? 128 * KD : 0) - W);
    a.qw = (W + 3) / 4;
    a.w = W; a.row_bytes = W * C;
    a.qw_inv = (int)(uint32_t)((0x100000000ull + (uint64_t)a.qw - 1) / (uint64_t)a.qw);
    // row pitch: an odd number of 16-byte units, so the 16 rows of an operand land in distinct bank groups
    a.lp = GEN ? (((W + 15) / 16 + 1) | 1) * 16 : W + 16;
    a.lt_half = (kRTM * a.lp + overhang + 15) & ~15;
    a.lt_bytes = 2 * a.lt_half;
    a.hp = ((std::max(plan ? 0 : cv->span, rows_padded) + 7) & ~7) + 8;
    a.hpd = 8;
    a.phash = d_phash; a.tile32_out = d_tile32; a.margin = ctx->margin_cur;
    size_t lds = std::max<size_t>((size_t)a.lt_bytes + (size_t)32 * a.hp, 4096);
    if (DH) {
        const KeAxisCoeffs *chd = ke_get_coeffs(ctx, g.w, 9), *cvd = ke_get_coeffs(ctx, g.h, 8);
        if (!chd || !cvd) return ke_fail(ctx, KE_EHIP, "coefficient upload failed");
        const KeMxTable *mxd = ke_get_mx(ctx, chd, 2 * KD);     // the whole row in two halves of KD steps
        const KeChunkTable *tv = ke_get_chunks(ctx, cvd, 3);
        if (!mxd || !tv) return ke_fail(ctx, KE_EHIP, "coefficient upload failed");
        if (mxd->tiles != 1 || mxd->base[0] != 0 || mxd->ks != 2 * KD) return KE_EUNSUPPORTED;
        a.mxd_frag = mxd->d_frag;
        a.hd_bias = chd->d_bias;
        a.vd_cpacked = tv->d_cpacked; a.vd_cstart = tv->d_cstart; a.vd_bias = cvd->d_bias;
        a.ndwcv = tv->ndwc;
        a.hpd = ((std::max(plan ? 0 : tv->cspan, rows_padded) + 7) & ~7) + 8;
        a.dhash = d_dhash; a.tile98_out = d_tile98;
        lds += (size_t)9 * a.hpd;
        lds = (lds + 15) & ~(size_t)15;
        a.x_off = (int)lds;
        lds += 4096;
    }
    // up to 80 KB two workgroups share a CU.  A tall image (its 32 x H transposed columns) would need more and run one
    // per CU; if cutting it into bands brings the columns back under that line, decline: the caller then runs this
    // kernel per band (measured 5.4-6.0 TB/s against 3.9-4.9 for one workgroup per CU)
    if (lds > 150 * 1024) return KE_EUNSUPPORTED;
    if (!plan && lds > 80 * 1024 && lds - (size_t)(32 * (a.hp - 136)) - (DH ? (size_t)(9 * (a.hpd - 136)) : 0) <= 80 * 1024) return KE_EUNSUPPORTED;
    if (lds > 64 * 1024)
        KE_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&ke_phash_fused_mx<W64, KS, DH, GEN, C, UNAL>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (g.n * (plan ? plan->bands : 1) > 0x7fffffffLL) return ke_fail(ctx, KE_EUNSUPPORTED, "group too large for one launch");
    hipLaunchKernelGGL((ke_phash_fused_mx<W64, KS, DH, GEN, C, UNAL>), dim3((unsigned)(g.n * (plan ? plan->bands : 1))), dim3(256), lds, ctx->stream, a);
    KE_HIP(ctx, hipGetLastError());
    return KE_OK;
}

// ---------------------------------------------------------------------------------------
// Wide rows (up to 2048 pixels): the same chain with the work cut differently.  A 32-row tile of such a row
// no longer fits LDS twice and one wave cannot hold the operands of a whole output tile, so: 16-row tiles,
// 512 threads, wave = (output tile jt, quarter kh of that tile's operand steps).  Each wave multiplies its KSH
// steps; the quarters kh = 1..3 hand their plane-combined partial sums (int32, wrap-around arithmetic, so the
// order of the additions does not matter) to the kh = 0 wave through LDS, which adds the bias, clips and writes
// the transposed column one tile later -- the exchange of the dHash leg above, four ways.  With DH the eight waves
// also take an eighth each of the dHash axis' operand steps (KDW per wave) and meet in wave 0 the same way.
// ---------------------------------------------------------------------------------------
constexpr int kRTW = 16;

template <int KSH, int QPT, bool DH, int KDW, int C, bool UNAL>
__global__ __launch_bounds__(512, 1) void ke_phash_fused_wide(const KeFusedArgs a) {
    constexpr uint32_t QB = 4u * C;              // bytes of a 4-pixel quad (RGB or RGBX)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *Lt = smem;
    uint8_t *HT = smem + a.lt_bytes;
    uint8_t *HTd = HT + 32 * a.hp;
    uint8_t *X = smem + a.x_off;                 // [tile parity][jt][kh - 1][lane] x 16 B, then [parity][wave - 1][lane] for dHash
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int jt = wv & 1, kh = wv >> 1;
    const int64_t img = a.bands ? blockIdx.x / a.bands : blockIdx.x;
    const int y_begin = a.bands ? (int)(blockIdx.x % a.bands) * a.band_rows : 0;
    const int QW = a.qw, LP = a.lp;
    const uint32_t row_bytes = UNAL ? (uint32_t)a.row_bytes : (uint32_t)QW * QB;
    const uint8_t *src = a.pixels + (a.offsets ? a.offsets[img] : (uint64_t)img * a.stride) + (size_t)y_begin * row_bytes;
    const int h = a.bands ? min(a.band_rows, a.h - y_begin) : a.h;          // rows this workgroup turns into columns
    const int quads_per_tile = kRTW * QW;
    const uint32_t image_bytes = (uint32_t)(a.h - y_begin) * row_bytes;     // bytes from src to the end of the image

    ke_v4i bf[KSH][3];
#pragma unroll
    for (int s = 0; s < KSH; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p)
            bf[s][p] = reinterpret_cast<const ke_v4i *>(a.mx_frag)[((jt * 4 * KSH + kh * KSH + s) * 3 + p) * 64 + lane];
    const int ocol = 16 * jt + (lane & 15);
    const int hbias = a.h_bias[ocol];
    const int a_off = (lane & 15) * LP + (jt ? a.mx_base1 : a.mx_base0) + 64 * KSH * kh + 16 * (lane >> 4);
    uint8_t *ht_dst = HT + (size_t)ocol * a.hp + 4 * (lane >> 4);
    ke_v4i *x_slot = reinterpret_cast<ke_v4i *>(X) + jt * 192 + lane;     // + 64 per quarter, + 384 for odd tiles
    constexpr int KDR = DH ? KDW : 1;
    ke_v4i bd[KDR][3];
    int dbias = 0;
    if (DH) {
#pragma unroll
        for (int d = 0; d < KDR; ++d)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                bd[d][p] = reinterpret_cast<const ke_v4i *>(a.mxd_frag)[((wv * KDW + d) * 3 + p) * 64 + lane];
        dbias = a.hd_bias[(lane & 15) < 9 ? (lane & 15) : 8];
    }
    const int d_off = (lane & 15) * LP + 64 * KDW * wv + 16 * (lane >> 4);
    uint8_t *htd_dst = HTd + (size_t)(lane & 15) * a.hpd + 4 * (lane >> 4);
    ke_v4i *xd_slot = reinterpret_cast<ke_v4i *>(X) + 768 + lane;          // + 64 per wave, + 448 for odd tiles

    uint32_t raw[QPT][C];
    auto load_tile = [&](int t) {
        const uint32_t tile_off = UNAL ? (uint32_t)t * (kRTW * row_bytes) : (uint32_t)t * ((uint32_t)quads_per_tile * QB);
        const uint8_t *tile_ptr = src + tile_off;
        const uint32_t lim = UNAL ? image_bytes - QB - tile_off : min(image_bytes - QB - tile_off, (uint32_t)(quads_per_tile - 1) * QB);
#pragma unroll
        for (int q = 0; q < QPT; ++q) {           // unconditional, clamped, non-temporal (see ke_phash_fused_mx)
            uint32_t vo = (uint32_t)(q * 512 + tid) * QB;
            if (UNAL) {                                           // rows are not quad-aligned: address by (row, quad)
                const uint32_t i = min((uint32_t)(q * 512 + tid), (uint32_t)(quads_per_tile - 1));
                const uint32_t row = __umulhi(i, (uint32_t)a.qw_inv);
                vo = row * row_bytes + (i - row * (uint32_t)QW) * QB;
            }
            const uint32_t off = vo < lim ? vo : lim;
            const uint32_t *p = reinterpret_cast<const uint32_t *>(tile_ptr + off);
#pragma unroll
            for (int c = 0; c < C; ++c) raw[q][c] = __builtin_nontemporal_load(p + c);
        }
    };
    auto store_luma = [&](uint8_t *dst) {
#pragma unroll
        for (int q = 0; q < QPT; ++q) {
            const int i = q * 512 + tid;
            const int row = (int)__umulhi((uint32_t)i, (uint32_t)a.qw_inv);          // i / QW, exact for i < 2^16
            if (i < quads_per_tile)
                *reinterpret_cast<uint32_t *>(dst + row * LP + (i - row * QW) * 4) = luma4_generic<C>(raw[q]);
        }
    };
    auto products = [&](const uint8_t *cur) -> ke_v4i {
        ke_v4i acc[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) acc[p] = ke_v4i{0, 0, 0, 0};
        const uint8_t *ap = cur + a_off;
#pragma unroll
        for (int s = 0; s < KSH; ++s) {
            const ke_v4i av = *reinterpret_cast<const ke_v4i *>(__builtin_assume_aligned(ap + 64 * s, 16));
#pragma unroll
            for (int p = 0; p < 3; ++p) acc[p] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bf[s][p], acc[p], 0, 0, 0);
        }
        ke_v4i part;
#pragma unroll
        for (int i = 0; i < 4; ++i) part[i] = combine_planes(acc[0][i], acc[1][i], acc[2][i], 0);
        return part;
    };
    auto dproducts = [&](const uint8_t *cur) -> ke_v4i {
        ke_v4i dacc[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) dacc[p] = ke_v4i{0, 0, 0, 0};
        const uint8_t *dp = cur + d_off;
#pragma unroll
        for (int d = 0; d < KDR; ++d) {
            const ke_v4i av = *reinterpret_cast<const ke_v4i *>(__builtin_assume_aligned(dp + 64 * d, 16));
#pragma unroll
            for (int p = 0; p < 3; ++p) dacc[p] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bd[d][p], dacc[p], 0, 0, 0);
        }
        ke_v4i part;
#pragma unroll
        for (int i = 0; i < 4; ++i) part[i] = combine_planes(dacc[0][i], dacc[1][i], dacc[2][i], 0);
        return part;
    };
    ke_v4i carry = {0, 0, 0, 0}, dcarry = {0, 0, 0, 0};
    auto dfinish = [&](int t) {                   // wave 0: tile t's dHash column bytes from the eight parts
        const ke_v4i *xs = xd_slot + (t & 1) * 448;
        ke_v4i sum = dcarry;
#pragma unroll
        for (int k = 0; k < 7; ++k) sum += xs[64 * k];
        uint32_t packed = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) packed |= (uint32_t)clip8_fixed((int)((uint32_t)sum[i] + (uint32_t)dbias)) << (8 * i);
        if ((lane & 15) < 9) *reinterpret_cast<uint32_t *>(htd_dst + t * kRTW) = packed ^ 0x80808080u;
    };
    auto finish = [&](int t) {                    // kh = 0 waves: tile t's column bytes from the four quarters
        const ke_v4i *xs = x_slot + (t & 1) * 384;
        const ke_v4i sum = carry + xs[0] + xs[64] + xs[128];
        uint32_t packed = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) packed |= (uint32_t)clip8_fixed((int)((uint32_t)sum[i] + (uint32_t)hbias)) << (8 * i);
        *reinterpret_cast<uint32_t *>(ht_dst + t * kRTW) = packed ^ 0x80808080u;
    };
    auto hpass = [&](int t, const uint8_t *cur) {
        if (kh) {
            x_slot[(t & 1) * 384 + (kh - 1) * 64] = products(cur);
        } else {
            if (t > 0) finish(t - 1);
            carry = products(cur);
        }
        if (DH) {
            if (wv) {
                xd_slot[(t & 1) * 448 + (wv - 1) * 64] = dproducts(cur);
            } else {
                if (t > 0) dfinish(t - 1);
                dcarry = dproducts(cur);
            }
        }
    };

    const int ntiles = (h + kRTW - 1) / kRTW;
    load_tile(0);
    store_luma(Lt);
    __syncthreads();
    for (int t = 0; t + 1 < ntiles; ++t) {
        load_tile(t + 1);
        __builtin_amdgcn_sched_barrier(0);
        hpass(t, Lt + (t & 1) * a.lt_half);
        __builtin_amdgcn_sched_barrier(0);
        store_luma(Lt + ((t + 1) & 1) * a.lt_half);
        __syncthreads();
    }
    if (UNAL) {
        if (tid == 0 && y_begin + h == a.h)
            patch_last_quad<C>(src + (size_t)(h - 1) * row_bytes, a.w, Lt + ((ntiles - 1) & 1) * a.lt_half + ((h - 1) - (ntiles - 1) * kRTW) * LP);
        __syncthreads();
    }
    hpass(ntiles - 1, Lt + ((ntiles - 1) & 1) * a.lt_half);
    __syncthreads();
    if (!kh) finish(ntiles - 1);
    if (DH && !wv) dfinish(ntiles - 1);
    __syncthreads();
    if (a.bands) {
        band_writeout(HT, a.hp, 32, h, a.hs + (size_t)img * 32 * a.hs_hp + y_begin, a.hs_hp, tid, 512);
        if (DH) band_writeout(HTd, a.hpd, 9, h, a.hsd + (size_t)img * 9 * a.hsd_hp + y_begin, a.hsd_hp, tid, 512);
        return;
    }
    // the tail is written for 256 threads; the other four waves are done (ended waves leave the barrier count)
    if (tid >= 256) return;
    fused_tail<(DH ? 1 : 0)>(a, Lt, HT, HTd, tid, img);
}

template <int KSH, int QPT, bool DH, int KDW, int C = 3, bool UNAL = false>
int launch_fused_wide(ke_ctx *ctx, const KeHashGroup &g, const KeAxisCoeffs *ch, const KeAxisCoeffs *cv, uint64_t *d_phash,
                      uint8_t *d_tile32, uint64_t *d_dhash, uint8_t *d_tile98, const KeBandPlan *plan = nullptr) {
    const int W = g.w;
    if ((W % 4 != 0) != UNAL || kRTW * ((W + 3) / 4) > 512 * QPT) return KE_EUNSUPPORTED;
    const KeMxTable *mx = ke_get_mx(ctx, ch, 4 * KSH);
    if (!mx) return ke_fail(ctx, KE_EHIP, "coefficient upload failed");
    if (mx->tiles != 2 || mx->ks != 4 * KSH) return KE_EUNSUPPORTED;
    KeFusedArgs a;
    std::memset(&a, 0, sizeof a);
    a.pixels = g.pixels; a.offsets = g.offsets; a.stride = g.stride; a.out_idx = g.out_idx; a.h = g.h;
    a.h_bias = ch->d_bias;
    a.v_packed = cv->d_packed; a.v_start = cv->d_start; a.v_bias = cv->d_bias;
    a.ndwv = cv->ndw;
    a.mx_frag = mx->d_frag; a.mx_base0 = mx->base[0]; a.mx_base1 = mx->base[1];
    const int rows_padded = plan ? plan->band_rows : ((g.h + kRTW - 1) / kRTW) * kRTW;
    if (plan) { a.bands = plan->bands; a.band_rows = plan->band_rows; a.hs = plan->hs; a.hsd = plan->hsd; a.hs_hp = plan->hs_hp; a.hsd_hp = plan->hsd_hp; }
    const int overhang = std::max(0, std::max(std::max(mx->base[0], mx->base[1]) + 64 * 4 * KSH, DH ? 512 * KDW : 0) - W);
    a.qw = (W + 3) / 4;
    a.w = W; a.row_bytes = W * C;
    a.qw_inv = (int)(uint32_t)((0x100000000ull + (uint64_t)a.qw - 1) / (uint64_t)a.qw);
    a.lp = (((W + 15) / 16 + 1) | 1) * 16;
    a.lt_half = (kRTW * a.lp + overhang + 15) & ~15;
    a.lt_bytes = 2 * a.lt_half;
    a.hp = ((std::max(plan ? 0 : cv->span, rows_padded) + 7) & ~7) + 8;
    a.hpd = 8;
    a.phash = d_phash; a.tile32_out = d_tile32; a.margin = ctx->margin_cur;
    size_t lds = (size_t)a.lt_bytes + (size_t)32 * a.hp;
    if (DH) {
        const KeAxisCoeffs *chd = ke_get_coeffs(ctx, g.w, 9), *cvd = ke_get_coeffs(ctx, g.h, 8);
        if (!chd || !cvd) return ke_fail(ctx, KE_EHIP, "coefficient upload failed");
        const KeMxTable *mxd = ke_get_mx(ctx, chd, 8 * KDW);    // the whole row in eight parts of KDW steps
        const KeChunkTable *tv = ke_get_chunks(ctx, cvd, 3);
        if (!mxd || !tv) return ke_fail(ctx, KE_EHIP, "coefficient upload failed");
        if (mxd->tiles != 1 || mxd->base[0] != 0 || mxd->ks != 8 * KDW) return KE_EUNSUPPORTED;
        a.mxd_frag = mxd->d_frag;
        a.hd_bias = chd->d_bias;
        a.vd_cpacked = tv->d_cpacked; a.vd_cstart = tv->d_cstart; a.vd_bias = cvd->d_bias;
        a.ndwcv = tv->ndwc;
        a.hpd = ((std::max(plan ? 0 : tv->cspan, rows_padded) + 7) & ~7) + 8;
        a.dhash = d_dhash; a.tile98_out = d_tile98;
        lds += (size_t)9 * a.hpd;
    }
    lds = (lds + 15) & ~(size_t)15;
    a.x_off = (int)lds;
    lds += 2 * 2 * 3 * 1024 + (DH ? 2 * 7 * 1024 : 0);
    if (lds > 150 * 1024) return KE_EUNSUPPORTED;     // one workgroup per CU
    // ... unless bands of rows bring it under 80 KB (two per CU): then the caller runs this kernel per band
    if (!plan && lds > 80 * 1024 && lds - (size_t)(32 * (a.hp - 136)) - (DH ? (size_t)(9 * (a.hpd - 136)) : 0) <= 80 * 1024) return KE_EUNSUPPORTED;
    if (lds > 64 * 1024)
        KE_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&ke_phash_fused_wide<KSH, QPT, DH, KDW, C, UNAL>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (g.n * (plan ? plan->bands : 1) > 0x7fffffffLL) return ke_fail(ctx, KE_EUNSUPPORTED, "group too large for one launch");
    hipLaunchKernelGGL((ke_phash_fused_wide<KSH, QPT, DH, KDW, C, UNAL>), dim3((unsigned)(g.n * (plan ? plan->bands : 1))), dim3(512), lds, ctx->stream, a);
    KE_HIP(ctx, hipGetLastError());
    return KE_OK;
}

// ---------------------------------------------------------------------------------------
// Banded path: any width/height.  ke_hband streams a band of rows of one image (luma + the
// horizontal taps of `nout` output columns, windows cut into cpo = 2^k chunks that sit in
// adjacent lanes and are summed with xor shuffles) and writes the clipped bytes, transposed and
// re-biased to signed, into a small global scratch hs[img][column][row]; ke_vtile then runs the
// vertical taps out of LDS and writes the (oh x ow) tile.  Scratch traffic is nout*H bytes per
// image each way (4 % of the pixel bytes at 512 px, 0.3 % at 4096 px).
// ---------------------------------------------------------------------------------------
struct KeBandArgs {
    const uint8_t *pixels;
    const uint64_t *offsets;
    uint64_t stride;
    int w, h;
    int qr;            // 4-pixel quads per row = ceil(w / 4)
    int lp;            // LDS pitch of one luma row (bytes, multiple of 8)
    int rt;            // rows per tile (rt * qr <= 2048)
    int band_rows;     // rows per workgroup (multiple of rt)
    int bands;         // workgroups per image
    int nout;          // output columns handled by this launch
    int nout_total;    // output columns of the axis (32 or 9): pitch of the scratch
    int out_first;     // first output column of this launch
    int cpo_log2;      // log2(chunks per output)
    int vcp_log2;      // log2(lanes per row group) >= log2(nout << cpo_log2)
    const int32_t *cpacked, *cstart, *cxor, *bias;
    int lt_half;       // bytes of one LDS luma tile buffer
    int bp;            // pitch of one output column of the band staging area in LDS (bytes, multiple of 4)
    uint8_t *hs;       // [img][nout][hp]
    int hp;
};

template <int NDWC, int C, bool ALIGNED>
__global__ __launch_bounds__(256, 3) void ke_hband(const KeBandArgs a) {
    constexpr int QPT = 8;
    constexpr int DW = C;   // dwords per quad: 4 pixels * C bytes
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const int64_t img = blockIdx.x / a.bands;
    const int band = blockIdx.x % a.bands;
    const uint8_t *src = a.pixels + (a.offsets ? a.offsets[img] : (uint64_t)img * a.stride);
    const int y_begin = band * a.band_rows;
    const int y_end = min(a.h, y_begin + a.band_rows);
    const uintptr_t last_dword = ((uintptr_t)src + (uintptr_t)a.w * a.h * C - 1) & ~(uintptr_t)3;

    // lane -> (virtual column, row group)
    const int vcl = tid & ((1 << a.vcp_log2) - 1);
    const int rg = tid >> a.vcp_log2, RG = 256 >> a.vcp_log2;
    const int nvc = a.nout << a.cpo_log2;
    const int vloc = vcl < nvc ? vcl : nvc - 1;
    const int o = vloc >> a.cpo_log2;                       // output column inside this launch
    const int vc = vloc + (a.out_first << a.cpo_log2);      // virtual column in the axis' tables
    const bool writer = vcl < nvc && (vloc & ((1 << a.cpo_log2) - 1)) == 0;
    // step p reads the window dword pair p ^ xs (xs in 0..3, host-chosen so a half-wave's lanes spread over the
    // LDS banks); the tap planes are loaded in the same permuted order
    const int xs = a.cxor[vc];
    int ck[NDWC][3];
#pragma unroll
    for (int p = 0; p < NDWC / 2; ++p)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int c = 0; c < 3; ++c) ck[2 * p + e][c] = a.cpacked[((size_t)vc * NDWC + 2 * (p ^ xs) + e) * 3 + c];
    const int cst = a.cstart[vc];
    const int sw[4] = {cst + 8 * (0 ^ xs), cst + 8 * (1 ^ xs), cst + 8 * (2 ^ xs), cst + 8 * (3 ^ xs)};
    const int obias = a.bias[a.out_first + o];

    // quad items of a tile: item = q*256 + tid -> (row r, quad j).  The mapping is the same for every tile,
    // so the pixel offset inside the tile and the LDS byte offset are computed once per thread.
    const int items = a.rt * a.qr;
    const int total_pix = a.w * a.h;                  // < 2^31 (checked on the host)
    int rel_pix[QPT], lds_off[QPT];
    {
        int r = tid / a.qr, j = tid % a.qr;
        const int dr = 256 / a.qr, dj = 256 % a.qr;
#pragma unroll
        for (int q = 0; q < QPT; ++q) {
            rel_pix[q] = r * a.w + 4 * j;
            lds_off[q] = (q * 256 + tid < items) ? r * a.lp + 4 * j : -1;
            r += dr; j += dj;
            if (j >= a.qr) { j -= a.qr; ++r; }
        }
    }

    uint32_t raw[QPT][DW];
    // a load = (wave-uniform tile pointer) + (clamped loop-invariant 32-bit byte offset): one v_min per quad
    const uint32_t image_bytes = (uint32_t)total_pix * C;      // < 2^31 (checked on the host)
    auto load_tile = [&](int ty0) {
        const uint32_t tile_off = (uint32_t)(ty0 * a.w) * C;
        const uint8_t *tile_ptr = src + tile_off;
#pragma unroll
        for (int q = 0; q < QPT; ++q) {
            const uint32_t rel = (uint32_t)rel_pix[q] * C;
            if (ALIGNED) {
                const uint32_t lim = image_bytes - 4u * C - tile_off;   // w % 4 == 0: the last quad ends with the image
                const uint32_t off = rel < lim ? rel : lim;
                const uint32_t *p = reinterpret_cast<const uint32_t *>(tile_ptr + off);
#pragma unroll
                for (int k = 0; k < DW; ++k) raw[q][k] = __builtin_nontemporal_load(p + k);
            } else {
                const uint32_t lim = image_bytes - C - tile_off;
                const uintptr_t ad = (uintptr_t)tile_ptr + (rel < lim ? rel : lim);
                const uintptr_t al = ad & ~(uintptr_t)3;
                const int sb = (int)(ad & 3);
                uint32_t wv[DW + 1];
#pragma unroll
                for (int k = 0; k <= DW; ++k) {   // every dword load is clamped to the last dword of the image
                    const uintptr_t x = al + 4 * k;
                    wv[k] = *reinterpret_cast<const uint32_t *>(x < last_dword ? x : last_dword);
                }
#pragma unroll
                for (int k = 0; k < DW; ++k) raw[q][k] = __builtin_amdgcn_alignbyte(wv[k + 1], wv[k], sb);
            }
        }
    };
    auto store_luma = [&](uint8_t *dst) {
#pragma unroll
        for (int q = 0; q < QPT; ++q)
            if (lds_off[q] >= 0) *reinterpret_cast<uint32_t *>(dst + lds_off[q]) = luma4_generic<C>(raw[q]);
    };

    const int ntiles = (y_end - y_begin + a.rt - 1) / a.rt;
    // The band's output bytes are staged in LDS and written out once at the end: a global store inside the
    // row loop makes hipcc drain vmcnt(0) at the loop entry, which would wait for the prefetch just issued.
    uint8_t *HB = smem + 2 * a.lt_half;
    uint8_t *hb_col = HB + o * a.bp;
    const int cpo = 1 << a.cpo_log2;
    load_tile(y_begin);
    store_luma(smem);
    __syncthreads();
    auto hpass = [&](int t, const uint8_t *cur) {
        for (int r = rg; r < a.rt; r += RG) {
            const uint8_t *lrow = cur + r * a.lp;
            const uint8_t *lr4[4] = {lrow + sw[0], lrow + sw[1], lrow + sw[2], lrow + sw[3]};
            int d0 = 0, d1 = 0, d2 = 0;
#pragma unroll
            for (int p = 0; p < NDWC / 2; ++p) {
                const uint2 v = *reinterpret_cast<const uint2 *>(__builtin_assume_aligned(lr4[p & 3] + 8 * (p & ~3), 8));
                d0 = __builtin_amdgcn_sdot4((int)v.x, ck[2 * p][0], d0, false);
                d1 = __builtin_amdgcn_sdot4((int)v.x, ck[2 * p][1], d1, false);
                d2 = __builtin_amdgcn_sdot4((int)v.x, ck[2 * p][2], d2, false);
                d0 = __builtin_amdgcn_sdot4((int)v.y, ck[2 * p + 1][0], d0, false);
                d1 = __builtin_amdgcn_sdot4((int)v.y, ck[2 * p + 1][1], d1, false);
                d2 = __builtin_amdgcn_sdot4((int)v.y, ck[2 * p + 1][2], d2, false);
            }
            int part = combine_planes(d0, d1, d2, 0);
            // chunks sit in adjacent lanes (cpo <= 8): DPP butterflies inside the quad, then the upper quad of
            // an 8-lane group shifted down onto the lower one (only the group's first lane is read afterwards).
            // DPP runs in the VALU; wave shuffles (ds_bpermute) would add three LDS round trips per row.
            {
                const int x1 = __builtin_amdgcn_update_dpp(0, part, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
                part += cpo > 1 ? x1 : 0;
                const int x2 = __builtin_amdgcn_update_dpp(0, part, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
                part += cpo > 2 ? x2 : 0;
                const int x4 = __builtin_amdgcn_update_dpp(0, part, 0x104, 0xF, 0xF, true);   // row_shl:4
                part += cpo > 4 ? x4 : 0;
                const int x8 = __builtin_amdgcn_update_dpp(0, part, 0x108, 0xF, 0xF, true);   // row_shl:8
                part += cpo > 8 ? x8 : 0;
                if (cpo > 16) part += __shfl_down(part, 16);                                   // across DPP rows (wave-uniform branch)
            }
            if (writer) hb_col[t * a.rt + r] = (uint8_t)(clip8_fixed((int)((uint32_t)part + (uint32_t)obias)) ^ 0x80);
        }
    };
    for (int t = 0; t + 1 < ntiles; ++t) {     // last tile peeled: nothing to prefetch (see ke_phash_fused_mx)
        load_tile(y_begin + (t + 1) * a.rt);
        __builtin_amdgcn_sched_barrier(0);
        hpass(t, smem + (t & 1) * a.lt_half);
        __builtin_amdgcn_sched_barrier(0);
        store_luma(smem + ((t + 1) & 1) * a.lt_half);
        __syncthreads();
    }
    hpass(ntiles - 1, smem + ((ntiles - 1) & 1) * a.lt_half);
    __syncthreads();
    // write the band out: whole dwords along y (y_begin and bp are multiples of 4; hs rows are padded)
    {
        const int wpc = (y_end - y_begin + 3) / 4;   // dwords per output column
        uint8_t *dst = a.hs + ((size_t)img * a.nout_total + a.out_first) * a.hp + y_begin;
        for (int e = tid; e < a.nout * wpc; e += 256) {
            const int col = e / wpc, k = e % wpc;
            *reinterpret_cast<uint32_t *>(dst + (size_t)col * a.hp + 4 * k) = *reinterpret_cast<const uint32_t *>(HB + col * a.bp + 4 * k);
        }
    }
}

struct KeVtileArgs {
    const uint8_t *hs;     // [img][ow][hp] signed bytes
    int hp, ow, oh;
    const int32_t *packed, *start, *bias;   // vertical axis, oh outputs, ndw dwords
    int ndw;
    int cg;                // columns staged in LDS per round
    uint8_t *tiles;        // [img][oh][ow]
};

__global__ __launch_bounds__(256) void ke_vtile(const KeVtileArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const int64_t img = blockIdx.x;
    const uint8_t *hs = a.hs + (size_t)img * a.ow * a.hp;
    for (int c0 = 0; c0 < a.ow; c0 += a.cg) {
        const int nc = min(a.cg, a.ow - c0);
        const int words = nc * a.hp / 4;
        for (int e = tid; e < words; e += 256)
            reinterpret_cast<uint32_t *>(smem)[e] = reinterpret_cast<const uint32_t *>(hs + (size_t)c0 * a.hp)[e];
        __syncthreads();
        for (int e = tid; e < a.oh * nc; e += 256) {
            const int yy = e % a.oh, c = e / a.oh;
            const uint8_t *col = smem + (size_t)c * a.hp + a.start[yy];
            const int32_t *cp = a.packed + (size_t)yy * a.ndw * 3;
            int d0 = 0, d1 = 0, d2 = 0;
            for (int j = 0; j < a.ndw; j += 2) {
                const uint2 v = *reinterpret_cast<const uint2 *>(__builtin_assume_aligned(col + 4 * j, 8));
                d0 = __builtin_amdgcn_sdot4((int)v.x, cp[3 * j + 0], d0, false);
                d1 = __builtin_amdgcn_sdot4((int)v.x, cp[3 * j + 1], d1, false);
                d2 = __builtin_amdgcn_sdot4((int)v.x, cp[3 * j + 2], d2, false);
                d0 = __builtin_amdgcn_sdot4((int)v.y, cp[3 * j + 3], d0, false);
                d1 = __builtin_amdgcn_sdot4((int)v.y, cp[3 * j + 4], d1, false);
                d2 = __builtin_amdgcn_sdot4((int)v.y, cp[3 * j + 5], d2, false);
            }
            a.tiles[(size_t)img * a.oh * a.ow + (size_t)yy * a.ow + c0 + c] =
                (uint8_t)clip8_fixed(combine_planes(d0, d1, d2, a.bias[yy]));
        }
        __syncthreads();
    }
}

template <int NDWC, int C, bool ALIGNED>
int launch_hband_one(ke_ctx *ctx, const KeBandArgs &a, int64_t n, size_t lds) {
    hipLaunchKernelGGL((ke_hband<NDWC, C, ALIGNED>), dim3((unsigned)(n * a.bands)), dim3(256), lds, ctx->stream, a);
    KE_HIP(ctx, hipGetLastError());
    return KE_OK;
}

template <int C, bool ALIGNED>
int launch_hband_ndwc(ke_ctx *ctx, const KeBandArgs &a, int64_t n, size_t lds, int ndwc) {
    switch (ndwc) {
        case 8: return launch_hband_one<8, C, ALIGNED>(ctx, a, n, lds);
        case 16: return launch_hband_one<16, C, ALIGNED>(ctx, a, n, lds);
        case 24: return launch_hband_one<24, C, ALIGNED>(ctx, a, n, lds);
        case 32: return launch_hband_one<32, C, ALIGNED>(ctx, a, n, lds);
        default: return KE_EUNSUPPORTED;
    }
}

int launch_vtile(ke_ctx *ctx, const uint8_t *hs, int hp, int64_t n, int ow, int oh, const KeAxisCoeffs *cvt, uint8_t *d_tiles);

// Banded target: src images -> (oh x ow) u8 tiles.  Returns KE_EUNSUPPORTED for shapes it does not take
// (Pillow's vertical-first rule, images under 4 pixels, windows beyond 8 chunks of 32 dwords).
int resample_banded(ke_ctx *ctx, const KeHashGroup &g, int ow, int oh, uint8_t *d_tiles, int filter = KE_FILTER_LANCZOS,
                    const float *box = nullptr) {
    if ((int64_t)g.h > (int64_t)g.w * 100 && oh < g.h) return KE_EUNSUPPORTED;
    if ((int64_t)g.w * g.h < 4 || g.w > 16384 || g.h > 65536 || (int64_t)g.w * g.h * g.channels >= (1LL << 31)) return KE_EUNSUPPORTED;
    const KeAxisCoeffs *chz = box ? ke_get_coeffs(ctx, g.w, ow, filter, box[0], box[2]) : ke_get_coeffs(ctx, g.w, ow, filter);
    const KeAxisCoeffs *cvt = box ? ke_get_coeffs(ctx, g.h, oh, filter, box[1], box[3]) : ke_get_coeffs(ctx, g.h, oh, filter);
    if (!chz || !cvt) return ke_fail(ctx, KE_EHIP, "coefficient upload failed");
    // chunks per output: the power of two that wastes the fewest padded dwords, chunk <= 32 dwords; when the
    // virtual columns of all outputs do not fit the 256 lanes, the outputs are split over several launches
    int best_log2 = -1, best_waste = 1 << 30;
    for (int l = 0; l <= 5; ++l) {
        const int cpo = 1 << l, ndwc = (((chz->ndw + cpo - 1) / cpo) + 7) & ~7;
        if (ndwc > 32) continue;
        const int launches = (ow * cpo + 255) / 256;
        const int waste = cpo * ndwc - chz->ndw + (ndwc > 24 ? 8 : 0) + 1000 * (launches - 1);   // mild preference for <= 24 (3 waves/SIMD)
        if (waste < best_waste) { best_waste = waste; best_log2 = l; }
    }
    if (best_log2 < 0) return KE_EUNSUPPORTED;
    const KeChunkTable *tc = ke_get_chunks(ctx, chz, 1 << best_log2, 8);
    if (!tc) return ke_fail(ctx, KE_EHIP, "coefficient upload failed");
    KeBandArgs a;
    std::memset(&a, 0, sizeof a);
    a.pixels = g.pixels; a.offsets = g.offsets; a.stride = g.stride; a.w = g.w; a.h = g.h;
    a.qr = (g.w + 3) / 4;
    if (a.qr > 2048) return KE_EUNSUPPORTED;
    a.lp = (4 * a.qr + 7) & ~7;
    a.rt = std::max(1, std::min(2048 / a.qr, g.h));
    // Rows per workgroup.  Upper bounds: about 2 MB of pixels (amortises the tap-plane loads of a workgroup; measured
    // 4-6 % over 512 KB bands), 480 rows and 24 KB of LDS for the band's staged output bytes.  The image is then cut
    // into EQUAL bands (a short last band leaves its CU idle at the end), each a whole number of 4-tile groups so that
    // bands start on a multiple of 4 rows.
    const int unit = 4 * a.rt;
    const int per_launch = std::min(ow, 256 >> best_log2);   // outputs whose virtual columns fit 256 lanes
    int64_t cap_rows = std::max<int64_t>(unit, ((int64_t)(2048 << 10) / ((int64_t)g.w * g.channels)) / unit * unit);
    cap_rows = std::min<int64_t>(cap_rows, std::max(unit, 480 / unit * unit));
    cap_rows = std::min<int64_t>(cap_rows, std::max<int64_t>(unit, ((24 * 1024) / per_launch - 8) / unit * unit));
    // small groups still have to fill the chip: aim for at least ~3000 workgroups per launch (4 per SIMD slot)
    const int64_t min_bands = std::min<int64_t>((3072 + g.n - 1) / g.n, (g.h + unit - 1) / unit);
    if (min_bands > 1) cap_rows = std::min<int64_t>(cap_rows, std::max<int64_t>(unit, ((g.h + min_bands - 1) / min_bands + unit - 1) / unit * unit));
    const int64_t want_bands = (g.h + cap_rows - 1) / cap_rows;
    const int64_t rows = std::min<int64_t>(cap_rows, (((g.h + want_bands - 1) / want_bands + unit - 1) / unit) * unit);
    a.band_rows = (int)rows;
    a.bands = (g.h + a.band_rows - 1) / a.band_rows;
    a.bp = ((a.band_rows + 3) & ~3) + 4;
    a.nout_total = ow; a.cpo_log2 = best_log2;
    a.cpacked = tc->d_cpacked; a.cstart = tc->d_cstart; a.cxor = tc->d_cxor; a.bias = chz->d_bias;
    const int tile_bytes = a.rt * a.lp + std::max(0, tc->cspan - a.lp) + 16;
    a.lt_half = (tile_bytes + 15) & ~15;
    const size_t lds = 2 * (size_t)a.lt_half + (size_t)per_launch * a.bp;
    if (lds > 64 * 1024) return KE_EUNSUPPORTED;
    a.hp = ((std::max(cvt->span, g.h + 4) + 7) & ~7) + 8;
    void *hs;
    KE_TRY(ke_reserve(ctx, KE_BUF_TMP, (size_t)g.n * ow * a.hp + 8192, &hs));     // + slack: ke_vtile_mx reads whole steps
    a.hs = (uint8_t *)hs;
    if ((int64_t)g.n * a.bands > 0x7fffffffLL) return ke_fail(ctx, KE_EUNSUPPORTED, "group too large for one launch");
    // "aligned" = every row starts on a quad boundary of the packed stream (w % 4 == 0) and every image starts on a
    // dword boundary, so a quad is DW whole dwords and the last quad of the image ends with the image.  Other widths, and
    // ragged batches in which some image starts off a dword boundary, take the funnel-shift loader.
    const bool aligned = g.w % 4 == 0 && !g.misaligned;
    for (int first = 0; first < ow; first += per_launch) {
        a.out_first = first;
        a.nout = std::min(per_launch, ow - first);
        int vcp = 1, vl = 0;
        while (vcp < (a.nout << best_log2)) { vcp <<= 1; ++vl; }
        a.vcp_log2 = vl;
        int rc;
        if (g.channels == 3) rc = aligned ? launch_hband_ndwc<3, true>(ctx, a, g.n, lds, tc->ndwc) : launch_hband_ndwc<3, false>(ctx, a, g.n, lds, tc->ndwc);
        else if (g.channels == 1) rc = aligned ? launch_hband_ndwc<1, true>(ctx, a, g.n, lds, tc->ndwc) : launch_hband_ndwc<1, false>(ctx, a, g.n, lds, tc->ndwc);
        else rc = aligned ? launch_hband_ndwc<4, true>(ctx, a, g.n, lds, tc->ndwc) : launch_hband_ndwc<4, false>(ctx, a, g.n, lds, tc->ndwc);
        if (rc != KE_OK) return rc;
    }
    return launch_vtile(ctx, a.hs, a.hp, g.n, ow, oh, cvt, d_tiles);
}

// Vertical taps of the banded paths on the matrix cores: T[yy][col] = sum_y k[yy][y] * hs[col][y] is the product of the tap
// matrix (A: 16 outputs x 64 rows per step, the KeMxTable image of the vertical axis) with the transposed columns, which
// are exactly a B operand as they lie in the scratch (lane = (column, 16 consecutive rows)).  One wave per 16 x 16 block
// of the output tile, operands straight from global memory / L2, no LDS.  The one-thread-per-output ke_vtile it replaces
// walked each window serially with its taps fetched from global memory and took as long as the horizontal pass on
// 4096-row images.
struct KeVtileMxArgs {
    const uint8_t *hs;      // [img][ow][hp] signed bytes
    int hp, ow, oh;
    const int32_t *frag;    // KeMxTable::frag of the vertical axis
    int ks, mt, nt;         // steps per output tile, output-row tiles, column tiles
    int base[8];            // first row of each output-row tile's window (multiples of 16)
    const int32_t *bias;
    uint8_t *tiles;         // [img][oh][ow]
    int64_t items;          // n * mt * nt
};

__global__ __launch_bounds__(256) void ke_vtile_mx(const KeVtileMxArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= a.items) return;
    const int per = a.mt * a.nt;
    const int64_t img = item / per;
    const int r = (int)(item % per), mt = r / a.nt, nt = r % a.nt;
    const int col = 16 * nt + (lane & 15);
    const uint8_t *bp = a.hs + ((size_t)img * a.ow + min(col, a.ow - 1)) * a.hp + a.base[mt] + 16 * (lane >> 4);
    const ke_v4i *fr = reinterpret_cast<const ke_v4i *>(a.frag) + (size_t)mt * a.ks * 3 * 64 + lane;
    ke_v4i acc[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) acc[p] = ke_v4i{0, 0, 0, 0};
    for (int s = 0; s < a.ks; ++s) {
        ke_v4i bv;
        __builtin_memcpy(&bv, bp + 64 * s, 16);                      // 8-byte aligned: two dwordx2 or one dwordx4
#pragma unroll
        for (int p = 0; p < 3; ++p) acc[p] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fr[(s * 3 + p) * 64], bv, acc[p], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int yy = 16 * mt + 4 * (lane >> 4) + i;
        if (yy < a.oh && col < a.ow)
            a.tiles[((size_t)img * a.oh + yy) * a.ow + col] = (uint8_t)clip8_fixed(combine_planes(acc[0][i], acc[1][i], acc[2][i], a.bias[yy]));
    }
}

// vertical taps of the banded paths: hs[img][ow][hp] signed bytes -> (oh x ow) u8 tiles
int launch_vtile(ke_ctx *ctx, const uint8_t *hs, int hp, int64_t n, int ow, int oh, const KeAxisCoeffs *cvt, uint8_t *d_tiles) {
    if (oh <= 128 && !getenv("KE_VTILE_VALU")) {
        const KeMxTable *mx = ke_get_mx(ctx, cvt);
        if (!mx) return ke_fail(ctx, KE_EHIP, "coefficient upload failed");
        KeVtileMxArgs m;
        std::memset(&m, 0, sizeof m);
        m.hs = hs; m.hp = hp; m.ow = ow; m.oh = oh;
        m.frag = mx->d_frag; m.ks = mx->ks; m.mt = mx->tiles; m.nt = (ow + 15) / 16;
        for (int j = 0; j < mx->tiles; ++j) m.base[j] = mx->base[j];
        m.bias = cvt->d_bias; m.tiles = d_tiles;
        m.items = n * m.mt * m.nt;
        const int64_t blocks = (m.items + 3) / 4;
        if (blocks > 0x7fffffffLL) return ke_fail(ctx, KE_EUNSUPPORTED, "group too large for one launch");
        hipLaunchKernelGGL(ke_vtile_mx, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, m);
        KE_HIP(ctx, hipGetLastError());
        return KE_OK;
    }
    KeVtileArgs v;
    v.hs = hs; v.hp = hp; v.ow = ow; v.oh = oh;
    v.packed = cvt->d_packed; v.start = cvt->d_start; v.bias = cvt->d_bias; v.ndw = cvt->ndw;
    v.cg = std::max(1, std::min(ow, (48 * 1024) / hp));
    v.tiles = d_tiles;
    if ((size_t)v.cg * hp > 64 * 1024) return KE_EUNSUPPORTED;
    hipLaunchKernelGGL(ke_vtile, dim3((unsigned)n), dim3(256), (size_t)v.cg * hp, ctx->stream, v);
    KE_HIP(ctx, hipGetLastError());
    return KE_OK;
}

// ---------------------------------------------------------------------------------------
// Rows wider than 2048 pixels (photographs): the horizontal pass of the banded path on the matrix cores.
// A workgroup owns a band of rows of one image and walks it in 16-row tiles; a tile is taken in strips of
// 2048 pixels (16 rows x 2048 luma bytes = one LDS buffer, two buffers), and the int32 accumulators of an
// output tile live across the strips of a tile.  1024 threads: wave = (output tile jt, eighth kh of that
// tile's operand steps); a step belongs to exactly one strip (tile bases and strip edges are multiples of
// 64), so a wave multiplies in a strip the steps of its eighth that fall into it.  After the last strip the
// eighths meet in the kh = 0 wave through LDS (one tile later, as in ke_phash_fused_wide), which clips and
// stages the band's transposed bytes; they leave for the global scratch at the end of the band and
// ke_vtile / ke_tiles_to_hashes finish as for every banded image.
// ---------------------------------------------------------------------------------------
struct KeStripArgs {
    const uint8_t *pixels;
    const uint64_t *offsets;
    uint64_t stride;
    int w, h;
    int nstrips;            // ceil(w / 2048)
    int band_rows, bands;   // rows per workgroup (multiple of 16), workgroups per image
    const int32_t *mx_frag; // KeMxTable::frag, 8 * KSH steps per tile, bases multiples of 64
    int base0, base1;
    const int32_t *bias;
    int ncols;              // output columns of the axis: 32 (pHash) or 9 (dHash)
    int bp;                 // pitch of one output column of the band staging area (bytes, multiple of 4)
    int x_off, hb_off;      // LDS offsets of the exchange slots and of the staging area
    uint8_t *hs;            // [img][32][hp]
    int hp;
};

constexpr int kSW = 2048;                 // strip width in pixels: 16 rows of it are one LDS buffer
constexpr int kSLP = kSW + 16;            // LDS pitch of a strip row (129 x 16 B: conflict-free operand rows)

// The strip geometry is compile-time on purpose: a run-time strip width (equal strips, no nearly empty last one)
// costs row/column registers per load slot, spills at 1024 threads x 128 registers and ran 25 % slower.
// NT: 16-output tiles of the axis (2: the 32 pHash columns, 1: the 9 dHash columns); the 16 waves split into NT groups
// of PARTS = 16 / NT, one part of a tile's operand steps each.
template <int KSH, int NT>
__global__ __launch_bounds__(1024) void ke_hstrips(const KeStripArgs a) {
    constexpr int PARTS = 16 / NT;
    constexpr int QPT = 16 * (kSW / 4) / 1024;     // 8 quads per thread per strip
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *Lt = smem;
    ke_v4i *X = reinterpret_cast<ke_v4i *>(smem + a.x_off);      // [parity][jt][kh - 1][lane]
    uint8_t *HB = smem + a.hb_off;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jt = NT == 2 ? (wv & 1) : 0, kh = NT == 2 ? (wv >> 1) : wv;
    const int64_t img = blockIdx.x / a.bands;
    const int band = blockIdx.x % a.bands;
    const uint8_t *src = a.pixels + (a.offsets ? a.offsets[img] : (uint64_t)img * a.stride);
    const int y_begin = band * a.band_rows;
    const int y_end = min(a.h, y_begin + a.band_rows);
    const int ntiles = (y_end - y_begin + 15) / 16;
    const int S = a.nstrips, nit = ntiles * S;

    ke_v4i bf[KSH][3];
#pragma unroll
    for (int i = 0; i < KSH; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p)
            bf[i][p] = reinterpret_cast<const ke_v4i *>(a.mx_frag)[((jt * PARTS * KSH + kh * KSH + i) * 3 + p) * 64 + lane];
    const int x_first = (jt ? a.base1 : a.base0) + 64 * KSH * kh;     // first tap column of this wave's steps (wave-uniform)
    const int ocol = 16 * jt + (lane & 15);
    const int obias = a.bias[min(ocol, a.ncols - 1)];
    const int a_row = (lane & 15) * kSLP + 16 * (lane >> 4);
    constexpr int XPAR = NT * (PARTS - 1) * 64;                  // slots of one tile parity
    ke_v4i *x_slot = X + jt * (PARTS - 1) * 64 + lane;           // + 64 per part, + XPAR for odd tiles
    uint8_t *hb_dst = HB + (size_t)ocol * a.bp + 4 * (lane >> 4);

    // pixel quads of a strip: item = q*1024 + tid -> (row = item / 512, quad = item % 512)
    const uint32_t row_bytes = (uint32_t)a.w * 3u;
    uint32_t raw[QPT][3];
    auto load_it = [&](int it) {
        const int t = it / S, s = it - t * S;
        const int x0 = s * kSW;
        const int qmax = (min(kSW, a.w - x0) >> 2) - 1;             // last quad of this strip's row segment
#pragma unroll
        for (int q = 0; q < QPT; ++q) {
            const int item = q * 1024 + tid;
            const int y = min(y_begin + 16 * t + (item >> 9), a.h - 1);
            const int c = min(item & 511, qmax);
            const uint32_t *p = reinterpret_cast<const uint32_t *>(src + (uint32_t)y * row_bytes + (uint32_t)(x0 + 4 * c) * 3u);
            raw[q][0] = __builtin_nontemporal_load(p);
            raw[q][1] = __builtin_nontemporal_load(p + 1);
            raw[q][2] = __builtin_nontemporal_load(p + 2);
        }
    };
    auto store_luma = [&](uint8_t *dst) {
#pragma unroll
        for (int q = 0; q < QPT; ++q) {
            const int item = q * 1024 + tid;
            *reinterpret_cast<uint32_t *>(dst + (item >> 9) * kSLP + 4 * (item & 511)) = luma4_biased(raw[q][0], raw[q][1], raw[q][2]);
        }
    };
    ke_v4i acc[3], carry = {0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < 3; ++p) acc[p] = ke_v4i{0, 0, 0, 0};
    auto finish = [&](int t) {                     // kh = 0 waves: tile t's column bytes from the eight parts
        const ke_v4i *xs = x_slot + (t & 1) * XPAR;
        ke_v4i sum = carry;
#pragma nounroll
        for (int k = 0; k < PARTS - 1; ++k) sum += xs[64 * k];     // not unrolled: 15 slots in flight would spill
        uint32_t packed = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) packed |= (uint32_t)clip8_fixed((int)((uint32_t)sum[i] + (uint32_t)obias)) << (8 * i);
        if (ocol < a.ncols) *reinterpret_cast<uint32_t *>(hb_dst + 16 * t) = packed ^ 0x80808080u;
    };
    auto compute = [&](int it, const uint8_t *cur) {
        const int t = it / S, s = it - t * S;
        const uint8_t *ap = cur + a_row;
#pragma unroll
        for (int i = 0; i < KSH; ++i) {
            const int x = x_first + 64 * i;
            if ((x >> 11) == s) {                  // wave-uniform: this step's 64 columns lie in the strip
                const ke_v4i av = *reinterpret_cast<const ke_v4i *>(__builtin_assume_aligned(ap + (x & (kSW - 1)), 16));
#pragma unroll
                for (int p = 0; p < 3; ++p) acc[p] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bf[i][p], acc[p], 0, 0, 0);
            }
        }
        if (s == S - 1) {                          // the tile is complete
            ke_v4i part;
#pragma unroll
            for (int i = 0; i < 4; ++i) part[i] = combine_planes(acc[0][i], acc[1][i], acc[2][i], 0);
#pragma unroll
            for (int p = 0; p < 3; ++p) acc[p] = ke_v4i{0, 0, 0, 0};
            if (kh) {
                x_slot[(t & 1) * XPAR + (kh - 1) * 64] = part;
            } else {
                if (t > 0) finish(t - 1);
                carry = part;
            }
        }
    };

    load_it(0);
    store_luma(Lt);
    __syncthreads();
    for (int it = 0; it + 1 < nit; ++it) {
        load_it(it + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(it, Lt + (it & 1) * (16 * kSLP));
        __builtin_amdgcn_sched_barrier(0);
        store_luma(Lt + ((it + 1) & 1) * (16 * kSLP));
        __syncthreads();
    }
    compute(nit - 1, Lt + ((nit - 1) & 1) * (16 * kSLP));
    __syncthreads();
    if (!kh) finish(ntiles - 1);
    __syncthreads();
    // write the band out: whole dwords along y (y_begin and bp are multiples of 4; hs rows are padded)
    {
        const int wpc = (y_end - y_begin + 3) / 4;
        uint8_t *dst = a.hs + (size_t)img * a.ncols * a.hp + y_begin;
        for (int e = tid; e < a.ncols * wpc; e += 1024) {
            const int col = e / wpc, k = e % wpc;
            *reinterpret_cast<uint32_t *>(dst + (size_t)col * a.hp + 4 * k) = *reinterpret_cast<const uint32_t *>(HB + col * a.bp + 4 * k);
        }
    }
}

int launch_vtile(ke_ctx *ctx, const uint8_t *hs, int hp, int64_t n, int ow, int oh, const KeAxisCoeffs *cvt, uint8_t *d_tiles);

// (32 x 32) or (8 x 9) tile of packed RGB images 2816..5300 pixels wide, large groups only
int resample_strips(ke_ctx *ctx, const KeHashGroup &g, int ow, int oh, uint8_t *d_tiles) {
    if (!((ow == 32 && oh == 32) || (ow == 9 && oh == 8))) return KE_EUNSUPPORTED;
    if (g.misaligned || g.channels != 3 || g.w % 4 || g.w <= 2048 || (int64_t)g.h > (int64_t)g.w * 100 || g.h == oh) return KE_EUNSUPPORTED;
    if ((int64_t)g.w * g.h * 3 >= (1LL << 31) || (uintptr_t)g.pixels % 4 || !(g.offsets || g.stride % 4 == 0)) return KE_EUNSUPPORTED;
    const KeAxisCoeffs *chz = ke_get_coeffs(ctx, g.w, ow);
    const KeAxisCoeffs *cvt = ke_get_coeffs(ctx, g.h, oh);
    if (!chz || !cvt) return ke_fail(ctx, KE_EHIP, "coefficient upload failed");
    const int nt = (ow + 15) / 16, parts = 16 / nt;
    // natural step count with 64-aligned bases decides the instantiation
    int ks = 0;
    for (int j = 0; j < nt; ++j) {
        int lo = g.w, hi = 0;
        for (int o = 16 * j; o < std::min(16 * j + 16, ow); ++o) { lo = std::min(lo, chz->bounds[2 * o]); hi = std::max(hi, chz->bounds[2 * o] + chz->bounds[2 * o + 1]); }
        ks = std::max(ks, (hi - (lo & ~63) + 63) / 64);
    }
    const int ksh = std::max((ks + parts - 1) / parts, 3);
    if (ksh > 6) return KE_EUNSUPPORTED;
    const KeMxTable *mx = ke_get_mx(ctx, chz, parts * ksh, true);
    if (!mx) return ke_fail(ctx, KE_EHIP, "coefficient upload failed");
    if (mx->tiles != nt || mx->ks != parts * ksh || mx->base[0] % 64 || (nt == 2 && mx->base[1] % 64)) return KE_EUNSUPPORTED;
    KeStripArgs a;
    std::memset(&a, 0, sizeof a);
    a.pixels = g.pixels; a.offsets = g.offsets; a.stride = g.stride; a.w = g.w; a.h = g.h;
    a.nstrips = (g.w + kSW - 1) / kSW;
    // a nearly empty last strip costs a full strip of load slots: rows that leave less than 3/8 of one stay banded
    if (g.w - (a.nstrips - 1) * kSW < 768) return KE_EUNSUPPORTED;
    // bands of up to 256 rows, equal, multiples of 16; only worth it when the bands fill the chip
    const int64_t want_bands = (g.h + 255) / 256;
    a.band_rows = (int)((((g.h + want_bands - 1) / want_bands) + 15) / 16 * 16);
    a.bands = (g.h + a.band_rows - 1) / a.band_rows;
    // few workgroups cannot fill the chip: such groups keep the banded kernel (smaller bands, up to 3072 workgroups);
    // KE_FUSED_MIN_IMAGES, when set, decides instead (as for the single-pass kernels)
    if (const char *e = getenv("KE_FUSED_MIN_IMAGES")) {
        if (g.n < atoll(e)) return KE_EUNSUPPORTED;
    } else if (g.n * a.bands < 2 * (int64_t)ctx->cu_count) {
        return KE_EUNSUPPORTED;
    }
    a.mx_frag = mx->d_frag; a.base0 = mx->base[0]; a.base1 = nt == 2 ? mx->base[1] : 0;
    a.bias = chz->d_bias;
    a.ncols = ow;
    a.bp = a.band_rows + 4;
    size_t lds = 2 * (size_t)16 * kSLP;
    a.x_off = (int)lds; lds += (size_t)2 * nt * (parts - 1) * 1024;
    a.hb_off = (int)lds; lds += (size_t)ow * a.bp;
    a.hp = ((std::max(cvt->span, g.h + 4) + 7) & ~7) + 8;
    void *hs;
    KE_TRY(ke_reserve(ctx, KE_BUF_TMP, (size_t)g.n * ow * a.hp + 8192, &hs));
    a.hs = (uint8_t *)hs;
    if ((int64_t)g.n * a.bands > 0x7fffffffLL) return ke_fail(ctx, KE_EUNSUPPORTED, "group too large for one launch");
    const dim3 grid((unsigned)(g.n * a.bands)), blk(1024);
#define KE_STRIPS(K, T) do { \
        KE_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&ke_hstrips<K, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((ke_hstrips<K, T>), grid, blk, lds, ctx->stream, a); } while (0)
    if (nt == 2) {
        switch (ksh) { case 3: KE_STRIPS(3, 2); break; case 4: KE_STRIPS(4, 2); break; case 5: KE_STRIPS(5, 2); break; default: KE_STRIPS(6, 2); break; }
    } else {
        switch (ksh) { case 3: KE_STRIPS(3, 1); break; case 4: KE_STRIPS(4, 1); break; case 5: KE_STRIPS(5, 1); break; default: KE_STRIPS(6, 1); break; }
    }
#undef KE_STRIPS
    KE_HIP(ctx, hipGetLastError());
    return launch_vtile(ctx, a.hs, a.hp, g.n, ow, oh, cvt, d_tiles);
}

// Generic target: src images -> (oh x ow) u8 tiles, two single-axis passes in Pillow's order.
int resample_generic(ke_ctx *ctx, const KeHashGroup &g, int ow, int oh, uint8_t *d_tiles, int filter = KE_FILTER_LANCZOS,
                     const float *box = nullptr) {
    const KeAxisCoeffs *chz = box ? ke_get_coeffs(ctx, g.w, ow, filter, box[0], box[2]) : ke_get_coeffs(ctx, g.w, ow, filter);
    const KeAxisCoeffs *cvt = box ? ke_get_coeffs(ctx, g.h, oh, filter, box[1], box[3]) : ke_get_coeffs(ctx, g.h, oh, filter);
    if (!chz || !cvt) return ke_fail(ctx, KE_EHIP, "coefficient upload failed");
    // Pillow's Image.resize shrinks very tall, narrow images vertically first (PIL/Image.py).
    const bool vertical_first = (int64_t)g.h > (int64_t)g.w * 100 && oh < g.h;
    const int mid_w = vertical_first ? g.w : ow, mid_h = vertical_first ? oh : g.h;
    const size_t mid_bytes = (size_t)mid_w * mid_h;
    const int64_t max_n = std::max<int64_t>(1, (int64_t)(((size_t)512 << 20) / mid_bytes));
    if (g.n > max_n) {   // bound the first-pass scratch: split the group
        for (int64_t f = 0; f < g.n; f += max_n) {
            KeHashGroup s = g;
            s.n = std::min(max_n, g.n - f);
            if (g.offsets) s.offsets = g.offsets + f; else s.pixels = g.pixels + (size_t)f * g.stride;
            KE_TRY(resample_generic(ctx, s, ow, oh, d_tiles + (size_t)f * ow * oh, filter, box));
        }
        return KE_OK;
    }
    void *tmp;
    KE_TRY(ke_reserve(ctx, KE_BUF_TMP, (size_t)g.n * mid_bytes, &tmp));
    const int b0 = (int)((mid_bytes + 255) / 256), b1 = (ow * oh + 255) / 256;
    if ((int64_t)g.n * b0 > 0x7fffffffLL) return ke_fail(ctx, KE_EUNSUPPORTED, "group too large for one launch");
    if (!vertical_first) {
        hipLaunchKernelGGL(ke_resample_pass<0>, dim3((unsigned)(g.n * b0)), dim3(256), 0, ctx->stream, g.pixels, g.offsets,
                           g.stride, g.channels, g.w, g.h, chz->d_bounds, chz->d_kk, chz->ksize, ow, (uint8_t *)tmp, b0);
        hipLaunchKernelGGL(ke_resample_pass<1>, dim3((unsigned)(g.n * b1)), dim3(256), 0, ctx->stream, (const uint8_t *)tmp,
                           (const uint64_t *)nullptr, (uint64_t)mid_bytes, 1, mid_w, mid_h, cvt->d_bounds, cvt->d_kk,
                           cvt->ksize, oh, d_tiles, b1);
    } else {
        hipLaunchKernelGGL(ke_resample_pass<1>, dim3((unsigned)(g.n * b0)), dim3(256), 0, ctx->stream, g.pixels, g.offsets,
                           g.stride, g.channels, g.w, g.h, cvt->d_bounds, cvt->d_kk, cvt->ksize, oh, (uint8_t *)tmp, b0);
        hipLaunchKernelGGL(ke_resample_pass<0>, dim3((unsigned)(g.n * b1)), dim3(256), 0, ctx->stream, (const uint8_t *)tmp,
                           (const uint64_t *)nullptr, (uint64_t)mid_bytes, 1, mid_w, mid_h, chz->d_bounds, chz->d_kk,
                           chz->ksize, ow, d_tiles, b1);
    }
    KE_HIP(ctx, hipGetLastError());
    return KE_OK;
}


// Picks the single-pass kernel for a group (row length, bytes per pixel, one or both hashes) and launches it.
// plan == NULL: one workgroup per image, hashes (and optional tiles) written directly.  plan != NULL: band mode, the
// transposed columns go to plan->hs / plan->hsd and the caller finishes with ke_vtile.  *did_d: dHash was covered too.
// Returns KE_EUNSUPPORTED when no single-pass kernel takes the shape (or its LDS needs exceed the CU).
int dispatch_single_pass(ke_ctx *ctx, const KeHashGroup &g, bool want_d, uint64_t *d_phash, uint8_t *d_t32, uint64_t *d_dhash,
                         uint8_t *d_t98, const KeBandPlan *plan, bool *did_d) {
    *did_d = false;
    const bool unal = g.w % 4 != 0;                // rows that do not end on a 4-pixel boundary: RGB, pHash leg only
    if (g.misaligned) return KE_EUNSUPPORTED;      // some image of the group does not start on a dword boundary: banded funnel-shift loader
    if (g.w <= 64 || g.w > (unal ? 1024 : g.channels == 3 ? 2816 : 2048) || g.h == 32 || g.h < 16 || (unal && g.channels != 3) ||
        (!plan && g.h > 4096) || (int64_t)g.h > (int64_t)g.w * 100 ||
        (!unal && ((uintptr_t)g.pixels % 4 || !(g.offsets || g.stride % 4 == 0))) ||
        (int64_t)g.w * g.h * g.channels >= (1LL << 31))
        return KE_EUNSUPPORTED;
    // ragged groups: ke_hash_images checks every offset and sets g.misaligned when one is not a multiple of 4
    const KeAxisCoeffs *ch = ke_get_coeffs(ctx, g.w, 32);
    const KeAxisCoeffs *cv = ke_get_coeffs(ctx, g.h, 32);
    if (!ch || !cv) return ke_fail(ctx, KE_EHIP, "coefficient upload failed");
    const bool both = want_d && g.h != 8 && g.channels == 3 && !unal;
    int rc = KE_EUNSUPPORTED;
#define KE_MX(W64, KS, DH, GEN, C) rc = launch_fused_mx<W64, KS, DH, GEN, C>(ctx, g, ch, cv, d_phash, d_t32, DH ? d_dhash : nullptr, DH ? d_t98 : nullptr, plan)
#define KE_WIDE(KSH, QPT, DH, KDW, C) rc = launch_fused_wide<KSH, QPT, DH, KDW, C>(ctx, g, ch, cv, d_phash, d_t32, DH ? d_dhash : nullptr, DH ? d_t98 : nullptr, plan)
    if (unal) {
        if (g.w <= 704) {
            switch ((g.w + 63) / 64) {
                case 2: rc = launch_fused_mx<2, 2, false, true, 3, true>(ctx, g, ch, cv, d_phash, d_t32, nullptr, nullptr, plan); break;
                case 3: rc = launch_fused_mx<3, 2, false, true, 3, true>(ctx, g, ch, cv, d_phash, d_t32, nullptr, nullptr, plan); break;
                case 4: rc = launch_fused_mx<4, 3, false, true, 3, true>(ctx, g, ch, cv, d_phash, d_t32, nullptr, nullptr, plan); break;
                case 5: rc = launch_fused_mx<5, 3, false, true, 3, true>(ctx, g, ch, cv, d_phash, d_t32, nullptr, nullptr, plan); break;
                case 6: rc = launch_fused_mx<6, 4, false, true, 3, true>(ctx, g, ch, cv, d_phash, d_t32, nullptr, nullptr, plan); break;
                case 7: rc = launch_fused_mx<7, 5, false, true, 3, true>(ctx, g, ch, cv, d_phash, d_t32, nullptr, nullptr, plan); break;
                case 8: rc = launch_fused_mx<8, 5, false, true, 3, true>(ctx, g, ch, cv, d_phash, d_t32, nullptr, nullptr, plan); break;
                case 9: rc = launch_fused_mx<9, 6, false, true, 3, true>(ctx, g, ch, cv, d_phash, d_t32, nullptr, nullptr, plan); break;
                case 10: rc = launch_fused_mx<10, 6, false, true, 3, true>(ctx, g, ch, cv, d_phash, d_t32, nullptr, nullptr, plan); break;
                default: rc = launch_fused_mx<11, 7, false, true, 3, true>(ctx, g, ch, cv, d_phash, d_t32, nullptr, nullptr, plan); break;
            }
            if (rc != KE_EUNSUPPORTED) return rc;
        }
        // wider rows stay on the banded kernel (funnel-shift loader): measured 3.9 TB/s here at 1599 pixels against 5.3 there
        if (g.w <= 1024) rc = launch_fused_wide<3, 8, false, 1, 3, true>(ctx, g, ch, cv, d_phash, d_t32, nullptr, nullptr, plan);
        return rc;
    }
    // ---- rows up to 768 pixels: 256-thread kernel, 32-row tiles
    if (g.w <= 768) {
        if (both && g.w <= 512) {                  // pHash + dHash in one pass over the pixels
            switch (g.w) {
                case 256: KE_MX(4, 3, true, false, 3); break;
                case 384: KE_MX(6, 4, true, false, 3); break;
                case 512: KE_MX(8, 5, true, false, 3); break;
                default:
                    switch ((g.w + 63) / 64) {
                        case 2: KE_MX(2, 2, true, true, 3); break;
                        case 3: KE_MX(3, 2, true, true, 3); break;
                        case 4: KE_MX(4, 3, true, true, 3); break;
                        case 5: KE_MX(5, 3, true, true, 3); break;
                        case 6: KE_MX(6, 4, true, true, 3); break;
                        case 7: KE_MX(7, 5, true, true, 3); break;
                        case 8: KE_MX(8, 5, true, true, 3); break;
                        default: break;
                    }
            }
            if (rc == KE_OK) { *did_d = true; return KE_OK; }
            if (rc != KE_EUNSUPPORTED) return rc;
            if (!plan) return KE_EUNSUPPORTED;     // both legs did not fit one workgroup per image: per band they do (one pass)
        }
        // both hashes of RGB rows of 516..768 pixels: the wide-row kernel below has a dHash leg for them, this one does not
        if (g.channels == 3 && !(both && g.w > 512)) {
            switch (g.w) {          // widths with their own instantiation: compile-time row length
                case 256: KE_MX(4, 3, false, false, 3); break;
                case 384: KE_MX(6, 4, false, false, 3); break;
                case 512: KE_MX(8, 5, false, false, 3); break;
                case 640: KE_MX(10, 6, false, false, 3); break;
                case 768: KE_MX(12, 7, false, false, 3); break;
                default:
                    switch ((g.w + 63) / 64) {   // any other multiple of 4: run-time row length, operand steps of the bucket
                        case 2: KE_MX(2, 2, false, true, 3); break;
                        case 3: KE_MX(3, 2, false, true, 3); break;
                        case 4: KE_MX(4, 3, false, true, 3); break;
                        case 5: KE_MX(5, 3, false, true, 3); break;
                        case 6: KE_MX(6, 4, false, true, 3); break;
                        case 7: KE_MX(7, 5, false, true, 3); break;
                        case 8: KE_MX(8, 5, false, true, 3); break;
                        case 9: KE_MX(9, 6, false, true, 3); break;
                        case 10: KE_MX(10, 6, false, true, 3); break;
                        case 11: KE_MX(11, 7, false, true, 3); break;
                        default: break;          // 708..764: 8 operand steps do not fit the register file beside the pixel loads
                    }
            }
            if (rc != KE_EUNSUPPORTED) return rc;
        }
        if (g.channels == 1) {                     // 1-byte pixels: the luma step is a sign flip, 4-byte loads
            switch ((g.w + 63) / 64) {
                case 2: KE_MX(2, 2, false, true, 1); break;
                case 3: KE_MX(3, 2, false, true, 1); break;
                case 4: KE_MX(4, 3, false, true, 1); break;
                case 5: KE_MX(5, 3, false, true, 1); break;
                case 6: KE_MX(6, 4, false, true, 1); break;
                case 7: KE_MX(7, 5, false, true, 1); break;
                case 8: KE_MX(8, 5, false, true, 1); break;
                case 9: KE_MX(9, 6, false, true, 1); break;
                case 10: KE_MX(10, 6, false, true, 1); break;
                case 11: KE_MX(11, 7, false, true, 1); break;
                case 12: KE_MX(12, 8, false, true, 1); break;
                default: break;
            }
            if (rc != KE_EUNSUPPORTED) return rc;
        }
        if (g.channels == 4 && g.w <= 640) {       // RGBX / RGBA rows: run-time row length per 64-pixel bucket
            switch ((g.w + 63) / 64) {
                case 2: KE_MX(2, 2, false, true, 4); break;
                case 3: KE_MX(3, 2, false, true, 4); break;
                case 4: KE_MX(4, 3, false, true, 4); break;
                case 5: KE_MX(5, 3, false, true, 4); break;
                case 6: KE_MX(6, 4, false, true, 4); break;
                case 7: KE_MX(7, 5, false, true, 4); break;
                case 8: KE_MX(8, 5, false, true, 4); break;
                case 9: KE_MX(9, 6, false, true, 4); break;
                case 10: KE_MX(10, 6, false, true, 4); break;
                default: break;
            }
            if (rc != KE_EUNSUPPORTED) return rc;
        }
    }
    // ---- wide rows (and what the kernel above left): 512-thread kernel, 16-row tiles
    if (g.w > (g.channels == 4 ? 640 : g.channels == 1 ? 768 : both ? 512 : 704)) {
        if (both && g.w <= 2048) {                 // both hashes in one pass
            if (g.w <= 1024) KE_WIDE(3, 8, true, 2, 3);
            else if (g.w <= 1536) KE_WIDE(4, 12, true, 3, 3);
            else KE_WIDE(5, 16, true, 4, 3);
            if (rc == KE_OK) { *did_d = true; return KE_OK; }
            if (rc != KE_EUNSUPPORTED) return rc;
            if (!plan) return KE_EUNSUPPORTED;     // as above
        }
        if (g.channels == 4) {
            if (g.w <= 1024) KE_WIDE(3, 8, false, 1, 4);
            else if (g.w <= 1536) KE_WIDE(4, 12, false, 1, 4);
            else KE_WIDE(5, 16, false, 1, 4);
        } else if (g.channels == 1) {
            if (g.w <= 1024) KE_WIDE(3, 8, false, 1, 1);
            else if (g.w <= 1536) KE_WIDE(4, 12, false, 1, 1);
            else KE_WIDE(5, 16, false, 1, 1);
        } else {
            if (g.w <= 1024) KE_WIDE(3, 8, false, 1, 3);
            else if (g.w <= 1536) KE_WIDE(4, 12, false, 1, 3);
            else if (g.w <= 2048) KE_WIDE(5, 16, false, 1, 3);
            else if (g.w <= 2560) KE_WIDE(6, 20, false, 1, 3);      // 2560x1440-class rows: below the strip kernel's range
            else KE_WIDE(7, 22, false, 1, 3);
        }
        if (rc != KE_EUNSUPPORTED) return rc;
    }
#undef KE_MX
#undef KE_WIDE
    // rows of 516..768 with both hashes wanted that the wide kernel could not take (LDS): pHash alone from the narrow one
    if (both && g.w > 512 && g.w <= 768) return dispatch_single_pass(ctx, g, false, d_phash, d_t32, nullptr, nullptr, plan, did_d);
    return KE_EUNSUPPORTED;
}

// The single-pass kernels per band of rows: for images whose transposed columns do not fit LDS, and for groups too
// small to fill the GPU with one workgroup per image.  Tiles come out of ke_vtile as for every banded image.
int resample_single_pass_banded(ke_ctx *ctx, const KeHashGroup &g, uint8_t *d_t32, uint8_t *d_t98, bool *did_d) {
    *did_d = false;
    if ((int64_t)g.h > (int64_t)g.w * 100 || g.h < 16) return KE_EUNSUPPORTED;
    const KeAxisCoeffs *cvt = ke_get_coeffs(ctx, g.h, 32);
    const KeAxisCoeffs *cvd = d_t98 ? ke_get_coeffs(ctx, g.h, 8) : nullptr;
    if (!cvt || (d_t98 && !cvd)) return ke_fail(ctx, KE_EHIP, "coefficient upload failed");
    KeBandPlan plan;
    // bands: at most 1024 rows (32 KB of columns in LDS), and enough of them that the launch has ~4 workgroups per CU
    const int64_t fill = (4 * (int64_t)ctx->cu_count + g.n - 1) / g.n;
    // ... and short enough that two workgroups share a CU (80 KB each) where the kernel's fixed LDS allows it
    const bool both = d_t98 != nullptr && g.channels == 3;
    const int64_t fixed = (g.w <= 768 && !(both && g.w > 512)) ? 64 * (int64_t)(g.w + 32) + (both ? 4096 : 0)
                                                               : 32 * (int64_t)(g.w + 32) + 12288 + (both ? 14336 : 0);
    int64_t max_rows = (80 * 1024 - fixed) / (both ? 41 : 32) / 32 * 32 - 32;
    if (max_rows < 128) max_rows = 1024;
    max_rows = std::min<int64_t>(max_rows, 1024);
    int64_t bands = std::max<int64_t>((g.h + max_rows - 1) / max_rows, std::min<int64_t>(fill, (g.h + 63) / 64));
    plan.band_rows = (int)(((g.h + bands - 1) / bands + 31) / 32 * 32);
    plan.bands = (g.h + plan.band_rows - 1) / plan.band_rows;
    plan.hs_hp = ((std::max(cvt->span, g.h + 4) + 7) & ~7) + 8;
    plan.hsd_hp = cvd ? ((std::max(cvd->span, g.h + 4) + 7) & ~7) + 8 : 8;
    void *hs;
    KE_TRY(ke_reserve(ctx, KE_BUF_TMP, (size_t)g.n * (32 * (size_t)plan.hs_hp + (cvd ? 9 * (size_t)plan.hsd_hp : 0)) + 8192, &hs));
    plan.hs = (uint8_t *)hs;
    plan.hsd = cvd ? plan.hs + (size_t)g.n * 32 * plan.hs_hp : nullptr;
    KE_TRY(dispatch_single_pass(ctx, g, d_t98 != nullptr, nullptr, nullptr, nullptr, nullptr, &plan, did_d));
    KE_TRY(launch_vtile(ctx, plan.hs, plan.hs_hp, g.n, 32, 32, cvt, d_t32));
    if (*did_d) KE_TRY(launch_vtile(ctx, plan.hsd, plan.hsd_hp, g.n, 9, 8, cvd, d_t98));
    return KE_OK;
}

}  // namespace

int ke_launch_hash_group(ke_ctx *ctx, const KeHashGroup &g, uint64_t *d_phash, uint64_t *d_dhash, uint8_t *d_tile32_out,
                         uint8_t *d_tile98_out, hipStream_t fused_stream) {
    KE_TRY(upload_dct_tables(ctx));
    const bool want_p = d_phash || d_tile32_out, want_d = d_dhash || d_tile98_out;
    bool p_done = false, d_done = false;
    // The single-pass kernels give one workgroup a whole image, so a small group of large images cannot fill the chip
    // that way; such a group runs the same kernels per band of rows (a single image is spread over the whole GPU).
    // Measured crossover on MI355X: about 200 images at 512x512 and 1024^2, 500-700 at 2048^2, none below ~400 KB per
    // image.  KE_FUSED_MIN_IMAGES overrides the threshold (tests set it to 1 to reach the one-workgroup-per-image form
    // with a few images).
    const int64_t image_bytes = (int64_t)g.w * g.h * g.channels;
    int64_t fused_min = image_bytes < 400 * 1024 ? 1 : image_bytes < (8 << 20) ? 192 : 512;
    if (const char *e = getenv("KE_FUSED_MIN_IMAGES")) fused_min = atoll(e);
    if (want_p && g.n >= fused_min) {
        bool did_d = false;
        // one workgroup per image: no scratch buffers, so the caller may put it on a side stream where it overlaps the
        // neighbouring size groups' kernels (what follows below uses the context's scratch and stays on the context's stream)
        hipStream_t home = ctx->stream;
        if (fused_stream) ctx->stream = fused_stream;
        const int rc = dispatch_single_pass(ctx, g, want_d, d_phash, d_tile32_out, d_dhash, d_tile98_out, nullptr, &did_d);
        ctx->stream = home;
        if (rc == KE_OK) { p_done = true; d_done = did_d; }
        else if (rc != KE_EUNSUPPORTED) return rc;
    }
    // ---- generic path, chunked so the first-pass scratch stays bounded
    if ((want_p && !p_done) || (want_d && !d_done)) {
        // the banded path needs 32*H bytes of scratch per image, the generic passes up to W*H: size chunks for
        // the former (the generic fallback re-chunks itself below)
        const size_t per_img = (size_t)40 * (std::max(g.h, 32) + 1024);
        const int64_t chunk = std::max<int64_t>(1, (int64_t)(((size_t)512 << 20) / per_img));
        for (int64_t f = 0; f < g.n; f += chunk) {
            KeHashGroup s = g;
            s.n = std::min(chunk, g.n - f);
            uint64_t sub_off = 0;
            if (g.offsets) s.offsets = g.offsets + f; else sub_off = (uint64_t)f * g.stride;
            if (!g.offsets) s.pixels = g.pixels + sub_off;
            // out_idx == NULL means "slot = position in this group": keep that true for the sub-chunk
            // by offsetting the output pointers instead.
            const int64_t slot0 = g.out_idx ? 0 : f;
            if (g.out_idx) s.out_idx = g.out_idx + f;
            void *t32 = nullptr, *t98 = nullptr;
            bool d_here = false;
            if (want_d && !d_done) {
                if (d_tile98_out) t98 = d_tile98_out + (size_t)f * 72;
                else KE_TRY(ke_reserve(ctx, KE_BUF_TILE98, (size_t)s.n * 72, &t98));
            }
            if (want_p && !p_done) {
                if (d_tile32_out) t32 = d_tile32_out + (size_t)f * 1024;
                else KE_TRY(ke_reserve(ctx, KE_BUF_TILE32, (size_t)s.n * 1024, &t32));
                int rb = resample_single_pass_banded(ctx, s, (uint8_t *)t32, (uint8_t *)t98, &d_here);
                if (rb == KE_EUNSUPPORTED) rb = resample_strips(ctx, s, 32, 32, (uint8_t *)t32);
                if (rb == KE_EUNSUPPORTED) rb = resample_banded(ctx, s, 32, 32, (uint8_t *)t32);
                if (rb == KE_EUNSUPPORTED) rb = resample_generic(ctx, s, 32, 32, (uint8_t *)t32);
                KE_TRY(rb);
            }
            if (want_d && !d_done && !d_here) {
                int rb = resample_strips(ctx, s, 9, 8, (uint8_t *)t98);
                if (rb == KE_EUNSUPPORTED) rb = resample_banded(ctx, s, 9, 8, (uint8_t *)t98);
                if (rb == KE_EUNSUPPORTED) rb = resample_generic(ctx, s, 9, 8, (uint8_t *)t98);
                KE_TRY(rb);
            }
            uint64_t *ph = (d_phash && !p_done) ? d_phash + slot0 : nullptr;
            uint64_t *dh = (d_dhash && !d_done) ? d_dhash + slot0 : nullptr;
            if (ph || dh) {
                hipLaunchKernelGGL(ke_tiles_to_hashes, dim3((unsigned)s.n), dim3(256), 0, ctx->stream, (const uint8_t *)t32,
                                   (const uint8_t *)t98, s.out_idx, ph, dh, (ph && ctx->margin_cur) ? ctx->margin_cur + slot0 : nullptr);
                KE_HIP(ctx, hipGetLastError());
            }
        }
    }
    return KE_OK;
}

// Luma + resize of a group to (oh x ow) tiles: what _to_grayscale does for the hashes (LANCZOS) and what
// the shipped refine stage does for its 32x32 / 128x128 thumbnails (BILINEAR).
int ke_launch_resize_group(ke_ctx *ctx, const KeHashGroup &g, int ow, int oh, int filter, uint8_t *d_tiles, const float *box) {
    if (ow <= 0 || oh <= 0 || ow > 4096 || oh > 4096) return ke_fail(ctx, KE_EINVAL, "bad output size %dx%d", ow, oh);
    const size_t per_img = (size_t)(ow + 8) * (std::max(g.h, oh) + 1024);
    const int64_t chunk = std::max<int64_t>(1, (int64_t)(((size_t)512 << 20) / per_img));
    for (int64_t f = 0; f < g.n; f += chunk) {
        KeHashGroup s = g;
        s.n = std::min(chunk, g.n - f);
        if (g.offsets) s.offsets = g.offsets + f; else s.pixels = g.pixels + (size_t)f * g.stride;
        uint8_t *dst = d_tiles + (size_t)f * ow * oh;
        int rc = resample_banded(ctx, s, ow, oh, dst, filter, box);
        if (rc == KE_EUNSUPPORTED) rc = resample_generic(ctx, s, ow, oh, dst, filter, box);
        KE_TRY(rc);
    }
    return KE_OK;
}

namespace {

// Tile aHash (src/ui/dup_refine_parallel.py:59-83): side = grid*tile luma thumbnail, one bit per pixel =
// pixel > mean of its tile, bits in (gy, gx, ty, tx) order, packed little-endian.  `a > sum/T^2` is decided
// in integers (a*T^2 > sum): exact for every tile size.
__global__ __launch_bounds__(256) void ke_tile_ahash_kernel(const uint8_t *__restrict__ tiles, int grid, int tile,
                                                            uint64_t *__restrict__ bits, int words) {
    extern __shared__ uint32_t s_sum[];   // grid*grid tile sums
    const int side = grid * tile, t2 = tile * tile;
    const uint8_t *img = tiles + (size_t)blockIdx.x * side * side;
    for (int g = threadIdx.x; g < grid * grid; g += 256) {
        const int gy = g / grid, gx = g % grid;
        uint32_t sum = 0;
        for (int ty = 0; ty < tile; ++ty)
            for (int tx = 0; tx < tile; ++tx) sum += img[(gy * tile + ty) * side + gx * tile + tx];
        s_sum[g] = sum;
    }
    __syncthreads();
    const int nbits = side * side;
    for (int wd = threadIdx.x; wd < words; wd += 256) {
        uint64_t v = 0;
        for (int b = 0; b < 64; ++b) {
            const int i = wd * 64 + b;
            if (i >= nbits) break;
            const int g = i / t2, r = i % t2;
            const int gy = g / grid, gx = g % grid, ty = r / tile, tx = r % tile;
            const uint32_t px = img[(gy * tile + ty) * side + gx * tile + tx];
            if (px * (uint32_t)t2 > s_sum[g]) v |= 1ull << b;
        }
        bits[(size_t)blockIdx.x * words + wd] = v;
    }
}

// Sum of absolute differences of two equally sized u8 thumbnails (the integer numerator of
// _mae01, src/ui/dup_refine_parallel.py:208-210).
__global__ __launch_bounds__(256) void ke_sad_pairs_kernel(const uint8_t *__restrict__ thumbs, int64_t pixels,
                                                           const int64_t *__restrict__ pa, const int64_t *__restrict__ pb,
                                                           uint64_t *__restrict__ out) {
    __shared__ unsigned long long s_red[256];
    const uint8_t *A = thumbs + (size_t)pa[blockIdx.x] * pixels, *B = thumbs + (size_t)pb[blockIdx.x] * pixels;
    unsigned long long local = 0;
    for (int64_t e = threadIdx.x; e < pixels; e += 256) {
        const int d = (int)A[e] - (int)B[e];
        local += (unsigned)(d < 0 ? -d : d);
    }
    s_red[threadIdx.x] = local;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) s_red[threadIdx.x] += s_red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = s_red[0];
}

}  // namespace

int ke_launch_tile_ahash(ke_ctx *ctx, const uint8_t *d_tiles, int64_t n, int grid, int tile, uint64_t *d_bits) {
    const int side = grid * tile, words = (side * side + 63) / 64;
    hipLaunchKernelGGL(ke_tile_ahash_kernel, dim3((unsigned)n), dim3(256), (size_t)grid * grid * sizeof(uint32_t), ctx->stream,
                       d_tiles, grid, tile, d_bits, words);
    KE_HIP(ctx, hipGetLastError());
    return KE_OK;
}

int ke_launch_sad_pairs(ke_ctx *ctx, const uint8_t *d_thumbs, int64_t pixels, const int64_t *d_pa, const int64_t *d_pb,
                        int64_t n_pairs, uint64_t *d_out) {
    hipLaunchKernelGGL(ke_sad_pairs_kernel, dim3((unsigned)n_pairs), dim3(256), 0, ctx->stream, d_thumbs, pixels, d_pa, d_pb, d_out);
    KE_HIP(ctx, hipGetLastError());
    return KE_OK;
}
