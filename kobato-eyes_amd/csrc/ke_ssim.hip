// ke_ssim.hip -- K5: windowed SSIM of candidate pairs for gfx950.
//
// Replaces dup.refine._compute_ssim (src/dup/refine.py:44-52) for equally sized images:
// luma exactly as convert("L"), x = L/255 in float32, then
// skimage.metrics.structural_similarity(x, y, data_range=1.0): 7x7 uniform window
// (scipy.ndimage.uniform_filter: axis 0 then axis 1, each pass summed in fp64 and stored as
// float32), K1=.01, K2=.03, sample covariance (49/48), mean over the image cropped by 3 px.
// Every float32 elementwise step is kept a separate rounding (the library is compiled with
// -ffp-contract=off) so the map matches NumPy's; window sums of seven float32 values are
// exact in fp64, so their order does not matter.
//
// Layout: see ke_ssim_waves below.  A second tiny kernel adds the partial sums of a pair in a fixed order
// (bitwise reproducible) and divides.
#include <type_traits>

#include "ke_internal.h"

namespace {

// ---------------------------------------------------------------------------------------
// v2 layout: one WAVE per (pair, 250-column block, band of rows), no LDS traffic on the data path and no
// barriers.  Lane l owns halo columns x0 + 4l .. x0 + 4l + 3 and slides down the rows: seven rows of both
// images live in registers, the five vertical window sums are updated incrementally (+ new row, - oldest
// row; sums of seven float32 values are exact in fp64, so the order cannot matter), rounded to float32
// (scipy's axis-0 pass), then the horizontal windows are gathered from the two neighbouring lanes with
// wave shuffles (axis-1 pass), again summed exactly in fp64.  division by 7: see mean7.  x = L/255 is formed arithmetically (exact).
// ---------------------------------------------------------------------------------------
constexpr int kBandRows = 64;     // interior rows per wave

struct SsimArgs {
    const uint8_t *images;
    const int64_t *pa, *pb;
    int w, h;
    int col_blocks, block_cols, bands, items_per_pair;
    int band_rows;     // interior rows per wave (fast kernel; the exact kernel walks kBandRows)
    int64_t n_items;
    double *partial;   // [pair][items_per_pair]; the fast kernel stores 2^26-scaled integer sums in the same slots
};

// float32(s / 7) for a window sum s, as scipy stores it (double quotient, then the cast): one multiply by RN(1/7).
// Why that is the same value for every s that occurs here: s is an exact multiple of 2^-42 below 8 (sums of seven
// float32 values that are 0 or at least 2^-19 -- x >= 1/255, products >= 1/65025, column means of those >= 1/7 of
// that), so the true quotient K/7 * 2^-42 is either exactly a float32 rounding midpoint or at least 2^-42/7 away
// from every one of them.  RN(s * r) is within one double ulp (<= 2^-50) of the true quotient, hence on the same
// side of every midpoint as the correctly rounded double quotient; and when the quotient IS a midpoint (K = 7M),
// s * r = Q (1 - 2^-54.2) rounds back to Q itself (Q has 25 significant bits), so the tie breaks identically.
__device__ __forceinline__ float mean7(double s) {
    const double r = 0x1.2492492492492p-3;           // RN(1/7) = (1/7)(1 - 2^-54.2)
    return (float)(s * r);
}

// Luma of four RGB(X) pixels held in p[0..3] (one pixel per dword, channel bytes 0..2): byte 2 of each 16.16 sum, packed.
__device__ __forceinline__ uint32_t luma4_from_pixels(const uint32_t *p, bool last_shifted) {
    constexpr uint32_t CHI = 0x001D964Cu, CLO = 0x002F468Bu;     // 19595 = 76*256+139, 38470 = 150*256+70, 7471 = 29*256+47
    uint32_t s[4];
#pragma unroll
    for (int k = 0; k < 3; ++k) s[k] = (__builtin_amdgcn_udot4(p[k], CHI, 0u, false) << 8) + __builtin_amdgcn_udot4(p[k], CLO, 0x8000u, false);
    // packed RGB: the fourth pixel sits in bytes 1..3 of its dword -- shift the weights, not the data
    s[3] = last_shifted ? (__builtin_amdgcn_udot4(p[3], CHI << 8, 0u, false) << 8) + __builtin_amdgcn_udot4(p[3], CLO << 8, 0x8000u, false)
                        : (__builtin_amdgcn_udot4(p[3], CHI, 0u, false) << 8) + __builtin_amdgcn_udot4(p[3], CLO, 0x8000u, false);
    return __builtin_amdgcn_perm(s[1], s[0], 0x0C0C0602u) | __builtin_amdgcn_perm(s[3], s[2], 0x06020C0Cu);
}

// Four consecutive pixels of one row starting `off` bytes into the image (any alignment) -> four luma bytes.
// base: the image pointer rounded down to a dword (wave-uniform), off: byte offset from it, last: offset of the last
// whole dword of the image -- all 32-bit, so a load is scalar base + clamped vector offset.
template <int C>
__device__ __forceinline__ uint32_t load_luma4(const uint8_t *base, uint32_t off, uint32_t last) {
    const uint32_t al = off & ~3u;
    const int sb = (int)(off & 3u);
    uint32_t wv[C + 1], d[C];
#pragma unroll
    for (int k = 0; k <= C; ++k) wv[k] = *reinterpret_cast<const uint32_t *>(base + min(al + 4u * k, last));
#pragma unroll
    for (int k = 0; k < C; ++k) d[k] = __builtin_amdgcn_alignbyte(wv[k + 1], wv[k], sb);
    if (C == 1) return d[0];
    uint32_t p[4];
    if (C == 3) {
        p[0] = d[0];
        p[1] = __builtin_amdgcn_alignbyte(d[1], d[0], 3);
        p[2] = __builtin_amdgcn_alignbyte(d[2 % C], d[1], 2);
        p[3] = d[2 % C];
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = d[k % C];
    }
    return luma4_from_pixels(p, C == 3);
}

// The same four luma bytes when the quad starts on a dword boundary (rows of a multiple of 4 pixels, lane columns a
// multiple of 4): C dword loads, no funnel shifts, bytes packed with two v_perm.
template <int C>
__device__ __forceinline__ uint32_t luma4_of_aligned(const uint32_t *q) {     // q: the C dwords of an aligned quad
    if (C == 1) return q[0];
    uint32_t p[4];
    if (C == 3) {
        p[0] = q[0];
        p[1] = __builtin_amdgcn_alignbyte(q[1], q[0], 3);
        p[2] = __builtin_amdgcn_alignbyte(q[2 % C], q[1], 2);
        p[3] = q[2 % C];
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = q[k % C];
    }
    return luma4_from_pixels(p, C == 3);
}

// x = L/255 in float32, correctly rounded for every L in 0..255 (checked for all 256 values in exact rational
// arithmetic): 1/255 = r_hi + r_lo to 48 bits, L * r_lo rounded, then one fma -- L/255 is never close to a float32
// rounding boundary (255 is odd), so the 2^-24 ulp this loses cannot change the result.
__device__ __forceinline__ float unit_of(uint32_t L) {
    const float r_hi = 0x1.010102p-8f, r_lo = -0x1.fdfdfep-33f;
    const float l = (float)L;
    return fmaf(l, r_hi, l * r_lo);
}

// PX pixels per lane (4 or 8): a wave covers 64*PX halo columns, 64*PX - 6 interior ones.
// AL: rows and lane columns start on dword boundaries (width a multiple of 4, aligned base): aligned loader, and the
// wave's interior width is rounded down to a multiple of 4 so that every block does.
template <int C, int PX, bool AL>
__global__ __launch_bounds__(256, 2) void ke_ssim_waves(const SsimArgs a) {
    constexpr int NW = PX / 4;                              // luma dwords per row per image per lane
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= a.n_items) return;
    const int64_t pair = item / a.items_per_pair;
    const int sub = (int)(item % a.items_per_pair);
    const int band = sub / a.col_blocks, cb = sub % a.col_blocks;
    const int block_cols = a.block_cols;                    // interior columns per wave: 64*PX - 6, or that rounded down to 4
    const int x0 = cb * block_cols;
    const int y0 = band * kBandRows;                       // first halo row; interior rows y0+3 ..
    const int y_int_end = min(a.h - 3, y0 + 3 + kBandRows);  // one past the last interior row of this band
    const size_t img_bytes = (size_t)a.w * a.h * C;
    const uint8_t *A = a.images + (size_t)a.pa[pair] * img_bytes;
    const uint8_t *B = a.images + (size_t)a.pb[pair] * img_bytes;
    // unaligned loader: dword-aligned bases (wave-uniform) + 32-bit byte offsets
    const uint32_t dA = (uint32_t)((uintptr_t)A & 3), dB = (uint32_t)((uintptr_t)B & 3);
    const uint8_t *A0 = A - dA, *B0 = B - dB;
    const uint32_t lastA = (dA + (uint32_t)img_bytes - 1u) & ~3u, lastB = (dB + (uint32_t)img_bytes - 1u) & ~3u;
    int xc = x0 + PX * lane;                                // this lane's first column
    xc = xc < a.w ? xc : a.w - 1;                           // lanes past the right edge load something valid; never used
    int xq[NW];                                             // AL: first column of each 4-pixel group, kept inside the row
#pragma unroll
    for (int n = 0; n < NW; ++n) xq[n] = min(x0 + PX * lane + 4 * n, a.w - 4);

    uint32_t ra[7][NW], rb[7][NW];                          // luma bytes of the seven most recent rows
    double sx[PX], sy[PX], sxx[PX], syy[PX], sxy[PX];
#pragma unroll
    for (int k = 0; k < PX; ++k) sx[k] = sy[k] = sxx[k] = syy[k] = sxy[k] = 0.0;
#pragma unroll
    for (int d = 0; d < 7; ++d)
#pragma unroll
        for (int n = 0; n < NW; ++n) ra[d][n] = rb[d][n] = 0;

    const float cov_norm = (float)(49.0 / 48.0);
    const float C1 = (float)(0.01 * 0.01), C2 = (float)(0.03 * 0.03);
    double local = 0.0;
    const int y_last = y_int_end + 3;                       // one past the last halo row
    // AL with 8 pixels per lane (2 waves/SIMD): the next row's quads are fetched one iteration ahead (raw dwords), so their
    // latency hides behind the arithmetic (+5 % at 512x512; with 4 pixels per lane the extra registers cost more than it gains)
    constexpr bool PF = AL && PX == 8;
    uint32_t qa[NW][C], qb[NW][C];
    auto fetch = [&](int y) {
        const int yc = min(y, y_last - 1);
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            const uint32_t o = ((uint32_t)yc * (uint32_t)a.w + (uint32_t)xq[n]) * C;
#pragma unroll
            for (int k = 0; k < C; ++k) {
                qa[n][k] = reinterpret_cast<const uint32_t *>(A + o)[k];
                qb[n][k] = reinterpret_cast<const uint32_t *>(B + o)[k];
            }
        }
    };
    if (PF) fetch(y0);
    for (int y = y0; y < y_last; ++y) {
        const uint32_t off = ((uint32_t)y * (uint32_t)a.w + (uint32_t)xc) * C;
        uint32_t na[NW], nb[NW];
        if (PF) {
#pragma unroll
            for (int n = 0; n < NW; ++n) { na[n] = luma4_of_aligned<C>(qa[n]); nb[n] = luma4_of_aligned<C>(qb[n]); }
            fetch(y + 1);
        }
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            if (PF) {
            } else if (AL) {
                const uint32_t o = ((uint32_t)y * (uint32_t)a.w + (uint32_t)xq[n]) * C;
                na[n] = luma4_of_aligned<C>(reinterpret_cast<const uint32_t *>(A + o));
                nb[n] = luma4_of_aligned<C>(reinterpret_cast<const uint32_t *>(B + o));
            } else {
                na[n] = load_luma4<C>(A0, dA + off + (uint32_t)n * 4 * C, lastA);
                nb[n] = load_luma4<C>(B0, dB + off + (uint32_t)n * 4 * C, lastB);
            }
        }
        // slide: drop the oldest row, add the new one.  The first six rows of a band subtract the zeros the
        // ring starts with (L = 0 -> x = 0), so no special case.
#pragma unroll
        for (int k = 0; k < PX; ++k) {
            const int n = k >> 2, sh = 8 * (k & 3);
            const float ox = unit_of((ra[0][n] >> sh) & 0xFF), oy = unit_of((rb[0][n] >> sh) & 0xFF);
            const float nx = unit_of((na[n] >> sh) & 0xFF), ny = unit_of((nb[n] >> sh) & 0xFF);
            sx[k] += (double)nx - (double)ox;
            sy[k] += (double)ny - (double)oy;
            sxx[k] += (double)(nx * nx) - (double)(ox * ox);
            syy[k] += (double)(ny * ny) - (double)(oy * oy);
            sxy[k] += (double)(nx * ny) - (double)(ox * oy);
        }
#pragma unroll
        for (int n = 0; n < NW; ++n) {
#pragma unroll
            for (int d = 0; d < 6; ++d) { ra[d][n] = ra[d + 1][n]; rb[d][n] = rb[d + 1][n]; }
            ra[6][n] = na[n]; rb[6][n] = nb[n];
        }
        if (y < y0 + 6) continue;                           // window not full yet (wave-uniform)
        // axis-0 means of this lane's columns, rounded to float32 as scipy stores them
        float m[5][PX + 6];                                 // [quantity][left 3 | own PX | right 3]
#pragma unroll
        for (int k = 0; k < PX; ++k) {
            m[0][3 + k] = mean7(sx[k]);
            m[1][3 + k] = mean7(sy[k]);
            m[2][3 + k] = mean7(sxx[k]);
            m[3][3 + k] = mean7(syy[k]);
            m[4][3 + k] = mean7(sxy[k]);
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) {
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                m[q][e] = __shfl_up(m[q][PX + e], 1);            // left lane's last three columns
                m[q][PX + 3 + e] = __shfl_down(m[q][3 + e], 1);  // right lane's first three columns
            }
        }
        // axis-1 means + SSIM; the window of own column k covers m[.][k .. k+6]
        double hs[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            hs[q] = 0.0;
#pragma unroll
            for (int d = 0; d < 7; ++d) hs[q] += (double)m[q][d];
        }
#pragma unroll
        for (int k = 0; k < PX; ++k) {
            if (k > 0) {
#pragma unroll
                for (int q = 0; q < 5; ++q) hs[q] += (double)m[q][k + 6] - (double)m[q][k - 1];
            }
            const int bc = PX * lane + k;                  // column inside the block
            const bool inside = (bc >= 3) && (bc < 3 + block_cols) && (x0 + bc < a.w - 3);
            const float ux = mean7(hs[0]), uy = mean7(hs[1]);
            const float uxx = mean7(hs[2]), uyy = mean7(hs[3]), uxy = mean7(hs[4]);
            const float mxx = ux * ux, myy = uy * uy, mxy = ux * uy;
            const float vx = cov_norm * (uxx - mxx);
            const float vy = cov_norm * (uyy - myy);
            const float vxy = cov_norm * (uxy - mxy);
            const float A1 = 2.0f * mxy + C1, A2 = 2.0f * vxy + C2;
            const float B1 = (mxx + myy) + C1, B2 = (vx + vy) + C2;
            const float S = (A1 * A2) / (B1 * B2);
            if (inside) local += (double)S;
        }
    }
    // wave reduction in a fixed order
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) local += __shfl_down(local, s);
    if (lane == 0) a.partial[item] = local;
}

// ---------------------------------------------------------------------------------------
// Fast variant (the default): the same wave layout, but every window sum is an exact INTEGER carried in float32.
//
// With integer luma L in 0..255 the five window sums of skimage's SSIM are integers below 2^24, hence exact in
// float32 in any order -- so the fp64 carries of the exact kernel (150 VALU instructions per pixel pair, VALU-issue
// bound at 0.2 of HBM) are unnecessary.  Further, SSIM needs vx + vy, vxy, ux*uy and ux^2 + uy^2 only, which follow
// from FOUR box filters over p = La + Lb and m = La - Lb instead of five over (x, y, xx, yy, xy):
//     SP = sum p, SM = sum m, SPP = sum p^2 (<= 49 * 510^2 < 2^24), SMM = sum m^2          (7 x 7 window)
//     2 ux uy      = (SP^2 - SM^2) c1        ux^2 + uy^2 = (SP^2 + SM^2) c1        c1 = 1 / (2 (49*255)^2)
//     2 vxy        = (Np - Nm) c2            vx + vy     = (Np + Nm) c2            c2 = 1 / (2*48*49*255^2)
//     Np = 49 SPP - SP^2,  Nm = 49 SMM - SM^2   (49/48 = skimage's sample-covariance normalisation)
// Np / Nm are formed without cancellation error: P = RN(S^2), e2 = fma(S, S, -P) (exact), e1 = fma(49, Q, -P) (exact
// whenever |N| < 2^24, i.e. exactly where cancellation would hurt), N = e1 - e2.  What differs from the exact kernel
// is therefore only what skimage's own float32 intermediates round away: |delta SSIM| stays below 1e-5 (measured over
// the golden pairs and the 100 000-image corpus' candidate edges; tests/test_gpu_parity.py), the bar being 1e-4.
//
// All (p, m) quantities travel as float2 so the sums are v_pk_add_f32 / v_pk_fma_f32 (two per instruction); the ring of
// the seven most recent rows is unrolled in place (no register moves); the 3-column halos come from the neighbouring
// lanes through ds_bpermute (LDS crossbar, no VALU issue slot); four quotients share one v_rcp_f32.
// ---------------------------------------------------------------------------------------
typedef float ke_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ ke_f2 pk_fma(ke_f2 a, ke_f2 b, ke_f2 c) { return __builtin_elementwise_fma(a, b, c); }

// 16.16 luma sum of one pixel held in a dword (channel bytes 0..2, or 1..3 when `shifted`); byte 2 of the result is L
__device__ __forceinline__ uint32_t luma_sum(uint32_t p, bool shifted) {
    constexpr uint32_t CHI = 0x001D964Cu, CLO = 0x002F468Bu;
    return shifted ? (__builtin_amdgcn_udot4(p, CHI << 8, 0u, false) << 8) + __builtin_amdgcn_udot4(p, CLO << 8, 0x8000u, false)
                   : (__builtin_amdgcn_udot4(p, CHI, 0u, false) << 8) + __builtin_amdgcn_udot4(p, CLO, 0x8000u, false);
}
__device__ __forceinline__ float byte2_f32(uint32_t s) { return (float)((s >> 16) & 0xFFu); }   // v_cvt_f32_ubyte2

// The C dwords of an aligned quad -> four luma values as floats (integers 0..255).  (A float32 form -- three byte->float
// conversions, an exact multiply-add chain, floor -- costs the same issue cycles as these 2 v_dot4 + v_lshl_add + conversion:
// its first fma with the 0.5 is a three-source instruction again, and it needs 10 more registers.)
template <int C>
__device__ __forceinline__ void quad_to_f32(const uint32_t *q, float *out) {
    if (C == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = (float)((q[0] >> (8 * k)) & 0xFFu);   // v_cvt_f32_ubyte0..3
    } else if (C == 3) {
        out[0] = byte2_f32(luma_sum(q[0], false));
        out[1] = byte2_f32(luma_sum(__builtin_amdgcn_alignbyte(q[1], q[0], 3), false));
        out[2] = byte2_f32(luma_sum(__builtin_amdgcn_alignbyte(q[2 % C], q[1], 2), false));
        out[3] = byte2_f32(luma_sum(q[2 % C], true));
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = byte2_f32(luma_sum(q[k % C], false));
    }
}

// keeps the compiler from folding (float)a + (float)b into an integer add + one more conversion (costs an instruction)
__device__ __forceinline__ float opaque(float x) { asm("" : "+v"(x)); return x; }

// Packed float32 operations with operand selection (VOP3P op_sel / neg): one instruction where the compiler's own choice for
// "difference and sum of the two halves of a pair" is two or three.
__device__ __forceinline__ ke_f2 pk_diff_sum(ke_f2 v) {          // {v.lo - v.hi, v.lo + v.hi}
    ke_f2 r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ ke_f2 pk_sum_diff(ke_f2 v) {          // {v.lo + v.hi, v.lo - v.hi}
    ke_f2 r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ ke_f2 pk_mul_hh_lh(ke_f2 x, ke_f2 y) {  // {x.hi * y.hi, x.lo * y.hi}
    ke_f2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ ke_f2 pk_mul_ll_hl(ke_f2 x, ke_f2 y) {  // {x.lo * y.lo, x.hi * y.lo}
    ke_f2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

template <int C, int PX, bool AL>
__global__ __launch_bounds__(256, 2) void ke_ssim_fast(const SsimArgs a) {
    constexpr int NW = PX / 4;
    const int lane = threadIdx.x & 63;
    // everything that names the work item is the same for the 64 lanes: kept in scalar registers, so that a row's loads are
    // "scalar row pointer + this lane's constant byte offset" and the row loop spends no vector instruction on addresses
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;
    if (item >= a.n_items) return;
    const int64_t pair = item / a.items_per_pair;
    const int sub = (int)(item % a.items_per_pair);
    const int band = sub / a.col_blocks, cb = sub % a.col_blocks;
    const int block_cols = a.block_cols;
    const int x0 = cb * block_cols;
    const int y0 = band * a.band_rows;                      // first halo row; interior rows y0 + 3 ..
    const int y_last = min(a.h - 3, y0 + 3 + a.band_rows) + 3;   // one past the last halo row (>= y0 + 7)
    const size_t img_bytes = (size_t)a.w * a.h * C;
    const uint32_t row_bytes = (uint32_t)a.w * C;
    const uint8_t *A = a.images + (size_t)a.pa[pair] * img_bytes;
    const uint8_t *B = a.images + (size_t)a.pb[pair] * img_bytes;
    const uint32_t dA = (uint32_t)((uintptr_t)A & 3), dB = (uint32_t)((uintptr_t)B & 3);
    const uint8_t *A0 = A - dA, *B0 = B - dB;
    const uint32_t lastA = (dA + (uint32_t)img_bytes - 1u) & ~3u, lastB = (dB + (uint32_t)img_bytes - 1u) & ~3u;
    int xc = x0 + PX * lane;
    xc = xc < a.w ? xc : a.w - 1;
    uint32_t voff[NW];                                      // AL: byte offset of each 4-pixel group inside its row
#pragma unroll
    for (int n = 0; n < NW; ++n) voff[n] = (uint32_t)min(x0 + PX * lane + 4 * n, a.w - 4) * C;

    // column k of this lane counts iff it is an interior column of the block and of the image
    bool ink[PX];                                           // loop-invariant lane masks (scalar register pairs)
#pragma unroll
    for (int k = 0; k < PX; ++k) {
        const int bc = PX * lane + k;
        ink[k] = (bc >= 3) && (bc < 3 + block_cols) && (x0 + bc < a.w - 3);
    }
    const int addr_l = ((lane + 63) & 63) * 4, addr_r = ((lane + 1) & 63) * 4;   // ds_bpermute byte addresses of the neighbours

    ke_f2 ring[7][PX];                                      // (p, m) of the seven most recent rows
    ke_f2 vs[PX], vq[PX];                                   // vertical window sums (sum p, sum m), (sum p^2, sum m^2)
#pragma unroll
    for (int k = 0; k < PX; ++k) {
        vs[k] = vq[k] = ke_f2{0.f, 0.f};
#pragma unroll
        for (int d = 0; d < 7; ++d) ring[d][k] = ke_f2{0.f, 0.f};
    }
    const float c1 = (float)(1.0 / (2.0 * (49.0 * 255.0) * (49.0 * 255.0)));
    const float c2 = (float)(1.0 / (2.0 * 48.0 * 49.0 * 255.0 * 255.0));
    const float C1 = (float)(0.01 * 0.01), C2 = (float)(0.03 * 0.03);
    // Row sums are quantised to 2^-26 (finer than the float32 they come from) and added as integers, so a pair's score does
    // not depend on how its rows are cut into bands or on the order of the partial sums: one launch or many give equal bits.
    long long local = 0;

    // The luma of the NEXT row travels from iteration to iteration as floats -- (La, Lb) of a column in one register pair:
    // its loads are issued at the top of an iteration and converted at the bottom, behind the arithmetic of the current row.
    ke_f2 xab[PX];
    uint32_t qa[NW][C], qb[NW][C];
    auto issue = [&](int y) {                               // raw quads of row y (clamped to the band)
        const uint32_t yc = (uint32_t)min(y, y_last - 1);
        const uint8_t *ra = A + (size_t)(yc * row_bytes), *rb = B + (size_t)(yc * row_bytes);
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            if (AL) {
                // the 32-bit offset is made opaque here so that its zero-extension stays next to the load: hoisted out of the
                // loop it becomes a 64-bit register pair and every load pays a 64-bit vector add instead of using the
                // "scalar base + 32-bit vector offset" addressing mode
                uint32_t vo = voff[n];
                asm volatile("" : "+v"(vo));
#pragma unroll
                for (int k = 0; k < C; ++k) {
                    qa[n][k] = *reinterpret_cast<const uint32_t *>(ra + vo + 4 * k);
                    qb[n][k] = *reinterpret_cast<const uint32_t *>(rb + vo + 4 * k);
                }
            } else {
                const uint32_t off = (yc * (uint32_t)a.w + (uint32_t)xc) * C + (uint32_t)n * 4 * C;
                qa[n][0] = load_luma4<C>(A0, dA + off, lastA);
                qb[n][0] = load_luma4<C>(B0, dB + off, lastB);
            }
        }
    };
    auto convert = [&]() {
        float fa[4], fb[4];
#pragma unroll
        for (int n = 0; n < NW; ++n) {
            if (AL) {
                quad_to_f32<C>(qa[n], fa);
                quad_to_f32<C>(qb[n], fb);
            } else {
                quad_to_f32<1>(qa[n], fa);
                quad_to_f32<1>(qb[n], fb);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) xab[4 * n + k] = ke_f2{fa[k], fb[k]};
        }
    };
    // slide the vertical windows: - the row that leaves (ring slot `slot` holds it; zeros at the band start), + the new row
    auto vertical = [&](auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
#pragma unroll
        for (int k = 0; k < PX; ++k) {
            vs[k] = vs[k] - ring[slot][k];
            vq[k] = pk_fma(-ring[slot][k], ring[slot][k], vq[k]);
            ring[slot][k] = pk_sum_diff(xab[k]);             // (La + Lb, La - Lb): one packed add
            vs[k] = vs[k] + ring[slot][k];
            vq[k] = pk_fma(ring[slot][k], ring[slot][k], vq[k]);
        }
    };
    auto outputs = [&](bool valid) {
        // the 7-wide horizontal windows of this lane's PX columns over w = [left 3 | own PX | right 3]: suffix sums of
        // w[0..6] towards the left, prefix sums of w[7..] towards the right, window k = suffix[k] + prefix[k + 6]
        // (2.25 packed adds per column and quantity pair instead of 2.5 for a sliding sum); the halo values, which arrive
        // through the LDS crossbar, sit at the far ends of both chains
        ke_f2 ls[3], lq[3], rs[3], rq[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                ls[e][c] = __int_as_float(__builtin_amdgcn_ds_bpermute(addr_l, __float_as_int(vs[PX - 3 + e][c])));
                lq[e][c] = __int_as_float(__builtin_amdgcn_ds_bpermute(addr_l, __float_as_int(vq[PX - 3 + e][c])));
                rs[e][c] = __int_as_float(__builtin_amdgcn_ds_bpermute(addr_r, __float_as_int(vs[e][c])));
                rq[e][c] = __int_as_float(__builtin_amdgcn_ds_bpermute(addr_r, __float_as_int(vq[e][c])));
            }
        }
        auto w_s = [&](int j) -> ke_f2 { return j < 3 ? ls[j] : j < PX + 3 ? vs[j - 3] : rs[j - PX - 3]; };
        auto w_q = [&](int j) -> ke_f2 { return j < 3 ? lq[j] : j < PX + 3 ? vq[j - 3] : rq[j - PX - 3]; };
        ke_f2 Ls[7], Lq[7], Rs[PX - 1], Rq[PX - 1];
        Ls[6] = w_s(6); Lq[6] = w_q(6);
#pragma unroll
        for (int j = 5; j >= 0; --j) { Ls[j] = w_s(j) + Ls[j + 1]; Lq[j] = w_q(j) + Lq[j + 1]; }
        Rs[0] = w_s(7); Rq[0] = w_q(7);
#pragma unroll
        for (int i = 1; i < PX - 1; ++i) { Rs[i] = Rs[i - 1] + w_s(7 + i); Rq[i] = Rq[i - 1] + w_q(7 + i); }
        ke_f2 nd[PX];                                       // (numerator, denominator) of the column's SSIM value
#pragma unroll
        for (int k = 0; k < PX; ++k) {
            const ke_f2 hs = k == 0 ? Ls[0] : k <= 6 ? Ls[k] + Rs[k - 1] : Rs[k - 1];
            const ke_f2 hq = k == 0 ? Lq[0] : k <= 6 ? Lq[k] + Rq[k - 1] : Rq[k - 1];
            const ke_f2 P = hs * hs;
            const ke_f2 e2 = pk_fma(hs, hs, -P);
            const ke_f2 e1 = pk_fma(ke_f2{49.f, 49.f}, hq, -P);
            const ke_f2 N = e1 - e2;                        // (Np, Nm)
            const ke_f2 d1 = pk_diff_sum(P);                // (SP^2 - SM^2, SP^2 + SM^2)
            const ke_f2 d2 = pk_diff_sum(N);                // (Np - Nm, Np + Nm)
            const ke_f2 ab1 = pk_fma(d1, ke_f2{c1, c1}, ke_f2{C1, C1});   // (A1, B1)
            const ke_f2 ab2 = pk_fma(d2, ke_f2{c2, c2}, ke_f2{C2, C2});   // (A2, B2)
            nd[k] = ab1 * ab2;
            nd[k][0] = ink[k] ? nd[k][0] : 0.f;
        }
        // four quotients per reciprocal: n0/d0 + .. + n3/d3 = (N01 D23 + N23 D01) / (D01 D23); every d >= C1*C2 = 9e-8.
        // (D01, n0 d1) and then (D01 D23, N01 D23) come out of one packed multiply each
        float rowsum = 0.f;
#pragma unroll
        for (int g = 0; g < PX; g += 4) {
            ke_f2 u = pk_mul_hh_lh(nd[g], nd[g + 1]);                  // (d0 d1, n0 d1)
            u[1] = __builtin_fmaf(nd[g + 1][0], nd[g][1], u[1]);       // (D01, N01)
            ke_f2 v = pk_mul_hh_lh(nd[g + 2], nd[g + 3]);
            v[1] = __builtin_fmaf(nd[g + 3][0], nd[g + 2][1], v[1]);   // (D23, N23)
            const ke_f2 t = pk_mul_ll_hl(u, v);                        // (D01 D23, N01 D23)
            const float Nn = __builtin_fmaf(v[1], u[0], t[1]);
            rowsum = __builtin_fmaf(Nn, __builtin_amdgcn_rcpf(t[0]), rowsum);
        }
        const int q = __float2int_rn(rowsum * 67108864.0f);   // |rowsum| <= PX + eps
        local += valid ? (long long)q : 0ll;
    };

    issue(y0);
    convert();
    // the first six halo rows only fill the windows (ring slots 0..5)
#define KE_FILL(S) do { issue(y0 + (S) + 1); vertical(std::integral_constant<int, (S)>{}); convert(); } while (0)
    KE_FILL(0); KE_FILL(1); KE_FILL(2); KE_FILL(3); KE_FILL(4); KE_FILL(5);
#undef KE_FILL
    // from the seventh row on every row completes a window; rows are taken seven at a time so that the ring slot of a
    // row is a compile-time constant (no register moves); rows past the band's end are computed on the clamped last
    // row and weighted 0
    for (int yb = y0 + 6; yb < y_last; yb += 7) {
#define KE_ROW(R) do { issue(yb + (R) + 1); vertical(std::integral_constant<int, ((R) + 6) % 7>{}); \
                       outputs(yb + (R) < y_last); convert(); } while (0)
        KE_ROW(0); KE_ROW(1); KE_ROW(2); KE_ROW(3); KE_ROW(4); KE_ROW(5); KE_ROW(6);
#undef KE_ROW
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) local += __shfl_down(local, s);
    if (lane == 0) reinterpret_cast<long long *>(a.partial)[item] = local;
}

__global__ void ke_ssim_finish_fixed(const long long *__restrict__ partial, int tiles, int64_t n_pairs, double denom,
                                     double *__restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    long long s = 0;
    for (int t = 0; t < tiles; ++t) s += partial[(size_t)p * tiles + t];
    out[p] = ((double)s * (1.0 / 67108864.0)) / denom;
}

__global__ void ke_ssim_finish(const double *__restrict__ partial, int tiles, int64_t n_pairs, double denom,
                               double *__restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    double s = 0.0;
    for (int t = 0; t < tiles; ++t) s += partial[(size_t)p * tiles + t];
    out[p] = s / denom;
}

__global__ void ke_fill_nan(double *out, int64_t n) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) out[p] = __longlong_as_double(0x7ff8000000000000LL);
}

}  // namespace

int ke_launch_ssim(ke_ctx *ctx, const uint8_t *d_images, int w, int h, int channels, const int64_t *d_pa,
                   const int64_t *d_pb, int64_t n_pairs, double *d_out) {
    if (w < 7 || h < 7) {  // skimage raises for images smaller than the window
        hipLaunchKernelGGL(ke_fill_nan, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, ctx->stream, d_out, n_pairs);
        KE_HIP(ctx, hipGetLastError());
        return KE_OK;
    }
    if ((int64_t)w * h * channels >= (1LL << 31)) return ke_fail(ctx, KE_EUNSUPPORTED, "image too large for the SSIM kernel");
    SsimArgs a;
    a.images = d_images; a.pa = d_pa; a.pb = d_pb;
    a.w = w; a.h = h;
    // pixels per lane: the choice that leaves the fewest idle lanes (ties -> 4, which holds 2 waves/SIMD more easily)
    const int cb4 = (w - 6 + 249) / 250, cb8 = (w - 6 + 505) / 506;
    const int px = (cb8 * 506 < cb4 * 250) ? 8 : 4;
    a.col_blocks = px == 8 ? cb8 : cb4;
    a.block_cols = 64 * px - 6;
    // aligned loads need dword-aligned quads: width a multiple of 4, image base and size multiples of 4, and every
    // block starting on a multiple of 4 -- one block, or blocks of 248 / 504 columns when that does not add a block
    bool al = w % 4 == 0 && (uintptr_t)d_images % 4 == 0 && ((size_t)w * h * channels) % 4 == 0;
    if (al && a.col_blocks > 1) {
        const int bca = a.block_cols & ~3;
        if ((w - 6 + bca - 1) / bca == a.col_blocks) a.block_cols = bca; else al = false;
    }
    a.band_rows = kBandRows;
    // Images of fewer than 4096 windows always take the exact kernel: skimage's float32 intermediates can be off by up to
    // ~1e-4 in a single window of a bright, nearly flat image (uxx - ux*ux cancels); over thousands of windows that
    // averages far below 1e-5, in a 7x7 image (one window) it is the whole score.  Such images cost nothing either way.
    const bool exact = ctx->ssim_exact || (int64_t)(w - 6) * (h - 6) < 4096;
    if (!exact) {
        // the fast kernel takes rows seven at a time: bands of 7*G interior rows, G as large as leaves >= ~8 waves per SIMD
        // of work in the launch (each band re-reads 6 halo rows), at most 18 (126 rows)
        const int64_t groups_total = (h - 6 + 6) / 7;
        const int64_t cols = px == 8 ? cb8 : cb4;
        int64_t G = groups_total * n_pairs * cols / 8192;
        G = std::max<int64_t>(2, std::min<int64_t>(18, G));
        const int64_t nb = (groups_total + G - 1) / G;
        G = (groups_total + nb - 1) / nb;                   // equalise the bands
        a.band_rows = (int)(7 * G);
    }
    a.bands = (h - 6 + a.band_rows - 1) / a.band_rows;
    a.items_per_pair = a.col_blocks * a.bands;
    a.n_items = n_pairs * a.items_per_pair;
    const int64_t blocks = (a.n_items + 3) / 4;
    if (blocks > 0x7fffffffLL) return ke_fail(ctx, KE_EUNSUPPORTED, "too many SSIM work items for one launch");
    void *part;
    KE_TRY(ke_reserve(ctx, KE_BUF_SCAN_AUX, (size_t)a.n_items * sizeof(double), &part));
    a.partial = (double *)part;
    const dim3 grid((unsigned)blocks), blk(256);
#define KE_SSIM_LAUNCH(CH, PXV) do { \
        if (exact) { if (al) hipLaunchKernelGGL((ke_ssim_waves<CH, PXV, true>), grid, blk, 0, ctx->stream, a); \
                               else hipLaunchKernelGGL((ke_ssim_waves<CH, PXV, false>), grid, blk, 0, ctx->stream, a); } \
        else { if (al) hipLaunchKernelGGL((ke_ssim_fast<CH, PXV, true>), grid, blk, 0, ctx->stream, a); \
               else hipLaunchKernelGGL((ke_ssim_fast<CH, PXV, false>), grid, blk, 0, ctx->stream, a); } } while (0)
    if (channels == 3) { if (px == 8) KE_SSIM_LAUNCH(3, 8); else KE_SSIM_LAUNCH(3, 4); }
    else if (channels == 1) { if (px == 8) KE_SSIM_LAUNCH(1, 8); else KE_SSIM_LAUNCH(1, 4); }
    else { if (px == 8) KE_SSIM_LAUNCH(4, 8); else KE_SSIM_LAUNCH(4, 4); }
#undef KE_SSIM_LAUNCH
    KE_HIP(ctx, hipGetLastError());
    if (exact)
        hipLaunchKernelGGL(ke_ssim_finish, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const double *)part, a.items_per_pair, n_pairs, (double)(w - 6) * (double)(h - 6), d_out);
    else
        hipLaunchKernelGGL(ke_ssim_finish_fixed, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const long long *)part, a.items_per_pair, n_pairs, (double)(w - 6) * (double)(h - 6), d_out);
    KE_HIP(ctx, hipGetLastError());
    return KE_OK;
}
