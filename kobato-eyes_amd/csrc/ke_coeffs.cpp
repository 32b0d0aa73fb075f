// ke_coeffs.cpp -- host-side LANCZOS(a=3) tap tables for the resampler kernels.
//
// Replaces the coefficient set-up that Pillow performs inside
// image.convert("L").resize(size, LANCZOS) (reference call site: src/sig/phash.py:24-25):
// per output sample a window [first, first+count) of the input axis, weights
// sinc(t)*sinc(t/3) evaluated at tap centres, normalised to sum 1 in double precision and
// quantised to 22-bit fixed point with round-half-away-from-zero.  An unchanged axis gets
// identity taps (one tap of weight 2^22), which reproduces Pillow's "skip this pass" exactly.
#include <algorithm>
#include <cmath>

#include "ke_internal.h"

namespace {

constexpr int kPrecisionBits = 22;
constexpr double kPi = 3.14159265358979323846;

inline double sinc_pi(double x) {
    if (x == 0.0) return 1.0;
    const double a = x * kPi;
    return std::sin(a) / a;
}

inline double lanczos3(double x) {
    return (x >= -3.0 && x < 3.0) ? sinc_pi(x) * sinc_pi(x / 3.0) : 0.0;
}

// Pillow's BILINEAR ("triangle") kernel, support 1: the resize of the shipped refine stage
// (src/ui/dup_refine_parallel.py:70 and :204)
inline double triangle(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}

// Pillow's BICUBIC (Keys cubic convolution, a = -0.5, support 2): the resize inside ImageOps.fit of the SSIM
// refine step (src/dup/refine.py:48-49)
inline double bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

}  // namespace

// [in0, in1): the source interval of the axis that maps onto the out_size outputs (Pillow's `box`, in the single
// precision its C entry point receives); the whole axis is (0, in_size).
void ke_build_axis_coeffs(int in_size, int out_size, KeAxisCoeffs &c, int filter, float in0, float in1) {
    c.in_size = in_size;
    c.out_size = out_size;
    c.bounds.assign(2 * (size_t)out_size, 0);
    if (in_size == out_size && in0 == 0.0f && in1 == (float)in_size) {
        c.ksize = 1;
        c.kk.assign((size_t)out_size, 1 << kPrecisionBits);
        for (int o = 0; o < out_size; ++o) {
            c.bounds[2 * o] = o;
            c.bounds[2 * o + 1] = 1;
        }
    } else {
        const double scale = (double)(in1 - in0) / (double)out_size;
        const double fscale = scale < 1.0 ? 1.0 : scale;
        const double support = (filter == KE_FILTER_BILINEAR ? 1.0 : filter == KE_FILTER_BICUBIC ? 2.0 : 3.0) * fscale;
        const double inv = 1.0 / fscale;
        c.ksize = (int)std::ceil(support) * 2 + 1;
        c.kk.assign((size_t)out_size * c.ksize, 0);
        std::vector<double> w((size_t)c.ksize);
        for (int o = 0; o < out_size; ++o) {
            const double center = in0 + (o + 0.5) * scale;
            int lo = (int)(center - support + 0.5);
            if (lo < 0) lo = 0;
            int hi = (int)(center + support + 0.5);
            if (hi > in_size) hi = in_size;
            const int cnt = hi - lo;
            double total = 0.0;
            for (int t = 0; t < cnt; ++t) {
                const double arg = (t + lo - center + 0.5) * inv;
                w[t] = filter == KE_FILTER_BILINEAR ? triangle(arg) : filter == KE_FILTER_BICUBIC ? bicubic(arg) : lanczos3(arg);
                total += w[t];
            }
            int32_t *k = &c.kk[(size_t)o * c.ksize];
            for (int t = 0; t < cnt; ++t) {
                const double v = total != 0.0 ? w[t] / total : w[t];
                k[t] = v < 0 ? (int32_t)(v * (double)(1 << kPrecisionBits) - 0.5)
                             : (int32_t)(v * (double)(1 << kPrecisionBits) + 0.5);
            }
            c.bounds[2 * o] = lo;
            c.bounds[2 * o + 1] = cnt;
        }
    }

    // ---- packed byte-plane layout for the fused kernel -----------------------------------
    int ndw = 0;
    c.start.assign((size_t)out_size, 0);
    for (int o = 0; o < out_size; ++o) {
        const int lo = c.bounds[2 * o], cnt = c.bounds[2 * o + 1];
        const int st = lo & ~7;
        c.start[o] = st;
        const int need = (lo + cnt - st + 3) / 4;
        if (need > ndw) ndw = need;
    }
    ndw = (ndw + 3) & ~3;
    c.ndw = ndw;
    c.span = 0;
    c.bias.assign((size_t)out_size, 0);
    c.packed.assign((size_t)out_size * ndw * 3, 0);
    for (int o = 0; o < out_size; ++o) {
        const int lo = c.bounds[2 * o], cnt = c.bounds[2 * o + 1], st = c.start[o];
        if (st + 4 * ndw > c.span) c.span = st + 4 * ndw;
        int64_t ksum = 0;
        for (int j = 0; j < ndw; ++j) {
            uint32_t plane[3] = {0, 0, 0};
            for (int e = 0; e < 4; ++e) {
                const int t = st + 4 * j + e - lo;
                const int32_t k = (t >= 0 && t < cnt) ? c.kk[(size_t)o * c.ksize + t] : 0;
                ksum += k;
                const int32_t b0 = ((k + 128) & 255) - 128;
                const int32_t r1 = (k - b0) >> 8;
                const int32_t b1 = ((r1 + 128) & 255) - 128;
                const int32_t b2 = (r1 - b1) >> 8;  // |k| <= 2^22 keeps this inside int8
                plane[0] |= (uint32_t)(b0 & 255) << (8 * e);
                plane[1] |= (uint32_t)(b1 & 255) << (8 * e);
                plane[2] |= (uint32_t)(b2 & 255) << (8 * e);
            }
            for (int p = 0; p < 3; ++p) c.packed[((size_t)o * ndw + j) * 3 + p] = (int32_t)plane[p];
        }
        c.bias[o] = (int32_t)(128 * ksum + (1 << (kPrecisionBits - 1)));
    }
}

// Byte planes of the taps as matrix-core B operands (see KeMxTable).
void ke_build_mx(const KeAxisCoeffs &c, KeMxTable &t, int min_ks, bool align64) {
    t.tiles = (c.out_size + 15) / 16;
    t.base.assign((size_t)t.tiles, 0);
    t.ks = std::max(1, min_ks);
    for (int j = 0; j < t.tiles; ++j) {
        int lo = c.in_size, hi = 0;
        for (int o = 16 * j; o < std::min(16 * j + 16, c.out_size); ++o) {
            lo = std::min(lo, c.bounds[2 * o]);
            hi = std::max(hi, c.bounds[2 * o] + c.bounds[2 * o + 1]);
        }
        // 64-aligned when that costs no extra step (the kernel's dHash leg shares A operands between axes then)
        const int ks16 = (hi - (lo & ~15) + 63) / 64, ks64 = (hi - (lo & ~63) + 63) / 64;
        t.base[j] = (align64 || ks64 == ks16) ? (lo & ~63) : (lo & ~15);
        t.ks = std::max(t.ks, align64 ? ks64 : ks16);
    }
    t.frag.assign((size_t)t.tiles * t.ks * 3 * 64 * 4, 0);
    for (int j = 0; j < t.tiles; ++j)
        for (int s = 0; s < t.ks; ++s)
            for (int l = 0; l < 64; ++l) {
                const int o = 16 * j + (l & 15);
                if (o >= c.out_size) continue;
                const int lo = c.bounds[2 * o], cnt = c.bounds[2 * o + 1];
                for (int e = 0; e < 16; ++e) {
                    const int tap = t.base[j] + 64 * s + 16 * (l >> 4) + e - lo;
                    const int32_t k = (tap >= 0 && tap < cnt) ? c.kk[(size_t)o * c.ksize + tap] : 0;
                    const int32_t b0 = ((k + 128) & 255) - 128;
                    const int32_t r1 = (k - b0) >> 8;
                    const int32_t b1 = ((r1 + 128) & 255) - 128;
                    const int32_t b2 = (r1 - b1) >> 8;
                    const int32_t plane[3] = {b0, b1, b2};
                    for (int p = 0; p < 3; ++p)
                        t.frag[((((size_t)j * t.ks + s) * 3 + p) * 64 + l) * 4 + e / 4] |=
                            (int32_t)((uint32_t)(plane[p] & 255) << (8 * (e % 4)));
                }
            }
}

// Cut every output's packed window into `cpo` equal chunks (see KeChunkTable).
void ke_build_chunked(const KeAxisCoeffs &c, int cpo, KeChunkTable &t, int ndwc_multiple) {
    t.cpo = cpo;
    t.ndwc = (((c.ndw + cpo - 1) / cpo) + ndwc_multiple - 1) / ndwc_multiple * ndwc_multiple;
    const int nv = c.out_size * cpo;
    t.cstart.assign((size_t)nv, 0);
    t.cpacked.assign((size_t)nv * t.ndwc * 3, 0);
    t.cspan = 0;
    for (int o = 0; o < c.out_size; ++o)
        for (int k = 0; k < cpo; ++k) {
            const int v = o * cpo + k;
            t.cstart[v] = c.start[o] + 4 * t.ndwc * k;
            if (t.cstart[v] + 4 * t.ndwc > t.cspan) t.cspan = t.cstart[v] + 4 * t.ndwc;
            for (int j = 0; j < t.ndwc; ++j) {
                const int src = k * t.ndwc + j;
                if (src >= c.ndw) continue;
                for (int p = 0; p < 3; ++p)
                    t.cpacked[((size_t)v * t.ndwc + j) * 3 + p] = c.packed[((size_t)o * c.ndw + src) * 3 + p];
            }
        }
    // Bank-conflict swizzle: lanes are virtual columns; 32 consecutive ones form a ds_read_b64 lane group and
    // step p reads 8 bytes at cstart + 8*(p ^ x).  Greedy choice of x per lane against the lanes already placed.
    t.cxor.assign((size_t)nv, 0);
    if (t.ndwc % 8 == 0) {
        const int np = t.ndwc / 2;
        for (int g0 = 0; g0 < nv; g0 += 32) {
            for (int v = g0; v < std::min(nv, g0 + 32); ++v) {
                int best_x = 0, best_hits = 1 << 30;
                for (int x = 0; x < 4; ++x) {
                    int hits = 0;
                    for (int p = 0; p < np; ++p) {
                        const int slot = ((t.cstart[v] / 8) + (p ^ x)) & 31;
                        for (int u = g0; u < v; ++u)
                            if ((((t.cstart[u] / 8) + (p ^ t.cxor[u])) & 31) == slot) ++hits;
                    }
                    if (hits < best_hits) { best_hits = hits; best_x = x; }
                }
                t.cxor[v] = best_x;
            }
        }
    }
}
