// ke_scan.hip -- K4: all-pairs Hamming candidate scan for gfx950.
//
// Replaces the bucket build + in-bucket pair loop of DuplicateScanner.build_clusters
// (src/dup/scanner.py:227-299).  The pair space i<j is cut into 1024 x 1024 tiles of the upper
// triangle; a workgroup owns one tile.
//
// The Hamming distance itself runs on the matrix cores: with every hash written out as 128 one-bit
// values [bits of x | bits of ~x], <row(x), row(y)> = popc(x & y) + popc(~x & ~y) = 64 - popc(x ^ y), so one
// v_mfma_scale_f32_16x16x128_f8f6f4 with fp4 (E2M1) operands -- bit 1 = 1.0, exact, sums <= 64 exact in
// f32 -- yields the 256 similarities of 16 x 16 hashes in 16 cycles.  A pre-pass expands the hashes once
// (64 B each, already in operand order); a wave keeps the operands of 128 row hashes in registers and
// walks the tile's columns out of LDS: 8 MFMAs per 16 columns, then one running maximum over the 32
// results per lane.  Only when a 128 x 16 block holds a similarity >= 64 - threshold does the wave
// re-inspect it and apply the full predicate (band lanes, ids, size ratio, bucket cap) to the original
// 64-bit hashes and append edges through one atomic cursor.  The VALU version of the same loop
// (2 xor + 2 popcount + min per pair) topped out at 7-8 Tpairs/s on VALU issue.
#include <cmath>

#include <hipcub/hipcub.hpp>

#include "ke_internal.h"

namespace {

constexpr int kThreads = 512;               // 8 waves
constexpr int kRT = 8;                      // 16-hash row tiles per wave
constexpr int kTile = 8 * kRT * 16;         // 1024 rows per workgroup ...
constexpr int kCols = 1024;                 // ... against 1024 columns
constexpr int kCPR = kTile / kCols;         // column chunks per row block
constexpr int kChunk = 256;                 // columns staged in LDS at a time (16 operand tiles, 16 KB)

typedef int ke_v8i __attribute__((ext_vector_type(8)));
typedef int ke_v4i __attribute__((ext_vector_type(4)));
typedef float ke_v4f __attribute__((ext_vector_type(4)));

struct ScanArgs {
    const uint64_t *hashes;
    const ke_v4i *expanded;  // operand tiles: [hash / 16][lane = group * 16 + hash % 16] x 16 B, zero beyond n
    const int64_t *ids;
    const int64_t *sizes;
    int64_t n;
    int nb;                 // row blocks
    int ncc;                // column chunks
    int64_t first_tile;     // ordinal of this launch's first tile in the triangle enumeration
    int64_t tile_step;
    int64_t n_tiles;        // total tiles in the triangle
    int threshold, band_bits, band_count;
    double size_ratio;
    const uint32_t *hist;   // nullable: band_count tables of 2^band_bits bucket sizes (band_bits <= 24) ...
    const uint32_t *blen;   // ... or, for wider bands, the size of hash i's bucket in band b at blen[b * n + i]
    unsigned long long cap;
    ke_edge *edges;
    int64_t capacity;
    unsigned long long *counters;  // [0] pairs, [1] sum of shared bands, [2] edges
    int xcd_remap;
};

__device__ __forceinline__ int popc64(uint32_t lo, uint32_t hi) { return __popc(lo) + __popc(hi); }

// Full predicate for one pair that already passed popcount <= threshold.
__device__ void consider_pair(const ScanArgs &a, int64_t gi, int64_t gj, uint64_t x, uint64_t y, int pc) {
    if (gi >= gj || gj >= a.n) return;
    if (a.ids && a.ids[gi] == a.ids[gj]) return;                       // src/dup/scanner.py:266
    if (a.sizes && a.size_ratio > 0.0) {                               // :358-370
        const int64_t sa = a.sizes[gi], sb = a.sizes[gj];
        if (sa > 0 && sb > 0) {
            const int64_t small = sa < sb ? sa : sb, large = sa < sb ? sb : sa;
            if (!((double)small / (double)large >= a.size_ratio)) return;
        }
    }
    const uint64_t d = x ^ y;
    const uint64_t mask = a.band_bits >= 64 ? ~0ull : ((1ull << a.band_bits) - 1ull);
    int bands = 0, shared = 0;
    for (int b = 0; b < a.band_count; ++b) {
        const int sh = b * a.band_bits;
        if (((d >> sh) & mask) != 0) continue;
        if (a.cap) {                                                   // KE_DUP_BUCKET_PAIR_CAP, :262-263
            const unsigned long long len = a.hist ? a.hist[((size_t)b << a.band_bits) + (size_t)((x >> sh) & mask)]
                                                  : a.blen[(size_t)b * (size_t)a.n + (size_t)gi];
            if (len * (len - 1) / 2 > a.cap) continue;
        }
        bands |= 1 << (b < 31 ? b : 31);
        ++shared;
    }
    if (!shared) return;
    atomicAdd(&a.counters[1], (unsigned long long)shared);
    const unsigned long long slot = atomicAdd(&a.counters[2], 1ull);
    if ((int64_t)slot < a.capacity) {
        ke_edge e;
        e.a = gi; e.b = gj; e.h = pc; e.bands = bands;
        a.edges[slot] = e;
    }
}

// 8 bits -> 8 E2M1 nibbles (bit n -> nibble n = 0b0010 = 1.0 or 0)
__device__ __forceinline__ uint32_t spread8(uint32_t b) {
    uint32_t t = (b | (b << 12)) & 0x000F000Fu;
    t = (t | (t << 6)) & 0x03030303u;
    t = (t | (t << 3)) & 0x11111111u;
    return t << 1;
}

// Operand image of the hashes.  Lane group g of a 16-hash tile holds 32 of the 128 values of each hash:
// g = 0, 1: bits 0..31 / 32..63 of x; g = 2, 3: the same bits of ~x.  A and B operands use the same image
// (the k <-> (group, nibble) map of the instruction is the same for both, so any fixed order works).
__global__ __launch_bounds__(256) void ke_scan_expand(const uint64_t *__restrict__ hashes, int64_t n, int64_t n_pad,
                                                       ke_v4i *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;     // = tile * 64 + lane
    if (e >= n_pad * 4) return;
    const int64_t i = (e >> 6) * 16 + (e & 15);
    const int g = (int)(e >> 4) & 3;
    ke_v4i v = {0, 0, 0, 0};
    if (i < n) {
        const uint64_t x = g < 2 ? hashes[i] : ~hashes[i];
        const uint32_t half = (uint32_t)(x >> (32 * (g & 1)));
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (int)spread8((half >> (8 * k)) & 0xFFu);
    }
    out[e] = v;
}

__global__ __launch_bounds__(kThreads) void ke_scan_tiles(const ScanArgs a) {
    __shared__ ke_v4i s_b[2][kChunk / 16 * 64];      // two chunks of 16 operand tiles
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // workgroup b runs on XCD b mod 8 (round-robin dispatch), each XCD has its own L2: XCD x takes the x-th eighth of this
    // launch's tiles, consecutive tiles (which share their row operands) on one L2 instead of on eight.  KE_SCAN_XCD=0: off.
    int64_t slot = blockIdx.x;
    if (a.xcd_remap) {
        const int64_t per = ((int64_t)gridDim.x + 7) / 8;
        slot = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if (slot >= (int64_t)gridDim.x) return;
    }
    const int64_t t = a.first_tile + slot * a.tile_step;
    if (t >= a.n_tiles) return;
    // Tiles of the upper triangle, row-major: row block rb owns column chunks kCPR*rb .. ncc-1, so
    // offset(rb) = rb*ncc - kCPR*rb*(rb-1)/2.  Invert with a float estimate and an exact fix-up.
    const double hc = 0.5 * kCPR, bq = (double)a.ncc + hc;
    int64_t rb = (int64_t)floor((bq - sqrt(fmax(bq * bq - 4.0 * hc * (double)t, 0.0))) / (2.0 * hc));
    if (rb < 0) rb = 0;
    if (rb >= a.nb) rb = a.nb - 1;
    while (rb > 0 && rb * a.ncc - kCPR * rb * (rb - 1) / 2 > t) --rb;
    while (rb + 1 < a.nb && (rb + 1) * a.ncc - kCPR * (rb + 1) * rb / 2 <= t) ++rb;
    const int64_t cc = kCPR * rb + (t - (rb * a.ncc - kCPR * rb * (rb - 1) / 2));
    const int64_t row0 = rb * kTile, col0 = cc * kCols;
    const int64_t wrow0 = row0 + (int64_t)wv * (kRT * 16);          // this wave's 128 rows
    if (tid == 0) {
        // pairs i < j < n this tile stands for, counted where they are evaluated: the shards' counts must add up to
        // n (n - 1) / 2 (counters[0]; the host checks it against its own arithmetic)
        const int64_t rows = min((int64_t)kTile, a.n - row0), cols = min((int64_t)kCols, a.n - col0);
        const unsigned long long pairs = cc == kCPR * rb ? (unsigned long long)(rows * (rows - 1) / 2) : (unsigned long long)(rows * cols);
        atomicAdd(&a.counters[0], pairs);
    }

    // row operands: registers for the whole tile (only dwords 0..3 of an fp4 operand are read)
    ke_v8i ra[kRT];
#pragma unroll
    for (int r = 0; r < kRT; ++r) {
        const ke_v4i v = a.expanded[(wrow0 / 16 + r) * 64 + lane];
        ra[r] = ke_v8i{v[0], v[1], v[2], v[3], 0, 0, 0, 0};
    }
    // The similarities are small non-negative integers held as floats, so their bit patterns order like
    // integers: the running maximum is an integer max (a float max would first canonicalise every input).
    const int thr_bits = __float_as_int((float)(64 - a.threshold));
    const int nchunks = (int)((min((int64_t)kCols, a.n - col0) + kChunk - 1) / kChunk);
    // column operands: 16 KB per chunk, two 16-byte pieces per thread, next chunk in flight during the products
    const ke_v4i *cbase = a.expanded + (col0 / 16) * 64;
    ke_v4i pre0 = cbase[tid], pre1 = cbase[kThreads + tid];
    s_b[0][tid] = pre0;
    s_b[0][kThreads + tid] = pre1;
    // every load so far has landed from here on: without this hipcc re-waits for the row operands inside the
    // loop with counts that also drain the column prefetch issued at the top of each iteration
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        {   // unconditional (the last iteration re-reads its own chunk): a guarded load is waited for at once
            const int cn = c + 1 < nchunks ? c + 1 : c;
            pre0 = cbase[cn * (kChunk / 16 * 64) + tid];
            pre1 = cbase[cn * (kChunk / 16 * 64) + kThreads + tid];
        }
        const ke_v4i *bt = s_b[c & 1];
        // diagonal tiles: a 128 x 16 block whose every column index is below the wave's first row holds no pair i < j --
        // the wave starts at the first block that can (wave-uniform)
        const int64_t below = wrow0 - (col0 + (int64_t)c * kChunk) - 15;
        const int ct0 = below < 0 ? 0 : (int)min((int64_t)(kChunk / 16), below / 16 + 1);
        for (int ct = ct0; ct < kChunk / 16; ++ct) {
            const ke_v4i bv = bt[ct * 64 + lane];
            const ke_v8i b = {bv[0], bv[1], bv[2], bv[3], 0, 0, 0, 0};
            ke_v4f acc[kRT];
#pragma unroll
            for (int r = 0; r < kRT; ++r)
                acc[r] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ra[r], b, ke_v4f{0, 0, 0, 0}, 4, 4, 0, 0, 0, 0);
            int m = 0;
#pragma unroll
            for (int r = 0; r < kRT; ++r) {
                m = max(max(m, __float_as_int(acc[r][0])), __float_as_int(acc[r][1]));
                m = max(max(m, __float_as_int(acc[r][2])), __float_as_int(acc[r][3]));
            }
            if (__ballot(m >= thr_bits) != 0ull) {
                // Rare: collect which of this lane's 32 results qualify into a bit mask, then walk the mask with the
                // original hashes (result map: lane l holds rows 4(l>>4) + i of row tile r, column l & 15).
                uint32_t hits = 0;
#pragma unroll
                for (int r = 0; r < kRT; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) hits |= __float_as_int(acc[r][i]) >= thr_bits ? 1u << (4 * r + i) : 0u;
                const int64_t gj = col0 + (int64_t)c * kChunk + ct * 16 + (lane & 15);
                if (gj >= a.n) hits = 0;
                const uint64_t y = hits ? a.hashes[gj] : 0ull;
                while (hits) {
                    const int k = __ffs(hits) - 1;
                    hits &= hits - 1;
                    const int64_t gi = wrow0 + (k >> 2) * 16 + 4 * (lane >> 4) + (k & 3);
                    if (gi < gj) {
                        const uint64_t x = a.hashes[gi];
                        consider_pair(a, gi, gj, x, y, __popcll(x ^ y));
                    }
                }
            }
        }
        s_b[(c + 1) & 1][tid] = pre0;
        s_b[(c + 1) & 1][kThreads + tid] = pre1;
        __syncthreads();
    }
}

__global__ void ke_band_hist(const uint64_t *__restrict__ hashes, int64_t n, int band_bits, int band_count,
                             uint32_t *__restrict__ hist) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t x = hashes[i];
    const uint64_t mask = (1ull << band_bits) - 1ull;
    for (int b = 0; b < band_count; ++b) atomicAdd(&hist[((size_t)b << band_bits) + (size_t)((x >> (b * band_bits)) & mask)], 1u);
}

// sum over buckets of C(len, 2) for the buckets the reference would walk (len >= 2, pairs <= cap when a cap is set):
// its "pairs total" funnel counter before the same-file-id exclusion (src/dup/scanner.py:258-270)
__global__ void ke_hist_pairs(const uint32_t *__restrict__ hist, size_t bins, unsigned long long cap, unsigned long long *out) {
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < bins; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long len = hist[i];
        const unsigned long long p = len * (len - 1) / 2;
        if (len >= 2 && (!cap || p <= cap)) s += p;
    }
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d);
    __shared__ unsigned long long part[4];                    // one atomic per workgroup: thousands to one address serialise
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0 && (part[0] | part[1] | part[2] | part[3])) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// Wide bands (2^band_bits bins do not fit a table): the band values are sorted together with their positions and the
// run lengths are read off the sorted keys.
__global__ void ke_band_keys(const uint64_t *__restrict__ hashes, int64_t n, int shift, uint64_t mask, uint64_t *__restrict__ keys,
                             uint32_t *__restrict__ pos) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = (hashes[i] >> shift) & mask;
    pos[i] = (uint32_t)i;
}

// thread i starts a run iff keys[i] != keys[i-1]; it walks to the end of its run, writes the length to every member's
// slot and adds the run's pairs.  Runs are short unless the corpus is degenerate (then the walk is long but correct).
__global__ void ke_run_lengths(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ pos, int64_t n, unsigned long long cap,
                               uint32_t *__restrict__ blen, unsigned long long *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i > 0 && keys[i - 1] == keys[i]) return;
    int64_t e = i + 1;
    while (e < n && keys[e] == keys[i]) ++e;
    const unsigned long long len = (unsigned long long)(e - i);
    for (int64_t k = i; k < e; ++k) blen[pos[k]] = (uint32_t)len;
    const unsigned long long p = len * (len - 1) / 2;
    if (len >= 2 && (!cap || p <= cap)) atomicAdd(out, p);
}

// ---- the "size=" funnel counter (src/dup/scanner.py:268-270, 358-370): bucket pairs that pass the size filter ----------
// Positions sorted by size once; per band a stable sort by band value then leaves every bucket contiguous with its sizes
// ascending, and a member's partners are the earlier members whose size is large enough -- found by a binary search whose
// test is the reference's own float division (smaller / larger >= ratio, monotone in the smaller size).
__global__ void ke_size_keys(const int64_t *__restrict__ sizes, int64_t n, uint64_t *__restrict__ keys, uint32_t *__restrict__ pos) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t s = sizes[i];
    keys[i] = s > 0 ? (uint64_t)s : 0ull;                     // a missing / non-positive size passes with everyone
    pos[i] = (uint32_t)i;
}

__global__ void ke_band_keys_of(const uint64_t *__restrict__ hashes, const uint32_t *__restrict__ order, int64_t n, int shift,
                                uint64_t mask, uint64_t *__restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = (hashes[order[i]] >> shift) & mask;
}

__global__ void ke_run_flags(const uint64_t *__restrict__ keys, int64_t n, uint32_t *__restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

// rank[i] = number of the run position i lies in (inclusive sum of the flags, minus one); starts[r] = first position of run r
__global__ void ke_run_starts(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ rank1, int64_t n, uint32_t *__restrict__ starts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i == 0 || keys[i] != keys[i - 1]) starts[rank1[i] - 1] = (uint32_t)i;
    if (i == n - 1) starts[rank1[i]] = (uint32_t)n;
}

__global__ void ke_size_pairs(const uint64_t *__restrict__ sz, const uint32_t *__restrict__ rank1, const uint32_t *__restrict__ starts,
                              int64_t n, double ratio, unsigned long long cap, unsigned long long *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c = 0;
    if (i < n) {
        const uint32_t r = rank1[i] - 1;
        const int64_t st = starts[r], en = starts[r + 1], m = en - st;
        if (m >= 2 && (!cap || (unsigned long long)m * (unsigned long long)(m - 1) / 2 <= cap)) {
            const uint64_t si = sz[i];
            if (si == 0) {
                c = (unsigned long long)(i - st);             // every earlier member is size-free as well
            } else {
                int64_t lo = st, hi = i;                      // first member with a positive size
                while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (sz[mid] == 0) lo = mid + 1; else hi = mid; }
                const int64_t free_n = lo - st;
                hi = i;                                       // first positive member j < i with sz[j] / si >= ratio
                const double larger = (double)si;
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if ((double)sz[mid] / larger >= ratio) hi = mid; else lo = mid + 1;
                }
                c = (unsigned long long)(i - lo) + (unsigned long long)free_n;
            }
        }
    }
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0 && (part[0] | part[1] | part[2] | part[3])) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

}  // namespace

int ke_launch_band_pairs_after_size(ke_ctx *ctx, const uint64_t *d_hashes, const int64_t *d_sizes, int64_t n, int band_bits,
                                    int band_count, double ratio, int64_t bucket_pair_cap, unsigned long long *d_out) {
    if (n > 0x7fffffffLL) return ke_fail(ctx, KE_EUNSUPPORTED, "the size funnel counter needs n < 2^31");
    const unsigned long long cap = bucket_pair_cap > 0 ? (unsigned long long)bucket_pair_cap : 0ull;
    size_t t_size = 0, t_band = 0, t_scan = 0;
    KE_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, t_size, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                                   (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n, 0, 64, ctx->stream));
    KE_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, t_band, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                                   (const uint64_t *)nullptr, (uint64_t *)nullptr, (int)n, 0, band_bits, ctx->stream));
    KE_HIP(ctx, hipcub::DeviceScan::InclusiveSum(nullptr, t_scan, (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n, ctx->stream));
    const size_t temp_bytes = std::max(t_size, std::max(t_band, t_scan));
    const size_t kb = ((size_t)n * 8 + 255) & ~(size_t)255, pb = ((size_t)(n + 1) * 4 + 255) & ~(size_t)255;
    void *scratch;
    KE_TRY(ke_reserve(ctx, KE_BUF_SCAN_AUX, 4 * kb + 4 * pb + temp_bytes + 256, &scratch));
    uint8_t *sp = (uint8_t *)scratch;
    uint64_t *k_a = (uint64_t *)sp, *k_b = (uint64_t *)(sp + kb), *sz_sorted = (uint64_t *)(sp + 2 * kb), *sz_band = (uint64_t *)(sp + 3 * kb);
    uint32_t *p_a = (uint32_t *)(sp + 4 * kb), *order = (uint32_t *)(sp + 4 * kb + pb), *rank1 = (uint32_t *)(sp + 4 * kb + 2 * pb),
             *starts = (uint32_t *)(sp + 4 * kb + 3 * pb);
    void *temp = sp + 4 * kb + 4 * pb;
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    const uint64_t mask = band_bits >= 64 ? ~0ull : ((1ull << band_bits) - 1ull);
    hipLaunchKernelGGL(ke_size_keys, grid, blk, 0, ctx->stream, d_sizes, n, k_a, p_a);
    size_t tb = temp_bytes;
    KE_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(temp, tb, (const uint64_t *)k_a, sz_sorted, (const uint32_t *)p_a, order, (int)n, 0, 64, ctx->stream));
    for (int b = 0; b < band_count; ++b) {
        hipLaunchKernelGGL(ke_band_keys_of, grid, blk, 0, ctx->stream, d_hashes, (const uint32_t *)order, n, b * band_bits, mask, k_a);
        tb = temp_bytes;
        KE_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(temp, tb, (const uint64_t *)k_a, k_b, (const uint64_t *)sz_sorted, sz_band, (int)n, 0,
                                                       band_bits, ctx->stream));
        uint32_t *flags = p_a;                                // the positions are not needed again
        hipLaunchKernelGGL(ke_run_flags, grid, blk, 0, ctx->stream, (const uint64_t *)k_b, n, flags);
        tb = temp_bytes;
        KE_HIP(ctx, hipcub::DeviceScan::InclusiveSum(temp, tb, (const uint32_t *)flags, rank1, (int)n, ctx->stream));
        hipLaunchKernelGGL(ke_run_starts, grid, blk, 0, ctx->stream, (const uint64_t *)k_b, (const uint32_t *)rank1, n, starts);
        hipLaunchKernelGGL(ke_size_pairs, grid, blk, 0, ctx->stream, (const uint64_t *)sz_band, (const uint32_t *)rank1,
                           (const uint32_t *)starts, n, ratio, cap, d_out);
        KE_HIP(ctx, hipGetLastError());
    }
    return KE_OK;
}

int ke_launch_scan(ke_ctx *ctx, const uint64_t *d_hashes, const int64_t *d_ids, const int64_t *d_sizes, int64_t n,
                   int part_index, int part_count, int threshold, int band_bits, int band_count, double size_ratio,
                   int64_t bucket_pair_cap, ke_edge *d_edges, int64_t capacity, unsigned long long *d_counters,
                   unsigned long long *pairs_evaluated, bool want_bucket_pairs) {
    ScanArgs a;
    a.hashes = d_hashes; a.ids = d_ids; a.sizes = (size_ratio > 0.0) ? d_sizes : nullptr;
    a.n = n;
    a.nb = (int)((n + kTile - 1) / kTile);
    a.ncc = (int)((n + kCols - 1) / kCols);
    a.n_tiles = (int64_t)a.nb * a.ncc - (int64_t)kCPR * a.nb * (a.nb - 1) / 2;
    a.first_tile = part_index;
    a.tile_step = part_count;
    a.threshold = threshold; a.band_bits = band_bits; a.band_count = band_count;
    a.size_ratio = size_ratio;
    a.hist = nullptr; a.blen = nullptr;
    a.cap = bucket_pair_cap > 0 ? (unsigned long long)bucket_pair_cap : 0ull;
    a.edges = d_edges; a.capacity = capacity; a.counters = d_counters;
    // Bucket sizes per band: needed by the pair cap, and they give the reference's "pairs total" counter (counters[3]).
    if ((bucket_pair_cap > 0 || want_bucket_pairs) && band_bits <= 24) {
        void *h;
        const size_t bins = (size_t)band_count << band_bits, bytes = bins * sizeof(uint32_t);
        KE_TRY(ke_reserve(ctx, KE_BUF_SCAN_HIST, bytes, &h));
        KE_HIP(ctx, hipMemsetAsync(h, 0, bytes, ctx->stream));
        hipLaunchKernelGGL(ke_band_hist, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_hashes, n, band_bits,
                           band_count, (uint32_t *)h);
        KE_HIP(ctx, hipGetLastError());
        a.hist = (const uint32_t *)h;
        if (want_bucket_pairs) {
            const unsigned blocks = (unsigned)std::min<size_t>((bins + 1023) / 1024, 256);
            hipLaunchKernelGGL(ke_hist_pairs, dim3(blocks), dim3(256), 0, ctx->stream, (const uint32_t *)h, bins, a.cap, d_counters + 3);
            KE_HIP(ctx, hipGetLastError());
        }
    } else if (bucket_pair_cap > 0 || want_bucket_pairs) {
        if (n > 0xffffffffLL) return ke_fail(ctx, KE_EUNSUPPORTED, "bands wider than 24 bits need n < 2^32");
        // sort (band value, position) per band; scratch: keys in/out, positions in/out, per-hash lengths, cub temp storage
        size_t temp_bytes = 0;
        KE_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                                       (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n, 0, band_bits, ctx->stream));
        const size_t kb = ((size_t)n * 8 + 255) & ~(size_t)255, pb = ((size_t)n * 4 + 255) & ~(size_t)255;
        void *scratch, *lens;
        KE_TRY(ke_reserve(ctx, KE_BUF_SCAN_AUX, 2 * kb + 2 * pb + temp_bytes + 256, &scratch));
        KE_TRY(ke_reserve(ctx, KE_BUF_SCAN_HIST, (size_t)band_count * n * sizeof(uint32_t), &lens));
        uint8_t *sp = (uint8_t *)scratch;
        uint64_t *k_in = (uint64_t *)sp, *k_out = (uint64_t *)(sp + kb);
        uint32_t *p_in = (uint32_t *)(sp + 2 * kb), *p_out = (uint32_t *)(sp + 2 * kb + pb);
        void *temp = sp + 2 * kb + 2 * pb;
        const uint64_t mask = band_bits >= 64 ? ~0ull : ((1ull << band_bits) - 1ull);
        const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
        for (int b = 0; b < band_count; ++b) {
            hipLaunchKernelGGL(ke_band_keys, grid, blk, 0, ctx->stream, d_hashes, n, b * band_bits, mask, k_in, p_in);
            KE_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, k_in, k_out, p_in, p_out, (int)n, 0, band_bits, ctx->stream));
            hipLaunchKernelGGL(ke_run_lengths, grid, blk, 0, ctx->stream, k_out, p_out, n, a.cap, (uint32_t *)lens + (size_t)b * n,
                               want_bucket_pairs ? d_counters + 3 : d_counters + 3);
            KE_HIP(ctx, hipGetLastError());
        }
        a.blen = (const uint32_t *)lens;
    }
    const int64_t my_tiles = a.n_tiles > part_index ? (a.n_tiles - part_index + part_count - 1) / part_count : 0;
    const int64_t n_pad = (int64_t)a.nb * kTile;     // the kernel reads whole tiles: operands beyond n are zero
    void *exp;
    KE_TRY(ke_reserve(ctx, KE_BUF_SCAN_EXP, (size_t)n_pad * 64, &exp));
    a.expanded = (const ke_v4i *)exp;
    ke_time_begin(ctx, KE_T_SCAN);
    if (my_tiles > 0) {
        hipLaunchKernelGGL(ke_scan_expand, dim3((unsigned)((n_pad * 4 + 255) / 256)), dim3(256), 0, ctx->stream, d_hashes, n, n_pad,
                           (ke_v4i *)exp);
        if (my_tiles > 0x7fffffffLL) return ke_fail(ctx, KE_EUNSUPPORTED, "too many tiles for one launch");
        static const int xcd = [] { const char *e = std::getenv("KE_SCAN_XCD"); return e ? std::atoi(e) : 0; }();
        a.xcd_remap = xcd;
        // with the remap the grid is padded to a multiple of 8 so that every slot exists on some workgroup
        const int64_t grid = xcd ? (my_tiles + 7) / 8 * 8 : my_tiles;
        hipLaunchKernelGGL(ke_scan_tiles, dim3((unsigned)grid), dim3(kThreads), 0, ctx->stream, a);
        KE_HIP(ctx, hipGetLastError());
    }
    ke_time_end(ctx, KE_T_SCAN);
    // pairs evaluated by this shard (host, O(row blocks)): per row block, the chunks right of the diagonal are
    // full rectangles, the kCPR chunks on the diagonal hold a partial triangle.
    unsigned long long pairs = 0;
    for (int64_t rb = 0; rb < a.nb; ++rb) {
        const int64_t off = rb * a.ncc - (int64_t)kCPR * rb * (rb - 1) / 2;
        const int64_t row0 = rb * kTile, rows = std::min<int64_t>(kTile, n - row0);
        for (int64_t cc = kCPR * rb; cc < a.ncc; ++cc) {
            const int64_t t = off + (cc - kCPR * rb);
            const bool diagonal = cc < kCPR * (rb + 1);
            if (!diagonal) {
                // every remaining chunk is a full rectangle except possibly the last one: count them arithmetically
                const int64_t last = a.ncc - 1;
                const int64_t t_last = off + (last - kCPR * rb);
                // chunks cc .. last-1 (full width)
                const int64_t first_full_t = t, end_full_t = t_last;   // [first_full_t, end_full_t)
                if (end_full_t > first_full_t) {
                    // count t in [first_full_t, end_full_t) with t % part_count == part_index
                    auto upto = [&](int64_t x) { return x <= part_index ? 0 : (x - part_index + part_count - 1) / part_count; };
                    pairs += (unsigned long long)(upto(end_full_t) - upto(first_full_t)) * (unsigned long long)(rows * kCols);
                }
                if (t_last % part_count == part_index && last >= cc)
                    pairs += (unsigned long long)(rows * std::min<int64_t>(kCols, n - last * kCols));
                break;
            }
            if (t % part_count != part_index) continue;
            const int64_t col0 = cc * kCols, cols = std::min<int64_t>(kCols, n - col0);
            for (int64_t j = col0; j < col0 + cols; ++j)
                pairs += (unsigned long long)std::max<int64_t>(0, std::min<int64_t>(rows, j - row0));
        }
    }
    *pairs_evaluated = pairs;
    return KE_OK;
}
