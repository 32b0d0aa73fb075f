// ke_scan.hip -- K4: all-pairs Hamming candidate scan for gfx950.
//
// Replaces the bucket build + in-bucket pair loop of DuplicateScanner.build_clusters
// (src/dup/scanner.py:227-299).  The pair space i<j is cut into TILE x TILE tiles of the upper
// triangle; a workgroup owns one tile: each thread keeps RI row hashes in registers, the
// column block sits in LDS and is read as wave-uniform (broadcast) 16-byte loads.  The hot
// loop is 2 xor + 2 popcount + a running minimum per pair; only when the minimum of a group
// of columns drops to the threshold does the thread re-walk that group and apply the full
// predicate (band lanes, ids, size ratio, bucket cap) and append edges through one atomic
// cursor.  No MFMA: nothing here is a contraction.
#include <cmath>

#include "ke_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kRI = 4;                      // row hashes per thread
constexpr int kTile = kThreads * kRI;       // 1024 rows per workgroup ...
constexpr int kCols = 256;                  // ... against 256 columns: fine-grained tiles keep the tail short at N = 100k
constexpr int kCPR = kTile / kCols;         // column chunks per row block
constexpr int kGroup = 8;                   // columns between threshold checks

struct ScanArgs {
    const uint64_t *hashes;
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
    const uint32_t *hist;   // nullable: band_count tables of 2^band_bits bucket sizes
    unsigned long long cap;
    ke_edge *edges;
    int64_t capacity;
    unsigned long long *counters;  // [0] pairs, [1] sum of shared bands, [2] edges
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
        if (a.hist) {                                                  // KE_DUP_BUCKET_PAIR_CAP, :262-263
            const unsigned long long len = a.hist[((size_t)b << a.band_bits) + (size_t)((x >> sh) & mask)];
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

__global__ __launch_bounds__(kThreads) void ke_scan_tiles(const ScanArgs a) {
    __shared__ __attribute__((aligned(16))) uint64_t s_cols[kCols];
    const int tid = threadIdx.x;
    const int64_t t = a.first_tile + (int64_t)blockIdx.x * a.tile_step;
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

    uint32_t xlo[kRI], xhi[kRI];
#pragma unroll
    for (int r = 0; r < kRI; ++r) {
        const int64_t gi = row0 + r * kThreads + tid;
        const uint64_t x = gi < a.n ? a.hashes[gi] : 0ull;
        xlo[r] = (uint32_t)x;
        xhi[r] = (uint32_t)(x >> 32);
    }
    {
        const int64_t gj = col0 + tid;
        s_cols[tid] = gj < a.n ? a.hashes[gj] : ~0ull;
    }
    __syncthreads();

    const int thr = a.threshold;
    const int ncols = (int)((a.n - col0) < kCols ? (a.n - col0) : kCols);
    for (int c0 = 0; c0 < ncols; c0 += kGroup) {
        int best = 64;
#pragma unroll
        for (int c = 0; c < kGroup; c += 2) {
            const uint4 yy = *reinterpret_cast<const uint4 *>(&s_cols[c0 + c]);  // two columns, broadcast read
#pragma unroll
            for (int r = 0; r < kRI; ++r) {
                const int p0 = popc64(xlo[r] ^ yy.x, xhi[r] ^ yy.y);
                const int p1 = popc64(xlo[r] ^ yy.z, xhi[r] ^ yy.w);
                best = min(best, min(p0, p1));
            }
        }
        if (best <= thr) {
            for (int c = 0; c < kGroup; ++c) {
                const uint64_t y = s_cols[c0 + c];
#pragma unroll
                for (int r = 0; r < kRI; ++r) {
                    const uint64_t x = ((uint64_t)xhi[r] << 32) | xlo[r];
                    const int pc = __popcll(x ^ y);
                    if (pc <= thr) consider_pair(a, row0 + r * kThreads + tid, col0 + c0 + c, x, y, pc);
                }
            }
        }
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

}  // namespace

int ke_launch_scan(ke_ctx *ctx, const uint64_t *d_hashes, const int64_t *d_ids, const int64_t *d_sizes, int64_t n,
                   int part_index, int part_count, int threshold, int band_bits, int band_count, double size_ratio,
                   int64_t bucket_pair_cap, ke_edge *d_edges, int64_t capacity, unsigned long long *d_counters,
                   unsigned long long *pairs_evaluated) {
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
    a.hist = nullptr;
    a.cap = bucket_pair_cap > 0 ? (unsigned long long)bucket_pair_cap : 0ull;
    a.edges = d_edges; a.capacity = capacity; a.counters = d_counters;
    if (bucket_pair_cap > 0) {
        void *h;
        const size_t bytes = ((size_t)band_count << band_bits) * sizeof(uint32_t);
        KE_TRY(ke_reserve(ctx, KE_BUF_SCAN_HIST, bytes, &h));
        KE_HIP(ctx, hipMemsetAsync(h, 0, bytes, ctx->stream));
        hipLaunchKernelGGL(ke_band_hist, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_hashes, n, band_bits,
                           band_count, (uint32_t *)h);
        KE_HIP(ctx, hipGetLastError());
        a.hist = (const uint32_t *)h;
    }
    const int64_t my_tiles = a.n_tiles > part_index ? (a.n_tiles - part_index + part_count - 1) / part_count : 0;
    ke_time_begin(ctx, KE_T_SCAN);
    if (my_tiles > 0) {
        if (my_tiles > 0x7fffffffLL) return ke_fail(ctx, KE_EUNSUPPORTED, "too many tiles for one launch");
        hipLaunchKernelGGL(ke_scan_tiles, dim3((unsigned)my_tiles), dim3(kThreads), 0, ctx->stream, a);
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
