// ke_internal.h -- shared declarations of libkeyes_hip.so (gfx950 only, no CPU fallback).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/keyes.h"

#define KE_API extern "C" __attribute__((visibility("default")))

// ---------------------------------------------------------------------------------------
// Lanczos coefficient tables (host side: ke_coeffs.cpp).
// plain  : Pillow's own layout, bounds[o] = (first tap, tap count), kk[o][ksize] 22-bit ints.
// packed : the layout the fused kernel eats.  For output o the window starts at byte
//          start[o] (multiple of 8) and spans 4*ndw bytes; tap weight k is split into three
//          balanced signed bytes k = b0 + 256*b1 + 65536*b2, stored per window dword as three
//          packed i8x4 words (one per byte plane) so a v_dot4_i32_i8 against four signed luma
//          bytes (L-128) accumulates each plane; bias[o] = 128*sum(k) + 2^21.
// ---------------------------------------------------------------------------------------
struct KeAxisCoeffs {
    int in_size = 0, out_size = 0;
    int ksize = 0;
    std::vector<int32_t> bounds;  // 2*out
    std::vector<int32_t> kk;      // out*ksize
    int ndw = 0;                  // window dwords (multiple of 4)
    int span = 0;                 // max(start[o] + 4*ndw)
    std::vector<int32_t> start;   // out
    std::vector<int32_t> bias;    // out
    std::vector<int32_t> packed;  // out*ndw*3
    // device copies
    int32_t *d_bounds = nullptr, *d_kk = nullptr, *d_start = nullptr, *d_bias = nullptr, *d_packed = nullptr;
    // chunked byte-plane layouts, by chunks-per-output (see KeChunkTable)
    std::map<int, struct KeChunkTable *> chunked;
    // matrix-core operand layouts of the same byte planes (see KeMxTable), built on first use, by (min_ks, align64)
    std::map<int, struct KeMxTable *> mx;
};

// The tap matrix as B operands of v_mfma_i32_16x16x64_i8: the resample of one axis is the banded product
// out[row][o] = sum_x luma[row][x] * k[o][x]; outputs are taken 16 at a time (tile j = o / 16), the taps of a tile
// live in x in [base[j], base[j] + 64*ks), and step s of tile j, byte plane p is one 64x16 operand whose lane l
// holds the 16 bytes k_p[o = 16j + (l & 15)][x = base[j] + 64s + 16(l >> 4) + 0..15] (zero outside the window
// and for o >= out_size).  frag index: (((j*ks + s)*3 + p)*64 + l)*4 dwords.
struct KeMxTable {
    int tiles = 0, ks = 0;
    std::vector<int32_t> base;   // tiles, multiples of 16
    std::vector<int32_t> frag;
    int32_t *d_frag = nullptr;
};

// Chunked byte-plane layout for long windows: every output's packed window is cut into `cpo` chunks of
// `ndwc` dwords; virtual column v = o*cpo + c starts at cstart[v] and owns cpacked[v][ndwc][3]; the
// chunks' plane sums are added before bias[o] and the clip.
struct KeChunkTable {
    int cpo = 0, ndwc = 0, cspan = 0;
    std::vector<int32_t> cstart, cpacked;
    // cxor[v] in 0..3: lane v visits window dword pair p at physical pair p ^ cxor[v], chosen on the host so
    // that the 32 lanes of a half-wave hit distinct LDS banks with ds_read_b64 (needs ndwc % 8 == 0)
    std::vector<int32_t> cxor;
    int32_t *d_cstart = nullptr, *d_cpacked = nullptr, *d_cxor = nullptr;
};


void ke_build_axis_coeffs(int in_size, int out_size, KeAxisCoeffs &out, int filter, float in0, float in1);
void ke_build_chunked(const KeAxisCoeffs &c, int cpo, KeChunkTable &out, int ndwc_multiple = 4);
void ke_build_mx(const KeAxisCoeffs &c, KeMxTable &out, int min_ks = 0, bool align64 = false);

struct KeDevBuf {
    void *ptr = nullptr;
    size_t bytes = 0;
};

enum { KE_T_HASH = 0, KE_T_SCAN = 1, KE_T_SSIM = 2, KE_T_SYNTH = 3, KE_T_JPEG = 4, KE_T_COUNT = 5 };
enum {
    KE_BUF_PIXELS = 0,   // staged input images
    KE_BUF_TMP,          // first-pass output of the generic resampler
    KE_BUF_TILE32,
    KE_BUF_TILE98,
    KE_BUF_OUT0,         // staged outputs
    KE_BUF_OUT1,
    KE_BUF_OUT2,         // staged tie margins
    KE_BUF_META,         // offsets / out_idx arrays
    KE_BUF_SCAN_IN,
    KE_BUF_SCAN_AUX,
    KE_BUF_SCAN_EDGES,
    KE_BUF_SCAN_CNT,
    KE_BUF_SCAN_HIST,
    KE_BUF_SCAN_EXP,     // hashes expanded to matrix-core operands (64 B each)
    KE_BUF_SSIM_IN,
    KE_BUF_SSIM_AUX,
    KE_BUF_COMM,         // gathered hash shards before they are put back into corpus order
    KE_BUF_COMM_EDGES,   // edge records: [send | world x recv]
    KE_BUF_JPEG_TABLES,  // Huffman tables of a JPEG batch
    KE_BUF_COUNT
};

// Pinned staging (north-star step 1): producers (decode threads) write pixels straight into page-locked host buffers;
// a submit enqueues the H2D copy on a copy stream and the hash kernels behind an event on the compute stream, so the copy
// of batch k+1 overlaps the kernels of batch k and the host never blocks between batches.
constexpr int KE_MAX_STAGE_SLOTS = 4;
struct KeStageSlot {
    uint8_t *h_px = nullptr, *d_px = nullptr;        // pinned pixels and their device twin
    uint64_t *h_meta = nullptr, *d_meta = nullptr;   // per size group: [byte offsets | output slots]
    uint64_t *h_ph = nullptr, *h_dh = nullptr, *d_ph = nullptr, *d_dh = nullptr;
    float *h_mg = nullptr, *d_mg = nullptr;
    hipEvent_t copied = nullptr, done = nullptr;
    bool in_flight = false;
    bool px_external = false;                         // h_px is the caller's memory, registered with the runtime
    uint64_t *user_ph = nullptr, *user_dh = nullptr;  // where the results go when the slot is waited for
    float *user_mg = nullptr;
    int64_t n = 0;
};
struct KeStage {
    size_t bytes = 0;
    int64_t max_images = 0;
    int n_slots = 0, next = 0;
    hipStream_t copy_stream = nullptr;
    KeStageSlot slot[KE_MAX_STAGE_SLOTS];
};

struct ke_ctx {
    int device = 0;
    int cu_count = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    KeDevBuf buf[KE_BUF_COUNT];
    std::map<std::tuple<int, int, uint32_t, uint32_t>, KeAxisCoeffs *> coeffs;   // key: (in_size, out_size * 4 + filter, bits of in0, in1)
    hipEvent_t ev0[KE_T_COUNT] = {}, ev1[KE_T_COUNT] = {};
    bool ev_valid[KE_T_COUNT] = {};
    KeStage *stage = nullptr;
    hipStream_t side[2] = {nullptr, nullptr};   // side streams of the ragged hash call: neighbouring size groups overlap their tails
    hipEvent_t side_fork = nullptr, side_join[2] = {nullptr, nullptr};
    int64_t edge_slots = 1024;       // edges carried by one record of ke_allgather_edges (follows the largest list seen)
    void *h_comm = nullptr;          // pinned landing zone of the gathered edge records
    size_t h_comm_bytes = 0;
    void *h_meta = nullptr;          // page-locked per-image records of a decode batch on their way to the device, and the statuses back
    size_t h_meta_bytes = 0;
    bool ssim_exact = false;         // ke_ssim_set_mode: false = integer-sum kernel (default), true = fp64-carry kernel
    float *margin_cur = nullptr;     // device array the hash kernels of the CURRENT call write tie margins to (slot = hash slot)
    bool dct_tables_ready = false;   // __constant__ tables are per device: uploaded once per context
};

// error helpers ------------------------------------------------------------------------------
int ke_fail(ke_ctx *ctx, int code, const char *fmt, ...);
#define KE_HIP(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return ke_fail((ctx), KE_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                           __FILE__, __LINE__);                                                  \
    } while (0)
#define KE_TRY(expr)            \
    do {                        \
        int rc_ = (expr);       \
        if (rc_ != KE_OK) return rc_; \
    } while (0)

// memory helpers (ke_api.hip) ------------------------------------------------------------------
bool ke_is_device_ptr(const void *p);
int ke_reserve(ke_ctx *ctx, int which, size_t bytes, void **out);
// Returns in *dev a device pointer holding `bytes` of `p` (p itself if it already is device memory,
// else a staged copy in buffer `which`).
int ke_to_device(ke_ctx *ctx, const void *p, size_t bytes, int which, const void **dev);
// taps of one axis; (in0, in1) = source interval (Pillow's box), in1 < 0 means the whole axis
const KeAxisCoeffs *ke_get_coeffs(ke_ctx *ctx, int in_size, int out_size, int filter = KE_FILTER_LANCZOS, float in0 = 0.0f,
                                  float in1 = -1.0f);
const KeChunkTable *ke_get_chunks(ke_ctx *ctx, const KeAxisCoeffs *c, int cpo, int ndwc_multiple = 4);
// min_ks: build with at least this many steps per tile (zero operands past the window); a cached table keeps its own
// align64: every tile's first tap column is a multiple of 64 (the strip kernel cuts rows at multiples of 64)
const KeMxTable *ke_get_mx(ke_ctx *ctx, const KeAxisCoeffs *c, int min_ks = 0, bool align64 = false);
void ke_time_begin(ke_ctx *ctx, int kind);
void ke_time_end(ke_ctx *ctx, int kind);

// kernel launchers (one per .hip file) ------------------------------------------------------------
struct KeHashGroup {
    const uint8_t *pixels;     // device
    const uint64_t *offsets;   // device, nullable (then image k is at k*stride)
    uint64_t stride;
    const int64_t *out_idx;    // device, nullable (then output slot k)
    int64_t n;
    int w, h, channels;
    bool misaligned = false;   // some image of the group starts at an address that is not a multiple of 4
};
// fused_stream (nullable): where the scratch-free one-workgroup-per-image kernel of the group may run instead of ctx->stream
int ke_launch_hash_group(ke_ctx *ctx, const KeHashGroup &g, uint64_t *d_phash, uint64_t *d_dhash,
                         uint8_t *d_tile32_out, uint8_t *d_tile98_out, hipStream_t fused_stream = nullptr);
// luma + resize of a group to (oh x ow) u8 tiles with the given filter (banded path, generic fallback)
// box = {x0, y0, x1, y1} source rectangle (Pillow's resize box), NULL = the whole image
int ke_launch_resize_group(ke_ctx *ctx, const KeHashGroup &g, int ow, int oh, int filter, uint8_t *d_tiles,
                           const float *box = nullptr);
int ke_launch_tile_ahash(ke_ctx *ctx, const uint8_t *d_tiles, int64_t n, int grid, int tile, uint64_t *d_bits);
int ke_launch_sad_pairs(ke_ctx *ctx, const uint8_t *d_thumbs, int64_t pixels, const int64_t *d_pa, const int64_t *d_pb,
                        int64_t n_pairs, uint64_t *d_out);
int ke_launch_band_pairs_after_size(ke_ctx *ctx, const uint64_t *d_hashes, const int64_t *d_sizes, int64_t n, int band_bits,
                                    int band_count, double ratio, int64_t bucket_pair_cap, unsigned long long *d_out);
int ke_launch_scan(ke_ctx *ctx, const uint64_t *d_hashes, const int64_t *d_ids, const int64_t *d_sizes, int64_t n,
                   int part_index, int part_count, int threshold, int band_bits, int band_count, double size_ratio,
                   int64_t bucket_pair_cap, ke_edge *d_edges, int64_t capacity, unsigned long long *d_counters,
                   unsigned long long *pairs_evaluated, bool want_bucket_pairs);
int ke_launch_ssim(ke_ctx *ctx, const uint8_t *d_images, int w, int h, int channels, const int64_t *d_pa,
                   const int64_t *d_pb, int64_t n_pairs, double *d_out);
int ke_launch_synth_rgb(ke_ctx *ctx, uint64_t seed, int64_t first, const int64_t *d_indices, int64_t n, int w, int h,
                        uint8_t *d_out);
int ke_launch_synth_hashes(ke_ctx *ctx, uint64_t seed, int64_t n, uint64_t *d_out);

// Host-side loops over the files of a batch (header parsing): [0, n) cut into contiguous ranges, one per thread; fn(lo, hi, t).
// KE_HOST_THREADS overrides the thread count (default: the hardware's, at most 16, at least 256 items per thread).
template <typename Fn>
static inline int ke_parallel_ranges(int64_t n, Fn fn) {
    int want = (int)std::thread::hardware_concurrency();
    if (const char *e = std::getenv("KE_HOST_THREADS")) want = std::atoi(e);
    int nt = (int)std::min<int64_t>(std::max(want, 1), std::min<int64_t>(16, n / 256));
    if (nt <= 1) { fn((int64_t)0, n, 0); return 1; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back([=]() { fn(n * t / nt, n * (t + 1) / nt, t); });
    for (auto &x : th) x.join();
    return nt;
}
