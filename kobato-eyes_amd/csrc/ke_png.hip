// ke_png.hip -- PNG decoding on the GPU (8-bit grayscale / RGB / RGBA / gray + alpha, palette and sub-byte files; Adam7 too): the other half of the decode step in
// front of the hash path.  Replaces `Image.open(path)` + pixel access of the reference's batch hasher
// (src/core/fastsig.py:31-34) for the files it takes; the format is lossless, so the pixels are Pillow's by construction of
// the two specifications followed in ke_png_core.h (zlib/deflate, the five scanline filters).
//
//   ke_png_gather    the IDAT payloads of every file -> one contiguous, 16-byte aligned zlib stream per image
//   ke_png_inflate   ONE THREAD PER IMAGE walks the bit stream (it is sequential by construction).  Canonical-code decoding by
//                    comparison: the 15 limits and bases of both codes sit in registers, the symbols in code order in 424 bytes
//                    of LDS per lane next to a 64-byte ring of stream bytes that is topped up at wave-uniform moments.
//                    Literals are written to their final place; an LZ77 copy is only recorded.  The kernel never waits for
//                    memory inside the symbol loop: a wait by one lane would stall the other 63.
//   ke_png_matches   ONE WAVE PER IMAGE makes the recorded copies, 64 per round, each as soon as the bytes it reads are final.
//   ke_png_unfilter  ONE WAVE PER IMAGE: a pixel needs its left, upper and upper-left neighbours, so lane l takes rows
//                    l, l + 64, ... and runs one group of four pixels behind lane l - 1; the row above arrives by a lane shift
//                    (lane 63 -> lane 0 through an LDS row).  The same pass sums the Adler-32 of the filtered bytes by rows
//                    and holds it against the stream's trailer.  An Adam7 file is seven such images one after another in
//                    the stream (ke_adam7_pass); the same loop runs once per pass and scatters the pixels (ADAM7 instances).
#include <algorithm>

#include "ke_internal.h"
#include "ke_lz_copies.h"
#include "ke_png_parse.h"

namespace {

struct KePngDev {
    KePngInfo info;
    uint64_t raw_off;      // filtered scanlines (height * (1 + width * channels) bytes) inside the scratch
    uint64_t rec_off;      // this image's LZ77 copy records (8 bytes each; at most one per 3 output bytes)
    uint64_t out_off;      // bytes into the caller's pixel buffer
    uint64_t z_off;        // the image's zlib stream inside the staged stream bytes (64 bits: a batch of textured files holds
                           // more than 4 GB of them, and a batch cut short there is a launch with half the chip idle)
};

struct KePngPiece {        // part of one IDAT payload: bytes [src, src + len) of the uploaded files -> [dst, dst + len) of the streams
    uint64_t src, dst;
    uint32_t len, pad;
};

constexpr int kWorkBytes = 160;       // per image in HBM: the code lengths of the block header being read, 8 per dword
constexpr uint32_t kPieceBytes = 1u << 18;

__global__ __launch_bounds__(256) void ke_png_gather(const KePngPiece *__restrict__ pieces, const uint8_t *__restrict__ files,
                                                     uint8_t *__restrict__ streams) {
    const KePngPiece pc = pieces[blockIdx.x];
    const uint8_t *src = files + pc.src;
    uint8_t *dst = streams + pc.dst;
    const uint32_t head = (uint32_t)((16 - (pc.dst & 15)) & 15);      // bytes up to the first aligned 16 of the destination
    if (threadIdx.x < head && threadIdx.x < pc.len) dst[threadIdx.x] = src[threadIdx.x];
    if (pc.len <= head) return;
    const uint32_t body = (pc.len - head) >> 4, tail0 = head + (body << 4);
    for (uint32_t k = threadIdx.x; k < body; k += 256) *reinterpret_cast<u32x4 *>(dst + head + 16 * k) = ld16(src + head + 16 * k);
    if (tail0 + threadIdx.x < pc.len) dst[tail0 + threadIdx.x] = src[tail0 + threadIdx.x];
}

// ---- inflate: the three policies of ke_inflate_zlib on the GPU

// The compressed bytes through a 64-byte ring per lane in LDS.  Waiting for a load stalls all 64 lanes, so inside the symbol
// loop the ring is topped up at wave-uniform moments (every 16th symbol): what was asked for 16 symbols ago is written to the
// ring, what has been consumed since is asked for, and nobody waits for memory that is still on its way.  Outside that loop
// (headers) and for a lane that outruns its ring the same two steps run back to back.
struct LdsStream {
    const u32x4 *z;        // global: this image's zlib stream (16-byte aligned)
    uint32_t *win;         // dword j of the ring at win[64 * (j & 15)]
    uint32_t nchunk;       // 16-byte chunks that hold stream data
    uint32_t avail, req;   // dwords landed in the ring / asked for (multiples of 4, avail <= req <= avail + 16)
    uint32_t t;
    u32x4 nx0, nx1, nx2, nx3;
    __device__ __forceinline__ void land1(int c, u32x4 v) {
        if (avail + 4 * c < req) {
            uint32_t *slot = win + 64 * ((avail + 4 * c) & 15);
            slot[0] = v.x; slot[64] = v.y; slot[128] = v.z; slot[192] = v.w;
        }
    }
    __device__ __forceinline__ void land() {
        land1(0, nx0); land1(1, nx1); land1(2, nx2); land1(3, nx3);
        avail = req;
    }
    __device__ __forceinline__ void ask1(uint32_t limit, u32x4 &v) {
        if (req + 4 <= limit) {
            const uint32_t chunk = req >> 2;
            v = chunk < nchunk ? z[chunk] : u32x4{0, 0, 0, 0};
            req += 4;
        }
    }
    __device__ __forceinline__ void ask(uint32_t next) {       // `next`: the reader's next dword; everything before its chunk is free
        const uint32_t limit = (next & ~3u) + 16;
        ask1(limit, nx0); ask1(limit, nx1); ask1(limit, nx2); ask1(limit, nx3);
    }
    __device__ __forceinline__ uint32_t word(uint32_t k) {
        if (k >= avail) {
            land();
            if (k >= avail) { ask(k); land(); }
        }
        return win[64 * (k & 15)];
    }
    __device__ __forceinline__ void tick(uint32_t next) {
        ++t;
        if ((__builtin_amdgcn_readfirstlane(t) & 15) == 0) { land(); ask(next); }
    }
};

// Literals go to their final place in HBM, four at a time; an LZ77 copy is only written down -- (destination, distance,
// length) -- and its bytes are left open: reading the source back here would stall the wave once per match, so ke_png_matches
// fills the copies in afterwards with a whole wave per image.  A dword that straddles the edge of a copy is written with
// zeros on the copy's side; the copy overwrites them later.
struct RecSink {
    uint8_t *p;
    uint32_t n, w;
    uint2 *rec;
    uint32_t nrec;
    int hold;           // streams that must wait with a match before the wave turns to the matches (matches_now)
    __device__ __forceinline__ uint32_t size() const { return n; }
    __device__ __forceinline__ void put(uint8_t b) {
        w |= (uint32_t)b << (8 * (n & 3));
        ++n;
        if ((n & 3) == 0) { *reinterpret_cast<uint32_t *>(p + n - 4) = w; w = 0; }
    }
    __device__ __forceinline__ void settle() {
        if (n & 3) *reinterpret_cast<uint32_t *>(p + (n & ~3u)) = w;
        w = 0;
    }
    __device__ __forceinline__ void finish() { settle(); }
    // ke_inflate_zlib's question once per turn: finish the held matches now?  Yes when `hold` streams wait with one, or all
    // that are still inside the symbol loop (the others would only watch).
    __device__ __forceinline__ bool matches_now(bool waiting) const {
        const int waiters = __popcll(__ballot(waiting)), inside = __popcll(__ballot(true));
        return waiters >= min(hold, inside);
    }
    __device__ __forceinline__ void copy(uint32_t dist, uint32_t len) {
        settle();
        rec[nrec++] = make_uint2(n, (dist << 9) | (len - 3));
        n += len;
    }
};

// eight table words held as named values (arrays indexed in loops end up in scratch memory before they are unrolled)
struct Oct {
    uint32_t a0, a1, a2, a3, a4, a5, a6, a7;
    __device__ __forceinline__ uint32_t get(int k) const {
        // each value through an empty asm: otherwise the selects below are folded into one load at a selected address and
        // the table stays in scratch memory for good
        uint32_t b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;
        asm("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
        uint32_t r = b0;
        r = k == 1 ? b1 : r; r = k == 2 ? b2 : r; r = k == 3 ? b3 : r; r = k == 4 ? b4 : r;
        r = k == 5 ? b5 : r; r = k == 6 ? b6 : r; r = k == 7 ? b7 : r;
        return r;
    }
    __device__ __forceinline__ void set(int k, uint32_t v) {
        a0 = k == 0 ? v : a0; a1 = k == 1 ? v : a1; a2 = k == 2 ? v : a2; a3 = k == 3 ? v : a3;
        a4 = k == 4 ? v : a4; a5 = k == 5 ? v : a5; a6 = k == 6 ? v : a6; a7 = k == 7 ? v : a7;
    }
};

struct LaneTab {           // limits and bases in registers, the symbols in this lane's slices of LDS, a header's code lengths in HBM
    Oct lim0, lim1, base0, base1;
    uint8_t *lsym_;        // low 8 bits of literal/length symbol i at lsym_[64 * i]
    uint32_t *lhigh_;      // bit 8 of symbol i: bit (i & 31) of lhigh_[64 * (i >> 5)]
    uint8_t *dsym_;        // distance symbol i at dsym_[64 * i]
    uint32_t *nib_;
    __device__ __forceinline__ uint32_t lim2(int which, int k) const { return which ? lim1.get(k) : lim0.get(k); }
    __device__ __forceinline__ void set_lim2(int which, int k, uint32_t v) { if (which) lim1.set(k, v); else lim0.set(k, v); }
    __device__ __forceinline__ uint32_t base2(int which, int k) const { return which ? base1.get(k) : base0.get(k); }
    __device__ __forceinline__ void set_base2(int which, int k, uint32_t v) { if (which) base1.set(k, v); else base0.set(k, v); }
    __device__ __forceinline__ uint32_t sym(int which, uint32_t i) const {
        if (which) return dsym_[64 * (i & 31)];
        i = i < 288 ? i : 0;
        return (uint32_t)lsym_[64 * i] | (((lhigh_[64 * (i >> 5)] >> (i & 31)) & 1u) << 8);
    }
    __device__ __forceinline__ void clear_syms(int which) {
        if (!which)
            for (int k = 0; k < 9; ++k) lhigh_[64 * k] = 0;
    }
    __device__ __forceinline__ void set_sym(int which, uint32_t i, uint32_t s) {
        if (which) { dsym_[64 * (i & 31)] = (uint8_t)s; return; }
        i = i < 288 ? i : 0;
        lsym_[64 * i] = (uint8_t)s;
        if (s & 256u) lhigh_[64 * (i >> 5)] |= 1u << (i & 31);
    }
    __device__ __forceinline__ uint32_t nibword(int k) const { return nib_[k]; }
    __device__ __forceinline__ void set_nibword(int k, uint32_t v) { nib_[k] = v; }
};

__global__ __launch_bounds__(64) void ke_png_inflate(const KePngDev *__restrict__ imgs, int64_t n, const uint8_t *__restrict__ streams,
                                                     uint8_t *__restrict__ raw, uint8_t *__restrict__ work, uint2 *__restrict__ records,
                                                     int32_t *__restrict__ status, uint32_t *__restrict__ adler, uint32_t *__restrict__ nrec,
                                                     int lanes, int hold) {
    __shared__ uint8_t s_lsym[288 * 64], s_dsym[32 * 64];
    __shared__ uint32_t s_lhigh[9 * 64], s_win[16 * 64];
    // `lanes` streams per wave (the rest of the wave idles): every path any lane takes is paid for by the whole wave, so while
    // there are SIMDs to spare a batch is spread over more, narrower waves
    const int64_t i = (int64_t)blockIdx.x * lanes + threadIdx.x;
    if ((int)threadIdx.x >= lanes || i >= n) return;
    const KePngDev &d = imgs[i];
    LdsStream src;
    src.z = reinterpret_cast<const u32x4 *>(streams + d.z_off);
    src.win = s_win + threadIdx.x;
    src.nchunk = (d.info.zlen + 15u) >> 4;
    src.avail = src.req = src.t = 0;
    KeBitsLsb<LdsStream> bits{&src, 0, 0, 0};
    RecSink sink{raw + d.raw_off, 0, 0, records + d.rec_off, 0, hold};
    LaneTab tab;
    tab.lsym_ = s_lsym + threadIdx.x;
    tab.lhigh_ = s_lhigh + threadIdx.x;
    tab.dsym_ = s_dsym + threadIdx.x;
    tab.nib_ = reinterpret_cast<uint32_t *>(work + (size_t)i * kWorkBytes);
    tab.lim0 = tab.lim1 = tab.base0 = tab.base1 = Oct{0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t want = d.info.raw_len;
    uint32_t trailer = 0;
    int rc = ke_inflate_zlib(bits, sink, d.info.zlen, want, tab, &trailer);
    if (rc == KE_PNG_OK && sink.n != want) rc = KE_PNG_CORRUPT;
    status[i] = rc;
    adler[i] = trailer;
    nrec[i] = sink.nrec;
}

// ---- the LZ77 copies, one wave per image (ke_lz_copies.h)
__global__ __launch_bounds__(64) void ke_png_matches(const KePngDev *__restrict__ imgs, uint8_t *__restrict__ raw,
                                                     const uint2 *__restrict__ records, const int32_t *__restrict__ status,
                                                     const uint32_t *__restrict__ nrec) {
    const int64_t i = blockIdx.x;
    if (status[i] != KE_PNG_OK) return;
    const KePngDev &d = imgs[i];
    ke_lz_make_copies(raw + d.raw_off, records + d.rec_off, nrec[i], 3u);      // a deflate match is at least 3 bytes long
}

// ---- unfilter

// One pixel of BPP bytes reconstructed (ke_png_recon per byte), without a branch: the predictor of every filter type is
// computed and the row's own picked by masks (m1..m4: all ones for the row's type, Sub / Up / Average / Paeth) -- the lanes of a
// wave hold 64 different rows, so a branch per type would run every arm anyway, once per channel.
template <int BPP>
__device__ __forceinline__ uint32_t recon_pixel(int m1, int m2, int m3, int m4, uint32_t x, uint32_t a, uint32_t b, uint32_t c) {
    uint32_t o = 0;
#pragma unroll
    for (int ch = 0; ch < BPP; ++ch) {
        const int xa = (int)((a >> (8 * ch)) & 255u), xb = (int)((b >> (8 * ch)) & 255u), xc = (int)((c >> (8 * ch)) & 255u);
        const int pa = abs(xb - xc), pb = abs(xa - xc), pc = abs(xa + xb - 2 * xc);
        const int nearest_bc = min(pb, pc), nearest = min(pa, nearest_bc);
        const int paeth = pa == nearest ? xa : (pb == nearest_bc ? xb : xc);      // ties: left, then above, then upper left
        const int pred = (xa & m1) | (xb & m2) | (((xa + xb) >> 1) & m3) | (paeth & m4);
        o |= ((((x >> (8 * ch)) & 255u) + (uint32_t)pred) & 255u) << (8 * ch);
    }
    return o;
}

// MAPPED: palette files and grayscale below 8 bits.  The filters work on the packed bytes (one byte is the filter unit);
// every byte then unpacks to 8 / depth samples, most significant bits first, and each sample goes out as the luma in the
// image's table.
// WIDE: 16-bit samples.  The filters' unit is then 2 * BPP bytes: the loop runs over half pixels of BPP bytes whose left
// neighbour is two places back, and what leaves is 8-bit (ke_png_parse.h: the samples' high bytes; grayscale clipped to 255;
// gray + alpha as RGBA).
template <int BPP, bool MAPPED, bool ADAM7, bool WIDE>
__global__ __launch_bounds__(64) void ke_png_unfilter(const KePngDev *__restrict__ imgs, const uint8_t *__restrict__ raw,
                                                      uint8_t *__restrict__ out, int32_t *__restrict__ status,
                                                      const uint32_t *__restrict__ adler) {
    extern __shared__ uint32_t s_row[];               // the pixels of lane 63's row, for lane 0's next one
    __shared__ uint8_t s_lut[MAPPED ? 256 : 4];
    const int64_t i = blockIdx.x;
    const KePngDev &d = imgs[i];
    if (d.info.fbpp != BPP || (d.info.mapped != 0) != MAPPED || (d.info.interlace != 0) != ADAM7 || (d.info.depth == 16) != WIDE ||
        status[i] != KE_PNG_OK)
        return;
    const int lane = threadIdx.x;
    if (MAPPED) {
        for (int k = lane; k < 256; k += 64) s_lut[k] = d.info.lut[k];
        __syncthreads();
    }
    const uint8_t *src = raw + d.raw_off;
    uint8_t *dst = out + d.out_off;
    const uint64_t total = d.info.raw_len;
    constexpr int kWords = BPP == 1 ? 1 : BPP;        // dwords of four pixels
    uint32_t bad = 0, s1 = 0, s2 = 0;                 // Adler totals mod 65521
    uint64_t pass_off = 0;                            // where this pass's rows start in the stream
    for (int pass = 0; pass < (ADAM7 ? 7 : 1); ++pass) {
        KeAdam7 geo{0, 0, 1, 1, d.info.width, d.info.height};
        int W = d.info.row_bytes / BPP;                   // filter units per row (pixels, or packed bytes)
        if (ADAM7) {
            geo = ke_adam7_pass(pass, d.info.width, d.info.height);
            if (geo.w == 0 || geo.h == 0) continue;
            W = ((geo.w * BPP * (MAPPED ? d.info.depth : WIDE ? 16 : 8) + 7) >> 3) / BPP;
        }
        const int H = geo.h;
        const uint32_t rb = (uint32_t)W * BPP, stride = rb + 1;
        const int groups = (W + 3) >> 2;                  // four pixels at a time
        const int period = groups > 64 ? groups : 64;     // lane 0 starts its next row only after lane 63 has started the one above
        const int rounds = (H + 63) >> 6;

        int row = lane, g = -lane;                        // this step's group; g < 0: not started
        uint32_t ft = 0;
        uint32_t o[4] = {0, 0, 0, 0};                     // the four pixels of the previous step (what lane + 1 sees above)
        uint32_t up3 = 0, up2 = 0;                        // upper-left of the next group: the last pixel(s) above of this one
        uint32_t a1 = 0, c32 = 0;                         // Adler: bytes, offset-weighted bytes of this row
        u32x4 ahead[kWords];                              // the filtered bytes of four groups
        uint32_t behind[4 * kWords];                      // the reconstructed bytes of four groups
        uint64_t b64 = 0;
        const int steps = rounds * period + 63;
        for (int t = 0; t <= steps; ++t) {
            // the row above: what the lane before produced in the previous step (same group index)
            uint32_t up[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) up[k] = (uint32_t)__shfl_up((int)o[k], 1);
            const bool active = g >= 0 && g < groups && row < H;
            if (lane == 0 && active && row > 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) up[k] = s_row[4 * g + k];
            }
            if (row == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) up[k] = 0;
            }
            if (active) {
                const uint8_t *rp = src + pass_off + (size_t)row * stride;
                if (g == 0) {
                    ft = rp[0];
                    bad |= ft > 4;
                    up3 = up2 = 0;
                    o[3] = o[2] = 0;                      // nothing to the left
                    a1 = ft; c32 = 0; b64 = 0;            // the filter byte sits at offset 0
                }
                const int valid = min(4, W - 4 * g);      // pixels of this group
                const uint32_t j0 = 1u + (uint32_t)g * (4 * BPP);
                // The filtered bytes of four groups at a time (16-byte loads): every lane reads its own row, 64 cache lines per
                // load instruction, and with dword loads each of them is fetched from L2 ten times over (the lines of all the
                // waves of a CU do not fit its L1).  A read may run up to 63 bytes past the row: the scratch has that slack.
                if ((g & 3) == 0) {
#pragma unroll
                    for (int q = 0; q < kWords; ++q) ahead[q] = ld16(rp + j0 + 16 * q);
                }
                uint32_t xw[kWords];
#pragma unroll
                for (int q = 0; q < kWords; ++q) {
                    // dword q of group (g & 3) among the 4 * kWords buffered: element (g & 3) * kWords + q
                    uint32_t v = 0;
#pragma unroll
                    for (int part = 0; part < 4; ++part) {
                        const int at = part * kWords + q;
                        const u32x4 from = ahead[at >> 2];
                        const uint32_t w = (at & 3) == 0 ? from.x : (at & 3) == 1 ? from.y : (at & 3) == 2 ? from.z : from.w;
                        v = (g & 3) == part ? w : v;
                    }
                    const int have = valid * BPP - 4 * q;                     // bytes of this dword that belong to the row
                    xw[q] = have >= 4 ? v : have <= 0 ? 0u : (v & ((1u << (8 * have)) - 1u));
                }
                // Adler-32 terms of these bytes: sum and offset-weighted sum
#pragma unroll
                for (int q = 0; q < kWords; ++q) {
                    const uint32_t sum = __builtin_amdgcn_udot4(xw[q], 0x01010101u, 0u, false);
                    a1 += sum;
                    c32 = __builtin_amdgcn_udot4(xw[q], 0x03020100u, c32, false);
                    b64 += (uint64_t)(j0 + 4 * q) * sum;
                }
                uint32_t x[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t v = 0;
#pragma unroll
                    for (int ch = 0; ch < BPP; ++ch) {
                        const int e = k * BPP + ch;
                        v |= ((xw[e >> 2] >> (8 * (e & 3))) & 255u) << (8 * ch);
                    }
                    x[k] = v;
                }
                const int m1 = -(int)(ft == 1), m2 = -(int)(ft == 2), m3 = -(int)(ft == 3), m4 = -(int)(ft >= 4);
                if (WIDE) {
                    const uint32_t l0 = o[2], l1 = o[3];
                    o[0] = recon_pixel<BPP>(m1, m2, m3, m4, x[0], l0, up[0], up2);
                    o[1] = recon_pixel<BPP>(m1, m2, m3, m4, x[1], l1, up[1], up3);
                    o[2] = recon_pixel<BPP>(m1, m2, m3, m4, x[2], o[0], up[2], up[0]);
                    o[3] = recon_pixel<BPP>(m1, m2, m3, m4, x[3], o[1], up[3], up[1]);
                } else {
                    uint32_t left = o[3], ul = up3;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        o[k] = recon_pixel<BPP>(m1, m2, m3, m4, x[k], left, up[k], ul);
                        left = o[k];
                        ul = up[k];
                    }
                }
                up2 = up[2];
                up3 = up[3];
                // the reconstructed bytes, in stream order
                uint32_t ow[kWords];
                if (BPP == 4) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) ow[q % kWords] = o[q];
                } else if (BPP == 3) {
                    ow[0] = o[0] | (o[1] << 24);
                    ow[1 % kWords] = (o[1] >> 8) | (o[2] << 16);
                    ow[2 % kWords] = (o[2] >> 16) | (o[3] << 8);
                } else if (BPP == 2) {
                    ow[0] = o[0] | (o[1] << 16);
                    ow[1 % kWords] = o[2] | (o[3] << 16);
                } else {
                    ow[0] = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
                }
                if (WIDE) {
                    // two pixels per group, 8 bits per sample leave; with or without interlacing (geo: the whole image then)
                    constexpr int kOut = BPP == 1 ? 1 : BPP == 3 ? 3 : 4;
                    uint8_t *wp = dst + (size_t)(geo.y0 + row * geo.dy) * d.info.width * kOut;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        if (2 * q < valid) {
                            uint32_t s[2 * BPP];                  // the pixel's bytes in stream order
#pragma unroll
                            for (int e = 0; e < 2 * BPP; ++e) {
                                const int at = q * 2 * BPP + e;
                                s[e] = (ow[(at >> 2) % kWords] >> (8 * (at & 3))) & 255u;
                            }
                            uint8_t *pp = wp + (size_t)(geo.x0 + (2 * g + q) * geo.dx) * kOut;
                            if (BPP == 1) {
                                pp[0] = (uint8_t)(s[0] ? 255u : s[1]);
                            } else if (BPP == 3) {
                                pp[0] = (uint8_t)s[0]; pp[1] = (uint8_t)s[2 % (2 * BPP)]; pp[2] = (uint8_t)s[4 % (2 * BPP)];
                            } else {
                                const uint32_t v = BPP == 2 ? s[0] * 0x010101u | (s[2 % (2 * BPP)] << 24)
                                                            : s[0] | (s[2 % (2 * BPP)] << 8) | (s[4 % (2 * BPP)] << 16) | (s[6 % (2 * BPP)] << 24);
                                __builtin_memcpy(pp, &v, 4);
                            }
                        }
                    }
                } else if (ADAM7) {
                    // the pass's pixels go to (x0 + k dx, y0 + row dy) of the whole image, one at a time
                    const int px = d.info.width;
                    uint8_t *wp = dst + (size_t)(geo.y0 + row * geo.dy) * px * (MAPPED || BPP == 2 ? 1 : BPP);
                    if (MAPPED) {
                        const int depth = d.info.depth, per = 8 / depth;
                        const uint32_t top = (1u << depth) - 1u;
                        for (int e = 0; e < valid; ++e) {
                            const uint32_t byte = (ow[0] >> (8 * e)) & 255u;
                            const int k0 = (4 * g + e) * per;
                            for (int q = 0; q < per && k0 + q < geo.w; ++q)
                                wp[geo.x0 + (k0 + q) * geo.dx] = s_lut[(byte >> (8 - depth * (q + 1))) & top];
                        }
                    } else {
                        for (int e = 0; e < valid; ++e) {
                            const size_t x = (size_t)(geo.x0 + (4 * g + e) * geo.dx);
                            if (BPP == 2 || BPP == 1) wp[x] = (uint8_t)o[e];
                            else for (int ch = 0; ch < BPP; ++ch) wp[x * BPP + ch] = (uint8_t)(o[e] >> (8 * ch));
                        }
                    }
                } else if (BPP == 2) {
                    // gray + alpha: only the gray samples leave (one byte per pixel, rows of `width` bytes)
                    const uint32_t gray = (o[0] & 255u) | ((o[1] & 255u) << 8) | ((o[2] & 255u) << 16) | ((o[3] & 255u) << 24);
                    uint8_t *wp = dst + (size_t)row * W + (size_t)g * 4;
                    if (valid == 4) __builtin_memcpy(wp, &gray, 4);
                    else for (int e = 0; e < valid; ++e) wp[e] = (uint8_t)(gray >> (8 * e));
                } else if (MAPPED) {
                    const int depth = d.info.depth, per = 8 / depth, px = d.info.width;
                    const uint32_t top = (1u << depth) - 1u;
                    uint8_t *wp = dst + (size_t)row * px;
                    for (int e = 0; e < valid; ++e) {
                        const uint32_t byte = (ow[0] >> (8 * e)) & 255u;
                        const int x0 = (4 * g + e) * per;
                        for (int q = 0; q < per && x0 + q < px; ++q) wp[x0 + q] = s_lut[(byte >> (8 - depth * (q + 1))) & top];
                    }
                } else if (((g | 3) + 1) * 4 <= W) {
                    // one of four whole groups in a row: gathered, written with 16-byte stores when the fourth is done
#pragma unroll
                    for (int part = 0; part < 4; ++part)
#pragma unroll
                        for (int q = 0; q < kWords; ++q) {
                            const int at = part * kWords + q;
                            behind[at] = (g & 3) == part ? ow[q] : behind[at];
                        }
                    if ((g & 3) == 3) {
                        uint8_t *wp = dst + (size_t)row * rb + (size_t)(g - 3) * (4 * BPP);
#pragma unroll
                        for (int q = 0; q < kWords; ++q) st16(wp + 16 * q, u32x4{behind[4 * q], behind[4 * q + 1], behind[4 * q + 2], behind[4 * q + 3]});
                    }
                } else {                                      // the groups at the end of a row that do not fill a block of four
                    uint8_t *wp = dst + (size_t)row * rb + (size_t)g * (4 * BPP);
                    if (valid == 4) {
#pragma unroll
                        for (int q = 0; q < kWords; ++q) __builtin_memcpy(wp + 4 * q, &ow[q], 4);
                    } else {
                        for (int e = 0; e < valid * BPP; ++e) wp[e] = (uint8_t)(ow[e >> 2] >> (8 * (e & 3)));
                    }
                }
                if (lane == 63) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) s_row[4 * g + k] = o[k];
                }
                if (g == groups - 1) {
                    // this row's share of the sums: byte j of the row sits (total - row * stride - j) bytes before the end
                    const uint64_t after = total - pass_off - (uint64_t)row * stride;
                    const uint64_t weighted = b64 + c32;
                    s1 = (s1 + a1 % 65521u) % 65521u;
                    s2 = (uint32_t)((s2 + (after % 65521u) * (uint64_t)(a1 % 65521u) + 65521u - weighted % 65521u) % 65521u);
                }
            }
            ++g;
            if (g == period) { g = 0; row += 64; }
        }
        pass_off += (uint64_t)stride * (uint64_t)H;
    }
    // s1 = 1 + sum of bytes, s2 = total + sum of (distance to the end) * byte   (mod 65521)
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) {
        s1 += (uint32_t)__shfl_xor((int)s1, sh);
        s2 += (uint32_t)__shfl_xor((int)s2, sh);
        bad |= (uint32_t)__shfl_xor((int)bad, sh);
    }
    if (lane == 0) {
        const uint32_t f1 = (1u + s1) % 65521u, f2 = (uint32_t)((total % 65521u + s2) % 65521u);
        if (bad || ((f2 << 16) | f1) != adler[i]) status[i] = KE_PNG_CORRUPT;
    }
}

}  // namespace

KE_API int ke_png_probe(const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n, int32_t *widths,
                        int32_t *heights, int32_t *channels, int32_t *status_out) {
    if (n < 0 || (n > 0 && (!files || !offsets || !sizes || !widths || !heights || !channels || !status_out))) return KE_EINVAL;
    ke_parallel_ranges(n, [=](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; ++i) {
            KePngInfo info;
            ke_parse_png(files + offsets[i], (size_t)sizes[i], nullptr, info);
            widths[i] = info.width; heights[i] = info.height; channels[i] = info.channels;
            status_out[i] = info.status;
        }
    });
    return KE_OK;
}

KE_API int ke_png_caveats(const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n, int32_t *flags_out) {
    if (n < 0 || (n > 0 && (!files || !offsets || !sizes || !flags_out))) return KE_EINVAL;
    ke_parallel_ranges(n, [=](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; ++i) {
            const uint8_t *p = files + offsets[i];
            const size_t size = (size_t)sizes[i];
            int32_t flags = 0;
            size_t pos = 8;
            while (pos + 12 <= size) {
                const size_t len = ((size_t)p[pos] << 24) | ((size_t)p[pos + 1] << 16) | ((size_t)p[pos + 2] << 8) | p[pos + 3];
                const uint8_t *type = p + pos + 4, *data = p + pos + 8;
                if (len > 0x7fffffffu || pos + 12 + len > size || std::memcmp(type, "IEND", 4) == 0) break;
                if (std::memcmp(type, "tRNS", 4) == 0) flags |= KE_CAVEAT_TRANSPARENCY;
                if (std::memcmp(type, "eXIf", 4) == 0) flags |= KE_CAVEAT_ORIENTATION;
                // EXIF blocks that ImageMagick-style writers put into text chunks ("Raw profile type exif" / "... APP1"): Pillow reads them
                if ((std::memcmp(type, "tEXt", 4) == 0 || std::memcmp(type, "zTXt", 4) == 0 || std::memcmp(type, "iTXt", 4) == 0) && len >= 16 &&
                    std::memcmp(data, "Raw profile type", 16) == 0)
                    flags |= KE_CAVEAT_ORIENTATION;
                pos += 12 + len;
            }
            flags_out[i] = flags;
        }
    });
    return KE_OK;
}

KE_API int ke_png_decode(ke_ctx *ctx, const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n,
                         uint8_t *pixels_out, const uint64_t *out_offsets, int32_t *status_out) {
    if (!ctx) return KE_EINVAL;
    if (n < 0 || (n > 0 && (!files || !offsets || !sizes || !pixels_out || !out_offsets || !status_out)))
        return ke_fail(ctx, KE_EINVAL, "NULL argument");
    if (n == 0) return KE_OK;
    if (ke_is_device_ptr(files)) return ke_fail(ctx, KE_EINVAL, "compressed files are parsed on the host: pass host memory (pinned staging is fine)");
    if (!ke_is_device_ptr(pixels_out)) return ke_fail(ctx, KE_EINVAL, "pixels_out must be device memory");
    for (const void *p : {(const void *)offsets, (const void *)sizes, (const void *)out_offsets, (const void *)status_out})
        if (ke_is_device_ptr(p)) return ke_fail(ctx, KE_EINVAL, "offsets/sizes/status are host arrays");
    KE_HIP(ctx, hipSetDevice(ctx->device));
    // ---- host: containers (chunk walk, CRCs of the header chunks), where each image's IDAT payloads lie
    // (on the host's threads, as for JPEG)
    struct Item { KePngDev d; size_t seg0, seg1; int64_t which; };
    struct Part {
        std::vector<Item> items;
        std::vector<KePngSeg> segs;
        uint64_t lo = ~0ull, hi = 0;
    };
    std::vector<Part> parts(16);
    const int nparts = ke_parallel_ranges(n, [&](int64_t first, int64_t last, int t) {
        Part &p = parts[(size_t)t];
        for (int64_t i = first; i < last; ++i) {
            Item it;
            it.seg0 = p.segs.size();
            ke_parse_png(files + offsets[i], (size_t)sizes[i], &p.segs, it.d.info);
            status_out[i] = it.d.info.status;
            if (it.d.info.status != KE_PNG_OK) continue;
            it.seg1 = p.segs.size();
            for (size_t s = it.seg0; s < it.seg1; ++s) p.segs[s].off += offsets[i];      // now relative to `files`
            it.d.out_off = out_offsets[i];
            it.which = i;
            p.lo = std::min(p.lo, offsets[i]);
            p.hi = std::max(p.hi, offsets[i] + sizes[i]);
            p.items.push_back(it);
        }
    });
    std::vector<Item> items;
    std::vector<KePngSeg> segs;
    uint64_t lo = ~0ull, hi = 0;
    for (int t = 0; t < nparts; ++t) {
        Part &p = parts[(size_t)t];
        for (Item &it : p.items) { it.seg0 += segs.size(); it.seg1 += segs.size(); }
        items.insert(items.end(), p.items.begin(), p.items.end());
        segs.insert(segs.end(), p.segs.begin(), p.segs.end());
        lo = std::min(lo, p.lo);
        hi = std::max(hi, p.hi);
    }
    if (items.empty()) return KE_OK;
    // lanes of one wave finish together at best: neighbours in the batch should have streams of like length
    std::stable_sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.d.info.zlen > b.d.info.zlen; });
    // the compressed files (one contiguous range of the caller's buffer) -> device
    void *d_files;
    KE_TRY(ke_reserve(ctx, KE_BUF_PIXELS, (size_t)(hi - lo) + 256, &d_files));
    KE_HIP(ctx, hipMemcpyAsync(d_files, files + lo, (size_t)(hi - lo), hipMemcpyHostToDevice, ctx->stream));
    // sub-batches bounded by scratch (streams + filtered scanlines + copy records)
    size_t free_b = 0, total_b = 0;
    KE_HIP(ctx, hipMemGetInfo(&free_b, &total_b));
    const uint64_t held = (uint64_t)ctx->buf[KE_BUF_SSIM_IN].bytes + ctx->buf[KE_BUF_TMP].bytes + ctx->buf[KE_BUF_SSIM_AUX].bytes;
    const uint64_t budget = std::max<uint64_t>((uint64_t)2 << 30, std::min<uint64_t>((held + (uint64_t)free_b) / 2, (uint64_t)160 << 30));
    std::vector<KePngDev> devs;
    std::vector<KePngPiece> pieces;
    std::vector<int32_t> st;
    size_t first = 0;
    ke_time_begin(ctx, KE_T_JPEG);
    while (first < items.size()) {
        uint64_t zbytes = 0, raw_bytes = 0, nrecs = 0;
        int max_groups = 0;
        size_t last = first;
        devs.clear();
        pieces.clear();
        while (last < items.size()) {
            Item &it = items[last];
            const uint64_t zl = ((uint64_t)it.d.info.zlen + 15) & ~15ull;
            const uint64_t want = it.d.info.raw_len;
            const uint64_t rw = (want + 32 + 15) & ~15ull, rc = want / 3 + 2;      // a copy covers at least 3 bytes
            if (last > first && zbytes + zl + raw_bytes + rw + (nrecs + rc) * 8 > budget) break;
            it.d.info.zoff = 0;
            it.d.z_off = zbytes;
            it.d.raw_off = raw_bytes;
            it.d.rec_off = nrecs;
            nrecs += rc;
            uint64_t at = zbytes;
            for (size_t s = it.seg0; s < it.seg1; ++s)
                for (uint32_t o = 0; o < segs[s].len; o += kPieceBytes) {
                    const uint32_t len = std::min(kPieceBytes, segs[s].len - o);
                    pieces.push_back(KePngPiece{segs[s].off - lo + o, at, len, 0});
                    at += len;
                }
            zbytes += zl;
            raw_bytes += rw;
            max_groups = std::max(max_groups, (it.d.info.row_bytes / (it.d.info.mapped ? 1 : it.d.info.fbpp) + 3) >> 2);
            devs.push_back(it.d);
            ++last;
        }
        const int64_t m = (int64_t)devs.size();
        void *d_streams, *d_imgs, *d_pieces, *d_raw, *d_work, *d_status, *d_adler, *d_rec, *d_nrec;
        KE_TRY(ke_reserve(ctx, KE_BUF_SSIM_AUX, (size_t)nrecs * 8, &d_rec));
        KE_TRY(ke_reserve(ctx, KE_BUF_TILE32, (size_t)m * 4, &d_nrec));
        KE_TRY(ke_reserve(ctx, KE_BUF_SSIM_IN, (size_t)zbytes + 256, &d_streams));      // the stream windows read up to 64 bytes past an image's data
        KE_TRY(ke_reserve(ctx, KE_BUF_META, (size_t)m * sizeof(KePngDev), &d_imgs));
        KE_TRY(ke_reserve(ctx, KE_BUF_JPEG_TABLES, pieces.size() * sizeof(KePngPiece), &d_pieces));
        KE_TRY(ke_reserve(ctx, KE_BUF_TMP, (size_t)raw_bytes + 128, &d_raw));
        KE_TRY(ke_reserve(ctx, KE_BUF_OUT1, (size_t)m * kWorkBytes, &d_work));
        KE_TRY(ke_reserve(ctx, KE_BUF_OUT0, (size_t)m * 4, &d_status));
        KE_TRY(ke_reserve(ctx, KE_BUF_OUT2, (size_t)m * 4, &d_adler));
        KE_HIP(ctx, hipMemcpyAsync(d_imgs, devs.data(), (size_t)m * sizeof(KePngDev), hipMemcpyHostToDevice, ctx->stream));
        KE_HIP(ctx, hipMemcpyAsync(d_pieces, pieces.data(), pieces.size() * sizeof(KePngPiece), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(ke_png_gather, dim3((unsigned)pieces.size()), dim3(256), 0, ctx->stream, (const KePngPiece *)d_pieces,
                           (const uint8_t *)d_files, (uint8_t *)d_streams);
        // 64 streams per wave whatever the batch: spreading a small batch over more, narrower waves (as the JPEG entropy kernels
        // do) makes this kernel slower -- 16 384 textured files 827 -> 1145 ms with 16 lanes per wave, four waves to a CU instead
        // of one.  Not the instruction cache (SQC_ICACHE_MISSES stay at 1e-6 of the requests either way); the narrower waves
        // execute 0.84x the instructions of a full one each, 3.3x as many in total.
        const int lanes = 64;
        static const int hold = [] { const char *e = std::getenv("KE_PNG_HOLD"); const int v = e ? std::atoi(e) : 6; return v < 1 ? 1 : (v > 64 ? 64 : v); }();
        hipLaunchKernelGGL(ke_png_inflate, dim3((unsigned)((m + lanes - 1) / lanes)), dim3(64), 0, ctx->stream, (const KePngDev *)d_imgs, m,
                           (const uint8_t *)d_streams, (uint8_t *)d_raw, (uint8_t *)d_work, (uint2 *)d_rec, (int32_t *)d_status,
                           (uint32_t *)d_adler, (uint32_t *)d_nrec, lanes, hold);
        hipLaunchKernelGGL(ke_png_matches, dim3((unsigned)m), dim3(64), 0, ctx->stream, (const KePngDev *)d_imgs, (uint8_t *)d_raw,
                           (const uint2 *)d_rec, (const int32_t *)d_status, (const uint32_t *)d_nrec);
        const size_t row_lds = (size_t)max_groups * 16;
        // one launch per kind present in the sub-batch (a workgroup whose image is of another kind returns at once)
        // kinds: (bytes per filter unit -- per half of one for 16-bit files --, or palette / sub-byte) x interlaced x 16-bit
        bool kinds[20] = {};
        for (const KePngDev &d : devs)
            kinds[(d.info.mapped ? 4 : d.info.fbpp - 1) + (d.info.interlace ? 5 : 0) + (d.info.depth == 16 ? 10 : 0)] = true;
#define KE_UNFILTER(BPP, MAPPED, ADAM7, WIDE)                                                                                       \
    hipLaunchKernelGGL((ke_png_unfilter<BPP, MAPPED, ADAM7, WIDE>), dim3((unsigned)m), dim3(64), row_lds, ctx->stream,              \
                       (const KePngDev *)d_imgs, (const uint8_t *)d_raw, pixels_out, (int32_t *)d_status, (const uint32_t *)d_adler)
#define KE_UNFILTER_KINDS(AT, ADAM7, WIDE)                \
    if (kinds[AT + 0]) KE_UNFILTER(1, false, ADAM7, WIDE); \
    if (kinds[AT + 1]) KE_UNFILTER(2, false, ADAM7, WIDE); \
    if (kinds[AT + 2]) KE_UNFILTER(3, false, ADAM7, WIDE); \
    if (kinds[AT + 3]) KE_UNFILTER(4, false, ADAM7, WIDE)
        KE_UNFILTER_KINDS(0, false, false);
        if (kinds[4]) KE_UNFILTER(1, true, false, false);
        KE_UNFILTER_KINDS(5, true, false);
        if (kinds[9]) KE_UNFILTER(1, true, true, false);
        KE_UNFILTER_KINDS(10, false, true);
        KE_UNFILTER_KINDS(15, true, true);
#undef KE_UNFILTER_KINDS
#undef KE_UNFILTER
        KE_HIP(ctx, hipGetLastError());
        st.resize((size_t)m);
        KE_HIP(ctx, hipMemcpyAsync(st.data(), d_status, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream));
        KE_HIP(ctx, hipStreamSynchronize(ctx->stream));                  // devs / pieces / st are host vectors; scratch is reused
        for (int64_t k = 0; k < m; ++k) status_out[items[first + (size_t)k].which] = st[(size_t)k];
        first = last;
    }
    ke_time_end(ctx, KE_T_JPEG);
    return KE_OK;
}
