// ke_png_core.h -- PNG decoding arithmetic shared by the HIP kernels (ke_png.hip) and by the CPU build the tests hold against
// Pillow (oracle/keyes_png_cpu.cpp): zlib/deflate decompression (RFC 1950 / 1951) of the concatenated IDAT data and the five
// scanline filters of the PNG specification (None, Sub, Up, Average, Paeth).  Lossless, so "the pixels Pillow yields" is a
// matter of following the two specifications; what is decoded is what `Image.open(path)` hands the reference's batch hasher
// (src/core/fastsig.py:31-34) for grayscale, RGB and RGBA files of 8 or 16 bits, interlaced (Adam7) or not.  Palette files and grayscale files of 1 / 2 / 4 bits are
// decoded to the luma Pillow's `convert("L")` makes of them -- which is what the reference's hashes see (src/sig/phash.py:25):
// samples unpacked most significant bits first, mapped through a 256-entry table (palette entry -> L by ImagingConvert's
// rounded 16-bit weights; 1 / 2 / 4-bit gray scaled to 0..255).
//
// The inflate is written against three small policies so that one body serves both builds:
//   Src   word(k): the k-th little-endian dword of the zlib stream (zeros past its end); tick(k): called once per symbol
//   Sink  put(byte), copy(distance, length), size(), finish()
//   Tab   the decoding tables of the current block (on the GPU: registers and per-lane slices of LDS)
// Symbols are decoded by canonical-code comparison (RFC 1951 3.2.2: codes of one length are consecutive and shorter codes
// sort first when left-aligned, so the length of the code in front of the reader is the number of per-length limits the next
// 15 bits reach): fifteen 16-bit limits and bases per code and the symbols in code order are all the state there is, which is
// what lets a GPU lane keep its tables in registers plus a few hundred bytes of LDS.
#pragma once

#include <stdint.h>

#ifdef __HIPCC__
#define KE_PNG_HD __host__ __device__ __forceinline__
#define KE_PNG_MEMBER __host__ __device__ __forceinline__
#else
#define KE_PNG_HD static inline
#define KE_PNG_MEMBER inline
#endif

enum { KE_PNG_OK = 0, KE_PNG_UNSUPPORTED = 1, KE_PNG_CORRUPT = 2 };
enum { KE_PNG_MAX_WIDTH = 16384 };

struct KePngInfo {
    int32_t status;
    int32_t width, height, channels;     // channels of the decoded pixels: 1 (L), 3 (RGB), 4 (RGBA)
    uint32_t zoff, zlen;                 // the zlib stream (all IDAT payloads, concatenated) inside the staged stream bytes
    int32_t row_bytes;                   // filtered bytes per scanline (without the filter-type byte)
    int32_t depth;                       // bits per sample in the file: 8, 1 / 2 / 4 for the mapped kinds below, or 16 -- then
                                         // the filters' unit is 2 * fbpp bytes and the pixels that leave are 8-bit (ke_png_parse.h)
    int32_t mapped;                      // palette files and grayscale below 8 bits: samples are indices into `lut`, the
    uint8_t lut[256];                    // luma Pillow's convert("L") gives each of them (the reference hashes that)
    int32_t fbpp;                        // bytes per pixel in the file = the filters' unit: `channels`, except gray + alpha (2):
                                         // such a file decodes to its gray samples alone -- what convert("L") makes of mode "LA",
                                         // i.e. what the reference hashes (src/sig/phash.py:25)
    int32_t interlace;                   // 1: Adam7 -- the stream holds seven reduced images one after another (ke_adam7_pass)
    uint32_t raw_len;                    // filtered bytes the stream inflates to: every row of every pass, filter-type bytes included
};

// Adam7 (PNG specification, "Interlace methods"): pass p (0..6) is the image of the pixels (x0 + k dx, y0 + j dy), w x h of
// them, filtered as an image of its own; a pass without pixels has no bytes in the stream at all.
struct KeAdam7 { int x0, y0, dx, dy, w, h; };
KE_PNG_HD KeAdam7 ke_adam7_pass(int p, int width, int height) {
    KeAdam7 a;
    a.x0 = (0x00102040 >> (4 * p)) & 15;     // 0 4 0 2 0 1 0
    a.y0 = (0x01020400 >> (4 * p)) & 15;     // 0 0 4 0 2 0 1
    a.dx = (0x01224488 >> (4 * p)) & 15;     // 8 8 4 4 2 2 1
    a.dy = (0x02244888 >> (4 * p)) & 15;     // 8 8 8 4 4 2 2
    a.w = (width - a.x0 + a.dx - 1) / a.dx;
    a.h = (height - a.y0 + a.dy - 1) / a.dy;
    if (a.w < 0) a.w = 0;
    if (a.h < 0) a.h = 0;
    return a;
}

KE_PNG_HD uint32_t ke_brev32(uint32_t v) {
#ifdef __HIP_DEVICE_COMPILE__
    return __brev(v);
#else
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
#endif
}

// LSB-first bit reader (RFC 1951 3.1.1) over the stream's dwords; after a refill at least 33 bits are in `acc`
template <typename Src>
struct KeBitsLsb {
    Src *src;
    uint32_t wpos;       // next dword to fetch
    uint64_t acc;
    int32_t n;
};

template <typename Src>
KE_PNG_HD void ke_lsb_refill(KeBitsLsb<Src> &b) {
    if (b.n <= 32) {
        b.acc |= (uint64_t)b.src->word(b.wpos++) << b.n;
        b.n += 32;
    }
}

template <typename Src>
KE_PNG_HD uint32_t ke_lsb_take(KeBitsLsb<Src> &b, int k) {      // k <= 16 bits that a refill has made available
    const uint32_t v = (uint32_t)b.acc & ((1u << k) - 1u);
    b.acc >>= k;
    b.n -= k;
    return v;
}

template <typename Src>
KE_PNG_HD uint32_t ke_lsb_bits(KeBitsLsb<Src> &b, int k) {
    ke_lsb_refill(b);
    return ke_lsb_take(b, k);
}

template <typename Src>
KE_PNG_HD uint64_t ke_lsb_consumed(const KeBitsLsb<Src> &b) { return (uint64_t)b.wpos * 32u - (uint64_t)b.n; }

// ---- reference table storage (the CPU build; the GPU keeps the same fields per lane, see ke_png.hip)
struct KeInflateTables {
    uint32_t lim_[2][8];                 // [0] literal/length, [1] distance: limits of lengths 2k, 2k+1 packed in halves
    uint32_t base_[2][8];                // sorted position - first code of each length, int16 halves
    uint16_t lsym_[288];
    uint8_t dsym_[32];
    uint32_t nib_[40];                   // code lengths of a block header, 8 per dword
    uint32_t lim2(int which, int k) const { return lim_[which][k]; }
    void set_lim2(int which, int k, uint32_t v) { lim_[which][k] = v; }
    uint32_t base2(int which, int k) const { return base_[which][k]; }
    void set_base2(int which, int k, uint32_t v) { base_[which][k] = v; }
    uint32_t sym(int which, uint32_t i) const { return which ? dsym_[i & 31] : lsym_[i < 288 ? i : 0]; }
    void clear_syms(int) {}
    void set_sym(int which, uint32_t i, uint32_t s) { if (which) dsym_[i & 31] = (uint8_t)s; else lsym_[i < 288 ? i : 0] = (uint16_t)s; }
    uint32_t nibword(int k) const { return nib_[k]; }
    void set_nibword(int k, uint32_t v) { nib_[k] = v; }
};

// 16-bit halves of the packed per-length arrays
KE_PNG_HD uint32_t ke_half_get(uint32_t word, int l) { return (l & 1) ? word >> 16 : word & 0xFFFFu; }
KE_PNG_HD uint32_t ke_half_set(uint32_t word, int l, uint32_t v) {
    return (l & 1) ? (word & 0xFFFFu) | (v << 16) : (word & 0xFFFF0000u) | (v & 0xFFFFu);
}

// code lengths pushed in order into the table's nibble words
template <typename Tab>
struct KeNibWriter {
    Tab &t;
    uint32_t cur, idx;
    KE_PNG_MEMBER void push(uint32_t v) {
        cur |= v << (4 * (idx & 7));
        if ((++idx & 7) == 0) { t.set_nibword((int)(idx >> 3) - 1, cur); cur = 0; }
    }
    KE_PNG_MEMBER void flush() { if (idx & 7) t.set_nibword((int)(idx >> 3), cur); }
};

// Builds the decoding tables of one code from the lengths at nibbles [nib0, nib0 + n) (RFC 1951 3.2.2).
// Returns 0 for a complete code (or one without any symbol, which decodes nothing), > 0 incomplete, < 0 over-subscribed;
// *unused = number of symbols without a code.
template <typename Tab>
KE_PNG_HD int ke_inflate_build(Tab &t, int which, int n, int nib0, int *unused) {
    // count[length], 9 bits each (n <= 288), in three 64-bit words: no addressable array needed
    uint64_t c0 = 0, c1 = 0, c2 = 0;
    uint32_t w = 0;
    for (int s = 0; s < n; ++s) {
        const int at = nib0 + s;
        if (s == 0 || (at & 7) == 0) w = t.nibword(at >> 3);
        const int l = (int)(w >> (4 * (at & 7))) & 15;
        c0 += l < 7 ? 1ull << (9 * l) : 0ull;
        c1 += (l >= 7 && l < 14) ? 1ull << (9 * (l - 7)) : 0ull;
        c2 += l >= 14 ? 1ull << (9 * (l - 14)) : 0ull;
    }
    auto cnt = [&](int l) -> uint32_t {
        return (uint32_t)((l < 7 ? c0 >> (9 * l) : l < 14 ? c1 >> (9 * (l - 7)) : c2 >> (9 * (l - 14))) & 511u);
    };
    *unused = (int)cnt(0);
    if ((int)cnt(0) == n) {                           // no codes: every limit is reached, nothing decodes
        t.set_lim2(which, 0, 0x00008000u);
        for (int k = 1; k < 8; ++k) t.set_lim2(which, k, 0);
        return 0;
    }
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        left -= (int)cnt(l);
        if (left < 0) return left;
    }
    // the symbols in code order: next position of every length counted up in the base slots
    {
        uint32_t off = 0, bw = 0;
        for (int l = 1; l <= 15; ++l) {
            bw = ke_half_set(bw, l, off);
            if (l & 1) { t.set_base2(which, l >> 1, bw); bw = 0; }
            off += cnt(l);
        }
    }
    t.clear_syms(which);
    for (int s = 0; s < n; ++s) {
        const int at = nib0 + s;
        if (s == 0 || (at & 7) == 0) w = t.nibword(at >> 3);
        const int l = (int)(w >> (4 * (at & 7))) & 15;
        if (l == 0) continue;
        const uint32_t bw = t.base2(which, l >> 1);
        const uint32_t idx = ke_half_get(bw, l);
        t.set_base2(which, l >> 1, ke_half_set(bw, l, idx + 1));
        t.set_sym(which, idx, (uint32_t)s);
    }
    {
        uint32_t code = 0, off = 0, lw = 0x8000u, bw = 0;
        for (int l = 1; l <= 15; ++l) {
            const uint32_t c = cnt(l);
            lw = ke_half_set(lw, l, (code + c) << (15 - l));         // one past the last code of this length, left-aligned
            bw = ke_half_set(bw, l, (off - code) & 0xFFFFu);
            if (l & 1) { t.set_lim2(which, l >> 1, lw); t.set_base2(which, l >> 1, bw); lw = bw = 0; }
            code = (code + c) << 1;
            off += c;
        }
    }
    return left;
}

// The limits and bases of both codes as plain values: what the symbol loop decodes from.  They are taken out of the table
// once per block and handed on one by one -- as values, never as a struct behind a reference: selecting among fields
// of an object in memory is rewritten by the compiler into one load at a computed address, which on the GPU pins the object
// in scratch memory and puts a memory round trip into every symbol.
#define KE_OCT_PARAMS(p) uint32_t p##0, uint32_t p##1, uint32_t p##2, uint32_t p##3, uint32_t p##4, uint32_t p##5, uint32_t p##6, uint32_t p##7
#define KE_OCT_ARGS(p) p##0, p##1, p##2, p##3, p##4, p##5, p##6, p##7
#define KE_OCT_LOAD(p, expr)                                                                                                   \
    const uint32_t p##0 = expr(0), p##1 = expr(1), p##2 = expr(2), p##3 = expr(3), p##4 = expr(4), p##5 = expr(5), p##6 = expr(6), \
                   p##7 = expr(7)

#define KE_LIM0(k) t.lim2(0, k)
#define KE_BASE0(k) t.base2(0, k)
#define KE_LIM1(k) t.lim2(1, k)
#define KE_BASE1(k) t.base2(1, k)

KE_PNG_HD uint32_t ke_oct_get(int k, KE_OCT_PARAMS(a)) {
    uint32_t r = a0;
    r = k == 1 ? a1 : r; r = k == 2 ? a2 : r; r = k == 3 ? a3 : r; r = k == 4 ? a4 : r;
    r = k == 5 ? a5 : r; r = k == 6 ? a6 : r; r = k == 7 ? a7 : r;
    return r;
}

// canonical decode of the code in the low bits of `lo` (LSB-first as read): its length (0 = no such code) and symbol
// (-1 then); `dist` selects between the literal/length code (l, b) and the distance code (m, c) -- per lane on the GPU
template <typename Tab>
KE_PNG_HD int ke_inflate_canon(Tab &t, int dist, uint32_t lo, int *len_out, KE_OCT_PARAMS(l), KE_OCT_PARAMS(b), KE_OCT_PARAMS(m),
                               KE_OCT_PARAMS(c)) {
    const uint32_t v = ke_brev32(lo) >> 17;          // the next 15 bits, first bit on top
    // limits reached, two per step: with bit 15 set in both halves, (v | 0x8000) - limit keeps that bit exactly where
    // v >= limit (limits are <= 0x8000, so no half borrows from the other)
    const uint32_t vv = (v | (v << 16)) | 0x80008000u;
    uint32_t hits = 0;
    hits += ((vv - (dist ? m0 : l0)) >> 15) & 0x00010001u;
    hits += ((vv - (dist ? m1 : l1)) >> 15) & 0x00010001u;
    hits += ((vv - (dist ? m2 : l2)) >> 15) & 0x00010001u;
    hits += ((vv - (dist ? m3 : l3)) >> 15) & 0x00010001u;
    hits += ((vv - (dist ? m4 : l4)) >> 15) & 0x00010001u;
    hits += ((vv - (dist ? m5 : l5)) >> 15) & 0x00010001u;
    hits += ((vv - (dist ? m6 : l6)) >> 15) & 0x00010001u;
    hits += ((vv - (dist ? m7 : l7)) >> 15) & 0x00010001u;
    const int reached = (int)((hits & 0xFFFFu) + (hits >> 16));
    const bool none = reached >= 15;                 // every limit reached: no code starts with these bits
    const int len = none ? 15 : reached + 1;
    const uint32_t bw = dist ? ke_oct_get(len >> 1, KE_OCT_ARGS(c)) : ke_oct_get(len >> 1, KE_OCT_ARGS(b));
    const int base = (int)(int16_t)ke_half_get(bw, len);
    const int sym = (int)t.sym(dist, (uint32_t)(base + (int)(v >> (15 - len))));
    *len_out = none ? 0 : len;
    return none ? -1 : sym;
}

// Inflates one zlib stream of `zlen` bytes into at most `limit` output bytes.  The Adler-32 trailer is handed back in
// *adler_out for the caller to hold against the output (the GPU does that where the output is read next, in parallel).
// Returns KE_PNG_OK or KE_PNG_CORRUPT.
template <typename Src, typename Sink, typename Tab>
KE_PNG_HD int ke_inflate_zlib(KeBitsLsb<Src> &b, Sink &out, uint32_t zlen, uint32_t limit, Tab &t, uint32_t *adler_out) {
    const uint64_t zbits = (uint64_t)zlen * 8u;
    // zlib header (RFC 1950): deflate, window <= 32K, no preset dictionary, check bits
    const uint32_t cmf = ke_lsb_bits(b, 8), flg = ke_lsb_bits(b, 8);
    if ((cmf & 15) != 8 || (cmf >> 4) > 7 || (flg & 32) || ((cmf << 8) | flg) % 31 != 0) return KE_PNG_CORRUPT;
    int last;
    do {
        if (ke_lsb_consumed(b) > zbits) return KE_PNG_CORRUPT;           // ran off the stream (only zeros come from there)
        last = (int)ke_lsb_bits(b, 1);
        const int type = (int)ke_lsb_bits(b, 2);
        if (type == 0) {                              // stored
            ke_lsb_take(b, b.n & 7);                  // to the byte boundary
            const uint32_t len = ke_lsb_bits(b, 16), nlen = ke_lsb_bits(b, 16);
            if ((len ^ 0xFFFFu) != nlen || out.size() + len > limit) return KE_PNG_CORRUPT;
            for (uint32_t k = 0; k < len; ++k) out.put((uint8_t)ke_lsb_bits(b, 8));
            continue;
        }
        if (type == 3) return KE_PNG_CORRUPT;
        int nlen, ndist;
        KeNibWriter<Tab> nw{t, 0, 0};
        if (type == 1) {                              // fixed codes (3.2.6)
            nlen = 288; ndist = 30;
            for (int s = 0; s < 144; ++s) nw.push(8);
            for (int s = 144; s < 256; ++s) nw.push(9);
            for (int s = 256; s < 280; ++s) nw.push(7);
            for (int s = 280; s < 288; ++s) nw.push(8);
            for (int s = 0; s < 30; ++s) nw.push(5);
            nw.flush();
        } else {                                      // dynamic codes (3.2.7)
            nlen = (int)ke_lsb_bits(b, 5) + 257;
            ndist = (int)ke_lsb_bits(b, 5) + 1;
            const int ncode = (int)ke_lsb_bits(b, 4) + 4;
            if (nlen > 286 || ndist > 30) return KE_PNG_CORRUPT;
            // the code length code: up to 19 lengths of 3 bits, sent in the order 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4,
            // 12, 3, 13, 2, 14, 1, 15; gathered as nibbles of a 64-bit + a 32-bit word
            uint64_t cl_lo = 0;
            uint32_t cl_hi = 0;
            for (int k = 0; k < ncode; ++k) {
                const int pos = k < 3 ? 16 + k : k == 3 ? 0 : (k & 1) ? 8 - ((k - 3) >> 1) : 8 + ((k - 4) >> 1);
                const uint32_t v = ke_lsb_bits(b, 3);
                if (pos < 16) cl_lo |= (uint64_t)v << (4 * pos); else cl_hi |= v << (4 * (pos - 16));
            }
            t.set_nibword(0, (uint32_t)cl_lo);
            t.set_nibword(1, (uint32_t)(cl_lo >> 32));
            t.set_nibword(2, cl_hi);
            int unused;
            if (ke_inflate_build(t, 1, 19, 0, &unused) != 0) return KE_PNG_CORRUPT;   // the code length code must be complete
            // the literal/length and distance code lengths, run-length coded with that code (it sits in the distance slots,
            // which the block's own distance code replaces below)
            KE_OCT_LOAD(q, KE_LIM1);
            KE_OCT_LOAD(r, KE_BASE1);
            uint32_t prev = 0;
            int idx = 0;
            while (idx < nlen + ndist) {
                ke_lsb_refill(b);
                int cl;
                const int sym = ke_inflate_canon(t, 1, (uint32_t)b.acc, &cl, KE_OCT_ARGS(q), KE_OCT_ARGS(r), KE_OCT_ARGS(q), KE_OCT_ARGS(r));
                if (cl == 0) return KE_PNG_CORRUPT;
                ke_lsb_take(b, cl);
                if (sym < 16) {
                    nw.push((uint32_t)sym);
                    prev = (uint32_t)sym;
                    ++idx;
                } else {
                    uint32_t len = 0;
                    int rep;
                    if (sym == 16) {
                        if (idx == 0) return KE_PNG_CORRUPT;
                        len = prev;
                        rep = 3 + (int)ke_lsb_bits(b, 2);
                    } else if (sym == 17) {
                        rep = 3 + (int)ke_lsb_bits(b, 3);
                    } else {
                        rep = 11 + (int)ke_lsb_bits(b, 7);
                    }
                    if (idx + rep > nlen + ndist) return KE_PNG_CORRUPT;
                    idx += rep;
                    while (rep--) nw.push(len);
                    prev = len;
                }
                if (ke_lsb_consumed(b) > zbits) return KE_PNG_CORRUPT;
            }
            nw.flush();
            if (((t.nibword(256 >> 3) >> (4 * (256 & 7))) & 15u) == 0) return KE_PNG_CORRUPT;      // no end-of-block code
        }
        // literal/length: an incomplete code is only allowed when it has a single code (zlib's rule); same for distances
        // (the fixed distance code, 30 of 32 five-bit codes, is incomplete by definition)
        int unused;
        int err = ke_inflate_build(t, 0, nlen, 0, &unused);
        if (type == 2 && (err < 0 || (err > 0 && nlen - unused != 1))) return KE_PNG_CORRUPT;
        err = ke_inflate_build(t, 1, ndist, nlen, &unused);
        if (type == 2 && (err < 0 || (err > 0 && ndist - unused != 1))) return KE_PNG_CORRUPT;
        // the block's symbols.  Written for lock-step execution (on the GPU all 64 lanes of a wave pay for every path any of
        // them takes, every symbol): one way out of the loop, damage remembered instead of returned mid-way.
        KE_OCT_LOAD(l, KE_LIM0);
        KE_OCT_LOAD(g, KE_BASE0);
        KE_OCT_LOAD(m, KE_LIM1);
        KE_OCT_LOAD(h, KE_BASE1);
        uint32_t bad = 0;
        int held = 0;                                 // a length symbol whose distance and copy wait for their turn (0 = none)
        for (;;) {
            b.src->tick(b.wpos);
            int sym = 0;
            if (!held) {
                ke_lsb_refill(b);                     // >= 33 bits: a code (<= 15) and its extra bits (<= 13)
                int cl;
                sym = ke_inflate_canon(t, 0, (uint32_t)b.acc, &cl, KE_OCT_ARGS(l), KE_OCT_ARGS(g), KE_OCT_ARGS(m), KE_OCT_ARGS(h));
                ke_lsb_take(b, cl);
                bad |= (uint32_t)(cl == 0);           // no such code (sym is -1 then)
                if ((uint32_t)sym < 256u) {
                    if (out.size() < limit) out.put((uint8_t)sym); else bad = 1;
                } else if (sym > 256) {
                    held = sym;
                }
            }
            // The second half of a match costs as much as a literal; on the GPU the whole wave pays for it in every turn in
            // which one of its 64 streams has a match.  So a stream holds its length symbol back (and sits idle) until the
            // sink says that enough streams are waiting: the turns in between are literals only.  Per stream nothing changes.
            if (out.matches_now(held != 0) && held) {
                // length 3..258 (3.2.5): eight codes of one length each, then groups of four per extra bit, then 258 itself
                const int ls = held - 257;
                held = 0;
                const int lx = (ls < 8 || ls >= 28) ? 0 : (ls - 4) >> 2;
                const uint32_t len = (ls < 8 ? 3u + (uint32_t)ls : ls >= 28 ? 258u : 3u + ((4u + ((uint32_t)ls & 3u)) << lx)) + ke_lsb_take(b, lx);
                ke_lsb_refill(b);
                int dl;
                const int ds = ke_inflate_canon(t, 1, (uint32_t)b.acc, &dl, KE_OCT_ARGS(l), KE_OCT_ARGS(g), KE_OCT_ARGS(m), KE_OCT_ARGS(h));
                ke_lsb_take(b, dl);
                // distance 1..32768: four codes of one distance each, then pairs per extra bit
                const int dx = ds < 4 ? 0 : ((ds - 2) >> 1) & 15;
                const uint32_t dist = (ds < 4 ? 1u + (uint32_t)(ds & 3) : 1u + ((2u + ((uint32_t)ds & 1u)) << dx)) + ke_lsb_take(b, dx);
                if (ls >= 29 || dl == 0 || ds >= 30 || dist > out.size() || out.size() + len > limit) bad = 1; else out.copy(dist, len);
            }
            if (sym == 256 || bad) break;
        }
        if (bad) return KE_PNG_CORRUPT;
    } while (!last);
    out.finish();
    // Adler-32 of the output, big-endian, at the next byte boundary
    ke_lsb_take(b, b.n & 7);
    uint32_t adler = 0;
    for (int k = 0; k < 4; ++k) adler = (adler << 8) | ke_lsb_bits(b, 8);
    if (ke_lsb_consumed(b) > zbits) return KE_PNG_CORRUPT;
    *adler_out = adler;
    return KE_PNG_OK;
}

// Adler-32 (RFC 1950) of `n` bytes, sequentially (the CPU build; the GPU sums it by rows, see ke_png.hip)
KE_PNG_HD uint32_t ke_adler32(const uint8_t *p, uint64_t n) {
    uint32_t s1 = 1, s2 = 0;
    while (n > 0) {
        const uint32_t run = n < 5552 ? (uint32_t)n : 5552u;
        for (uint32_t k = 0; k < run; ++k) { s1 += p[k]; s2 += s1; }
        s1 %= 65521u; s2 %= 65521u;
        p += run;
        n -= run;
    }
    return (s2 << 16) | s1;
}

// ---- scanline filters (PNG specification, "Filter algorithms"): x the filtered byte, a left, b above, c above-left
KE_PNG_HD int ke_paeth(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

KE_PNG_HD int ke_png_recon(int type, int x, int a, int b, int c) {
    const int pred = type == 0 ? 0 : type == 1 ? a : type == 2 ? b : type == 3 ? (a + b) >> 1 : ke_paeth(a, b, c);
    return (x + pred) & 255;
}

// row of `rb` bytes, bpp bytes per pixel; raw: the filtered bytes (after the filter-type byte); prev: the reconstructed
// previous row or NULL for the first; out: result
KE_PNG_HD int ke_png_unfilter_row(int type, const uint8_t *raw, const uint8_t *prev, uint8_t *out, int rb, int bpp) {
    if (type > 4) return KE_PNG_CORRUPT;
    for (int i = 0; i < rb; ++i) {
        const int a = i >= bpp ? out[i - bpp] : 0;
        const int b = prev ? prev[i] : 0;
        const int c = (prev && i >= bpp) ? prev[i - bpp] : 0;
        out[i] = (uint8_t)ke_png_recon(type, raw[i], a, b, c);
    }
    return KE_PNG_OK;
}
