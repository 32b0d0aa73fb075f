// ke_png_core.h -- PNG decoding arithmetic shared by the HIP kernels (ke_png.hip) and by the CPU build the tests hold against
// Pillow (oracle/keyes_png_cpu.cpp): zlib/deflate decompression (RFC 1950 / 1951) of the concatenated IDAT data and the five
// scanline filters of the PNG specification (None, Sub, Up, Average, Paeth).  Lossless, so "the pixels Pillow yields" is a
// matter of following the two specifications; what is decoded is what `Image.open(path)` hands the reference's batch hasher
// (src/core/fastsig.py:31-34) for 8-bit grayscale, RGB and RGBA files without interlacing.  Everything else (palette, 16-bit,
// 1/2/4-bit, gray+alpha, Adam7) is refused by the parser and stays with Pillow.
#pragma once

#include <stdint.h>

#ifdef __HIPCC__
#define KE_PNG_HD __host__ __device__ __forceinline__
#else
#define KE_PNG_HD static inline
#endif

enum { KE_PNG_OK = 0, KE_PNG_UNSUPPORTED = 1, KE_PNG_CORRUPT = 2 };

// Canonical Huffman decoding tables of one deflate block (RFC 1951 3.2.2): count[len] codes of each length, symbols sorted by
// (length, value).  One pair (literal/length, distance) per decoding thread.
struct KeInflateTables {
    uint16_t lcount[16];
    uint16_t lsym[288];
    uint16_t dcount[16];
    uint16_t dsym[32];
};

struct KePngInfo {
    int32_t status;
    int32_t width, height, channels;     // channels 1 (L), 3 (RGB), 4 (RGBA)
    uint32_t zoff, zlen;                 // the zlib stream (all IDAT payloads, concatenated by the host) inside the staged bytes
};

// LSB-first bit reader over bytes fetched through `Src::byte(pos)` (RFC 1951 3.1.1)
template <typename Src>
struct KeBitsLsb {
    Src *src;
    uint32_t pos, end;
    uint64_t acc;
    int32_t n;
    int32_t overrun;
};

template <typename Src>
KE_PNG_HD void ke_lsb_fill(KeBitsLsb<Src> &b) {
    while (b.n <= 56) {
        uint32_t byte = 0;
        if (b.pos < b.end) byte = b.src->byte(b.pos++); else b.overrun += 1;
        b.acc |= (uint64_t)byte << b.n;
        b.n += 8;
    }
}

template <typename Src>
KE_PNG_HD uint32_t ke_lsb_bits(KeBitsLsb<Src> &b, int k) {      // k <= 16
    ke_lsb_fill(b);
    const uint32_t v = (uint32_t)(b.acc & ((1u << k) - 1u));
    b.acc >>= k;
    b.n -= k;
    return v;
}

// one symbol with the canonical tables (the decode of zlib's puff.c, which follows the RFC directly); -1 = invalid code
template <typename Src>
KE_PNG_HD int ke_inflate_symbol(KeBitsLsb<Src> &b, const uint16_t *count, const uint16_t *sym) {
    ke_lsb_fill(b);
    int code = 0, first = 0, index = 0;
    uint32_t bits = (uint32_t)b.acc;
    for (int len = 1; len <= 15; ++len) {
        code |= (int)(bits & 1u);
        bits >>= 1;
        const int cnt = count[len];
        if (code - cnt < first) {
            b.acc >>= len;
            b.n -= len;
            return sym[index + (code - first)];
        }
        index += cnt;
        first += cnt;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

// count[] / sym[] from code lengths (RFC 1951 3.2.2); returns 0 for a complete code, > 0 incomplete, < 0 over-subscribed
KE_PNG_HD int ke_inflate_build(const uint8_t *lengths, int n, uint16_t *count, uint16_t *sym) {
    for (int l = 0; l < 16; ++l) count[l] = 0;
    for (int s = 0; s < n; ++s) count[lengths[s]] += 1;
    if (count[0] == n) return 0;                      // no codes: complete, but unusable
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return left;
    }
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    for (int s = 0; s < n; ++s)
        if (lengths[s] != 0) sym[offs[lengths[s]]++] = (uint16_t)s;
    return left;
}

// Sink: out.put(byte), out.get(distance) (a byte written `distance` positions back), out.size() bytes written so far.
// Inflates one zlib stream of at most `limit` output bytes; returns KE_PNG_OK or KE_PNG_CORRUPT.  `work`: 320 bytes of
// scratch for the code lengths of a dynamic block.
template <typename Src, typename Sink>
KE_PNG_HD int ke_inflate_zlib(KeBitsLsb<Src> &b, Sink &out, uint32_t limit, KeInflateTables &t, uint8_t *work) {
    const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097,
                                6145, 8193, 12289, 16385, 24577};
    const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    // zlib header (RFC 1950): deflate, window <= 32K, no preset dictionary, check bits
    const uint32_t cmf = ke_lsb_bits(b, 8), flg = ke_lsb_bits(b, 8);
    if ((cmf & 15) != 8 || (cmf >> 4) > 7 || (flg & 32) || ((cmf << 8) | flg) % 31 != 0) return KE_PNG_CORRUPT;
    uint32_t s1 = 1, s2 = 0, pending = 0;             // Adler-32 of the output
    int last;
    do {
        last = (int)ke_lsb_bits(b, 1);
        const int type = (int)ke_lsb_bits(b, 2);
        if (type == 0) {                              // stored
            const int drop = b.n & 7;                 // to the byte boundary
            b.acc >>= drop;
            b.n -= drop;
            const uint32_t len = ke_lsb_bits(b, 16), nlen = ke_lsb_bits(b, 16);
            if ((len ^ 0xFFFFu) != nlen || out.size() + len > limit) return KE_PNG_CORRUPT;
            for (uint32_t k = 0; k < len; ++k) {
                const uint32_t v = ke_lsb_bits(b, 8);
                out.put((uint8_t)v);
                s1 += v; s2 += s1;
                if (++pending == 5552) { s1 %= 65521u; s2 %= 65521u; pending = 0; }
            }
            continue;
        }
        if (type == 3) return KE_PNG_CORRUPT;
        if (type == 1) {                              // fixed codes (3.2.6)
            for (int s = 0; s < 144; ++s) work[s] = 8;
            for (int s = 144; s < 256; ++s) work[s] = 9;
            for (int s = 256; s < 280; ++s) work[s] = 7;
            for (int s = 280; s < 288; ++s) work[s] = 8;
            ke_inflate_build(work, 288, t.lcount, t.lsym);
            for (int s = 0; s < 30; ++s) work[s] = 5;
            ke_inflate_build(work, 30, t.dcount, t.dsym);
        } else {                                      // dynamic codes (3.2.7)
            const int nlen = (int)ke_lsb_bits(b, 5) + 257, ndist = (int)ke_lsb_bits(b, 5) + 1, ncode = (int)ke_lsb_bits(b, 4) + 4;
            if (nlen > 286 || ndist > 30) return KE_PNG_CORRUPT;
            for (int k = 0; k < 19; ++k) work[k] = 0;
            for (int k = 0; k < ncode; ++k) work[order[k]] = (uint8_t)ke_lsb_bits(b, 3);
            if (ke_inflate_build(work, 19, t.lcount, t.lsym) != 0) return KE_PNG_CORRUPT;   // the code length code must be complete
            int idx = 0;
            while (idx < nlen + ndist) {
                int sym = ke_inflate_symbol(b, t.lcount, t.lsym);
                if (sym < 0) return KE_PNG_CORRUPT;
                if (sym < 16) {
                    work[32 + idx++] = (uint8_t)sym;
                } else {
                    int len = 0, rep;
                    if (sym == 16) {
                        if (idx == 0) return KE_PNG_CORRUPT;
                        len = work[32 + idx - 1];
                        rep = 3 + (int)ke_lsb_bits(b, 2);
                    } else if (sym == 17) {
                        rep = 3 + (int)ke_lsb_bits(b, 3);
                    } else {
                        rep = 11 + (int)ke_lsb_bits(b, 7);
                    }
                    if (idx + rep > nlen + ndist) return KE_PNG_CORRUPT;
                    while (rep--) work[32 + idx++] = (uint8_t)len;
                }
            }
            if (work[32 + 256] == 0) return KE_PNG_CORRUPT;               // no end-of-block code
            // literal/length: incomplete codes are only allowed with a single code (zlib's rule); same for distances
            int err = ke_inflate_build(work + 32, nlen, t.lcount, t.lsym);
            if (err < 0 || (err > 0 && nlen - t.lcount[0] != 1)) return KE_PNG_CORRUPT;
            err = ke_inflate_build(work + 32 + nlen, ndist, t.dcount, t.dsym);
            if (err < 0 || (err > 0 && ndist - t.dcount[0] != 1)) return KE_PNG_CORRUPT;
        }
        for (;;) {                                    // the block's symbols
            int sym = ke_inflate_symbol(b, t.lcount, t.lsym);
            if (sym < 0) return KE_PNG_CORRUPT;
            if (sym < 256) {
                if (out.size() >= limit) return KE_PNG_CORRUPT;
                out.put((uint8_t)sym);
                s1 += (uint32_t)sym; s2 += s1;
                if (++pending == 5552) { s1 %= 65521u; s2 %= 65521u; pending = 0; }
                continue;
            }
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) return KE_PNG_CORRUPT;
            const uint32_t len = lbase[sym] + ke_lsb_bits(b, lext[sym]);
            const int ds = ke_inflate_symbol(b, t.dcount, t.dsym);
            if (ds < 0 || ds >= 30) return KE_PNG_CORRUPT;
            const uint32_t dist = dbase[ds] + ke_lsb_bits(b, dext[ds]);
            if (dist > out.size() || out.size() + len > limit) return KE_PNG_CORRUPT;
            for (uint32_t k = 0; k < len; ++k) {
                const uint32_t v = out.get(dist);
                out.put((uint8_t)v);
                s1 += v; s2 += s1;
                if (++pending == 5552) { s1 %= 65521u; s2 %= 65521u; pending = 0; }
            }
        }
    } while (!last);
    // Adler-32 trailer, big-endian, at the next byte boundary
    const int drop = b.n & 7;
    b.acc >>= drop;
    b.n -= drop;
    s1 %= 65521u; s2 %= 65521u;
    uint32_t adler = 0;
    for (int k = 0; k < 4; ++k) adler = (adler << 8) | ke_lsb_bits(b, 8);
    if (b.overrun > 8) return KE_PNG_CORRUPT;         // the filler's look-ahead past the data is fine, reading real bits there is not
    if (adler != ((s2 << 16) | s1)) return KE_PNG_CORRUPT;
    return KE_PNG_OK;
}

// ---- scanline filters (PNG specification, "Filter algorithms"): row of `rb` bytes, bpp bytes per pixel;
// raw: the filtered bytes (after the filter-type byte); prev: the reconstructed previous row or NULL for the first; out: result
KE_PNG_HD int ke_paeth(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

KE_PNG_HD int ke_png_unfilter_row(int type, const uint8_t *raw, const uint8_t *prev, uint8_t *out, int rb, int bpp) {
    if (type > 4) return KE_PNG_CORRUPT;
    for (int i = 0; i < rb; ++i) {
        const int a = i >= bpp ? out[i - bpp] : 0;
        const int b = prev ? prev[i] : 0;
        const int c = (prev && i >= bpp) ? prev[i - bpp] : 0;
        int v = raw[i];
        if (type == 1) v += a;
        else if (type == 2) v += b;
        else if (type == 3) v += (a + b) >> 1;
        else if (type == 4) v += ke_paeth(a, b, c);
        out[i] = (uint8_t)v;
    }
    return KE_PNG_OK;
}
