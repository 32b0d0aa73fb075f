// ke_gif_core.h -- GIF: the container walk and the LZW arithmetic shared by the HIP kernels (ke_gif.hip) and the CPU build the
// tests hold against Pillow (oracle/keyes_gif_cpu.cpp).  Replaces `Image.open(path)` + pixel access of the reference's batch
// hasher (src/core/fastsig.py:31-34) for GIF files -- a keeper format of src/dup/scanner.py:16-28 -- as far as the first frame,
// which is what Image.open shows (an animation's later frames are never looked at by the hashes).  What leaves is the luma
// `convert("L")` makes of the frame (src/sig/phash.py:25): palette index -> ImagingConvert's rounded weights of the palette
// entry (the local palette if the frame has one, else the global one; no palette or an identity gray ramp: the index itself).
//
// The container is walked the way GifImagePlugin._open / _seek(0) walk it (extensions skipped block by block, the graphic
// control extension's transparency index noted -- it changes nothing for a frame that covers the screen, every pixel is
// written -- the first image descriptor taken).  The LZW stream is read the way Pillow's GifDecode.c reads it: sub-blocks are
// taken only when they are whole, codes grow when `next == codemask` at the moment an entry is added, a code beyond `next` is
// damage, `next` itself is the previous string plus its own first character, and decoding is over when the last pixel is written
// -- no end code needed, nothing behind it looked at.  Everything else is refused and stays with Pillow: a first frame that
// does not cover the logical screen exactly, code sizes outside 2..8, an end code / block terminator / end of file before the
// last pixel (Pillow then reads on into whatever follows, or raises), unknown block tags.
//
// A dictionary entry is kept as (where its string was last written, its length): the string of a new entry is the previous
// string plus the first character of the current one, and those bytes lie next to each other in the output -- so expanding a
// code is a copy from earlier output (possibly overlapping its own destination by one character), exactly the LZ77 copies of
// the PNG path, and the same kernel makes them (ke_png_matches).
#pragma once

#include <stdint.h>

#ifdef __HIPCC__
#define KE_GIF_HD __host__ __device__ __forceinline__
#else
#define KE_GIF_HD static inline
#endif

enum { KE_GIF_OK = 0, KE_GIF_UNSUPPORTED = 1, KE_GIF_CORRUPT = 2 };
enum { KE_GIF_MAX_PIXELS = 1 << 23 };      // a copy's distance has 23 bits in the record shared with the PNG path

struct KeGifInfo {
    int32_t status;
    int32_t width, height, channels;     // channels = 1: luma
    int32_t bits;                        // LZW minimum code size
    int32_t interlace;
    uint32_t data_off;                   // the first sub-block's size byte, from the start of the file
    uint8_t lut[256];                    // index -> luma
};

#include <cstring>

static inline void ke_parse_gif(const uint8_t *p, size_t size, KeGifInfo &info) {
    std::memset(&info, 0, sizeof info);
    info.status = KE_GIF_CORRUPT;
    if (size < 13 || (std::memcmp(p, "GIF87a", 6) != 0 && std::memcmp(p, "GIF89a", 6) != 0)) return;      // Pillow: "not a GIF file"
    info.status = KE_GIF_UNSUPPORTED;
    const int W = p[6] | (p[7] << 8), H = p[8] | (p[9] << 8);
    if (W == 0 || H == 0 || (int64_t)W * H > KE_GIF_MAX_PIXELS) return;
    size_t pos = 13;
    const uint8_t *global = nullptr, *local = nullptr;
    int global_n = 0, local_n = 0;
    if (p[10] & 128) {
        global_n = 2 << (p[10] & 7);
        if (pos + 3 * (size_t)global_n > size) return;
        global = p + pos;
        pos += 3 * (size_t)global_n;
    }
    // GifImageFile.data(): a size byte, then that many bytes; size 0 or the end of the file: nothing.  (A block the file ends in
    // is read short by Pillow; such files are refused here.)
    bool bad = false;
    auto data = [&](const uint8_t *&blk, size_t &n) {
        if (pos >= size) return false;
        n = p[pos++];
        if (n == 0) return false;
        if (pos + n > size) { bad = true; return false; }
        blk = p + pos;
        pos += n;
        return true;
    };
    for (;;) {
        if (pos >= size) return;
        const uint8_t tag = p[pos++];
        if (tag == 0x3B) return;                                   // trailer before any image: Pillow has nothing to show
        if (tag == 0x21) {
            // _seek's order of reads, to the letter: the label, one block, for a NETSCAPE2.0 application extension one more,
            // then blocks until one is empty -- also when the first was (what follows a lone terminator is then eaten as blocks);
            // a comment extension ends at its own first empty block
            if (pos >= size) return;
            const uint8_t label = p[pos++];
            const uint8_t *blk = nullptr;
            size_t n = 0;
            bool have = data(blk, n);
            if (label == 0xFE) {
                while (have) have = data(blk, n);
            } else {
                if (label == 0xF9 && have && n < 4) return;        // Pillow indexes the block's first four bytes
                if (label == 0xFF && have && n >= 11 && std::memcmp(blk, "NETSCAPE2.0", 11) == 0) data(blk, n);
                while (data(blk, n)) {}
            }
            if (bad) return;
        } else if (tag == 0x2C) {
            if (pos + 9 > size) return;
            const int x0 = p[pos] | (p[pos + 1] << 8), y0 = p[pos + 2] | (p[pos + 3] << 8);
            const int w = p[pos + 4] | (p[pos + 5] << 8), h = p[pos + 6] | (p[pos + 7] << 8);
            const uint8_t flags = p[pos + 8];
            pos += 9;
            if (x0 != 0 || y0 != 0 || w != W || h != H) return;    // a frame inside (or beyond) the screen: Pillow's canvas rules
            info.interlace = (flags & 64) ? 1 : 0;
            if (flags & 128) {
                local_n = 2 << (flags & 7);
                if (pos + 3 * (size_t)local_n > size) return;
                local = p + pos;
                pos += 3 * (size_t)local_n;
            }
            if (pos + 1 > size) return;
            info.bits = p[pos++];
            if (info.bits < 2 || info.bits > 8) return;
            break;
        } else {
            return;                                                // not a tag of the format (Pillow steps over such bytes)
        }
    }
    // the frame's palette: local if present, else global; an identity gray ramp counts as none (GifImageFile._is_palette_needed)
    const uint8_t *pal = local ? local : global;
    const int pal_n = local ? local_n : global_n;
    for (int k = 0; k < 256; ++k) info.lut[k] = (uint8_t)k;
    bool needed = false;
    for (int k = 0; k < pal_n; ++k) needed = needed || !(pal[3 * k] == k && pal[3 * k + 1] == k && pal[3 * k + 2] == k);
    if (needed) {
        for (int k = 0; k < 256; ++k) {
            // indices beyond the palette are black (the palette is padded with zeros)
            info.lut[k] = k < pal_n ? (uint8_t)((pal[3 * k] * 19595u + pal[3 * k + 1] * 38470u + pal[3 * k + 2] * 7471u + 0x8000u) >> 16) : 0;
        }
    }
    info.width = W;
    info.height = H;
    info.channels = 1;
    info.data_off = (uint32_t)pos;
    info.status = KE_GIF_OK;
}

// The stored row a decoded row goes to: rows come as they are, or in the four passes of an interlaced frame (every 8th from
// 0, every 8th from 4, every 4th from 2, every 2nd from 1).
KE_GIF_HD int ke_gif_row(int k, int height, int interlace) {
    if (!interlace) return k;
    const int n0 = (height + 7) >> 3, n1 = (height + 3) >> 3, n2 = (height + 1) >> 2;
    if (k < n0) return 8 * k;
    k -= n0;
    if (k < n1) return 8 * k + 4;
    k -= n1;
    if (k < n2) return 4 * k + 2;
    k -= n2;
    return 2 * k + 1;
}

// Src: byte(pos) of the file, asked for ascending positions.  Dict: set(code, pos, len), get(code, pos, len).  Sink: literal(byte), copy(from, len) -- both append;
// `want` pixels end the decoding wherever they fall.
template <typename Src, typename Dict, typename Sink>
KE_GIF_HD int ke_gif_lzw(Src &src, uint32_t pos, uint32_t size, int bits, uint32_t want, Dict &dict, Sink &sink) {
    const uint32_t clear = 1u << bits, end = clear + 1;
    uint32_t next = clear + 2, codesize = (uint32_t)bits + 1, codemask = (1u << codesize) - 1;
    bool fresh = true;                   // the next code is the first after a clear code (or the first of all)
    uint32_t bitbuf = 0, bitcount = 0, blocksize = 0;
    uint32_t out = 0, last_pos = 0, last_len = 0;
    while (out < want) {
        while (bitcount < codesize) {
            if (blocksize > 0) {
                bitbuf |= (uint32_t)src.byte(pos++) << bitcount;
                bitcount += 8;
                --blocksize;
            } else {
                if (pos >= size) return KE_GIF_CORRUPT;            // Pillow: "image file is truncated"
                const uint32_t n = src.byte(pos);
                if (pos + 1 + n > size) return KE_GIF_CORRUPT;     // a sub-block is taken whole or not at all
                if (n == 0) return KE_GIF_UNSUPPORTED;             // the list ends before the image does
                blocksize = n;
                ++pos;
            }
        }
        const uint32_t c = bitbuf & codemask;
        bitbuf >>= codesize;
        bitcount -= codesize;
        if (c == clear) {
            next = clear + 2;
            codesize = (uint32_t)bits + 1;
            codemask = (1u << codesize) - 1;
            fresh = true;
            continue;
        }
        if (c == end) return KE_GIF_UNSUPPORTED;
        const uint32_t at = out;
        uint32_t len;
        if (fresh) {
            if (c > clear) return KE_GIF_CORRUPT;
            sink.literal((uint8_t)c);
            len = 1;
            fresh = false;
        } else {
            if (c > next) return KE_GIF_CORRUPT;
            if (c < clear) {
                sink.literal((uint8_t)c);
                len = 1;
            } else {
                uint32_t from = last_pos;
                len = last_len + 1;
                if (c != next) dict.get(c, from, len);
                const uint32_t take = len < want - out ? len : want - out;
                sink.copy(from, take);
            }
            if (next < 4096) {
                dict.set(next, last_pos, last_len + 1);
                if (next == codemask && codesize < 12) {
                    ++codesize;
                    codemask = (1u << codesize) - 1;
                }
                ++next;
            }
        }
        last_pos = at;
        last_len = len;
        out += len;
    }
    return KE_GIF_OK;
}
