// ke_png_parse.h -- host-side PNG container parsing for the GPU decoder (and the CPU build the tests hold against Pillow):
// signature, IHDR, where the IDAT payloads lie (their concatenation is one zlib stream per image), IEND; chunk CRCs are
// verified where Pillow's ChunkStream verifies them (a damaged file raises there and the reference drops it,
// src/core/fastsig.py:36-37).  8-bit grayscale, RGB
// and RGBA are taken, interlaced (Adam7) or not, 16-bit files as the 8-bit pixels Pillow opens them to (palette / sub-byte
// gray / gray + alpha files decode to the luma convert("L") gives them), and of an animated PNG frame 0, which is what Image.open shows.
#pragma once

#include <cstring>
#include <vector>

#include "ke_png_core.h"

struct KeCrcTables {
    uint32_t t[4][256];
    KeCrcTables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int k = 1; k < 4; ++k) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xFF];
    }
};

static inline const uint32_t *ke_crc_tables() {
    static const KeCrcTables tables;                 // initialised once, also when several parsing threads get here together
    return &tables.t[0][0];
}

static inline uint32_t ke_crc32(uint32_t crc, const uint8_t *p, size_t n) {
    const uint32_t(*t)[256] = reinterpret_cast<const uint32_t(*)[256]>(ke_crc_tables());
    crc = ~crc;
    while (n >= 4) {
        uint32_t w;
        std::memcpy(&w, p, 4);
        crc ^= w;
        crc = t[3][crc & 0xFF] ^ t[2][(crc >> 8) & 0xFF] ^ t[1][(crc >> 16) & 0xFF] ^ t[0][crc >> 24];
        p += 4;
        n -= 4;
    }
    while (n--) crc = t[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
    return ~crc;
}

struct KePngSeg { uint64_t off; uint32_t len; };   // one IDAT payload: offset from the start of the file, length

// Walks the container: IHDR into `info`, the IDAT payloads appended to `segs` in file order (their concatenation is the
// image's zlib stream; info.zlen = its length).  segs == nullptr: headers only.  CRCs are verified for every chunk except
// IDAT -- as Pillow does (PngImagePlugin.ChunkStream.crc for the chunks it parses, while PngImageFile.load_read steps over the
// CRC of image data); damage inside IDAT is caught by the zlib stream's own structure and Adler-32.
static inline void ke_parse_png(const uint8_t *p, size_t size, std::vector<KePngSeg> *segs, KePngInfo &info) {
    std::memset(&info, 0, sizeof info);
    info.status = KE_PNG_CORRUPT;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (size < 8 + 25 || std::memcmp(p, sig, 8) != 0) return;
    auto be32 = [&](size_t o) { return ((uint32_t)p[o] << 24) | ((uint32_t)p[o + 1] << 16) | ((uint32_t)p[o + 2] << 8) | p[o + 3]; };
    size_t pos = 8;
    bool have_ihdr = false, ended = false, palette = false, have_plte = false, idat_closed = false;
    // animated PNG: Image.open shows frame 0, which is the IDAT image (the "default image", or the first frame when an fcTL chunk
    // stands in front of it -- its region must then be the whole image, or Pillow decodes the stream into that region only).
    // PngImageFile.load_end stops at the next frame's fcTL without looking at anything behind it.
    bool have_actl = false, have_fctl = false;
    uint32_t actl_frames = 0;
    const size_t s0 = segs ? segs->size() : 0;
    uint64_t zlen = 0;
    auto animated = [&]() { return have_actl && (uint64_t)actl_frames + (have_fctl ? 0 : 1) > 1; };
    while (pos + 12 <= size || (pos + 8 <= size && zlen != 0 && animated())) {
        if (segs && zlen != 0 && animated() && std::memcmp(p + pos + 4, "IDAT", 4) != 0) {
            // the first chunk behind the image data of an animation: the next frame's fcTL (Pillow stops there, whatever
            // follows) or IEND; any other chunk there is handled by code paths this parser does not mirror
            if (std::memcmp(p + pos + 4, "fcTL", 4) == 0 || std::memcmp(p + pos + 4, "IEND", 4) == 0) ended = true;
            else { info.status = KE_PNG_UNSUPPORTED; segs->resize(s0); return; }
            break;
        }
        if (pos + 12 > size) break;
        const uint32_t len = be32(pos);
        if (len > 0x7fffffffu || pos + 12 + (size_t)len > size) break;
        const uint8_t *type = p + pos + 4, *data = p + pos + 8;
        const bool idat = std::memcmp(type, "IDAT", 4) == 0;
        if (segs && !idat && ke_crc32(0, type, 4 + (size_t)len) != be32(pos + 8 + len)) break;      // Pillow: "broken PNG file"
        if (!have_ihdr) {
            if (std::memcmp(type, "IHDR", 4) != 0 || len != 13) return;
            info.width = (int32_t)be32(pos + 8);
            info.height = (int32_t)be32(pos + 12);
            const int depth = data[8], ctype = data[9], comp = data[10], filt = data[11], lace = data[12];
            if (info.width <= 0 || info.height <= 0 || comp != 0 || filt != 0) return;
            info.channels = (ctype == 0 || ctype == 3 || ctype == 4) ? 1 : ctype == 2 ? 3 : ctype == 6 ? 4 : 0;
            info.fbpp = ctype == 4 ? 2 : info.channels;
            const bool sub8 = (ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4);
            // 16 bits per sample (not for palettes): Pillow opens such a file to 8-bit pixels -- RGB / RGBA from the samples' high
            // bytes (raw modes "RGB;16B", "RGBA;16B"), gray + alpha to RGBA (L, L, L, A) the same way ("LA;16B"), gray to mode
            // "I;16", which convert("L") and convert("RGB") clip to 255 (PngImagePlugin._MODES; ImagingConvert I;16 -> L)
            const bool wide = depth == 16 && ctype != 3;
            if (wide && ctype == 4) info.channels = 4;
            if (!(depth == 8 || sub8 || wide) || info.channels == 0 || lace > 1 || info.width > (wide ? KE_PNG_MAX_WIDTH / 2 : KE_PNG_MAX_WIDTH) ||
                (uint64_t)info.width * info.height > (1ull << 28)) {
                info.status = KE_PNG_UNSUPPORTED;
                if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16) || ctype > 6 || ctype == 1 || ctype == 5 || lace > 1)
                    info.status = KE_PNG_CORRUPT;
                return;
            }
            info.depth = depth;
            info.mapped = (ctype == 3 || sub8) ? 1 : 0;
            info.row_bytes = (int32_t)(((int64_t)info.width * info.fbpp * depth + 7) / 8);
            info.interlace = lace;
            uint64_t raw_len = ((uint64_t)info.row_bytes + 1) * (uint64_t)info.height;
            if (lace) {
                raw_len = 0;
                for (int pass = 0; pass < 7; ++pass) {
                    const KeAdam7 a = ke_adam7_pass(pass, info.width, info.height);
                    if (a.w > 0 && a.h > 0) raw_len += (((uint64_t)a.w * info.fbpp * depth + 7) / 8 + 1) * (uint64_t)a.h;
                }
            }
            info.raw_len = (uint32_t)raw_len;            // at most 2^28 pixels of 4 bytes + a filter byte per row
            palette = ctype == 3;
            if (info.mapped && !palette)                      // 1 / 2 / 4-bit gray: Pillow's "1", "L;2", "L;4" unpackers scale to 0..255
                for (int v = 0; v < (1 << depth); ++v) info.lut[v] = (uint8_t)(v * 255 / ((1 << depth) - 1));
            have_ihdr = true;
            if (!segs) { info.status = KE_PNG_OK; return; }
        } else if (std::memcmp(type, "PLTE", 4) == 0) {
            if (len % 3 != 0 || len > 768 || zlen != 0) break;               // Pillow: "broken PNG file" / palette after the image data
            if (palette) {
                // what convert("L") makes of a palette entry: ImagingConvert's L24, (19595 R + 38470 G + 7471 B + 0x8000) >> 16;
                // indices beyond the palette read as black
                for (uint32_t k = 0; k < len / 3; ++k)
                    info.lut[k] = (uint8_t)((data[3 * k] * 19595u + data[3 * k + 1] * 38470u + data[3 * k + 2] * 7471u + 0x8000u) >> 16);
                have_plte = true;
            }
        } else if (idat) {
            if (palette && !have_plte) break;
            if (idat_closed) break;                    // IDAT, another chunk, IDAT: Pillow's load_read ends the stream at the first
                                                       // non-IDAT chunk ("image file is truncated"), libpng rejects the file too
            if (len) segs->push_back(KePngSeg{(uint64_t)(pos + 8), len});
            zlen += len;
        } else if (std::memcmp(type, "IEND", 4) == 0) {
            ended = true;
            break;
        } else if (std::memcmp(type, "acTL", 4) == 0 || std::memcmp(type, "fcTL", 4) == 0 || std::memcmp(type, "fdAT", 4) == 0) {
            // in front of the image data: one acTL with a sane frame count, at most one fcTL (sequence number 0, the whole image);
            // anything else about these chunks -- repeated, malformed, fdAT before IDAT, any of them behind the image data of a
            // file that is not animated -- is Pillow's to judge
            bool fine = zlen == 0;
            if (fine && type[0] == 'a') {
                fine = !have_actl && !have_fctl && len == 8 && be32(pos + 8) != 0 && be32(pos + 8) <= 0x7FFFFFFFu;
                have_actl = true;
                actl_frames = fine ? be32(pos + 8) : 0;
            } else if (fine && type[1] == 'c') {
                fine = !have_fctl && len == 26 && be32(pos + 8) == 0 && be32(pos + 12) == (uint32_t)info.width &&
                       be32(pos + 16) == (uint32_t)info.height && be32(pos + 20) == 0 && be32(pos + 24) == 0;
                have_fctl = true;
            } else {
                fine = false;
            }
            if (!fine) {
                info.status = KE_PNG_UNSUPPORTED;
                segs->resize(s0);
                return;
            }
        }
        if (!idat && zlen != 0) idat_closed = true;
        pos += 12 + (size_t)len;
    }
    if (zlen != 0 && animated() && pos + 8 > size) ended = true;      // an animation whose file ends behind frame 0's data: Pillow stops reading too
    if (!have_ihdr || zlen == 0 || zlen > 0xF0000000ull || !ended) {   // no IEND: truncated (Pillow raises unless LOAD_TRUNCATED_IMAGES)
        if (segs) segs->resize(s0);
        return;
    }
    info.zlen = (uint32_t)zlen;
    info.status = KE_PNG_OK;
}
