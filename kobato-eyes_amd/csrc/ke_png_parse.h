// ke_png_parse.h -- host-side PNG container parsing for the GPU decoder (and the CPU build the tests hold against Pillow):
// signature, IHDR, the IDAT payloads (concatenated into one zlib stream per image), IEND; chunk CRCs are verified as Pillow's
// ChunkStream does (a damaged file raises there and the reference drops it, src/core/fastsig.py:36-37).  8-bit grayscale, RGB
// and RGBA without interlacing are taken; palette, 16-bit, sub-byte depths, gray+alpha and Adam7 files are KE_PNG_UNSUPPORTED
// and stay with Pillow.
#pragma once

#include <cstring>
#include <vector>

#include "ke_png_core.h"

static inline const uint32_t *ke_crc_tables() {
    static uint32_t t[4][256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int k = 1; k < 4; ++k) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xFF];
        ready = true;
    }
    return &t[0][0];
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

// Appends the image's zlib stream to `stream` (info.zoff / zlen refer to it).  stream == nullptr: headers only.
static inline void ke_parse_png(const uint8_t *p, size_t size, std::vector<uint8_t> *stream, KePngInfo &info) {
    std::memset(&info, 0, sizeof info);
    info.status = KE_PNG_CORRUPT;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (size < 8 + 25 || std::memcmp(p, sig, 8) != 0) return;
    auto be32 = [&](size_t o) { return ((uint32_t)p[o] << 24) | ((uint32_t)p[o + 1] << 16) | ((uint32_t)p[o + 2] << 8) | p[o + 3]; };
    size_t pos = 8;
    bool have_ihdr = false, have_idat = false, ended = false;
    const size_t z0 = stream ? stream->size() : 0;
    while (pos + 12 <= size) {
        const uint32_t len = be32(pos);
        if (len > 0x7fffffffu || pos + 12 + (size_t)len > size) return;
        const uint8_t *type = p + pos + 4, *data = p + pos + 8;
        if (stream && ke_crc32(0, type, 4 + (size_t)len) != be32(pos + 8 + len)) return;      // Pillow: "broken PNG file"
        if (!have_ihdr) {
            if (std::memcmp(type, "IHDR", 4) != 0 || len != 13) return;
            info.width = (int32_t)be32(pos + 8);
            info.height = (int32_t)be32(pos + 12);
            const int depth = data[8], ctype = data[9], comp = data[10], filt = data[11], lace = data[12];
            if (info.width <= 0 || info.height <= 0 || comp != 0 || filt != 0) return;
            info.channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 6 ? 4 : 0;
            if (depth != 8 || info.channels == 0 || lace != 0 || (uint64_t)info.width * info.height > (1ull << 28)) {
                info.status = KE_PNG_UNSUPPORTED;
                if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16) || ctype > 6 || ctype == 1 || ctype == 5 || lace > 1)
                    info.status = KE_PNG_CORRUPT;
                return;
            }
            have_ihdr = true;
            if (!stream) { info.status = KE_PNG_OK; return; }
        } else if (std::memcmp(type, "IDAT", 4) == 0) {
            stream->insert(stream->end(), data, data + len);
            have_idat = true;
        } else if (std::memcmp(type, "IEND", 4) == 0) {
            ended = true;
            break;
        } else if (std::memcmp(type, "acTL", 4) == 0) {
            info.status = KE_PNG_UNSUPPORTED;          // animated PNG: Pillow's frame handling decides what is seen
            stream->resize(z0);
            return;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || !have_idat || !ended) {          // no IEND: truncated (Pillow raises unless LOAD_TRUNCATED_IMAGES)
        if (stream) stream->resize(z0);
        return;
    }
    info.zoff = (uint32_t)z0;
    info.zlen = (uint32_t)(stream->size() - z0);
    while (stream->size() % 4) stream->push_back(0);    // the next image's stream starts on a dword
    info.status = KE_PNG_OK;
}
