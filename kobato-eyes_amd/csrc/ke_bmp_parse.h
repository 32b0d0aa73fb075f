// ke_bmp_parse.h -- host-side BMP header parsing for the GPU unpacker (ke_bmp.hip) and the CPU build the tests hold against
// Pillow (oracle/keyes_bmp_cpu.cpp).  Replaces `Image.open(path)` + pixel access of the reference's batch hasher
// (src/core/fastsig.py:31-34) for the uncompressed files Pillow's BmpImagePlugin opens with its "raw" decoder: 24-bit BGR,
// 32-bit BGRX, the 32-bit BITFIELDS layouts that are byte permutations (with or without alpha), and 8-bit palette files -- the
// last as the luma `convert("L")` makes of them, which is what the reference's hashes see (src/sig/phash.py:25).  The header
// is read the way BmpImageFile._bitmap reads it (field positions, the defaults for `colors` and for the data offset, rows of
// ((width * bits + 31) >> 3) & ~3 bytes, bottom-up unless the height's top byte is 0xFF).  Everything else -- OS/2 headers,
// RLE, 1 / 4 / 16 bits, other masks, odd palette sizes -- is KE_BMP_UNSUPPORTED and stays with Pillow; a file whose pixel data
// ends early is KE_BMP_CORRUPT (Pillow: "image file is truncated", the reference drops the file, src/core/fastsig.py:36-37).
#pragma once

#include <stdint.h>

#include <cstring>

enum { KE_BMP_OK = 0, KE_BMP_UNSUPPORTED = 1, KE_BMP_CORRUPT = 2 };
enum { KE_BMP_MAX_WIDTH = 65535 };

struct KeBmpInfo {
    int32_t status;
    int32_t width, height, channels;     // channels of the pixels that leave: 1 (luma of a palette file), 3 (RGB), 4 (RGBA)
    int32_t bits;                        // 8, 24 or 32 per pixel in the file
    int32_t topdown;                     // rows stored top to bottom (otherwise the first stored row is the bottom one)
    uint32_t data_off, stride;           // pixel data from the start of the file; bytes per stored row
    uint8_t pick[4];                     // 24 / 32 bits: the byte of a stored pixel that gives R, G, B, A
    uint8_t lut[256];                    // 8 bits: palette index -> luma (ImagingConvert's rounded weights, as for PNG palettes)
};

static inline void ke_parse_bmp(const uint8_t *p, size_t size, KeBmpInfo &info) {
    std::memset(&info, 0, sizeof info);
    info.status = KE_BMP_CORRUPT;
    if (size < 18 || p[0] != 'B' || p[1] != 'M') return;                       // Pillow: "Not a BMP file"
    auto le32 = [&](size_t o) { return (uint32_t)p[o] | ((uint32_t)p[o + 1] << 8) | ((uint32_t)p[o + 2] << 16) | ((uint32_t)p[o + 3] << 24); };
    auto le16 = [&](size_t o) { return (uint32_t)p[o] | ((uint32_t)p[o + 1] << 8); };
    info.status = KE_BMP_UNSUPPORTED;
    uint64_t offset = le32(10);
    const uint32_t hs = le32(14);
    if (!(hs == 40 || hs == 52 || hs == 56 || hs == 64 || hs == 108 || hs == 124) || 14 + (size_t)hs > size) return;
    const size_t hd = 18;                                                       // header_data: the header without its size field
    const bool flip = p[hd + 7] == 0xFF;
    const uint64_t width = le32(hd), height = flip ? (1ull << 32) - le32(hd + 4) : le32(hd + 4);
    const uint32_t bits = le16(hd + 10), comp = le32(hd + 12);
    uint64_t colors = le32(hd + 28);
    if (width == 0 || height == 0 || width > KE_BMP_MAX_WIDTH || height > (1u << 28) || width * height > (1ull << 28)) return;
    size_t pos = 14 + (size_t)hs;
    uint32_t mask[4] = {0, 0, 0, 0};
    if (comp == 3) {                                                            // BITFIELDS
        if (hs >= 52) {
            for (int k = 0; k < 3; ++k) mask[k] = le32(hd + 36 + 4 * (size_t)k);
            if (hs >= 56) mask[3] = le32(hd + 48);
        } else {
            if (pos + 12 > size) return;
            for (int k = 0; k < 3; ++k) mask[k] = le32(pos + 4 * (size_t)k);
            pos += 12;
        }
    }
    if (colors == 0) colors = 1ull << bits;
    if (offset == 14 + (uint64_t)hs && bits <= 8) offset += 4 * colors;
    if (!(bits == 8 || bits == 24 || bits == 32)) return;
    info.channels = bits == 8 ? 1 : 3;
    info.pick[0] = 2; info.pick[1] = 1; info.pick[2] = 0; info.pick[3] = 3;    // BGR / BGRX
    if (comp == 3) {
        if (bits == 32) {
            // the layouts BmpImagePlugin lists (MASK_MODES): byte permutations of R, G, B and a fourth byte that is alpha or unused
            static const struct { uint32_t m[4]; uint8_t pick[4]; int alpha; } known[8] = {
                {{0xFF0000u, 0xFF00u, 0xFFu, 0x0u}, {2, 1, 0, 3}, 0},           // BGRX
                {{0xFF000000u, 0xFF0000u, 0xFF00u, 0x0u}, {3, 2, 1, 0}, 0},     // XBGR
                {{0xFF000000u, 0xFF00u, 0xFFu, 0x0u}, {3, 1, 0, 2}, 0},         // BGXR
                {{0xFF000000u, 0xFF0000u, 0xFF00u, 0xFFu}, {3, 2, 1, 0}, 1},    // ABGR
                {{0xFFu, 0xFF00u, 0xFF0000u, 0xFF000000u}, {0, 1, 2, 3}, 1},    // RGBA
                {{0xFF0000u, 0xFF00u, 0xFFu, 0xFF000000u}, {2, 1, 0, 3}, 1},    // BGRA
                {{0xFF000000u, 0xFF00u, 0xFFu, 0xFF0000u}, {3, 1, 0, 2}, 1},    // BGAR
                {{0x0u, 0x0u, 0x0u, 0x0u}, {2, 1, 0, 3}, 1},                    // BGRA
            };
            int found = -1;
            for (int k = 0; k < 8 && found < 0; ++k)
                if (std::memcmp(known[k].m, mask, sizeof mask) == 0) found = k;
            if (found < 0) return;
            std::memcpy(info.pick, known[found].pick, 4);
            info.channels = known[found].alpha ? 4 : 3;
        } else if (!(bits == 24 && mask[0] == 0xFF0000u && mask[1] == 0xFF00u && mask[2] == 0xFFu)) {
            return;
        }
    } else if (comp != 0) {
        return;                                                                 // RLE and the rest
    }
    if (bits == 8) {
        // Pillow reads `colors` entries of B, G, R, X behind the header.  If entry k is (k, k, k) for every one of them the file
        // is mode "L" with the stored bytes as samples -- whatever their value, also beyond a short palette; otherwise mode "P",
        // where an index beyond the palette is black.  Two colours are another case in the plugin (mode "1"): left to it.
        if (colors == 2 || colors > 256 || pos + 4 * (size_t)colors > size) return;
        bool gray = true;
        for (uint64_t k = 0; k < colors; ++k) {
            const uint8_t *e = p + pos + 4 * (size_t)k;
            info.lut[k] = (uint8_t)((e[2] * 19595u + e[1] * 38470u + e[0] * 7471u + 0x8000u) >> 16);
            gray = gray && e[0] == k && e[1] == k && e[2] == k;
        }
        if (gray)
            for (int k = 0; k < 256; ++k) info.lut[k] = (uint8_t)k;
        pos += 4 * (size_t)colors;
    }
    if (offset == 0) offset = pos;                                              // Pillow: `offset or self.fp.tell()`
    if (offset > 0xFFFFFFFFull) return;
    info.width = (int32_t)width;
    info.height = (int32_t)height;
    info.bits = (int32_t)bits;
    info.topdown = flip ? 1 : 0;
    info.stride = (uint32_t)((((uint64_t)width * bits + 31) >> 3) & ~3ull);
    if (offset > size || (uint64_t)info.stride * height > size - offset) {      // the raw decoder wants every row whole
        info.status = KE_BMP_CORRUPT;
        return;
    }
    info.data_off = (uint32_t)offset;
    info.status = KE_BMP_OK;
}
