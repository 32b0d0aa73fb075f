// ke_tiff_parse.h -- host-side TIFF directory parsing for the GPU unpacker (ke_tiff.hip) and the CPU build the tests hold
// against Pillow (oracle/keyes_tiff_cpu.cpp).  Replaces `Image.open(path)` + pixel access of the reference's batch hasher
// (src/core/fastsig.py:31-34) for the uncompressed 8-bit files Pillow's TiffImagePlugin opens with its own "raw" decoder (every
// compressed TIFF goes through libtiff there and stays with Pillow here): grayscale (BlackIsZero and WhiteIsZero), RGB, RGB
// with an unassociated alpha or an unspecified fourth sample, palette files -- the last as the luma `convert("L")` makes of
// them (src/sig/phash.py:25).  The first directory is the image (Image.open shows frame 0).  The directory is read as
// ImageFileDirectory_v2.load reads it (both byte orders; a value that does not fit its four bytes lies at an offset; an entry
// of unknown type or without data is skipped; at an entry whose data cannot be read whole the reading stops; a later entry of
// a tag replaces an earlier one) and
// the pixel layout is derived as TiffImageFile._setup derives it (defaults, the BitsPerSample fix-ups, strips of RowsPerStrip
// rows, "every strip covers the image: only the last offset counts").  What is taken is a whitelist: any tag of the ones
// _setup / load_end look at that is not in its plain form -- other compression, planar layout, fill order, orientation other
// than 1, sample formats, tiles, EXIF / GPS sub-directories or an XMP packet (both can carry an orientation that Pillow
// applies at load time), a strip count that does not match the rows -- makes the file KE_TIFF_UNSUPPORTED and leaves it to
// Pillow; a strip that ends behind the file is KE_TIFF_CORRUPT (Pillow: "image file is truncated").
#pragma once

#include <stdint.h>

#include <cstring>
#include <vector>

enum { KE_TIFF_OK = 0, KE_TIFF_UNSUPPORTED = 1, KE_TIFF_CORRUPT = 2 };

struct KeTiffInfo {
    int32_t status;
    int32_t width, height, channels;     // channels of the pixels that leave: 1 (gray, or the luma of a palette file), 3, 4
    int32_t spp;                         // samples per pixel in the file: 1, 3 or 4
    int32_t mapped;                      // 1: every sample goes through `lut` (WhiteIsZero: inverted; palette: luma)
    int32_t rows_per_strip, nstrips;     // strip s holds rows s * rows_per_strip ...; its offset is strips[first + s]
    uint8_t lut[256];
};

// strips: the offsets of the image's strips are appended (nothing is appended unless the status is KE_TIFF_OK)
static inline void ke_parse_tiff(const uint8_t *p, size_t size, std::vector<uint32_t> *strips, KeTiffInfo &info) {
    std::memset(&info, 0, sizeof info);
    info.status = KE_TIFF_CORRUPT;
    if (size < 8) return;
    const bool le = p[0] == 'I' && p[1] == 'I', be = p[0] == 'M' && p[1] == 'M';
    if (!le && !be) return;
    auto rd16 = [&](size_t o) { return le ? (uint32_t)p[o] | ((uint32_t)p[o + 1] << 8) : ((uint32_t)p[o] << 8) | p[o + 1]; };
    auto rd32 = [&](size_t o) {
        return le ? (uint32_t)p[o] | ((uint32_t)p[o + 1] << 8) | ((uint32_t)p[o + 2] << 16) | ((uint32_t)p[o + 3] << 24)
                  : ((uint32_t)p[o] << 24) | ((uint32_t)p[o + 1] << 16) | ((uint32_t)p[o + 2] << 8) | p[o + 3];
    };
    const uint32_t magic = rd16(2);
    if (magic != 42 && magic != 43) return;                        // Pillow: not identified
    info.status = KE_TIFF_UNSUPPORTED;
    if (magic == 43) return;                                       // BigTIFF
    const size_t ifd = rd32(4);
    if (ifd < 8 || ifd + 2 > size) return;
    const size_t n = rd16(ifd);
    if (ifd + 2 + 12 * n + 4 > size) return;                       // the entries and the pointer to the next directory
    struct Tag { int type; uint32_t count; size_t at; bool have; };
    enum { T_WIDTH, T_LENGTH, T_BITS, T_COMP, T_PHOTO, T_FILL, T_STRIPS, T_ORIENT, T_SPP, T_ROWS, T_XRES, T_YRES, T_PLANAR, T_UNIT,
           T_COLORMAP, T_TILES, T_EXTRA, T_FORMAT, T_XMP, T_EXIF, T_GPS, T_INTEROP, T_WMP, T_COUNT };
    static const uint32_t ids[T_COUNT] = {256, 257, 258, 259, 262, 266, 273, 274, 277, 278, 282, 283, 284, 296, 320, 324, 338, 339, 700,
                                          34665, 34853, 40965, 0xBC01};
    Tag tags[T_COUNT];
    for (Tag &t : tags) t = Tag{0, 0, 0, false};
    static const int unit[17] = {0, 1, 1, 2, 4, 8, 1, 1, 2, 4, 8, 4, 8, 4, 0, 0, 8};
    for (size_t k = 0; k < n; ++k) {
        const size_t e = ifd + 2 + 12 * k;
        const uint32_t id = rd16(e), type = rd16(e + 2), count = rd32(e + 4);
        if (type > 16 || unit[type] == 0) continue;                // unknown type: the entry is skipped
        const uint64_t bytes = (uint64_t)count * unit[type];
        if (bytes == 0) continue;                                  // no data: skipped (an earlier entry of the tag stays)
        size_t at = e + 8;
        if (bytes > 4) {
            const uint64_t off = rd32(e + 8);
            if (off + bytes > size) break;                         // cannot be read whole: Pillow's load() gives up HERE, the
            at = (size_t)off;                                      // entries behind this one are never seen
        }
        for (int t = 0; t < T_COUNT; ++t)
            if (ids[t] == id) tags[t] = Tag{(int)type, count, at, true};
    }
    // an integer tag in its plain form: SHORT or LONG; `which`-th value
    auto is_int = [&](const Tag &t) { return t.type == 3 || t.type == 4; };
    auto value = [&](const Tag &t, uint32_t which) { return t.type == 3 ? rd16(t.at + 2 * (size_t)which) : rd32(t.at + 4 * (size_t)which); };
    auto scalar = [&](int which, uint32_t absent, uint32_t &out) {     // false: present in a form this reader leaves to Pillow
        const Tag &t = tags[which];
        out = absent;
        if (!t.have) return true;
        if (!is_int(t) || t.count != 1) return false;
        out = value(t, 0);
        return true;
    };
    if (tags[T_WMP].have || tags[T_EXIF].have || tags[T_GPS].have || tags[T_INTEROP].have || tags[T_XMP].have || tags[T_TILES].have) return;
    uint32_t comp, planar, photo, fill, orient, spp, W, H, rows, dummy;
    if (!scalar(T_COMP, 1, comp) || !scalar(T_PLANAR, 1, planar) || !scalar(T_PHOTO, 0, photo) || !scalar(T_FILL, 1, fill) ||
        !scalar(T_ORIENT, 1, orient) || !scalar(T_SPP, 1, spp))
        return;
    if (comp != 1 || planar != 1 || fill != 1 || orient != 1 || photo > 3) return;
    if (!tags[T_WIDTH].have || !tags[T_LENGTH].have || !scalar(T_WIDTH, 0, W) || !scalar(T_LENGTH, 0, H)) return;
    if (W == 0 || H == 0 || W > 65535 || (uint64_t)W * H > (1ull << 28)) return;
    // resolution tags are read by _setup too: one value each (any numeric type), or absent
    for (int t : {T_XRES, T_YRES, T_UNIT})
        if (tags[t].have && tags[t].count != 1) return;
    (void)dummy;
    if (tags[T_FORMAT].have) {
        if (!is_int(tags[T_FORMAT]) || tags[T_FORMAT].count == 0) return;
        for (uint32_t k = 0; k < tags[T_FORMAT].count; ++k)
            if (value(tags[T_FORMAT], k) != 1) return;
    }
    // BitsPerSample with _setup's fix-ups: more values than samples are cut, a single value stands for all samples
    const Tag &bits = tags[T_BITS];
    if (!bits.have || !is_int(bits) || bits.count == 0 || spp == 0 || spp > 4) return;
    uint32_t nbits = bits.count;
    if (spp < nbits) nbits = spp;
    else if (spp > nbits && nbits != 1) return;
    for (uint32_t k = 0; k < nbits; ++k)
        if (value(bits, k) != 8) return;
    int extra = -1;                                                // no ExtraSamples tag
    if (tags[T_EXTRA].have) {
        if (!is_int(tags[T_EXTRA]) || tags[T_EXTRA].count != 1) return;
        extra = (int)value(tags[T_EXTRA], 0);
    }
    // (byte order, photometric, sample format, fill order, bits, extra samples) -> mode: the 8-bit rows of TiffImagePlugin.OPEN_INFO
    for (int k = 0; k < 256; ++k) info.lut[k] = (uint8_t)k;
    if ((photo == 0 || photo == 1) && spp == 1 && extra < 0) {
        info.channels = 1;
        if (photo == 0) {                                          // WhiteIsZero: raw mode "L;I"
            info.mapped = 1;
            for (int k = 0; k < 256; ++k) info.lut[k] = (uint8_t)(255 - k);
        }
    } else if (photo == 2 && spp == 3 && extra < 0) {
        info.channels = 3;
    } else if (photo == 2 && spp == 4 && (extra < 0 || extra == 2)) {
        info.channels = 4;                                         // unassociated alpha (or no ExtraSamples at all: Pillow reads RGBA)
    } else if (photo == 2 && spp == 4 && extra == 0) {
        info.channels = 3;                                         // "RGBX": the fourth sample is dropped
    } else if (photo == 3 && spp == 1 && extra < 0) {
        const Tag &cm = tags[T_COLORMAP];                          // 256 reds, 256 greens, 256 blues, 16 bits each: Pillow keeps v // 256
        if (!cm.have || cm.type != 3 || cm.count != 768) return;
        info.channels = 1;
        info.mapped = 1;
        for (int k = 0; k < 256; ++k) {
            const uint32_t r = rd16(cm.at + 2 * (size_t)k) >> 8, g = rd16(cm.at + 2 * (size_t)(256 + k)) >> 8, b = rd16(cm.at + 2 * (size_t)(512 + k)) >> 8;
            info.lut[k] = (uint8_t)((r * 19595u + g * 38470u + b * 7471u + 0x8000u) >> 16);
        }
    } else {
        return;
    }
    info.spp = (int32_t)spp;
    // strips
    const Tag &so = tags[T_STRIPS];
    if (!so.have || !is_int(so) || so.count == 0) return;
    if (!scalar(T_ROWS, H, rows) || rows == 0) return;
    uint32_t first = 0, count = so.count;
    if (rows == H) {                                               // "every tile covers the image: only use the last offset"
        first = so.count - 1;
        count = 1;
    } else if ((uint64_t)count != ((uint64_t)H + rows - 1) / rows) {
        return;                                                    // fewer strips leave rows blank, more start over at the top: Pillow's business
    }
    if (count > (1u << 20)) return;                                // (a strip per row of a 1-pixel-wide giant: not worth a gigabyte of offsets)
    const uint64_t stride = (uint64_t)W * spp;
    for (uint32_t s = 0; s < count; ++s) {
        const uint64_t off = value(so, first + s);
        const uint64_t y0 = (uint64_t)s * rows, nrows = (y0 + rows <= H ? rows : H - y0);
        if (off > size || stride * nrows > size - off) {           // the raw decoder wants every strip whole
            info.status = KE_TIFF_CORRUPT;
            return;
        }
    }
    if (strips)
        for (uint32_t s = 0; s < count; ++s) strips->push_back(value(so, first + s));
    info.width = (int32_t)W;
    info.height = (int32_t)H;
    info.rows_per_strip = (int32_t)(rows > H ? H : rows);
    info.nstrips = (int32_t)count;
    info.status = KE_TIFF_OK;
}
