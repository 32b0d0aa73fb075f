// keyes_png_cpu.cpp -- TEST INFRASTRUCTURE: the PNG decoding arithmetic of the product (kobato-eyes_amd/csrc/ke_png_core.h,
// ke_png_parse.h -- the headers the HIP kernels compile) driven sequentially on the CPU, so that the CPU test suite can hold
// it against the installed Pillow without a GPU.  Only tests/ load this library; the product never does.
#include <vector>

#include "../kobato-eyes_amd/csrc/ke_png_parse.h"

namespace {
struct MemSrc {
    const uint8_t *p;
    uint32_t len;
    uint32_t word(uint32_t k) const {
        uint32_t w = 0;
        for (uint32_t j = 0; j < 4; ++j)
            if ((uint64_t)k * 4 + j < len) w |= (uint32_t)p[(size_t)k * 4 + j] << (8 * j);
        return w;
    }
    void tick(uint32_t) {}
};
struct VecSink {
    std::vector<uint8_t> &v;
    void put(uint8_t b) { v.push_back(b); }
    void copy(uint32_t dist, uint32_t len) {
        for (uint32_t k = 0; k < len; ++k) v.push_back(v[v.size() - dist]);
    }
    uint32_t size() const { return (uint32_t)v.size(); }
    bool matches_now(bool) const { return true; }     // one stream: a match is finished where it starts
    void finish() {}
};
}  // namespace

extern "C" {

int ko_png_probe(const uint8_t *file, uint64_t size, int32_t *w, int32_t *h, int32_t *ch) {
    KePngInfo info;
    ke_parse_png(file, (size_t)size, nullptr, info);
    *w = info.width; *h = info.height; *ch = info.channels;
    return info.status;
}

int ko_png_decode(const uint8_t *file, uint64_t size, uint8_t *out) {
    KePngInfo info;
    std::vector<KePngSeg> segs;
    ke_parse_png(file, (size_t)size, &segs, info);
    if (info.status != KE_PNG_OK) return info.status;
    std::vector<uint8_t> stream;
    for (const KePngSeg &s : segs) stream.insert(stream.end(), file + s.off, file + s.off + s.len);
    const size_t want = info.raw_len;
    std::vector<uint8_t> raw;
    raw.reserve(want);
    MemSrc src{stream.data(), info.zlen};
    KeBitsLsb<MemSrc> bits{&src, 0, 0, 0};
    VecSink sink{raw};
    KeInflateTables t;
    uint32_t adler = 0;
    if (ke_inflate_zlib(bits, sink, info.zlen, (uint32_t)want, t, &adler) != KE_PNG_OK) return KE_PNG_CORRUPT;
    if (raw.size() != want || ke_adler32(raw.data(), raw.size()) != adler) return KE_PNG_CORRUPT;
    // every pass (one for a file without interlacing) is a filtered image of its own: unfilter its rows, then put its samples
    // where they belong -- packed samples unpacked and mapped to luma, gray + alpha reduced to the gray samples
    const bool wide = info.depth == 16;                      // 16-bit samples, big-endian: 8-bit pixels leave (ke_png_parse.h)
    const int out_bpp = wide ? info.channels : (info.mapped || info.fbpp != info.channels) ? 1 : info.channels;
    const int unit = info.mapped ? 1 : info.fbpp * (wide ? 2 : 1);
    const uint8_t *at = raw.data();
    std::vector<uint8_t> rows;
    for (int pass = 0; pass < (info.interlace ? 7 : 1); ++pass) {
        KeAdam7 a{0, 0, 1, 1, info.width, info.height};
        if (info.interlace) a = ke_adam7_pass(pass, info.width, info.height);
        if (a.w == 0 || a.h == 0) continue;
        const int rb = (int)(((int64_t)a.w * info.fbpp * info.depth + 7) / 8);
        rows.assign((size_t)rb * a.h, 0);
        for (int y = 0; y < a.h; ++y) {
            const uint8_t *row = at + (size_t)y * (rb + 1);
            uint8_t *cur = rows.data() + (size_t)y * rb;
            if (ke_png_unfilter_row(row[0], row + 1, y ? cur - rb : nullptr, cur, rb, unit) != KE_PNG_OK) return KE_PNG_CORRUPT;
            uint8_t *dst = out + ((size_t)(a.y0 + y * a.dy) * info.width + a.x0) * out_bpp;
            for (int x = 0; x < a.w; ++x, dst += (size_t)a.dx * out_bpp) {
                if (wide) {
                    const uint8_t *s = cur + (size_t)x * unit;
                    if (info.fbpp == 1) dst[0] = s[0] ? 255 : s[1];                          // I;16 -> L: clipped
                    else if (info.fbpp == 2) { dst[0] = dst[1] = dst[2] = s[0]; dst[3] = s[2]; }   // LA;16B -> RGBA
                    else for (int c = 0; c < out_bpp; ++c) dst[c] = s[2 * c];                // the high bytes
                } else if (info.mapped) {
                    const int bit = x * info.depth;
                    dst[0] = info.lut[(cur[bit >> 3] >> (8 - info.depth - (bit & 7))) & ((1u << info.depth) - 1u)];
                } else if (out_bpp == 1) {
                    dst[0] = cur[(size_t)x * info.fbpp];
                } else {
                    for (int c = 0; c < out_bpp; ++c) dst[c] = cur[(size_t)x * out_bpp + c];
                }
            }
        }
        at += (size_t)(rb + 1) * a.h;
    }
    return KE_PNG_OK;
}

// literals / copies / bytes copied of a file's zlib stream (what the benchmarks quote next to a decode rate)
int ko_png_stats(const uint8_t *file, uint64_t size, uint64_t *literals, uint64_t *copies, uint64_t *copied_bytes) {
    KePngInfo info;
    std::vector<KePngSeg> segs;
    ke_parse_png(file, (size_t)size, &segs, info);
    if (info.status != KE_PNG_OK) return info.status;
    std::vector<uint8_t> stream;
    for (const KePngSeg &s : segs) stream.insert(stream.end(), file + s.off, file + s.off + s.len);
    struct CountSink {
        uint64_t lit = 0, cp = 0, cpb = 0;
        void put(uint8_t) { ++lit; }
        void copy(uint32_t, uint32_t len) { ++cp; cpb += len; }
        uint32_t size() const { return (uint32_t)(lit + cpb); }
        bool matches_now(bool) const { return true; }
        void finish() {}
    } sink;
    MemSrc src{stream.data(), info.zlen};
    KeBitsLsb<MemSrc> bits{&src, 0, 0, 0};
    KeInflateTables t;
    uint32_t adler = 0;
    const int rc = ke_inflate_zlib(bits, sink, info.zlen, 0xFFFFFFFFu, t, &adler);
    *literals = sink.lit; *copies = sink.cp; *copied_bytes = sink.cpb;
    return rc;
}

}  // extern "C"
