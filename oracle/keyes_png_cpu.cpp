// keyes_png_cpu.cpp -- TEST INFRASTRUCTURE: the PNG decoding arithmetic of the product (kobato-eyes_amd/csrc/ke_png_core.h,
// ke_png_parse.h -- the headers the HIP kernels compile) driven sequentially on the CPU, so that the CPU test suite can hold
// it against the installed Pillow without a GPU.  Only tests/ load this library; the product never does.
#include <vector>

#include "../kobato-eyes_amd/csrc/ke_png_parse.h"

namespace {
struct MemSrc {
    const uint8_t *p;
    uint32_t byte(uint32_t pos) const { return p[pos]; }
};
struct VecSink {
    std::vector<uint8_t> &v;
    void put(uint8_t b) { v.push_back(b); }
    uint32_t get(uint32_t dist) const { return v[v.size() - dist]; }
    uint32_t size() const { return (uint32_t)v.size(); }
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
    std::vector<uint8_t> stream;
    ke_parse_png(file, (size_t)size, &stream, info);
    if (info.status != KE_PNG_OK) return info.status;
    const int rb = info.width * info.channels;
    std::vector<uint8_t> raw;
    raw.reserve((size_t)(rb + 1) * info.height);
    MemSrc src{stream.data() + info.zoff};
    KeBitsLsb<MemSrc> bits{&src, 0, info.zlen, 0, 0, 0};
    VecSink sink{raw};
    KeInflateTables t;
    uint8_t work[352];
    if (ke_inflate_zlib(bits, sink, (uint32_t)((size_t)(rb + 1) * info.height), t, work) != KE_PNG_OK) return KE_PNG_CORRUPT;
    if (raw.size() != (size_t)(rb + 1) * info.height) return KE_PNG_CORRUPT;
    for (int y = 0; y < info.height; ++y) {
        const uint8_t *row = raw.data() + (size_t)y * (rb + 1);
        if (ke_png_unfilter_row(row[0], row + 1, y ? out + (size_t)(y - 1) * rb : nullptr, out + (size_t)y * rb, rb, info.channels) != KE_PNG_OK)
            return KE_PNG_CORRUPT;
    }
    return KE_PNG_OK;
}

}  // extern "C"
