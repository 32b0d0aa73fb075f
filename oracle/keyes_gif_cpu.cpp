// keyes_gif_cpu.cpp -- TEST INFRASTRUCTURE: the GIF container walk and LZW arithmetic of the product
// (kobato-eyes_amd/csrc/ke_gif_core.h, the header ke_gif.hip compiles) driven sequentially on the CPU, so that the CPU test suite
// can hold it against the installed Pillow without a GPU.  Only tests/ load this library; the product never does.
#include <vector>

#include "../kobato-eyes_amd/csrc/ke_gif_core.h"

namespace {
struct MemSrc {
    const uint8_t *p;
    uint8_t byte(uint32_t pos) const { return p[pos]; }
};
struct VecDict {
    uint32_t at[4096], n[4096];
    void set(uint32_t code, uint32_t pos, uint32_t len) { at[code] = pos; n[code] = len; }
    void get(uint32_t code, uint32_t &pos, uint32_t &len) const { pos = at[code]; len = n[code]; }
};
struct VecSink {
    std::vector<uint8_t> &v;
    void literal(uint8_t b) { v.push_back(b); }
    void copy(uint32_t from, uint32_t len) {
        for (uint32_t k = 0; k < len; ++k) v.push_back(v[from + k]);      // may run into its own output by one character
    }
};
}  // namespace

extern "C" {

int ko_gif_probe(const uint8_t *file, uint64_t size, int32_t *w, int32_t *h, int32_t *ch) {
    KeGifInfo info;
    ke_parse_gif(file, (size_t)size, info);
    *w = info.width; *h = info.height; *ch = info.channels;
    return info.status;
}

int ko_gif_decode(const uint8_t *file, uint64_t size, uint8_t *out) {
    KeGifInfo info;
    ke_parse_gif(file, (size_t)size, info);
    if (info.status != KE_GIF_OK) return info.status;
    const uint32_t want = (uint32_t)info.width * (uint32_t)info.height;
    std::vector<uint8_t> idx;
    idx.reserve(want);
    MemSrc src{file};
    static thread_local VecDict dict;
    VecSink sink{idx};
    const int rc = ke_gif_lzw(src, info.data_off, (uint32_t)size, info.bits, want, dict, sink);
    if (rc != KE_GIF_OK) return rc;
    for (int k = 0; k < info.height; ++k) {
        uint8_t *dst = out + (size_t)ke_gif_row(k, info.height, info.interlace) * info.width;
        for (int x = 0; x < info.width; ++x) dst[x] = info.lut[idx[(size_t)k * info.width + x]];
    }
    return KE_GIF_OK;
}

}  // extern "C"
