// keyes_tiff_cpu.cpp -- TEST INFRASTRUCTURE: the TIFF directory parsing of the product (kobato-eyes_amd/csrc/ke_tiff_parse.h,
// the header ke_tiff.hip compiles) with the unpacking done sequentially on the CPU, so that the CPU test suite can hold it
// against the installed Pillow without a GPU.  Only tests/ load this library; the product never does.
#include "../kobato-eyes_amd/csrc/ke_tiff_parse.h"

extern "C" {

int ko_tiff_probe(const uint8_t *file, uint64_t size, int32_t *w, int32_t *h, int32_t *ch) {
    KeTiffInfo info;
    ke_parse_tiff(file, (size_t)size, nullptr, info);
    *w = info.width; *h = info.height; *ch = info.channels;
    return info.status;
}

int ko_tiff_decode(const uint8_t *file, uint64_t size, uint8_t *out) {
    KeTiffInfo info;
    std::vector<uint32_t> strips;
    ke_parse_tiff(file, (size_t)size, &strips, info);
    if (info.status != KE_TIFF_OK) return info.status;
    for (int y = 0; y < info.height; ++y) {
        const uint8_t *row = file + strips[(size_t)(y / info.rows_per_strip)] + (size_t)(y % info.rows_per_strip) * info.width * info.spp;
        uint8_t *dst = out + (size_t)y * info.width * info.channels;
        for (int x = 0; x < info.width; ++x)
            for (int c = 0; c < info.channels; ++c) {
                const uint8_t v = row[(size_t)x * info.spp + c];
                dst[(size_t)x * info.channels + c] = info.mapped ? info.lut[v] : v;
            }
    }
    return KE_TIFF_OK;
}

}  // extern "C"
