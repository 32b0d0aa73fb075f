// keyes_bmp_cpu.cpp -- TEST INFRASTRUCTURE: the BMP header parsing of the product (kobato-eyes_amd/csrc/ke_bmp_parse.h, the
// header ke_bmp.hip compiles) with the unpacking done sequentially on the CPU, so that the CPU test suite can hold it against
// the installed Pillow without a GPU.  Only tests/ load this library; the product never does.
#include "../kobato-eyes_amd/csrc/ke_bmp_parse.h"

extern "C" {

int ko_bmp_probe(const uint8_t *file, uint64_t size, int32_t *w, int32_t *h, int32_t *ch) {
    KeBmpInfo info;
    ke_parse_bmp(file, (size_t)size, info);
    *w = info.width; *h = info.height; *ch = info.channels;
    return info.status;
}

int ko_bmp_decode(const uint8_t *file, uint64_t size, uint8_t *out) {
    KeBmpInfo info;
    ke_parse_bmp(file, (size_t)size, info);
    if (info.status != KE_BMP_OK) return info.status;
    const int bpp = info.bits / 8;
    for (int y = 0; y < info.height; ++y) {
        const uint8_t *row = file + info.data_off + (size_t)(info.topdown ? y : info.height - 1 - y) * info.stride;
        uint8_t *dst = out + (size_t)y * info.width * info.channels;
        for (int x = 0; x < info.width; ++x)
            for (int c = 0; c < info.channels; ++c)
                dst[(size_t)x * info.channels + c] = info.bits == 8 ? info.lut[row[x]] : row[(size_t)x * bpp + info.pick[c]];
    }
    return KE_BMP_OK;
}

}  // extern "C"
