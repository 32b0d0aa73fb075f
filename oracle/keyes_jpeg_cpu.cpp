// keyes_jpeg_cpu.cpp -- TEST INFRASTRUCTURE: the JPEG decoding arithmetic of the product (kobato-eyes_amd/csrc/ke_jpeg_core.h,
// ke_jpeg_parse.h -- the very headers the HIP kernels compile) driven sequentially on the CPU, so that the CPU test suite can
// hold it against the installed Pillow (libjpeg-turbo) without a GPU.  Only tests/ load this library; the product never does.
#include <cstdlib>
#include <vector>

#include "../kobato-eyes_amd/csrc/ke_jpeg_parse.h"

extern "C" {

// widths/heights/channels/status of one file (status KE_JPEG_*)
int ko_jpeg_probe(const uint8_t *file, uint64_t size, int32_t *w, int32_t *h, int32_t *ch) {
    KeJpegTables tables;
    KeJpegInfo info;
    ke_parse_jpeg(file, (size_t)size, tables, info);
    *w = info.width; *h = info.height; *ch = info.ncomp;
    return info.status;
}

// decodes into out (h * w * channels bytes, packed); returns KE_JPEG_*
int ko_jpeg_decode(const uint8_t *file, uint64_t size, uint8_t *out) {
    KeJpegTables tables;
    KeJpegInfo info;
    ke_parse_jpeg(file, (size_t)size, tables, info);
    if (info.status != KE_JPEG_OK) return info.status;
    std::vector<uint8_t> planes[3];
    for (int c = 0; c < info.ncomp; ++c) planes[c].assign((size_t)info.plane_w[c] * info.plane_h[c], 0);
    KeBits bits;
    ke_bits_init(bits, file, info.scan_offset, info.scan_end);
    int pred[3] = {0, 0, 0};
    int32_t blk[64];
    int restart_left = info.restart_interval;
    for (int my = 0; my < info.mcus_y; ++my)
        for (int mx = 0; mx < info.mcus_x; ++mx) {
            if (info.restart_interval && restart_left == 0) {
                if (ke_bits_restart(bits) != KE_JPEG_OK) return KE_JPEG_CORRUPT;
                pred[0] = pred[1] = pred[2] = 0;
                restart_left = info.restart_interval;
            }
            for (int c = 0; c < info.ncomp; ++c)
                for (int by = 0; by < info.vs[c]; ++by)
                    for (int bx = 0; bx < info.hs[c]; ++bx) {
                        if (ke_decode_block(bits, tables.pool[info.huff_dc[c]], tables.pool[info.huff_ac[c]], info.quant[c], kKeZigzag,
                                            pred[c], blk) != KE_JPEG_OK)
                            return KE_JPEG_CORRUPT;
                        uint8_t *dst = planes[c].data() + (size_t)((my * info.vs[c] + by) * 8) * info.plane_w[c] + (mx * info.hs[c] + bx) * 8;
                        ke_idct_islow(blk, dst, info.plane_w[c]);
                    }
            --restart_left;
        }
    if (bits.overrun > 8) return KE_JPEG_CORRUPT;          // ran past the data (a few zero bytes of look-ahead are normal)
    for (int y = 0; y < info.height; ++y)
        for (int x = 0; x < info.width; ++x) {
            const int Y = planes[0][(size_t)y * info.plane_w[0] + x];
            if (info.ncomp == 1) { out[(size_t)y * info.width + x] = (uint8_t)Y; continue; }
            const int cb = ke_upsample_at(planes[1].data(), info.plane_w[1], info.comp_w[1], info.comp_h[1], info.hmax, info.vmax, x, y);
            const int cr = ke_upsample_at(planes[2].data(), info.plane_w[2], info.comp_w[2], info.comp_h[2], info.hmax, info.vmax, x, y);
            ke_ycc_to_rgb(Y, cb, cr, out + ((size_t)y * info.width + x) * 3);
        }
    return KE_JPEG_OK;
}

// ke_upsample4 (what the colour kernel calls) against ke_upsample_at (what the decode above calls) on random planes of every
// geometry: 0 = equal everywhere
int ko_jpeg_upsample_selftest(void) {
    uint32_t rng = 12345;
    for (int cw = 1; cw <= 21; ++cw)
        for (int ch = 1; ch <= 9; ++ch)
            for (int mode = 0; mode < 3; ++mode) {
                const int hf = mode ? 2 : 1, vf = mode == 2 ? 2 : 1;
                const int pw = ((cw + 7) / 8) * 8 + 8;
                std::vector<uint8_t> plane((size_t)pw * (ch + 8));
                for (auto &v : plane) { rng = rng * 1664525u + 1013904223u; v = (uint8_t)(rng >> 24); }
                const int W = cw * hf, H = ch * vf;                  // (odd image sizes only drop the last column / row)
                for (int y = 0; y < H; ++y)
                    for (int x0 = 0; x0 < W; x0 += 4) {
                        int got[4];
                        ke_upsample4(plane.data(), pw, cw, ch, hf, vf, x0, y, got);
                        for (int k = 0; k < 4 && x0 + k < W; ++k)
                            if (got[k] != ke_upsample_at(plane.data(), pw, cw, ch, hf, vf, x0 + k, y)) return 1 + cw * 1000 + ch * 10 + mode;
                    }
            }
    return 0;
}

}  // extern "C"
