// keyes_jpeg_cpu.cpp -- TEST INFRASTRUCTURE: the JPEG decoding arithmetic of the product (kobato-eyes_amd/csrc/ke_jpeg_core.h,
// ke_jpeg_parse.h -- the very headers the HIP kernels compile) driven sequentially on the CPU, so that the CPU test suite can
// hold it against the installed Pillow (libjpeg-turbo) without a GPU.  Only tests/ load this library; the product never does.
#include <cstdlib>
#include <vector>

#include "../kobato-eyes_amd/csrc/ke_jpeg_parse.h"

namespace {

struct TableReader {       // the reader of ke_prog_*: Huffman symbols through the parsed look-up tables
    KeBits &b;
    const KeJpegTables &tables;
    const int32_t *ids;    // table of each slot
    int sym(int slot) { return ke_huff_decode(b, tables.pool[(size_t)ids[slot]]); }
    uint32_t bits(int k) {
        ke_bits_fill(b);
        const uint32_t v = ke_bits_peek(b, k);
        ke_bits_skip(b, k);
        return v;
    }
    int bit() { return (int)bits(1); }
};

// every scan of a progressive file into coef[c] (blocks of the padded planes, 64 int16 each, natural order)
int decode_progressive(const uint8_t *file, const KeJpegTables &tables, const KeJpegInfo &info, const std::vector<KeJpegScan> &scans,
                       std::vector<int16_t> *coef) {
    for (int si = 0; si < info.nscans; ++si) {
        const KeJpegScan &sc = scans[info.first_scan + (size_t)si];
        KeBits bits;
        ke_bits_init(bits, file, sc.offset, sc.end);
        TableReader rd{bits, tables, sc.ss == 0 ? sc.dc_tab : sc.ac_tab};
        int pred[3] = {0, 0, 0};
        uint32_t eobrun = 0;
        int restart_left = sc.restart_interval;
        auto block = [&](int k, int c, int brow, int bcol) -> int {
            int16_t *blk = coef[c].data() + ((size_t)brow * (info.plane_w[c] >> 3) + bcol) * 64;
            if (sc.ss == 0) {
                if (sc.ah == 0) {
                    int v;
                    if (ke_prog_dc_first(rd, k, pred[k], sc.al, &v) != KE_JPEG_OK) return KE_JPEG_CORRUPT;
                    blk[0] = (int16_t)v;
                } else if (rd.bit()) {
                    blk[0] = (int16_t)(blk[0] | (1 << sc.al));
                }
                return KE_JPEG_OK;
            }
            return sc.ah == 0 ? ke_prog_ac_first(rd, k, blk, kKeZigzag, sc.ss, sc.se, sc.al, eobrun)
                              : ke_prog_ac_refine(rd, k, blk, kKeZigzag, sc.ss, sc.se, sc.al, eobrun);
        };
        auto restart = [&]() -> int {
            if (sc.restart_interval && restart_left == 0) {
                if (ke_bits_restart(bits) != KE_JPEG_OK) return KE_JPEG_CORRUPT;
                pred[0] = pred[1] = pred[2] = 0;
                eobrun = 0;
                restart_left = sc.restart_interval;
            }
            return KE_JPEG_OK;
        };
        if (sc.ncomp > 1) {                                  // interleaved: MCUs of hs x vs blocks per component, padding blocks coded
            for (int my = 0; my < info.mcus_y; ++my)
                for (int mx = 0; mx < info.mcus_x; ++mx) {
                    if (restart() != KE_JPEG_OK) return KE_JPEG_CORRUPT;
                    for (int k = 0; k < sc.ncomp; ++k) {
                        const int c = sc.comp[k];
                        for (int by = 0; by < info.vs[c]; ++by)
                            for (int bx = 0; bx < info.hs[c]; ++bx)
                                if (block(k, c, my * info.vs[c] + by, mx * info.hs[c] + bx) != KE_JPEG_OK) return KE_JPEG_CORRUPT;
                    }
                    --restart_left;
                }
        } else {                                             // one component: an MCU is one block, only the blocks that hold samples
            const int c = sc.comp[0], bw = (info.comp_w[c] + 7) >> 3, bh = (info.comp_h[c] + 7) >> 3;
            for (int brow = 0; brow < bh; ++brow)
                for (int bcol = 0; bcol < bw; ++bcol) {
                    if (restart() != KE_JPEG_OK) return KE_JPEG_CORRUPT;
                    if (block(0, c, brow, bcol) != KE_JPEG_OK) return KE_JPEG_CORRUPT;
                    --restart_left;
                }
        }
        if (ke_bits_ran_dry(bits)) return KE_JPEG_CORRUPT;
    }
    return KE_JPEG_OK;
}

}  // namespace

extern "C" {

// widths/heights/channels/status of one file (status KE_JPEG_*)
int ko_jpeg_probe(const uint8_t *file, uint64_t size, int32_t *w, int32_t *h, int32_t *ch) {
    KeJpegTables tables;
    KeJpegInfo info;
    ke_parse_jpeg(file, (size_t)size, tables, info, false);
    *w = info.width; *h = info.height; *ch = info.ncomp;
    return info.status;
}

// decodes into out (h * w * channels bytes, packed); returns KE_JPEG_*
int ko_jpeg_decode(const uint8_t *file, uint64_t size, uint8_t *out) {
    KeJpegTables tables;
    KeJpegInfo info;
    std::vector<KeJpegScan> scans;
    ke_parse_jpeg(file, (size_t)size, tables, info, true, &scans);
    if (info.status != KE_JPEG_OK) return info.status;
    bool narrow = true;                 // every block inside ke_idct_islow's bound (ke_jpeg_core.h); the file is handed back otherwise
    std::vector<uint8_t> planes[3];
    for (int c = 0; c < info.ncomp; ++c) planes[c].assign((size_t)info.plane_w[c] * info.plane_h[c], 0);
    if (info.progressive) {
        std::vector<int16_t> coef[3];
        for (int c = 0; c < info.ncomp; ++c) coef[c].assign((size_t)info.plane_w[c] * info.plane_h[c], 0);
        if (decode_progressive(file, tables, info, scans, coef) != KE_JPEG_OK) return KE_JPEG_CORRUPT;
        for (int c = 0; c < info.ncomp; ++c) {
            const int bpr = info.plane_w[c] >> 3, rows = info.plane_h[c] >> 3;
            for (int brow = 0; brow < rows; ++brow)
                for (int bcol = 0; bcol < bpr; ++bcol) {
                    int32_t blk[64];
                    const int16_t *src = coef[c].data() + ((size_t)brow * bpr + bcol) * 64;
                    for (int k = 0; k < 64; ++k) blk[k] = (int32_t)src[k] * (int32_t)info.quant[c][k];
                    narrow &= ke_idct_islow(blk, planes[c].data() + (size_t)brow * 8 * info.plane_w[c] + bcol * 8, info.plane_w[c]);
                }
        }
    }
    KeBits bits;
    ke_bits_init(bits, file, info.scan_offset, info.scan_end);
    int pred[3] = {0, 0, 0};
    int32_t blk[64];
    int restart_left = info.restart_interval;
    for (int my = 0; my < (info.progressive ? 0 : info.mcus_y); ++my)
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
                        narrow &= ke_idct_islow(blk, dst, info.plane_w[c]);
                    }
            --restart_left;
        }
    if (!info.progressive && ke_bits_ran_dry(bits)) return KE_JPEG_CORRUPT;     // used bits that are not in the file
    if (!narrow) return KE_JPEG_UNSUPPORTED;
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
