// ke_jpeg_parse.h -- host-side JPEG header parsing for the GPU decoder (and for the CPU build the tests compare against
// Pillow): markers up to the first scan, Huffman tables into decoding form, the geometry the kernels need.  Everything
// outside baseline sequential Huffman JPEG with 8-bit samples, one scan, grayscale or YCbCr at 4:4:4 / 4:2:2 / 4:2:0 is
// reported as KE_JPEG_UNSUPPORTED so that the caller decodes such a file with Pillow, as the reference does for every file
// (src/core/fastsig.py:31-34).
#pragma once

#include <cstring>
#include <vector>

#include "ke_jpeg_core.h"

static const uint8_t kKeZigzag[64] = KE_ZZ;

// DHT payload (counts[16], symbols) -> decoding form; false if the table is malformed
static inline bool ke_build_huff(const uint8_t *counts, const uint8_t *symbols, int nsym, KeHuffTable &t) {
    std::memset(&t, 0, sizeof t);
    int total = 0;
    for (int l = 0; l < 16; ++l) total += counts[l];
    if (total != nsym || total > 256) return false;
    std::memcpy(t.huffval, symbols, (size_t)nsym);
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        const int cnt = counts[l - 1];
        if (cnt == 0) {
            t.maxcode[l] = -1;
        } else {
            t.valoffset[l] = k - code;
            if (code + cnt > (1 << l)) return false;
            for (int i = 0; i < cnt; ++i, ++k, ++code) {
                if (l <= 9) {
                    const int first = code << (9 - l), span = 1 << (9 - l);
                    for (int j = 0; j < span; ++j) t.look[first + j] = (uint16_t)((l << 8) | symbols[k]);
                }
            }
            t.maxcode[l] = code - 1;
        }
        code <<= 1;
    }
    t.maxcode[17] = 0x7fffffff;
    return true;
}

struct KeJpegTables {            // the batch's pool of distinct Huffman tables (files made by one encoder share them)
    std::vector<KeHuffTable> pool;
    std::vector<std::vector<uint8_t>> keys;
    int intern(const uint8_t *counts, const uint8_t *symbols, int nsym) {
        std::vector<uint8_t> key(counts, counts + 16);
        key.insert(key.end(), symbols, symbols + nsym);
        for (size_t i = 0; i < keys.size(); ++i)
            if (keys[i] == key) return (int)i;
        KeHuffTable t;
        if (!ke_build_huff(counts, symbols, nsym, t)) return -1;
        pool.push_back(t);
        keys.push_back(key);
        return (int)pool.size() - 1;
    }
};

// find_end = false: headers only (probing sizes); the end of the entropy-coded segment is then left unset
static inline void ke_parse_jpeg(const uint8_t *p, size_t size, KeJpegTables &tables, KeJpegInfo &info, bool find_end = true) {
    std::memset(&info, 0, sizeof info);
    info.status = KE_JPEG_UNSUPPORTED;
    if (size < 4 || p[0] != 0xFF || p[1] != 0xD8) { info.status = KE_JPEG_CORRUPT; return; }
    uint16_t qt[4][64];
    bool have_q[4] = {false, false, false, false};
    int dc_tab[4] = {-1, -1, -1, -1}, ac_tab[4] = {-1, -1, -1, -1};
    int comp_id[3] = {0, 0, 0};
    bool have_sof = false;
    int adobe_transform = -1;
    size_t pos = 2;
    while (pos + 4 <= size) {
        if (p[pos] != 0xFF) { info.status = KE_JPEG_CORRUPT; return; }
        while (pos < size && p[pos] == 0xFF) ++pos;          // fill bytes
        if (pos >= size) break;
        const int m = p[pos++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) break;
        if (pos + 2 > size) break;
        const size_t len = ((size_t)p[pos] << 8) | p[pos + 1];
        if (len < 2 || pos + len > size) { info.status = KE_JPEG_CORRUPT; return; }
        const uint8_t *seg = p + pos + 2;
        const size_t n = len - 2;
        if (m == 0xDB) {                                     // DQT
            size_t o = 0;
            while (o < n) {
                const int pq = seg[o] >> 4, tq = seg[o] & 15;
                ++o;
                if (tq > 3 || pq > 1 || o + (pq ? 128 : 64) > n) { info.status = KE_JPEG_CORRUPT; return; }
                for (int k = 0; k < 64; ++k) {
                    const int v = pq ? ((seg[o] << 8) | seg[o + 1]) : seg[o];
                    o += pq ? 2 : 1;
                    qt[tq][kKeZigzag[k]] = (uint16_t)v;
                }
                if (pq) return;                              // 16-bit tables: 12-bit data or an unusual encoder -> Pillow
                have_q[tq] = true;
            }
        } else if (m == 0xC4) {                              // DHT
            size_t o = 0;
            while (o + 17 <= n) {
                const int tc = seg[o] >> 4, th = seg[o] & 15;
                int total = 0;
                for (int l = 0; l < 16; ++l) total += seg[o + 1 + l];
                if (tc > 1 || th > 3 || o + 17 + (size_t)total > n || total > 256) { info.status = KE_JPEG_CORRUPT; return; }
                const int id = tables.intern(seg + o + 1, seg + o + 17, total);
                if (id < 0) { info.status = KE_JPEG_CORRUPT; return; }
                (tc ? ac_tab : dc_tab)[th] = id;
                o += 17 + (size_t)total;
            }
        } else if (m == 0xC0 || m == 0xC1) {                 // SOF0 / SOF1 (sequential Huffman)
            if (n < 6 || have_sof) return;
            if (seg[0] != 8) return;                         // sample precision
            info.height = (seg[1] << 8) | seg[2];
            info.width = (seg[3] << 8) | seg[4];
            info.ncomp = seg[5];
            if ((info.ncomp != 1 && info.ncomp != 3) || n < 6 + 3 * (size_t)info.ncomp) return;
            if (info.width < 1 || info.height < 1) return;
            for (int c = 0; c < info.ncomp; ++c) {
                comp_id[c] = seg[6 + 3 * c];
                info.hs[c] = seg[7 + 3 * c] >> 4;
                info.vs[c] = seg[7 + 3 * c] & 15;
                info.tq[c] = seg[8 + 3 * c];
                if (info.tq[c] > 3) return;
            }
            have_sof = true;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return;                                          // progressive, lossless, arithmetic, hierarchical
        } else if (m == 0xDD) {                              // DRI
            if (n < 2) { info.status = KE_JPEG_CORRUPT; return; }
            info.restart_interval = (seg[0] << 8) | seg[1];
        } else if (m == 0xEE) {                              // Adobe
            if (n >= 12 && std::memcmp(seg, "Adobe", 5) == 0) adobe_transform = seg[11];
        } else if (m == 0xDA) {                              // SOS
            if (!have_sof || n < 1 || seg[0] != info.ncomp || n < 1 + 2 * (size_t)info.ncomp + 3) return;   // one interleaved scan only
            for (int c = 0; c < info.ncomp; ++c) {
                if (seg[1 + 2 * c] != comp_id[c]) return;
                info.td[c] = seg[2 + 2 * c] >> 4;
                info.ta[c] = seg[2 + 2 * c] & 15;
                if (info.td[c] > 3 || info.ta[c] > 3 || dc_tab[info.td[c]] < 0 || ac_tab[info.ta[c]] < 0 || !have_q[info.tq[c]]) {
                    info.status = KE_JPEG_CORRUPT;
                    return;
                }
                info.huff_dc[c] = dc_tab[info.td[c]];
                info.huff_ac[c] = ac_tab[info.ta[c]];
                std::memcpy(info.quant[c], qt[info.tq[c]], sizeof qt[0]);
            }
            const uint8_t *tail = seg + 1 + 2 * info.ncomp;
            if (tail[0] != 0 || tail[1] != 63 || tail[2] != 0) return;
            // colour space as libjpeg guesses it (jdapimin.c default_decompress_parms): 3 components are YCbCr unless an
            // Adobe marker says transform 0 or the component ids spell "RGB"
            if (info.ncomp == 3) {
                if (adobe_transform == 0) return;
                if (adobe_transform < 0 && comp_id[0] == 'R' && comp_id[1] == 'G' && comp_id[2] == 'B') return;
                // luma at full resolution, both chroma planes alike, factors 1x1 / 2x1 / 2x2
                const int h = info.hs[0], v = info.vs[0];
                if (info.hs[1] != 1 || info.vs[1] != 1 || info.hs[2] != 1 || info.vs[2] != 1) return;
                if (!((h == 1 && v == 1) || (h == 2 && v == 1) || (h == 2 && v == 2))) return;
            } else {
                info.hs[0] = info.vs[0] = 1;                 // a single component is never subsampled (its factors are ignored)
            }
            info.hmax = info.hs[0];
            info.vmax = info.vs[0];
            info.mcus_x = (info.width + 8 * info.hmax - 1) / (8 * info.hmax);
            info.mcus_y = (info.height + 8 * info.vmax - 1) / (8 * info.vmax);
            for (int c = 0; c < info.ncomp; ++c) {
                info.comp_w[c] = (info.width * info.hs[c] + info.hmax - 1) / info.hmax;
                info.comp_h[c] = (info.height * info.vs[c] + info.vmax - 1) / info.vmax;
                info.plane_w[c] = info.mcus_x * 8 * info.hs[c];
                info.plane_h[c] = info.mcus_y * 8 * info.vs[c];
            }
            if ((uint64_t)info.width * info.height > (1ull << 28)) return;
            info.scan_offset = (uint32_t)(pos + len);
            if (!find_end) { info.scan_end = (uint32_t)size; info.status = KE_JPEG_OK; return; }
            // the entropy-coded segment runs to the next marker that is not RSTn (normally EOI); 0xFF bytes are rare in it
            size_t e = pos + len;
            while (e + 1 < size) {
                const void *hit = std::memchr(p + e, 0xFF, size - 1 - e);
                if (!hit) { e = size; break; }
                e = (size_t)((const uint8_t *)hit - p);
                if (p[e + 1] != 0 && !(p[e + 1] >= 0xD0 && p[e + 1] <= 0xD7)) break;
                e += 2;
            }
            if (e + 1 >= size) { info.status = KE_JPEG_CORRUPT; return; }      // no EOI: truncated, Pillow raises
            if (p[e + 1] != 0xD9) return;                    // another scan or table follows: not a single-scan file
            info.scan_end = (uint32_t)e;
            info.status = KE_JPEG_OK;
            return;
        }
        pos += len;
    }
}
