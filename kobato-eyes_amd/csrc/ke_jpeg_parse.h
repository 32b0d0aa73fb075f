// ke_jpeg_parse.h -- host-side JPEG header parsing for the GPU decoder (and for the CPU build the tests compare against
// Pillow): markers up to the first scan (every scan of a progressive file), Huffman tables into decoding form, the geometry
// the kernels need.  Everything outside Huffman-coded JPEG with 8-bit samples -- sequential with one interleaved scan, or
// progressive -- grayscale or YCbCr at 4:4:4 / 4:2:2 / 4:2:0 is reported as KE_JPEG_UNSUPPORTED so that the caller decodes such a file with Pillow, as the reference does for every file
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
            if (code + cnt >= (1 << l)) return false;        // jdhuff.c: no code may be all ones
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

// jpeg_make_d_derived_tbl's checks on a table a scan is about to use (JERR_BAD_HUFF_TABLE: Pillow raises): the codes of every
// length leave the all-ones code free, and a DC table holds categories 0..15 only.  `dht` = 16 counts, then the symbols.
static inline bool ke_dht_usable(const uint8_t *dht, bool dc) {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        const int cnt = dht[l - 1];
        if (cnt && code + cnt >= (1 << l)) return false;
        code = (code + cnt) << 1;
        k += cnt;
    }
    if (dc)
        for (int i = 0; i < k; ++i)
            if (dht[16 + i] > 15) return false;
    return true;
}

struct KeJpegTables {            // the batch's pool of distinct Huffman tables (files made by one encoder share them)
    std::vector<KeHuffTable> pool;
    std::vector<std::vector<uint8_t>> keys;
    bool sequential_only = false;    // the tables of progressive files are not wanted (the GPU builds them from the DHT bytes)
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

// the entropy-coded segment that starts at `from` runs to the next marker that is not RSTn; 0xFF bytes are rare in it.
// Returns the position of that marker's 0xFF, or `size` when there is none.
static inline size_t ke_jpeg_segment_end(const uint8_t *p, size_t size, size_t from) {
    size_t e = from;
    while (e + 1 < size) {
        const void *hit = std::memchr(p + e, 0xFF, size - 1 - e);
        if (!hit) return size;
        e = (size_t)((const uint8_t *)hit - p);
        if (p[e + 1] != 0 && !(p[e + 1] >= 0xD0 && p[e + 1] <= 0xD7)) return e;
        e += 2;
    }
    return size;
}

// find_end = false: headers only (probing sizes); the end of the entropy-coded segment is then left unset.
// defer_end (with find_end): a sequential file's entropy-coded segment is NOT walked here -- scan_end is left at the file size
// and *defer_end is set: the caller applies the very same rule (ke_jpeg_segment_end, then "the marker must be EOI") where the
// bytes already are (the GPU decoder, ke_jpeg_find_end).  The walk reads every byte of the file; on the host it was 40 of the
// 60 ms a 65 536-file batch spent before its first kernel.  Progressive files are walked here as before: their scans' headers
// lie between the segments.
// scans: where the scans of a progressive file go (info.first_scan / nscans index it); without it such a file is
// KE_JPEG_UNSUPPORTED as before.
static inline void ke_parse_jpeg(const uint8_t *p, size_t size, KeJpegTables &tables, KeJpegInfo &info, bool find_end = true,
                                 std::vector<KeJpegScan> *scans = nullptr, bool *defer_end = nullptr) {
    if (defer_end) *defer_end = false;
    std::memset(&info, 0, sizeof info);
    info.status = KE_JPEG_UNSUPPORTED;
    if (size < 4 || p[0] != 0xFF || p[1] != 0xD8) { info.status = KE_JPEG_CORRUPT; return; }
    uint16_t qt[4][64];
    bool have_q[4] = {false, false, false, false};
    bool latched[3] = {false, false, false};
    int dc_tab[4] = {-1, -1, -1, -1}, ac_tab[4] = {-1, -1, -1, -1};
    uint32_t dc_dht[4] = {0, 0, 0, 0}, ac_dht[4] = {0, 0, 0, 0};
    int comp_id[3] = {0, 0, 0};
    bool have_sof = false, progressive = false, have_geometry = false;
    const size_t scan0 = scans ? scans->size() : 0;
    int coef_al[3][10];                                  // bit position the scans have brought zigzag coefficients 0..9 to (-1: never coded)
    for (auto &row : coef_al)
        for (int &v : row) v = -1;
    struct Undo {                                        // whatever way the function is left: no scans of a file that is not taken
        std::vector<KeJpegScan> *scans;
        size_t keep;
        const KeJpegInfo &info;
        ~Undo() { if (scans && info.status != KE_JPEG_OK) scans->resize(keep); }
    } undo{scans, scan0, info};
    auto give_up = [&](int status) { info.status = status; };
    int adobe_transform = -1;
    size_t pos = 2;
    while (pos + 2 <= size) {
        if (p[pos] != 0xFF) { info.status = KE_JPEG_CORRUPT; return; }
        while (pos < size && p[pos] == 0xFF) ++pos;          // fill bytes
        if (pos >= size) break;
        const int m = p[pos++];
        if ((m >= 0xD0 && m <= 0xD7) || m == 0x01) continue; // RSTn / TEM between segments: jdmarker.c reads on
        // what read_markers knows: SOFn, DHT, SOS, EOI, DQT, DNL, DRI, APPn, COM (DAC belongs to arithmetic coding: not taken
        // here).  A second SOI, JPGn and the reserved codes
        // are fatal there (JERR_SOI_DUPLICATE / JERR_UNKNOWN_MARKER), and Pillow's own scan stops at codes below 0xC0
        if (!((m >= 0xC0 && m <= 0xCF && m != 0xC8 && m != 0xCC) || m == 0xD9 || m == 0xDA || m == 0xDB || m == 0xDC || m == 0xDD || (m >= 0xE0 && m <= 0xEF) ||
              m == 0xFE))
            return give_up(KE_JPEG_UNSUPPORTED);
        if (m == 0xD9) {
            if (progressive && have_geometry && scans && scans->size() > scan0) {
                // libjpeg smooths blocks whose first AC coefficients have not been refined to the last bit (jdcoefct.c
                // smoothing_ok: coefficients 1..9 of any component still at al > 0 or never coded, DC known everywhere):
                // such a file is left to Pillow
                bool dc_known = true, unrefined = false;
                for (int c = 0; c < info.ncomp; ++c) {
                    dc_known = dc_known && coef_al[c][0] >= 0;
                    for (int z = 1; z < 10; ++z) unrefined = unrefined || coef_al[c][z] != 0;
                }
                if (dc_known && unrefined) return give_up(KE_JPEG_UNSUPPORTED);
                info.first_scan = (uint32_t)scan0;
                info.nscans = (int32_t)(scans->size() - scan0);
                info.scan_end = (uint32_t)(pos - 1);
                info.status = KE_JPEG_OK;
                return;
            }
            break;
        }
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
                const int id = (progressive && tables.sequential_only) ? 0 : tables.intern(seg + o + 1, seg + o + 17, total);
                if (id < 0) { info.status = KE_JPEG_CORRUPT; return; }
                (tc ? ac_tab : dc_tab)[th] = id;
                (tc ? ac_dht : dc_dht)[th] = (uint32_t)(seg + o + 1 - p);
                o += 17 + (size_t)total;
            }
            if (o != n) { info.status = KE_JPEG_CORRUPT; return; }     // get_dht: JERR_BAD_LENGTH
        } else if (m == 0xC0 || m == 0xC1 || (m == 0xC2 && (scans || !find_end))) {   // SOF0 / SOF1 (sequential), SOF2 (progressive)
            if (n < 6 || have_sof) return;
            progressive = m == 0xC2;
            if (seg[0] != 8) return;                         // sample precision
            info.height = (seg[1] << 8) | seg[2];
            info.width = (seg[3] << 8) | seg[4];
            info.ncomp = seg[5];
            if ((info.ncomp != 1 && info.ncomp != 3) || n != 6 + 3 * (size_t)info.ncomp) return;     // get_sof: JERR_BAD_LENGTH
            if (info.width < 1 || info.height < 1 || info.width > 65500 || info.height > 65500) return;   // JPEG_MAX_DIMENSION
            for (int c = 0; c < info.ncomp; ++c) {
                comp_id[c] = seg[6 + 3 * c];
                info.hs[c] = seg[7 + 3 * c] >> 4;
                info.vs[c] = seg[7 + 3 * c] & 15;
                info.tq[c] = seg[8 + 3 * c];
                if (info.tq[c] > 3) return;
                if (info.hs[c] < 1 || info.hs[c] > 4 || info.vs[c] < 1 || info.vs[c] > 4) return;     // jdinput.c: JERR_BAD_SAMPLING
            }
            have_sof = true;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return;                                          // progressive, lossless, arithmetic, hierarchical
        } else if (m == 0xDD) {                              // DRI
            if (n != 2) { info.status = KE_JPEG_CORRUPT; return; }     // get_dri: JERR_BAD_LENGTH
            info.restart_interval = (seg[0] << 8) | seg[1];
        } else if (m == 0xEE) {                              // Adobe
            if (n >= 12 && std::memcmp(seg, "Adobe", 5) == 0) adobe_transform = seg[11];
        } else if (m == 0xDA) {                              // SOS
            if (!have_sof || n < 1) return give_up(KE_JPEG_UNSUPPORTED);
            const int ns = seg[0];
            if (ns < 1 || ns > info.ncomp || n != 1 + 2 * (size_t)ns + 3) return give_up(KE_JPEG_UNSUPPORTED);   // get_sos: JERR_BAD_LENGTH
            if (!progressive && ns != info.ncomp) return give_up(KE_JPEG_UNSUPPORTED);       // sequential: one interleaved scan only
            KeJpegScan sc;
            std::memset(&sc, 0, sizeof sc);
            sc.ncomp = ns;
            const uint8_t *tail = seg + 1 + 2 * ns;
            sc.ss = tail[0]; sc.se = tail[1]; sc.ah = tail[2] >> 4; sc.al = tail[2] & 15;
            for (int k = 0; k < ns; ++k) {
                int c = -1;
                for (int q = 0; q < info.ncomp; ++q)
                    if (seg[1 + 2 * k] == comp_id[q]) c = q;
                if (c < 0 || (k > 0 && c <= sc.comp[k - 1])) return give_up(KE_JPEG_UNSUPPORTED);    // components in frame order
                if (!progressive && c != k) return give_up(KE_JPEG_UNSUPPORTED);
                sc.comp[k] = c;
                const int td = seg[2 + 2 * k] >> 4, ta = seg[2 + 2 * k] & 15;
                if (td > 3 || ta > 3) return give_up(KE_JPEG_CORRUPT);
                const bool need_dc = !progressive || (sc.ss == 0 && sc.ah == 0), need_ac = !progressive || sc.ss > 0;
                if ((need_dc && dc_tab[td] < 0) || (need_ac && ac_tab[ta] < 0) || !have_q[info.tq[c]]) return give_up(KE_JPEG_CORRUPT);
                if ((need_dc && !ke_dht_usable(p + dc_dht[td], true)) || (need_ac && !ke_dht_usable(p + ac_dht[ta], false)))
                    return give_up(KE_JPEG_CORRUPT);
                sc.dc_tab[k] = need_dc ? dc_tab[td] : 0;
                sc.ac_tab[k] = need_ac ? ac_tab[ta] : 0;
                sc.dc_dht[k] = need_dc ? dc_dht[td] : 0;
                sc.ac_dht[k] = need_ac ? ac_dht[ta] : 0;
                if (!progressive) {
                    info.td[c] = td; info.ta[c] = ta;
                    info.huff_dc[c] = dc_tab[td];
                    info.huff_ac[c] = ac_tab[ta];
                }
                if (!latched[c]) {                            // jdinput.c latch_quant_tables: the table in effect at the component's
                    std::memcpy(info.quant[c], qt[info.tq[c]], sizeof qt[0]);   // FIRST scan stays, a DQT between scans notwithstanding
                    latched[c] = true;
                }
            }
            if (!progressive) {
                if (sc.ss != 0 || sc.se != 63 || sc.ah != 0 || sc.al != 0) return give_up(KE_JPEG_UNSUPPORTED);
            } else {
                // jdphuff.c start_pass: a DC scan has se = 0, an AC scan one component; a refinement continues one bit lower
                const bool dc = sc.ss == 0;
                if (sc.se > 63 || sc.ss > sc.se || (dc && sc.se != 0) || (!dc && ns != 1) || sc.al > 13 ||
                    (sc.ah != 0 && sc.al != sc.ah - 1))
                    return give_up(KE_JPEG_UNSUPPORTED);
                for (int k = 0; k < ns; ++k)
                    for (int z = sc.ss; z <= sc.se && z < 10; ++z) coef_al[sc.comp[k]][z] = sc.al;
            }
            if (!have_geometry) {
                // colour space as libjpeg guesses it (jdapimin.c default_decompress_parms): 3 components are YCbCr unless an
                // Adobe marker says transform 0 or the component ids spell "RGB"
                if (info.ncomp == 3) {
                    if (adobe_transform == 0) return give_up(KE_JPEG_UNSUPPORTED);
                    if (adobe_transform < 0 && comp_id[0] == 'R' && comp_id[1] == 'G' && comp_id[2] == 'B') return give_up(KE_JPEG_UNSUPPORTED);
                    // luma at full resolution, both chroma planes alike, factors 1x1 / 2x1 / 2x2 / 1x2 (4:4:0)
                    const int h = info.hs[0], v = info.vs[0];
                    if (info.hs[1] != 1 || info.vs[1] != 1 || info.hs[2] != 1 || info.vs[2] != 1) return give_up(KE_JPEG_UNSUPPORTED);
                    if (!((h == 1 && v == 1) || (h == 2 && v == 1) || (h == 2 && v == 2) || (h == 1 && v == 2))) return give_up(KE_JPEG_UNSUPPORTED);
                } else {
                    info.hs[0] = info.vs[0] = 1;             // a single component is never subsampled (its factors are ignored)
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
                if ((uint64_t)info.width * info.height > (1ull << 28)) return give_up(KE_JPEG_UNSUPPORTED);
                info.progressive = progressive ? 1 : 0;
                info.scan_offset = (uint32_t)(pos + len);
                have_geometry = true;
                if (!find_end) { info.scan_end = (uint32_t)size; info.status = KE_JPEG_OK; return; }
            }
            if (defer_end && !progressive) {                 // the caller finds the end (same rule, on the device)
                if (pos + len + 1 >= size) return give_up(KE_JPEG_CORRUPT);
                info.scan_end = (uint32_t)size;
                info.status = KE_JPEG_OK;
                *defer_end = true;
                return;
            }
            const size_t e = ke_jpeg_segment_end(p, size, pos + len);
            if (e + 1 >= size) return give_up(KE_JPEG_CORRUPT);           // no EOI: truncated, Pillow raises
            if (!progressive) {
                if (p[e + 1] != 0xD9) return give_up(KE_JPEG_UNSUPPORTED);   // another scan or table follows: not a single-scan file
                info.scan_end = (uint32_t)e;
                info.status = KE_JPEG_OK;
                return;
            }
            sc.offset = (uint32_t)(pos + len);
            sc.end = (uint32_t)e;
            sc.restart_interval = info.restart_interval;
            scans->push_back(sc);
            pos = e;                                         // on to the marker behind the scan
            continue;
        }
        pos += len;
    }
}
