// ke_jpeg_core.h -- baseline JPEG decoding arithmetic shared by the HIP kernels (ke_jpeg.hip) and by the CPU build that the
// tests compare against Pillow (oracle/keyes_jpeg_cpu.cpp).  Plain C++ without allocation; KE_HD marks what the device
// compiles too.
//
// What is restated here is libjpeg's decoder as Pillow drives it for `Image.open(path)` -- the decode step of the
// reference's batch hasher (src/core/fastsig.py:31-34) and of safe_load_image (src/utils/image_io.py:60-138):
//   entropy decoding   : ITU T.81 F.2 (sequential Huffman, 8-bit), restart intervals, byte stuffing;
//   dequantise + IDCT  : jidctint.c `jpeg_idct_islow` (the default dct_method), 13-bit constants, 2 extra bits after pass 1;
//   upsampling         : jdsample.c "fancy" triangle filters h2v1 / h2v2 (do_fancy_upsampling is libjpeg's default), edge
//                        rows and columns replicated as jdmainct.c presents them;
//   colour             : jdcolor.c YCbCr -> RGB with its 16-bit fixed-point tables.
//   progressive files  : T.81 G.1.2 (spectral selection, successive approximation), scan by scan into one coefficient array;
// Arithmetic-coded, 12-bit, CMYK/YCCK, RGB-coded and 4:4:0 / 4:1:1-style files are refused by the parser
// (KE_JPEG_UNSUPPORTED) and stay with Pillow.
#pragma once

#include <stdint.h>

#ifdef __HIPCC__
#define KE_HD __host__ __device__ __forceinline__
#define KE_HD_STATIC static __host__ __device__ __forceinline__
#else
#define KE_HD static inline
#define KE_HD_STATIC static inline
#endif

enum { KE_JPEG_OK = 0, KE_JPEG_UNSUPPORTED = 1, KE_JPEG_CORRUPT = 2 };

// One Huffman table in decoding form: a 9-bit look-ahead (value = length << 8 | symbol, 0 = longer code or invalid) and the
// canonical arrays of T.81 F.2.2.3 for the longer codes.
struct KeHuffTable {
    uint16_t look[512];
    int32_t maxcode[18];     // maxcode[l] = largest code of length l, -1 if none; maxcode[17] = sentinel
    int32_t valoffset[17];   // huffval index of the first code of length l minus that code
    uint8_t huffval[256];
};

struct KeJpegInfo {
    int32_t status;
    int32_t width, height, ncomp;
    int32_t hs[3], vs[3], tq[3], td[3], ta[3];
    int32_t hmax, vmax;
    int32_t mcus_x, mcus_y;
    int32_t comp_w[3], comp_h[3];      // downsampled size in samples (ceil)
    int32_t plane_w[3], plane_h[3];    // padded to whole MCUs (multiples of 8 * sampling factor)
    int32_t restart_interval;
    uint32_t scan_offset, scan_end;    // entropy-coded bytes [scan_offset, scan_end) inside the file
    int32_t huff_dc[3], huff_ac[3];    // indices into the batch's table pool
    uint16_t quant[3][64];             // per component, natural (row-major) order
    int32_t progressive;               // SOF2: the scans below instead of the one [scan_offset, scan_end)
    int32_t nscans;
    uint32_t first_scan;               // index of the image's first KeJpegScan in the batch's scan list
};

// One scan of a progressive file (T.81 G.1): a band [ss, se] of the zigzag sequence at bit position al (ah: the position the
// previous scan of the band stopped at; 0 = first scan), for one component (any AC scan) or several (DC scans).
struct KeJpegScan {
    uint32_t offset, end;              // entropy-coded bytes [offset, end) inside the file
    int32_t restart_interval;
    int32_t ncomp, comp[3];            // component indices of the frame
    int32_t ss, se, ah, al;
    int32_t dc_tab[3], ac_tab[3];      // per scan component: indices into the batch's table pool (host-side decoding)
    uint32_t dc_dht[3], ac_dht[3];     // ... and where the table's DHT payload (16 counts, then the symbols) lies in the file
};

// zigzag position -> natural (row-major) index
#define KE_ZZ { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, \
                42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 }

// ---- bit reader over the entropy-coded segment (T.81 F.2.2.5): 0xFF00 -> 0xFF, RSTn handled by the caller, any other marker
// ends the data (zeros are fed from there on and `overrun` counts them: libjpeg warns, Pillow raises on truncated files).
struct KeBits {
    const uint8_t *p;
    uint32_t pos, end;
    uint64_t acc;       // bits left-aligned
    int32_t n;          // valid bits in acc
    int32_t overrun;    // zero bytes fed behind the data
    int32_t marker;     // pending marker byte (0 = none)
    int32_t next_rst;   // number the next restart marker must carry (jdmarker.c next_restart_num)
};

KE_HD void ke_bits_init(KeBits &b, const uint8_t *p, uint32_t pos, uint32_t end) {
    b.p = p; b.pos = pos; b.end = end; b.acc = 0; b.n = 0; b.overrun = 0; b.marker = 0; b.next_rst = 0;
}

KE_HD void ke_bits_fill(KeBits &b) {
    while (b.n <= 56) {
        uint32_t byte = 0;
        if (b.marker == 0 && b.pos < b.end) {
            byte = b.p[b.pos];
            if (byte == 0xFF) {
                uint32_t next = b.pos + 1 < b.end ? b.p[b.pos + 1] : 0xD9u;
                if (next == 0) {
                    b.pos += 2;                      // stuffed zero
                } else {
                    b.marker = (int32_t)next;        // RSTn / EOI / anything else: stop here, feed zeros
                    byte = 0;
                    b.overrun += 1;
                }
            } else {
                b.pos += 1;
            }
        } else {
            b.overrun += 1;
        }
        b.acc |= (uint64_t)byte << (56 - b.n);
        b.n += 8;
    }
}

KE_HD uint32_t ke_bits_peek(const KeBits &b, int k) { return (uint32_t)(b.acc >> (64 - k)); }
KE_HD void ke_bits_skip(KeBits &b, int k) { b.acc <<= k; b.n -= k; }

// Whether bits that are not in the file have been consumed: the zero bytes the filler feeds behind the data sit at the tail of
// the buffer, so as long as it still holds 8 * overrun bits none of them was used (a valid stream only ever looks at them).
// libjpeg marks such a segment (insufficient_data: the rest of it decodes to zero blocks, with a warning Pillow drops); the
// decoders here refuse the file instead.
KE_HD bool ke_bits_ran_dry(const KeBits &b) { return b.overrun * 8 > b.n; }

// Restart: drop the partial byte, expect the RSTn marker that the filler stopped at (or that comes next) -- with the number
// that is due: anything else and libjpeg resynchronises by a heuristic of its own (jpeg_resync_to_restart), so the file is
// refused -- and continue behind it.
KE_HD int ke_bits_restart(KeBits &b) {
    if (ke_bits_ran_dry(b)) return KE_JPEG_CORRUPT;
    if (b.marker == 0) {                             // the filler has not reached a marker yet
        // bytes buffered in acc beyond the current byte boundary were not used: rewind to the byte boundary of what was
        // consumed, stepping over stuffed zeros
        const int whole = b.n / 8;                   // whole unread bytes in the buffer
        for (int k = 0; k < whole; ++k) {
            if (b.pos >= 2 && b.p[b.pos - 1] == 0 && b.p[b.pos - 2] == 0xFF) b.pos -= 2;
            else b.pos -= 1;
        }
        // ... and find the marker as jdmarker.c next_marker does: data left over in front of it is skipped (libjpeg warns of
        // extraneous bytes; the pixels do not depend on them).  How far the filler had read ahead must not matter: with the
        // marker already seen the same bytes are dropped with the buffer below.
        uint32_t q = b.pos;
        int found = 0;
        while (q + 1 < b.end) {
            if (b.p[q] != 0xFF) { ++q; continue; }
            if (b.p[q + 1] == 0) { q += 2; continue; }           // a stuffed zero: data
            found = b.p[q + 1];                                     // (0xFF fill bytes in front of a marker: not taken, as in the filler)
            break;
        }
        if (!found) return KE_JPEG_CORRUPT;
        b.pos = q;
        b.marker = found;
    }
    if (b.marker != 0xD0 + b.next_rst) return KE_JPEG_CORRUPT;
    b.next_rst = (b.next_rst + 1) & 7;
    b.pos += 2;
    b.acc = 0; b.n = 0; b.marker = 0; b.overrun = 0;
    return KE_JPEG_OK;
}

// One Huffman symbol; returns -1 for an invalid code.
KE_HD int ke_huff_decode(KeBits &b, const KeHuffTable &t) {
    ke_bits_fill(b);
    const uint32_t look = t.look[ke_bits_peek(b, 9)];
    if (look) {
        ke_bits_skip(b, (int)(look >> 8));
        return (int)(look & 0xFF);
    }
    int32_t code = (int32_t)ke_bits_peek(b, 10);
    int l = 10;
    while (l <= 16 && code > t.maxcode[l]) {
        ++l;
        code = (int32_t)ke_bits_peek(b, l);
    }
    if (l > 16) return -1;
    ke_bits_skip(b, l);
    return t.huffval[(code + t.valoffset[l]) & 0xFF];
}

// `s` more bits as a signed difference / coefficient (F.2.2.1 EXTEND)
KE_HD int ke_receive_extend(KeBits &b, int s) {
    if (s == 0) return 0;
    ke_bits_fill(b);
    const int v = (int)ke_bits_peek(b, s);
    ke_bits_skip(b, s);
    return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
}

// One 8x8 block: coefficients dequantised into `blk` (natural order, int32).  Returns KE_JPEG_OK or KE_JPEG_CORRUPT.
// zz: the zigzag table (KE_ZZ) wherever the caller keeps it.
template <typename Blk>
KE_HD int ke_decode_block(KeBits &b, const KeHuffTable &dc, const KeHuffTable &ac, const uint16_t *quant, const uint8_t *zz,
                          int &dc_pred, Blk &blk) {
    for (int k = 0; k < 64; ++k) blk[k] = 0;
    int s = ke_huff_decode(b, dc);
    if (s < 0 || s > 11) return KE_JPEG_CORRUPT;
    dc_pred += ke_receive_extend(b, s);
    blk[0] = dc_pred * (int)quant[0];
    for (int k = 1; k < 64;) {
        const int rs = ke_huff_decode(b, ac);
        if (rs < 0) return KE_JPEG_CORRUPT;
        const int r = rs >> 4;
        s = rs & 15;
        if (s == 0) {
            if (r != 15) break;                      // end of block
            k += 16;
            continue;
        }
        k += r;
        if (k > 63) return KE_JPEG_CORRUPT;
        const int nat = zz[k];
        blk[nat] = ke_receive_extend(b, s) * (int)quant[nat];
        ++k;
    }
    return KE_JPEG_OK;
}

// ---- progressive scans (T.81 G.1.2, as jdphuff.c decodes them).  Rd is the caller's reader: sym(slot) = one Huffman symbol
// of the scan's table `slot` (-1 = invalid code), bits(k) = k raw bits (k <= 16), bit().  Blk holds the block's 64
// coefficients in natural order (int16 values); zz is the zigzag table.

// first DC scan: the difference is coded as in a sequential file, the coefficient is stored shifted left by al
template <typename Rd>
KE_HD int ke_prog_dc_first(Rd &rd, int slot, int &pred, int al, int *value) {
    const int s = rd.sym(slot);
    if (s < 0 || s > 11) return KE_JPEG_CORRUPT;     // 11 bits of DC difference at most for 8-bit samples
    int diff = 0;
    if (s) {
        const int v = (int)rd.bits(s);
        diff = v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
    }
    pred += diff;
    *value = (int)((uint32_t)pred << al);
    return KE_JPEG_OK;
}

// first scan of an AC band (G.1.2.2): run/size symbols, end-of-band runs that span blocks
template <typename Rd, typename Blk>
KE_HD int ke_prog_ac_first(Rd &rd, int slot, Blk &blk, const uint8_t *zz, int ss, int se, int al, uint32_t &eobrun) {
    if (eobrun > 0) { --eobrun; return KE_JPEG_OK; }
    for (int k = ss; k <= se; ++k) {
        const int rs = rd.sym(slot);
        if (rs < 0) return KE_JPEG_CORRUPT;
        const int r = rs >> 4, s = rs & 15;
        if (s) {
            k += r;
            if (k > 63) return KE_JPEG_CORRUPT;
            const int v = (int)rd.bits(s);
            const int coef = v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
            blk[zz[k]] = (int16_t)((uint32_t)coef << al);
        } else if (r == 15) {
            k += 15;                                 // sixteen zeros
        } else {
            eobrun = 1u << r;                        // this block and eobrun - 1 more end here
            if (r) eobrun += rd.bits(r);
            --eobrun;
            break;
        }
    }
    return KE_JPEG_OK;
}

// refinement scan of an AC band (G.1.2.3): one more bit for every coefficient that is already nonzero, new +-1 << al
// coefficients in between
template <typename Rd, typename Blk>
KE_HD int ke_prog_ac_refine(Rd &rd, int slot, Blk &blk, const uint8_t *zz, int ss, int se, int al, uint32_t &eobrun) {
    const int p1 = 1 << al, m1 = -(1 << al);
    int k = ss;
    if (eobrun == 0) {
        for (; k <= se; ++k) {
            const int rs = rd.sym(slot);
            if (rs < 0) return KE_JPEG_CORRUPT;
            int r = rs >> 4, s = rs & 15;
            if (s) {
                s = rd.bit() ? p1 : m1;              // the size of a new coefficient is always 1; its sign follows
            } else if (r != 15) {
                eobrun = 1u << r;
                if (r) eobrun += rd.bits(r);
                break;                               // the rest of the band belongs to the end-of-band run
            }
            // skip r still-zero coefficients; every nonzero one passed on the way takes a correction bit
            do {
                const int pos = zz[k];
                const int c = blk[pos];
                if (c != 0) {
                    if (rd.bit() && (c & p1) == 0) blk[pos] = (int16_t)(c + (c >= 0 ? p1 : m1));
                } else if (--r < 0) {
                    break;
                }
                ++k;
            } while (k <= se);
            if (s) {
                if (k > 63) return KE_JPEG_CORRUPT;
                blk[zz[k]] = (int16_t)s;
            }
        }
    }
    if (eobrun > 0) {
        for (; k <= se; ++k) {
            const int pos = zz[k];
            const int c = blk[pos];
            if (c != 0 && rd.bit() && (c & p1) == 0) blk[pos] = (int16_t)(c + (c >= 0 ? p1 : m1));
        }
        --eobrun;
    }
    return KE_JPEG_OK;
}

// ---- jidctint.c jpeg_idct_islow on a dequantised block; out: 8 rows of 8 samples at `stride` bytes
#define KE_FIX_0_298631336 2446
#define KE_FIX_0_390180644 3196
#define KE_FIX_0_541196100 4433
#define KE_FIX_0_765366865 6270
#define KE_FIX_0_899976223 7373
#define KE_FIX_1_175875602 9633
#define KE_FIX_1_501321110 12299
#define KE_FIX_1_847759065 15137
#define KE_FIX_1_961570560 16069
#define KE_FIX_2_053119869 16819
#define KE_FIX_2_562915447 20995
#define KE_FIX_3_072711026 25172

KE_HD int ke_descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
KE_HD uint8_t ke_range_limit(int x) {      // libjpeg's range_limit table at offset CENTERJSAMPLE: clamp(x + 128, 0, 255)
    x += 128;
    return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x));
}

// jpeg_idct_islow on one dequantised block.  Returns whether the block is NARROW: every input of the column pass and every
// input of the row pass inside (-2^14, 2^14).  The function follows jidctint.c's C arithmetic (32-bit ints); the library
// Pillow links (libjpeg-turbo with its SIMD routines) dequantises and adds sample pairs in 16 bits and saturates the column
// pass's results to 16 bits, and the two agree exactly as long as nothing leaves 16 bits -- which the bound guarantees (pairs of
// values below 2^14 add within 16 bits).  Files an encoder wrote stay far inside it (a flat white block is 1016 in, 4064
// between the passes); a file with a block outside is one whose pixels depend on the build of libjpeg, and the decoders hand
// it back as KE_JPEG_UNSUPPORTED instead of choosing.  (Seen with quantisation tables overwritten by large steps: the C form
// and Pillow's differ there, tests/test_gpu_jpeg.py::test_files_with_blocks_beyond_the_16_bit_bound_are_handed_back.)
//
// The products go through a policy: under the bound every multiplicand is below 2^16, so the device passes one that uses the
// 24-bit multiplier (a quarter of the 32-bit one's cost on CDNA; ke_jpeg.hip) -- for a block that is not narrow the results
// are not used.
constexpr int kKeIdctBoundBits = 14;

struct KeMulPlain {
    KE_HD_STATIC int mul(int a, int k) { return a * k; }
};

template <typename Blk, typename M = KeMulPlain>
KE_HD bool ke_idct_islow(Blk &blk, uint8_t *out, int stride) {
    int hi = 0, lo = 0;                               // extremes over both passes' inputs (one max3 / min3 per pair on the GPU)
    for (int k = 0; k < 64; ++k) {
        hi = blk[k] > hi ? blk[k] : hi;
        lo = blk[k] < lo ? blk[k] : lo;
    }
    // pass 1: columns, results scaled up by 2 bits, in place
    for (int c = 0; c < 8; ++c) {
        const int i0 = blk[c], i1 = blk[8 + c], i2 = blk[16 + c], i3 = blk[24 + c], i4 = blk[32 + c], i5 = blk[40 + c],
                  i6 = blk[48 + c], i7 = blk[56 + c];
        if ((i1 | i2 | i3 | i4 | i5 | i6 | i7) == 0) {
            const int dcval = i0 * 4;                // << PASS1_BITS
            for (int r = 0; r < 8; ++r) blk[8 * r + c] = dcval;
            continue;
        }
        int z2 = i2, z3 = i6;
        int z1 = M::mul(z2 + z3, KE_FIX_0_541196100);
        int tmp2 = z1 + M::mul(z3, -KE_FIX_1_847759065);
        int tmp3 = z1 + M::mul(z2, KE_FIX_0_765366865);
        z2 = i0; z3 = i4;
        int tmp0 = (z2 + z3) * 8192, tmp1 = (z2 - z3) * 8192;   // << CONST_BITS
        const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = i7; tmp1 = i5; tmp2 = i3; tmp3 = i1;
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int z4 = tmp1 + tmp3;
        const int z5 = M::mul(z3 + z4, KE_FIX_1_175875602);
        tmp0 = M::mul(tmp0, KE_FIX_0_298631336); tmp1 = M::mul(tmp1, KE_FIX_2_053119869);
        tmp2 = M::mul(tmp2, KE_FIX_3_072711026); tmp3 = M::mul(tmp3, KE_FIX_1_501321110);
        z1 = M::mul(z1, -KE_FIX_0_899976223); z2 = M::mul(z2, -KE_FIX_2_562915447);
        z3 = M::mul(z3, -KE_FIX_1_961570560); z4 = M::mul(z4, -KE_FIX_0_390180644);
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        blk[c] = ke_descale(tmp10 + tmp3, 11);       // CONST_BITS - PASS1_BITS
        blk[56 + c] = ke_descale(tmp10 - tmp3, 11);
        blk[8 + c] = ke_descale(tmp11 + tmp2, 11);
        blk[48 + c] = ke_descale(tmp11 - tmp2, 11);
        blk[16 + c] = ke_descale(tmp12 + tmp1, 11);
        blk[40 + c] = ke_descale(tmp12 - tmp1, 11);
        blk[24 + c] = ke_descale(tmp13 + tmp0, 11);
        blk[32 + c] = ke_descale(tmp13 - tmp0, 11);
    }
    for (int k = 0; k < 64; ++k) {
        hi = blk[k] > hi ? blk[k] : hi;
        lo = blk[k] < lo ? blk[k] : lo;
    }
    // pass 2: rows, descale by CONST_BITS + PASS1_BITS + 3, level shift and clamp
    for (int r = 0; r < 8; ++r) {
        const int w0 = blk[8 * r], w1 = blk[8 * r + 1], w2 = blk[8 * r + 2], w3 = blk[8 * r + 3], w4 = blk[8 * r + 4],
                  w5 = blk[8 * r + 5], w6 = blk[8 * r + 6], w7 = blk[8 * r + 7];
        uint8_t *o = out + (long)r * stride;
        if ((w1 | w2 | w3 | w4 | w5 | w6 | w7) == 0) {
            const uint8_t dcval = ke_range_limit(ke_descale(w0, 5));     // PASS1_BITS + 3
            for (int c = 0; c < 8; ++c) o[c] = dcval;
            continue;
        }
        int z2 = w2, z3 = w6;
        int z1 = M::mul(z2 + z3, KE_FIX_0_541196100);
        int tmp2 = z1 + M::mul(z3, -KE_FIX_1_847759065);
        int tmp3 = z1 + M::mul(z2, KE_FIX_0_765366865);
        int tmp0 = (w0 + w4) * 8192, tmp1 = (w0 - w4) * 8192;
        const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w7; tmp1 = w5; tmp2 = w3; tmp3 = w1;
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int z4 = tmp1 + tmp3;
        const int z5 = M::mul(z3 + z4, KE_FIX_1_175875602);
        tmp0 = M::mul(tmp0, KE_FIX_0_298631336); tmp1 = M::mul(tmp1, KE_FIX_2_053119869);
        tmp2 = M::mul(tmp2, KE_FIX_3_072711026); tmp3 = M::mul(tmp3, KE_FIX_1_501321110);
        z1 = M::mul(z1, -KE_FIX_0_899976223); z2 = M::mul(z2, -KE_FIX_2_562915447);
        z3 = M::mul(z3, -KE_FIX_1_961570560); z4 = M::mul(z4, -KE_FIX_0_390180644);
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        o[0] = ke_range_limit(ke_descale(tmp10 + tmp3, 18));
        o[7] = ke_range_limit(ke_descale(tmp10 - tmp3, 18));
        o[1] = ke_range_limit(ke_descale(tmp11 + tmp2, 18));
        o[6] = ke_range_limit(ke_descale(tmp11 - tmp2, 18));
        o[2] = ke_range_limit(ke_descale(tmp12 + tmp1, 18));
        o[5] = ke_range_limit(ke_descale(tmp12 - tmp1, 18));
        o[3] = ke_range_limit(ke_descale(tmp13 + tmp0, 18));
        o[4] = ke_range_limit(ke_descale(tmp13 - tmp0, 18));
    }
    return hi < (1 << kKeIdctBoundBits) && lo > -(1 << kKeIdctBoundBits);
}

// ---- upsampling (jdsample.c) for one output sample (x, y) of a chroma plane with factors (hs, vs) relative to luma
// plane: downsampled component, `pw` bytes per row; cw x ch = its real size (the padded columns / rows beyond are not used:
// libjpeg replicates the last real column and row).
KE_HD int ke_upsample_at(const uint8_t *plane, int pw, int cw, int ch, int hfac, int vfac, int x, int y) {
    if (hfac == 1 && vfac == 1) return plane[(long)y * pw + x];
    if (hfac == 1) {                                 // h1v2_fancy_upsample (4:4:0, e.g. a 4:2:2 file turned by jpegtran): the nearer
        const int r0 = y >> 1;                       // row weighs 3, the one above (even rows, bias 1) or below (odd, bias 2) weighs 1
        int r1 = (y & 1) ? r0 + 1 : r0 - 1;          // -- whatever the component's width (jinit_upsampler asks for no minimum here)
        r1 = r1 < 0 ? 0 : (r1 > ch - 1 ? ch - 1 : r1);
        return (plane[(long)r0 * pw + x] * 3 + plane[(long)r1 * pw + x] + ((y & 1) ? 2 : 1)) >> 2;
    }
    // jinit_upsampler picks the fancy filters only for components more than two samples wide; narrower ones are replicated
    if (cw <= 2) return plane[(long)(y / vfac) * pw + x / hfac];
    if (hfac == 2 && vfac == 1) {                    // h2v1_fancy_upsample
        const uint8_t *row = plane + (long)y * pw;
        const int i = x >> 1;
        if ((x & 1) == 0) {
            if (i == 0) return row[0];
            return (row[i] * 3 + row[i - 1] + 1) >> 2;
        }
        if (i == cw - 1) return row[i];
        return (row[i] * 3 + row[i + 1] + 2) >> 2;
    }
    // h2v2_fancy_upsample: nearer row weighs 3, the other 1 (above for even output rows, below for odd), edges replicated
    const int r0 = y >> 1;
    int r1 = (y & 1) ? r0 + 1 : r0 - 1;
    r1 = r1 < 0 ? 0 : (r1 > ch - 1 ? ch - 1 : r1);
    const uint8_t *in0 = plane + (long)r0 * pw, *in1 = plane + (long)r1 * pw;
    const int i = x >> 1;
    const int cur = in0[i] * 3 + in1[i];
    if ((x & 1) == 0) {
        if (i == 0) return (cur * 4 + 8) >> 4;
        return (cur * 3 + (in0[i - 1] * 3 + in1[i - 1]) + 8) >> 4;
    }
    if (i == cw - 1) return (cur * 4 + 7) >> 4;
    return (cur * 3 + (in0[i + 1] * 3 + in1[i + 1]) + 7) >> 4;
}

// The same for the four output samples x0 .. x0+3 of row y at once (x0 a multiple of 4): the chroma neighbourhood is read
// once.  Must equal ke_upsample_at sample by sample (oracle/keyes_jpeg_cpu.cpp checks it).
KE_HD void ke_upsample4(const uint8_t *plane, int pw, int cw, int ch, int hfac, int vfac, int x0, int y, int *out) {
    if (hfac == 1 && vfac == 2) {                             // 4:4:0
        for (int k = 0; k < 4; ++k) out[k] = ke_upsample_at(plane, pw, cw, ch, 1, 2, x0 + k, y);
        return;
    }
    if (hfac == 1) {
        const uint8_t *row = plane + (long)y * pw + x0;
        for (int k = 0; k < 4; ++k) out[k] = row[k];
        return;
    }
    const int i0 = x0 >> 1;
    if (cw <= 2) {
        const uint8_t *row = plane + (long)(y / vfac) * pw;
        out[0] = out[1] = row[i0];
        out[2] = out[3] = row[i0 + 1];                        // columns beyond the image: the padded plane holds them, unused
        return;
    }
    const int r0 = vfac == 2 ? (y >> 1) : y;
    int r1 = r0;
    if (vfac == 2) {
        r1 = (y & 1) ? r0 + 1 : r0 - 1;
        r1 = r1 < 0 ? 0 : (r1 > ch - 1 ? ch - 1 : r1);
    }
    const uint8_t *in0 = plane + (long)r0 * pw, *in1 = plane + (long)r1 * pw;
    int c[4];
    for (int j = 0; j < 4; ++j) {
        int col = i0 - 1 + j;
        col = col < 0 ? 0 : (col > cw - 1 ? cw - 1 : col);
        c[j] = vfac == 2 ? in0[col] * 3 + in1[col] : in0[col];
    }
    const bool first = i0 == 0, last0 = i0 == cw - 1, last1 = i0 + 1 >= cw - 1;
    if (vfac == 2) {
        out[0] = first ? (c[1] * 4 + 8) >> 4 : (c[1] * 3 + c[0] + 8) >> 4;
        out[1] = last0 ? (c[1] * 4 + 7) >> 4 : (c[1] * 3 + c[2] + 7) >> 4;
        out[2] = (c[2] * 3 + c[1] + 8) >> 4;
        out[3] = last1 ? (c[2] * 4 + 7) >> 4 : (c[2] * 3 + c[3] + 7) >> 4;
    } else {
        out[0] = first ? c[1] : (c[1] * 3 + c[0] + 1) >> 2;
        out[1] = last0 ? c[1] : (c[1] * 3 + c[2] + 2) >> 2;
        out[2] = (c[2] * 3 + c[1] + 1) >> 2;
        out[3] = last1 ? c[2] : (c[2] * 3 + c[3] + 2) >> 2;
    }
}

// ---- jdcolor.c ycc_rgb_convert for one pixel
KE_HD void ke_ycc_to_rgb(int y, int cb, int cr, uint8_t *rgb) {
    const int xb = cb - 128, xr = cr - 128;
    const int r = y + ((91881 * xr + 32768) >> 16);                              // Cr_r_tab
    const int g = y + ((-22554 * xb + 32768 + (-46802) * xr) >> 16);              // Cb_g_tab + Cr_g_tab
    const int b = y + ((116130 * xb + 32768) >> 16);                             // Cb_b_tab
    rgb[0] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    rgb[1] = (uint8_t)(g < 0 ? 0 : (g > 255 ? 255 : g));
    rgb[2] = (uint8_t)(b < 0 ? 0 : (b > 255 ? 255 : b));
}
