// ke_jpeg.hip -- baseline JPEG decoding on the GPU: the decode step in front of the hash path (north-star step 1; SURVEY 8 f2,
// "GPU-side JPEG decode").  Replaces `Image.open(path)` + pixel access of the reference's batch hasher
// (src/core/fastsig.py:31-34) for the files it can take; the pixels are libjpeg's, bit for bit (ke_jpeg_core.h says which
// of its routines are restated), so the hashes are those of the Pillow route.
//
// Three kernels per batch:
//   ke_jpeg_entropy : ONE THREAD PER IMAGE walks the image's entropy-coded segment (Huffman decoding is sequential within a
//                     scan; the parallelism is across the batch -- a 16 384-image batch is one wave per CU, 100 000 images
//                     fill the chip) and writes the non-zero dequantised coefficients into a zero-filled int16 array;
//   ke_jpeg_idct    : one thread per 8x8 block, jpeg_idct_islow, samples into padded component planes;
//   ke_jpeg_colour  : one thread per 4 output pixels: fancy upsampling + YCbCr -> RGB (or the luma plane as it is), packed
//                     8-bit pixels where ke_hash_images / ke_ssim_pairs expect them.
// Compressed bytes cross PCIe (a tenth of the pixels); files the parser refuses (progressive, CMYK, ...) are reported per
// image and stay with Pillow.
#include <algorithm>

#include "ke_internal.h"
#include "ke_jpeg_parse.h"

namespace {

struct KeJpegDev {                 // per image, device side
    KeJpegInfo info;
    uint64_t file_off;             // into the uploaded file bytes
    uint64_t coef_off;             // int16 units into the coefficient array
    uint64_t plane_off[3];         // bytes into the plane array
    uint64_t out_off;              // bytes into the caller's pixel buffer
    int32_t block_base[3];         // first block of each component inside the image's coefficient range
    int32_t blocks_total;
};

__constant__ uint8_t c_zigzag[64] = KE_ZZ;

__global__ __launch_bounds__(64) void ke_jpeg_entropy(const KeJpegDev *__restrict__ imgs, int64_t n, const uint8_t *__restrict__ files,
                                                      const KeHuffTable *__restrict__ tables, int16_t *__restrict__ coefs,
                                                      int32_t *__restrict__ status) {
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const KeJpegDev &d = imgs[i];
    const KeJpegInfo &in = d.info;
    KeBits bits;
    ke_bits_init(bits, files + d.file_off, in.scan_offset, in.scan_end);
    int pred[3] = {0, 0, 0};
    int restart_left = in.restart_interval;
    int rc = KE_JPEG_OK;
    int16_t *base = coefs + d.coef_off;
    for (int my = 0; my < in.mcus_y && rc == KE_JPEG_OK; ++my) {
        for (int mx = 0; mx < in.mcus_x && rc == KE_JPEG_OK; ++mx) {
            if (in.restart_interval && restart_left == 0) {
                if (ke_bits_restart(bits) != KE_JPEG_OK) { rc = KE_JPEG_CORRUPT; break; }
                pred[0] = pred[1] = pred[2] = 0;
                restart_left = in.restart_interval;
            }
            for (int c = 0; c < in.ncomp && rc == KE_JPEG_OK; ++c) {
                const KeHuffTable &dc = tables[in.huff_dc[c]], &ac = tables[in.huff_ac[c]];
                const uint16_t *q = in.quant[c];
                const int bpr = in.plane_w[c] >> 3;                       // blocks per row of this component
                for (int by = 0; by < in.vs[c] && rc == KE_JPEG_OK; ++by) {
                    for (int bx = 0; bx < in.hs[c]; ++bx) {
                        int16_t *blk = base + ((size_t)d.block_base[c] + (size_t)(my * in.vs[c] + by) * bpr + (mx * in.hs[c] + bx)) * 64;
                        int s = ke_huff_decode(bits, dc);
                        if (s < 0 || s > 11) { rc = KE_JPEG_CORRUPT; break; }
                        pred[c] += ke_receive_extend(bits, s);
                        blk[0] = (int16_t)(pred[c] * (int)q[0]);
                        for (int k = 1; k < 64;) {
                            const int rs = ke_huff_decode(bits, ac);
                            if (rs < 0) { rc = KE_JPEG_CORRUPT; break; }
                            const int r = rs >> 4;
                            s = rs & 15;
                            if (s == 0) {
                                if (r != 15) break;
                                k += 16;
                                continue;
                            }
                            k += r;
                            if (k > 63) { rc = KE_JPEG_CORRUPT; break; }
                            const int nat = c_zigzag[k];
                            blk[nat] = (int16_t)(ke_receive_extend(bits, s) * (int)q[nat]);
                            ++k;
                        }
                        if (rc != KE_JPEG_OK) break;
                    }
                }
            }
            --restart_left;
        }
    }
    if (rc == KE_JPEG_OK && bits.overrun > 8) rc = KE_JPEG_CORRUPT;     // ran past the data: truncated file (Pillow raises)
    status[i] = rc;
}

__global__ __launch_bounds__(256) void ke_jpeg_idct(const KeJpegDev *__restrict__ imgs, const int16_t *__restrict__ coefs,
                                                    uint8_t *__restrict__ planes) {
    const KeJpegDev &d = imgs[blockIdx.y];
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= d.blocks_total) return;
    const KeJpegInfo &in = d.info;
    int c = 0;
    if (in.ncomp == 3) c = b >= d.block_base[2] ? 2 : (b >= d.block_base[1] ? 1 : 0);
    const int local = b - d.block_base[c];
    const int bpr = in.plane_w[c] >> 3;
    const int brow = local / bpr, bcol = local - brow * bpr;
    const int16_t *src = coefs + d.coef_off + (size_t)b * 64;
    int blk[64];
#pragma unroll
    for (int k = 0; k < 8; ++k) {                                        // 128 bytes per block: eight 16-byte loads
        const int4 v = reinterpret_cast<const int4 *>(src)[k];
        const int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            blk[8 * k + 2 * j] = (int)(int16_t)(w[j] & 0xFFFF);
            blk[8 * k + 2 * j + 1] = w[j] >> 16;
        }
    }
    uint8_t rows[64];
    ke_idct_islow(blk, rows, 8);
    uint8_t *dst = planes + d.plane_off[c] + (size_t)(brow * 8) * in.plane_w[c] + bcol * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        uint2 v;
        v.x = rows[8 * r] | (rows[8 * r + 1] << 8) | (rows[8 * r + 2] << 16) | ((uint32_t)rows[8 * r + 3] << 24);
        v.y = rows[8 * r + 4] | (rows[8 * r + 5] << 8) | (rows[8 * r + 6] << 16) | ((uint32_t)rows[8 * r + 7] << 24);
        *reinterpret_cast<uint2 *>(dst + (size_t)r * in.plane_w[c]) = v;
    }
}

__global__ __launch_bounds__(256) void ke_jpeg_colour(const KeJpegDev *__restrict__ imgs, const uint8_t *__restrict__ planes,
                                                      uint8_t *__restrict__ out) {
    const KeJpegDev &d = imgs[blockIdx.y];
    const KeJpegInfo &in = d.info;
    const int quads_per_row = (in.width + 3) >> 2;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= quads_per_row * in.height) return;
    const int y = q / quads_per_row, x0 = (q - y * quads_per_row) * 4;
    const uint8_t *Y = planes + d.plane_off[0];
    uint8_t *dst = out + d.out_off;
    for (int x = x0; x < min(x0 + 4, in.width); ++x) {
        const int yy = Y[(size_t)y * in.plane_w[0] + x];
        if (in.ncomp == 1) {
            dst[(size_t)y * in.width + x] = (uint8_t)yy;
        } else {
            const int cb = ke_upsample_at(planes + d.plane_off[1], in.plane_w[1], in.comp_w[1], in.comp_h[1], in.hmax, in.vmax, x, y);
            const int cr = ke_upsample_at(planes + d.plane_off[2], in.plane_w[2], in.comp_w[2], in.comp_h[2], in.hmax, in.vmax, x, y);
            ke_ycc_to_rgb(yy, cb, cr, dst + ((size_t)y * in.width + x) * 3);
        }
    }
}

}  // namespace

KE_API int ke_jpeg_probe(const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n, int32_t *widths,
                         int32_t *heights, int32_t *channels, int32_t *status_out) {
    if (n < 0 || (n > 0 && (!files || !offsets || !sizes || !widths || !heights || !channels || !status_out))) return KE_EINVAL;
    KeJpegTables tables;
    for (int64_t i = 0; i < n; ++i) {
        KeJpegInfo info;
        ke_parse_jpeg(files + offsets[i], (size_t)sizes[i], tables, info);
        widths[i] = info.width; heights[i] = info.height; channels[i] = info.ncomp;
        status_out[i] = info.status;
        if (tables.pool.size() > 64) { tables.pool.clear(); tables.keys.clear(); }
    }
    return KE_OK;
}

KE_API int ke_jpeg_decode(ke_ctx *ctx, const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n,
                          uint8_t *pixels_out, const uint64_t *out_offsets, int32_t *status_out) {
    if (!ctx) return KE_EINVAL;
    if (n < 0 || (n > 0 && (!files || !offsets || !sizes || !pixels_out || !out_offsets || !status_out)))
        return ke_fail(ctx, KE_EINVAL, "NULL argument");
    if (n == 0) return KE_OK;
    if (ke_is_device_ptr(files)) return ke_fail(ctx, KE_EINVAL, "compressed files are parsed on the host: pass host memory (pinned staging is fine)");
    if (!ke_is_device_ptr(pixels_out)) return ke_fail(ctx, KE_EINVAL, "pixels_out must be device memory");
    for (const void *p : {(const void *)offsets, (const void *)sizes, (const void *)out_offsets, (const void *)status_out})
        if (ke_is_device_ptr(p)) return ke_fail(ctx, KE_EINVAL, "offsets/sizes/status are host arrays");
    KE_HIP(ctx, hipSetDevice(ctx->device));
    // ---- host: headers, Huffman tables, geometry
    KeJpegTables tables;
    std::vector<KeJpegDev> devs;
    std::vector<int64_t> which;                       // batch position of every decodable image
    devs.reserve((size_t)n);
    uint64_t lo = ~0ull, hi = 0;
    for (int64_t i = 0; i < n; ++i) {
        KeJpegDev d;
        ke_parse_jpeg(files + offsets[i], (size_t)sizes[i], tables, d.info);
        status_out[i] = d.info.status;
        if (d.info.status != KE_JPEG_OK) continue;
        d.file_off = offsets[i];
        d.out_off = out_offsets[i];
        lo = std::min(lo, offsets[i]);
        hi = std::max(hi, offsets[i] + sizes[i]);
        devs.push_back(d);
        which.push_back(i);
    }
    if (devs.empty()) return KE_OK;
    // compressed bytes of the decodable files (one contiguous range of the caller's buffer) -> device
    void *d_files;
    KE_TRY(ke_reserve(ctx, KE_BUF_PIXELS, (size_t)(hi - lo) + 64, &d_files));
    KE_HIP(ctx, hipMemcpyAsync(d_files, files + lo, (size_t)(hi - lo), hipMemcpyHostToDevice, ctx->stream));
    void *d_tables;
    KE_TRY(ke_reserve(ctx, KE_BUF_JPEG_TABLES, tables.pool.size() * sizeof(KeHuffTable), &d_tables));
    KE_HIP(ctx, hipMemcpyAsync(d_tables, tables.pool.data(), tables.pool.size() * sizeof(KeHuffTable), hipMemcpyHostToDevice, ctx->stream));
    // sub-batches bounded by scratch: coefficients (2 B per sample) + planes (1 B per sample)
    const uint64_t budget = (uint64_t)6 << 30;
    size_t first = 0;
    std::vector<int32_t> st;
    ke_time_begin(ctx, KE_T_JPEG);
    while (first < devs.size()) {
        uint64_t coef_units = 0, plane_bytes = 0;
        size_t last = first;
        int max_blocks = 0, max_quads = 0;
        while (last < devs.size() && last - first < 65535) {
            KeJpegDev &d = devs[last];
            uint64_t blocks = 0, pl = 0;
            for (int c = 0; c < d.info.ncomp; ++c) {
                blocks += (uint64_t)(d.info.plane_w[c] >> 3) * (d.info.plane_h[c] >> 3);
                pl += (uint64_t)d.info.plane_w[c] * d.info.plane_h[c];
            }
            if (last > first && (coef_units + blocks * 64) * 2 + plane_bytes + pl > budget) break;
            d.file_off -= lo;
            d.coef_off = coef_units;
            uint64_t b = 0;
            for (int c = 0; c < 3; ++c) {
                d.block_base[c] = (int32_t)b;
                d.plane_off[c] = plane_bytes;
                if (c < d.info.ncomp) {
                    b += (uint64_t)(d.info.plane_w[c] >> 3) * (d.info.plane_h[c] >> 3);
                    plane_bytes += ((uint64_t)d.info.plane_w[c] * d.info.plane_h[c] + 15) & ~15ull;
                }
            }
            d.blocks_total = (int32_t)blocks;
            coef_units += blocks * 64;
            max_blocks = std::max(max_blocks, (int)blocks);
            max_quads = std::max(max_quads, ((d.info.width + 3) >> 2) * d.info.height);
            ++last;
        }
        const int64_t m = (int64_t)(last - first);
        void *d_imgs, *d_coef, *d_planes, *d_status;
        KE_TRY(ke_reserve(ctx, KE_BUF_META, (size_t)m * sizeof(KeJpegDev), &d_imgs));
        KE_TRY(ke_reserve(ctx, KE_BUF_TMP, (size_t)coef_units * 2 + 64, &d_coef));
        KE_TRY(ke_reserve(ctx, KE_BUF_SSIM_IN, (size_t)plane_bytes + 64, &d_planes));
        KE_TRY(ke_reserve(ctx, KE_BUF_OUT0, (size_t)m * 4, &d_status));
        KE_HIP(ctx, hipMemcpyAsync(d_imgs, devs.data() + first, (size_t)m * sizeof(KeJpegDev), hipMemcpyHostToDevice, ctx->stream));
        KE_HIP(ctx, hipMemsetAsync(d_coef, 0, (size_t)coef_units * 2, ctx->stream));
        hipLaunchKernelGGL(ke_jpeg_entropy, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, ctx->stream, (const KeJpegDev *)d_imgs, m,
                           (const uint8_t *)d_files, (const KeHuffTable *)d_tables, (int16_t *)d_coef, (int32_t *)d_status);
        hipLaunchKernelGGL(ke_jpeg_idct, dim3((unsigned)((max_blocks + 255) / 256), (unsigned)m), dim3(256), 0, ctx->stream,
                           (const KeJpegDev *)d_imgs, (const int16_t *)d_coef, (uint8_t *)d_planes);
        hipLaunchKernelGGL(ke_jpeg_colour, dim3((unsigned)((max_quads + 255) / 256), (unsigned)m), dim3(256), 0, ctx->stream,
                           (const KeJpegDev *)d_imgs, (const uint8_t *)d_planes, pixels_out);
        KE_HIP(ctx, hipGetLastError());
        st.resize((size_t)m);
        KE_HIP(ctx, hipMemcpyAsync(st.data(), d_status, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream));
        KE_HIP(ctx, hipStreamSynchronize(ctx->stream));                  // devs / st are host vectors; scratch is reused
        for (int64_t k = 0; k < m; ++k) status_out[which[first + (size_t)k]] = st[(size_t)k];
        first = last;
    }
    ke_time_end(ctx, KE_T_JPEG);
    return KE_OK;
}
