// ke_tiff.hip -- uncompressed 8-bit TIFF files unpacked on the GPU: the decode step in front of the hash path for
// the fifth of the formats the reference ranks as keepers (src/dup/scanner.py:16-28).  Replaces `Image.open(path)` + pixel access
// of the reference's batch hasher (src/core/fastsig.py:31-34) for the files ke_tiff_parse.h takes -- those Pillow opens with its
// own "raw" decoder; every compressed TIFF is libtiff's there and stays with Pillow.  As for BMP there is nothing to decode:
// the files travel to the device as they are and one kernel gathers the strips into packed rows -- gray and RGB(A) as stored,
// an unspecified fourth sample dropped, WhiteIsZero inverted, a palette file as the luma the reference's hashes see.
//
//   ke_tiff_unpack   blockIdx.x = image, blockIdx.y = a band of rows; row y lies in strip y / rows_per_strip.  HBM-bound: bytes in
//                    + bytes out, each once.
#include <algorithm>
#include <vector>

#include "ke_internal.h"
#include "ke_tiff_parse.h"

namespace {

struct KeTiffDev {
    uint64_t out_off;      // bytes into the caller's pixel buffer
    uint64_t strip_first;  // index of the image's first strip in the batch's list of strip positions
    int32_t width, height, spp, channels, mapped, rows_per_strip;
    uint8_t lut[256];
};

constexpr int kRowsPerBlock = 8;      // at least; more for images taller than 65 535 bands of them

__global__ __launch_bounds__(256) void ke_tiff_unpack(const KeTiffDev *__restrict__ imgs, const uint64_t *__restrict__ strips,
                                                      const uint8_t *__restrict__ files, uint8_t *__restrict__ out, int rows) {
    __shared__ uint8_t s_lut[256];
    const KeTiffDev &d = imgs[blockIdx.x];
    const int y0 = blockIdx.y * rows;
    if (y0 >= d.height) return;
    const int W = d.width, spp = d.spp, ch = d.channels;
    if (d.mapped) {
        s_lut[threadIdx.x] = d.lut[threadIdx.x];
        __syncthreads();
    }
    const int y1 = min(y0 + rows, d.height);
    for (int y = y0; y < y1; ++y) {
        const uint8_t *row = files + strips[d.strip_first + (uint64_t)(y / d.rows_per_strip)] + (size_t)(y % d.rows_per_strip) * W * spp;
        uint8_t *dst = out + d.out_off + (size_t)y * W * ch;
        if (spp == ch) {                                       // the row as it is (through the table for mapped files: one sample per pixel)
            const int n = W * ch;
            for (int k = threadIdx.x; k < n; k += 256) dst[k] = d.mapped ? s_lut[row[k]] : row[k];
        } else {                                               // four samples stored, three leave
            for (int x = threadIdx.x; x < W; x += 256) {
                uint32_t v;
                __builtin_memcpy(&v, row + 4 * (size_t)x, 4);
                uint8_t *w = dst + 3 * (size_t)x;
                w[0] = (uint8_t)v; w[1] = (uint8_t)(v >> 8); w[2] = (uint8_t)(v >> 16);
            }
        }
    }
}

}  // namespace

KE_API int ke_tiff_probe(const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n, int32_t *widths,
                         int32_t *heights, int32_t *channels, int32_t *status_out) {
    if (n < 0 || (n > 0 && (!files || !offsets || !sizes || !widths || !heights || !channels || !status_out))) return KE_EINVAL;
    ke_parallel_ranges(n, [=](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; ++i) {
            KeTiffInfo info;
            ke_parse_tiff(files + offsets[i], (size_t)sizes[i], nullptr, info);
            widths[i] = info.width; heights[i] = info.height; channels[i] = info.channels;
            status_out[i] = info.status;
        }
    });
    return KE_OK;
}

KE_API int ke_tiff_caveats(const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n, int32_t *flags_out) {
    if (n < 0 || (n > 0 && (!files || !offsets || !sizes || !flags_out))) return KE_EINVAL;
    // files that carry an orientation (the tag, an EXIF directory, an XMP packet) are refused by the parser: Pillow turns them
    for (int64_t i = 0; i < n; ++i) flags_out[i] = 0;
    return KE_OK;
}

KE_API int ke_tiff_decode(ke_ctx *ctx, const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n,
                          uint8_t *pixels_out, const uint64_t *out_offsets, int32_t *status_out) {
    if (!ctx) return KE_EINVAL;
    if (n < 0 || (n > 0 && (!files || !offsets || !sizes || !pixels_out || !out_offsets || !status_out)))
        return ke_fail(ctx, KE_EINVAL, "NULL argument");
    if (n == 0) return KE_OK;
    if (ke_is_device_ptr(files)) return ke_fail(ctx, KE_EINVAL, "the files' directories are parsed on the host: pass host memory (pinned staging is fine)");
    if (!ke_is_device_ptr(pixels_out)) return ke_fail(ctx, KE_EINVAL, "pixels_out must be device memory");
    for (const void *p : {(const void *)offsets, (const void *)sizes, (const void *)out_offsets, (const void *)status_out})
        if (ke_is_device_ptr(p)) return ke_fail(ctx, KE_EINVAL, "offsets/sizes/status are host arrays");
    KE_HIP(ctx, hipSetDevice(ctx->device));
    // directories on the host's threads, each part with its own list of strips
    struct Part {
        std::vector<KeTiffInfo> infos;
        std::vector<uint32_t> strips;
        std::vector<size_t> first;
        int64_t lo = 0;
    };
    std::vector<Part> parts(16);
    const int nparts = ke_parallel_ranges(n, [&](int64_t a, int64_t b, int t) {
        Part &p = parts[(size_t)t];
        p.lo = a;
        p.infos.resize((size_t)(b - a));
        p.first.resize((size_t)(b - a));
        for (int64_t i = a; i < b; ++i) {
            p.first[(size_t)(i - a)] = p.strips.size();
            ke_parse_tiff(files + offsets[i], (size_t)sizes[i], &p.strips, p.infos[(size_t)(i - a)]);
        }
    });
    std::vector<KeTiffDev> devs;
    std::vector<uint64_t> strips;          // positions relative to `files` for now
    uint64_t lo = ~0ull, hi = 0;
    int max_height = 0;
    for (int t = 0; t < nparts; ++t) {
        const Part &p = parts[(size_t)t];
        for (size_t k = 0; k < p.infos.size(); ++k) {
            const KeTiffInfo &info = p.infos[k];
            const int64_t i = p.lo + (int64_t)k;
            status_out[i] = info.status;
            if (info.status != KE_TIFF_OK) continue;
            KeTiffDev d;
            d.out_off = out_offsets[i];
            d.strip_first = strips.size();
            d.width = info.width; d.height = info.height; d.spp = info.spp; d.channels = info.channels; d.mapped = info.mapped;
            d.rows_per_strip = info.rows_per_strip;
            std::memcpy(d.lut, info.lut, 256);
            for (int s = 0; s < info.nstrips; ++s) strips.push_back(offsets[i] + p.strips[p.first[k] + (size_t)s]);
            lo = std::min(lo, offsets[i]);
            hi = std::max(hi, offsets[i] + sizes[i]);
            max_height = std::max(max_height, info.height);
            devs.push_back(d);
        }
    }
    if (devs.empty()) return KE_OK;
    for (uint64_t &s : strips) s -= lo;
    void *d_files, *d_imgs, *d_strips;
    KE_TRY(ke_reserve(ctx, KE_BUF_SSIM_IN, (size_t)(hi - lo) + 16, &d_files));
    KE_TRY(ke_reserve(ctx, KE_BUF_META, devs.size() * sizeof(KeTiffDev), &d_imgs));
    KE_TRY(ke_reserve(ctx, KE_BUF_SSIM_AUX, strips.size() * 8, &d_strips));
    ke_time_begin(ctx, KE_T_JPEG);
    KE_HIP(ctx, hipMemcpyAsync(d_files, files + lo, (size_t)(hi - lo), hipMemcpyHostToDevice, ctx->stream));
    KE_HIP(ctx, hipMemcpyAsync(d_imgs, devs.data(), devs.size() * sizeof(KeTiffDev), hipMemcpyHostToDevice, ctx->stream));
    KE_HIP(ctx, hipMemcpyAsync(d_strips, strips.data(), strips.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    const size_t kMaxX = 1u << 30;
    const int rows = std::max(kRowsPerBlock, (max_height + 65534) / 65535);
    for (size_t first = 0; first < devs.size(); first += kMaxX) {
        const size_t m = std::min(kMaxX, devs.size() - first);
        hipLaunchKernelGGL(ke_tiff_unpack, dim3((unsigned)m, (unsigned)((max_height + rows - 1) / rows)), dim3(256), 0, ctx->stream,
                           (const KeTiffDev *)d_imgs + first, (const uint64_t *)d_strips, (const uint8_t *)d_files, pixels_out, rows);
    }
    KE_HIP(ctx, hipGetLastError());
    ke_time_end(ctx, KE_T_JPEG);
    KE_HIP(ctx, hipStreamSynchronize(ctx->stream));                  // devs / strips are host vectors; the scratch is reused
    return KE_OK;
}
