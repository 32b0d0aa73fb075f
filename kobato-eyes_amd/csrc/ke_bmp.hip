// ke_bmp.hip -- uncompressed BMP files unpacked on the GPU: the decode step in front of the hash path for the
// third of the formats the reference ranks as keepers (src/dup/scanner.py:16-28).  Replaces `Image.open(path)` + pixel access
// of the reference's batch hasher (src/core/fastsig.py:31-34) for the files ke_bmp_parse.h takes; there is nothing to decode,
// only to move: stored rows are bottom-up and padded, pixels are B, G, R(, X) -- the files travel to the device as they are
// (a BMP is its pixels, so that copy is the batch's traffic) and one kernel writes the packed top-down rows the hash kernels
// read, a palette file as the luma the reference's hashes see (src/sig/phash.py:25).
//
//   ke_bmp_unpack   blockIdx.x = image, blockIdx.y = a band of rows; a thread takes four pixels at a time (16-byte loads and
//                   stores where the format allows).  HBM-bound: bytes in + bytes out, each once.
#include <algorithm>
#include <vector>

#include "ke_bmp_parse.h"
#include "ke_internal.h"

namespace {

struct KeBmpDev {
    uint64_t src;          // the image's pixel data inside the uploaded files
    uint64_t out_off;      // bytes into the caller's pixel buffer
    int32_t width, height, bits, channels, topdown;
    uint32_t stride;
    uint8_t pick[4];
    uint8_t lut[256];
};

constexpr int kRowsPerBlock = 8;      // at least; more for images taller than 65 535 bands of them

__device__ __forceinline__ uint32_t ld4(const uint8_t *p) {       // any alignment
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

__global__ __launch_bounds__(256) void ke_bmp_unpack(const KeBmpDev *__restrict__ imgs, const uint8_t *__restrict__ files,
                                                     uint8_t *__restrict__ out, int rows) {
    __shared__ uint8_t s_lut[256];
    const KeBmpDev &d = imgs[blockIdx.x];
    const int y0 = blockIdx.y * rows;
    if (y0 >= d.height) return;
    const int W = d.width, bits = d.bits, ch = d.channels;
    if (bits == 8) {
        s_lut[threadIdx.x] = d.lut[threadIdx.x];
        __syncthreads();
    }
    const int p0 = d.pick[0], p1 = d.pick[1], p2 = d.pick[2], p3 = d.pick[3];
    const int y1 = min(y0 + rows, d.height);
    const int quads = (W + 3) >> 2;
    for (int y = y0; y < y1; ++y) {
        const uint8_t *row = files + d.src + (size_t)(d.topdown ? y : d.height - 1 - y) * d.stride;
        uint8_t *dst = out + d.out_off + (size_t)y * W * ch;
        for (int q = threadIdx.x; q < quads; q += 256) {
            const int x = 4 * q, n = min(4, W - x);
            if (bits == 8) {
                // a stored row is padded to four bytes, so the load stays inside it
                const uint32_t v = ld4(row + x);
                uint32_t o = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) o |= (uint32_t)s_lut[(v >> (8 * k)) & 255u] << (8 * k);
                if (n == 4) __builtin_memcpy(dst + x, &o, 4);
                else for (int k = 0; k < n; ++k) dst[x + k] = (uint8_t)(o >> (8 * k));
            } else if (bits == 24) {
                if (n == 4) {
                    const uint32_t a = ld4(row + 3 * x), b = ld4(row + 3 * x + 4), c = ld4(row + 3 * x + 8);
                    // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3  ->  R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
                    const uint32_t o0 = ((a >> 16) & 255u) | (a & 0xFF00u) | ((a & 255u) << 16) | ((b >> 8) & 255u) << 24;
                    const uint32_t o1 = (b & 255u) | ((a >> 24) << 8) | ((c & 255u) << 16) | (b & 0xFF000000u);
                    const uint32_t o2 = ((b >> 16) & 255u) | ((c >> 24) << 8) | (c & 0xFF0000u) | ((c >> 8) & 255u) << 24;
                    uint8_t *w = dst + 3 * x;
                    __builtin_memcpy(w, &o0, 4);
                    __builtin_memcpy(w + 4, &o1, 4);
                    __builtin_memcpy(w + 8, &o2, 4);
                } else {
                    for (int k = 0; k < n; ++k) {
                        const uint8_t *s = row + 3 * (x + k);
                        uint8_t *w = dst + 3 * (x + k);
                        w[0] = s[2]; w[1] = s[1]; w[2] = s[0];
                    }
                }
            } else {
                for (int k = 0; k < n; ++k) {
                    const uint32_t v = ld4(row + 4 * (size_t)(x + k));
                    const uint32_t r = (v >> (8 * p0)) & 255u, g = (v >> (8 * p1)) & 255u, b = (v >> (8 * p2)) & 255u;
                    uint8_t *w = dst + (size_t)ch * (x + k);
                    if (ch == 4) {
                        const uint32_t o = r | (g << 8) | (b << 16) | (((v >> (8 * p3)) & 255u) << 24);
                        __builtin_memcpy(w, &o, 4);
                    } else {
                        w[0] = (uint8_t)r; w[1] = (uint8_t)g; w[2] = (uint8_t)b;
                    }
                }
            }
        }
    }
}

}  // namespace

KE_API int ke_bmp_probe(const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n, int32_t *widths,
                        int32_t *heights, int32_t *channels, int32_t *status_out) {
    if (n < 0 || (n > 0 && (!files || !offsets || !sizes || !widths || !heights || !channels || !status_out))) return KE_EINVAL;
    ke_parallel_ranges(n, [=](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; ++i) {
            KeBmpInfo info;
            ke_parse_bmp(files + offsets[i], (size_t)sizes[i], info);
            widths[i] = info.width; heights[i] = info.height; channels[i] = info.channels;
            status_out[i] = info.status;
        }
    });
    return KE_OK;
}

KE_API int ke_bmp_caveats(const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n, int32_t *flags_out) {
    if (n < 0 || (n > 0 && (!files || !offsets || !sizes || !flags_out))) return KE_EINVAL;
    // the format carries no orientation tag; an alpha channel shows in ke_bmp_probe's channel count
    for (int64_t i = 0; i < n; ++i) flags_out[i] = 0;
    return KE_OK;
}

KE_API int ke_bmp_decode(ke_ctx *ctx, const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n,
                         uint8_t *pixels_out, const uint64_t *out_offsets, int32_t *status_out) {
    if (!ctx) return KE_EINVAL;
    if (n < 0 || (n > 0 && (!files || !offsets || !sizes || !pixels_out || !out_offsets || !status_out)))
        return ke_fail(ctx, KE_EINVAL, "NULL argument");
    if (n == 0) return KE_OK;
    if (ke_is_device_ptr(files)) return ke_fail(ctx, KE_EINVAL, "the files' headers are parsed on the host: pass host memory (pinned staging is fine)");
    if (!ke_is_device_ptr(pixels_out)) return ke_fail(ctx, KE_EINVAL, "pixels_out must be device memory");
    for (const void *p : {(const void *)offsets, (const void *)sizes, (const void *)out_offsets, (const void *)status_out})
        if (ke_is_device_ptr(p)) return ke_fail(ctx, KE_EINVAL, "offsets/sizes/status are host arrays");
    KE_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<KeBmpDev> devs;
    devs.reserve((size_t)n);
    uint64_t lo = ~0ull, hi = 0;
    int max_height = 0;
    std::vector<KeBmpInfo> infos((size_t)n);                     // the headers are read on the host's threads
    ke_parallel_ranges(n, [&](int64_t a, int64_t b, int) {
        for (int64_t i = a; i < b; ++i) ke_parse_bmp(files + offsets[i], (size_t)sizes[i], infos[(size_t)i]);
    });
    for (int64_t i = 0; i < n; ++i) {
        const KeBmpInfo &info = infos[(size_t)i];
        status_out[i] = info.status;
        if (info.status != KE_BMP_OK) continue;
        KeBmpDev d;
        d.src = offsets[i] + info.data_off;                          // relative to `files` for now
        d.out_off = out_offsets[i];
        d.width = info.width; d.height = info.height; d.bits = info.bits; d.channels = info.channels; d.topdown = info.topdown;
        d.stride = info.stride;
        std::memcpy(d.pick, info.pick, 4);
        std::memcpy(d.lut, info.lut, 256);
        lo = std::min(lo, d.src);
        hi = std::max(hi, d.src + (uint64_t)info.stride * (uint64_t)info.height);
        max_height = std::max(max_height, info.height);
        devs.push_back(d);
    }
    if (devs.empty()) return KE_OK;
    for (KeBmpDev &d : devs) d.src -= lo;
    // the pixel data (one contiguous range of the caller's buffer) -> device; a 32-bit row may be read up to its last dword only
    void *d_files, *d_imgs;
    KE_TRY(ke_reserve(ctx, KE_BUF_SSIM_IN, (size_t)(hi - lo) + 16, &d_files));
    KE_TRY(ke_reserve(ctx, KE_BUF_META, devs.size() * sizeof(KeBmpDev), &d_imgs));
    ke_time_begin(ctx, KE_T_JPEG);
    KE_HIP(ctx, hipMemcpyAsync(d_files, files + lo, (size_t)(hi - lo), hipMemcpyHostToDevice, ctx->stream));
    KE_HIP(ctx, hipMemcpyAsync(d_imgs, devs.data(), devs.size() * sizeof(KeBmpDev), hipMemcpyHostToDevice, ctx->stream));
    // blockIdx.y covers the tallest image; the blocks beyond a shorter one's rows return at once
    const size_t kMaxX = 1u << 30;
    const int rows = std::max(kRowsPerBlock, (max_height + 65534) / 65535);
    for (size_t first = 0; first < devs.size(); first += kMaxX) {
        const size_t m = std::min(kMaxX, devs.size() - first);
        hipLaunchKernelGGL(ke_bmp_unpack, dim3((unsigned)m, (unsigned)((max_height + rows - 1) / rows)), dim3(256), 0, ctx->stream,
                           (const KeBmpDev *)d_imgs + first, (const uint8_t *)d_files, pixels_out, rows);
    }
    KE_HIP(ctx, hipGetLastError());
    ke_time_end(ctx, KE_T_JPEG);
    KE_HIP(ctx, hipStreamSynchronize(ctx->stream));                  // devs is a host vector; the scratch is reused
    return KE_OK;
}
