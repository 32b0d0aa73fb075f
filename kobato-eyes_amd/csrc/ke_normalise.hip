// ke_normalise.hip -- what the reference's defensive loader does to decoded pixels before the refine stage sees them, for images that
// are already on the device (GPU-decoded JPEG / PNG files):
//   * ImageOps.exif_transpose (src/utils/image_io.py:116-120 via _exif_transpose_safely; src/ui/dup_refine_parallel.py:67-70): the
//     EXIF orientation 2..8 applied as Pillow's Image.transpose does -- FLIP_LEFT_RIGHT, ROTATE_180, FLIP_TOP_BOTTOM, TRANSPOSE,
//     ROTATE_270, TRANSVERSE, ROTATE_90 -- a pure permutation of the pixels;
//   * _convert_image_to_rgb for RGBA input (src/utils/image_io.py:137-151): Image.alpha_composite over an opaque white
//     background, then convert("RGB").  Pillow's AlphaComposite.c in integers: with a destination alpha of 255 the two
//     coefficients are sa * 128 and 255 * 128 - sa * 128, the channel is (c * coef1 + 255 * coef2 + 0x4000) divided by 255
//     through ((t >> 8) + t) >> 8 and shifted down by 7; a source alpha of 0 leaves the background.
//   * img.thumbnail((4096, 4096), LANCZOS) for a side over 4096 (src/utils/image_io.py:122-124): Pillow's two-pass resample
//     works on every band of an RGB image with the arithmetic it uses for a single-band one, so the image is split into three
//     planes, each goes through the resampler that serves the hashes (ke_launch_resize_group: bit-exact with Pillow's
//     integer resample), and the three results are interleaved again (ke_thumbnail_rgb).
// Without this, every phone photograph (they all carry an orientation tag) and every RGBA PNG left the GPU route of the refine
// stage and was decoded a second time by Pillow on the host.
#include "ke_internal.h"

namespace {

struct NormItem {
    uint64_t src_off, dst_off;
    int32_t w, h;            // source size
    int32_t channels;        // 3 or 4 (source)
    int32_t orientation;     // 1..8
};

__global__ __launch_bounds__(256) void ke_normalise_kernel(const uint8_t *__restrict__ src, const NormItem *__restrict__ items,
                                                           int64_t first, uint8_t *__restrict__ dst) {
    const NormItem it = items[first + blockIdx.y];
    const int o = it.orientation, w = it.w, h = it.h;
    const int dw = o >= 5 ? h : w, dh = o >= 5 ? w : h;
    const int64_t total = (int64_t)dw * dh;
    const uint8_t *s = src + it.src_off;
    uint8_t *d = dst + it.dst_off;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int y = (int)(e / dw), x = (int)(e - (int64_t)y * dw);
        int sx = x, sy = y;
        switch (o) {
            case 2: sx = w - 1 - x; break;
            case 3: sx = w - 1 - x; sy = h - 1 - y; break;
            case 4: sy = h - 1 - y; break;
            case 5: sx = y; sy = x; break;
            case 6: sx = y; sy = h - 1 - x; break;
            case 7: sx = w - 1 - y; sy = h - 1 - x; break;
            case 8: sx = w - 1 - y; sy = x; break;
            default: break;
        }
        const uint8_t *p = s + ((size_t)sy * w + sx) * it.channels;
        uint32_t r = p[0], g = p[1], b = p[2];
        if (it.channels == 4) {
            const uint32_t sa = p[3];
            if (sa == 0) {
                r = g = b = 255;
            } else {
                const uint32_t c1 = sa * 128u, c2 = 255u * 128u - c1;
                auto over = [&](uint32_t c) {
                    const uint32_t t = c * c1 + 255u * c2 + (0x80u << 7);
                    return ((((t >> 8) + t) >> 8) >> 7);
                };
                r = over(r); g = over(g); b = over(b);
            }
        }
        uint8_t *q = d + (size_t)e * 3;
        q[0] = (uint8_t)r; q[1] = (uint8_t)g; q[2] = (uint8_t)b;
    }
}

// interleaved RGB <-> three planes
__global__ __launch_bounds__(256) void ke_split_rgb(const uint8_t *__restrict__ src, int64_t pixels, uint8_t *__restrict__ planes) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < pixels; e += (int64_t)gridDim.x * 256) {
        const uint8_t *p = src + 3 * e;
        planes[e] = p[0];
        planes[pixels + e] = p[1];
        planes[2 * pixels + e] = p[2];
    }
}

__global__ __launch_bounds__(256) void ke_merge_rgb(const uint8_t *__restrict__ planes, int64_t pixels, uint8_t *__restrict__ dst) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < pixels; e += (int64_t)gridDim.x * 256) {
        uint8_t *q = dst + 3 * e;
        q[0] = planes[e];
        q[1] = planes[pixels + e];
        q[2] = planes[2 * pixels + e];
    }
}

}  // namespace

KE_API int ke_thumbnail_rgb(ke_ctx *ctx, const uint8_t *src, int32_t width, int32_t height, int32_t out_w, int32_t out_h, int32_t filter,
                            uint8_t *dst) {
    if (!ctx) return KE_EINVAL;
    if (!src || !dst) return ke_fail(ctx, KE_EINVAL, "NULL argument");
    if (width <= 0 || height <= 0 || out_w <= 0 || out_h <= 0) return ke_fail(ctx, KE_EINVAL, "sizes must be positive");
    if (filter != KE_FILTER_LANCZOS && filter != KE_FILTER_BILINEAR && filter != KE_FILTER_BICUBIC)
        return ke_fail(ctx, KE_EINVAL, "unknown filter %d", filter);
    if (!ke_is_device_ptr(src) || !ke_is_device_ptr(dst)) return ke_fail(ctx, KE_EINVAL, "src and dst are device memory");
    KE_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t in_px = (int64_t)width * height, out_px = (int64_t)out_w * out_h;
    void *planes_in, *planes_out;
    KE_TRY(ke_reserve(ctx, KE_BUF_SSIM_AUX, (size_t)(3 * in_px) + 64, &planes_in));
    KE_TRY(ke_reserve(ctx, KE_BUF_SSIM_IN, (size_t)(3 * out_px) + 64, &planes_out));
    const unsigned gin = (unsigned)std::min<int64_t>((in_px + 255) / 256, 1 << 16), gout = (unsigned)std::min<int64_t>((out_px + 255) / 256, 1 << 16);
    hipLaunchKernelGGL(ke_split_rgb, dim3(gin), dim3(256), 0, ctx->stream, src, in_px, (uint8_t *)planes_in);
    KeHashGroup g{(const uint8_t *)planes_in, nullptr, (uint64_t)in_px, nullptr, 3, width, height, 1};
    KE_TRY(ke_launch_resize_group(ctx, g, out_w, out_h, filter, (uint8_t *)planes_out, nullptr));
    hipLaunchKernelGGL(ke_merge_rgb, dim3(gout), dim3(256), 0, ctx->stream, (const uint8_t *)planes_out, out_px, dst);
    KE_HIP(ctx, hipGetLastError());
    KE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KE_OK;
}

KE_API int ke_normalise_rgb(ke_ctx *ctx, const uint8_t *src, const uint64_t *src_offsets, const int32_t *widths, const int32_t *heights,
                            const int32_t *channels, const int32_t *orientations, int64_t n, uint8_t *dst, const uint64_t *dst_offsets) {
    if (!ctx) return KE_EINVAL;
    if (n < 0 || (n > 0 && (!src || !src_offsets || !widths || !heights || !channels || !orientations || !dst || !dst_offsets)))
        return ke_fail(ctx, KE_EINVAL, "NULL argument");
    if (n == 0) return KE_OK;
    if (!ke_is_device_ptr(src) || !ke_is_device_ptr(dst)) return ke_fail(ctx, KE_EINVAL, "src and dst are device memory");
    KE_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<NormItem> items((size_t)n);
    int64_t largest = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (widths[i] <= 0 || heights[i] <= 0 || (channels[i] != 3 && channels[i] != 4) || orientations[i] < 1 || orientations[i] > 8)
            return ke_fail(ctx, KE_EINVAL, "image %lld: %dx%d, %d channels, orientation %d", (long long)i, widths[i], heights[i], channels[i],
                           orientations[i]);
        items[(size_t)i] = NormItem{src_offsets[i], dst_offsets[i], widths[i], heights[i], channels[i], orientations[i]};
        largest = std::max(largest, (int64_t)widths[i] * heights[i]);
    }
    void *d_items;
    KE_TRY(ke_reserve(ctx, KE_BUF_META, (size_t)n * sizeof(NormItem), &d_items));
    KE_HIP(ctx, hipMemcpyAsync(d_items, items.data(), (size_t)n * sizeof(NormItem), hipMemcpyHostToDevice, ctx->stream));
    const unsigned gx = (unsigned)std::min<int64_t>((largest + 255) / 256, 4096);
    for (int64_t first = 0; first < n; first += 65535) {
        const unsigned gy = (unsigned)std::min<int64_t>(65535, n - first);
        hipLaunchKernelGGL(ke_normalise_kernel, dim3(gx, gy), dim3(256), 0, ctx->stream, src, (const NormItem *)d_items, first, dst);
    }
    KE_HIP(ctx, hipGetLastError());
    KE_HIP(ctx, hipStreamSynchronize(ctx->stream));       // `items` is a host vector
    return KE_OK;
}
