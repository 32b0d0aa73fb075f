// ke_png.hip -- PNG decoding on the GPU (8-bit grayscale / RGB / RGBA, no interlacing): the other half of the decode step in
// front of the hash path.  Replaces `Image.open(path)` + pixel access of the reference's batch hasher
// (src/core/fastsig.py:31-34) for the files it takes; the format is lossless, so the pixels are Pillow's by construction of
// the two specifications followed in ke_png_core.h (zlib/deflate, the five scanline filters).
//
// Deflate is one sequential bit stream per image and the filters chain every byte to its left and upper neighbours, so, as
// for JPEG, the parallelism is across the batch: ONE THREAD PER IMAGE inflates (canonical Huffman tables and a stream window in
// LDS, output and LZ77 back-references in HBM), a second kernel -- again one thread per image -- undoes the filters row by
// row.  Sequential work per image is ~25x that of a JPEG of the same size (every byte is a symbol), which bounds the rate;
// it still beats the host cores by an order of magnitude once thousands of files are in flight.
#include <algorithm>

#include "ke_internal.h"
#include "ke_png_parse.h"

namespace {

struct KePngDev {
    KePngInfo info;
    uint64_t raw_off;      // filtered scanlines (height * (1 + width * channels) bytes) inside the scratch
    uint64_t out_off;      // bytes into the caller's pixel buffer
};

constexpr int kPngWin = 64, kPngWinPitch = kPngWin + 4;

struct LdsStream {         // the compressed bytes through a 64-byte window per lane in LDS
    const uint8_t *z;      // global: this image's zlib stream
    uint8_t *win;
    uint32_t win_pos;
    __device__ __forceinline__ void load(uint32_t pos) {
        win_pos = pos & ~3u;
        const uint8_t *src = z + win_pos;
#pragma unroll
        for (int k = 0; k < kPngWin / 4; ++k) {
            uint32_t w;
            __builtin_memcpy(&w, src + 4 * k, 4);
            reinterpret_cast<uint32_t *>(win)[k] = w;
        }
    }
    __device__ __forceinline__ uint32_t byte(uint32_t pos) {
        if (pos - win_pos >= (uint32_t)kPngWin) load(pos);
        return win[pos - win_pos];
    }
};

struct GlobalSink {        // decompressed bytes straight into HBM; back-references read them from there
    uint8_t *p;
    uint32_t n;
    __device__ __forceinline__ void put(uint8_t b) { p[n++] = b; }
    __device__ __forceinline__ uint32_t get(uint32_t dist) const { return p[n - dist]; }
    __device__ __forceinline__ uint32_t size() const { return n; }
};

__global__ __launch_bounds__(64) void ke_png_inflate(const KePngDev *__restrict__ imgs, int64_t n, const uint8_t *__restrict__ streams,
                                                     uint8_t *__restrict__ raw, uint8_t *__restrict__ work, int32_t *__restrict__ status) {
    __shared__ KeInflateTables s_tab[64];
    __shared__ __attribute__((aligned(16))) uint8_t s_win[64 * kPngWinPitch];
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const KePngDev &d = imgs[i];
    LdsStream src;
    src.z = streams + d.info.zoff;
    src.win = s_win + threadIdx.x * kPngWinPitch;
    src.load(0);
    KeBitsLsb<LdsStream> bits{&src, 0, d.info.zlen, 0, 0, 0};
    GlobalSink sink{raw + d.raw_off, 0};
    const uint32_t want = (uint32_t)(d.info.width * d.info.channels + 1) * (uint32_t)d.info.height;
    int rc = ke_inflate_zlib(bits, sink, want, s_tab[threadIdx.x], work + (size_t)i * 352);
    if (rc == KE_PNG_OK && sink.n != want) rc = KE_PNG_CORRUPT;
    status[i] = rc;
}

__global__ __launch_bounds__(64) void ke_png_unfilter(const KePngDev *__restrict__ imgs, int64_t n, const uint8_t *__restrict__ raw,
                                                      uint8_t *__restrict__ out, int32_t *__restrict__ status) {
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n || status[i] != KE_PNG_OK) return;
    const KePngDev &d = imgs[i];
    const int rb = d.info.width * d.info.channels, bpp = d.info.channels;
    const uint8_t *src = raw + d.raw_off;
    uint8_t *dst = out + d.out_off;
    for (int y = 0; y < d.info.height; ++y) {
        const uint8_t *row = src + (size_t)y * (rb + 1);
        if (ke_png_unfilter_row(row[0], row + 1, y ? dst + (size_t)(y - 1) * rb : nullptr, dst + (size_t)y * rb, rb, bpp) != KE_PNG_OK) {
            status[i] = KE_PNG_CORRUPT;
            return;
        }
    }
}

}  // namespace

KE_API int ke_png_probe(const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n, int32_t *widths,
                        int32_t *heights, int32_t *channels, int32_t *status_out) {
    if (n < 0 || (n > 0 && (!files || !offsets || !sizes || !widths || !heights || !channels || !status_out))) return KE_EINVAL;
    for (int64_t i = 0; i < n; ++i) {
        KePngInfo info;
        ke_parse_png(files + offsets[i], (size_t)sizes[i], nullptr, info);
        widths[i] = info.width; heights[i] = info.height; channels[i] = info.channels;
        status_out[i] = info.status;
    }
    return KE_OK;
}

KE_API int ke_png_decode(ke_ctx *ctx, const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n,
                         uint8_t *pixels_out, const uint64_t *out_offsets, int32_t *status_out) {
    if (!ctx) return KE_EINVAL;
    if (n < 0 || (n > 0 && (!files || !offsets || !sizes || !pixels_out || !out_offsets || !status_out)))
        return ke_fail(ctx, KE_EINVAL, "NULL argument");
    if (n == 0) return KE_OK;
    if (ke_is_device_ptr(files)) return ke_fail(ctx, KE_EINVAL, "compressed files are parsed on the host: pass host memory");
    if (!ke_is_device_ptr(pixels_out)) return ke_fail(ctx, KE_EINVAL, "pixels_out must be device memory");
    for (const void *p : {(const void *)offsets, (const void *)sizes, (const void *)out_offsets, (const void *)status_out})
        if (ke_is_device_ptr(p)) return ke_fail(ctx, KE_EINVAL, "offsets/sizes/status are host arrays");
    KE_HIP(ctx, hipSetDevice(ctx->device));
    // ---- host: containers, CRCs, one zlib stream per image
    std::vector<uint8_t> streams;
    std::vector<KePngDev> devs;
    std::vector<int64_t> which;
    uint64_t raw_bytes = 0;
    for (int64_t i = 0; i < n; ++i) {
        KePngDev d;
        ke_parse_png(files + offsets[i], (size_t)sizes[i], &streams, d.info);
        status_out[i] = d.info.status;
        if (d.info.status != KE_PNG_OK) continue;
        if (streams.size() > 0xF0000000ull) return ke_fail(ctx, KE_EUNSUPPORTED, "more than 4 GB of compressed PNG data in one call");
        d.raw_off = raw_bytes;
        d.out_off = out_offsets[i];
        raw_bytes += (((uint64_t)d.info.width * d.info.channels + 1) * d.info.height + 15) & ~15ull;
        devs.push_back(d);
        which.push_back(i);
    }
    if (devs.empty()) return KE_OK;
    const int64_t m = (int64_t)devs.size();
    streams.resize(streams.size() + 256);                // the stream windows read up to 64 bytes past an image's data
    void *d_streams, *d_imgs, *d_raw, *d_work, *d_status;
    KE_TRY(ke_reserve(ctx, KE_BUF_PIXELS, streams.size(), &d_streams));
    KE_TRY(ke_reserve(ctx, KE_BUF_META, (size_t)m * sizeof(KePngDev), &d_imgs));
    KE_TRY(ke_reserve(ctx, KE_BUF_TMP, (size_t)raw_bytes + 64, &d_raw));
    KE_TRY(ke_reserve(ctx, KE_BUF_SSIM_IN, (size_t)m * 352, &d_work));
    KE_TRY(ke_reserve(ctx, KE_BUF_OUT0, (size_t)m * 4, &d_status));
    KE_HIP(ctx, hipMemcpyAsync(d_streams, streams.data(), streams.size(), hipMemcpyHostToDevice, ctx->stream));
    KE_HIP(ctx, hipMemcpyAsync(d_imgs, devs.data(), (size_t)m * sizeof(KePngDev), hipMemcpyHostToDevice, ctx->stream));
    ke_time_begin(ctx, KE_T_JPEG);
    hipLaunchKernelGGL(ke_png_inflate, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, ctx->stream, (const KePngDev *)d_imgs, m,
                       (const uint8_t *)d_streams, (uint8_t *)d_raw, (uint8_t *)d_work, (int32_t *)d_status);
    hipLaunchKernelGGL(ke_png_unfilter, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, ctx->stream, (const KePngDev *)d_imgs, m,
                       (const uint8_t *)d_raw, pixels_out, (int32_t *)d_status);
    KE_HIP(ctx, hipGetLastError());
    ke_time_end(ctx, KE_T_JPEG);
    std::vector<int32_t> st((size_t)m);
    KE_HIP(ctx, hipMemcpyAsync(st.data(), d_status, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream));
    KE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t k = 0; k < m; ++k) status_out[which[(size_t)k]] = st[(size_t)k];
    return KE_OK;
}
