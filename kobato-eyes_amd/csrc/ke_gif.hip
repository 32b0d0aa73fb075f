// ke_gif.hip -- the first frame of GIF files decoded on the GPU: the decode step in front of the hash path for the
// fourth of the formats the reference ranks as keepers (src/dup/scanner.py:16-28).  Replaces `Image.open(path)` + pixel access of
// the reference's batch hasher (src/core/fastsig.py:31-34); what leaves is the luma `convert("L")` makes of the frame, which is
// what the reference's hashes see (src/sig/phash.py:25).  The arithmetic is ke_gif_core.h's (held against Pillow on the CPU).
//
//   ke_gif_codes    ONE THREAD PER IMAGE walks the code stream (it is sequential by construction).  A code below the clear
//                   code is a pixel and goes to its place; any other is a string that lies in the output already -- recorded as
//                   a copy (ke_gif_core.h), not made: a lane that waited for its own earlier stores would stall the other 63.
//                   The dictionary (4 096 x {where, how long}) is a slice of HBM per image; the file's bytes come through a
//                   16-byte register window that is refilled one step ahead.
//   ke_gif_copies   ONE WAVE PER IMAGE makes the recorded copies, 64 per round (ke_lz_copies.h, the PNG path's).
//   ke_gif_rows     index -> luma through the frame's table, rows put where an interlaced frame wants them.
#include <algorithm>
#include <vector>

#include "ke_internal.h"

#include "ke_gif_core.h"
#include "ke_lz_copies.h"

namespace {

struct KeGifDev {
    uint64_t file_off;     // the file inside the uploaded bytes
    uint64_t idx_off;      // the frame's indices (width * height bytes) inside the scratch
    uint64_t rec_off;      // this image's copy records (8 bytes each; a copy covers at least 2 pixels)
    uint64_t out_off;      // bytes into the caller's pixel buffer
    uint32_t file_size, data_off;
    int32_t width, height, bits, interlace;
    uint8_t lut[256];
};

constexpr uint32_t kMaxCopy = 511 + 2;            // the record's length field

struct WindowSrc {                                // the file's bytes at ascending positions, 16 at a time, the next 16 on their way
    const uint8_t *file;
    uint32_t base, limit;                         // window = [base, base + 16); nothing is read at or beyond `limit`
    uint64_t lo, hi, nlo, nhi;
    __device__ __forceinline__ void load(uint32_t at, uint64_t &a, uint64_t &b) const {
        a = b = 0;
        if (at < limit) {                         // the uploaded files end with slack: 16 bytes from a position inside are there
            const u32x4 v = ld16(file + at);
            a = (uint64_t)v.x | ((uint64_t)v.y << 32);
            b = (uint64_t)v.z | ((uint64_t)v.w << 32);
        }
    }
    __device__ __forceinline__ void start(uint32_t at) {
        base = at;
        load(at, lo, hi);
        load(at + 16, nlo, nhi);
    }
    __device__ __forceinline__ uint32_t byte(uint32_t pos) {
        if (pos >= base + 16) {                   // positions ascend by one: the next window, and the one after it requested
            base += 16;
            lo = nlo; hi = nhi;
            load(base + 16, nlo, nhi);
        }
        const uint32_t k = pos - base;
        return (uint32_t)((k < 8 ? lo >> (8 * k) : hi >> (8 * (k - 8))) & 255ull);
    }
};

struct HbmDict {
    uint2 *e;
    __device__ __forceinline__ void set(uint32_t code, uint32_t pos, uint32_t len) { e[code] = make_uint2(pos, len); }
    __device__ __forceinline__ void get(uint32_t code, uint32_t &pos, uint32_t &len) const {
        const uint2 v = e[code];
        pos = v.x; len = v.y;
    }
};

struct RecSink {
    uint8_t *idx;
    uint2 *rec;
    uint32_t out, nrec;
    __device__ __forceinline__ void literal(uint8_t b) { idx[out++] = b; }
    __device__ __forceinline__ void copy(uint32_t from, uint32_t len) {
        const uint32_t dist = out - from;
        // A copied string has at least 2 characters; only the frame's last one can be cut to 1 -- it is recorded as 2, the
        // second byte lands in the slack behind the frame's indices.  Pieces of at most 513, none of them a single byte.
        if (len == 1) len = 2;
        while (len) {
            const uint32_t take = len > kMaxCopy ? (len - kMaxCopy == 1 ? kMaxCopy - 1 : kMaxCopy) : len;
            rec[nrec++] = make_uint2(out, (dist << 9) | (take - 2));
            out += take;
            len -= take;
        }
    }
};

__global__ __launch_bounds__(64) void ke_gif_codes(const KeGifDev *__restrict__ imgs, int64_t n, const uint8_t *__restrict__ files,
                                                 uint8_t *__restrict__ indices, uint2 *__restrict__ records, uint2 *__restrict__ dicts,
                                                 int32_t *__restrict__ status, uint32_t *__restrict__ nrec) {
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const KeGifDev &d = imgs[i];
    WindowSrc src;
    src.file = files + d.file_off;
    src.limit = d.file_size;
    src.start(d.data_off);
    HbmDict dict{dicts + (size_t)i * 4096};
    RecSink sink{indices + d.idx_off, records + d.rec_off, 0, 0};
    const uint32_t want = (uint32_t)d.width * (uint32_t)d.height;
    status[i] = ke_gif_lzw(src, d.data_off, d.file_size, d.bits, want, dict, sink);
    nrec[i] = sink.nrec;
}

__global__ __launch_bounds__(64) void ke_gif_copies(const KeGifDev *__restrict__ imgs, uint8_t *__restrict__ indices,
                                                    const uint2 *__restrict__ records, const int32_t *__restrict__ status,
                                                    const uint32_t *__restrict__ nrec) {
    const int64_t i = blockIdx.x;
    if (status[i] != KE_GIF_OK) return;
    const KeGifDev &d = imgs[i];
    ke_lz_make_copies(indices + d.idx_off, records + d.rec_off, nrec[i], 2u);       // a copied string has at least 2 characters
}

constexpr int kRowsPerBlock = 8;

__global__ __launch_bounds__(256) void ke_gif_rows(const KeGifDev *__restrict__ imgs, const uint8_t *__restrict__ indices,
                                                   const int32_t *__restrict__ status, uint8_t *__restrict__ out, int rows) {
    __shared__ uint8_t s_lut[256];
    const int64_t i = blockIdx.x;
    const KeGifDev &d = imgs[i];
    const int k0 = blockIdx.y * rows;
    if (status[i] != KE_GIF_OK || k0 >= d.height) return;
    s_lut[threadIdx.x] = d.lut[threadIdx.x];
    __syncthreads();
    const int W = d.width, k1 = min(k0 + rows, d.height);
    for (int k = k0; k < k1; ++k) {                        // k: the row as it was decoded
        const uint8_t *src = indices + d.idx_off + (size_t)k * W;
        uint8_t *dst = out + d.out_off + (size_t)ke_gif_row(k, d.height, d.interlace) * W;
        for (int x = threadIdx.x; x < W; x += 256) dst[x] = s_lut[src[x]];
    }
}

}  // namespace

KE_API int ke_gif_probe(const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n, int32_t *widths,
                        int32_t *heights, int32_t *channels, int32_t *status_out) {
    if (n < 0 || (n > 0 && (!files || !offsets || !sizes || !widths || !heights || !channels || !status_out))) return KE_EINVAL;
    ke_parallel_ranges(n, [=](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; ++i) {
            KeGifInfo info;
            ke_parse_gif(files + offsets[i], (size_t)sizes[i], info);
            widths[i] = info.width; heights[i] = info.height; channels[i] = info.channels;
            status_out[i] = info.status;
        }
    });
    return KE_OK;
}

KE_API int ke_gif_caveats(const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n, int32_t *flags_out) {
    if (n < 0 || (n > 0 && (!files || !offsets || !sizes || !flags_out))) return KE_EINVAL;
    for (int64_t i = 0; i < n; ++i) flags_out[i] = 0;               // (the decoder yields luma: only the hashing seams take it)
    return KE_OK;
}

KE_API int ke_gif_decode(ke_ctx *ctx, const uint8_t *files, const uint64_t *offsets, const uint64_t *sizes, int64_t n,
                         uint8_t *pixels_out, const uint64_t *out_offsets, int32_t *status_out) {
    if (!ctx) return KE_EINVAL;
    if (n < 0 || (n > 0 && (!files || !offsets || !sizes || !pixels_out || !out_offsets || !status_out)))
        return ke_fail(ctx, KE_EINVAL, "NULL argument");
    if (n == 0) return KE_OK;
    if (ke_is_device_ptr(files)) return ke_fail(ctx, KE_EINVAL, "the files' containers are walked on the host: pass host memory (pinned staging is fine)");
    if (!ke_is_device_ptr(pixels_out)) return ke_fail(ctx, KE_EINVAL, "pixels_out must be device memory");
    for (const void *p : {(const void *)offsets, (const void *)sizes, (const void *)out_offsets, (const void *)status_out})
        if (ke_is_device_ptr(p)) return ke_fail(ctx, KE_EINVAL, "offsets/sizes/status are host arrays");
    KE_HIP(ctx, hipSetDevice(ctx->device));
    struct Item { KeGifDev d; int64_t which; };
    std::vector<Item> items;
    items.reserve((size_t)n);
    uint64_t lo = ~0ull, hi = 0;
    std::vector<KeGifInfo> infos((size_t)n);                     // the containers are walked on the host's threads
    ke_parallel_ranges(n, [&](int64_t a, int64_t b, int) {
        for (int64_t i = a; i < b; ++i) ke_parse_gif(files + offsets[i], (size_t)sizes[i], infos[(size_t)i]);
    });
    for (int64_t i = 0; i < n; ++i) {
        const KeGifInfo &info = infos[(size_t)i];
        status_out[i] = info.status;
        if (info.status != KE_GIF_OK) continue;
        Item it;
        it.which = i;
        it.d.file_off = offsets[i];
        it.d.out_off = out_offsets[i];
        it.d.file_size = (uint32_t)sizes[i];
        it.d.data_off = info.data_off;
        it.d.width = info.width; it.d.height = info.height; it.d.bits = info.bits; it.d.interlace = info.interlace;
        std::memcpy(it.d.lut, info.lut, 256);
        lo = std::min(lo, offsets[i]);
        hi = std::max(hi, offsets[i] + sizes[i]);
        items.push_back(it);
    }
    if (items.empty()) return KE_OK;
    // lanes of one wave finish together at best: neighbours in the batch should have streams of like length
    std::stable_sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.d.file_size > b.d.file_size; });
    void *d_files;
    KE_TRY(ke_reserve(ctx, KE_BUF_PIXELS, (size_t)(hi - lo) + 256, &d_files));
    KE_HIP(ctx, hipMemcpyAsync(d_files, files + lo, (size_t)(hi - lo), hipMemcpyHostToDevice, ctx->stream));
    // sub-batches bounded by scratch: indices (1 B per pixel) + copy records (8 B per 2 pixels at worst) + 32 KB of dictionary
    size_t free_b = 0, total_b = 0;
    KE_HIP(ctx, hipMemGetInfo(&free_b, &total_b));
    const uint64_t held = (uint64_t)ctx->buf[KE_BUF_SSIM_IN].bytes + ctx->buf[KE_BUF_TMP].bytes + ctx->buf[KE_BUF_SSIM_AUX].bytes;
    const uint64_t budget = std::max<uint64_t>((uint64_t)2 << 30, std::min<uint64_t>((held + (uint64_t)free_b) / 2, (uint64_t)160 << 30));
    std::vector<KeGifDev> devs;
    std::vector<int32_t> st;
    size_t first = 0;
    ke_time_begin(ctx, KE_T_JPEG);
    while (first < items.size()) {
        uint64_t idx_bytes = 0, nrecs = 0;
        int max_height = 0;
        size_t last = first;
        devs.clear();
        while (last < items.size()) {
            Item &it = items[last];
            const uint64_t px = (uint64_t)it.d.width * it.d.height;
            const uint64_t ib = (px + 64 + 15) & ~15ull, rc = px / 2 + 2;
            if (last > first && idx_bytes + ib + (nrecs + rc) * 8 + (uint64_t)(last - first + 1) * 32768 > budget) break;
            it.d.file_off -= lo;
            it.d.idx_off = idx_bytes;
            it.d.rec_off = nrecs;
            idx_bytes += ib;
            nrecs += rc;
            max_height = std::max(max_height, it.d.height);
            devs.push_back(it.d);
            ++last;
        }
        const int64_t m = (int64_t)devs.size();
        void *d_imgs, *d_idx, *d_rec, *d_dict, *d_status, *d_nrec;
        KE_TRY(ke_reserve(ctx, KE_BUF_META, (size_t)m * sizeof(KeGifDev), &d_imgs));
        KE_TRY(ke_reserve(ctx, KE_BUF_TMP, (size_t)idx_bytes + 128, &d_idx));
        KE_TRY(ke_reserve(ctx, KE_BUF_SSIM_AUX, (size_t)nrecs * 8, &d_rec));
        KE_TRY(ke_reserve(ctx, KE_BUF_SSIM_IN, (size_t)m * 32768, &d_dict));
        KE_TRY(ke_reserve(ctx, KE_BUF_OUT0, (size_t)m * 4, &d_status));
        KE_TRY(ke_reserve(ctx, KE_BUF_TILE32, (size_t)m * 4, &d_nrec));
        KE_HIP(ctx, hipMemcpyAsync(d_imgs, devs.data(), (size_t)m * sizeof(KeGifDev), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(ke_gif_codes, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, ctx->stream, (const KeGifDev *)d_imgs, m, (const uint8_t *)d_files,
                           (uint8_t *)d_idx, (uint2 *)d_rec, (uint2 *)d_dict, (int32_t *)d_status, (uint32_t *)d_nrec);
        hipLaunchKernelGGL(ke_gif_copies, dim3((unsigned)m), dim3(64), 0, ctx->stream, (const KeGifDev *)d_imgs, (uint8_t *)d_idx,
                           (const uint2 *)d_rec, (const int32_t *)d_status, (const uint32_t *)d_nrec);
        const int rows = std::max(kRowsPerBlock, (max_height + 65534) / 65535);
        hipLaunchKernelGGL(ke_gif_rows, dim3((unsigned)m, (unsigned)((max_height + rows - 1) / rows)), dim3(256), 0, ctx->stream,
                           (const KeGifDev *)d_imgs, (const uint8_t *)d_idx, (const int32_t *)d_status, pixels_out, rows);
        KE_HIP(ctx, hipGetLastError());
        st.resize((size_t)m);
        KE_HIP(ctx, hipMemcpyAsync(st.data(), d_status, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream));
        KE_HIP(ctx, hipStreamSynchronize(ctx->stream));                  // devs / st are host vectors; the scratch is reused
        for (int64_t k = 0; k < m; ++k) status_out[items[first + (size_t)k].which] = st[(size_t)k];
        first = last;
    }
    ke_time_end(ctx, KE_T_JPEG);
    return KE_OK;
}
