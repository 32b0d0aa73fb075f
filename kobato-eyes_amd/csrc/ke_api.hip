// ke_api.hip -- C ABI of libkeyes_hip.so: context, memory staging, argument checking and
// dispatch into the kernel launchers.  See include/keyes.h for the contract and the
// reference lines each entry point replaces.
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cmath>
#include <numeric>
#include <unordered_map>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "ke_internal.h"

static thread_local std::string g_create_err;

int ke_fail(ke_ctx *ctx, int code, const char *fmt, ...) {
    char msg[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(msg, sizeof msg, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = msg;
    else
        g_create_err = msg;
    (void)hipGetLastError();  // clear the sticky error so later calls can proceed
    return code;
}

bool ke_is_device_ptr(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    std::memset(&a, 0, sizeof a);
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;  // plain host memory
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

int ke_reserve(ke_ctx *ctx, int which, size_t bytes, void **out) {
    KeDevBuf &b = ctx->buf[which];
    if (bytes == 0) bytes = 16;
    if (b.bytes < bytes) {
        if (b.ptr) {
            KE_HIP(ctx, hipStreamSynchronize(ctx->stream));
            KE_HIP(ctx, hipFree(b.ptr));
            b.ptr = nullptr;
            b.bytes = 0;
        }
        size_t want = (bytes + 255) & ~(size_t)255;
        KE_HIP(ctx, hipMalloc(&b.ptr, want));
        b.bytes = want;
    }
    *out = b.ptr;
    return KE_OK;
}

int ke_to_device(ke_ctx *ctx, const void *p, size_t bytes, int which, const void **dev) {
    if (!p) {
        *dev = nullptr;
        return KE_OK;
    }
    if (ke_is_device_ptr(p)) {
        *dev = p;
        return KE_OK;
    }
    void *d = nullptr;
    KE_TRY(ke_reserve(ctx, which, bytes, &d));
    if (bytes) KE_HIP(ctx, hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, ctx->stream));
    *dev = d;
    return KE_OK;
}

static int upload_i32(ke_ctx *ctx, const std::vector<int32_t> &v, int32_t **out) {
    KE_HIP(ctx, hipMalloc((void **)out, std::max<size_t>(v.size(), 4) * sizeof(int32_t)));
    if (!v.empty())
        KE_HIP(ctx, hipMemcpy(*out, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    return KE_OK;
}

// Tap tables are cached per (axis length, outputs, filter, source interval).  A library of many image shapes would
// grow the cache without bound, so the public entry points drop it once it holds more than kMaxCoeffTables tables
// (after draining the stream: kernels in flight may still read them).
constexpr size_t kMaxCoeffTables = 2048;

static void free_coeff_cache(ke_ctx *ctx) {
    for (auto &kv : ctx->coeffs) {
        KeAxisCoeffs *c = kv.second;
        for (int32_t *p : {c->d_bounds, c->d_kk, c->d_start, c->d_bias, c->d_packed})
            if (p) (void)hipFree(p);
        for (auto &ck : c->chunked) {
            if (ck.second->d_cstart) (void)hipFree(ck.second->d_cstart);
            if (ck.second->d_cpacked) (void)hipFree(ck.second->d_cpacked);
            if (ck.second->d_cxor) (void)hipFree(ck.second->d_cxor);
            delete ck.second;
        }
        for (auto &mk : c->mx) {
            if (mk.second->d_frag) (void)hipFree(mk.second->d_frag);
            delete mk.second;
        }
        delete c;
    }
    ctx->coeffs.clear();
}

static void trim_coeff_cache(ke_ctx *ctx) {
    if (ctx->coeffs.size() <= kMaxCoeffTables) return;
    (void)hipStreamSynchronize(ctx->stream);
    free_coeff_cache(ctx);
}

const KeAxisCoeffs *ke_get_coeffs(ke_ctx *ctx, int in_size, int out_size, int filter, float in0, float in1) {
    if (in1 < 0.0f) { in0 = 0.0f; in1 = (float)in_size; }
    uint32_t b0, b1;
    std::memcpy(&b0, &in0, 4);
    std::memcpy(&b1, &in1, 4);
    auto key = std::make_tuple(in_size, out_size * 4 + filter, b0, b1);
    auto it = ctx->coeffs.find(key);
    if (it != ctx->coeffs.end()) return it->second;
    auto *c = new KeAxisCoeffs();
    ke_build_axis_coeffs(in_size, out_size, *c, filter, in0, in1);
    if (upload_i32(ctx, c->bounds, &c->d_bounds) || upload_i32(ctx, c->kk, &c->d_kk) ||
        upload_i32(ctx, c->start, &c->d_start) || upload_i32(ctx, c->bias, &c->d_bias) ||
        upload_i32(ctx, c->packed, &c->d_packed)) {
        delete c;
        return nullptr;
    }
    ctx->coeffs[key] = c;
    return c;
}

const KeChunkTable *ke_get_chunks(ke_ctx *ctx, const KeAxisCoeffs *cc, int cpo, int ndwc_multiple) {
    auto *c = const_cast<KeAxisCoeffs *>(cc);
    const int key = cpo * 64 + ndwc_multiple;
    auto it = c->chunked.find(key);
    if (it != c->chunked.end()) return it->second;
    auto *t = new KeChunkTable();
    ke_build_chunked(*c, cpo, *t, ndwc_multiple);
    if (upload_i32(ctx, t->cstart, &t->d_cstart) || upload_i32(ctx, t->cpacked, &t->d_cpacked) ||
        upload_i32(ctx, t->cxor, &t->d_cxor)) {
        delete t;
        return nullptr;
    }
    c->chunked[key] = t;
    return t;
}

const KeMxTable *ke_get_mx(ke_ctx *ctx, const KeAxisCoeffs *cc, int min_ks, bool align64) {
    auto *c = const_cast<KeAxisCoeffs *>(cc);
    const int key = min_ks * 2 + (align64 ? 1 : 0);
    auto it = c->mx.find(key);
    if (it != c->mx.end()) return it->second;
    auto *t = new KeMxTable();
    ke_build_mx(*c, *t, min_ks, align64);
    if (upload_i32(ctx, t->frag, &t->d_frag)) {
        delete t;
        return nullptr;
    }
    c->mx[key] = t;
    return t;
}

void ke_time_begin(ke_ctx *ctx, int kind) {
    ctx->ev_valid[kind] = false;
    (void)hipEventRecord(ctx->ev0[kind], ctx->stream);
}

void ke_time_end(ke_ctx *ctx, int kind) {
    if (hipEventRecord(ctx->ev1[kind], ctx->stream) == hipSuccess) ctx->ev_valid[kind] = true;
}

// ---- context -------------------------------------------------------------------------------
KE_API int ke_abi_version(void) { return KE_ABI_VERSION; }

KE_API const char *ke_create_error(void) { return g_create_err.c_str(); }

KE_API ke_ctx *ke_create(int device_id) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        ke_fail(nullptr, KE_ENODEV, "no HIP device visible (libkeyes_hip has no CPU fallback)");
        return nullptr;
    }
    if (device_id < 0 || device_id >= count) {
        ke_fail(nullptr, KE_EINVAL, "device_id %d out of range [0,%d)", device_id, count);
        return nullptr;
    }
    if (hipSetDevice(device_id) != hipSuccess) {
        ke_fail(nullptr, KE_EHIP, "hipSetDevice(%d) failed", device_id);
        return nullptr;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) {
        ke_fail(nullptr, KE_EHIP, "hipGetDeviceProperties failed");
        return nullptr;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        ke_fail(nullptr, KE_ENODEV, "device %d is %s; this library is built for gfx950 only", device_id,
                prop.gcnArchName);
        return nullptr;
    }
    auto *ctx = new ke_ctx();
    ctx->device = device_id;
    ctx->cu_count = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        ke_fail(nullptr, KE_EHIP, "hipStreamCreate failed");
        delete ctx;
        return nullptr;
    }
    ctx->stream = ctx->own_stream;
    for (int k = 0; k < KE_T_COUNT; ++k) {
        (void)hipEventCreate(&ctx->ev0[k]);
        (void)hipEventCreate(&ctx->ev1[k]);
    }
    return ctx;
}

static void stage_free(ke_ctx *ctx);

KE_API void ke_destroy(ke_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    stage_free(ctx);
    for (int k = 0; k < 2; ++k) {
        if (ctx->side[k]) (void)hipStreamDestroy(ctx->side[k]);
        if (ctx->side_join[k]) (void)hipEventDestroy(ctx->side_join[k]);
    }
    if (ctx->side_fork) (void)hipEventDestroy(ctx->side_fork);
    if (ctx->h_comm) (void)hipHostFree(ctx->h_comm);
    if (ctx->h_meta) (void)hipHostFree(ctx->h_meta);
    for (auto &b : ctx->buf)
        if (b.ptr) (void)hipFree(b.ptr);
    free_coeff_cache(ctx);
    for (int k = 0; k < KE_T_COUNT; ++k) {
        (void)hipEventDestroy(ctx->ev0[k]);
        (void)hipEventDestroy(ctx->ev1[k]);
    }
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

KE_API const char *ke_last_error(ke_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

KE_API int ke_set_stream(ke_ctx *ctx, void *hip_stream) {
    if (!ctx) return KE_EINVAL;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return KE_OK;
}

KE_API void *ke_get_stream(ke_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

KE_API int ke_synchronize(ke_ctx *ctx) {
    if (!ctx) return KE_EINVAL;
    KE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KE_OK;
}

KE_API int ke_device_info(ke_ctx *ctx, char *name, size_t name_len, int32_t *compute_units, int64_t *total_mem) {
    if (!ctx) return KE_EINVAL;
    hipDeviceProp_t prop;
    KE_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_len) std::snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (total_mem) *total_mem = (int64_t)prop.totalGlobalMem;
    return KE_OK;
}

KE_API int ke_malloc(ke_ctx *ctx, size_t bytes, void **out) {
    if (!ctx || !out) return KE_EINVAL;
    KE_HIP(ctx, hipMalloc(out, bytes ? bytes : 16));
    return KE_OK;
}

KE_API int ke_free(ke_ctx *ctx, void *p) {
    if (!ctx) return KE_EINVAL;
    KE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    KE_HIP(ctx, hipFree(p));
    return KE_OK;
}

KE_API int ke_host_alloc(ke_ctx *ctx, size_t bytes, void **out) {
    if (!ctx || !out) return KE_EINVAL;
    KE_HIP(ctx, hipSetDevice(ctx->device));
    KE_HIP(ctx, hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault));
    return KE_OK;
}

KE_API int ke_host_pack(uint8_t *dst, const uint8_t *const *srcs, const uint64_t *offsets, const uint64_t *sizes, int64_t n) {
    if (n < 0 || (n > 0 && (!dst || !srcs || !offsets || !sizes))) return KE_EINVAL;
    ke_parallel_ranges(n, [=](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; ++i) std::memcpy(dst + offsets[i], srcs[i], (size_t)sizes[i]);
    });
    return KE_OK;
}

KE_API int ke_host_read_files(const char *const *paths, int64_t n, uint8_t *dst, uint64_t capacity, uint64_t *offsets, uint64_t *sizes,
                              uint64_t *needed_out) {
    if (n < 0 || (n > 0 && (!paths || !offsets || !sizes)) || !needed_out) return KE_EINVAL;
    // sizes first (a file that is not there or is not a regular file counts 0 bytes), then the layout, then the bytes; no
    // descriptor stays open between the two passes (a batch is tens of thousands of files, the descriptor limit is not)
    ke_parallel_ranges(n, [&](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; ++i) {
            struct stat st;
            sizes[i] = (paths[i] && ::stat(paths[i], &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) ? (uint64_t)st.st_size : 0;
        }
    });
    uint64_t at = 0;
    for (int64_t i = 0; i < n; ++i) { offsets[i] = at; at += sizes[i]; }
    *needed_out = at + 64;
    const bool fits = dst && at + 64 <= capacity;
    if (fits)
        ke_parallel_ranges(n, [&](int64_t lo, int64_t hi, int) {
            for (int64_t i = lo; i < hi; ++i) {
                if (sizes[i] == 0) continue;
                uint64_t got = 0;
                const int fd = ::open(paths[i], O_RDONLY | O_CLOEXEC);
                if (fd >= 0) {
                    while (got < sizes[i]) {
                        const ssize_t r = ::read(fd, dst + offsets[i] + got, (size_t)(sizes[i] - got));
                        if (r <= 0) break;
                        got += (uint64_t)r;
                    }
                    ::close(fd);
                }
                if (got != sizes[i]) std::memset(dst + offsets[i], 0, (size_t)sizes[i]);      // unreadable or shrunk meanwhile: not a decodable file
            }
        });
    if (!fits) return KE_ENOMEM;
    std::memset(dst + at, 0, 64);
    return KE_OK;
}

KE_API int ke_host_free(ke_ctx *ctx, void *p) {
    if (!ctx) return KE_EINVAL;
    KE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    KE_HIP(ctx, hipHostFree(p));
    return KE_OK;
}

KE_API int ke_memcpy(ke_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx || (!dst && bytes) || (!src && bytes)) return KE_EINVAL;
    if (!bytes) return KE_OK;
    KE_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, ctx->stream));
    KE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KE_OK;
}

KE_API double ke_last_kernel_ms(ke_ctx *ctx, int32_t kind) {
    if (!ctx || kind < 0 || kind >= KE_T_COUNT || !ctx->ev_valid[kind]) return -1.0;
    if (hipEventSynchronize(ctx->ev1[kind]) != hipSuccess) return -1.0;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->ev0[kind], ctx->ev1[kind]) != hipSuccess) return -1.0;
    return (double)ms;
}

// ---- hashing -------------------------------------------------------------------------------
namespace {

struct HashOutputs {
    uint64_t *d_phash = nullptr, *d_dhash = nullptr;
    uint8_t *d_t32 = nullptr, *d_t98 = nullptr;
    bool stage_p = false, stage_d = false, stage_t32 = false, stage_t98 = false;
};

// Chunk size for host-resident pixel input: bounded staging buffer.
constexpr size_t kStageBytes = (size_t)1 << 30;

struct MarginScope {   // the kernels of one call find the margin array in the context; cleared on every exit path
    ke_ctx *ctx;
    ~MarginScope() { ctx->margin_cur = nullptr; }
};

int hash_uniform_impl(ke_ctx *ctx, const uint8_t *pixels, int64_t n, int w, int h, int ch, uint64_t *phash_out,
                      uint64_t *dhash_out, uint8_t *t32_out, uint8_t *t98_out, float *margin_out = nullptr) {
    if (!ctx) return KE_EINVAL;
    if (n < 0 || (n > 0 && !pixels)) return ke_fail(ctx, KE_EINVAL, "pixels is NULL");
    if (w <= 0 || h <= 0) return ke_fail(ctx, KE_EINVAL, "width/height must be positive (got %dx%d)", w, h);
    if (ch != 1 && ch != 3 && ch != 4) return ke_fail(ctx, KE_EINVAL, "channels must be 1, 3 or 4 (got %d)", ch);
    if (margin_out && !phash_out) return ke_fail(ctx, KE_EINVAL, "margin_out needs phash_out");
    if (n == 0) return KE_OK;
    KE_HIP(ctx, hipSetDevice(ctx->device));
    trim_coeff_cache(ctx);
    const size_t img_bytes = (size_t)w * h * ch;
    const bool in_dev = ke_is_device_ptr(pixels);
    const bool stage_m = margin_out && !ke_is_device_ptr(margin_out);
    MarginScope margin_scope{ctx};
    HashOutputs o;
    o.stage_p = phash_out && !ke_is_device_ptr(phash_out);
    o.stage_d = dhash_out && !ke_is_device_ptr(dhash_out);
    o.stage_t32 = t32_out && !ke_is_device_ptr(t32_out);
    o.stage_t98 = t98_out && !ke_is_device_ptr(t98_out);
    const int64_t chunk = in_dev ? n : std::max<int64_t>(1, (int64_t)(kStageBytes / img_bytes));
    ke_time_begin(ctx, KE_T_HASH);
    for (int64_t first = 0; first < n; first += chunk) {
        const int64_t m = std::min(chunk, n - first);
        const void *d_px = nullptr;
        KE_TRY(ke_to_device(ctx, pixels + (size_t)first * img_bytes, (size_t)m * img_bytes, KE_BUF_PIXELS, &d_px));
        void *tmp;
        o.d_phash = phash_out ? phash_out + first : nullptr;
        o.d_dhash = dhash_out ? dhash_out + first : nullptr;
        o.d_t32 = t32_out ? t32_out + (size_t)first * 1024 : nullptr;
        o.d_t98 = t98_out ? t98_out + (size_t)first * 72 : nullptr;
        if (o.stage_p) { KE_TRY(ke_reserve(ctx, KE_BUF_OUT0, (size_t)m * 8, &tmp)); o.d_phash = (uint64_t *)tmp; }
        if (o.stage_d) { KE_TRY(ke_reserve(ctx, KE_BUF_OUT1, (size_t)m * 8, &tmp)); o.d_dhash = (uint64_t *)tmp; }
        ctx->margin_cur = margin_out ? margin_out + first : nullptr;
        if (stage_m) { KE_TRY(ke_reserve(ctx, KE_BUF_OUT2, (size_t)m * 4, &tmp)); ctx->margin_cur = (float *)tmp; }
        // tiles requested by a host caller are produced straight into the scratch tile buffers
        KeHashGroup g{(const uint8_t *)d_px, nullptr, img_bytes, nullptr, m, w, h, ch};
        uint8_t *d_t32 = o.stage_t32 ? nullptr : o.d_t32, *d_t98 = o.stage_t98 ? nullptr : o.d_t98;
        if (o.stage_t32) { KE_TRY(ke_reserve(ctx, KE_BUF_SSIM_IN, (size_t)m * 1024, &tmp)); d_t32 = (uint8_t *)tmp; }
        if (o.stage_t98) { KE_TRY(ke_reserve(ctx, KE_BUF_SSIM_AUX, (size_t)m * 72, &tmp)); d_t98 = (uint8_t *)tmp; }
        KE_TRY(ke_launch_hash_group(ctx, g, o.d_phash, o.d_dhash, d_t32, d_t98));
        if (o.stage_p)
            KE_HIP(ctx, hipMemcpyAsync(phash_out + first, o.d_phash, (size_t)m * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (o.stage_d)
            KE_HIP(ctx, hipMemcpyAsync(dhash_out + first, o.d_dhash, (size_t)m * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (stage_m)
            KE_HIP(ctx, hipMemcpyAsync(margin_out + first, ctx->margin_cur, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (o.stage_t32)
            KE_HIP(ctx, hipMemcpyAsync(t32_out + (size_t)first * 1024, d_t32, (size_t)m * 1024, hipMemcpyDeviceToHost, ctx->stream));
        if (o.stage_t98)
            KE_HIP(ctx, hipMemcpyAsync(t98_out + (size_t)first * 72, d_t98, (size_t)m * 72, hipMemcpyDeviceToHost, ctx->stream));
        if (!in_dev || o.stage_p || o.stage_d || o.stage_t32 || o.stage_t98 || stage_m)
            KE_HIP(ctx, hipStreamSynchronize(ctx->stream));  // staging buffers are reused by the next chunk
    }
    ke_time_end(ctx, KE_T_HASH);
    return KE_OK;
}

}  // namespace

KE_API int ke_hash_uniform(ke_ctx *ctx, const uint8_t *pixels, int64_t n, int32_t width, int32_t height,
                           int32_t channels, uint64_t *phash_out, uint64_t *dhash_out) {
    return hash_uniform_impl(ctx, pixels, n, width, height, channels, phash_out, dhash_out, nullptr, nullptr);
}

KE_API int ke_hash_uniform_ex(ke_ctx *ctx, const uint8_t *pixels, int64_t n, int32_t width, int32_t height,
                              int32_t channels, uint64_t *phash_out, uint64_t *dhash_out, float *margin_out) {
    return hash_uniform_impl(ctx, pixels, n, width, height, channels, phash_out, dhash_out, nullptr, nullptr, margin_out);
}

KE_API int ke_luma_tiles_uniform(ke_ctx *ctx, const uint8_t *pixels, int64_t n, int32_t width, int32_t height,
                                 int32_t channels, uint8_t *tile32_out, uint8_t *tile98_out) {
    return hash_uniform_impl(ctx, pixels, n, width, height, channels, nullptr, nullptr, tile32_out, tile98_out);
}

static int hash_images_impl(ke_ctx *ctx, const uint8_t *pixels, const uint64_t *offsets, const int32_t *widths,
                            const int32_t *heights, int32_t channels, int64_t n, uint64_t *phash_out,
                            uint64_t *dhash_out, int32_t *status_out, float *margin_out) {
    if (!ctx) return KE_EINVAL;
    if (margin_out && !phash_out) return ke_fail(ctx, KE_EINVAL, "margin_out needs phash_out");
    if (n < 0 || (n > 0 && (!pixels || !widths || !heights)))
        return ke_fail(ctx, KE_EINVAL, "pixels/widths/heights must be non-NULL");
    if (channels != 1 && channels != 3 && channels != 4)
        return ke_fail(ctx, KE_EINVAL, "channels must be 1, 3 or 4 (got %d)", channels);
    if (ke_is_device_ptr(widths) || ke_is_device_ptr(heights) || ke_is_device_ptr(offsets) ||
        ke_is_device_ptr(status_out))
        return ke_fail(ctx, KE_EINVAL, "offsets/widths/heights/status are metadata and must be host arrays");
    if (n == 0) return KE_OK;
    KE_HIP(ctx, hipSetDevice(ctx->device));
    trim_coeff_cache(ctx);
    // byte offset of every image, then group equal shapes (one launch group per distinct size)
    std::vector<uint64_t> off((size_t)n);
    uint64_t run = 0;
    for (int64_t i = 0; i < n; ++i) {
        off[i] = offsets ? offsets[i] : run;
        if (widths[i] > 0 && heights[i] > 0) run += (uint64_t)widths[i] * heights[i] * channels;
    }
    // size groups: an image's group id comes from a one-entry cache or a hash lookup (the shape list of a library is short),
    // the members are then laid out by a counting sort -- linear in n with small constants (a million-image call spends its
    // host time here)
    std::map<std::pair<int, int>, std::vector<int64_t>> groups;
    {
        std::unordered_map<uint64_t, int32_t> ids;
        std::vector<int32_t> gid((size_t)n, -1);
        std::vector<int64_t> counts;
        std::vector<std::pair<int, int>> shapes;
        uint64_t last_key = ~0ull;
        int32_t last_id = -1;
        for (int64_t i = 0; i < n; ++i) {
            const bool ok = widths[i] > 0 && heights[i] > 0;
            if (status_out) status_out[i] = ok ? KE_IMG_OK : KE_IMG_BAD_SHAPE;
            if (!ok) continue;
            const uint64_t key = ((uint64_t)(uint32_t)widths[i] << 32) | (uint32_t)heights[i];
            if (key != last_key) {
                auto it = ids.find(key);
                if (it == ids.end()) {
                    it = ids.emplace(key, (int32_t)shapes.size()).first;
                    shapes.emplace_back(widths[i], heights[i]);
                    counts.push_back(0);
                }
                last_key = key;
                last_id = it->second;
            }
            gid[i] = last_id;
            ++counts[last_id];
        }
        std::vector<std::vector<int64_t> *> slots(shapes.size());
        for (size_t k = 0; k < shapes.size(); ++k) {
            auto &v = groups[shapes[k]];
            v.reserve((size_t)counts[k]);
            slots[k] = &v;
        }
        for (int64_t i = 0; i < n; ++i)
            if (gid[i] >= 0) slots[gid[i]]->push_back(i);
    }
    const bool in_dev = ke_is_device_ptr(pixels);
    const bool p_dev = phash_out && ke_is_device_ptr(phash_out), d_dev = dhash_out && ke_is_device_ptr(dhash_out);
    // outputs are produced in a device array of n slots, then copied out once
    void *tmp;
    uint64_t *d_ph = nullptr, *d_dh = nullptr;
    if (phash_out) {
        if (p_dev) d_ph = phash_out;
        else { KE_TRY(ke_reserve(ctx, KE_BUF_OUT0, (size_t)n * 8, &tmp)); d_ph = (uint64_t *)tmp; }
        KE_HIP(ctx, hipMemsetAsync(d_ph, 0, (size_t)n * 8, ctx->stream));
    }
    if (dhash_out) {
        if (d_dev) d_dh = dhash_out;
        else { KE_TRY(ke_reserve(ctx, KE_BUF_OUT1, (size_t)n * 8, &tmp)); d_dh = (uint64_t *)tmp; }
        KE_HIP(ctx, hipMemsetAsync(d_dh, 0, (size_t)n * 8, ctx->stream));
    }
    const bool m_dev = margin_out && ke_is_device_ptr(margin_out);
    MarginScope margin_scope{ctx};
    if (margin_out) {
        if (m_dev) ctx->margin_cur = margin_out;
        else { KE_TRY(ke_reserve(ctx, KE_BUF_OUT2, (size_t)n * 4, &tmp)); ctx->margin_cur = (float *)tmp; }
        KE_HIP(ctx, hipMemsetAsync(ctx->margin_cur, 0, (size_t)n * 4, ctx->stream));
    }
    ke_time_begin(ctx, KE_T_HASH);
    if (in_dev) {
        // device-resident batch: one metadata upload for all shape groups ([offsets | output slots] per group),
        // then the groups are launched back to back with no host synchronisation in between
        std::vector<uint64_t> meta_h((size_t)2 * n);
        size_t cursor = 0;
        std::vector<std::pair<size_t, size_t>> spans;   // (meta offset, count) per group, in map order
        for (auto &kv : groups) {
            const std::vector<int64_t> &idx = kv.second;
            spans.emplace_back(cursor, idx.size());
            for (size_t k = 0; k < idx.size(); ++k) {
                meta_h[cursor + k] = off[idx[k]];
                meta_h[cursor + idx.size() + k] = (uint64_t)idx[k];
            }
            cursor += 2 * idx.size();
        }
        void *meta;
        KE_TRY(ke_reserve(ctx, KE_BUF_META, std::max<size_t>(cursor, 2) * 8, &meta));
        if (cursor) KE_HIP(ctx, hipMemcpyAsync(meta, meta_h.data(), cursor * 8, hipMemcpyHostToDevice, ctx->stream));
        KE_HIP(ctx, hipStreamSynchronize(ctx->stream));   // meta_h is a local
        // Neighbouring size groups overlap: the one-workgroup-per-image kernels of the groups alternate between two side
        // streams (they use no scratch), so the tail of one launch -- its last workgroups draining -- runs beside the head of
        // the next instead of leaving the chip half empty 81 times in a mixed-resolution batch.  Whatever a group needs
        // beyond that kernel (banded paths, second passes) stays on the context's stream.
        const bool overlap = groups.size() > 1 && !getenv("KE_NO_GROUP_OVERLAP");
        if (overlap && !ctx->side[0]) {
            for (int k = 0; k < 2; ++k) {
                KE_HIP(ctx, hipStreamCreateWithFlags(&ctx->side[k], hipStreamNonBlocking));
                KE_HIP(ctx, hipEventCreateWithFlags(&ctx->side_join[k], hipEventDisableTiming));
            }
            KE_HIP(ctx, hipEventCreateWithFlags(&ctx->side_fork, hipEventDisableTiming));
        }
        if (overlap) {
            KE_HIP(ctx, hipEventRecord(ctx->side_fork, ctx->stream));          // outputs zeroed, metadata in place
            for (int k = 0; k < 2; ++k) KE_HIP(ctx, hipStreamWaitEvent(ctx->side[k], ctx->side_fork, 0));
        }
        size_t gi = 0;
        int rc_groups = KE_OK;
        for (auto &kv : groups) {
            const int w = kv.first.first, h = kv.first.second;
            const uint64_t *d_off = (const uint64_t *)meta + spans[gi].first;
            const int64_t m = (int64_t)spans[gi].second;
            KeHashGroup g{pixels, d_off, (uint64_t)w * h * channels, (const int64_t *)(d_off + m), m, w, h, channels};
            // the dword loaders of the single-pass and aligned banded kernels want every image of the group on a dword
            // boundary; a packed stream loses that after the first image whose byte size is not a multiple of 4
            for (int64_t idx : kv.second)
                if (((uintptr_t)pixels + off[idx]) % 4 != 0) { g.misaligned = true; break; }
            rc_groups = ke_launch_hash_group(ctx, g, d_ph, d_dh, nullptr, nullptr, overlap ? ctx->side[gi & 1] : nullptr);
            if (rc_groups != KE_OK) break;
            ++gi;
        }
        if (overlap) {                                                          // join, also on the error path
            for (int k = 0; k < 2; ++k) {
                KE_HIP(ctx, hipEventRecord(ctx->side_join[k], ctx->side[k]));
                KE_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_join[k], 0));
            }
        }
        KE_TRY(rc_groups);
    } else {
        for (auto &kv : groups) {
            const int w = kv.first.first, h = kv.first.second;
            const size_t img_bytes = (size_t)w * h * channels;
            const std::vector<int64_t> &idx = kv.second;
            const int64_t chunk = std::max<int64_t>(1, (int64_t)(kStageBytes / img_bytes));
            for (size_t first = 0; first < idx.size(); first += (size_t)chunk) {
                const int64_t m = (int64_t)std::min<size_t>((size_t)chunk, idx.size() - first);
                std::vector<uint64_t> goff((size_t)m);
                std::vector<int64_t> gidx(idx.begin() + first, idx.begin() + first + m);
                // pack this chunk of host images into the staging buffer
                KE_TRY(ke_reserve(ctx, KE_BUF_PIXELS, (size_t)m * img_bytes, &tmp));
                for (int64_t k = 0; k < m; ++k) {
                    goff[k] = (uint64_t)k * img_bytes;
                    KE_HIP(ctx, hipMemcpyAsync((uint8_t *)tmp + goff[k], pixels + off[gidx[k]], img_bytes,
                                               hipMemcpyHostToDevice, ctx->stream));
                }
                void *meta;
                KE_TRY(ke_reserve(ctx, KE_BUF_META, (size_t)m * 16, &meta));
                KE_HIP(ctx, hipMemcpyAsync(meta, goff.data(), (size_t)m * 8, hipMemcpyHostToDevice, ctx->stream));
                KE_HIP(ctx, hipMemcpyAsync((uint8_t *)meta + (size_t)m * 8, gidx.data(), (size_t)m * 8,
                                           hipMemcpyHostToDevice, ctx->stream));
                KeHashGroup g{(const uint8_t *)tmp, (const uint64_t *)meta, img_bytes,
                              (const int64_t *)((uint8_t *)meta + (size_t)m * 8), m, w, h, channels};
                KE_TRY(ke_launch_hash_group(ctx, g, d_ph, d_dh, nullptr, nullptr));
                KE_HIP(ctx, hipStreamSynchronize(ctx->stream));  // goff/gidx and the staging buffers are reused
            }
        }
    }
    ke_time_end(ctx, KE_T_HASH);
    if (phash_out && !p_dev)
        KE_HIP(ctx, hipMemcpyAsync(phash_out, d_ph, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (dhash_out && !d_dev)
        KE_HIP(ctx, hipMemcpyAsync(dhash_out, d_dh, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (margin_out && !m_dev)
        KE_HIP(ctx, hipMemcpyAsync(margin_out, ctx->margin_cur, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    KE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KE_OK;
}

KE_API int ke_hash_images(ke_ctx *ctx, const uint8_t *pixels, const uint64_t *offsets, const int32_t *widths,
                          const int32_t *heights, int32_t channels, int64_t n, uint64_t *phash_out,
                          uint64_t *dhash_out, int32_t *status_out) {
    return hash_images_impl(ctx, pixels, offsets, widths, heights, channels, n, phash_out, dhash_out, status_out, nullptr);
}

KE_API int ke_hash_images_ex(ke_ctx *ctx, const uint8_t *pixels, const uint64_t *offsets, const int32_t *widths,
                             const int32_t *heights, int32_t channels, int64_t n, uint64_t *phash_out,
                             uint64_t *dhash_out, int32_t *status_out, float *margin_out) {
    return hash_images_impl(ctx, pixels, offsets, widths, heights, channels, n, phash_out, dhash_out, status_out, margin_out);
}

// ---- pinned staging ---------------------------------------------------------------------------
static void stage_free(ke_ctx *ctx) {
    KeStage *st = ctx->stage;
    if (!st) return;
    (void)hipStreamSynchronize(ctx->stream);
    if (st->copy_stream) (void)hipStreamSynchronize(st->copy_stream);
    for (int k = 0; k < st->n_slots; ++k) {
        KeStageSlot &s = st->slot[k];
        if (s.h_px && s.px_external) { (void)hipHostUnregister(s.h_px); s.h_px = nullptr; }
        for (void *p : {(void *)s.h_px, (void *)s.h_meta, (void *)s.h_ph, (void *)s.h_dh, (void *)s.h_mg})
            if (p) (void)hipHostFree(p);
        for (void *p : {(void *)s.d_px, (void *)s.d_meta, (void *)s.d_ph, (void *)s.d_dh, (void *)s.d_mg})
            if (p) (void)hipFree(p);
        if (s.copied) (void)hipEventDestroy(s.copied);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    if (st->copy_stream) (void)hipStreamDestroy(st->copy_stream);
    delete st;
    ctx->stage = nullptr;
}

KE_API int ke_stage_destroy(ke_ctx *ctx) {
    if (!ctx) return KE_EINVAL;
    stage_free(ctx);
    return KE_OK;
}

static int stage_create(ke_ctx *ctx, void *const *external, size_t bytes_per_buffer, int64_t max_images, int32_t n_buffers);

KE_API int ke_stage_create(ke_ctx *ctx, size_t bytes_per_buffer, int64_t max_images, int32_t n_buffers) {
    return stage_create(ctx, nullptr, bytes_per_buffer, max_images, n_buffers);
}

KE_API int ke_stage_create_shared(ke_ctx *ctx, void *const *buffers, size_t bytes_per_buffer, int64_t max_images, int32_t n_buffers) {
    if (!ctx) return KE_EINVAL;
    if (!buffers) return ke_fail(ctx, KE_EINVAL, "NULL argument");
    for (int k = 0; k < n_buffers && k < KE_MAX_STAGE_SLOTS; ++k)
        if (!buffers[k] || ((uintptr_t)buffers[k] & 4095) || (bytes_per_buffer & 255))
            return ke_fail(ctx, KE_EINVAL, "shared staging buffers must be page-aligned and a multiple of 256 bytes long");
    return stage_create(ctx, buffers, bytes_per_buffer, max_images, n_buffers);
}

static int stage_create(ke_ctx *ctx, void *const *external, size_t bytes_per_buffer, int64_t max_images, int32_t n_buffers) {
    if (!ctx) return KE_EINVAL;
    if (bytes_per_buffer < 4096 || max_images <= 0 || n_buffers < 1 || n_buffers > KE_MAX_STAGE_SLOTS)
        return ke_fail(ctx, KE_EINVAL, "bad staging geometry (%zu bytes, %lld images, %d buffers)", bytes_per_buffer,
                       (long long)max_images, n_buffers);
    KE_HIP(ctx, hipSetDevice(ctx->device));
    stage_free(ctx);
    auto *st = new KeStage();
    ctx->stage = st;
    st->bytes = (bytes_per_buffer + 255) & ~(size_t)255;
    st->max_images = max_images;
    st->n_slots = n_buffers;
    auto fail = [&](const char *what) { stage_free(ctx); return ke_fail(ctx, KE_ENOMEM, "staging: %s failed", what); };
    if (hipStreamCreateWithFlags(&st->copy_stream, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate");
    for (int k = 0; k < n_buffers; ++k) {
        KeStageSlot &s = st->slot[k];
        const size_t m8 = (size_t)max_images * 8;
        if (external) {
            // the caller's pages (shared memory that decoder PROCESSES write into), page-locked and mapped for the copy engine
            if (hipHostRegister(external[k], st->bytes, hipHostRegisterDefault) != hipSuccess) return fail("hipHostRegister(pixels)");
            s.h_px = (uint8_t *)external[k];
            s.px_external = true;
        } else if (hipHostMalloc((void **)&s.h_px, st->bytes, hipHostMallocDefault) != hipSuccess) {
            return fail("hipHostMalloc(pixels)");
        }
        if (hipHostMalloc((void **)&s.h_meta, 2 * m8, hipHostMallocDefault) != hipSuccess) return fail("hipHostMalloc(meta)");
        if (hipHostMalloc((void **)&s.h_ph, m8, hipHostMallocDefault) != hipSuccess) return fail("hipHostMalloc");
        if (hipHostMalloc((void **)&s.h_dh, m8, hipHostMallocDefault) != hipSuccess) return fail("hipHostMalloc");
        if (hipHostMalloc((void **)&s.h_mg, m8 / 2, hipHostMallocDefault) != hipSuccess) return fail("hipHostMalloc");
        if (hipMalloc((void **)&s.d_px, st->bytes + 64) != hipSuccess) return fail("hipMalloc(pixels)");
        if (hipMalloc((void **)&s.d_meta, 2 * m8) != hipSuccess) return fail("hipMalloc(meta)");
        if (hipMalloc((void **)&s.d_ph, m8) != hipSuccess || hipMalloc((void **)&s.d_dh, m8) != hipSuccess ||
            hipMalloc((void **)&s.d_mg, m8 / 2) != hipSuccess)
            return fail("hipMalloc(results)");
        if (hipEventCreateWithFlags(&s.copied, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess)
            return fail("hipEventCreate");
    }
    return KE_OK;
}

static int stage_collect(ke_ctx *ctx, KeStageSlot &s) {
    if (!s.in_flight) return KE_OK;
    KE_HIP(ctx, hipEventSynchronize(s.done));
    if (s.user_ph) std::memcpy(s.user_ph, s.h_ph, (size_t)s.n * 8);
    if (s.user_dh) std::memcpy(s.user_dh, s.h_dh, (size_t)s.n * 8);
    if (s.user_mg) std::memcpy(s.user_mg, s.h_mg, (size_t)s.n * 4);
    s.in_flight = false;
    return KE_OK;
}

KE_API int ke_stage_acquire(ke_ctx *ctx, int32_t *slot_out, uint8_t **host_ptr_out, size_t *bytes_out) {
    if (!ctx || !slot_out || !host_ptr_out) return KE_EINVAL;
    KeStage *st = ctx->stage;
    if (!st) return ke_fail(ctx, KE_EINVAL, "ke_stage_create has not been called");
    const int k = st->next;
    st->next = (k + 1) % st->n_slots;
    KE_TRY(stage_collect(ctx, st->slot[k]));      // the buffer's previous batch: copy and kernels done, results handed over
    *slot_out = k;
    *host_ptr_out = st->slot[k].h_px;
    if (bytes_out) *bytes_out = st->bytes;
    return KE_OK;
}

KE_API int ke_stage_wait(ke_ctx *ctx, int32_t slot) {
    if (!ctx) return KE_EINVAL;
    KeStage *st = ctx->stage;
    if (!st || slot < -1 || slot >= st->n_slots) return ke_fail(ctx, KE_EINVAL, "bad staging slot %d", slot);
    if (slot >= 0) return stage_collect(ctx, st->slot[slot]);
    for (int k = 0; k < st->n_slots; ++k) KE_TRY(stage_collect(ctx, st->slot[(st->next + k) % st->n_slots]));   // oldest first
    return KE_OK;
}

KE_API int ke_stage_submit_hash(ke_ctx *ctx, int32_t slot, const uint64_t *offsets, const int32_t *widths, const int32_t *heights,
                                const int32_t *channels, int64_t n, uint64_t *phash_out, uint64_t *dhash_out,
                                int32_t *status_out, float *margin_out) {
    if (!ctx) return KE_EINVAL;
    KeStage *st = ctx->stage;
    if (!st || slot < 0 || slot >= st->n_slots) return ke_fail(ctx, KE_EINVAL, "bad staging slot %d", slot);
    KeStageSlot &s = st->slot[slot];
    if (s.in_flight) return ke_fail(ctx, KE_EINVAL, "staging slot %d is still in flight", slot);
    if (n < 0 || n > st->max_images) return ke_fail(ctx, KE_EINVAL, "batch of %lld images exceeds the staging limit %lld", (long long)n, (long long)st->max_images);
    if (n > 0 && (!offsets || !widths || !heights || !channels)) return ke_fail(ctx, KE_EINVAL, "offsets/widths/heights/channels must be non-NULL");
    if (margin_out && !phash_out) return ke_fail(ctx, KE_EINVAL, "margin_out needs phash_out");
    for (const void *p : {(const void *)phash_out, (const void *)dhash_out, (const void *)margin_out})
        if (p && ke_is_device_ptr(p)) return ke_fail(ctx, KE_EINVAL, "staged results are returned to host arrays");
    if (n == 0) return KE_OK;
    KE_HIP(ctx, hipSetDevice(ctx->device));
    trim_coeff_cache(ctx);
    // size groups over (width, height, channels); metadata goes into the slot's pinned block so nothing here has to
    // outlive the call on the stack
    std::map<std::tuple<int, int, int>, std::vector<int64_t>> groups;
    size_t used = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int c = channels[i];
        const bool ok = widths[i] > 0 && heights[i] > 0 && (c == 1 || c == 3 || c == 4);
        const size_t bytes = ok ? (size_t)widths[i] * heights[i] * c : 0;
        const bool inside = ok && offsets[i] <= st->bytes && bytes <= st->bytes - offsets[i];
        if (status_out) status_out[i] = inside ? KE_IMG_OK : KE_IMG_BAD_SHAPE;
        if (!inside) continue;
        groups[std::make_tuple(widths[i], heights[i], c)].push_back(i);
        used = std::max(used, (size_t)offsets[i] + bytes);
    }
    size_t cursor = 0;
    std::vector<std::pair<size_t, size_t>> spans;
    for (auto &kv : groups) {
        const std::vector<int64_t> &idx = kv.second;
        spans.emplace_back(cursor, idx.size());
        for (size_t k = 0; k < idx.size(); ++k) {
            s.h_meta[cursor + k] = offsets[idx[k]];
            s.h_meta[cursor + idx.size() + k] = (uint64_t)idx[k];
        }
        cursor += 2 * idx.size();
    }
    // copy stream: pixels + metadata to the slot's device twin
    if (used) KE_HIP(ctx, hipMemcpyAsync(s.d_px, s.h_px, used, hipMemcpyHostToDevice, st->copy_stream));
    if (cursor) KE_HIP(ctx, hipMemcpyAsync(s.d_meta, s.h_meta, cursor * 8, hipMemcpyHostToDevice, st->copy_stream));
    KE_HIP(ctx, hipEventRecord(s.copied, st->copy_stream));
    // compute stream: behind the copy, and behind whatever batch was submitted before
    KE_HIP(ctx, hipStreamWaitEvent(ctx->stream, s.copied, 0));
    if (phash_out) KE_HIP(ctx, hipMemsetAsync(s.d_ph, 0, (size_t)n * 8, ctx->stream));
    if (dhash_out) KE_HIP(ctx, hipMemsetAsync(s.d_dh, 0, (size_t)n * 8, ctx->stream));
    if (margin_out) KE_HIP(ctx, hipMemsetAsync(s.d_mg, 0, (size_t)n * 4, ctx->stream));
    MarginScope margin_scope{ctx};
    ctx->margin_cur = margin_out ? s.d_mg : nullptr;
    ke_time_begin(ctx, KE_T_HASH);
    size_t gi = 0;
    for (auto &kv : groups) {
        const int w = std::get<0>(kv.first), h = std::get<1>(kv.first), c = std::get<2>(kv.first);
        const uint64_t *d_off = s.d_meta + spans[gi].first;
        const int64_t m = (int64_t)spans[gi].second;
        KeHashGroup g{s.d_px, d_off, (uint64_t)w * h * c, (const int64_t *)(d_off + m), m, w, h, c};
        for (int64_t idx : kv.second)
            if (((uintptr_t)s.d_px + offsets[idx]) % 4 != 0) { g.misaligned = true; break; }
        KE_TRY(ke_launch_hash_group(ctx, g, phash_out ? s.d_ph : nullptr, dhash_out ? s.d_dh : nullptr, nullptr, nullptr));
        ++gi;
    }
    ke_time_end(ctx, KE_T_HASH);
    if (phash_out) KE_HIP(ctx, hipMemcpyAsync(s.h_ph, s.d_ph, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (dhash_out) KE_HIP(ctx, hipMemcpyAsync(s.h_dh, s.d_dh, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (margin_out) KE_HIP(ctx, hipMemcpyAsync(s.h_mg, s.d_mg, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    KE_HIP(ctx, hipEventRecord(s.done, ctx->stream));
    s.user_ph = phash_out; s.user_dh = dhash_out; s.user_mg = margin_out; s.n = n;
    s.in_flight = true;
    return KE_OK;
}

// ---- scan ------------------------------------------------------------------------------------
KE_API int ke_hamming_scan(ke_ctx *ctx, const uint64_t *hashes, const int64_t *ids, const int64_t *sizes, int64_t n,
                           int32_t part_index, int32_t part_count, int32_t threshold, int32_t band_bits,
                           int32_t band_count, double size_ratio, int64_t bucket_pair_cap, ke_edge *edges_out,
                           int64_t capacity, int64_t *n_edges_out, uint64_t *counters_out) {
    if (!ctx) return KE_EINVAL;
    if (n < 0 || (n > 0 && !hashes)) return ke_fail(ctx, KE_EINVAL, "hashes is NULL");
    if (!n_edges_out) return ke_fail(ctx, KE_EINVAL, "n_edges_out is NULL");
    // DuplicateScanConfig.__post_init__ / DuplicateScanner.__init__ (src/dup/scanner.py:155-166, 208)
    if (band_bits <= 0) return ke_fail(ctx, KE_EINVAL, "band_bits must be positive");
    if (band_count <= 0) return ke_fail(ctx, KE_EINVAL, "band_count must be positive");
    if ((int64_t)band_bits * band_count > 64) return ke_fail(ctx, KE_EINVAL, "band config too large");
    if (threshold < 0 || threshold > 64) return ke_fail(ctx, KE_EINVAL, "hamming_threshold must be in [0, 64]");
    if (part_count <= 0 || part_index < 0 || part_index >= part_count)
        return ke_fail(ctx, KE_EINVAL, "bad shard %d/%d", part_index, part_count);
    if (capacity < 0 || (capacity > 0 && !edges_out)) return ke_fail(ctx, KE_EINVAL, "edges_out is NULL");
    *n_edges_out = 0;
    if (counters_out) std::memset(counters_out, 0, 4 * sizeof(uint64_t));
    if (n < 2) return KE_OK;
    KE_HIP(ctx, hipSetDevice(ctx->device));
    const void *d_h, *d_ids, *d_sizes;
    // three inputs share one staging buffer when they come from the host
    const bool any_host = !ke_is_device_ptr(hashes) || (ids && !ke_is_device_ptr(ids)) || (sizes && !ke_is_device_ptr(sizes));
    if (any_host) {
        void *base;
        KE_TRY(ke_reserve(ctx, KE_BUF_SCAN_IN, (size_t)n * 24, &base));
        auto stage = [&](const void *src, size_t slot, const void **dst) -> int {
            if (!src) { *dst = nullptr; return KE_OK; }
            if (ke_is_device_ptr(src)) { *dst = src; return KE_OK; }
            void *d = (uint8_t *)base + slot * (size_t)n * 8;
            KE_HIP(ctx, hipMemcpyAsync(d, src, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
            *dst = d;
            return KE_OK;
        };
        KE_TRY(stage(hashes, 0, &d_h));
        KE_TRY(stage(ids, 1, &d_ids));
        KE_TRY(stage(sizes, 2, &d_sizes));
    } else {
        d_h = hashes; d_ids = ids; d_sizes = sizes;
    }
    const bool edges_dev = capacity > 0 && ke_is_device_ptr(edges_out);
    ke_edge *d_edges = edges_out;
    void *tmp;
    if (capacity > 0 && !edges_dev) {
        KE_TRY(ke_reserve(ctx, KE_BUF_SCAN_EDGES, (size_t)capacity * sizeof(ke_edge), &tmp));
        d_edges = (ke_edge *)tmp;
    }
    KE_TRY(ke_reserve(ctx, KE_BUF_SCAN_CNT, 4 * sizeof(unsigned long long), &tmp));
    unsigned long long *d_cnt = (unsigned long long *)tmp;
    KE_HIP(ctx, hipMemsetAsync(d_cnt, 0, 4 * sizeof(unsigned long long), ctx->stream));
    unsigned long long pairs = 0;
    KE_TRY(ke_launch_scan(ctx, (const uint64_t *)d_h, (const int64_t *)d_ids, (const int64_t *)d_sizes, n, part_index,
                          part_count, threshold, band_bits, band_count, size_ratio, bucket_pair_cap, d_edges, capacity,
                          d_cnt, &pairs, counters_out != nullptr));
    unsigned long long h_cnt[4];
    KE_HIP(ctx, hipMemcpyAsync(h_cnt, d_cnt, sizeof h_cnt, hipMemcpyDeviceToHost, ctx->stream));
    KE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_edges_out = (int64_t)h_cnt[2];
    if (h_cnt[0] != pairs)    // the kernel's own count of the pairs its tiles stand for against the host's closed form
        return ke_fail(ctx, KE_EHIP, "scan evaluated %llu pairs, expected %llu", h_cnt[0], pairs);
    if (counters_out)
        for (int k = 0; k < 4; ++k) counters_out[k] = h_cnt[k];
    if (capacity > 0 && !edges_dev) {
        const int64_t m = std::min<int64_t>(capacity, (int64_t)h_cnt[2]);
        if (m > 0) {
            KE_HIP(ctx, hipMemcpyAsync(edges_out, d_edges, (size_t)m * sizeof(ke_edge), hipMemcpyDeviceToHost, ctx->stream));
            KE_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    return KE_OK;
}

KE_API int ke_band_pairs_after_size(ke_ctx *ctx, const uint64_t *hashes, const int64_t *sizes, int64_t n, int32_t band_bits,
                                    int32_t band_count, double size_ratio, int64_t bucket_pair_cap, uint64_t *count_out) {
    if (!ctx) return KE_EINVAL;
    if (!count_out) return ke_fail(ctx, KE_EINVAL, "count_out is NULL");
    if (n < 0 || (n > 0 && (!hashes || !sizes))) return ke_fail(ctx, KE_EINVAL, "hashes / sizes is NULL");
    if (band_bits <= 0 || band_count <= 0 || (int64_t)band_bits * band_count > 64) return ke_fail(ctx, KE_EINVAL, "bad band config");
    if (!(size_ratio > 0.0)) return ke_fail(ctx, KE_EINVAL, "size_ratio must be positive (without a size filter the counter equals the bucket pairs)");
    *count_out = 0;
    if (n < 2) return KE_OK;
    KE_HIP(ctx, hipSetDevice(ctx->device));
    const void *d_h = hashes, *d_s = sizes;
    if (!ke_is_device_ptr(hashes) || !ke_is_device_ptr(sizes)) {
        void *base;
        KE_TRY(ke_reserve(ctx, KE_BUF_SCAN_IN, (size_t)n * 24, &base));
        if (!ke_is_device_ptr(hashes)) {
            KE_HIP(ctx, hipMemcpyAsync(base, hashes, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
            d_h = base;
        }
        if (!ke_is_device_ptr(sizes)) {
            void *d = (uint8_t *)base + (size_t)n * 16;
            KE_HIP(ctx, hipMemcpyAsync(d, sizes, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
            d_s = d;
        }
    }
    void *tmp;
    KE_TRY(ke_reserve(ctx, KE_BUF_SCAN_CNT, 4 * sizeof(unsigned long long), &tmp));
    unsigned long long *d_cnt = (unsigned long long *)tmp;
    KE_HIP(ctx, hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), ctx->stream));
    KE_TRY(ke_launch_band_pairs_after_size(ctx, (const uint64_t *)d_h, (const int64_t *)d_s, n, band_bits, band_count, size_ratio,
                                           bucket_pair_cap, d_cnt));
    unsigned long long h = 0;
    KE_HIP(ctx, hipMemcpyAsync(&h, d_cnt, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    KE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *count_out = h;
    return KE_OK;
}

// ---- ssim ------------------------------------------------------------------------------------
KE_API int ke_ssim_set_mode(ke_ctx *ctx, int32_t mode) {
    if (!ctx) return KE_EINVAL;
    if (mode != KE_SSIM_FAST && mode != KE_SSIM_EXACT) return ke_fail(ctx, KE_EINVAL, "unknown SSIM mode %d", mode);
    ctx->ssim_exact = mode == KE_SSIM_EXACT;
    return KE_OK;
}

KE_API int ke_ssim_pairs_uniform(ke_ctx *ctx, const uint8_t *images, int64_t n_images, int32_t width, int32_t height,
                                 int32_t channels, const int64_t *pair_a, const int64_t *pair_b, int64_t n_pairs,
                                 double *ssim_out) {
    if (!ctx) return KE_EINVAL;
    if (n_pairs < 0 || n_images < 0) return ke_fail(ctx, KE_EINVAL, "negative count");
    if (n_pairs == 0) return KE_OK;
    if (!images || !pair_a || !pair_b || !ssim_out) return ke_fail(ctx, KE_EINVAL, "NULL argument");
    if (width <= 0 || height <= 0) return ke_fail(ctx, KE_EINVAL, "width/height must be positive");
    if (channels != 1 && channels != 3 && channels != 4) return ke_fail(ctx, KE_EINVAL, "channels must be 1, 3 or 4");
    KE_HIP(ctx, hipSetDevice(ctx->device));
    if (!ke_is_device_ptr(pair_a) && !ke_is_device_ptr(pair_b)) {
        for (int64_t k = 0; k < n_pairs; ++k)
            if (pair_a[k] < 0 || pair_a[k] >= n_images || pair_b[k] < 0 || pair_b[k] >= n_images)
                return ke_fail(ctx, KE_EINVAL, "pair %lld indexes outside [0,%lld)", (long long)k, (long long)n_images);
    }
    const void *d_img, *d_pa, *d_pb;
    KE_TRY(ke_to_device(ctx, images, (size_t)n_images * width * height * channels, KE_BUF_SSIM_IN, &d_img));
    void *aux;
    KE_TRY(ke_reserve(ctx, KE_BUF_SSIM_AUX, (size_t)n_pairs * 24, &aux));
    auto stage = [&](const int64_t *src, int slot, const void **dst) -> int {
        if (ke_is_device_ptr(src)) { *dst = src; return KE_OK; }
        void *d = (uint8_t *)aux + (size_t)slot * n_pairs * 8;
        KE_HIP(ctx, hipMemcpyAsync(d, src, (size_t)n_pairs * 8, hipMemcpyHostToDevice, ctx->stream));
        *dst = d;
        return KE_OK;
    };
    KE_TRY(stage(pair_a, 0, &d_pa));
    KE_TRY(stage(pair_b, 1, &d_pb));
    const bool out_dev = ke_is_device_ptr(ssim_out);
    double *d_out = out_dev ? ssim_out : (double *)((uint8_t *)aux + (size_t)2 * n_pairs * 8);
    ke_time_begin(ctx, KE_T_SSIM);
    KE_TRY(ke_launch_ssim(ctx, (const uint8_t *)d_img, width, height, channels, (const int64_t *)d_pa,
                          (const int64_t *)d_pb, n_pairs, d_out));
    ke_time_end(ctx, KE_T_SSIM);
    if (!out_dev) {
        KE_HIP(ctx, hipMemcpyAsync(ssim_out, d_out, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, ctx->stream));
        KE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return KE_OK;
}

static int fit_box(int width, int height, int out_w, int out_h, float box[4]) {
    // PIL/ImageOps.py fit(): the same double-precision steps in the same order
    const double live_ratio = (double)width / (double)height, out_ratio = (double)out_w / (double)out_h;
    double cw, chh;
    if (live_ratio == out_ratio) { cw = width; chh = height; }
    else if (live_ratio >= out_ratio) { cw = out_ratio * height; chh = height; }
    else { cw = width; chh = width / out_ratio; }
    const double left = (width - cw) * 0.5, top = (height - chh) * 0.5;
    box[0] = (float)left; box[1] = (float)top; box[2] = (float)(left + cw); box[3] = (float)(top + chh);
    return (box[0] < 0 || box[1] < 0 || box[2] > width || box[3] > height) ? KE_EINVAL : KE_OK;
}

KE_API int ke_ssim_pairs(ke_ctx *ctx, const uint8_t *pixels, const uint64_t *offsets, const int32_t *widths, const int32_t *heights,
                         int32_t channels, int64_t n_images, const int64_t *pair_a, const int64_t *pair_b, int64_t n_pairs,
                         double *ssim_out, int32_t *status_out) {
    if (!ctx) return KE_EINVAL;
    if (n_pairs < 0 || n_images < 0) return ke_fail(ctx, KE_EINVAL, "negative count");
    if (n_pairs == 0) return KE_OK;
    if (!pixels || !widths || !heights || !pair_a || !pair_b || !ssim_out) return ke_fail(ctx, KE_EINVAL, "NULL argument");
    if (channels != 1 && channels != 3 && channels != 4) return ke_fail(ctx, KE_EINVAL, "channels must be 1, 3 or 4");
    for (const void *p : {(const void *)offsets, (const void *)widths, (const void *)heights, (const void *)pair_a, (const void *)pair_b,
                          (const void *)ssim_out, (const void *)status_out})
        if (p && ke_is_device_ptr(p)) return ke_fail(ctx, KE_EINVAL, "offsets/widths/heights/pairs/outputs are host arrays");
    KE_HIP(ctx, hipSetDevice(ctx->device));
    trim_coeff_cache(ctx);
    const double nan = std::nan("");
    // byte offsets; classify the pairs
    std::vector<uint64_t> off((size_t)n_images);
    uint64_t run = 0;
    for (int64_t i = 0; i < n_images; ++i) {
        off[i] = offsets ? offsets[i] : run;
        if (widths[i] > 0 && heights[i] > 0) run += (uint64_t)widths[i] * heights[i] * channels;
    }
    uint64_t total_bytes = 0;
    for (int64_t i = 0; i < n_images; ++i)
        if (widths[i] > 0 && heights[i] > 0) total_bytes = std::max<uint64_t>(total_bytes, off[i] + (uint64_t)widths[i] * heights[i] * channels);
    std::map<std::pair<int, int>, std::vector<int64_t>> by_size;        // common (w, h) -> pairs
    for (int64_t k = 0; k < n_pairs; ++k) {
        ssim_out[k] = nan;
        const int64_t ia = pair_a[k], ib = pair_b[k];
        const bool ok = ia >= 0 && ia < n_images && ib >= 0 && ib < n_images && widths[ia] > 0 && heights[ia] > 0 && widths[ib] > 0 &&
                        heights[ib] > 0;
        if (!ok) { if (status_out) status_out[k] = KE_PAIR_BAD_IMAGE; continue; }
        const int w = std::min(widths[ia], widths[ib]), h = std::min(heights[ia], heights[ib]);   // src/dup/refine.py:45-47
        if (w < 7 || h < 7) { if (status_out) status_out[k] = KE_PAIR_TOO_SMALL; continue; }
        if (status_out) status_out[k] = KE_PAIR_OK;
        by_size[{w, h}].push_back(k);
    }
    if (by_size.empty()) return KE_OK;
    const void *d_px;
    KE_TRY(ke_to_device(ctx, pixels, (size_t)total_bytes, KE_BUF_PIXELS, &d_px));
    ke_time_begin(ctx, KE_T_SSIM);
    for (auto &kv : by_size) {
        const int w = kv.first.first, h = kv.first.second;
        const std::vector<int64_t> &ks = kv.second;
        // plane stack of this common size: the images of one source size sit next to each other, so a fit launch writes
        // its whole group straight into place
        std::map<std::pair<int, int>, std::vector<int64_t>> groups;     // source (w, h) -> images, first use order
        std::map<int64_t, int64_t> slot;                                 // image -> position inside its group
        for (int64_t k : ks)
            for (int64_t i : {pair_a[k], pair_b[k]})
                if (!slot.count(i)) {
                    auto &g = groups[{widths[i], heights[i]}];
                    slot[i] = (int64_t)g.size();
                    g.push_back(i);
                }
        int64_t base = 0;
        std::map<std::pair<int, int>, int64_t> group_base;
        for (auto &g : groups) { group_base[g.first] = base; base += (int64_t)g.second.size(); }
        const size_t plane = (size_t)w * h;
        void *planes, *meta, *aux;
        KE_TRY(ke_reserve(ctx, KE_BUF_SSIM_IN, (size_t)base * plane, &planes));
        KE_TRY(ke_reserve(ctx, KE_BUF_META, (size_t)base * 8, &meta));
        std::vector<uint64_t> meta_h((size_t)base);
        for (auto &g : groups)
            for (size_t j = 0; j < g.second.size(); ++j) meta_h[(size_t)group_base[g.first] + j] = off[g.second[j]];
        KE_HIP(ctx, hipMemcpyAsync(meta, meta_h.data(), (size_t)base * 8, hipMemcpyHostToDevice, ctx->stream));
        KE_HIP(ctx, hipStreamSynchronize(ctx->stream));                  // meta_h is a local
        for (auto &g : groups) {
            const int sw = g.first.first, sh = g.first.second;
            const int64_t gb = group_base[g.first], m = (int64_t)g.second.size();
            KeHashGroup hg{(const uint8_t *)d_px, (const uint64_t *)meta + gb, (uint64_t)sw * sh * channels, nullptr, m, sw, sh, channels};
            for (int64_t i : g.second)
                if (((uintptr_t)d_px + off[i]) % 4 != 0) { hg.misaligned = true; break; }
            float box[4];
            if (fit_box(sw, sh, w, h, box) != KE_OK) return ke_fail(ctx, KE_EINVAL, "crop box outside the image");
            KE_TRY(ke_launch_resize_group(ctx, hg, w, h, KE_FILTER_BICUBIC, (uint8_t *)planes + (size_t)gb * plane, box));
        }
        // pair indices into the plane stack, then one SSIM launch
        const int64_t np = (int64_t)ks.size();
        KE_TRY(ke_reserve(ctx, KE_BUF_SSIM_AUX, (size_t)np * 24, &aux));
        std::vector<int64_t> idx((size_t)2 * np);
        for (int64_t j = 0; j < np; ++j) {
            const int64_t ia = pair_a[ks[j]], ib = pair_b[ks[j]];
            idx[j] = group_base[{widths[ia], heights[ia]}] + slot[ia];
            idx[np + j] = group_base[{widths[ib], heights[ib]}] + slot[ib];
        }
        KE_HIP(ctx, hipMemcpyAsync(aux, idx.data(), (size_t)2 * np * 8, hipMemcpyHostToDevice, ctx->stream));
        double *d_out = (double *)((uint8_t *)aux + (size_t)2 * np * 8);
        KE_TRY(ke_launch_ssim(ctx, (const uint8_t *)planes, w, h, 1, (const int64_t *)aux, (const int64_t *)aux + np, np, d_out));
        std::vector<double> scores((size_t)np);
        KE_HIP(ctx, hipMemcpyAsync(scores.data(), d_out, (size_t)np * 8, hipMemcpyDeviceToHost, ctx->stream));
        KE_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int64_t j = 0; j < np; ++j) ssim_out[ks[j]] = scores[j];
    }
    ke_time_end(ctx, KE_T_SSIM);
    return KE_OK;
}

// ---- shipped refine stage: thumbnails, tile aHash, pixel MAE --------------------------------------
static int resize_luma_common(ke_ctx *ctx, const uint8_t *pixels, int64_t n, int32_t width, int32_t height, int32_t channels,
                              int32_t out_w, int32_t out_h, int32_t filter, const float *box, uint8_t *tiles_out);

KE_API int ke_resize_luma_uniform(ke_ctx *ctx, const uint8_t *pixels, int64_t n, int32_t width, int32_t height,
                                  int32_t channels, int32_t out_w, int32_t out_h, int32_t filter, uint8_t *tiles_out) {
    return resize_luma_common(ctx, pixels, n, width, height, channels, out_w, out_h, filter, nullptr, tiles_out);
}

KE_API int ke_fit_luma_uniform(ke_ctx *ctx, const uint8_t *pixels, int64_t n, int32_t width, int32_t height,
                               int32_t channels, int32_t out_w, int32_t out_h, int32_t filter, uint8_t *tiles_out) {
    if (!ctx) return KE_EINVAL;
    if (width <= 0 || height <= 0 || out_w <= 0 || out_h <= 0) return ke_fail(ctx, KE_EINVAL, "sizes must be positive");
    float box[4];
    if (fit_box(width, height, out_w, out_h, box) != KE_OK)                  // Pillow raises ValueError here
        return ke_fail(ctx, KE_EINVAL, "crop box outside the image");
    return resize_luma_common(ctx, pixels, n, width, height, channels, out_w, out_h, filter, box, tiles_out);
}

static int resize_luma_common(ke_ctx *ctx, const uint8_t *pixels, int64_t n, int32_t width, int32_t height, int32_t channels,
                              int32_t out_w, int32_t out_h, int32_t filter, const float *box, uint8_t *tiles_out) {
    if (!ctx) return KE_EINVAL;
    if (n < 0 || (n > 0 && (!pixels || !tiles_out))) return ke_fail(ctx, KE_EINVAL, "NULL argument");
    if (width <= 0 || height <= 0 || out_w <= 0 || out_h <= 0) return ke_fail(ctx, KE_EINVAL, "sizes must be positive");
    if (channels != 1 && channels != 3 && channels != 4) return ke_fail(ctx, KE_EINVAL, "channels must be 1, 3 or 4");
    if (filter != KE_FILTER_LANCZOS && filter != KE_FILTER_BILINEAR && filter != KE_FILTER_BICUBIC)
        return ke_fail(ctx, KE_EINVAL, "unknown filter %d", filter);
    if (n == 0) return KE_OK;
    KE_HIP(ctx, hipSetDevice(ctx->device));
    trim_coeff_cache(ctx);
    const size_t img_bytes = (size_t)width * height * channels, out_bytes = (size_t)out_w * out_h;
    const bool in_dev = ke_is_device_ptr(pixels), out_dev = ke_is_device_ptr(tiles_out);
    const int64_t chunk = in_dev ? n : std::max<int64_t>(1, (int64_t)(kStageBytes / img_bytes));
    ke_time_begin(ctx, KE_T_HASH);
    for (int64_t f = 0; f < n; f += chunk) {
        const int64_t m = std::min(chunk, n - f);
        const void *d_px;
        KE_TRY(ke_to_device(ctx, pixels + (size_t)f * img_bytes, (size_t)m * img_bytes, KE_BUF_PIXELS, &d_px));
        uint8_t *d_out = tiles_out + (size_t)f * out_bytes;
        void *tmp;
        if (!out_dev) { KE_TRY(ke_reserve(ctx, KE_BUF_SSIM_IN, (size_t)m * out_bytes, &tmp)); d_out = (uint8_t *)tmp; }
        KeHashGroup g{(const uint8_t *)d_px, nullptr, img_bytes, nullptr, m, width, height, channels};
        KE_TRY(ke_launch_resize_group(ctx, g, out_w, out_h, filter, d_out, box));
        if (!out_dev)
            KE_HIP(ctx, hipMemcpyAsync(tiles_out + (size_t)f * out_bytes, d_out, (size_t)m * out_bytes, hipMemcpyDeviceToHost, ctx->stream));
        if (!in_dev || !out_dev) KE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    ke_time_end(ctx, KE_T_HASH);
    return KE_OK;
}

KE_API int ke_tile_ahash(ke_ctx *ctx, const uint8_t *tiles, int64_t n, int32_t grid, int32_t tile, uint64_t *bits_out) {
    if (!ctx) return KE_EINVAL;
    if (n < 0 || (n > 0 && (!tiles || !bits_out))) return ke_fail(ctx, KE_EINVAL, "NULL argument");
    if (grid <= 0 || tile <= 0 || (int64_t)grid * tile > 1024 || grid > 64) return ke_fail(ctx, KE_EINVAL, "bad grid/tile %d/%d", grid, tile);
    if (n == 0) return KE_OK;
    KE_HIP(ctx, hipSetDevice(ctx->device));
    const size_t side = (size_t)grid * tile, words = (side * side + 63) / 64;
    const void *d_t;
    KE_TRY(ke_to_device(ctx, tiles, (size_t)n * side * side, KE_BUF_SSIM_IN, &d_t));
    const bool out_dev = ke_is_device_ptr(bits_out);
    uint64_t *d_bits = bits_out;
    void *tmp;
    if (!out_dev) { KE_TRY(ke_reserve(ctx, KE_BUF_OUT0, (size_t)n * words * 8, &tmp)); d_bits = (uint64_t *)tmp; }
    KE_TRY(ke_launch_tile_ahash(ctx, (const uint8_t *)d_t, n, grid, tile, d_bits));
    if (!out_dev) {
        KE_HIP(ctx, hipMemcpyAsync(bits_out, d_bits, (size_t)n * words * 8, hipMemcpyDeviceToHost, ctx->stream));
        KE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return KE_OK;
}

KE_API int ke_sad_pairs(ke_ctx *ctx, const uint8_t *thumbs, int64_t n_thumbs, int64_t pixels, const int64_t *pair_a,
                        const int64_t *pair_b, int64_t n_pairs, uint64_t *sad_out) {
    if (!ctx) return KE_EINVAL;
    if (n_pairs < 0 || n_thumbs < 0 || pixels <= 0) return ke_fail(ctx, KE_EINVAL, "bad counts");
    if (n_pairs == 0) return KE_OK;
    if (!thumbs || !pair_a || !pair_b || !sad_out) return ke_fail(ctx, KE_EINVAL, "NULL argument");
    KE_HIP(ctx, hipSetDevice(ctx->device));
    if (!ke_is_device_ptr(pair_a) && !ke_is_device_ptr(pair_b))
        for (int64_t k = 0; k < n_pairs; ++k)
            if (pair_a[k] < 0 || pair_a[k] >= n_thumbs || pair_b[k] < 0 || pair_b[k] >= n_thumbs)
                return ke_fail(ctx, KE_EINVAL, "pair %lld indexes outside [0,%lld)", (long long)k, (long long)n_thumbs);
    const void *d_t, *d_pa, *d_pb;
    KE_TRY(ke_to_device(ctx, thumbs, (size_t)n_thumbs * pixels, KE_BUF_SSIM_IN, &d_t));
    void *aux;
    KE_TRY(ke_reserve(ctx, KE_BUF_SSIM_AUX, (size_t)n_pairs * 24, &aux));
    auto stage = [&](const int64_t *src, int slot, const void **dst) -> int {
        if (ke_is_device_ptr(src)) { *dst = src; return KE_OK; }
        void *d = (uint8_t *)aux + (size_t)slot * n_pairs * 8;
        KE_HIP(ctx, hipMemcpyAsync(d, src, (size_t)n_pairs * 8, hipMemcpyHostToDevice, ctx->stream));
        *dst = d;
        return KE_OK;
    };
    KE_TRY(stage(pair_a, 0, &d_pa));
    KE_TRY(stage(pair_b, 1, &d_pb));
    const bool out_dev = ke_is_device_ptr(sad_out);
    uint64_t *d_out = out_dev ? sad_out : (uint64_t *)((uint8_t *)aux + (size_t)2 * n_pairs * 8);
    KE_TRY(ke_launch_sad_pairs(ctx, (const uint8_t *)d_t, pixels, (const int64_t *)d_pa, (const int64_t *)d_pb, n_pairs, d_out));
    if (!out_dev) {
        KE_HIP(ctx, hipMemcpyAsync(sad_out, d_out, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, ctx->stream));
        KE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return KE_OK;
}

// ---- synthetic corpus --------------------------------------------------------------------------
KE_API int ke_synth_rgb(ke_ctx *ctx, uint64_t seed, int64_t first_index, int64_t n, int32_t width, int32_t height,
                        uint8_t *rgb_out) {
    if (!ctx) return KE_EINVAL;
    if (n < 0 || width <= 0 || height <= 0 || first_index < 0 || (n > 0 && !rgb_out))
        return ke_fail(ctx, KE_EINVAL, "bad synth arguments");
    if (n == 0) return KE_OK;
    KE_HIP(ctx, hipSetDevice(ctx->device));
    const size_t img_bytes = (size_t)width * height * 3;
    if (ke_is_device_ptr(rgb_out)) {
        ke_time_begin(ctx, KE_T_SYNTH);
        KE_TRY(ke_launch_synth_rgb(ctx, seed, first_index, nullptr, n, width, height, rgb_out));
        ke_time_end(ctx, KE_T_SYNTH);
        return KE_OK;
    }
    const int64_t chunk = std::max<int64_t>(1, (int64_t)(kStageBytes / img_bytes));
    for (int64_t f = 0; f < n; f += chunk) {
        const int64_t m = std::min(chunk, n - f);
        void *d;
        KE_TRY(ke_reserve(ctx, KE_BUF_PIXELS, (size_t)m * img_bytes, &d));
        KE_TRY(ke_launch_synth_rgb(ctx, seed, first_index + f, nullptr, m, width, height, (uint8_t *)d));
        KE_HIP(ctx, hipMemcpyAsync(rgb_out + (size_t)f * img_bytes, d, (size_t)m * img_bytes, hipMemcpyDeviceToHost, ctx->stream));
        KE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return KE_OK;
}

KE_API int ke_synth_rgb_indexed(ke_ctx *ctx, uint64_t seed, const int64_t *indices, int64_t n, int32_t width, int32_t height,
                                uint8_t *rgb_out) {
    if (!ctx) return KE_EINVAL;
    if (n < 0 || width <= 0 || height <= 0 || (n > 0 && (!rgb_out || !indices))) return ke_fail(ctx, KE_EINVAL, "bad synth arguments");
    if (n == 0) return KE_OK;
    if (!ke_is_device_ptr(rgb_out)) return ke_fail(ctx, KE_EINVAL, "ke_synth_rgb_indexed writes device memory only");
    KE_HIP(ctx, hipSetDevice(ctx->device));
    const void *d_idx;
    KE_TRY(ke_to_device(ctx, indices, (size_t)n * 8, KE_BUF_META, &d_idx));
    ke_time_begin(ctx, KE_T_SYNTH);
    KE_TRY(ke_launch_synth_rgb(ctx, seed, 0, (const int64_t *)d_idx, n, width, height, rgb_out));
    ke_time_end(ctx, KE_T_SYNTH);
    return KE_OK;
}

KE_API int ke_synth_hashes(ke_ctx *ctx, uint64_t seed, int64_t n, uint64_t *hashes_out) {
    if (!ctx) return KE_EINVAL;
    if (n < 0 || (n > 0 && !hashes_out)) return ke_fail(ctx, KE_EINVAL, "bad synth arguments");
    if (n == 0) return KE_OK;
    KE_HIP(ctx, hipSetDevice(ctx->device));
    if (ke_is_device_ptr(hashes_out)) return ke_launch_synth_hashes(ctx, seed, n, hashes_out);
    void *d;
    KE_TRY(ke_reserve(ctx, KE_BUF_SCAN_IN, (size_t)n * 8, &d));
    KE_TRY(ke_launch_synth_hashes(ctx, seed, n, (uint64_t *)d));
    KE_HIP(ctx, hipMemcpyAsync(hashes_out, d, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    KE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return KE_OK;
}
