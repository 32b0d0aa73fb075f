// ke_ssim.hip -- K5: windowed SSIM of candidate pairs for gfx950.
//
// Replaces dup.refine._compute_ssim (src/dup/refine.py:44-52) for equally sized images:
// luma exactly as convert("L"), x = L/255 in float32, then
// skimage.metrics.structural_similarity(x, y, data_range=1.0): 7x7 uniform window
// (scipy.ndimage.uniform_filter: axis 0 then axis 1, each pass summed in fp64 and stored as
// float32), K1=.01, K2=.03, sample covariance (49/48), mean over the image cropped by 3 px.
// Every float32 elementwise step is kept a separate rounding (the library is compiled with
// -ffp-contract=off) so the map matches NumPy's; window sums of seven float32 values are
// exact in fp64, so their order does not matter.
//
// Layout: one workgroup per (pair, 16x64 tile of interior pixels).  The tile plus a 3 px halo
// of both images is converted once to float32 in LDS; the vertical pass writes five window
// means per halo column to LDS; the horizontal pass finishes the means, evaluates S and the
// workgroup writes one fp64 partial sum.  A second tiny kernel adds the partials of a pair in
// a fixed order (bitwise reproducible) and divides.
#include "ke_internal.h"

namespace {

constexpr int kTH = 16, kTW = 64;
constexpr int kHH = kTH + 6, kHW = kTW + 6;   // halo tile
constexpr int kPitch = kHW + 1;

struct SsimArgs {
    const uint8_t *images;
    const int64_t *pa, *pb;
    int w, h, ch;
    int tiles_x, tiles_y;
    double *partial;   // [pair][tiles_y*tiles_x]
};

__device__ __forceinline__ float unit_luma(const uint8_t *p, int ch) {
    const int l = ch == 1 ? p[0] : (int)((19595u * p[0] + 38470u * p[1] + 7471u * p[2] + 0x8000u) >> 16);
    return (float)l / 255.0f;
}

__global__ __launch_bounds__(256) void ke_ssim_tiles(const SsimArgs a) {
    __shared__ float s_x[kHH][kPitch], s_y[kHH][kPitch];
    __shared__ float s_m[5][kTH][kPitch];
    __shared__ double s_red[256];
    const int tid = threadIdx.x;
    const int tiles = a.tiles_x * a.tiles_y;
    const int64_t pair = blockIdx.x / tiles;
    const int tile = blockIdx.x % tiles;
    const int ty = tile / a.tiles_x, tx = tile % a.tiles_x;
    const int y0 = ty * kTH, x0 = tx * kTW;   // halo origin == first interior pixel - 3
    const size_t img_bytes = (size_t)a.w * a.h * a.ch;
    const uint8_t *A = a.images + (size_t)a.pa[pair] * img_bytes;
    const uint8_t *B = a.images + (size_t)a.pb[pair] * img_bytes;
    for (int e = tid; e < kHH * kHW; e += 256) {
        const int r = e / kHW, c = e % kHW;
        const int y = y0 + r, x = x0 + c;
        float fx = 0.f, fy = 0.f;
        if (y < a.h && x < a.w) {
            const size_t off = ((size_t)y * a.w + x) * a.ch;
            fx = unit_luma(A + off, a.ch);
            fy = unit_luma(B + off, a.ch);
        }
        s_x[r][c] = fx;
        s_y[r][c] = fy;
    }
    __syncthreads();
    // vertical pass (scipy axis 0): rows r..r+6 of the halo -> interior row r
    for (int e = tid; e < kTH * kHW; e += 256) {
        const int r = e / kHW, c = e % kHW;
        double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
#pragma unroll
        for (int d = 0; d < 7; ++d) {
            const float vx = s_x[r + d][c], vy = s_y[r + d][c];
            sx += (double)vx;
            sy += (double)vy;
            sxx += (double)(vx * vx);
            syy += (double)(vy * vy);
            sxy += (double)(vx * vy);
        }
        s_m[0][r][c] = (float)(sx / 7.0);
        s_m[1][r][c] = (float)(sy / 7.0);
        s_m[2][r][c] = (float)(sxx / 7.0);
        s_m[3][r][c] = (float)(syy / 7.0);
        s_m[4][r][c] = (float)(sxy / 7.0);
    }
    __syncthreads();
    const float cov_norm = (float)(49.0 / 48.0);
    const float C1 = (float)(0.01 * 0.01), C2 = (float)(0.03 * 0.03);
    double local = 0.0;
    for (int e = tid; e < kTH * kTW; e += 256) {
        const int r = e / kTW, c = e % kTW;
        if (y0 + r + 3 >= a.h - 3 || x0 + c + 3 >= a.w - 3) continue;   // outside the cropped interior
        double s[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            double acc = 0;
#pragma unroll
            for (int d = 0; d < 7; ++d) acc += (double)s_m[q][r][c + d];
            s[q] = acc / 7.0;
        }
        const float ux = (float)s[0], uy = (float)s[1], uxx = (float)s[2], uyy = (float)s[3], uxy = (float)s[4];
        const float mxx = ux * ux, myy = uy * uy, mxy = ux * uy;
        const float vx = cov_norm * (uxx - mxx);
        const float vy = cov_norm * (uyy - myy);
        const float vxy = cov_norm * (uxy - mxy);
        const float A1 = 2.0f * mxy + C1, A2 = 2.0f * vxy + C2;
        const float B1 = (mxx + myy) + C1, B2 = (vx + vy) + C2;
        const float S = (A1 * A2) / (B1 * B2);
        local += (double)S;
    }
    s_red[tid] = local;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (tid < k) s_red[tid] += s_red[tid + k];
        __syncthreads();
    }
    if (tid == 0) a.partial[(size_t)pair * tiles + tile] = s_red[0];
}

__global__ void ke_ssim_finish(const double *__restrict__ partial, int tiles, int64_t n_pairs, double denom,
                               double *__restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    double s = 0.0;
    for (int t = 0; t < tiles; ++t) s += partial[(size_t)p * tiles + t];
    out[p] = s / denom;
}

__global__ void ke_fill_nan(double *out, int64_t n) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) out[p] = __longlong_as_double(0x7ff8000000000000LL);
}

}  // namespace

int ke_launch_ssim(ke_ctx *ctx, const uint8_t *d_images, int w, int h, int channels, const int64_t *d_pa,
                   const int64_t *d_pb, int64_t n_pairs, double *d_out) {
    if (w < 7 || h < 7) {  // skimage raises for images smaller than the window
        hipLaunchKernelGGL(ke_fill_nan, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, ctx->stream, d_out, n_pairs);
        KE_HIP(ctx, hipGetLastError());
        return KE_OK;
    }
    SsimArgs a;
    a.images = d_images; a.pa = d_pa; a.pb = d_pb;
    a.w = w; a.h = h; a.ch = channels;
    a.tiles_x = (w - 6 + kTW - 1) / kTW;
    a.tiles_y = (h - 6 + kTH - 1) / kTH;
    const int tiles = a.tiles_x * a.tiles_y;
    if ((int64_t)tiles * n_pairs > 0x7fffffffLL) return ke_fail(ctx, KE_EUNSUPPORTED, "too many SSIM tiles for one launch");
    void *part;
    KE_TRY(ke_reserve(ctx, KE_BUF_SCAN_AUX, (size_t)n_pairs * tiles * sizeof(double), &part));
    a.partial = (double *)part;
    hipLaunchKernelGGL(ke_ssim_tiles, dim3((unsigned)(n_pairs * tiles)), dim3(256), 0, ctx->stream, a);
    KE_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(ke_ssim_finish, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const double *)part, tiles, n_pairs, (double)(w - 6) * (double)(h - 6), d_out);
    KE_HIP(ctx, hipGetLastError());
    return KE_OK;
}
