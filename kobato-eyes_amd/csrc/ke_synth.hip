// ke_synth.hip -- counter-based synthetic corpus generated directly in HBM (BASELINE configs;
// DESIGN.md "Synthetic data").  Integer only, so the CPU oracle reproduces it bit for bit:
// 16x16 blocky low-frequency cells per base image, +-4 per-pixel noise, variants (every
// index = 9 mod 10 from 19 on) copy an earlier base with a brightness shift and 1/32 of the
// cells replaced; from index 1000 on half of the variants are "low-noise": they reuse their base's
// per-pixel noise and replace 1..4 of 32 cells, so that their SSIM against the base spreads around the
// reference's 0.95 threshold (BASELINE configs[3]).
#include "ke_internal.h"

namespace {

__device__ __host__ inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
constexpr int64_t kLowNoiseFrom = 1000;   // the first 1000 images (BASELINE configs[0]) keep the round-1 corpus
constexpr uint64_t P1 = 0xD6E8FEB86659FD93ull, P2 = 0xCA5A826395121157ull, P3 = 0x9E6C63D0676A9A99ull;

// indices: nullable list of corpus positions to generate (then image k of the launch is indices[k]), else first + k
__global__ __launch_bounds__(256) void ke_synth_rgb_kernel(uint64_t seed, int64_t first, const int64_t *__restrict__ indices,
                                                            int w, int h, uint8_t *__restrict__ out, int blocks_per_image) {
    const int64_t k = blockIdx.x / blocks_per_image;
    const int blk = blockIdx.x % blocks_per_image;
    const int64_t index = indices ? indices[k] : first + k;
    const uint64_t r = splitmix64(seed ^ (0x51ED270B0E3A6F5Dull + (uint64_t)index * P1));
    const bool variant = index >= 10 && index % 10 == 9;
    int64_t base = index, noise_index = index;
    int delta = 0, cell_q = 1;
    if (variant) {
        base = (int64_t)((r >> 8) % (uint64_t)index);
        if (base % 10 == 9) base -= 1;
        delta = (int)((r >> 44) % 7) - 3;
        if (index >= kLowNoiseFrom && ((r >> 52) & 1)) {     // low-noise class: the base's own per-pixel noise, 1..4 of 32 cells replaced
            noise_index = base;
            cell_q = 1 + (int)((r >> 53) & 3);
        }
    }
    uint8_t *img = out + (size_t)k * w * h * 3;
    const int npix = w * h;
    for (int p = blk * 256 + threadIdx.x; p < npix; p += blocks_per_image * 256) {
        const int y = p / w, x = p % w;
        const int cy = (int)(((int64_t)y * 16) / h), cx = (int)(((int64_t)x * 16) / w);
        uint64_t cell = splitmix64((seed + 1) ^ ((uint64_t)base * P1 + (uint64_t)cy * P2 + (uint64_t)cx * P3));
        if (variant) {
            const uint64_t m = splitmix64((seed + 2) ^ ((uint64_t)index * P1 + (uint64_t)cy * P2 + (uint64_t)cx * P3));
            if ((int)(m & 31) < cell_q) cell = m >> 8;
        }
        const uint64_t u = splitmix64(seed ^ ((uint64_t)noise_index * P1 + (uint64_t)y * P2 + (uint64_t)x * P3));
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int v = (int)((cell >> (8 * c)) & 0xFF) + (int)((u >> (20 * c)) & 7) - 4 + delta;
            img[(size_t)p * 3 + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

__global__ void ke_synth_hashes_kernel(uint64_t seed, int64_t n, uint64_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t n_plain = n - n / 10;
    if (i < n_plain) {
        out[i] = splitmix64(seed + (uint64_t)i);
        return;
    }
    const uint64_t r = splitmix64(seed ^ (0x7F4A7C159E3779B9ull + (uint64_t)i * P2));
    const uint64_t v = splitmix64(seed + (r % (uint64_t)n_plain));
    int k = (int)((r >> 40) % 11);
    uint64_t flipped = 0, s = r;
    while (k > 0) {
        s = splitmix64(s);
        const uint64_t bit = 1ull << (s & 63);
        if (flipped & bit) continue;
        flipped |= bit;
        --k;
    }
    out[i] = v ^ flipped;
}

}  // namespace

int ke_launch_synth_rgb(ke_ctx *ctx, uint64_t seed, int64_t first, const int64_t *d_indices, int64_t n, int w, int h, uint8_t *d_out) {
    const int64_t npix = (int64_t)w * h;
    int bpi = (int)std::min<int64_t>((npix + 1023) / 1024, 64);
    if (bpi < 1) bpi = 1;
    // keep each launch below 2^31 blocks
    const int64_t per_launch = 0x7fffffffLL / bpi;
    for (int64_t f = 0; f < n; f += per_launch) {
        const int64_t m = std::min(per_launch, n - f);
        hipLaunchKernelGGL(ke_synth_rgb_kernel, dim3((unsigned)(m * bpi)), dim3(256), 0, ctx->stream, seed, first + f,
                           d_indices ? d_indices + f : nullptr, w, h, d_out + (size_t)f * npix * 3, bpi);
        KE_HIP(ctx, hipGetLastError());
    }
    return KE_OK;
}

int ke_launch_synth_hashes(ke_ctx *ctx, uint64_t seed, int64_t n, uint64_t *d_out) {
    hipLaunchKernelGGL(ke_synth_hashes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, seed, n, d_out);
    KE_HIP(ctx, hipGetLastError());
    return KE_OK;
}
