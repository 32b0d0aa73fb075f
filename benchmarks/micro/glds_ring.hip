#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// each wave: DMA rows of 64*12 bytes into its own LDS ring (depth 4), read them back, sum per lane
template <int D>
__global__ void k(const uint8_t *src, uint32_t *out, int rows, uint32_t row_bytes) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[4][D + 1][64 * 4];   // a dwordx3 DMA places its lanes 16 bytes apart
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint8_t *base = src + (size_t)(blockIdx.x * 4 + wave) * rows * row_bytes;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)&lds[wave][0][0];
    uint32_t vo = lane * 12;
    auto issue = [&](int r) {
        const uint8_t *rp = base + (size_t)((uint32_t)min(r, rows - 1) * row_bytes);
        const uint32_t dst = lds0 + (uint32_t)(r % (D + 1)) * 1024u;
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(vo), "s"(dst), "s"(rp) : "memory");
    };
    for (int r = 0; r < D; ++r) issue(r);
    uint32_t acc0 = 0, acc1 = 0, acc2 = 0;
    for (int r = 0; r < rows; ++r) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D - 1) : "memory");
        const uint32_t *p = &lds[wave][r % (D + 1)][lane * 4];
        const uint32_t a = p[0], b = p[1], c = p[2];
        issue(r + D);
        acc0 += a * (uint32_t)(r + 1); acc1 ^= b + r; acc2 += c;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t *o = out + ((size_t)(blockIdx.x * 4 + wave) * 64 + lane) * 3;
    o[0] = acc0; o[1] = acc1; o[2] = acc2;
}
int main() {
    const int waves = 4096, rows = 200; const uint32_t rb = 768;
    std::vector<uint8_t> h((size_t)waves * rows * rb);
    uint32_t s = 12345; for (auto &v : h) { s = s * 1664525u + 1013904223u; v = s >> 24; }
    uint8_t *d; uint32_t *o;
    hipMalloc(&d, h.size()); hipMalloc(&o, (size_t)waves * 64 * 3 * 4);
    hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k<3>), waves / 4, 256, 0, 0, d, o, rows, rb);
    std::vector<uint32_t> r((size_t)waves * 64 * 3);
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (int w = 0; w < waves; ++w) for (int l = 0; l < 64; ++l) {
        uint32_t a0 = 0, a1 = 0, a2 = 0;
        for (int y = 0; y < rows; ++y) {
            const uint32_t *p = (const uint32_t *)(h.data() + ((size_t)w * rows + y) * rb + l * 12);
            a0 += p[0] * (uint32_t)(y + 1); a1 ^= p[1] + y; a2 += p[2];
        }
        const uint32_t *g = &r[((size_t)w * 64 + l) * 3];
        if (g[0] != a0 || g[1] != a1 || g[2] != a2) ++bad;
    }
    printf("glds ring: bad=%zu of %d\n", bad, waves * 64);
    return bad != 0;
}
