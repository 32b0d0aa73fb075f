// Streaming-read ceiling of one MI355X: what a kernel that only reads (non-temporal dwordx4, one
// accumulate per dword, nothing written but one partial per workgroup) reaches on the same footprint the
// fused hash kernel sweeps (78.6 GB).  The number the hash kernel's achieved GB/s is compared with in DESIGN.md.
// hipcc --offload-arch=gfx950 -O3 -o hbm_read.bin hbm_read.hip && ./hbm_read.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void sweep(const u32x4 *__restrict__ src, size_t per_wg, uint32_t *__restrict__ out) {
    const u32x4 *p = src + (size_t)blockIdx.x * per_wg + threadIdx.x;
    uint32_t acc = 0;
    for (size_t i = 0; i < per_wg; i += 256 * UNROLL) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * 256) : p[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    acc += __shfl_xor(acc, 1);
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

template <int UNROLL, bool NT>
void run(const char *name, const u32x4 *buf, size_t bytes, int wgs, uint32_t *out) {
    const size_t per_wg = bytes / 16 / wgs / (256 * UNROLL) * (256 * UNROLL);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((sweep<UNROLL, NT>), dim3(wgs), dim3(256), 0, 0, buf, per_wg, out);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((sweep<UNROLL, NT>), dim3(wgs), dim3(256), 0, 0, buf, per_wg, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-28s wgs=%6d  %.3f ms  %.3f TB/s\n", name, wgs, best, (double)per_wg * 16 * wgs / (best * 1e-3) / 1e12);
}

int main() {
    const size_t bytes = 78643200000ull;   // 100 000 x 512 x 512 x 3
    u32x4 *buf;
    uint32_t *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 1 << 22) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    for (int wgs : {2048, 8192, 32768, 100000}) {
        run<4, true>("nt x4 unroll 4", buf, bytes, wgs, out);
        run<8, true>("nt x4 unroll 8", buf, bytes, wgs, out);
        run<4, false>("plain x4 unroll 4", buf, bytes, wgs, out);
    }
    hipFree(buf); hipFree(out);
    return 0;
}
