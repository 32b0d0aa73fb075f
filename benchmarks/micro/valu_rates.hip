// Instruction-rate microbenchmark for the integer VALU ops the kernels lean on (gfx950).
// hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define OPS_PER_ITER 16
template <int OP>
__global__ __launch_bounds__(256) void rate(int iters, const int *in, int *out) {
    int a[OPS_PER_ITER];
    const int b = in[threadIdx.x & 63], c = in[64 + (threadIdx.x & 63)];
#pragma unroll
    for (int k = 0; k < OPS_PER_ITER; ++k) a[k] = in[k] + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < OPS_PER_ITER; ++k) {
            if (OP == 0) a[k] = __builtin_amdgcn_sdot4(a[k], b, a[k], false);
            if (OP == 1) a[k] = (int)__builtin_amdgcn_udot4((unsigned)a[k], (unsigned)b, (unsigned)a[k], false);
            if (OP == 2) a[k] = __builtin_popcount((unsigned)a[k]) + c;          // v_bcnt_u32_b32 with accumulate
            if (OP == 3) a[k] = a[k] ^ b;
            if (OP == 4) a[k] = __builtin_amdgcn_alignbyte(a[k], b, 3);
            if (OP == 5) a[k] = (a[k] & 0xffffff) * (b & 0xffffff) + c;          // v_mad_u32_u24
            if (OP == 6) a[k] = min(a[k], b);
            if (OP == 7) a[k] = a[k] * b;                                          // v_mul_lo_u32
            if (OP == 8) a[k] = __builtin_amdgcn_sdot4(a[k], b, c, false) ;       // dot4 non-accumulating chain dependency via src
        }
    }
    int s = 0;
#pragma unroll
    for (int k = 0; k < OPS_PER_ITER; ++k) s += a[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
double run(const char *name, int *d_in, int *d_out) {
    const int iters = 4096, blocks = 256 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(256), 0, 0, 64, d_in, d_out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(256), 0, 0, iters, d_in, d_out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4 * iters * OPS_PER_ITER;
    const double per_simd_per_us = wave_instr / (1024.0) / (ms * 1e3);   // wave-instructions per SIMD per microsecond
    printf("%-28s %8.3f ms  %7.1f wave-instr/SIMD/us  => %5.2f cycles per wave-instr at 2.4 GHz\n", name, ms, per_simd_per_us,
           2400.0 / per_simd_per_us);
    return ms;
}

int main() {
    int *d_in, *d_out;
    std::vector<int> h(256);
    for (int i = 0; i < 256; ++i) h[i] = i * 2654435761u;
    hipMalloc(&d_in, 1024); hipMalloc(&d_out, 256 * 8 * 256 * 4);
    hipMemcpy(d_in, h.data(), 1024, hipMemcpyHostToDevice);
    run<0>("v_dot4c_i32_i8 (acc chain)", d_in, d_out);
    run<8>("v_dot4_i32_i8 (3-src)", d_in, d_out);
    run<1>("v_dot4_u32_u8", d_in, d_out);
    run<2>("v_bcnt_u32_b32", d_in, d_out);
    run<3>("v_xor_b32", d_in, d_out);
    run<4>("v_alignbyte_b32", d_in, d_out);
    run<5>("v_mad_u32_u24", d_in, d_out);
    run<6>("v_min_i32", d_in, d_out);
    run<7>("v_mul_lo_u32", d_in, d_out);
    return 0;
}
