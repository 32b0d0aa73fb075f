// Instruction-rate microbenchmark for the integer VALU ops the kernels lean on (gfx950).
// hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define OPS_PER_ITER 16
template <int OP>
__global__ __launch_bounds__(256) void rate(int iters, const int *in, int *out, unsigned long long *clk) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
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
            if (OP == 9) a[k] = __float_as_int(fmaf(__int_as_float(a[k]), 1.0000001f, 0.5f));   // v_fma_f32 reference
            if (OP == 10) a[k] = a[k] + b;
            if (OP == 11) a[k] = (a[k] << 3) + b;                                  // v_lshl_add_u32
            if (OP == 12) a[k] = __builtin_amdgcn_perm(a[k], b, 0x06020400);
        }
    }
    int s = 0;
#pragma unroll
    for (int k = 0; k < OPS_PER_ITER; ++k) s += a[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) {   // shader cycles and 100 MHz ticks spent by one wave
        clk[0] = __builtin_amdgcn_s_memtime() - t0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int OP>
double run(const char *name, int *d_in, int *d_out) {
    const int iters = 4096, blocks = 256 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    static unsigned long long *d_clk = nullptr;
    if (!d_clk) hipMalloc(&d_clk, 16);
    hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(256), 0, 0, 64, d_in, d_out, (unsigned long long *)nullptr);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(256), 0, 0, iters, d_in, d_out, d_clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4 * iters * OPS_PER_ITER;
    const double per_simd_per_us = wave_instr / (1024.0) / (ms * 1e3);   // wave-instructions per SIMD per microsecond
    unsigned long long clk[2];
    hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost);
    const double mhz = (double)clk[0] / (double)clk[1] * 100.0;   // shader clock held during the loop
    printf("%-28s %8.3f ms  %7.1f wave-instr/SIMD/us  clock %6.0f MHz  => %5.2f shader cycles per wave-instr per SIMD\n", name, ms,
           per_simd_per_us, mhz, mhz / per_simd_per_us);
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
    run<9>("v_fma_f32", d_in, d_out);
    run<10>("v_add_u32", d_in, d_out);
    run<11>("v_lshl_add_u32", d_in, d_out);
    run<12>("v_perm_b32", d_in, d_out);
    return 0;
}
