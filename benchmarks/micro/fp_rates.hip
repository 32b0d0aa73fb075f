// Issue cost of the floating-point ops the SSIM kernel leans on (gfx950): fp64 add/fma, f32<->f64 converts,
// packed fp32.  Same harness as valu_rates.hip: 8 waves/SIMD, 16 independent chains per lane.
// hipcc --offload-arch=gfx950 -O3 -o fp_rates.bin fp_rates.hip && ./fp_rates.bin
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
#define CH 16
template <int OP>
__global__ __launch_bounds__(256) void rate(int iters, const float *in, float *out, unsigned long long *clk) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double d[CH];
    float f[CH];
    f2 p[CH];
    const float b = in[threadIdx.x & 63], c = in[64 + (threadIdx.x & 63)];
    const double bd = b, cd = c;
    const f2 bp = {b, c}, cp = {c, b};
#pragma unroll
    for (int k = 0; k < CH; ++k) { f[k] = in[k] + threadIdx.x; d[k] = f[k]; p[k] = f2{f[k], f[k] + 1.0f}; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            if (OP == 0) d[k] = d[k] + bd;                                   // v_add_f64
            if (OP == 1) d[k] = __builtin_fma(d[k], bd, cd);                 // v_fma_f64
            if (OP == 2) d[k] = d[k] * bd;                                   // v_mul_f64
            if (OP == 3) { f[k] = (float)((double)f[k]); asm volatile("" : "+v"(f[k])); }   // cvt pair (kept by the asm barrier)
            if (OP == 4) f[k] = __builtin_fmaf(f[k], b, c);                  // v_fma_f32
            if (OP == 5) p[k] = p[k] * bp + cp;                              // v_pk_fma_f32 (contracted) or pk_mul + pk_add
            if (OP == 6) p[k] = p[k] * bp;                                   // v_pk_mul_f32
            if (OP == 7) p[k] = p[k] + bp;                                   // v_pk_add_f32
            if (OP == 8) f[k] = f[k] / b;                                    // correctly rounded fp32 division
            if (OP == 9) { unsigned u = __float_as_uint(f[k]); f[k] = (float)((u >> 8) & 0xFFu) + c; }  // cvt_f32_ubyte1 + add
        }
    }
    float s = 0;
#pragma unroll
    for (int k = 0; k < CH; ++k) s += f[k] + (float)d[k] + p[k].x + p[k].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - t0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int OP>
void run(const char *name, float *d_in, float *d_out, double per_iter) {
    const int iters = 2048, blocks = 256 * 8;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    static unsigned long long *d_clk = nullptr;
    if (!d_clk) (void)hipMalloc(&d_clk, 16);
    hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(256), 0, 0, 64, d_in, d_out, (unsigned long long *)nullptr);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(256), 0, 0, iters, d_in, d_out, d_clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long clk[2];
    (void)hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost);
    const double mhz = (double)clk[0] / (double)clk[1] * 100.0;
    const double groups = (double)blocks * 4 * iters * CH;              // source-level ops per SIMD-set
    const double per_simd_per_us = groups / 1024.0 / (ms * 1e3);
    printf("%-34s %8.3f ms  clock %5.0f MHz  => %6.2f shader cycles per wave64 source op (%s)\n", name, ms, mhz, mhz / per_simd_per_us,
           per_iter > 1 ? "several instructions" : "one instruction");
}

int main() {
    float *d_in, *d_out;
    float h[256];
    for (int i = 0; i < 256; ++i) h[i] = 1.0f + (i % 7) * 1e-3f;
    (void)hipMalloc(&d_in, 1024); (void)hipMalloc(&d_out, 256 * 8 * 256 * 4);
    (void)hipMemcpy(d_in, h, 1024, hipMemcpyHostToDevice);
    run<0>("v_add_f64", d_in, d_out, 1);
    run<1>("v_fma_f64", d_in, d_out, 1);
    run<2>("v_mul_f64", d_in, d_out, 1);
    run<3>("cvt f32->f64->f32", d_in, d_out, 2);
    run<4>("v_fma_f32", d_in, d_out, 1);
    run<5>("pk fp32 mul+add (2 lanes)", d_in, d_out, 2);
    run<6>("v_pk_mul_f32 (2 lanes)", d_in, d_out, 1);
    run<7>("v_pk_add_f32 (2 lanes)", d_in, d_out, 1);
    run<8>("fp32 division (IEEE)", d_in, d_out, 10);
    run<9>("cvt_f32_ubyte + add", d_in, d_out, 2);
    return 0;
}
