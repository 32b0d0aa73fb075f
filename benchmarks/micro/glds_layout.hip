// Where do the bytes of one global_load_lds_{dword,dwordx3,dwordx4} land?  One wave, source dword j of lane l = l*16 + j.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int DW>
__global__ void k(const uint32_t *src, uint32_t *out) {
    __shared__ uint32_t lds[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) lds[i] = 0xDEAD0000u + i;
    __syncthreads();
    const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds + 64;   // +64 bytes
    uint32_t vo = lane * 64;          // lane's source: 16 dwords apart
    uint32_t keep;
    if (DW == 1)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(vo), "s"(dst), "s"(src) : "memory");
    else if (DW == 3)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, %3\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(vo), "s"(dst), "s"(src) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(vo), "s"(dst), "s"(src) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
    uint32_t *src, *out;
    hipMallocManaged(&src, 64 * 16 * 4); hipMallocManaged(&out, 1024 * 4);
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 16; ++j) src[l * 16 + j] = l * 16 + j;
    for (int dw : {1, 3, 4}) {
        if (dw == 1) hipLaunchKernelGGL(k<1>, 1, 64, 0, 0, src, out);
        if (dw == 3) hipLaunchKernelGGL(k<3>, 1, 64, 0, 0, src, out);
        if (dw == 4) hipLaunchKernelGGL(k<4>, 1, 64, 0, 0, src, out);
        (void)hipDeviceSynchronize();
        printf("DW=%d: LDS dwords 12..60 (dst starts at dword 16):\n", dw);
        for (int i = 12; i < 60; ++i) {
            if ((out[i] >> 16) == 0xDEAD) printf(" [%d]=--", i);
            else printf(" [%d]=L%u.%u", i, out[i] / 16, out[i] % 16);
        }
        int written = 0, last = 0;
        for (int i = 0; i < 1024; ++i) if ((out[i] >> 16) != 0xDEAD) { ++written; last = i; }
        printf("\n  written dwords: %d, last index %d\n", written, last);
    }
    return 0;
}
