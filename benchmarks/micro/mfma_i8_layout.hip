// Checks the operand / result lane maps of v_mfma_i32_16x16x64_i8 on gfx950 with exact integer data
// (the horizontal resampling taps of the fused hash kernel run on it).
// hipcc --offload-arch=gfx950 -O3 -o mfma_i8_layout.bin mfma_i8_layout.hip && ./mfma_i8_layout.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void probe(const int8_t *A, const int8_t *B, int *D) {   // A[16][64] row-major, B[64][16] row-major
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    v4i a, b, c = {0, 0, 0, 0};
    int8_t ab[16], bb[16];
    for (int j = 0; j < 16; ++j) {
        ab[j] = A[r * 64 + 16 * g + j];
        bb[j] = B[(16 * g + j) * 16 + r];
    }
    memcpy(&a, ab, 16);
    memcpy(&b, bb, 16);
    c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = c[i];
}

int main() {
    int8_t hA[16 * 64], hB[64 * 16];
    int hD[256], ref[256];
    uint32_t s = 12345;
    for (auto &v : hA) { s = s * 1664525u + 1013904223u; v = (int8_t)(s >> 24); }
    for (auto &v : hB) { s = s * 1664525u + 1013904223u; v = (int8_t)(s >> 24); }
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            int acc = 0;
            for (int k = 0; k < 64; ++k) acc += (int)hA[i * 64 + k] * (int)hB[k * 16 + j];
            ref[i * 16 + j] = acc;
        }
    int8_t *dA, *dB;
    int *dD;
    (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dD, sizeof hD);
    (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
    printf("v_mfma_i32_16x16x64_i8 layout A[l&15][16(l>>4)+j] B[16(l>>4)+j][l&15] D[4(l>>4)+i][l&15]: %s (%d mismatches)\n",
           bad ? "WRONG" : "ok", bad);
    return bad != 0;
}
