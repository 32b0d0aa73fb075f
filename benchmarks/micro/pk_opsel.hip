#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));
// {a.lo - a.hi, a.lo + a.hi}
__device__ __forceinline__ f2 diff_sum(f2 a) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(a));
    return r;
}
// {a.lo + a.hi, a.lo - a.hi}
__device__ __forceinline__ f2 sum_diff(f2 a) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(a));
    return r;
}
__global__ void k(const f2 *in, f2 *out) {
    int i = threadIdx.x;
    out[2 * i] = diff_sum(in[i]);
    out[2 * i + 1] = sum_diff(in[i]);
}
int main() {
    f2 *in, *out;
    hipMallocManaged(&in, 64 * sizeof(f2)); hipMallocManaged(&out, 128 * sizeof(f2));
    for (int i = 0; i < 64; ++i) in[i] = f2{(float)(i * 3 + 1), (float)(i * 7 + 2)};
    hipLaunchKernelGGL(k, 1, 64, 0, 0, in, out);
    hipDeviceSynchronize();
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        float a = in[i][0], b = in[i][1];
        if (out[2*i][0] != a - b || out[2*i][1] != a + b || out[2*i+1][0] != a + b || out[2*i+1][1] != a - b) ++bad;
    }
    printf("bad=%d  sample: in=(%g,%g) diff_sum=(%g,%g) sum_diff=(%g,%g)\n", bad, in[5][0], in[5][1], out[10][0], out[10][1], out[11][0], out[11][1]);
    return bad != 0;
}
