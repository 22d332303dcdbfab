// Does v_mfma_f32_32x32x16_f16 keep fp16 subnormal INPUTS (needed by the hi/lo split of the precise mode)?
// a = subnormal fp16 (2^-20), b = 1.0: a flushed input gives 0, a kept one 16 * 2^-20 per output element.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__global__ void k(float* out, float av) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)av; b[i] = (_Float16)1.0f; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float vals[3] = {9.5367431640625e-07f /* 2^-20 */, 5.9604644775390625e-08f /* 2^-24, smallest */, 6.103515625e-05f /* 2^-14 normal */};
    for (int i = 0; i < 3; ++i) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, vals[i]);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%g  sum=%g  expected=%g  %s\n", vals[i], h, 16.0 * vals[i], h == 16.0f * vals[i] ? "KEPT" : "FLUSHED/INEXACT");
    }
    return 0;
}
