// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands on gfx950: operand lane layout and per-lane scale semantics.
// hipcc --offload-arch=gfx950 -O2 tools/probes/mx_probe.hip -o /tmp/mx_probe && /tmp/mx_probe
#include <hip/hip_runtime.h>
#include <hip/hip_fp8.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;

// A: [32 rows][64 k] bytes (e4m3), B: [32 cols][64 k] bytes, sa / sb: [32][2] E8M0 scale bytes per (row, k block)
__global__ void probe(const unsigned char* A, const unsigned char* B, const unsigned char* sa, const unsigned char* sb, float* D) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    i32x8 a, b;
    const int* pa = reinterpret_cast<const int*>(A + r * 64 + h * 32);
    const int* pb = reinterpret_cast<const int*>(B + r * 64 + h * 32);
    for (int i = 0; i < 8; ++i) { a[i] = pa[i]; b[i] = pb[i]; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    const int scale_a = sa[r * 2 + h], scale_b = sb[r * 2 + h];
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        D[row * 32 + r] = c[reg];
    }
}

static float e4m3_to_f(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7);
    return s ? -f : f;
}

int main() {
    unsigned char hA[32 * 64], hB[32 * 64], hsa[64], hsb[64];
    srand(1);
    for (int i = 0; i < 32 * 64; ++i) { hA[i] = (unsigned char)(rand() & 0x7f) | ((rand() & 1) << 7); hB[i] = (unsigned char)(rand() & 0x7f) | ((rand() & 1) << 7); }
    for (int i = 0; i < 32 * 64; ++i) { if ((hA[i] & 0x7f) == 0x7f) hA[i] &= 0xfe; if ((hB[i] & 0x7f) == 0x7f) hB[i] &= 0xfe; }   // no NaN
    for (int i = 0; i < 64; ++i) { hsa[i] = 127 - (rand() % 5); hsb[i] = 127 + (rand() % 4) - 6; }
    unsigned char *dA, *dB, *dsa, *dsb; float* dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dD, 32 * 32 * 4);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa, 64, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 64, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dsa, dsb, dD);
    float hD[32 * 32];
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    double worst = 0, mag = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double s = 0;
            for (int kb = 0; kb < 2; ++kb) {
                double p = 0;
                for (int k = 0; k < 32; ++k) p += (double)e4m3_to_f(hA[i * 64 + kb * 32 + k]) * e4m3_to_f(hB[j * 64 + kb * 32 + k]);
                s += p * ldexp(1.0, (int)hsa[i * 2 + kb] - 127) * ldexp(1.0, (int)hsb[j * 2 + kb] - 127);
            }
            worst = fmax(worst, fabs(s - hD[i * 32 + j])); mag = fmax(mag, fabs(s));
        }
    printf("max |D - ref| = %g  (max |ref| = %g)  -> %s\n", worst, mag, worst <= 1e-5 * mag ? "layout + per-lane scales CONFIRMED" : "MISMATCH");
    return 0;
}
