// Which operand bytes does each lane's scale of v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3) apply to?  One-hot experiments.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;

__global__ void probe(const unsigned char* A, const unsigned char* B, const int* sa, const int* sb, float* D, int opsel_a) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    i32x8 a, b;
    const int* pa = reinterpret_cast<const int*>(A + lane * 32);      // lane-major: byte t of lane l at A[l*32 + t]
    const int* pb = reinterpret_cast<const int*>(B + lane * 32);
    for (int i = 0; i < 8; ++i) { a[i] = pa[i]; b[i] = pb[i]; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    if (opsel_a == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa[lane], 0, sb[lane]);
    else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 1, sa[lane], 0, sb[lane]);
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        D[row * 32 + r] = c[reg];
    }
}

int main() {
    unsigned char hA[64 * 32], hB[64 * 32]; int hsa[64], hsb[64]; float hD[32 * 32];
    unsigned char *dA, *dB; int *dsa, *dsb; float* dD;
    (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dsa, 256); (void)hipMalloc(&dsb, 256); (void)hipMalloc(&dD, sizeof hD);
    const unsigned char ONE = 0x38;      // e4m3 1.0
    auto run = [&](int opsel) {
        (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        (void)hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(dA, dB, dsa, dsb, dD, opsel);
        (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    };
    // 1. pairing: A one-hot at (lane 3 + 32 ha, byte ba), B one-hot at (lane 7 + 32 hb, byte bb): D[3][7] = 1 iff same k
    printf("pairing (A lane half ha, byte ba) <-> (B lane half hb, byte bb) with D[3][7] != 0:\n");
    for (int i = 0; i < 64; ++i) hsa[i] = hsb[i] = 127;
    for (int ha = 0; ha < 2; ++ha)
        for (int ba = 0; ba < 32; ba += 5) {
            for (int hb = 0; hb < 2; ++hb)
                for (int bb = 0; bb < 32; ++bb) {
                    memset(hA, 0, sizeof hA); memset(hB, 0, sizeof hB);
                    hA[(3 + 32 * ha) * 32 + ba] = ONE; hB[(7 + 32 * hb) * 32 + bb] = ONE;
                    run(0);
                    if (hD[3 * 32 + 7] != 0.f) printf("  A(h%d,b%2d) <-> B(h%d,b%2d) = %g\n", ha, ba, hb, bb, hD[3 * 32 + 7]);
                }
        }
    // 2. scale lanes: B all ones; A one-hot at (lane 3 + 32 ha, byte ba); scale_a = 2^1 in lane ls only (all four bytes of the VGPR)
    printf("which scale_a lane doubles the contribution of A(lane 3 + 32 ha, byte ba)  [opsel 0; scale VGPR = 0x80808080 in lane ls]:\n");
    for (int ha = 0; ha < 2; ++ha)
        for (int ba = 0; ba < 32; ++ba) {
            memset(hA, 0, sizeof hA); memset(hB, ONE, sizeof hB);
            hA[(3 + 32 * ha) * 32 + ba] = ONE;
            char line[256]; int n = 0;
            for (int ls = 0; ls < 64; ++ls) {
                for (int i = 0; i < 64; ++i) { hsa[i] = 0x7f7f7f7f; hsb[i] = 0x7f7f7f7f; }
                hsa[ls] = (int)0x80808080;
                run(0);
                if (hD[3 * 32 + 7] != 1.f) n += snprintf(line + n, sizeof line - n, " ls=%d(x%g)", ls, hD[3 * 32 + 7]);
            }
            printf("  A(h%d,b%2d):%s\n", ha, ba, n ? line : " none");
        }
    // 3. which BYTE of the scale VGPR is used (opsel 0 / 1): lane 3 scale = 0x7f7f7f80 | ...
    for (int opsel = 0; opsel < 2; ++opsel)
        for (int byte = 0; byte < 4; ++byte) {
            memset(hA, 0, sizeof hA); memset(hB, ONE, sizeof hB);
            hA[3 * 32 + 0] = ONE;
            for (int i = 0; i < 64; ++i) { hsa[i] = 0x7f7f7f7f; hsb[i] = 0x7f7f7f7f; }
            hsa[3] = 0x7f7f7f7f + (1 << (8 * byte));
            run(opsel);
            printf("opsel_a %d, scale byte %d raised in lane 3 -> D[3][7] = %g\n", opsel, byte, hD[3 * 32 + 7]);
        }
    return 0;
}
