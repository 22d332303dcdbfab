// Micro-benchmark: sustained v_mfma_f32_32x32x16_f16 rate on this device (random data),
//   (a) operands resident in registers, (b) operands re-read from LDS with ds_read_b128 (1 KiB per MFMA per
//   wave, the ratio of the conv kernels' 64x64 wave tile), software pipelined one step ahead.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(16))) float f16v;

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_reg(const h8* in, float* out, int iters) {
    h8 a0 = in[threadIdx.x], a1 = in[threadIdx.x + 256], b0 = in[threadIdx.x + 512], b1 = in[threadIdx.x + 768];
    f16v c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
    for (int i = 0; i < iters; ++i) {
        c00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c11, 0, 0, 0);
    }
    f16v s = c00 + c01 + c10 + c11;
    float t = 0; for (int r = 0; r < 16; ++r) t += s[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_lds(const h8* in, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[128 * 72 * 2];   // A and B tiles, 144-B rows
    for (int i = threadIdx.x; i < 128 * 72 * 2 / 8; i += blockDim.x) ((h8*)lds)[i] = in[i % 1024];
    __syncthreads();
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5, w = threadIdx.x >> 6;
    const unsigned short* A = lds + (w % 2) * 64 * 72 + l31 * 72 + h * 8;
    const unsigned short* B = lds + 128 * 72 + (w / 2 % 2) * 64 * 72 + l31 * 72 + h * 8;
    f16v c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
    h8 a0 = *(const h8*)(A), a1 = *(const h8*)(A + 32 * 72), b0 = *(const h8*)(B), b1 = *(const h8*)(B + 32 * 72);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int nk = ((kk + 1) & 3) * 16;
            h8 na0 = *(const h8*)(A + nk), na1 = *(const h8*)(A + 32 * 72 + nk);
            h8 nb0 = *(const h8*)(B + nk), nb1 = *(const h8*)(B + 32 * 72 + nk);
            __builtin_amdgcn_sched_barrier(0);
            c00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, c00, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, c01, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c11, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
    }
    f16v s = c00 + c01 + c10 + c11;
    float t = 0; for (int r = 0; r < 16; ++r) t += s[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

typedef __attribute__((ext_vector_type(4))) float f4v;
// 16x16x32: a 64x64 wave tile = 4x4 tiles; per K=32 step 4 A + 4 B fragments (16 B each) feed 16 MFMAs
template <int WAVES, int LDR = 72>
__global__ __launch_bounds__(64 * WAVES) void k_lds16(const h8* in, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[128 * LDR * 2];
    for (int i = threadIdx.x; i < 128 * LDR * 2 / 8; i += blockDim.x) ((h8*)lds)[i] = in[i % 1024];
    __syncthreads();
    const int lane = threadIdx.x & 63, l15 = lane & 15, q = lane >> 4, w = threadIdx.x >> 6;
    const unsigned short* A = lds + (w % 2) * 64 * LDR + l15 * LDR + q * 8;
    const unsigned short* B = lds + 128 * LDR + (w / 2 % 2) * 64 * LDR + l15 * LDR + q * 8;
    f4v c[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) c[i][j] = f4v{0, 0, 0, 0};
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = *(const h8*)(A + i * 16 * LDR); b[i] = *(const h8*)(B + i * 16 * LDR); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int nk = ((kk + 1) & 1) * 32;
            h8 na[4], nb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { na[i] = *(const h8*)(A + i * 16 * LDR + nk); nb[i] = *(const h8*)(B + i * 16 * LDR + nk); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], c[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = na[i]; b[i] = nb[i]; }
        }
    }
    float t = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) t += c[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <typename F> double run(F f, int blocks, int threads, const h8* in, float* out, int iters, int mfma_per_iter) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(f, dim3(blocks), dim3(threads), 0, 0, in, out, iters / 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(f, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 5.0 * blocks * (threads / 64) * (double)iters * mfma_per_iter * 32768.0;
    return flops / (ms * 1e-3) / 1e12;
}

int main() {
    h8* in; float* out;
    hipMalloc(&in, 4096 * sizeof(h8)); hipMalloc(&out, 2048 * 1024 * sizeof(float));
    _Float16* hbuf = (_Float16*)malloc(4096 * 8 * 2);
    for (int i = 0; i < 4096 * 8; ++i) hbuf[i] = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
    hipMemcpy(in, hbuf, 4096 * 16, hipMemcpyHostToDevice);
    printf("reg  4 waves/CU (1/SIMD): %.0f TFLOP/s\n", run(k_reg<4>, 256, 256, in, out, 20000, 4));
    printf("reg  8 waves/CU (2/SIMD): %.0f TFLOP/s\n", run(k_reg<4>, 512, 256, in, out, 20000, 4));
    printf("lds  4 waves/CU (1/SIMD): %.0f TFLOP/s\n", run(k_lds<4>, 256, 256, in, out, 5000, 16));
    printf("lds  8 waves/CU (2/SIMD): %.0f TFLOP/s\n", run(k_lds<4>, 512, 256, in, out, 5000, 16));
    // 16x16x32: 32 MFMAs of 16384 FLOP per iteration = 16 "32x32x16-equivalents"
    printf("lds16x16x32 4 waves/CU: %.0f TFLOP/s\n", run(k_lds16<4>, 256, 256, in, out, 5000, 16));
    printf("lds16x16x32 8 waves/CU: %.0f TFLOP/s\n", run(k_lds16<4>, 512, 256, in, out, 5000, 16));
    // 160-byte rows: conflict-free for the 16-row x 4 k-group fragments (144-byte rows are 2-way conflicted)
    printf("lds16x16x32 4 waves/CU, 160-B rows: %.0f TFLOP/s\n", run(k_lds16<4, 80>, 256, 256, in, out, 5000, 16));
    printf("lds16x16x32 8 waves/CU, 160-B rows: %.0f TFLOP/s\n", run(k_lds16<4, 80>, 512, 256, in, out, 5000, 16));
    printf("lds  4 waves/CU (1/SIMD) again: %.0f TFLOP/s\n", run(k_lds<4>, 256, 256, in, out, 5000, 16));
    return 0;
}
