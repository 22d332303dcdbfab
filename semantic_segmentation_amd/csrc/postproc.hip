// Fake-image post-processing of the Unet step (running_files/train_end2end_jsrt.py:197-200): global min-max scaling of
// the generator output to [0,1], conversion to uint8, per-plane histogram equalisation (torchvision 0.14.1 F.equalize) and
// gamma 0.5 (F.adjust_gamma; its uint8 table is built by the host, steps.py), back to float.  Non-differentiable byte
// work: three launches instead of ~25 element-wise torch launches.  Integer histogram / LUT arithmetic follows
// torchvision's published algorithm; the float steps use non-contracted IEEE operations in torch's order.  Bit-exact
// against oracle/postproc.py (tests/test_steps_gpu.py).
//   pass 1: per-block min/max partials of the whole batch
//   pass 2: u8 = trunc(clamp((x-min)/(max-min) * 255 + 0.5)); per-image 256-bin histogram (LDS, then integer atomics)
//   pass 3: per-image LUT  lut[v] = (exclusive_cumsum[v] + step/2) / step,  step = (npix - hist[last non-zero bin]) / 255
//           (identity when step == 0), composed with the gamma table, applied per pixel
#include "common.hpp"

namespace {

constexpr int PP_BLOCKS = 512;      // min/max partials

__device__ __forceinline__ void block_minmax(float& mn, float& mx, float* red) {
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o, 64));
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w] = mn; red[4 + w] = mx; }
    __syncthreads();
    mn = fminf(fminf(red[0], red[1]), fminf(red[2], red[3]));
    mx = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    __syncthreads();
}

__global__ __launch_bounds__(256) void pp_minmax_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ part) {
    __shared__ float red[8];
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = x[i];
        mn = fminf(mn, v); mx = fmaxf(mx, v);
    }
    block_minmax(mn, mx, red);
    if (threadIdx.x == 0) { part[blockIdx.x] = mn; part[PP_BLOCKS + blockIdx.x] = mx; }
}

// every block re-reduces the (<= 512) partials: cheaper than another launch
__device__ __forceinline__ void global_minmax(const float* part, int nparts, float& mn, float& mx, float* red) {
    mn = INFINITY; mx = -INFINITY;
    for (int i = threadIdx.x; i < nparts; i += 256) { mn = fminf(mn, part[i]); mx = fmaxf(mx, part[PP_BLOCKS + i]); }
    block_minmax(mn, mx, red);
}

__device__ __forceinline__ int to_u8(float v, float mn, float d) {
    // torch: ((x - min) / (max - min)).mul(255).add_(0.5).clamp_(0, 255).to(uint8)   -- separate roundings, truncation
    float f = __fdiv_rn(__fsub_rn(v, mn), d);
    f = __fadd_rn(__fmul_rn(f, 255.f), 0.5f);
    f = fminf(fmaxf(f, 0.f), 255.f);
    return (int)f;
}

__global__ __launch_bounds__(256) void pp_hist_kernel(const float* __restrict__ x, int64_t hw, const float* __restrict__ part,
                                                      int nparts, int* __restrict__ hist) {
    __shared__ float red[8];
    __shared__ int lh[256];
    float mn, mx;
    global_minmax(part, nparts, mn, mx, red);
    const float d = __fsub_rn(mx, mn);
    lh[threadIdx.x] = 0;
    __syncthreads();
    const float* img = x + (int64_t)blockIdx.y * hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256)
        atomicAdd(&lh[to_u8(img[i], mn, d)], 1);
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(hist + blockIdx.y * 256 + threadIdx.x, lh[threadIdx.x]);
}

__global__ __launch_bounds__(256) void pp_apply_kernel(const float* __restrict__ x, int64_t hw, const float* __restrict__ part,
                                                       int nparts, const int* __restrict__ hist,
                                                       const float* __restrict__ gamma_lut, float* __restrict__ out) {
    __shared__ float red[8];
    __shared__ int csum[256];
    __shared__ float lut[256];
    __shared__ int s_step;
    float mn, mx;
    global_minmax(part, nparts, mn, mx, red);
    const float d = __fsub_rn(mx, mn);
    const int* h = hist + blockIdx.y * 256;
    csum[threadIdx.x] = h[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        int last = 0, total = 0, run = 0;
        for (int v = 0; v < 256; ++v) {
            const int c = csum[v];
            if (c > 0) last = c;
            total += c;
            csum[v] = run;                       // exclusive prefix sum
            run += c;
        }
        s_step = (total - last) / 255;
    }
    __syncthreads();
    {
        const int step = s_step, v = threadIdx.x;
        int e = v;                               // step == 0: identity (torchvision returns the image unchanged)
        if (step > 0) {
            e = (csum[v] + step / 2) / step;
            e = e < 0 ? 0 : (e > 255 ? 255 : e);
        }
        lut[v] = gamma_lut[e];
    }
    __syncthreads();
    const float* img = x + (int64_t)blockIdx.y * hw;
    float* o = out + (int64_t)blockIdx.y * hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256)
        o[i] = lut[to_u8(img[i], mn, d)];
}

}  // namespace

extern "C" int64_t gs_fake_postprocess_ws_floats(int N) { return N > 0 ? 2 * PP_BLOCKS + 256 * (int64_t)N : 0; }

extern "C" int gs_fake_postprocess(const float* x, float* out, float* ws, const float* gamma_lut, int N, int64_t hw,
                                   void* stream) {
    GS_CHECK_ARG(x && out && ws && gamma_lut && N > 0 && hw > 0, "gs_fake_postprocess: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)N * hw;
    int nb = (int)cdiv64(n, 256 * 8);
    if (nb > PP_BLOCKS) nb = PP_BLOCKS;
    if (nb < 1) nb = 1;
    int* hist = reinterpret_cast<int*>(ws + 2 * PP_BLOCKS);
    if (hipMemsetAsync(hist, 0, (size_t)N * 256 * sizeof(int), s) != hipSuccess) return GS_ELAUNCH;
    pp_minmax_kernel<<<nb, 256, 0, s>>>(x, n, ws);
    GS_CHECK_LAUNCH("gs_fake_postprocess");
    int bpi = (int)cdiv64(hw, 256 * 8);
    if (bpi > 64) bpi = 64;
    if (bpi < 1) bpi = 1;
    dim3 grid(bpi, N);
    pp_hist_kernel<<<grid, 256, 0, s>>>(x, hw, ws, nb, hist);
    GS_CHECK_LAUNCH("gs_fake_postprocess");
    pp_apply_kernel<<<grid, 256, 0, s>>>(x, hw, ws, nb, hist, gamma_lut, out);
    GS_CHECK_LAUNCH("gs_fake_postprocess");
    return GS_OK;
}
