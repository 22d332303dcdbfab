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

// a product that stays a product: the file is built with -ffp-contract=fast and __fmul_rn / __fadd_rn are plain operators to
// the compiler, so mul + add pairs would otherwise be fused into one FMA (one rounding instead of torch's two)
__device__ __forceinline__ float mul_keep(float a, float b) {
    float r = a * b;
    asm volatile("" : "+v"(r));
    return r;
}


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
    f = mul_keep(f, 255.f) + 0.5f;
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

// ---- ISIC variant: the `fake_trans` chain of running_files/train_end2end_isic.py:178-184,263-264 on the uint8 RGB batch ----
// RandomEqualize / RandomPosterize(4) / RandomAdjustSharpness(0.3) / RandomAutocontrast / ColorJitter(saturation) of
// torchvision 0.14.1 (functional_tensor.py) with the per-call random decisions made by the host.  uint8 planes live in the
// workspace; float steps are non-contracted IEEE operations in torchvision's order (bit-exact against oracle/postproc.py).
__global__ __launch_bounds__(256) void it_u8_kernel(const float* __restrict__ x, int64_t hw, const float* __restrict__ part,
                                                    int nparts, uint8_t* __restrict__ A, int* __restrict__ hist) {
    __shared__ float red[8];
    __shared__ int lh[256];
    float mn, mx;
    global_minmax(part, nparts, mn, mx, red);
    const float d = __fsub_rn(mx, mn);
    lh[threadIdx.x] = 0;
    __syncthreads();
    const float* img = x + (int64_t)blockIdx.y * hw;
    uint8_t* a = A + (int64_t)blockIdx.y * hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256) {
        const int v = to_u8(img[i], mn, d);
        a[i] = (uint8_t)v;
        if (hist) atomicAdd(&lh[v], 1);
    }
    if (hist) {
        __syncthreads();
        if (lh[threadIdx.x]) atomicAdd(hist + blockIdx.y * 256 + threadIdx.x, lh[threadIdx.x]);
    }
}

// A[i] = posterize(equalise_lut[A[i]]) in place; hist == NULL: no equalisation
__global__ __launch_bounds__(256) void it_lut_kernel(uint8_t* __restrict__ A, int64_t hw, const int* __restrict__ hist,
                                                     int pmask) {
    __shared__ int csum[256];
    __shared__ int lut[256];
    __shared__ int s_step;
    int e = threadIdx.x;
    if (hist) {
        csum[threadIdx.x] = hist[blockIdx.y * 256 + threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 0) {
            int last = 0, total = 0, run = 0;
            for (int v = 0; v < 256; ++v) {
                const int c = csum[v];
                if (c > 0) last = c;
                total += c;
                csum[v] = run;
                run += c;
            }
            s_step = (total - last) / 255;
        }
        __syncthreads();
        const int step = s_step;
        if (step > 0) {
            e = (csum[threadIdx.x] + step / 2) / step;
            e = e < 0 ? 0 : (e > 255 ? 255 : e);
        }
    }
    lut[threadIdx.x] = e & pmask;
    __syncthreads();
    uint8_t* a = A + (int64_t)blockIdx.y * hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256) a[i] = (uint8_t)lut[a[i]];
}

__device__ __forceinline__ int blend_u8(int a, int b, float r1, float r2) {
    float v = mul_keep(r1, (float)a) + mul_keep(r2, (float)b);
    v = fminf(fmaxf(v, 0.f), 255.f);
    return (int)v;
}

// adjust_sharpness: blend(img, blurred, ratio); blurred = round(3x3 [[1,1,1],[1,5,1],[1,1,1]]/13) inside, img on the border
__global__ __launch_bounds__(256) void it_sharp_kernel(const uint8_t* __restrict__ A, uint8_t* __restrict__ B, int H, int W,
                                                       float r1, float r2) {
    const int64_t hw = (int64_t)H * W;
    const uint8_t* a = A + (int64_t)blockIdx.y * hw;
    uint8_t* b = B + (int64_t)blockIdx.y * hw;
    const float k1 = __fdiv_rn(1.f, 13.f), k5 = __fdiv_rn(5.f, 13.f);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256) {
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        const int v = a[i];
        int blur = v;
        if (y >= 1 && y < H - 1 && x >= 1 && x < W - 1) {
            float acc = 0.f;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx)
                    acc = acc + mul_keep((dy == 0 && dx == 0) ? k5 : k1, (float)a[i + (int64_t)dy * W + dx]);
            blur = (int)rintf(acc);
        }
        b[i] = (uint8_t)blend_u8(v, blur, r1, r2);
    }
}

// per-plane max of v and of 255 - v (both start at 0)
__global__ __launch_bounds__(256) void it_minmax_kernel(const uint8_t* __restrict__ A, int64_t hw, int* __restrict__ mm) {
    __shared__ int red[2][4];
    const uint8_t* a = A + (int64_t)blockIdx.y * hw;
    int mx = 0, nmn = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256) {
        const int v = a[i];
        mx = max(mx, v); nmn = max(nmn, 255 - v);
    }
    for (int o = 32; o > 0; o >>= 1) { mx = max(mx, __shfl_xor(mx, o, 64)); nmn = max(nmn, __shfl_xor(nmn, o, 64)); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = mx; red[1][threadIdx.x >> 6] = nmn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(mm + 2 * blockIdx.y, max(max(red[0][0], red[0][1]), max(red[0][2], red[0][3])));
        atomicMax(mm + 2 * blockIdx.y + 1, max(max(red[1][0], red[1][1]), max(red[1][2], red[1][3])));
    }
}

// autocontrast (per plane) -> saturation blend with the grey image (3 channels) -> float / 255
__global__ __launch_bounds__(256) void it_final_kernel(const uint8_t* __restrict__ A, float* __restrict__ out, int C, int64_t hw,
                                                       const int* __restrict__ mm, float s1, float s2, int sat_on) {
    const int n = blockIdx.y;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256) {
        int v[4];
        for (int c = 0; c < C; ++c) {
            int u = A[((int64_t)n * C + c) * hw + i];
            if (mm) {
                const float mx = (float)mm[2 * (n * C + c)], mn = (float)(255 - mm[2 * (n * C + c) + 1]);
                float scale = __fdiv_rn(255.f, __fsub_rn(mx, mn)), lo = mn;
                if (!isfinite(scale)) { scale = 1.f; lo = 0.f; }
                float f = __fmul_rn(__fsub_rn((float)u, lo), scale);
                f = fminf(fmaxf(f, 0.f), 255.f);
                u = (int)f;
            }
            v[c] = u;
        }
        if (sat_on && C == 3) {
            const float g = (mul_keep(0.2989f, (float)v[0]) + mul_keep(0.587f, (float)v[1])) + mul_keep(0.114f, (float)v[2]);
            const int grey = (int)g;
            for (int c = 0; c < 3; ++c) v[c] = blend_u8(v[c], grey, s1, s2);
        }
        for (int c = 0; c < C; ++c) out[((int64_t)n * C + c) * hw + i] = __fdiv_rn((float)v[c], 255.f);
    }
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

// workspace: [2 * PP_BLOCKS floats min/max partials][planes * 256 ints histogram][planes * 2 ints plane max / 255 - min]
// [planes * hw bytes A][planes * hw bytes B]
extern "C" int64_t gs_isic_fake_trans_ws_bytes(int planes, int64_t hw) {
    if (planes <= 0 || hw <= 0) return 0;
    const int64_t plane_bytes = ((int64_t)planes * hw + 255) / 256 * 256;
    return (int64_t)(2 * PP_BLOCKS) * 4 + (int64_t)planes * 258 * 4 + 2 * plane_bytes + 256;
}

extern "C" int gs_isic_fake_trans(const float* x, float* out, void* ws, int N, int C, int H, int W, int equalize_on,
                                  int bits, int sharpness_on, float sharp_r1, float sharp_r2, int autocontrast_on,
                                  int saturation_on, float sat_r1, float sat_r2, void* stream) {
    GS_CHECK_ARG(x && out && ws && N > 0 && C >= 1 && C <= 4 && H > 0 && W > 0 && bits >= 1 && bits <= 8,
                 "gs_isic_fake_trans: bad arguments (1..4 channels)");
    GS_CHECK_ARG((int64_t)N * C <= 65535, "gs_isic_fake_trans: too many planes");
    hipStream_t s = (hipStream_t)stream;
    const int planes = N * C;
    const int64_t hw = (int64_t)H * W, n = (int64_t)planes * hw;
    float* part = reinterpret_cast<float*>(ws);
    int* hist = reinterpret_cast<int*>(part + 2 * PP_BLOCKS);
    int* mm = hist + (int64_t)planes * 256;
    uint8_t* A = reinterpret_cast<uint8_t*>(mm + (int64_t)planes * 2);
    A = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(A) + 255) / 256 * 256);
    uint8_t* B = A + ((int64_t)planes * hw + 255) / 256 * 256;
    int nb = (int)cdiv64(n, 256 * 8);
    nb = nb > PP_BLOCKS ? PP_BLOCKS : (nb < 1 ? 1 : nb);
    if (hipMemsetAsync(hist, 0, (size_t)planes * 258 * sizeof(int), s) != hipSuccess) return GS_ELAUNCH;
    pp_minmax_kernel<<<nb, 256, 0, s>>>(x, n, part);
    int bpi = (int)cdiv64(hw, 256 * 8);
    bpi = bpi > 64 ? 64 : (bpi < 1 ? 1 : bpi);
    dim3 grid(bpi, planes);
    it_u8_kernel<<<grid, 256, 0, s>>>(x, hw, part, nb, A, equalize_on ? hist : nullptr);
    it_lut_kernel<<<grid, 256, 0, s>>>(A, hw, equalize_on ? hist : nullptr, (-(1 << (8 - bits))) & 0xFF);
    uint8_t* cur = A;
    if (sharpness_on && H > 2 && W > 2) {
        it_sharp_kernel<<<grid, 256, 0, s>>>(A, B, H, W, sharp_r1, sharp_r2);
        cur = B;
    }
    if (autocontrast_on) it_minmax_kernel<<<grid, 256, 0, s>>>(cur, hw, mm);
    it_final_kernel<<<dim3(bpi, N), 256, 0, s>>>(cur, out, C, hw, autocontrast_on ? mm : nullptr, sat_r1, sat_r2, saturation_on);
    GS_CHECK_LAUNCH("gs_isic_fake_trans");
    return GS_OK;
}
