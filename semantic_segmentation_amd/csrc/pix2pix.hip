// Pix2Pix generator up-path: the DARTS-style mixed transposed convolution
//   y = sum_j softmax(arch)[j] * ConvTranspose2d_j(x),  (k,s,p) = (4,2,1), (6,2,2), (8,2,3)
// (models_pix2pix/networks.py:486-511, architecture_pix2pix/operations.py:14-39) is linear and centre
// aligned (k - 2p = 2 for all three), hence EXACTLY one ConvTranspose2d(k=8,s=2,p=3) with the merged kernel
//   Wm = w2*W8 + w1*pad1(W6) + w0*pad2(W4)            (116 taps -> 64 taps, -45 % FLOPs)
// This file merges the three fp32 parameter tensors straight into the 16-bit K-major packs the MFMA
// engine consumes (one pass over the 1.09 GB of generator weights, HBM-bound), and splits the merged
// weight gradient back into dW4/dW6/dW8 plus the three architecture-weight dot products.
//
// Sub-pixel decomposition of the stride-2 transposed conv: output pixel (2i+py, 2j+px) only sees taps
// ky == py+1 (mod 2), kx == px+1 (mod 2): 4 classes x 16 taps, input offset dy = (py+3-ky)/2.
//   fwd pack   P[c][t][Cout][Cin],  c = py*2+px, t = a*4+b, ky = 2a + (1-py), kx = 2b + (1-px)
//   dgrad pack D[ky*8+kx][Cin][Cout]   (dX = stride-2 conv of dY with the un-flipped kernel)
#include "common.hpp"

namespace {

__device__ __forceinline__ float merged_tap(const float* w4, const float* w6, const float* w8, int ky, int kx,
                                            float s0, float s1, float s2) {
    float v = s2 * w8[ky * 8 + kx];
    if (ky >= 1 && ky <= 6 && kx >= 1 && kx <= 6) v += s1 * w6[(ky - 1) * 6 + kx - 1];
    if (ky >= 2 && ky <= 5 && kx >= 2 && kx <= 5) v += s0 * w4[(ky - 2) * 4 + kx - 2];
    return v;
}

template <int DT>
__global__ __launch_bounds__(256) void upconv_merge_pack_kernel(const float* __restrict__ W4, const float* __restrict__ W6,
                                                                const float* __restrict__ W8, const float* __restrict__ sm,
                                                                unsigned short* pf, unsigned short* pd, float* wm32,
                                                                int Cin, int Cout) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)Cin * Cout) return;
    const int ci = (int)(idx % Cin), co = (int)(idx / Cin);
    const float s0 = sm[0], s1 = sm[1], s2 = sm[2];
    const int64_t pair = (int64_t)ci * Cout + co;
    const float* w4 = W4 + pair * 16;
    const float* w6 = W6 + pair * 36;
    const float* w8 = W8 + pair * 64;
    const int64_t plane = (int64_t)Cout * Cin;
    for (int ky = 0; ky < 8; ++ky)
        for (int kx = 0; kx < 8; ++kx) {
            const float v = merged_tap(w4, w6, w8, ky, kx, s0, s1, s2);
            const unsigned short h = Elem<DT>::from_f(v);
            if (pf) {
                const int py = 1 - (ky & 1), px = 1 - (kx & 1);
                const int c = py * 2 + px, t = (ky >> 1) * 4 + (kx >> 1);
                pf[((int64_t)(c * 16 + t)) * plane + (int64_t)co * Cin + ci] = h;
            }
            if (pd) pd[((int64_t)(ky * 8 + kx)) * plane + (int64_t)ci * Cout + co] = h;
            if (wm32) wm32[pair * 64 + ky * 8 + kx] = v;
        }
}

// dWm [4][16][Cout][Cin] fp32 (class major, from gs_conv_wgrad per class) -> dW4/dW6/dW8 (reference layouts,
// OVERWRITE, scaled by gscale*softmax weight) and dots[j] += gscale * <dWm window_j, W_j>.
__global__ __launch_bounds__(256) void upconv_split_wgrad_kernel(const float* __restrict__ dWm, const float* __restrict__ W4,
                                                                 const float* __restrict__ W6, const float* __restrict__ W8,
                                                                 const float* __restrict__ sm, float gscale, float* dW4,
                                                                 float* dW6, float* dW8, float* dots, int Cin, int Cout, float* dots_ws) {
    __shared__ float red[3][4];
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    if (idx < (int64_t)Cin * Cout) {
        const int ci = (int)(idx % Cin), co = (int)(idx / Cin);
        const float s0 = sm[0] * gscale, s1 = sm[1] * gscale, s2 = sm[2] * gscale;
        const int64_t pair = (int64_t)ci * Cout + co;
        const int64_t plane = (int64_t)Cout * Cin;
        for (int ky = 0; ky < 8; ++ky)
            for (int kx = 0; kx < 8; ++kx) {
                const int py = 1 - (ky & 1), px = 1 - (kx & 1);
                const int c = py * 2 + px, t = (ky >> 1) * 4 + (kx >> 1);
                const float g = dWm[((int64_t)(c * 16 + t)) * plane + (int64_t)co * Cin + ci];
                dW8[pair * 64 + ky * 8 + kx] = s2 * g;
                d2 += g * W8[pair * 64 + ky * 8 + kx];
                if (ky >= 1 && ky <= 6 && kx >= 1 && kx <= 6) {
                    const int o = (ky - 1) * 6 + kx - 1;
                    dW6[pair * 36 + o] = s1 * g;
                    d1 += g * W6[pair * 36 + o];
                }
                if (ky >= 2 && ky <= 5 && kx >= 2 && kx <= 5) {
                    const int o = (ky - 2) * 4 + kx - 2;
                    dW4[pair * 16 + o] = s0 * g;
                    d0 += g * W4[pair * 16 + o];
                }
            }
    }
    d0 = wave_sum(d0); d1 = wave_sum(d1); d2 = wave_sum(d2);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = d0; red[1][w] = d1; red[2][w] = d2; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const float v = (red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]) * gscale;
        // deterministic form: one slot per block, summed in block order by dots_reduce_kernel
        if (dots_ws) dots_ws[(int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = v;
        else atomicAdd(dots + threadIdx.x, v);
    }
}

// ---- tiled versions ------------------------------------------------------------------------------------
// The element-wise kernels above walk the [Cin][Cout][taps] parameters with one (ci, co) pair per lane: every load
// touches 64 cache lines and one of the packs is written 2 bytes at a time (measured 583 us per call, 17x the HBM
// time of the 1.09 GB of generator weights).  Here a block moves a (co x ci) tile through LDS: the parameters are
// read as contiguous runs, the pack is written as contiguous 128-byte runs.  The LDS column index is XOR-swizzled
// with the slot so that the transposing side of the exchange is (at most 2-way) conflict free.
__device__ __forceinline__ int tap_slot(int tap) {                     // class-major slot of tap (ky, kx)
    const int ky = tap >> 3, kx = tap & 7;
    return ((1 - (ky & 1)) * 2 + (1 - (kx & 1))) * 16 + (ky >> 1) * 4 + (kx >> 1);
}

// U merged taps per lane with every load issued up front: the k6 / k4 windows are read unconditionally from a clamped
// index and weighted by 0 outside the window (per-iteration conditional loads left one load in flight per lane)
template <int U>
__device__ __forceinline__ void merged_taps_batch(const float* __restrict__ W4, const float* __restrict__ W6,
                                                  const float* __restrict__ W8, const int64_t (&pair)[U], const int (&tap)[U],
                                                  float s0, float s1, float s2, float (&out)[U]) {
    float a8[U], a6[U], a4[U], m6[U], m4[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int ky = tap[u] >> 3, kx = tap[u] & 7;
        const bool in6 = ky >= 1 && ky <= 6 && kx >= 1 && kx <= 6, in4 = ky >= 2 && ky <= 5 && kx >= 2 && kx <= 5;
        a8[u] = W8[pair[u] * 64 + tap[u]];
        a6[u] = W6[pair[u] * 36 + (in6 ? (ky - 1) * 6 + kx - 1 : 0)];
        a4[u] = W4[pair[u] * 16 + (in4 ? (ky - 2) * 4 + kx - 2 : 0)];
        m6[u] = in6 ? s1 : 0.f;
        m4[u] = in4 ? s0 : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float v = s2 * a8[u];                      // same association as merged_tap()
        v += m6[u] * a6[u];
        v += m4[u] * a4[u];
        out[u] = v;
    }
}

// forward pack: tile 8 co x 64 ci, LDS [64 slots][8 co][64 ci]
template <int DT>
__global__ __launch_bounds__(256) void upconv_merge_pack_fwd_tiled(const float* __restrict__ W4, const float* __restrict__ W6,
                                                                   const float* __restrict__ W8, const float* __restrict__ sm,
                                                                   unsigned short* pf, int Cin, int Cout) {
    __shared__ unsigned short lds[64 * 8 * 64];
    const int co0 = blockIdx.x * 8, ci0 = blockIdx.y * 64;
    const float s0 = sm[0], s1 = sm[1], s2 = sm[2];
    constexpr int U = 8;
    for (int base = threadIdx.x; base < 64 * 8 * 64; base += 256 * U) {
        int64_t pair[U]; int tap[U], dst[U]; float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * 256;
            tap[u] = idx & 63;
            const int co_l = (idx >> 6) & 7, ci_l = idx >> 9;
            pair[u] = (int64_t)(ci0 + ci_l) * Cout + co0 + co_l;
            const int slot = tap_slot(tap[u]);
            dst[u] = (slot * 8 + co_l) * 64 + (ci_l ^ ((slot & 31) << 1));
        }
        merged_taps_batch<U>(W4, W6, W8, pair, tap, s0, s1, s2, v);
#pragma unroll
        for (int u = 0; u < U; ++u) lds[dst[u]] = Elem<DT>::from_f(v[u]);
    }
    __syncthreads();
    const int64_t plane = (int64_t)Cout * Cin;
    for (int idx = threadIdx.x; idx < 64 * 8 * 64; idx += 256) {
        const int ci_l = idx & 63, co_l = (idx >> 6) & 7, slot = idx >> 9;
        pf[slot * plane + (int64_t)(co0 + co_l) * Cin + ci0 + ci_l] = lds[(slot * 8 + co_l) * 64 + (ci_l ^ ((slot & 31) << 1))];
    }
}

// dgrad pack D[ky*8+kx][Cin][Cout]: tile 64 co x 8 ci, LDS [64 taps][8 ci][64 co]
template <int DT>
__global__ __launch_bounds__(256) void upconv_merge_pack_dgrad_tiled(const float* __restrict__ W4, const float* __restrict__ W6,
                                                                     const float* __restrict__ W8, const float* __restrict__ sm,
                                                                     unsigned short* pd, int Cin, int Cout) {
    __shared__ unsigned short lds[64 * 8 * 64];
    const int co0 = blockIdx.x * 64, ci0 = blockIdx.y * 8;
    const float s0 = sm[0], s1 = sm[1], s2 = sm[2];
    constexpr int U = 8;
    for (int base = threadIdx.x; base < 64 * 8 * 64; base += 256 * U) {
        int64_t pair[U]; int tap[U], dst[U]; float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * 256;
            tap[u] = idx & 63;
            const int co_l = (idx >> 6) & 63, ci_l = idx >> 12;
            pair[u] = (int64_t)(ci0 + ci_l) * Cout + co0 + co_l;
            dst[u] = (tap[u] * 8 + ci_l) * 64 + (co_l ^ ((tap[u] & 31) << 1));
        }
        merged_taps_batch<U>(W4, W6, W8, pair, tap, s0, s1, s2, v);
#pragma unroll
        for (int u = 0; u < U; ++u) lds[dst[u]] = Elem<DT>::from_f(v[u]);
    }
    __syncthreads();
    const int64_t plane = (int64_t)Cout * Cin;
    for (int idx = threadIdx.x; idx < 64 * 8 * 64; idx += 256) {
        const int co_l = idx & 63, ci_l = (idx >> 6) & 7, tap = idx >> 9;
        pd[tap * plane + (int64_t)(ci0 + ci_l) * Cout + co0 + co_l] = lds[(tap * 8 + ci_l) * 64 + (co_l ^ ((tap & 31) << 1))];
    }
}

// gradient split: tile 8 co x 32 ci, LDS fp32 [64 slots][8 co][32 ci]
// (nparts > 1: dWm is a stack of split-K slabs [nparts][64 slots][Cout][Cin] of the weight-gradient GEMMs, summed here in part
// order -- the separate ordered-sum pass wrote and re-read the 134 MB of a 1024 -> 512 layer's merged gradient)
__global__ __launch_bounds__(256) void upconv_split_wgrad_tiled(const float* __restrict__ dWm, const float* __restrict__ W4,
                                                                const float* __restrict__ W6, const float* __restrict__ W8,
                                                                const float* __restrict__ sm, float gscale, float* dW4,
                                                                float* dW6, float* dW8, float* dots, int Cin, int Cout, float* dots_ws,
                                                                int nparts, int64_t pstride) {
    __shared__ float lds[64 * 8 * 32];
    __shared__ float red[3][4];
    const int co0 = blockIdx.x * 8, ci0 = blockIdx.y * 32;
    const int64_t plane = (int64_t)Cout * Cin;
    if (nparts <= 1) {
#pragma unroll 8
        for (int idx = threadIdx.x; idx < 64 * 8 * 32; idx += 256) {
            const int ci_l = idx & 31, co_l = (idx >> 5) & 7, slot = idx >> 8;
            lds[(slot * 8 + co_l) * 32 + (ci_l ^ (slot & 31))] = dWm[slot * plane + (int64_t)(co0 + co_l) * Cin + ci0 + ci_l];
        }
    } else {
        constexpr int U = 8;
        for (int base = threadIdx.x; base < 64 * 8 * 32; base += 256 * U) {
            float v[U];
            const float* src[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * 256;
                const int ci_l = idx & 31, co_l = (idx >> 5) & 7, slot = idx >> 8;
                src[u] = dWm + slot * plane + (int64_t)(co0 + co_l) * Cin + ci0 + ci_l;
                v[u] = *src[u];
            }
            for (int p = 1; p < nparts; ++p) {
                float t[U];
#pragma unroll
                for (int u = 0; u < U; ++u) t[u] = src[u][(int64_t)p * pstride];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] += t[u];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * 256;
                const int ci_l = idx & 31, co_l = (idx >> 5) & 7, slot = idx >> 8;
                lds[(slot * 8 + co_l) * 32 + (ci_l ^ (slot & 31))] = v[u];
            }
        }
    }
    __syncthreads();
    const float s0 = sm[0] * gscale, s1 = sm[1] * gscale, s2 = sm[2] * gscale;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    constexpr int U = 8;
    for (int base = threadIdx.x; base < 64 * 8 * 32; base += 256 * U) {
        int64_t pair[U]; int tap[U]; float g[U], a8[U], a6[U], a4[U]; bool in6[U], in4[U]; int o6[U], o4[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                    // every load up front (clamped indices, masked later)
            const int idx = base + u * 256;
            tap[u] = idx & 63;
            const int co_l = (idx >> 6) & 7, ci_l = idx >> 9;
            const int ky = tap[u] >> 3, kx = tap[u] & 7;
            const int slot = tap_slot(tap[u]);
            g[u] = lds[(slot * 8 + co_l) * 32 + (ci_l ^ (slot & 31))];
            pair[u] = (int64_t)(ci0 + ci_l) * Cout + co0 + co_l;
            in6[u] = ky >= 1 && ky <= 6 && kx >= 1 && kx <= 6;
            in4[u] = ky >= 2 && ky <= 5 && kx >= 2 && kx <= 5;
            o6[u] = in6[u] ? (ky - 1) * 6 + kx - 1 : 0;
            o4[u] = in4[u] ? (ky - 2) * 4 + kx - 2 : 0;
            a8[u] = W8[pair[u] * 64 + tap[u]];
            a6[u] = W6[pair[u] * 36 + o6[u]];
            a4[u] = W4[pair[u] * 16 + o4[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            dW8[pair[u] * 64 + tap[u]] = s2 * g[u];
            d2 += g[u] * a8[u];
            if (in6[u]) { dW6[pair[u] * 36 + o6[u]] = s1 * g[u]; d1 += g[u] * a6[u]; }
            if (in4[u]) { dW4[pair[u] * 16 + o4[u]] = s0 * g[u]; d0 += g[u] * a4[u]; }
        }
    }
    d0 = wave_sum(d0); d1 = wave_sum(d1); d2 = wave_sum(d2);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = d0; red[1][w] = d1; red[2][w] = d2; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const float v = (red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]) * gscale;
        // deterministic form: one slot per block, summed in block order by dots_reduce_kernel
        if (dots_ws) dots_ws[(int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = v;
        else atomicAdd(dots + threadIdx.x, v);
    }
}

// dots[j] += sum over blocks of ws[b][j], in block order (fp64 running sum): 3 lanes, latency irrelevant (<= 8192 blocks)
__global__ void dots_reduce_kernel(const float* __restrict__ ws, int nblocks, float* dots) {
    __shared__ double red[3][64];
    const int j = threadIdx.x / 64, l = threadIdx.x % 64;
    double s = 0.0;
    for (int b = l; b < nblocks; b += 64) s += (double)ws[(int64_t)b * 3 + j];
    red[j][l] = s;
    __syncthreads();
    if (l == 0) {
        double t = 0.0;
        for (int i = 0; i < 64; ++i) t += red[j][i];
        dots[j] += (float)t;
    }
}

}  // namespace

namespace {
// ---------------------------------------------------------------------------------------------------------------------
// Outermost generator layer (networks.py:588-593 with the merged 8x8 kernel): ConvTranspose2d(Cin -> 1..4 channels,
// k 8, stride 2, pad 3) + bias + tanh, straight to the fp32 NCHW image.  On the MFMA engine this layer fills 1 of 64
// N columns; here it is a direct VALU kernel: a block stages the (8+4)^2 input pixels of an 8x8 patch of input positions
// and the four sub-pixel classes' weights (the cached class-major pack [4][16][cpad][Cin]) in LDS; wave = class (weights
// are wave-uniform LDS broadcasts), lane = input position; 16 taps x Cin/8 chunks of packed dot products.
//   class (py,px), tap (iy,ix): input (y + py + 1 - iy, x + px + 1 - ix) -> output (2y+py, 2x+px)   [geom_convT_class]
// ---------------------------------------------------------------------------------------------------------------------
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

template <int DT>
__device__ __forceinline__ float dot8(const uint4& a, const uint4& b, float acc) {
    if constexpr (DT == GS_F16) {
        const unsigned int av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, av[i]), __builtin_bit_cast(half2_t, bv[i]), acc, false);
        return acc;
    } else {
        float fa[8], fb[8];
        unpack8<DT>(a, fa); unpack8<DT>(b, fb);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += fa[i] * fb[i];
        return acc;
    }
}

constexpr int UO_T = 8, UO_HW = UO_T + 4, UO_MAXC = 4;

template <int DT, int NCO>
__global__ __launch_bounds__(256) void upconv8_image_fwd_kernel(const unsigned short* __restrict__ x, int in_stride, int in_coff,
                                                                const unsigned short* __restrict__ pf, int cpad,
                                                                const float* __restrict__ bias, float* __restrict__ out,
                                                                unsigned short* __restrict__ u, int N, int h, int w, int Cin,
                                                                int act) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    const int nch = Cin >> 3;
    const int xstride = Cin + 8;                          // +16 bytes per pixel: conflict-free 16-byte reads across lanes
    unsigned short* xs = lds;                             // [12*12][xstride]
    unsigned short* ws = lds + UO_HW * UO_HW * xstride;   // [4 classes][16 taps][NCO][Cin]
    const int tiles_x = (w + UO_T - 1) / UO_T, tiles_y = (h + UO_T - 1) / UO_T;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int y0 = ty * UO_T, x0 = tx * UO_T;
    const int t = threadIdx.x;
    for (int i = t; i < UO_HW * UO_HW * nch; i += 256) {
        const int c = i % nch, p = i / nch;
        const int py_ = p / UO_HW, px_ = p - py_ * UO_HW;
        const int gy = y0 + py_ - 2, gx = x0 + px_ - 2;
        uint4 v = make_uint4(0, 0, 0, 0);
        if ((unsigned)gy < (unsigned)h && (unsigned)gx < (unsigned)w)
            v = *reinterpret_cast<const uint4*>(x + ((int64_t)(n * h + gy) * w + gx) * in_stride + in_coff + c * 8);
        *reinterpret_cast<uint4*>(xs + p * xstride + c * 8) = v;
    }
    for (int i = t; i < 64 * NCO * nch; i += 256) {
        const int c = i % nch;
        int r = i / nch;
        const int co = r % NCO; r /= NCO;                 // r = cls*16 + tap
        *reinterpret_cast<uint4*>(ws + ((int64_t)r * NCO + co) * Cin + c * 8) =
            *reinterpret_cast<const uint4*>(pf + ((int64_t)r * cpad + co) * Cin + c * 8);
    }
    __syncthreads();
    const int cls = t >> 6, lane = t & 63;
    const int py = cls >> 1, px = cls & 1;
    const int qy = lane >> 3, qx = lane & 7;
    float acc[NCO];
#pragma unroll
    for (int co = 0; co < NCO; ++co) acc[co] = 0.f;
    for (int iy = 0; iy < 4; ++iy)
        for (int ix = 0; ix < 4; ++ix) {
            const unsigned short* xp = xs + ((qy + py + 1 - iy + 2) * UO_HW + (qx + px + 1 - ix + 2)) * xstride;
            const unsigned short* wp = ws + (int64_t)((cls * 16 + iy * 4 + ix) * NCO) * Cin;
            for (int c = 0; c < nch; ++c) {
                const uint4 a = *reinterpret_cast<const uint4*>(xp + c * 8);
#pragma unroll
                for (int co = 0; co < NCO; ++co)
                    acc[co] = dot8<DT>(a, *reinterpret_cast<const uint4*>(wp + co * Cin + c * 8), acc[co]);
            }
        }
    const int iy_ = y0 + qy, ix_ = x0 + qx;
    if (iy_ < h && ix_ < w) {
        const int oy = 2 * iy_ + py, ox = 2 * ix_ + px, OH = 2 * h, OW = 2 * w;
        float pre[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int co = 0; co < NCO; ++co) {
            const float v = acc[co] + (bias ? bias[co] : 0.f);
            pre[co] = v;
            out[((int64_t)(n * NCO + co) * OH + oy) * OW + ox] = act_fwd(v, act);
        }
        if (u) *reinterpret_cast<uint4*>(u + ((int64_t)(n * OH + oy) * OW + ox) * cpad) = pack8<DT>(pre);
    }
}
}  // namespace

extern "C" int gs_upconv8_image_fwd(const void* x, int in_pix_stride, int in_coff, const void* pack_fwd, int cpad,
                                    const float* bias, float* out, void* u, int N, int h, int w, int Cin, int Cout, int act,
                                    int dtype, void* stream) {
    GS_CHECK_ARG(x && pack_fwd && out && N > 0 && h > 0 && w > 0, "gs_upconv8_image_fwd: bad arguments");
    GS_CHECK_ARG(Cin > 0 && Cin % 8 == 0 && Cout >= 1 && Cout <= UO_MAXC && cpad == 8, "gs_upconv8_image_fwd: Cin %% 8, 1 <= Cout <= 4, cpad == 8");
    GS_CHECK_ARG(in_pix_stride >= in_coff + Cin && in_pix_stride % 8 == 0 && in_coff % 8 == 0, "gs_upconv8_image_fwd: bad input stride");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_upconv8_image_fwd: bad dtype");
    const size_t lds = ((size_t)UO_HW * UO_HW * (Cin + 8) + (size_t)64 * Cout * Cin) * 2;
    GS_CHECK_ARG(lds <= 64 * 1024, "gs_upconv8_image_fwd: Cin %d x Cout %d does not fit the 64 KB LDS tile", Cin, Cout);
    const int64_t blocks = (int64_t)N * cdiv(h, UO_T) * cdiv(w, UO_T);
    GS_CHECK_ARG(blocks < 2147483000LL, "gs_upconv8_image_fwd: too many tiles");
    hipStream_t st = (hipStream_t)stream;
#define UO_LAUNCH(DT, NCO)                                                                                           \
    upconv8_image_fwd_kernel<DT, NCO><<<(int)blocks, 256, lds, st>>>((const unsigned short*)x, in_pix_stride, in_coff,   \
                                                                     (const unsigned short*)pack_fwd, cpad, bias, out,   \
                                                                     (unsigned short*)u, N, h, w, Cin, act)
    if (dtype == GS_F16) {
        if (Cout == 1) UO_LAUNCH(GS_F16, 1); else if (Cout == 2) UO_LAUNCH(GS_F16, 2);
        else if (Cout == 3) UO_LAUNCH(GS_F16, 3); else UO_LAUNCH(GS_F16, 4);
    } else {
        if (Cout == 1) UO_LAUNCH(GS_BF16, 1); else if (Cout == 2) UO_LAUNCH(GS_BF16, 2);
        else if (Cout == 3) UO_LAUNCH(GS_BF16, 3); else UO_LAUNCH(GS_BF16, 4);
    }
#undef UO_LAUNCH
    GS_CHECK_LAUNCH("gs_upconv8_image_fwd");
    return GS_OK;
}

// ---- weight gradient of the same layer (the merged 8x8 / s2 / p3 transposed conv to ONE image channel, networks.py:588-593) ------
// dW[ci][ky][kx] = sum over input pixels (n, a, b) of x[n,a,b][ci] * du[n, 2a + ky - 3, 2b + kx - 3]   (zero outside the image),
// stored where gs_upconv_split_wgrad reads the merged gradient: dwm[cls][tap][0][ci] with cls = 2 (1 - ky % 2) + (1 - kx % 2),
// tap = 4 (ky / 2) + kx / 2.  As a GEMM this is [Cin x pixels] . [pixels x 64 taps] with ONE output channel: the generic engine
// pads it to 8 couts per sub-pixel class and ran it at 1.7 ms for batch 32 (134 ms ... 0.13 ms at batch 2).  Here a block walks
// tiles of 64 input pixels of one row: the 8 x 136 window of du (fp32) and the 64 x 128 tile of x sit in LDS, a thread owns one
// input channel and four kernel rows (32 accumulators) and reads the du window at wave-uniform addresses (broadcast); four
// pixels per iteration share their overlapping window columns.  Blocks are persistent (<= UW_MAX_BLOCKS): one partial
// [64 taps][128 channels] per block, summed in block order by the second kernel -- deterministic, no atomics.
namespace {
constexpr int UW_PX = 64, UW_COLS = 2 * UW_PX + 8, UW_MAX_BLOCKS = 512;

template <int DT>
__global__ __launch_bounds__(256) void upconv8_image_wgrad_kernel(const unsigned short* __restrict__ x, int xs,
                                                                  const unsigned short* __restrict__ du, int dus,
                                                                  float* __restrict__ ws, int N, int h, int w, int Cin,
                                                                  int tiles_x, int ntiles) {
    __shared__ __attribute__((aligned(16))) float dl[8][UW_COLS];
    __shared__ __attribute__((aligned(16))) unsigned short xl[UW_PX][128];
    const int t = threadIdx.x, ci = t & 127, kyh = t >> 7;
    const int c0 = blockIdx.y * 128;
    float acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    const int OH = 2 * h, OW = 2 * w;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int r = tile;
        const int tx = r % tiles_x; r /= tiles_x;
        const int a = r % h;
        const int n = r / h;
        const int x0 = tx * UW_PX;
        __syncthreads();                                   // the previous tile has been consumed
        for (int i = t; i < 8 * UW_COLS; i += 256) {
            const int rr = i / UW_COLS, cc = i - rr * UW_COLS;
            const int oy = 2 * a - 3 + rr, ox = 2 * x0 - 3 + cc;
            float v = 0.f;
            if ((unsigned)oy < (unsigned)OH && (unsigned)ox < (unsigned)OW) v = Elem<DT>::to_f(du[((int64_t)(n * OH + oy) * OW + ox) * dus]);
            dl[rr][cc] = v;
        }
        for (int i = t; i < UW_PX * 16; i += 256) {
            const int px = i >> 4, c8 = i & 15;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (x0 + px < w) v = *reinterpret_cast<const uint4*>(x + ((int64_t)(n * h + a) * w + x0 + px) * xs + c0 + c8 * 8);
            *reinterpret_cast<uint4*>(&xl[px][c8 * 8]) = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int p = 0; p < UW_PX; p += 4) {
            float xv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[j] = Elem<DT>::to_f(xl[p + j][ci]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float2* row = reinterpret_cast<const float2*>(&dl[4 * kyh + k][2 * p]);      // wave-uniform: broadcast reads
                float d[14];
#pragma unroll
                for (int q = 0; q < 7; ++q) { const float2 v = row[q]; d[2 * q] = v.x; d[2 * q + 1] = v.y; }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int kx = 0; kx < 8; ++kx) acc[k][kx] += xv[j] * d[2 * j + kx];
            }
        }
    }
    float* dst = ws + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 64) * 128;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int kx = 0; kx < 8; ++kx) dst[((4 * kyh + k) * 8 + kx) * 128 + ci] = acc[k][kx];
}

// 16 outputs per block; 16 lane groups each sum every 16th partial (eight loads in flight), then the 16 group sums are added in
// group order: a fixed tree, deterministic.  (One thread per output walking all <= 512 partials took as long as the main kernel.)
__global__ __launch_bounds__(256) void upconv8_image_wgrad_reduce(const float* __restrict__ ws, float* __restrict__ dwm, int nblk, int Cin) {
    __shared__ float part[16][17];
    const int t = threadIdx.x, o = t & 15, g = t >> 4;
    const int i = blockIdx.x * 16 + o;                     // (tap64, channel); 64 * Cin is a multiple of 16
    const int c = i % Cin, tap = i / Cin, ky = tap >> 3, kx = tap & 7;
    const float* src = ws + ((int64_t)(c >> 7) * nblk * 64 + tap) * 128 + (c & 127);
    float s = 0.f;
    int b = g;
    for (; b + 7 * 16 < nblk; b += 8 * 16) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(b + 16 * u) * 64 * 128];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < nblk; b += 16) s += src[(int64_t)b * 64 * 128];
    part[g][o] = s;
    __syncthreads();
    if (g == 0) {
        float r = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) r += part[q][o];
        const int cls = 2 * (1 - (ky & 1)) + (1 - (kx & 1)), t16 = 4 * (ky >> 1) + (kx >> 1);
        dwm[(int64_t)(cls * 16 + t16) * Cin + c] = r;
    }
}

}  // namespace

static int uw_blocks(int N, int h, int w, int Cin) {
    const int64_t ntiles = (int64_t)N * h * cdiv(w, UW_PX);
    int nb = UW_MAX_BLOCKS / (Cin / 128);
    if (nb < 1) nb = 1;
    return (int)(ntiles < nb ? ntiles : nb);
}

extern "C" int gs_upconv8_image_wgrad_ok(int Cin, int Cout) { return Cout == 1 && Cin >= 128 && Cin % 128 == 0; }
extern "C" int64_t gs_upconv8_image_wgrad_ws_floats(int N, int h, int w, int Cin) {
    if (N <= 0 || h <= 0 || w <= 0 || Cin < 128 || Cin % 128) return 0;
    return (int64_t)(Cin / 128) * uw_blocks(N, h, w, Cin) * 64 * 128;
}
extern "C" int gs_upconv8_image_wgrad(const void* x, int x_pix_stride, const void* du, int du_pix_stride, float* ws, float* dwm, int N,
                                      int h, int w, int Cin, int dtype, void* stream) {
    GS_CHECK_ARG(x && du && ws && dwm && N > 0 && h > 0 && w > 0, "gs_upconv8_image_wgrad: bad arguments");
    GS_CHECK_ARG(Cin >= 128 && Cin % 128 == 0 && x_pix_stride >= Cin && x_pix_stride % 8 == 0 && du_pix_stride >= 1,
                 "gs_upconv8_image_wgrad: Cin %% 128 == 0, x stride %% 8 == 0");
    GS_CHECK_ARG(((uintptr_t)x & 15) == 0, "gs_upconv8_image_wgrad: x must be 16-byte aligned");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_upconv8_image_wgrad: bad dtype");
    const int tiles_x = cdiv(w, UW_PX);
    const int64_t ntiles = (int64_t)N * h * tiles_x;
    GS_CHECK_ARG(ntiles < 2147483000LL && (int64_t)N * 4 * h * w < 2147483000LL, "gs_upconv8_image_wgrad: too many pixels");
    const int nb = uw_blocks(N, h, w, Cin);
    dim3 grid(nb, Cin / 128);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GS_F16)
        upconv8_image_wgrad_kernel<GS_F16><<<grid, 256, 0, st>>>((const unsigned short*)x, x_pix_stride, (const unsigned short*)du,
                                                                  du_pix_stride, ws, N, h, w, Cin, tiles_x, (int)ntiles);
    else
        upconv8_image_wgrad_kernel<GS_BF16><<<grid, 256, 0, st>>>((const unsigned short*)x, x_pix_stride, (const unsigned short*)du,
                                                                   du_pix_stride, ws, N, h, w, Cin, tiles_x, (int)ntiles);
    upconv8_image_wgrad_reduce<<<64 * Cin / 16, 256, 0, st>>>(ws, dwm, nb, Cin);
    GS_CHECK_LAUNCH("gs_upconv8_image_wgrad");
    return GS_OK;
}

extern "C" int gs_upconv_merge_pack(const float* w4, const float* w6, const float* w8, const float* softmax3,
                                    void* pack_fwd, void* pack_dgrad, float* merged_f32, int Cin, int Cout, int dtype,
                                    void* stream) {
    GS_CHECK_ARG(w4 && w6 && w8 && softmax3 && (pack_fwd || pack_dgrad || merged_f32) && Cin > 0 && Cout > 0,
                 "gs_upconv_merge_pack: bad arguments");
    const int64_t n = (int64_t)Cin * Cout;
    const int nb = (int)cdiv64(n, 256);
    hipStream_t s = (hipStream_t)stream;
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_upconv_merge_pack: bad dtype");
    if (pack_fwd && !pack_dgrad && !merged_f32 && Cin % 64 == 0 && Cout % 8 == 0) {
        dim3 grid(Cout / 8, Cin / 64);
        if (dtype == GS_F16) upconv_merge_pack_fwd_tiled<GS_F16><<<grid, 256, 0, s>>>(w4, w6, w8, softmax3, (unsigned short*)pack_fwd, Cin, Cout);
        else upconv_merge_pack_fwd_tiled<GS_BF16><<<grid, 256, 0, s>>>(w4, w6, w8, softmax3, (unsigned short*)pack_fwd, Cin, Cout);
        GS_CHECK_LAUNCH("gs_upconv_merge_pack");
        return GS_OK;
    }
    if (pack_dgrad && !pack_fwd && !merged_f32 && Cout % 64 == 0 && Cin % 8 == 0) {
        dim3 grid(Cout / 64, Cin / 8);
        if (dtype == GS_F16) upconv_merge_pack_dgrad_tiled<GS_F16><<<grid, 256, 0, s>>>(w4, w6, w8, softmax3, (unsigned short*)pack_dgrad, Cin, Cout);
        else upconv_merge_pack_dgrad_tiled<GS_BF16><<<grid, 256, 0, s>>>(w4, w6, w8, softmax3, (unsigned short*)pack_dgrad, Cin, Cout);
        GS_CHECK_LAUNCH("gs_upconv_merge_pack");
        return GS_OK;
    }
    if (dtype == GS_F16)
        upconv_merge_pack_kernel<GS_F16><<<nb, 256, 0, s>>>(w4, w6, w8, softmax3, (unsigned short*)pack_fwd,
                                                             (unsigned short*)pack_dgrad, merged_f32, Cin, Cout);
    else if (dtype == GS_BF16)
        upconv_merge_pack_kernel<GS_BF16><<<nb, 256, 0, s>>>(w4, w6, w8, softmax3, (unsigned short*)pack_fwd,
                                                              (unsigned short*)pack_dgrad, merged_f32, Cin, Cout);
    else GS_CHECK_ARG(false, "gs_upconv_merge_pack: bad dtype");
    GS_CHECK_LAUNCH("gs_upconv_merge_pack");
    return GS_OK;
}

static int split_wgrad_blocks(int Cin, int Cout) {
    if (Cin % 32 == 0 && Cout % 8 == 0) return (Cout / 8) * (Cin / 32);
    return (int)cdiv64((int64_t)Cin * Cout, 256);
}

static int upconv_split_wgrad_launch(const float* dwm, const float* w4, const float* w6, const float* w8,
                                     const float* softmax3, float gscale, float* dw4, float* dw6, float* dw8,
                                     float* dots3, int Cin, int Cout, float* dots_ws, void* stream, int nparts = 1,
                                     int64_t pstride = 0) {
    GS_CHECK_ARG(dwm && w4 && w6 && w8 && softmax3 && dw4 && dw6 && dw8 && dots3 && Cin > 0 && Cout > 0,
                 "gs_upconv_split_wgrad: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (Cin % 32 == 0 && Cout % 8 == 0) {
        dim3 grid(Cout / 8, Cin / 32);
        upconv_split_wgrad_tiled<<<grid, 256, 0, st>>>(dwm, w4, w6, w8, softmax3, gscale, dw4, dw6, dw8, dots3, Cin, Cout,
                                                       dots_ws, nparts, pstride);
    } else {
        GS_CHECK_ARG(nparts <= 1, "gs_upconv_split_wgrad_parts: needs Cin %% 32 == 0 and Cout %% 8 == 0");
        upconv_split_wgrad_kernel<<<split_wgrad_blocks(Cin, Cout), 256, 0, st>>>(dwm, w4, w6, w8, softmax3, gscale, dw4, dw6,
                                                                                 dw8, dots3, Cin, Cout, dots_ws);
    }
    GS_CHECK_LAUNCH("gs_upconv_split_wgrad");
    if (dots_ws) {
        dots_reduce_kernel<<<1, 192, 0, st>>>(dots_ws, split_wgrad_blocks(Cin, Cout), dots3);
        GS_CHECK_LAUNCH("gs_upconv_split_wgrad");
    }
    return GS_OK;
}

extern "C" int gs_upconv_split_wgrad(const float* dwm, const float* w4, const float* w6, const float* w8,
                                     const float* softmax3, float gscale, float* dw4, float* dw6, float* dw8,
                                     float* dots3, int Cin, int Cout, void* stream) {
    return upconv_split_wgrad_launch(dwm, w4, w6, w8, softmax3, gscale, dw4, dw6, dw8, dots3, Cin, Cout, nullptr, stream);
}

extern "C" int64_t gs_upconv_split_wgrad_ws_floats(int Cin, int Cout) {
    return (Cin > 0 && Cout > 0) ? (int64_t)3 * split_wgrad_blocks(Cin, Cout) : 0;
}

// dwm = `nparts` split-K slabs [64 slots][Cout][Cin] at a distance of `part_stride` floats (the output of
// gs_conv_wgrad_slabs_batch for the four classes): summed in part order while splitting.  0: not covered (sum first).
extern "C" int gs_upconv_split_wgrad_parts_ok(int Cin, int Cout) { return (Cin % 32 == 0 && Cout % 8 == 0) ? 1 : 0; }
extern "C" int gs_upconv_split_wgrad_parts(const float* slabs, int nparts, int64_t part_stride, const float* w4, const float* w6,
                                           const float* w8, const float* softmax3, float gscale, float* dw4, float* dw6,
                                           float* dw8, float* dots3, float* ws, int Cin, int Cout, void* stream) {
    GS_CHECK_ARG(ws != nullptr && nparts >= 1 && part_stride >= (int64_t)64 * Cin * Cout, "gs_upconv_split_wgrad_parts: bad arguments");
    return upconv_split_wgrad_launch(slabs, w4, w6, w8, softmax3, gscale, dw4, dw6, dw8, dots3, Cin, Cout, ws, stream, nparts,
                                     part_stride);
}

extern "C" int gs_upconv_split_wgrad_det(const float* dwm, const float* w4, const float* w6, const float* w8,
                                         const float* softmax3, float gscale, float* dw4, float* dw6, float* dw8,
                                         float* dots3, float* ws, int Cin, int Cout, void* stream) {
    GS_CHECK_ARG(ws != nullptr, "gs_upconv_split_wgrad_det: null workspace");
    return upconv_split_wgrad_launch(dwm, w4, w6, w8, softmax3, gscale, dw4, dw6, dw8, dots3, Cin, Cout, ws, stream);
}
