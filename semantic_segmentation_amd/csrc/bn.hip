// Train-mode BatchNorm split around the MFMA convolutions, fused with the activation, the 2x2 max-pool
// and the skip-concat write (all HBM-bound: 16-byte vector accesses, one pass per tensor).
//   forward : conv epilogue -> per-tile partial sums -> gs_bn_finalize -> gs_bn_act_apply
//   backward: gs_bn_act_bwd_reduce -> gs_bn_bwd_coeffs -> gs_bn_act_bwd_apply
// Reference semantics: torch.nn.BatchNorm2d(train) + ReLU/LeakyReLU + MaxPool2d(2) + torch.cat
// (unet/unet_parts.py:17-21,34,67 ; models_pix2pix/networks.py:583-586,606-607,642-657).
#include <stdlib.h>

#include "common.hpp"

namespace {

constexpr int RED_SLICES = 64;

// ---- two-stage deterministic reduction of [ntiles][2][C] partials ---------------------------------
// stage 1: grid (ceil(C/32), nslices); block 256 = 8 tile-lanes x 32 channels -> out1[slice][2][C] (double)
__global__ __launch_bounds__(256) void reduce_partials_stage1(const float* __restrict__ part, int ntiles, int C,
                                                              int tiles_per_slice, double* __restrict__ out1) {
    __shared__ double red[2][8][32];
    const int cl = threadIdx.x & 31, tl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const int t0 = blockIdx.y * tiles_per_slice, t1 = min(ntiles, t0 + tiles_per_slice);
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        for (int tt = t0 + tl; tt < t1; tt += 8) {
            s1 += (double)part[((int64_t)tt * 2 + 0) * C + c];
            s2 += (double)part[((int64_t)tt * 2 + 1) * C + c];
        }
    }
    red[0][tl][cl] = s1;
    red[1][tl][cl] = s2;
    __syncthreads();
    if (tl == 0 && c < C) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { a += red[0][i][cl]; b += red[1][i][cl]; }
        out1[((int64_t)blockIdx.y * 2 + 0) * C + c] = a;
        out1[((int64_t)blockIdx.y * 2 + 1) * C + c] = b;
    }
}

// sum of the stage-1 slices for 32 channels per block: 8 slice-lanes per channel through LDS (fixed order)
// (T = double: the stage-1 slices; T = float: the tile partials themselves, where there are few enough to skip stage 1)
template <typename T>
__device__ __forceinline__ void sum_slices_32x8(const T* __restrict__ s, int nslices, int C, double& s1, double& s2) {
    __shared__ double red[2][8][32];
    const int cl = threadIdx.x & 31, tl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double a = 0.0, b = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int i = tl; i < nslices; i += 8) { a += (double)s[((int64_t)i * 2 + 0) * C + c]; b += (double)s[((int64_t)i * 2 + 1) * C + c]; }
    }
    red[0][tl][cl] = a;
    red[1][tl][cl] = b;
    __syncthreads();
    s1 = 0.0; s2 = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { s1 += red[0][i][cl]; s2 += red[1][i][cl]; }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const T* __restrict__ s, int nslices, int C, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* running_mean, float* running_var, float momentum,
                                                          float eps, float* scale, float* shift, float* mean_out,
                                                          float* invstd_out) {
    double s1, s2;
    sum_slices_32x8(s, nslices, C, s1, s2);
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    if (c >= C || threadIdx.x >= 32) return;
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    const float sc = gm * invstd;
    scale[c] = sc;
    shift[c] = bt - (float)mean * sc;
    mean_out[c] = (float)mean;
    invstd_out[c] = invstd;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) {
        const double unb = count > 1.0 ? var * (count / (count - 1.0)) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

__global__ void bn_eval_coeffs_kernel(int C, const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, float* scale, float* shift, float* mean, float* invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float is = 1.f / sqrtf(rv[c] + eps);
    const float sc = (gamma ? gamma[c] : 1.f) * is;
    scale[c] = sc;
    shift[c] = (beta ? beta[c] : 0.f) - rm[c] * sc;
    mean[c] = rm[c];
    invstd[c] = is;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_coeffs_kernel(const T* __restrict__ s, int nslices, int C, double count,
                                                            float gscale, float* dgamma, float* dbeta, float* c1, float* c2) {
    double s1, s2;
    sum_slices_32x8(s, nslices, C, s1, s2);
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    if (c >= C || threadIdx.x >= 32) return;
    if (dbeta) dbeta[c] = (float)(s1 * gscale);
    if (dgamma) dgamma[c] = (float)(s2 * gscale);
    c1[c] = (float)(s1 / count);
    c2[c] = (float)(s2 / count);
}

// Traversal direction of the three element-wise passes (GSSEG_BN_REV bit mask: 1 forward apply, 2 backward reduce, 4 backward
// apply run tail first).  Each pass streams a tensor that the previous kernel has just written (or read) head to tail: its
// tail is what the 256 MB Infinity Cache still holds, so a pass that starts there takes part of its reads from the cache.
static int bn_traversal() {
    static const int v = getenv("GSSEG_BN_REV") ? atoi(getenv("GSSEG_BN_REV")) : 3;     // measured: 14.62 -> 14.53 ms per step
    return v;
}

// ---- forward apply -------------------------------------------------------------------------------
struct ApplyArgs {
    const unsigned short* y;
    const float* scale;
    const float* shift;
    unsigned short* z;
    unsigned short* zp;
    const uint8_t* keep;
    float keep_scale;
    int act, N, H, W, C, zs, zc;
    int rev;                      // traverse the tensor tail first (see bn_traversal())
};

template <int DT, bool POOL, bool GENERIC>
__global__ __launch_bounds__(256) void bn_act_apply_kernel(const ApplyArgs a) {
    const float slope = a.act == GS_ACT_RELU ? 0.f : (a.act == GS_ACT_LEAKY02 ? 0.2f : 1.f);
    auto fwd_a = [&](float v) __attribute__((always_inline)) { return GENERIC ? act_fwd(v, a.act) : (v > 0.f ? v : v * slope); };
    const int nch = a.C >> 3;
    const int PH = POOL ? (a.H + 1) / 2 : a.H, PW = POOL ? (a.W + 1) / 2 : a.W;
    const int64_t total = (int64_t)a.N * PH * PW * nch;
    for (int64_t lidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; lidx < total;
         lidx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t idx = a.rev ? total - 1 - lidx : lidx;
        const int ch = (int)(idx % nch);
        int64_t pidx = idx / nch;
        const int px = (int)(pidx % PW); pidx /= PW;
        const int py = (int)(pidx % PH);
        const int n = (int)(pidx / PH);
        const int c0 = ch * 8;
        float sc[8], sh[8];
        if (a.scale) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { sc[i] = a.scale[c0 + i]; sh[i] = a.shift[c0 + i]; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) { sc[i] = 1.f; sh[i] = 0.f; }
        }
        if (!POOL) {
            const int64_t pix = ((int64_t)n * a.H + py) * a.W + px;
            float v[8];
            unpack8<DT>(*reinterpret_cast<const uint4*>(a.y + pix * a.C + c0), v);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = fwd_a(v[i] * sc[i] + sh[i]);
            if (a.keep) {
                const uint2 k = *reinterpret_cast<const uint2*>(a.keep + pix * a.C + c0);
                const unsigned int kw[2] = {k.x, k.y};
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] *= ((kw[i >> 2] >> (8 * (i & 3))) & 0xffu) ? a.keep_scale : 0.f;
            }
            st16(a.z + pix * a.zs + a.zc + c0, pack8<DT>(v));
        } else {
            float mx[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) mx[i] = -INFINITY;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int yy = 2 * py + dy, xx = 2 * px + dx;
                    if (yy < a.H && xx < a.W) {
                        const int64_t pix = ((int64_t)n * a.H + yy) * a.W + xx;
                        float v[8];
                        unpack8<DT>(*reinterpret_cast<const uint4*>(a.y + pix * a.C + c0), v);
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = fwd_a(v[i] * sc[i] + sh[i]);
                        const uint4 pk = pack8<DT>(v);
                        st16(a.z + pix * a.zs + a.zc + c0, pk);
                        // pool the ROUNDED values: the next layer sees exactly max over the stored z
                        float r[8];
                        unpack8<DT>(pk, r);
#pragma unroll
                        for (int i = 0; i < 8; ++i) mx[i] = fmaxf(mx[i], r[i]);
                    }
                }
            if (py < a.H / 2 && px < a.W / 2) {
                const int64_t pp = ((int64_t)n * (a.H / 2) + py) * (a.W / 2) + px;
                st16(a.zp + pp * a.C + c0, pack8<DT>(mx));
            }
        }
    }
}

// ---- backward ------------------------------------------------------------------------------------
struct BwdArgs {
    const unsigned short* y;
    const unsigned short* dza;
    const unsigned short* dzp;
    const unsigned short* dzb;      // second dense gradient source with its own activation (act_b)
    const uint8_t* keep;            // dropout keep-mask applied to dza (dense [N,H,W,C])
    float keep_scale;
    int act_b;
    const float *scale, *shift, *mean, *invstd, *c1, *c2;
    float* partials;
    unsigned short* dy;
    int sa, ca, act, bn, N, H, W, C;
    int tile_units;   // work units (pixels or 2x2 windows) per tile
    int rev;          // blocks take the tiles last to first (see bn_traversal())
    // HEAD source (instead of dza): the gradient w.r.t. this activation is that of a pointwise head with <= 4 outputs,
    // dz[p][c] = sum_k head_dl[n][k][hw] * head_w[k][c] -- formed here from the 2..16 MB of logit gradients instead of
    // being written as a [N,H,W,C] tensor by the head's data-gradient kernel and read back twice
    const float* head_dl;
    const float* head_w;
    int head_n;
};

// gradient w.r.t. z at one pixel for 8 channels: concat/skip part + max-pool routed part.
// v[] are the pre-activation values (y*scale+shift) of THIS pixel; for the pool part the caller supplies
// `win` = whether this pixel is the first arg-max of its window, per channel.
// GENERIC = some activation is tanh (runtime switch per element); otherwise the activations are the slope family
// (identity / ReLU / LeakyReLU) and their value / derivative is one compare + select -- the per-element uniform
// `switch` of act_fwd/act_grad costs a scalar branch per element and held these kernels at ~3 TB/s.
// EXTRA: the second dense source (dzb) and / or the dropout keep-mask are present (Pix2Pix); without them their loads and
// registers are compiled out (145-164 VGPRs otherwise: three waves per SIMD)
template <int DT, bool POOL, bool APPLY, bool GENERIC, int HEAD = 0, bool EXTRA = true>   // HEAD: 0 = tensor sources; 2 / 4 = a head of <= 2 / 4 outputs
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const BwdArgs a) {
    static_assert(!HEAD || (!POOL && !GENERIC), "the head source: plain pixels, slope-family activation");
    constexpr int NH = HEAD ? HEAD : 1;
    constexpr int HEAD_CHUNK = 2048;                        // pixels whose logit gradients are staged in LDS at a time
    __shared__ float hdl[HEAD ? NH * HEAD_CHUNK : 1];
    __shared__ float red[2][256 * 8 / 8][8];   // [stat][thread][8 channels] -- reduced below
    const float slope_a = a.act == GS_ACT_RELU ? 0.f : (a.act == GS_ACT_LEAKY02 ? 0.2f : 1.f);
    const float slope_b = a.act_b == GS_ACT_RELU ? 0.f : (a.act_b == GS_ACT_LEAKY02 ? 0.2f : 1.f);
    auto grad_a = [&](float v) __attribute__((always_inline)) { return GENERIC ? act_grad(v, a.act) : (v > 0.f ? 1.f : slope_a); };
    auto grad_b = [&](float v) __attribute__((always_inline)) { return GENERIC ? act_grad(v, a.act_b) : (v > 0.f ? 1.f : slope_b); };
    auto fwd_a = [&](float v) __attribute__((always_inline)) { return GENERIC ? act_fwd(v, a.act) : (v > 0.f ? v : v * slope_a); };
    const int nch = a.C >> 3;
    const int PH = POOL ? (a.H + 1) / 2 : a.H, PW = POOL ? (a.W + 1) / 2 : a.W;
    const int units = a.N * PH * PW;                       // host guarantees < 2^31
    // thread -> (chunk, unit-lane): chunk fastest so a wave reads contiguous channels
    const int lanes_per_unit = nch < 256 ? nch : 256;      // threads covering the channel chunks of one unit
    const int unit_lanes = 256 / lanes_per_unit;           // units processed concurrently per block
    const int chl = threadIdx.x % lanes_per_unit, ul = threadIdx.x / lanes_per_unit;
    const int tile = a.rev ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
    const int u0 = tile * a.tile_units;
    int u1 = u0 + a.tile_units < units ? u0 + a.tile_units : units;
    if (ul >= unit_lanes) u1 = u0;   // leftover threads (256 % lanes_per_unit) only join the barriers
    constexpr int UNR = 4;           // plain path: four pixels per lane in flight (the head source with eight: no faster)

    for (int ch = chl; ch - chl < nch; ch += lanes_per_unit) {      // uniform trip count: the block barriers below
        const bool ch_ok = ch < nch;
        const int c0 = ch_ok ? ch * 8 : 0;
        if (!ch_ok) u1 = u0;
        float sc[8], sh[8], mu[8], is[8], k1[8], k2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            sc[i] = a.scale ? a.scale[c0 + i] : 1.f;
            sh[i] = a.shift ? a.shift[c0 + i] : 0.f;
            mu[i] = a.mean ? a.mean[c0 + i] : 0.f;
            is[i] = a.invstd ? a.invstd[c0 + i] : 1.f;
            k1[i] = (APPLY && a.c1) ? a.c1[c0 + i] : 0.f;
            k2[i] = (APPLY && a.c2) ? a.c2[c0 + i] : 0.f;
        }
        float s1[8], s2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
        float hw[NH][8];                                    // HEAD: the head's weights for this channel chunk
        if (HEAD) {
#pragma unroll
            for (int k = 0; k < NH; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) hw[k][i] = k < a.head_n ? a.head_w[k * a.C + c0 + i] : 0.f;
        }
        const int HWp = a.H * a.W;

        if (!POOL) {
            // a unit IS a pixel: no index decomposition at all.
            // HEAD: the tile is walked in chunks of HEAD_CHUNK pixels whose logit gradients the block stages in LDS first (one
            // coalesced pass with the (image, pixel) split per loaded element) -- as scattered per-lane loads they made the
            // pass slower: 91 -> 83 us (reduce, 285 MB; the tensor path reads 536 MB in 99 us).
            const int tile_end = u0 + a.tile_units < units ? u0 + a.tile_units : units;      // block-uniform
            const int chunk = HEAD ? HEAD_CHUNK : (tile_end - u0 > 0 ? tile_end - u0 : 1);
            for (int cb = u0; cb < tile_end; cb += chunk) {
            const int ce = cb + chunk < tile_end ? cb + chunk : tile_end;
            if (HEAD) {
                __syncthreads();
                for (int i = threadIdx.x; i < ce - cb; i += 256) {
                    const int p = cb + i, n = p / HWp, hwp = p - n * HWp;
#pragma unroll
                    for (int q = 0; q < NH; ++q) hdl[q * HEAD_CHUNK + i] = q < a.head_n ? a.head_dl[((int64_t)n * a.head_n + q) * HWp + hwp] : 0.f;
                }
                __syncthreads();
            }
            const int ue = ce < u1 ? ce : u1;                 // (u1 = u0 for a thread that only joins the barriers)
            for (int ub = cb + ul; ub < ue; ub += unit_lanes * UNR) {
                uint4 ry[UNR], rg[UNR], rb[UNR];
                uint2 rk[UNR];
                bool ok[UNR];
                float hd[UNR][NH];
#pragma unroll
                for (int k = 0; k < UNR; ++k) {
                    const int u = ub + k * unit_lanes;
                    ok[k] = u < ue;
                    const int64_t pix = ok[k] ? u : cb;
                    ry[k] = *reinterpret_cast<const uint4*>(a.y + pix * a.C + c0);
                    if (HEAD) {
#pragma unroll
                        for (int q = 0; q < NH; ++q) hd[k][q] = hdl[q * HEAD_CHUNK + (int)pix - cb];
                        rg[k] = make_uint4(0, 0, 0, 0);
                    } else {
                        rg[k] = a.dza ? *reinterpret_cast<const uint4*>(a.dza + pix * a.sa + a.ca + c0) : make_uint4(0, 0, 0, 0);
                    }
                    rb[k] = (EXTRA && a.dzb) ? *reinterpret_cast<const uint4*>(a.dzb + pix * a.C + c0) : make_uint4(0, 0, 0, 0);
                    rk[k] = (EXTRA && a.keep) ? *reinterpret_cast<const uint2*>(a.keep + pix * a.C + c0) : make_uint2(0, 0);
                }
#pragma unroll
                for (int k = 0; k < UNR; ++k) {
                    if (!ok[k]) continue;
                    const int64_t pix = ub + k * unit_lanes;
                    float yv[8], g[8], gb[8], out[8];
                    unpack8<DT>(ry[k], yv);
                    unpack8<DT>(rg[k], g);
                    unpack8<DT>(rb[k], gb);
                    if (HEAD) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            float t = hd[k][0] * hw[0][i];
#pragma unroll
                            for (int q = 1; q < NH; ++q) t += hd[k][q] * hw[q][i];
                            g[i] = t;
                        }
                    }
                    if (EXTRA && a.keep) {
                        const unsigned int kw[2] = {rk[k].x, rk[k].y};
#pragma unroll
                        for (int i = 0; i < 8; ++i) g[i] *= ((kw[i >> 2] >> (8 * (i & 3))) & 0xffu) ? a.keep_scale : 0.f;
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float v = yv[i] * sc[i] + sh[i];
                        const float gh = g[i] * grad_a(v) + ((EXTRA && a.dzb) ? gb[i] * grad_b(v) : 0.f);
                        const float xh = (yv[i] - mu[i]) * is[i];
                        if (APPLY) out[i] = a.bn ? sc[i] * (gh - k1[i] - xh * k2[i]) : gh;
                        else { s1[i] += gh; s2[i] += gh * xh; }
                    }
                    if (APPLY) st16(a.dy + pix * a.C + c0, pack8<DT>(out));
                }
            }
            }       // chunks
        } else {
            const int PH2 = a.H / 2, PW2 = a.W / 2;
            for (int u = u0 + ul; u < u1; u += unit_lanes) {
                const int px = u % PW;
                const int r = u / PW;
                const int py = r % PH;
                const int n = r / PH;
                // 2x2 window: all nine loads are issued up front (clamped addresses; `ok` masks the results),
                // then the stored (rounded) z of the 4 pixels is recomputed and dzp routed to the first max
                uint4 ry[4], rg[4];
                bool ok[4];
                int64_t pixk[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int yy = 2 * py + (k >> 1), xx = 2 * px + (k & 1);
                    ok[k] = yy < a.H && xx < a.W;
                    pixk[k] = ((int64_t)n * a.H + (yy < a.H ? yy : a.H - 1)) * a.W + (xx < a.W ? xx : a.W - 1);
                    ry[k] = *reinterpret_cast<const uint4*>(a.y + pixk[k] * a.C + c0);
                    rg[k] = a.dza ? *reinterpret_cast<const uint4*>(a.dza + pixk[k] * a.sa + a.ca + c0) : make_uint4(0, 0, 0, 0);
                }
                const bool pooled = (py < PH2) && (px < PW2);
                uint4 rp = make_uint4(0, 0, 0, 0);
                if (pooled && a.dzp) {
                    const int64_t pp = ((int64_t)n * PH2 + py) * PW2 + px;
                    rp = *reinterpret_cast<const uint4*>(a.dzp + pp * a.C + c0);
                }
                float yv[4][8], zr[4][8], gp[8];
                unpack8<DT>(rp, gp);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    unpack8<DT>(ry[k], yv[k]);
                    float zt[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) zt[i] = fwd_a(yv[k][i] * sc[i] + sh[i]);
                    const uint4 pk = pack8<DT>(zt);
                    unpack8<DT>(pk, zr[k]);
                    if (!ok[k]) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) zr[k][i] = -INFINITY;
                    }
                }
                int amax[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    int best = 0; float bv = zr[0][i];
#pragma unroll
                    for (int k = 1; k < 4; ++k) if (zr[k][i] > bv) { bv = zr[k][i]; best = k; }
                    amax[i] = best;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (!ok[k]) continue;
                    float g[8], out[8];
                    unpack8<DT>(rg[k], g);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float gz = g[i] + (amax[i] == k ? gp[i] : 0.f);
                        const float v = yv[k][i] * sc[i] + sh[i];
                        const float gh = gz * grad_a(v);
                        const float xh = (yv[k][i] - mu[i]) * is[i];
                        if (APPLY) out[i] = a.bn ? sc[i] * (gh - k1[i] - xh * k2[i]) : gh;
                        else { s1[i] += gh; s2[i] += gh * xh; }
                    }
                    if (APPLY) st16(a.dy + pixk[k] * a.C + c0, pack8<DT>(out));
                }
            }
        }
        if (!APPLY) {
            // reduce over the unit lanes of this block (fixed order -> deterministic)
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) { red[0][threadIdx.x][i] = s1[i]; red[1][threadIdx.x][i] = s2[i]; }
            __syncthreads();
            // one lane per (statistic, channel): unit_lanes LDS reads each, summed in ascending unit-lane order (the
            // order of the former 8-lane serial loop, which was 6-10 % of the kernel on the 64-channel layers)
            const int nv = lanes_per_unit * 8;
            for (int v = threadIdx.x; v < 2 * nv; v += 256) {
                const int st = v >= nv ? 1 : 0, idx = v - st * nv;
                const int cl = idx >> 3, i = idx & 7;
                if (ch - chl + cl >= nch) continue;
                float tsum = 0.f;
                for (int k = 0; k < unit_lanes; ++k) tsum += red[st][k * lanes_per_unit + cl][i];
                a.partials[(int64_t)tile * 2 * a.C + st * a.C + (ch - chl + cl) * 8 + i] = tsum;
            }
        }
    }
}

// ---- strided column sums ---------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void colsum_stage1(const unsigned short* __restrict__ t, int ps, int coff,
                                                     int64_t npix, int C, int64_t pix_per_block, float* ws,
                                                     int H, int W, int y0, int x0, int h, int w) {
    __shared__ float red[256][8];
    const int nch = C >> 3;
    const int lpu = nch < 256 ? nch : 256, ulanes = 256 / lpu;
    const int chl = threadIdx.x % lpu, ul = threadIdx.x / lpu;
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    int64_t p1 = p0 + pix_per_block < npix ? p0 + pix_per_block : npix;
    if (ul >= ulanes) p1 = p0;
    for (int ch = chl; ch - chl < nch; ch += lpu) {                 // uniform trip count (block barriers below)
        float s[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = 0.f;
        if (ch >= nch) p1 = p0;
        for (int64_t p = p0 + ul; p < p1; p += ulanes) {
            float v[8];
            const int xx = (int)(p % w);
            const int64_t r = p / w;
            const int yy = (int)(r % h);
            const int64_t n = r / h;
            const int64_t pp = (n * H + y0 + yy) * W + x0 + xx;
            unpack8<DT>(*reinterpret_cast<const uint4*>(t + pp * ps + coff + ch * 8), v);
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] += v[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) red[threadIdx.x][i] = s[i];
        __syncthreads();
        for (int v = threadIdx.x; v < lpu * 8; v += 256) {            // one lane per channel, ascending unit-lane order
            const int cl = v >> 3, i = v & 7;
            if (ch - chl + cl >= nch) continue;
            float tt = 0.f;
            for (int k = 0; k < ulanes; ++k) tt += red[k * lpu + cl][i];
            ws[(int64_t)blockIdx.x * C + (ch - chl + cl) * 8 + i] = tt;
        }
    }
}
// block 256 = 8 row-lanes x 32 channels; fixed summation order -> deterministic
__global__ __launch_bounds__(256) void colsum_stage2(const float* ws, int nblocks, int C, float gscale, float* out) {
    __shared__ double red[8][32];
    const int cl = threadIdx.x & 31, tl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s = 0.0;
    if (c < C) {
        // eight independent loads in flight per lane (the chain of dependent adds was latency bound: 34 us for 1024 rows)
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int i = tl;
        for (; i + 56 < nblocks; i += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ws[(int64_t)(i + 8 * u) * C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += (double)v[u];
        }
        for (; i < nblocks; i += 8) a[0] += (double)ws[(int64_t)i * C + c];
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    red[tl][cl] = s;
    __syncthreads();
    if (tl == 0 && c < C) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i][cl];
        out[c] = (float)(t * gscale);
    }
}

// Column sums out of a convolution's tile partials ([ntiles][2][Cfull], sum slot): the bias gradient of the transposed
// convolution whose output is the channel range [coff, coff + C) of the tensor that convolution wrote.
// stage 1: grid (ceil(C/32), nslices) -> out1[slice][C] (double); final: the slices (T = double) or the tiles (T = float).
__global__ __launch_bounds__(256) void partials_colsum_stage1(const float* __restrict__ part, int ntiles, int Cfull, int coff, int C,
                                                              int tiles_per_slice, double* __restrict__ out1) {
    __shared__ double red[8][32];
    const int cl = threadIdx.x & 31, tl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const int t0 = blockIdx.y * tiles_per_slice, t1 = min(ntiles, t0 + tiles_per_slice);
    double s1 = 0.0;
    if (c < C)
        for (int tt = t0 + tl; tt < t1; tt += 8) s1 += (double)part[(int64_t)tt * 2 * Cfull + coff + c];
    red[tl][cl] = s1;
    __syncthreads();
    if (tl == 0 && c < C) {
        double a = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) a += red[i][cl];
        out1[(int64_t)blockIdx.y * C + c] = a;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void partials_colsum_final(const T* __restrict__ s, int n, int64_t stride, int C, float gscale,
                                                             float* __restrict__ out) {
    __shared__ double red[8][32];
    const int cl = threadIdx.x & 31, tl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double a = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int i = tl; i < n; i += 8) a += (double)s[(int64_t)i * stride + c];
    }
    red[tl][cl] = a;
    __syncthreads();
    if (tl == 0 && c < C) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i][cl];
        out[c] = (float)(t * gscale);
    }
}

// Up to this many tiles the finalising kernels sum the tile partials themselves (one launch instead of two: each is at the
// ~5 us launch floor, and the deep levels and the batch-2 steps of config 3 have few tiles); GSSEG_BN_DIRECT_TILES overrides.
int direct_tiles() {
    static const int v = getenv("GSSEG_BN_DIRECT_TILES") ? atoi(getenv("GSSEG_BN_DIRECT_TILES")) : 512;
    return v;
}

int reduce_partials(const float* partials, int ntiles, int C, double** out, int* nslices, hipStream_t s) {
    // stage-1 output lives right behind the partials: [ntiles][2][C] floats, then [64][2][C] doubles
    int slices = ntiles < RED_SLICES ? ntiles : RED_SLICES;
    const int tps = cdiv(ntiles, slices);
    slices = cdiv(ntiles, tps);
    size_t off = ((size_t)ntiles * 2 * C * sizeof(float) + 15) & ~(size_t)15;
    double* o1 = (double*)((char*)partials + off);
    reduce_partials_stage1<<<dim3(cdiv(C, 32), slices), 256, 0, s>>>(partials, ntiles, C, tps, o1);
    *out = o1;
    *nslices = slices;
    return 0;
}

}  // namespace

// Size (in floats) a partials buffer needs for `ntiles` tiles of C channels, including the stage-1 area.
extern "C" int64_t gs_bn_partials_floats(int ntiles, int C) {
    return (int64_t)ntiles * 2 * C + 4 + (int64_t)RED_SLICES * 2 * C * 2;
}

extern "C" int gs_bn_finalize(const float* partials, int ntiles, int C, double count, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                              float* scale, float* shift, float* mean, float* invstd, void* stream) {
    GS_CHECK_ARG(partials && scale && shift && mean && invstd && ntiles > 0 && C > 0 && count > 0,
                 "gs_bn_finalize: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (ntiles <= direct_tiles()) {
        bn_finalize_kernel<float><<<cdiv(C, 32), 256, 0, s>>>(partials, ntiles, C, count, gamma, beta, running_mean, running_var,
                                                               momentum, eps, scale, shift, mean, invstd);
        GS_CHECK_LAUNCH("gs_bn_finalize");
        return GS_OK;
    }
    double* o1; int ns;
    reduce_partials(partials, ntiles, C, &o1, &ns, s);
    bn_finalize_kernel<double><<<cdiv(C, 32), 256, 0, s>>>(o1, ns, C, count, gamma, beta, running_mean, running_var,
                                                            momentum, eps, scale, shift, mean, invstd);
    GS_CHECK_LAUNCH("gs_bn_finalize");
    return GS_OK;
}

extern "C" int gs_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                                 const float* running_var, float eps, float* scale, float* shift, float* mean,
                                 float* invstd, void* stream) {
    GS_CHECK_ARG(C > 0 && running_mean && running_var && scale && shift && mean && invstd, "gs_bn_eval_coeffs: bad args");
    bn_eval_coeffs_kernel<<<cdiv(C, 128), 128, 0, (hipStream_t)stream>>>(C, gamma, beta, running_mean, running_var, eps,
                                                                         scale, shift, mean, invstd);
    GS_CHECK_LAUNCH("gs_bn_eval_coeffs");
    return GS_OK;
}

extern "C" int gs_bn_act_apply(const void* y, const float* scale, const float* shift, int act, void* z,
                               int z_pix_stride, int z_coff, void* zp, const uint8_t* keep_mask, float keep_scale,
                               int N, int H, int W, int C, int dtype, void* stream) {
    GS_CHECK_ARG(y && z && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "gs_bn_act_apply: bad arguments");
    GS_CHECK_ARG(z_pix_stride >= z_coff + C && z_pix_stride % 8 == 0 && z_coff % 8 == 0, "gs_bn_act_apply: bad z stride");
    GS_CHECK_ARG((scale == nullptr) == (shift == nullptr), "gs_bn_act_apply: scale/shift must both be given or NULL");
    GS_CHECK_ARG(!(zp && keep_mask), "gs_bn_act_apply: pool + dropout not supported together");
    ApplyArgs a{(const unsigned short*)y, scale, shift, (unsigned short*)z, (unsigned short*)zp, keep_mask, keep_scale,
                act, N, H, W, C, z_pix_stride, z_coff, bn_traversal() & 1};
    const bool pool = zp != nullptr;
    const int PH = pool ? (H + 1) / 2 : H, PW = pool ? (W + 1) / 2 : W;
    const int64_t total = (int64_t)N * PH * PW * (C / 8);
    int64_t blocks = cdiv64(total, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipStream_t s = (hipStream_t)stream;
    const bool generic = act == GS_ACT_TANH;
    if (dtype == GS_F16) {
        if (generic) {
            if (pool) bn_act_apply_kernel<GS_F16, true, true><<<(int)blocks, 256, 0, s>>>(a);
            else bn_act_apply_kernel<GS_F16, false, true><<<(int)blocks, 256, 0, s>>>(a);
        } else {
            if (pool) bn_act_apply_kernel<GS_F16, true, false><<<(int)blocks, 256, 0, s>>>(a);
            else bn_act_apply_kernel<GS_F16, false, false><<<(int)blocks, 256, 0, s>>>(a);
        }
    } else if (dtype == GS_BF16) {
        if (generic) {
            if (pool) bn_act_apply_kernel<GS_BF16, true, true><<<(int)blocks, 256, 0, s>>>(a);
            else bn_act_apply_kernel<GS_BF16, false, true><<<(int)blocks, 256, 0, s>>>(a);
        } else {
            if (pool) bn_act_apply_kernel<GS_BF16, true, false><<<(int)blocks, 256, 0, s>>>(a);
            else bn_act_apply_kernel<GS_BF16, false, false><<<(int)blocks, 256, 0, s>>>(a);
        }
    } else {
        GS_CHECK_ARG(false, "gs_bn_act_apply: bad dtype");
    }
    GS_CHECK_LAUNCH("gs_bn_act_apply");
    return GS_OK;
}

static int bwd_tile_units(int64_t units) {
    int64_t tu = cdiv64(units, 1024);
    if (tu < 64) tu = 64;
    return (int)tu;
}

extern "C" int gs_bn_bwd_tiles(int N, int H, int W) {
    // the same tile count is valid for the pooled (2x2 window) and the plain variant: use the larger (plain)
    const int64_t units = (int64_t)N * H * W;
    return (int)cdiv64(units, bwd_tile_units(units));
}

static int launch_bwd(const BwdArgs& a0, bool apply, int dtype, hipStream_t s, int* ntiles_out) {
    BwdArgs a = a0;
    a.rev = (bn_traversal() >> (apply ? 2 : 1)) & 1;
    const bool pool = a.dzp != nullptr;
    const int PH = pool ? (a.H + 1) / 2 : a.H, PW = pool ? (a.W + 1) / 2 : a.W;
    const int64_t units = (int64_t)a.N * PH * PW;
    // tiles are defined on the PLAIN pixel count so gs_bn_bwd_tiles() bounds both variants
    const int ntiles_max = gs_bn_bwd_tiles(a.N, a.H, a.W);
    a.tile_units = (int)cdiv64(units, ntiles_max);
    if (a.tile_units < 1) a.tile_units = 1;
    const int ntiles = (int)cdiv64(units, a.tile_units);
    if (ntiles_out) *ntiles_out = ntiles;
    const bool generic = a.act == GS_ACT_TANH || (a.dzb && a.act_b == GS_ACT_TANH);
#define LAUNCH(DT, G)                                                                                   \
    do {                                                                                                \
        if (pool) {                                                                                     \
            if (apply) bn_act_bwd_kernel<DT, true, true, G><<<ntiles, 256, 0, s>>>(a);                  \
            else bn_act_bwd_kernel<DT, true, false, G><<<ntiles, 256, 0, s>>>(a);                       \
        } else if (a.dzb != nullptr || a.keep != nullptr) {                                             \
            if (apply) bn_act_bwd_kernel<DT, false, true, G><<<ntiles, 256, 0, s>>>(a);                 \
            else bn_act_bwd_kernel<DT, false, false, G><<<ntiles, 256, 0, s>>>(a);                      \
        } else {                                                                                        \
            if (apply) bn_act_bwd_kernel<DT, false, true, G, 0, false><<<ntiles, 256, 0, s>>>(a);       \
            else bn_act_bwd_kernel<DT, false, false, G, 0, false><<<ntiles, 256, 0, s>>>(a);            \
        }                                                                                               \
    } while (0)
    if (a.head_dl != nullptr) {                            // the head source: plain pixels, slope-family activation (host checks)
#define LAUNCH_HEAD(DT, NC)                                                                             \
    do {                                                                                                \
        if (apply) bn_act_bwd_kernel<DT, false, true, false, NC, false><<<ntiles, 256, 0, s>>>(a);      \
        else bn_act_bwd_kernel<DT, false, false, false, NC, false><<<ntiles, 256, 0, s>>>(a);           \
    } while (0)
        if (dtype == GS_F16) { if (a.head_n <= 2) LAUNCH_HEAD(GS_F16, 2); else LAUNCH_HEAD(GS_F16, 4); }
        else { if (a.head_n <= 2) LAUNCH_HEAD(GS_BF16, 2); else LAUNCH_HEAD(GS_BF16, 4); }
#undef LAUNCH_HEAD
        return 0;
    }
    if (dtype == GS_F16) { if (generic) LAUNCH(GS_F16, true); else LAUNCH(GS_F16, false); }
    else { if (generic) LAUNCH(GS_BF16, true); else LAUNCH(GS_BF16, false); }
#undef LAUNCH
    return 0;
}

// number of tiles the reduce kernel actually writes (<= gs_bn_bwd_tiles)
extern "C" int gs_bn_bwd_tiles_used(int N, int H, int W, int pooled) {
    const int PH = pooled ? (H + 1) / 2 : H, PW = pooled ? (W + 1) / 2 : W;
    const int64_t units = (int64_t)N * PH * PW;
    const int ntiles_max = gs_bn_bwd_tiles(N, H, W);
    int64_t tu = cdiv64(units, ntiles_max);
    if (tu < 1) tu = 1;
    return (int)cdiv64(units, tu);
}

extern "C" int gs_bn_act_bwd_reduce(const void* y, const void* dz_a, int sa, int coff_a, const void* dzp,
                                    const void* dz_b, int act_b, const uint8_t* keep_mask, float keep_scale,
                                    const float* scale, const float* shift, const float* mean, const float* invstd,
                                    int act, float* partials, int N, int H, int W, int C, int dtype, void* stream) {
    GS_CHECK_ARG(!(dzp && (dz_b || keep_mask)), "gs_bn_act_bwd_reduce: pooled gradient cannot be combined with dz_b / keep_mask");
    GS_CHECK_ARG(y && partials && (dz_a || dzp || dz_b) && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && C <= 2048,
                 "gs_bn_act_bwd_reduce: bad arguments");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_bn_act_bwd_reduce: bad dtype");
    GS_CHECK_ARG((int64_t)N * H * W < 2147483647LL, "gs_bn_act_bwd_reduce: more than 2^31 pixels");
    GS_CHECK_ARG(!dz_a || (sa >= coff_a + C && sa % 8 == 0 && coff_a % 8 == 0), "gs_bn_act_bwd_reduce: bad dz stride");
    BwdArgs a{};
    a.y = (const unsigned short*)y; a.dza = (const unsigned short*)dz_a; a.dzp = (const unsigned short*)dzp;
    a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.c1 = nullptr; a.c2 = nullptr;
    a.dzb = (const unsigned short*)dz_b; a.act_b = act_b; a.keep = keep_mask; a.keep_scale = keep_scale;
    a.partials = partials; a.dy = nullptr; a.sa = sa; a.ca = coff_a; a.act = act; a.bn = 1;
    a.N = N; a.H = H; a.W = W; a.C = C;
    launch_bwd(a, false, dtype, (hipStream_t)stream, nullptr);
    GS_CHECK_LAUNCH("gs_bn_act_bwd_reduce");
    return GS_OK;
}

extern "C" int gs_bn_bwd_coeffs(const float* partials, int ntiles, int C, double count, float gscale, float* dgamma,
                                float* dbeta, float* c1, float* c2, void* stream) {
    GS_CHECK_ARG(partials && c1 && c2 && ntiles > 0 && C > 0 && count > 0, "gs_bn_bwd_coeffs: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (ntiles <= direct_tiles()) {
        bn_bwd_coeffs_kernel<float><<<cdiv(C, 32), 256, 0, s>>>(partials, ntiles, C, count, gscale, dgamma, dbeta, c1, c2);
        GS_CHECK_LAUNCH("gs_bn_bwd_coeffs");
        return GS_OK;
    }
    double* o1; int ns;
    reduce_partials(partials, ntiles, C, &o1, &ns, s);
    bn_bwd_coeffs_kernel<double><<<cdiv(C, 32), 256, 0, s>>>(o1, ns, C, count, gscale, dgamma, dbeta, c1, c2);
    GS_CHECK_LAUNCH("gs_bn_bwd_coeffs");
    return GS_OK;
}

extern "C" int gs_bn_act_bwd_apply(const void* y, const void* dz_a, int sa, int coff_a, const void* dzp,
                                   const void* dz_b, int act_b, const uint8_t* keep_mask, float keep_scale,
                                   const float* scale, const float* shift, const float* mean, const float* invstd,
                                   const float* c1, const float* c2, int act, int bn, void* dy, int N, int H, int W,
                                   int C, int dtype, void* stream) {
    GS_CHECK_ARG(!(dzp && (dz_b || keep_mask)), "gs_bn_act_bwd_apply: pooled gradient cannot be combined with dz_b / keep_mask");
    GS_CHECK_ARG(y && dy && (dz_a || dzp || dz_b) && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && C <= 2048,
                 "gs_bn_act_bwd_apply: bad arguments");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_bn_act_bwd_apply: bad dtype");
    GS_CHECK_ARG((int64_t)N * H * W < 2147483647LL, "gs_bn_act_bwd_apply: more than 2^31 pixels");
    GS_CHECK_ARG(!bn || (scale && shift && mean && invstd && c1 && c2), "gs_bn_act_bwd_apply: bn=1 needs all coefficients");
    GS_CHECK_ARG(!dz_a || (sa >= coff_a + C && sa % 8 == 0 && coff_a % 8 == 0), "gs_bn_act_bwd_apply: bad dz stride");
    BwdArgs a{};
    a.y = (const unsigned short*)y; a.dza = (const unsigned short*)dz_a; a.dzp = (const unsigned short*)dzp;
    a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.c1 = c1; a.c2 = c2;
    a.dzb = (const unsigned short*)dz_b; a.act_b = act_b; a.keep = keep_mask; a.keep_scale = keep_scale;
    a.partials = nullptr; a.dy = (unsigned short*)dy; a.sa = sa; a.ca = coff_a; a.act = act; a.bn = bn;
    a.N = N; a.H = H; a.W = W; a.C = C;
    launch_bwd(a, true, dtype, (hipStream_t)stream, nullptr);
    GS_CHECK_LAUNCH("gs_bn_act_bwd_apply");
    return GS_OK;
}

// BatchNorm backward of the stage in front of a pointwise head (unet/unet_parts.py:74 after :19-21): the gradient source is
// the head's logit gradient and weight, not a tensor (see BwdArgs::head_dl).
static int bn_head_args(const char* who, BwdArgs& a, const void* y, const float* dl, const float* w_head, int ncls,
                        const float* scale, const float* shift, const float* mean, const float* invstd, int act, int N, int H,
                        int W, int C, int dtype) {
    GS_CHECK_ARG(y && dl && w_head && ncls >= 1 && ncls <= 4 && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && C <= 2048,
                 "%s: bad arguments", who);
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "%s: bad dtype", who);
    GS_CHECK_ARG((int64_t)N * H * W < 2147483647LL, "%s: more than 2^31 pixels", who);
    GS_CHECK_ARG(act == GS_ACT_NONE || act == GS_ACT_RELU || act == GS_ACT_LEAKY02, "%s: activation %d not supported", who, act);
    GS_CHECK_ARG(scale && shift && mean && invstd, "%s: needs the BatchNorm coefficients", who);
    a.y = (const unsigned short*)y; a.head_dl = dl; a.head_w = w_head; a.head_n = ncls;
    a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.act = act; a.bn = 1;
    a.N = N; a.H = H; a.W = W; a.C = C;
    return GS_OK;
}

extern "C" int gs_bn_act_bwd_reduce_head(const void* y, const float* dl, const float* w_head, int ncls, const float* scale,
                                         const float* shift, const float* mean, const float* invstd, int act, float* partials,
                                         int N, int H, int W, int C, int dtype, void* stream) {
    BwdArgs a{};
    int rc = bn_head_args("gs_bn_act_bwd_reduce_head", a, y, dl, w_head, ncls, scale, shift, mean, invstd, act, N, H, W, C, dtype);
    if (rc) return rc;
    GS_CHECK_ARG(partials != nullptr, "gs_bn_act_bwd_reduce_head: null partials");
    a.partials = partials;
    launch_bwd(a, false, dtype, (hipStream_t)stream, nullptr);
    GS_CHECK_LAUNCH("gs_bn_act_bwd_reduce_head");
    return GS_OK;
}

extern "C" int gs_bn_act_bwd_apply_head(const void* y, const float* dl, const float* w_head, int ncls, const float* scale,
                                        const float* shift, const float* mean, const float* invstd, const float* c1,
                                        const float* c2, int act, void* dy, int N, int H, int W, int C, int dtype, void* stream) {
    BwdArgs a{};
    int rc = bn_head_args("gs_bn_act_bwd_apply_head", a, y, dl, w_head, ncls, scale, shift, mean, invstd, act, N, H, W, C, dtype);
    if (rc) return rc;
    GS_CHECK_ARG(dy && c1 && c2, "gs_bn_act_bwd_apply_head: null pointer");
    a.c1 = c1; a.c2 = c2; a.dy = (unsigned short*)dy;
    launch_bwd(a, true, dtype, (hipStream_t)stream, nullptr);
    GS_CHECK_LAUNCH("gs_bn_act_bwd_apply_head");
    return GS_OK;
}

extern "C" int gs_bn_partials_colsum(const float* partials, int ntiles, int Cfull, int coff, int C, float gscale, float* out,
                                     void* stream) {
    GS_CHECK_ARG(partials && out && ntiles > 0 && Cfull > 0 && C > 0 && coff >= 0 && coff + C <= Cfull,
                 "gs_bn_partials_colsum: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (ntiles <= direct_tiles()) {
        partials_colsum_final<float><<<cdiv(C, 32), 256, 0, s>>>(partials + coff, ntiles, (int64_t)2 * Cfull, C, gscale, out);
    } else {
        int slices = ntiles < RED_SLICES ? ntiles : RED_SLICES;
        const int tps = cdiv(ntiles, slices);
        slices = cdiv(ntiles, tps);
        double* o1 = (double*)((char*)partials + (((size_t)ntiles * 2 * Cfull * sizeof(float) + 15) & ~(size_t)15));   // stage-1 area
        partials_colsum_stage1<<<dim3(cdiv(C, 32), slices), 256, 0, s>>>(partials, ntiles, Cfull, coff, C, tps, o1);
        partials_colsum_final<double><<<cdiv(C, 32), 256, 0, s>>>(o1, slices, (int64_t)C, C, gscale, out);
    }
    GS_CHECK_LAUNCH("gs_bn_partials_colsum");
    return GS_OK;
}

extern "C" int gs_colsum(const void* t, int pix_stride, int coff, int N, int H, int W, int y0, int x0, int h, int w,
                         int C, float gscale, float* ws, float* out, int dtype, void* stream) {
    GS_CHECK_ARG(t && ws && out && N > 0 && h > 0 && w > 0 && y0 >= 0 && x0 >= 0 && y0 + h <= H && x0 + w <= W,
                 "gs_colsum: bad region");
    GS_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 2048 && pix_stride % 8 == 0 && coff % 8 == 0, "gs_colsum: bad channels");
    hipStream_t s = (hipStream_t)stream;
    const int64_t npix = (int64_t)N * h * w;
    int64_t ppb = cdiv64(npix, 1024);
    if (ppb < 32) ppb = 32;
    const int nb = (int)cdiv64(npix, ppb);
    if (dtype == GS_F16)
        colsum_stage1<GS_F16><<<nb, 256, 0, s>>>((const unsigned short*)t, pix_stride, coff, npix, C, ppb, ws, H, W, y0, x0, h, w);
    else if (dtype == GS_BF16)
        colsum_stage1<GS_BF16><<<nb, 256, 0, s>>>((const unsigned short*)t, pix_stride, coff, npix, C, ppb, ws, H, W, y0, x0, h, w);
    else GS_CHECK_ARG(false, "gs_colsum: bad dtype");
    colsum_stage2<<<cdiv(C, 32), 256, 0, s>>>(ws, nb, C, gscale, out);
    GS_CHECK_LAUNCH("gs_colsum");
    return GS_OK;
}
