// Loss heads: fused BCE/CE + global-batch Dice (one pass over the logits), generic dice_loss, and the
// mean-reduced GAN / L1 / BCE losses.  All HBM-bound single-pass reductions: per-thread fp32 partials ->
// wave shuffle -> block -> fixed [<=1024 blocks] slab -> one finalising block in double (deterministic).
// Reference: running_files/train_end2end_jsrt.py:136-138,181-183; util/dice_score.py:5-28;
//            models_pix2pix/networks.py:263-281.
#include "common.hpp"

namespace {

constexpr int LOSS_MAX_BLOCKS = 1024;
constexpr float DICE_EPS = 1e-6f;

__device__ __forceinline__ float softplus_neg_abs(float x) { return log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// writes NS block sums to ws[s*LOSS_MAX_BLOCKS + blockIdx.x]
template <int NS>
__device__ __forceinline__ void block_store_sums(float (&v)[NS], float* ws) {
    __shared__ float red[NS][4];
#pragma unroll
    for (int s = 0; s < NS; ++s) v[s] = wave_sum(v[s]);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) red[s][w] = v[s];
    }
    __syncthreads();
    if (threadIdx.x < NS) ws[threadIdx.x * LOSS_MAX_BLOCKS + blockIdx.x] =
        red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

// sum of ws[s][0..nb) in double, by one block of 256 threads; result broadcast
__device__ __forceinline__ double final_sum(const float* ws, int s, int nb, double* red) {
    double v = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) v += (double)ws[s * LOSS_MAX_BLOCKS + i];
    __syncthreads();
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    return red[0];
}

// ---- segmentation loss ---------------------------------------------------------------------------
// sums: 0 = sum ce/bce, 1 = sum p*t, 2 = sum p, 3 = sum t
__global__ __launch_bounds__(256) void seg_loss_fwd_kernel(const float* __restrict__ x, const uint8_t* __restrict__ m,
                                                           int N, int C, int64_t HW, float* ws) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t total = (int64_t)N * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int t = m[i];
        if (C == 1) {
            const float v = x[i];
            const float tf = (float)t;
            s[0] += fmaxf(v, 0.f) - v * tf + softplus_neg_abs(v);
            const float p = sigmoidf_(v);
            s[1] += p * tf; s[2] += p; s[3] += tf;
        } else {
            const int64_t n = i / HW, hw = i - n * HW;
            const float* xp = x + n * C * HW + hw;
            float mx = -INFINITY;
            for (int c = 0; c < C; ++c) mx = fmaxf(mx, xp[(int64_t)c * HW]);
            float den = 0.f, xt = 0.f;
            for (int c = 0; c < C; ++c) {
                const float v = xp[(int64_t)c * HW];
                den += expf(v - mx);
                if (c == t) xt = v;
            }
            const float lse = mx + logf(den);
            s[0] += lse - xt;
            s[1] += expf(xt - lse);     // p_t = sum_c p_c * onehot_c
            s[2] += 1.f;                // sum_c p_c
            s[3] += 1.f;                // sum_c onehot_c
        }
    }
    block_store_sums<4>(s, ws);
}

__global__ __launch_bounds__(256) void seg_loss_finalize_kernel(const float* ws, int nb, double npix, float* out) {
    __shared__ double red[256];
    const double ce = final_sum(ws, 0, nb, red);
    const double spt = final_sum(ws, 1, nb, red);
    const double sp = final_sum(ws, 2, nb, red);
    const double st = final_sum(ws, 3, nb, red);
    if (threadIdx.x == 0) {
        // dice_score.py:12-17 in fp32 like the reference
        const float inter = 2.f * (float)spt;
        float sets = (float)sp + (float)st;
        if (sets == 0.f) sets = inter;
        const float dice = (inter + DICE_EPS) / (sets + DICE_EPS);
        const float l_ce = (float)(ce / npix);
        out[0] = l_ce + (1.f - dice);
        out[1] = l_ce;
        out[2] = 1.f - dice;
        out[3] = inter;
        out[4] = (float)sp;
        out[5] = (float)st;
        out[6] = 1.f;       // multiplier of the Dice part of the gradient (world size in global-batch Dice mode)
        out[7] = 0.f;
    }
}

__global__ __launch_bounds__(256) void seg_loss_bwd_kernel(const float* __restrict__ x, const uint8_t* __restrict__ m,
                                                           const float* __restrict__ out, const float* gout, float gscale,
                                                           float* __restrict__ dx, int N, int C, int64_t HW) {
    const int64_t total = (int64_t)N * HW;
    const float g = (gout ? gout[0] : 1.f) * gscale;
    const float inter = out[3], sp = out[4], st = out[5];
    const float S = sp + st;
    const bool degenerate = (S == 0.f);               // torch.where(sets_sum == 0, inter, sets_sum): dice == 1, grad 0
    const float dmul = out[6];                         // 1, or the world size when out[3..5] are global sums
    const float den = (S + DICE_EPS), inv2 = dmul / (den * den);
    const float inv_m = 1.f / (float)total;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int t = m[i];
        if (C == 1) {
            const float v = x[i], tf = (float)t;
            const float p = sigmoidf_(v);
            // d(1-dice)/dp = -(2 t (S+eps) - (I+eps)) / (S+eps)^2
            const float gd = degenerate ? 0.f : -(2.f * tf * den - (inter + DICE_EPS)) * inv2;
            dx[i] = g * ((p - tf) * inv_m + gd * p * (1.f - p));
        } else {
            const int64_t n = i / HW, hw = i - n * HW;
            const float* xp = x + n * C * HW + hw;
            float* dp = dx + n * C * HW + hw;
            float mx = -INFINITY;
            for (int c = 0; c < C; ++c) mx = fmaxf(mx, xp[(int64_t)c * HW]);
            float dsum = 0.f;
            for (int c = 0; c < C; ++c) dsum += expf(xp[(int64_t)c * HW] - mx);
            const float rden = 1.f / dsum;
            // g_c = d(1-dice)/dp_c ; softmax backward dx_c = p_c (g_c - sum_k p_k g_k)
            const float g1 = degenerate ? 0.f : -(2.f * den - (inter + DICE_EPS)) * inv2;   // onehot = 1
            const float g0 = degenerate ? 0.f : (inter + DICE_EPS) * inv2;                  // onehot = 0
            float pt = 0.f;
            for (int c = 0; c < C; ++c) if (c == t) pt = expf(xp[(int64_t)c * HW] - mx) * rden;
            const float dot = pt * g1 + (1.f - pt) * g0;
            for (int c = 0; c < C; ++c) {
                const float p = expf(xp[(int64_t)c * HW] - mx) * rden;
                const float oh = (c == t) ? 1.f : 0.f;
                dp[(int64_t)c * HW] = g * ((p - oh) * inv_m + p * ((c == t ? g1 : g0) - dot));
            }
        }
    }
}

// ---- generic dice_loss ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dice_fwd_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                       int64_t n, float* ws) {
    float s[3] = {0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float a = p[i], b = t[i];
        s[0] += a * b; s[1] += a; s[2] += b;
    }
    block_store_sums<3>(s, ws);
}
__global__ __launch_bounds__(256) void dice_finalize_kernel(const float* ws, int nb, float* out) {
    __shared__ double red[256];
    const double spt = final_sum(ws, 0, nb, red);
    const double sp = final_sum(ws, 1, nb, red);
    const double st = final_sum(ws, 2, nb, red);
    if (threadIdx.x == 0) {
        const float inter = 2.f * (float)spt;
        float sets = (float)sp + (float)st;
        if (sets == 0.f) sets = inter;
        out[0] = 1.f - (inter + DICE_EPS) / (sets + DICE_EPS);
        out[1] = inter; out[2] = (float)sp; out[3] = (float)st;
    }
}
__global__ __launch_bounds__(256) void dice_bwd_kernel(const float* __restrict__ t, const float* __restrict__ out,
                                                       const float* gout, float* __restrict__ dp, int64_t n) {
    const float g = gout ? gout[0] : 1.f;
    const float inter = out[1], S = out[2] + out[3];
    const bool degenerate = (S == 0.f);
    const float den = S + DICE_EPS, inv2 = 1.f / (den * den);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dp[i] = degenerate ? 0.f : -g * (2.f * t[i] * den - (inter + DICE_EPS)) * inv2;
}

// ---- mean-reduced elementwise losses -------------------------------------------------------------
__device__ __forceinline__ float mean_loss_term(float x, float t, float cval, int mode) {
    switch (mode) {
        case 0: return fmaxf(x, 0.f) - x * cval + softplus_neg_abs(x);
        case 1: return (x - cval) * (x - cval);
        case 2: return x * cval;
        case 3: return fabsf(x - t);
        default: return fmaxf(x, 0.f) - x * t + softplus_neg_abs(x);
    }
}
__device__ __forceinline__ float mean_loss_grad(float x, float t, float cval, int mode) {
    switch (mode) {
        case 0: return sigmoidf_(x) - cval;
        case 1: return 2.f * (x - cval);
        case 2: return cval;
        case 3: return x > t ? 1.f : (x < t ? -1.f : 0.f);     // torch sign(): 0 at equality
        default: return sigmoidf_(x) - t;
    }
}
__global__ __launch_bounds__(256) void mean_loss_fwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                            float cval, int mode, int64_t n, float* ws) {
    float s[1] = {0.f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        s[0] += mean_loss_term(x[i], t ? t[i] : 0.f, cval, mode);
    block_store_sums<1>(s, ws);
}
__global__ __launch_bounds__(256) void mean_loss_finalize_kernel(const float* ws, int nb, double n, float* out) {
    __shared__ double red[256];
    const double s = final_sum(ws, 0, nb, red);
    if (threadIdx.x == 0) out[0] = (float)(s / n);
}
__global__ __launch_bounds__(256) void mean_loss_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                            float cval, int mode, int64_t n, const float* gout,
                                                            float gscale, float* __restrict__ dx) {
    const float g = (gout ? gout[0] : 1.f) * gscale / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dx[i] = g * mean_loss_grad(x[i], t ? t[i] : 0.f, cval, mode);
}

inline int loss_blocks(int64_t n) {
    int64_t b = cdiv64(n, 256 * 4);
    if (b < 1) b = 1;
    if (b > LOSS_MAX_BLOCKS) b = LOSS_MAX_BLOCKS;
    return (int)b;
}

}  // namespace

extern "C" int gs_seg_loss_fwd(const float* logits, const uint8_t* mask, int N, int C, int H, int W, float* ws,
                               float* out, void* stream) {
    GS_CHECK_ARG(logits && mask && ws && out && N > 0 && C > 0 && C <= 64 && H > 0 && W > 0, "gs_seg_loss_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t npix = (int64_t)N * H * W;
    const int nb = loss_blocks(npix);
    seg_loss_fwd_kernel<<<nb, 256, 0, s>>>(logits, mask, N, C, (int64_t)H * W, ws);
    seg_loss_finalize_kernel<<<1, 256, 0, s>>>(ws, nb, (double)npix, out);
    GS_CHECK_LAUNCH("gs_seg_loss_fwd");
    return GS_OK;
}

extern "C" int gs_seg_loss_bwd(const float* logits, const uint8_t* mask, const float* out, const float* gout,
                               float gscale, float* dlogits, int N, int C, int H, int W, void* stream) {
    GS_CHECK_ARG(logits && mask && out && dlogits && N > 0 && C > 0 && H > 0 && W > 0, "gs_seg_loss_bwd: bad arguments");
    const int64_t npix = (int64_t)N * H * W;
    int64_t nb = cdiv64(npix, 256);
    if (nb > 4096) nb = 4096;
    seg_loss_bwd_kernel<<<(int)nb, 256, 0, (hipStream_t)stream>>>(logits, mask, out, gout, gscale, dlogits, N, C,
                                                                 (int64_t)H * W);
    GS_CHECK_LAUNCH("gs_seg_loss_bwd");
    return GS_OK;
}

// ---- per-item Dice coefficients in one launch (dice_score.py:5-17 with reduce_batch_first=False; unet/evaluate.py:29-43) --
// item b has n_per consecutive elements; grid (DB_BLOCKS, B); ws[b][blk][3] partial sums; the finalising block turns them
// into dice_b = (2 sum(pt) + eps) / (sum p + sum t + eps) (sets == 0 -> inter, dice_score.py:14) and their mean.
constexpr int DB_BLOCKS = 16;

__global__ __launch_bounds__(256) void dice_batched_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                           int64_t n_per, float* __restrict__ ws) {
    __shared__ float red[3][4];
    const float* pp = p + (int64_t)blockIdx.y * n_per;
    const float* tt = t + (int64_t)blockIdx.y * n_per;
    float s[3] = {0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_per; i += (int64_t)gridDim.x * 256) {
        const float a = pp[i], b = tt[i];
        s[0] += a * b; s[1] += a; s[2] += b;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) s[k] = wave_sum(s[k]);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s[0]; red[1][threadIdx.x >> 6] = s[1]; red[2][threadIdx.x >> 6] = s[2]; }
    __syncthreads();
    if (threadIdx.x < 3)
        ws[((int64_t)blockIdx.y * DB_BLOCKS + blockIdx.x) * 3 + threadIdx.x] =
            red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

// fused evaluation head: prediction = sigmoid(logit) > 0.5 (C == 1) or arg-max over classes (first maximum), one item per
// (sample, foreground class); sums are pixel counts (exact in fp32 per block)
__global__ __launch_bounds__(256) void eval_dice_kernel(const float* __restrict__ x, const uint8_t* __restrict__ m, int C,
                                                        int64_t HW, float* __restrict__ ws) {
    __shared__ float red[9][4];
    const int K = C == 1 ? 1 : C - 1;
    const float* xp = x + (int64_t)blockIdx.y * C * HW;
    const uint8_t* mp = m + (int64_t)blockIdx.y * HW;
    float s[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};       // [k][inter, pred, true], K <= 3
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += (int64_t)gridDim.x * 256) {
        const int t = mp[i];
        int pred;
        if (C == 1) {
            pred = sigmoidf_(xp[i]) > 0.5f ? 1 : 0;
        } else {
            float best = xp[i];
            pred = 0;
            for (int c = 1; c < C; ++c) {
                const float v = xp[(int64_t)c * HW + i];
                if (v > best) { best = v; pred = c; }
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k < K) {
                const float pk = pred == k + 1 ? 1.f : 0.f, tk = t == k + 1 ? 1.f : 0.f;
                s[3 * k] += pk * tk; s[3 * k + 1] += pk; s[3 * k + 2] += tk;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) s[k] = wave_sum(s[k]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) red[k][threadIdx.x >> 6] = s[k];
    }
    __syncthreads();
    if (threadIdx.x < 3 * K) {
        const int k = threadIdx.x / 3, q = threadIdx.x - 3 * k;
        ws[(((int64_t)blockIdx.y * K + k) * DB_BLOCKS + blockIdx.x) * 3 + q] =
            red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    }
}

// out[0] = mean of the B coefficients, out[1 + b] = dice_b
// metric 0: Dice (dice_score.py:12-16); 1: Jaccard index with smooth = 1 (train_end2end_isic.py:40-53)
__global__ __launch_bounds__(256) void dice_batched_finalize_kernel(const float* __restrict__ ws, int B, int nblk,
                                                                    float* __restrict__ out, int metric = 0) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) {
        double spt = 0.0, sp = 0.0, st = 0.0;
        for (int j = 0; j < nblk; ++j) {
            const float* q = ws + ((int64_t)b * DB_BLOCKS + j) * 3;
            spt += (double)q[0]; sp += (double)q[1]; st += (double)q[2];
        }
        float d;
        if (metric == 1) {
            const float I = (float)spt, S = (float)sp + (float)st;
            d = (I + 1.f) / (S - I + 1.f);
        } else {
            const float inter = 2.f * (float)spt;
            float sets = (float)sp + (float)st;
            if (sets == 0.f) sets = inter;
            d = (inter + DICE_EPS) / (sets + DICE_EPS);
        }
        out[1 + b] = d;
        acc += (double)d;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] / (double)B);
}

static int dice_batched_blocks(int64_t n_per) {
    int64_t nb = cdiv64(n_per, 256 * 8);
    return (int)(nb < 1 ? 1 : (nb > DB_BLOCKS ? DB_BLOCKS : nb));
}

// ---- BCE + Jaccard loss of the ISIC variant (train_end2end_isic.py:40-56,247-249; one class) -----------------------------
// per sample i: I_i = sum p t, S_i = sum (p + t), jac_i = (I_i + 1) / (S_i - I_i + 1); loss = mean BCE + 1 - mean_i jac_i.
// fwd: grid (DB_BLOCKS, N) partial sums [bce, I, S]; finalize: out[0] = loss, out[1] = bce, out[2] = 1 - mean jac,
// out[4 + 2 i] = I_i, out[5 + 2 i] = S_i (kept for the backward pass).
__global__ __launch_bounds__(256) void jaccard_fwd_kernel(const float* __restrict__ x, const uint8_t* __restrict__ m,
                                                          int64_t HW, float* __restrict__ ws) {
    __shared__ float red[3][4];
    const float* xp = x + (int64_t)blockIdx.y * HW;
    const uint8_t* mp = m + (int64_t)blockIdx.y * HW;
    float s[3] = {0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += (int64_t)gridDim.x * 256) {
        const float v = xp[i], tf = (float)mp[i];
        s[0] += fmaxf(v, 0.f) - v * tf + softplus_neg_abs(v);
        const float p = sigmoidf_(v);
        s[1] += p * tf; s[2] += p + tf;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) s[k] = wave_sum(s[k]);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s[0]; red[1][threadIdx.x >> 6] = s[1]; red[2][threadIdx.x >> 6] = s[2]; }
    __syncthreads();
    if (threadIdx.x < 3)
        ws[((int64_t)blockIdx.y * DB_BLOCKS + blockIdx.x) * 3 + threadIdx.x] =
            red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

__global__ __launch_bounds__(256) void jaccard_finalize_kernel(const float* __restrict__ ws, int N, int nblk, int64_t HW,
                                                               float* __restrict__ out) {
    __shared__ double red[2][256];
    double bce = 0.0, jac = 0.0;
    for (int b = threadIdx.x; b < N; b += 256) {
        double sb = 0.0, si = 0.0, ss = 0.0;
        for (int j = 0; j < nblk; ++j) {
            const float* q = ws + ((int64_t)b * DB_BLOCKS + j) * 3;
            sb += (double)q[0]; si += (double)q[1]; ss += (double)q[2];
        }
        const float I = (float)si, S = (float)ss;
        out[4 + 2 * b] = I; out[5 + 2 * b] = S;
        bce += sb;
        jac += (double)((I + 1.f) / (S - I + 1.f));
    }
    red[0][threadIdx.x] = bce; red[1][threadIdx.x] = jac;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float l_bce = (float)(red[0][0] / ((double)N * (double)HW));
        const float l_jac = 1.f - (float)(red[1][0] / (double)N);
        out[0] = l_bce + l_jac; out[1] = l_bce; out[2] = l_jac; out[3] = 0.f;
    }
}

// dx = g * [ (p - t) / (N HW)  -  p (1 - p) / N * ( t (D) - (I + 1)(1 - t) ) / D^2 ],  D = S - I + 1
__global__ __launch_bounds__(256) void jaccard_bwd_kernel(const float* __restrict__ x, const uint8_t* __restrict__ m,
                                                          const float* __restrict__ out, const float* gout, float gscale,
                                                          float* __restrict__ dx, int N, int64_t HW) {
    const float g = (gout ? gout[0] : 1.f) * gscale;
    const float inv_m = 1.f / ((float)N * (float)HW), inv_n = 1.f / (float)N;
    const int n = blockIdx.y;
    const float I = out[4 + 2 * n], S = out[5 + 2 * n];
    const float D = S - I + 1.f, inv_d2 = 1.f / (D * D);
    const float* xp = x + (int64_t)n * HW;
    const uint8_t* mp = m + (int64_t)n * HW;
    float* dp = dx + (int64_t)n * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += (int64_t)gridDim.x * 256) {
        const float tf = (float)mp[i];
        const float p = sigmoidf_(xp[i]);
        const float djac = (tf * D - (I + 1.f) * (1.f - tf)) * inv_d2;        // d jac_n / d p
        dp[i] = g * ((p - tf) * inv_m - p * (1.f - p) * inv_n * djac);
    }
}

extern "C" int64_t gs_jaccard_loss_out_floats(int N) { return N > 0 ? 4 + 2 * (int64_t)N : 0; }

extern "C" int gs_jaccard_seg_loss_fwd(const float* logits, const uint8_t* mask, int N, int64_t HW, float* ws, float* out,
                                       void* stream) {
    GS_CHECK_ARG(logits && mask && ws && out && N > 0 && N <= 65535 && HW > 0, "gs_jaccard_seg_loss_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int nb = dice_batched_blocks(HW);
    jaccard_fwd_kernel<<<dim3(nb, N), 256, 0, s>>>(logits, mask, HW, ws);
    jaccard_finalize_kernel<<<1, 256, 0, s>>>(ws, N, nb, HW, out);
    GS_CHECK_LAUNCH("gs_jaccard_seg_loss_fwd");
    return GS_OK;
}

extern "C" int gs_jaccard_seg_loss_bwd(const float* logits, const uint8_t* mask, const float* out, const float* gout,
                                       float gscale, float* dlogits, int N, int64_t HW, void* stream) {
    GS_CHECK_ARG(logits && mask && out && dlogits && N > 0 && N <= 65535 && HW > 0, "gs_jaccard_seg_loss_bwd: bad arguments");
    int64_t nb = cdiv64(HW, 256 * 4);
    if (nb > 256) nb = 256;
    if (nb < 1) nb = 1;
    jaccard_bwd_kernel<<<dim3((int)nb, N), 256, 0, (hipStream_t)stream>>>(logits, mask, out, gout, gscale, dlogits, N, HW);
    GS_CHECK_LAUNCH("gs_jaccard_seg_loss_bwd");
    return GS_OK;
}

extern "C" int64_t gs_dice_batched_ws_floats(int B) { return B > 0 ? (int64_t)B * DB_BLOCKS * 3 : 0; }

extern "C" int gs_dice_coeff_batched(const float* p, const float* t, int B, int64_t n_per, float* ws, float* out,
                                     void* stream) {
    GS_CHECK_ARG(p && t && ws && out && B > 0 && B <= 65535 && n_per > 0, "gs_dice_coeff_batched: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int nb = dice_batched_blocks(n_per);
    dice_batched_kernel<<<dim3(nb, B), 256, 0, s>>>(p, t, n_per, ws);
    dice_batched_finalize_kernel<<<1, 256, 0, s>>>(ws, B, nb, out);
    GS_CHECK_LAUNCH("gs_dice_coeff_batched");
    return GS_OK;
}

static int eval_metric(const float* logits, const uint8_t* mask, int N, int C, int64_t HW, int metric, float* ws, float* out,
                       void* stream, const char* who) {
    GS_CHECK_ARG(logits && mask && ws && out && N > 0 && N <= 65535 && C >= 1 && C <= 4 && HW > 0, "%s: bad arguments (C <= 4)", who);
    hipStream_t s = (hipStream_t)stream;
    const int K = C == 1 ? 1 : C - 1;
    const int nb = dice_batched_blocks(HW);
    eval_dice_kernel<<<dim3(nb, N), 256, 0, s>>>(logits, mask, C, HW, ws);
    dice_batched_finalize_kernel<<<1, 256, 0, s>>>(ws, N * K, nb, out, metric);
    GS_CHECK_LAUNCH(who);
    return GS_OK;
}

extern "C" int gs_eval_dice(const float* logits, const uint8_t* mask, int N, int C, int64_t HW, float* ws, float* out,
                            void* stream) {
    return eval_metric(logits, mask, N, C, HW, 0, ws, out, stream, "gs_eval_dice");
}

// the ISIC script's validation metric (train_end2end_isic.py:58-84): thresholded prediction, per-sample Jaccard index, mean
extern "C" int gs_eval_jaccard(const float* logits, const uint8_t* mask, int N, int64_t HW, float* ws, float* out, void* stream) {
    return eval_metric(logits, mask, N, 1, HW, 1, ws, out, stream, "gs_eval_jaccard");
}

extern "C" int gs_dice_loss_fwd(const float* p, const float* t, int64_t n, float* ws, float* out, void* stream) {
    GS_CHECK_ARG(p && t && ws && out && n > 0, "gs_dice_loss_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int nb = loss_blocks(n);
    dice_fwd_kernel<<<nb, 256, 0, s>>>(p, t, n, ws);
    dice_finalize_kernel<<<1, 256, 0, s>>>(ws, nb, out);
    GS_CHECK_LAUNCH("gs_dice_loss_fwd");
    return GS_OK;
}

extern "C" int gs_dice_loss_bwd(const float* t, const float* out, const float* gout, float* dp, int64_t n,
                                void* stream) {
    GS_CHECK_ARG(t && out && dp && n > 0, "gs_dice_loss_bwd: bad arguments");
    int64_t nb = cdiv64(n, 256);
    if (nb > 4096) nb = 4096;
    dice_bwd_kernel<<<(int)nb, 256, 0, (hipStream_t)stream>>>(t, out, gout, dp, n);
    GS_CHECK_LAUNCH("gs_dice_loss_bwd");
    return GS_OK;
}

extern "C" int gs_mean_loss_fwd(const float* x, const float* t, float cval, int mode, int64_t n, float* ws,
                                float* out, void* stream) {
    GS_CHECK_ARG(x && ws && out && n > 0 && mode >= 0 && mode <= 4, "gs_mean_loss_fwd: bad arguments");
    GS_CHECK_ARG(mode < 3 || t, "gs_mean_loss_fwd: mode %d needs a target tensor", mode);
    hipStream_t s = (hipStream_t)stream;
    const int nb = loss_blocks(n);
    mean_loss_fwd_kernel<<<nb, 256, 0, s>>>(x, t, cval, mode, n, ws);
    mean_loss_finalize_kernel<<<1, 256, 0, s>>>(ws, nb, (double)n, out);
    GS_CHECK_LAUNCH("gs_mean_loss_fwd");
    return GS_OK;
}

extern "C" int gs_mean_loss_bwd(const float* x, const float* t, float cval, int mode, int64_t n, const float* gout,
                                float gscale, float* dx, void* stream) {
    GS_CHECK_ARG(x && dx && n > 0 && mode >= 0 && mode <= 4, "gs_mean_loss_bwd: bad arguments");
    GS_CHECK_ARG(mode < 3 || t, "gs_mean_loss_bwd: mode %d needs a target tensor", mode);
    int64_t nb = cdiv64(n, 256);
    if (nb > 4096) nb = 4096;
    mean_loss_bwd_kernel<<<(int)nb, 256, 0, (hipStream_t)stream>>>(x, t, cval, mode, n, gout, gscale, dx);
    GS_CHECK_LAUNCH("gs_mean_loss_bwd");
    return GS_OK;
}
