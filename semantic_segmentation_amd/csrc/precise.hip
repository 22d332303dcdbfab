// Precise mode of the U-Net forward: activations and weights travel as PAIRS of 16-bit values
// v = hi + lo (hi = 16-bit(v), lo = 16-bit(v - hi): ~22 significand bits for fp16), the MFMA contractions run over the K
// concatenation [x_hi | x_lo | x_hi] . [w_hi | w_hi | w_lo] (every product exact, one fp32 accumulator) -- conv3x3.hip /
// igemm.hip carry those; this file holds the memory-bound kernels around them:
//   gs_pack_weight_split      fp32 conv / transposed-conv weight -> [taps][Cout][3*Cin] segment pack
//   gs_conv_smallcin_fwd_split  the 1..4-channel first conv (unet_parts.py:16), fp32 image and weights -> y pair
//   gs_bn_act_apply_split     z = act(bn(y_hi + y_lo)) -> z pair (+ 2x2 max-pooled pair), unet_parts.py:17-18,20-21,34
//   gs_head1x1_fwd_split      OutConv (unet_parts.py:74) reading the z pair
// Opt-in (UNet(..., precise=True)); the default engine stores single 16-bit values.
#include "common.hpp"

namespace {

// ---- weight pack ------------------------------------------------------------------------------------------------
// conv (transposed == 0): w fp32 [Cout][Cin][taps]  -> out[t][co][s*Cin + ci]
// convT (transposed == 1): w fp32 [Cin][Cout][taps] -> out[t][co][s*Cin + ci]
// segment s: 0, 1 -> hi(w), 2 -> lo(w)
template <int DT>
__global__ __launch_bounds__(256) void pack_split_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
                                                         int Cout, int Cin, int taps, int transposed) {
    const int64_t total = (int64_t)taps * Cout * Cin;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ci = (int)(i % Cin);
        const int64_t r = i / Cin;
        const int co = (int)(r % Cout), t = (int)(r / Cout);
        const float v = transposed ? w[((int64_t)ci * Cout + co) * taps + t] : w[((int64_t)co * Cin + ci) * taps + t];
        const unsigned short hi = Elem<DT>::from_f(v);
        const unsigned short lo = Elem<DT>::from_f(v - Elem<DT>::to_f(hi));
        unsigned short* o = out + ((int64_t)t * Cout + co) * 3 * Cin + ci;
        o[0] = hi; o[Cin] = hi; o[2 * Cin] = lo;
    }
}


// ---- generic segment pack (mixed precision plans of the U-Net forward) ---------------------------------------------
// out[t][co][k], k running over the concatenation of up to GS_SEG_MAX segments; segment j covers input channels
// [ci0[j], ci0[j] + len[j]) and carries hi(w) (kind 0), lo(w) = 16-bit(w - hi(w)) (kind 1) or zeros (kind 2: pads K to the
// multiple of 64 the conv kernels walk, e.g. the 32-channel second conv of UNet3D).  One launch packs every descriptor (after an
// optimiser step all packs are stale).
constexpr int SEGPACK_MAX = 24;
struct SegPackArgs {
    const float* w[SEGPACK_MAX];
    unsigned short* out[SEGPACK_MAX];
    int Cout[SEGPACK_MAX], Cin[SEGPACK_MAX], taps[SEGPACK_MAX], transposed[SEGPACK_MAX], K[SEGPACK_MAX];
    int nseg[SEGPACK_MAX], kind[SEGPACK_MAX][GS_SEG_MAX], ci0[SEGPACK_MAX][GS_SEG_MAX], len[SEGPACK_MAX][GS_SEG_MAX];
    int first[SEGPACK_MAX + 1];          // first block of each descriptor
    int n;
};

template <int DT>
__global__ __launch_bounds__(256) void pack_segs_kernel(const SegPackArgs a) {
    int d = 0;
    while (d + 1 < a.n && (int)blockIdx.x >= a.first[d + 1]) ++d;
    const int nb = a.first[d + 1] - a.first[d], b = blockIdx.x - a.first[d];
    const int Cout = a.Cout[d], Cin = a.Cin[d], taps = a.taps[d], K = a.K[d];
    const float* __restrict__ w = a.w[d];
    unsigned short* __restrict__ out = a.out[d];
    // one thread per (co, ci) pair of the SOURCE, all taps: the fp32 weights are read once, each tap written per segment
    const int64_t pairs = (int64_t)Cout * Cin;
    for (int64_t i = (int64_t)b * 256 + threadIdx.x; i < pairs; i += (int64_t)nb * 256) {
        const int ci = (int)(i % Cin), co = (int)(i / Cin);
        const float* src = a.transposed[d] ? w + ((int64_t)ci * Cout + co) * taps : w + ((int64_t)co * Cin + ci) * taps;
        int koff = 0;
        for (int j = 0; j < a.nseg[d]; ++j) {
            const int c0 = a.ci0[d][j], ln = a.len[d][j];
            if (ci >= c0 && ci < c0 + ln) {
                const int k = koff + ci - c0;
                for (int t = 0; t < taps; ++t) {
                    const float v = src[t];
                    const unsigned short hi = Elem<DT>::from_f(v);
                    const int kd = a.kind[d][j];
                    out[((int64_t)t * Cout + co) * K + k] = kd == 0 ? hi : (kd == 1 ? Elem<DT>::from_f(v - Elem<DT>::to_f(hi)) : (unsigned short)0);
                }
            }
            koff += ln;
        }
    }
}

// ---- weight packs of "q" stages (FP8 correction segment, common.hpp) ------------------------------------------------------
// out[t][co] = one row of 4*Cin bytes: [w_hi 16-bit (Cin) | per 32 channels a 64-byte chunk: w_hi8 (32 x e4m3) | w_lo8 (32 x e4m3)].
// Both e4m3 planes of a cout row share ONE power-of-two scale derived from the row's own amax (all taps): w_hi8 = e4m3(hi * 2^e),
// w_lo8 = e4m3(lo * 2^(e + LS)) with amax * 2^e in [64, 128); e goes to wexp[co] (the conv kernel turns it into the E8M0 scale bytes
// of its B operand: per-lane = per cout).  One wave per cout row: amax pass + pack pass over Cin * taps fp32 values.
constexpr int Q8PACK_MAX = 24;
struct Q8PackArgs {
    const float* w[Q8PACK_MAX];
    unsigned char* out[Q8PACK_MAX];
    int* wexp[Q8PACK_MAX];
    int Cout[Q8PACK_MAX], Cin[Q8PACK_MAX], taps[Q8PACK_MAX];
    int first[Q8PACK_MAX + 1];           // first block of each descriptor (4 rows per block)
    int n;
};

template <int DT>
__global__ __launch_bounds__(256) void pack_q8_kernel(const Q8PackArgs a) {
    int d = 0;
    while (d + 1 < a.n && (int)blockIdx.x >= a.first[d + 1]) ++d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int co = (blockIdx.x - a.first[d]) * 4 + wave;
    const int Cout = a.Cout[d], Cin = a.Cin[d], taps = a.taps[d];
    if (co >= Cout) return;
    const float* __restrict__ src = a.w[d] + (int64_t)co * Cin * taps;
    const int n = Cin * taps;
    float amax = 0.f;
    for (int i = lane; i < n; i += 64) amax = fmaxf(amax, fabsf(src[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    int e = 0;
    if (amax > 0.f && amax < INFINITY) {
        int ex;
        frexpf(amax, &ex);                              // amax = m * 2^ex, m in [0.5, 1)
        e = 7 - ex;                                      // amax * 2^e in [64, 128)
    }
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    if (lane == 0) a.wexp[d][co] = e;
    const float s_hi = __builtin_ldexpf(1.f, e), s_lo = __builtin_ldexpf(1.f, e + Q8Shift<DT>::v);
    const int64_t rowb = (int64_t)4 * Cin;
    for (int ci = lane; ci < Cin; ci += 64) {
        const int qoff = 2 * Cin + q8_off(ci);
        for (int t = 0; t < taps; ++t) {
            const float v = src[ci * taps + t];
            const unsigned short hi = Elem<DT>::from_f(v);
            const float hf = Elem<DT>::to_f(hi);
            unsigned char* row = a.out[d] + ((int64_t)t * Cout + co) * rowb;
            reinterpret_cast<unsigned short*>(row)[ci] = hi;
            row[qoff] = (unsigned char)(cvt4_e4m3(hf * s_hi, 0.f, 0.f, 0.f) & 0xffu);
            row[qoff + 32] = (unsigned char)(cvt4_e4m3((v - hf) * s_lo, 0.f, 0.f, 0.f) & 0xffu);
        }
    }
}

// ---- q plane of a tensor that was written hi-only (the transposed conv's half of a concat buffer): hi8 from the stored hi plane,
// lo8 = 0 (the stage's x_lo . w_hi term then covers the other channels only -- exactly what the 16-bit "lo_len" segment did)
template <int DT>
__global__ __launch_bounds__(256) void q8_from_hi_kernel(const unsigned short* __restrict__ x, unsigned char* __restrict__ q, int64_t P,
                                                         int C, int xs, int xc) {
    const int nch = C >> 3;
    const float s_hi = __builtin_ldexpf(1.f, GS_Q8_XH_EXP);
    const int64_t total = P * nch;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const int64_t p = i / nch;
        const int c = xc + ch * 8;
        float h[8];
        unpack8<DT>(*reinterpret_cast<const uint4*>(x + p * xs + c), h);
        uint2 hi8;
        hi8.x = cvt4_e4m3(h[0] * s_hi, h[1] * s_hi, h[2] * s_hi, h[3] * s_hi);
        hi8.y = cvt4_e4m3(h[4] * s_hi, h[5] * s_hi, h[6] * s_hi, h[7] * s_hi);
        unsigned char* dst = q + p * xs * 2 + q8_off(c);
        *reinterpret_cast<uint2*>(dst) = make_uint2(0u, 0u);
        *reinterpret_cast<uint2*>(dst + 32) = hi8;
    }
}

// ---- first conv: fp32 NCHW image, fp32 weights, VALU; one thread per (pixel, 8-channel chunk) ----------------------
constexpr int SCP_TILE = 1024;      // output pixels per block = per BatchNorm tile (== gs_conv_smallcin_mtiles)
constexpr int SCP_MAX_W = 8192;

struct SCPArgs {
    const float* x; const float* w; unsigned short* y_hi; unsigned short* y_lo; float* bnp;
    int N, Cin, H, W, Cout, k, pad;
};

template <int DT>
__global__ __launch_bounds__(256) void smallcin_split_kernel(const SCPArgs a) {
    __shared__ float wl[SCP_MAX_W];            // [tap][Cout]
    __shared__ float red[2][256][8];
    const int T = a.Cin * a.k * a.k;
    for (int i = threadIdx.x; i < T * a.Cout; i += 256) {
        const int co = i % a.Cout, tap = i / a.Cout;
        wl[i] = a.w[(int64_t)co * T + tap];
    }
    __syncthreads();
    const int nch = a.Cout >> 3;
    const int lanes = 256 / nch;
    const int ch = threadIdx.x % nch, pl = threadIdx.x / nch;
    const int64_t M = (int64_t)a.N * a.H * a.W;
    const int64_t m0 = (int64_t)blockIdx.x * SCP_TILE;
    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
    if (pl < lanes) {
        for (int p = pl; p < SCP_TILE; p += lanes) {
            const int64_t m = m0 + p;
            if (m >= M) break;
            const int ox = (int)(m % a.W);
            const int64_t r = m / a.W;
            const int oy = (int)(r % a.H);
            const int n = (int)(r / a.H);
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = 0.f;
            int tap = 0;
            for (int ci = 0; ci < a.Cin; ++ci)
                for (int ky = 0; ky < a.k; ++ky)
                    for (int kx = 0; kx < a.k; ++kx, ++tap) {
                        const int iy = oy - a.pad + ky, ix = ox - a.pad + kx;
                        float xv = 0.f;
                        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                            xv = a.x[(((int64_t)n * a.Cin + ci) * a.H + iy) * a.W + ix];
                        const float* wp = wl + tap * a.Cout + ch * 8;
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc[i] += xv * wp[i];
                    }
#pragma unroll
            for (int i = 0; i < 8; ++i) { s1[i] += acc[i]; s2[i] += acc[i] * acc[i]; }
            uint4 hi, lo;
            split8<DT>(acc, hi, lo);
            *reinterpret_cast<uint4*>(a.y_hi + m * a.Cout + ch * 8) = hi;
            *reinterpret_cast<uint4*>(a.y_lo + m * a.Cout + ch * 8) = lo;
        }
    }
    if (a.bnp) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { red[0][threadIdx.x][i] = s1[i]; red[1][threadIdx.x][i] = s2[i]; }
        __syncthreads();
        if (pl == 0) {
            float t1[8], t2[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { t1[i] = 0.f; t2[i] = 0.f; }
            for (int q = 0; q < lanes; ++q)
#pragma unroll
                for (int i = 0; i < 8; ++i) { t1[i] += red[0][q * nch + ch][i]; t2[i] += red[1][q * nch + ch][i]; }
            float* dst = a.bnp + (int64_t)blockIdx.x * 2 * a.Cout + ch * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) { dst[i] = t1[i]; dst[a.Cout + i] = t2[i]; }
        }
    }
}

// Cout == 32 (the first conv of UNet3D, GenSeg-3D/UNet3D/unet3d.py:28, run as a 2-D conv over three depth slices): ONE THREAD PER OUTPUT
// PIXEL holding all 32 channels -- the [tap][32] weights come from LDS at wave-uniform addresses (8 broadcast ds_read_b128 + 32 FMAs per
// tap) and the image value and its bounds test are formed once per tap, not once per (tap, 8-channel chunk) as in the generic
// kernel above (334 us for the 128^3 volume, VALU-bound on index arithmetic).  The wave's 64 pixel rows (64 B of hi, 64 B of lo
// each) are assembled in LDS (16-byte slot s of row r at s ^ ((r >> 2) & 3): conflict-free both ways) and leave as 1 KB stores of
// complete lines.  BatchNorm partial sums through a padded LDS transpose, one row per SCP_TILE pixels as above.
constexpr int SC32_PAD = 33, SC32_MAX_T = 64;
template <int DT>
__global__ __launch_bounds__(256) void smallcin_split32_kernel(const SCPArgs a) {
    __shared__ float wl[SC32_MAX_T * 32];                       // [tap][32]
    __shared__ __attribute__((aligned(16))) float stage[256 * SC32_PAD];     // output staging (4 waves x 8 KB), then the statistics transpose
    const int T = a.Cin * a.k * a.k;
    for (int i = threadIdx.x; i < T * 32; i += 256) {
        const int co = i & 31, tap = i >> 5;
        wl[i] = a.w[co * T + tap];
    }
    __syncthreads();
    const int M = a.N * a.H * a.W;                               // host guarantees < 2^31
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* const wst = reinterpret_cast<unsigned char*>(stage) + wave * 8192;
    for (int sub = 0; sub < SCP_TILE / 256; ++sub) {
        const int m = blockIdx.x * SCP_TILE + sub * 256 + threadIdx.x;
        const bool live = m < M;
        const int mm = live ? m : M - 1;
        const int ox = mm % a.W;
        const int r = mm / a.W;
        const int oy = r % a.H;
        const int n = r / a.H;
        float acc[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) acc[c] = 0.f;
        const float* xn = a.x + (int64_t)n * a.Cin * a.H * a.W;
        int tap = 0;
        for (int ci = 0; ci < a.Cin; ++ci)
            for (int ky = 0; ky < a.k; ++ky) {
                const int iy = oy - a.pad + ky;
                for (int kx = 0; kx < a.k; ++kx, ++tap) {
                    const int ix = ox - a.pad + kx;
                    float xv = 0.f;
                    if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) xv = xn[(ci * a.H + iy) * a.W + ix];
                    const float4* wp = reinterpret_cast<const float4*>(wl + tap * 32);
#pragma unroll
                    for (int c4 = 0; c4 < 8; ++c4) {
                        const float4 w4 = wp[c4];
                        acc[4 * c4 + 0] += xv * w4.x;
                        acc[4 * c4 + 1] += xv * w4.y;
                        acc[4 * c4 + 2] += xv * w4.z;
                        acc[4 * c4 + 3] += xv * w4.w;
                    }
                }
            }
        // the pair rows of the wave's 64 pixels -> LDS -> complete lines
        {
            const int sw = (lane >> 2) & 3;
#pragma unroll
            for (int c8 = 0; c8 < 4; ++c8) {
                uint4 hi, lo;
                float v8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v8[i] = acc[c8 * 8 + i];
                split8<DT>(v8, hi, lo);
                *reinterpret_cast<uint4*>(wst + lane * 64 + ((c8 ^ sw) << 4)) = hi;
                *reinterpret_cast<uint4*>(wst + 4096 + lane * 64 + ((c8 ^ sw) << 4)) = lo;
            }
            __builtin_amdgcn_wave_barrier();
            const int m_w = m - lane;                                   // the wave's first pixel
            unsigned char* gh = reinterpret_cast<unsigned char*>(a.y_hi) + (int64_t)m_w * 64;
            unsigned char* gl = reinterpret_cast<unsigned char*>(a.y_lo) + (int64_t)m_w * 64;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 16 + (lane >> 2);
                const int off = row * 64 + (((lane & 3) ^ ((row >> 2) & 3)) << 4);
                const uint4 vh = *reinterpret_cast<const uint4*>(wst + off);
                const uint4 vl = *reinterpret_cast<const uint4*>(wst + 4096 + off);
                if (m_w + row < M) {
                    *reinterpret_cast<uint4*>(gh + it * 1024 + lane * 16) = vh;
                    *reinterpret_cast<uint4*>(gl + it * 1024 + lane * 16) = vl;
                }
            }
        }
        if (a.bnp) {
            // [256 pixels][33] transpose: lane (c, q) sums 32 pixels of channel c, then the eight pixel groups are added in order
            const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
            __syncthreads();                                            // every wave has left its output staging
#pragma unroll
            for (int i = 0; i < 32; ++i) stage[threadIdx.x * SC32_PAD + i] = live ? acc[i] : 0.f;
            __syncthreads();
            float t1 = 0.f, t2 = 0.f;
            for (int i = 0; i < 32; ++i) {
                const float v = stage[(q * 32 + i) * SC32_PAD + c];
                t1 += v; t2 += v * v;
            }
            __syncthreads();
            stage[q * 32 + c] = t1;
            stage[256 + q * 32 + c] = t2;
            __syncthreads();
            if (threadIdx.x < 64) {
                const int cc = threadIdx.x & 31, st = threadIdx.x >> 5;
                const float* sp = stage + st * 256 + cc;
                float t = 0.f;
#pragma unroll
                for (int g = 0; g < 8; ++g) t += sp[g * 32];
                float* dst = a.bnp + (int64_t)blockIdx.x * 64 + st * 32 + cc;
                *dst = sub == 0 ? t : *dst + t;                          // same thread, same address on every pass
            }
        }
        __syncthreads();
    }
}

// ---- BatchNorm apply + activation (+ 2x2 max-pool) on pairs ---------------------------------------------------------
struct ApplySArgs {
    const unsigned short* y_hi; const unsigned short* y_lo;
    const float* scale; const float* shift;
    unsigned short* z_hi; unsigned short* z_lo;       // both with (zs, zc)
    unsigned short* zp_hi; unsigned short* zp_lo;     // pooled pair, both with pixel stride zps (channel offset 0)
    int act, N, H, W, C, zs, zc, zps;
    int zp_q8 = 0;                                    // the pooled lo plane is a q plane (FP8 correction chunks, common.hpp)
};

// Plain form: a thread owns ONE 8-channel chunk for the whole launch (scale / shift live in registers, no index divisions:
// pixels are a flat range, y dense with stride C, z with stride zs) and keeps UNR pixels in flight -- 2 x UNR 16-byte loads
// issued before the first use.  The first version (one element per grid-stride iteration, 64-bit div / mod per element,
// coefficients re-loaded) ran the level-0 tensors at 3.1 TB/s of its four streams.  z_lo may be NULL (no consumer reads it).
template <int DT, int LOF>
__global__ __launch_bounds__(256) void bn_act_apply_split_kernel(const ApplySArgs a) {
    constexpr int UNR = 4;
    constexpr bool LO = LOF == 1;
    const float q_slo = __builtin_ldexpf(1.f, GS_Q8_XH_EXP + Q8Shift<DT>::v), q_shi = __builtin_ldexpf(1.f, GS_Q8_XH_EXP);
    const float slope = a.act == GS_ACT_RELU ? 0.f : (a.act == GS_ACT_LEAKY02 ? 0.2f : 1.f);
    const int nch = a.C >> 3;                              // a power-of-two-free divisor of the thread count (host-checked)
    const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int ch = (int)(gt % nch);
    const int64_t pl = gt / nch, npl = ((int64_t)gridDim.x * 256) / nch;
    const int c0 = ch * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = a.scale ? a.scale[c0 + i] : 1.f; sh[i] = a.shift ? a.shift[c0 + i] : 0.f; }
    const int64_t P = (int64_t)a.N * a.H * a.W;
    const unsigned short* yh = a.y_hi + c0;
    const unsigned short* yl = a.y_lo + c0;
    unsigned short* zh = a.z_hi + a.zc + c0;
    unsigned short* zl = LO ? a.z_lo + a.zc + c0 : nullptr;
    unsigned char* zq = LOF == 2 ? reinterpret_cast<unsigned char*>(a.z_lo) + q8_off(a.zc + c0) : nullptr;
    for (int64_t p0 = pl; p0 < P; p0 += npl * UNR) {
        uint4 vh[UNR], vl[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t p = p0 + u * npl;
            if (p < P) {
                vh[u] = *reinterpret_cast<const uint4*>(yh + p * a.C);
                vl[u] = *reinterpret_cast<const uint4*>(yl + p * a.C);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t p = p0 + u * npl;
            if (p < P) {
                float v[8];
                join8<DT>(vh[u], vl[u], v);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float t = v[i] * sc[i] + sh[i];
                    v[i] = t > 0.f ? t : t * slope;
                }
                if (LO) {
                    uint4 hi, lo;
                    split8<DT>(v, hi, lo);
                    st16(zh + p * a.zs, hi);
                    st16(zl + p * a.zs, lo);
                } else if (LOF == 2) {
                    const uint4 hi = pack8<DT>(v);
                    uint2 lo8, hi8;
                    q8_of8<DT>(v, hi, q_slo, q_shi, lo8, hi8);
                    st16(zh + p * a.zs, hi);
                    *reinterpret_cast<uint2*>(zq + p * a.zs * 2) = lo8;
                    *reinterpret_cast<uint2*>(zq + p * a.zs * 2 + 32) = hi8;
                } else {
                    st16(zh + p * a.zs, pack8<DT>(v));
                }
            }
        }
    }
}

// Pooled form: a thread owns one chunk; units are 2x2 windows (32-bit index arithmetic), the four pixels of a window are
// loaded before the first use.  The pooled pair is the maximum of the STORED pair values.
template <int DT, int LOF>
__global__ __launch_bounds__(256) void bn_act_apply_split_pool_kernel(const ApplySArgs a) {
    constexpr bool LO = LOF == 1;
    const float q_slo = __builtin_ldexpf(1.f, GS_Q8_XH_EXP + Q8Shift<DT>::v), q_shi = __builtin_ldexpf(1.f, GS_Q8_XH_EXP);
    const float slope = a.act == GS_ACT_RELU ? 0.f : (a.act == GS_ACT_LEAKY02 ? 0.2f : 1.f);
    const int nch = a.C >> 3;
    const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int ch = (int)(gt % nch);
    const int pl = (int)(gt / nch), npl = (int)(((int64_t)gridDim.x * 256) / nch);
    const int c0 = ch * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = a.scale ? a.scale[c0 + i] : 1.f; sh[i] = a.shift ? a.shift[c0 + i] : 0.f; }
    const int PH = (a.H + 1) / 2, PW = (a.W + 1) / 2;
    const int units = a.N * PH * PW;                       // host guarantees < 2^31
    for (int u = pl; u < units; u += npl) {
        const int px = u % PW;
        const int r = u / PW;
        const int py = r % PH, n = r / PH;
        uint4 vh[4], vl[4];
        bool ok[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int yy = 2 * py + (q >> 1), xx = 2 * px + (q & 1);
            ok[q] = yy < a.H && xx < a.W;
            if (ok[q]) {
                const int64_t pix = ((int64_t)n * a.H + yy) * a.W + xx;
                vh[q] = *reinterpret_cast<const uint4*>(a.y_hi + pix * a.C + c0);
                vl[q] = *reinterpret_cast<const uint4*>(a.y_lo + pix * a.C + c0);
            }
        }
        float mx[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) mx[i] = -INFINITY;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!ok[q]) continue;
            const int yy = 2 * py + (q >> 1), xx = 2 * px + (q & 1);
            const int64_t pix = ((int64_t)n * a.H + yy) * a.W + xx;
            float v[8];
            join8<DT>(vh[q], vl[q], v);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float t = v[i] * sc[i] + sh[i];
                v[i] = t > 0.f ? t : t * slope;
            }
            uint4 hi, lo;
            split8<DT>(v, hi, lo);
            st16(a.z_hi + pix * a.zs + a.zc + c0, hi);
            if (LO) st16(a.z_lo + pix * a.zs + a.zc + c0, lo);
            if (LOF == 2) {
                uint2 lo8, hi8;
                q8_of8<DT>(v, hi, q_slo, q_shi, lo8, hi8);
                unsigned char* zq = reinterpret_cast<unsigned char*>(a.z_lo) + pix * a.zs * 2 + q8_off(a.zc + c0);
                *reinterpret_cast<uint2*>(zq) = lo8;
                *reinterpret_cast<uint2*>(zq + 32) = hi8;
            }
            float rr[8];
            if (LOF != 0) join8<DT>(hi, lo, rr);             // the pooled value: the maximum of the pair values (16-bit lo precision)
            else unpack8<DT>(hi, rr);
#pragma unroll
            for (int i = 0; i < 8; ++i) mx[i] = fmaxf(mx[i], rr[i]);
        }
        if (py < a.H / 2 && px < a.W / 2) {
            const int64_t pp = ((int64_t)n * (a.H / 2) + py) * (a.W / 2) + px;
            uint4 hi, lo;
            split8<DT>(mx, hi, lo);
            st16(a.zp_hi + pp * a.zps + c0, hi);
            if (a.zp_lo && !a.zp_q8) st16(a.zp_lo + pp * a.zps + c0, lo);
            if (a.zp_lo && a.zp_q8) {
                uint2 lo8, hi8;
                q8_of8<DT>(mx, hi, q_slo, q_shi, lo8, hi8);
                unsigned char* zq = reinterpret_cast<unsigned char*>(a.zp_lo) + pp * a.zps * 2 + q8_off(c0);
                *reinterpret_cast<uint2*>(zq) = lo8;
                *reinterpret_cast<uint2*>(zq + 32) = hi8;
            }
        }
    }
}

// 3-D form (UNet3D's analysis blocks, GenSeg-3D/UNet3D/unet3d.py:29-36: BatchNorm3d + ReLU + MaxPool3d(2)): units are 2x2x2 windows of
// the [NB*D, H, W] voxel grid (even D, H, W: host-checked), the eight voxels of a window are loaded before the first use; the z pair is
// stored as bn_act_apply_split_kernel stores it and the pooled pair is the maximum of the STORED pair values -- bit-identical to
// gs_bn_act_apply_split followed by gs_maxpool3d_fwd_pair, without reading the z pair back (537 MB at 128^3).
template <int DT, int LOF>
__global__ __launch_bounds__(256) void bn_act_apply_split_pool3d_kernel(const ApplySArgs a, int D) {
    constexpr bool LO = LOF == 1;
    const float slope = a.act == GS_ACT_RELU ? 0.f : (a.act == GS_ACT_LEAKY02 ? 0.2f : 1.f);
    const int nch = a.C >> 3;
    const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int ch = (int)(gt % nch);
    const int pl = (int)(gt / nch), npl = (int)(((int64_t)gridDim.x * 256) / nch);
    const int c0 = ch * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = a.scale ? a.scale[c0 + i] : 1.f; sh[i] = a.shift ? a.shift[c0 + i] : 0.f; }
    const int PD = D / 2, PH = a.H / 2, PW = a.W / 2, NB = a.N / D;
    const int units = NB * PD * PH * PW;                   // host guarantees < 2^31
    for (int u = pl; u < units; u += npl) {
        const int px = u % PW;
        int r = u / PW;
        const int py = r % PH; r /= PH;
        const int pd = r % PD, nb = r / PD;
        uint4 vh[8], vl[8];
        int64_t pix[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            pix[q] = (((int64_t)nb * D + 2 * pd + (q >> 2)) * a.H + 2 * py + ((q >> 1) & 1)) * a.W + 2 * px + (q & 1);
            vh[q] = *reinterpret_cast<const uint4*>(a.y_hi + pix[q] * a.C + c0);
            vl[q] = *reinterpret_cast<const uint4*>(a.y_lo + pix[q] * a.C + c0);
        }
        float mx[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) mx[i] = -INFINITY;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float v[8];
            join8<DT>(vh[q], vl[q], v);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float t = v[i] * sc[i] + sh[i];
                v[i] = t > 0.f ? t : t * slope;
            }
            uint4 hi, lo;
            split8<DT>(v, hi, lo);
            st16(a.z_hi + pix[q] * a.zs + a.zc + c0, hi);
            if (LO) st16(a.z_lo + pix[q] * a.zs + a.zc + c0, lo);
            float rr[8];
            if (LO) join8<DT>(hi, lo, rr);
            else unpack8<DT>(hi, rr);
#pragma unroll
            for (int i = 0; i < 8; ++i) mx[i] = fmaxf(mx[i], rr[i]);
        }
        const int64_t pp = (((int64_t)nb * PD + pd) * PH + py) * PW + px;
        uint4 hi, lo;
        split8<DT>(mx, hi, lo);
        st16(a.zp_hi + pp * a.zps + c0, hi);
        if (a.zp_lo) st16(a.zp_lo + pp * a.zps + c0, lo);
    }
}

// ---- pointwise head, Cin == 64, on a pair of dense inputs: 8 lanes per pixel ---------------------------------------
struct HeadSArgs {
    const unsigned short* x_hi; const unsigned short* x_lo; const float* w; const float* bias; float* y;
    int N, HW, Cout;
    const float* scale = nullptr; const float* shift = nullptr;      // BatchNorm + activation on the load path (x = a conv output pair)
    int act = GS_ACT_NONE;
};

__device__ __forceinline__ float sum8_dpp_p(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    return v;
}

template <int DT>
__global__ __launch_bounds__(256) void head1x1_split_kernel(const HeadSArgs a) {
    const int ch = threadIdx.x & 7, pl = threadIdx.x >> 3;
    float w[4][8], bv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        bv[c] = (c < a.Cout && a.bias) ? a.bias[c] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) w[c][i] = c < a.Cout ? a.w[c * 64 + ch * 8 + i] : 0.f;
    }
    const int M = a.N * a.HW;
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = a.scale ? a.scale[ch * 8 + i] : 1.f; sh[i] = a.scale ? a.shift[ch * 8 + i] : 0.f; }
    const float slope = a.act == GS_ACT_RELU ? 0.f : (a.act == GS_ACT_LEAKY02 ? 0.2f : 1.f);
    for (int m = blockIdx.x * 32 + pl; m < M; m += gridDim.x * 32) {
        float v[8], s[4];
        join8<DT>(*reinterpret_cast<const uint4*>(a.x_hi + (int64_t)m * 64 + ch * 8),
                  *reinterpret_cast<const uint4*>(a.x_lo + (int64_t)m * 64 + ch * 8), v);
        if (a.scale) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float t = v[i] * sc[i] + sh[i];
                v[i] = t > 0.f ? t : t * slope;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) t += v[i] * w[c][i];
            s[c] = sum8_dpp_p(t);
        }
        if (ch == 0) {
            const int n = m / a.HW, hw = m - n * a.HW;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < a.Cout) a.y[((int64_t)n * a.Cout + c) * a.HW + hw] = s[c] + bv[c];
        }
    }
}

}  // namespace

extern "C" int gs_pack_weight_split(const float* w, void* pack, int Cout, int Cin, int taps, int transposed, int dtype,
                                    void* stream) {
    GS_CHECK_ARG(w && pack && Cout > 0 && Cin > 0 && taps > 0, "gs_pack_weight_split: bad arguments");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_pack_weight_split: bad dtype");
    const int64_t total = (int64_t)taps * Cout * Cin;
    int64_t nb = cdiv64(total, 256);
    if (nb > 4096) nb = 4096;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GS_F16) pack_split_kernel<GS_F16><<<(int)nb, 256, 0, s>>>(w, (unsigned short*)pack, Cout, Cin, taps, transposed);
    else pack_split_kernel<GS_BF16><<<(int)nb, 256, 0, s>>>(w, (unsigned short*)pack, Cout, Cin, taps, transposed);
    GS_CHECK_LAUNCH("gs_pack_weight_split");
    return GS_OK;
}


extern "C" int gs_pack_weight_segs(int n, const GsSegPackDesc* descs, int dtype, void* stream) {
    GS_CHECK_ARG(n > 0 && descs != nullptr, "gs_pack_weight_segs: no descriptors");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_pack_weight_segs: bad dtype");
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < n; base += SEGPACK_MAX) {
        SegPackArgs a;
        a.n = n - base < SEGPACK_MAX ? n - base : SEGPACK_MAX;
        int blocks = 0;
        for (int i = 0; i < a.n; ++i) {
            const GsSegPackDesc& d = descs[base + i];
            GS_CHECK_ARG(d.w && d.pack && d.Cout > 0 && d.Cin > 0 && d.taps > 0 && d.nseg >= 1 && d.nseg <= GS_SEG_MAX,
                         "gs_pack_weight_segs: descriptor %d: bad arguments", base + i);
            int K = 0;
            for (int j = 0; j < d.nseg; ++j) {
                GS_CHECK_ARG((d.kind[j] >= 0 && d.kind[j] <= 2) && d.ci0[j] >= 0 && d.len[j] > 0 && d.ci0[j] + d.len[j] <= d.Cin,
                             "gs_pack_weight_segs: descriptor %d segment %d out of range", base + i, j);
                a.kind[i][j] = d.kind[j]; a.ci0[i][j] = d.ci0[j]; a.len[i][j] = d.len[j];
                K += d.len[j];
            }
            a.w[i] = d.w; a.out[i] = (unsigned short*)d.pack;
            a.Cout[i] = d.Cout; a.Cin[i] = d.Cin; a.taps[i] = d.taps; a.transposed[i] = d.transposed; a.K[i] = K; a.nseg[i] = d.nseg;
            a.first[i] = blocks;
            int64_t b = cdiv64((int64_t)d.Cout * d.Cin, 256);
            if (b > 1024) b = 1024;
            blocks += (int)b;
        }
        a.first[a.n] = blocks;
        if (dtype == GS_F16) pack_segs_kernel<GS_F16><<<blocks, 256, 0, s>>>(a);
        else pack_segs_kernel<GS_BF16><<<blocks, 256, 0, s>>>(a);
        GS_CHECK_LAUNCH("gs_pack_weight_segs");
    }
    return GS_OK;
}

extern "C" int gs_pack_weight_q8(int n, const GsQ8PackDesc* descs, int dtype, void* stream) {
    GS_CHECK_ARG(n > 0 && descs != nullptr, "gs_pack_weight_q8: no descriptors");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_pack_weight_q8: bad dtype");
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < n; base += Q8PACK_MAX) {
        Q8PackArgs a;
        a.n = n - base < Q8PACK_MAX ? n - base : Q8PACK_MAX;
        int blocks = 0;
        for (int i = 0; i < a.n; ++i) {
            const GsQ8PackDesc& d = descs[base + i];
            GS_CHECK_ARG(d.w && d.pack && d.wexp && d.Cout > 0 && d.Cin > 0 && d.Cin % 32 == 0 && d.taps > 0,
                         "gs_pack_weight_q8: descriptor %d: needs w, pack, wexp and Cin %% 32 == 0", base + i);
            a.w[i] = d.w; a.out[i] = (unsigned char*)d.pack; a.wexp[i] = d.wexp;
            a.Cout[i] = d.Cout; a.Cin[i] = d.Cin; a.taps[i] = d.taps;
            a.first[i] = blocks;
            blocks += cdiv(d.Cout, 4);
        }
        a.first[a.n] = blocks;
        if (dtype == GS_F16) pack_q8_kernel<GS_F16><<<blocks, 256, 0, s>>>(a);
        else pack_q8_kernel<GS_BF16><<<blocks, 256, 0, s>>>(a);
        GS_CHECK_LAUNCH("gs_pack_weight_q8");
    }
    return GS_OK;
}

extern "C" int gs_q8_from_hi(const void* x, void* q, int64_t pixels, int C, int pix_stride, int coff, int dtype, void* stream) {
    GS_CHECK_ARG(x && q && pixels > 0 && C > 0 && C % 32 == 0 && coff % 32 == 0 && pix_stride % 8 == 0 && pix_stride >= coff + C,
                 "gs_q8_from_hi: C and coff must be multiples of 32 inside the pixel stride");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_q8_from_hi: bad dtype");
    int64_t nb = cdiv64(pixels * (C / 8), 256 * 4);
    if (nb > 8192) nb = 8192;
    if (nb < 1) nb = 1;
    if (dtype == GS_F16) q8_from_hi_kernel<GS_F16><<<(int)nb, 256, 0, (hipStream_t)stream>>>((const unsigned short*)x, (unsigned char*)q, pixels, C, pix_stride, coff);
    else q8_from_hi_kernel<GS_BF16><<<(int)nb, 256, 0, (hipStream_t)stream>>>((const unsigned short*)x, (unsigned char*)q, pixels, C, pix_stride, coff);
    GS_CHECK_LAUNCH("gs_q8_from_hi");
    return GS_OK;
}

extern "C" int gs_conv_smallcin_fwd_split(const float* x, const float* w, void* y_hi, void* y_lo, float* bn_partials, int N,
                                          int Cin, int H, int W, int Cout, int k, int pad, int dtype, void* stream) {
    GS_CHECK_ARG(x && w && y_hi && y_lo, "gs_conv_smallcin_fwd_split: null pointer");
    GS_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin >= 1 && Cin <= 4 && k >= 1 && 2 * pad == k - 1,
                 "gs_conv_smallcin_fwd_split: stride-1 'same' convolutions of 1..4 channels only");
    GS_CHECK_ARG(Cout % 8 == 0 && Cout >= 8 && Cout <= 256 && (256 % (Cout / 8)) == 0 && Cin * k * k * Cout <= SCP_MAX_W,
                 "gs_conv_smallcin_fwd_split: Cout=%d not supported", Cout);
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_conv_smallcin_fwd_split: bad dtype");
    SCPArgs a{x, w, (unsigned short*)y_hi, (unsigned short*)y_lo, bn_partials, N, Cin, H, W, Cout, k, pad};
    const int nb = (int)cdiv64((int64_t)N * H * W, SCP_TILE);
    hipStream_t s = (hipStream_t)stream;
    if (Cout == 32 && Cin * k * k <= SC32_MAX_T && (int64_t)N * H * W < 2147483647LL && (((uintptr_t)y_hi | (uintptr_t)y_lo) & 15) == 0 &&
        getenv("GSSEG_SC32_OFF") == nullptr) {       // (getenv per call: the kernel test switches forms in one process)       // one thread per pixel, complete-line stores (the UNet3D stem)
        if (dtype == GS_F16) smallcin_split32_kernel<GS_F16><<<nb, 256, 0, s>>>(a);
        else smallcin_split32_kernel<GS_BF16><<<nb, 256, 0, s>>>(a);
        GS_CHECK_LAUNCH("gs_conv_smallcin_fwd_split");
        return GS_OK;
    }
    if (dtype == GS_F16) smallcin_split_kernel<GS_F16><<<nb, 256, 0, s>>>(a);
    else smallcin_split_kernel<GS_BF16><<<nb, 256, 0, s>>>(a);
    GS_CHECK_LAUNCH("gs_conv_smallcin_fwd_split");
    return GS_OK;
}

static int bn_act_apply_split_impl(const void* y_hi, const void* y_lo, const float* scale, const float* shift, int act,
                                   void* z_hi, void* z_lo, int z_pix_stride, int z_coff, void* zp_hi, void* zp_lo,
                                   int zp_pix_stride, int N, int H, int W, int C, int dtype, void* stream, int z_q8, int zp_q8);
extern "C" int gs_bn_act_apply_split(const void* y_hi, const void* y_lo, const float* scale, const float* shift, int act,
                                     void* z_hi, void* z_lo, int z_pix_stride, int z_coff, void* zp_hi, void* zp_lo,
                                     int zp_pix_stride, int N, int H, int W, int C, int dtype, void* stream) {
    return bn_act_apply_split_impl(y_hi, y_lo, scale, shift, act, z_hi, z_lo, z_pix_stride, z_coff, zp_hi, zp_lo, zp_pix_stride, N, H, W,
                                   C, dtype, stream, 0, 0);
}
// the same with the lo planes as Q PLANES (FP8 correction chunks, common.hpp): z_q8 / zp_q8 != 0 -> z_lo / zp_lo point at the q
// plane of their buffer (its byte 0 = channel 0 of the buffer; z_coff selects the chunk), which takes the place of the 16-bit lo
// plane for consumers that run a "q" stage.  Channel offsets and C must be multiples of 32 for a q plane.
extern "C" int gs_bn_act_apply_split_q8(const void* y_hi, const void* y_lo, const float* scale, const float* shift, int act,
                                        void* z_hi, void* z_lo, int z_q8, int z_pix_stride, int z_coff, void* zp_hi, void* zp_lo,
                                        int zp_q8, int zp_pix_stride, int N, int H, int W, int C, int dtype, void* stream) {
    GS_CHECK_ARG((!z_q8 || (z_lo && z_coff % 32 == 0 && C % 32 == 0)) && (!zp_q8 || (zp_lo && C % 32 == 0)),
                 "gs_bn_act_apply_split_q8: a q plane needs its pointer and channel counts / offsets that are multiples of 32");
    return bn_act_apply_split_impl(y_hi, y_lo, scale, shift, act, z_hi, z_lo, z_pix_stride, z_coff, zp_hi, zp_lo, zp_pix_stride, N, H, W,
                                   C, dtype, stream, z_q8 ? 1 : 0, zp_q8 ? 1 : 0);
}
static int bn_act_apply_split_impl(const void* y_hi, const void* y_lo, const float* scale, const float* shift, int act,
                                   void* z_hi, void* z_lo, int z_pix_stride, int z_coff, void* zp_hi, void* zp_lo,
                                   int zp_pix_stride, int N, int H, int W, int C, int dtype, void* stream, int z_q8, int zp_q8) {
    GS_CHECK_ARG(y_hi && y_lo && z_hi && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "gs_bn_act_apply_split: bad arguments");
    GS_CHECK_ARG(z_pix_stride >= z_coff + C && z_pix_stride % 8 == 0 && z_coff % 8 == 0, "gs_bn_act_apply_split: bad z stride");
    GS_CHECK_ARG((scale == nullptr) == (shift == nullptr), "gs_bn_act_apply_split: scale/shift must both be given or NULL");
    GS_CHECK_ARG((zp_hi != nullptr || zp_lo == nullptr) && (zp_hi == nullptr || (zp_pix_stride >= C && zp_pix_stride % 8 == 0)),
                 "gs_bn_act_apply_split: bad pooled output");
    GS_CHECK_ARG(act == GS_ACT_NONE || act == GS_ACT_RELU || act == GS_ACT_LEAKY02, "gs_bn_act_apply_split: activation %d", act);
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_bn_act_apply_split: bad dtype");
    GS_CHECK_ARG((int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) < 2147483000LL, "gs_bn_act_apply_split: too many pixels");
    ApplySArgs a{(const unsigned short*)y_hi, (const unsigned short*)y_lo, scale, shift, (unsigned short*)z_hi,
                 (unsigned short*)z_lo, (unsigned short*)zp_hi, (unsigned short*)zp_lo, act, N, H, W, C, z_pix_stride, z_coff,
                 zp_pix_stride};
    a.zp_q8 = zp_q8;
    const bool pool = zp_hi != nullptr;
    const int lof = z_lo == nullptr ? 0 : (z_q8 ? 2 : 1);
    const int nch = C / 8;
    // the grid's thread count must be a multiple of the chunk count (a thread keeps its chunk): blocks of 256 threads,
    // nch | 256 * blocks
    const int PH = pool ? (H + 1) / 2 : H, PW = pool ? (W + 1) / 2 : W;
    const int64_t work = (int64_t)N * PH * PW * nch / (pool ? 1 : 4);
    int64_t blocks = cdiv64(work, 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    int bq = 1;                                            // smallest block multiple with nch | 256 * bq
    while ((256 * bq) % nch != 0) ++bq;
    blocks = cdiv64(blocks, bq) * bq;
    hipStream_t s = (hipStream_t)stream;
#define GS_APPLY_SPLIT(DT)                                                                         \
    do {                                                                                           \
        if (pool) {                                                                                \
            if (lof == 1) bn_act_apply_split_pool_kernel<DT, 1><<<(int)blocks, 256, 0, s>>>(a);    \
            else if (lof == 2) bn_act_apply_split_pool_kernel<DT, 2><<<(int)blocks, 256, 0, s>>>(a); \
            else bn_act_apply_split_pool_kernel<DT, 0><<<(int)blocks, 256, 0, s>>>(a);             \
        } else {                                                                                   \
            if (lof == 1) bn_act_apply_split_kernel<DT, 1><<<(int)blocks, 256, 0, s>>>(a);         \
            else if (lof == 2) bn_act_apply_split_kernel<DT, 2><<<(int)blocks, 256, 0, s>>>(a);    \
            else bn_act_apply_split_kernel<DT, 0><<<(int)blocks, 256, 0, s>>>(a);                  \
        }                                                                                          \
    } while (0)
    if (dtype == GS_F16) GS_APPLY_SPLIT(GS_F16);
    else GS_APPLY_SPLIT(GS_BF16);
#undef GS_APPLY_SPLIT
    GS_CHECK_LAUNCH("gs_bn_act_apply_split");
    return GS_OK;
}

extern "C" int gs_bn_act_apply_split_pool3d(const void* y_hi, const void* y_lo, const float* scale, const float* shift, int act,
                                            void* z_hi, void* z_lo, int z_pix_stride, int z_coff, void* zp_hi, void* zp_lo,
                                            int zp_pix_stride, int NB, int D, int H, int W, int C, int dtype, void* stream) {
    GS_CHECK_ARG(y_hi && y_lo && z_hi && zp_hi && NB > 0 && D > 1 && H > 1 && W > 1 && C > 0 && C % 8 == 0, "gs_bn_act_apply_split_pool3d: bad arguments");
    GS_CHECK_ARG(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "gs_bn_act_apply_split_pool3d: even volume dims only");
    GS_CHECK_ARG(z_pix_stride >= z_coff + C && z_pix_stride % 8 == 0 && z_coff % 8 == 0 && zp_pix_stride >= C && zp_pix_stride % 8 == 0,
                 "gs_bn_act_apply_split_pool3d: bad strides");
    GS_CHECK_ARG((scale == nullptr) == (shift == nullptr), "gs_bn_act_apply_split_pool3d: scale/shift must both be given or NULL");
    GS_CHECK_ARG(act == GS_ACT_NONE || act == GS_ACT_RELU || act == GS_ACT_LEAKY02, "gs_bn_act_apply_split_pool3d: activation %d", act);
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_bn_act_apply_split_pool3d: bad dtype");
    GS_CHECK_ARG((int64_t)NB * D * H * W < 2147483000LL, "gs_bn_act_apply_split_pool3d: too many voxels");
    ApplySArgs a{(const unsigned short*)y_hi, (const unsigned short*)y_lo, scale, shift, (unsigned short*)z_hi, (unsigned short*)z_lo,
                 (unsigned short*)zp_hi, (unsigned short*)zp_lo, act, NB * D, H, W, C, z_pix_stride, z_coff, zp_pix_stride};
    const int nch = C / 8;
    int64_t blocks = cdiv64((int64_t)NB * (D / 2) * (H / 2) * (W / 2) * nch, 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    int bq = 1;                                            // smallest block multiple with nch | 256 * bq (a thread keeps its chunk)
    while ((256 * bq) % nch != 0) ++bq;
    blocks = cdiv64(blocks, bq) * bq;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GS_F16) {
        if (z_lo) bn_act_apply_split_pool3d_kernel<GS_F16, 1><<<(int)blocks, 256, 0, s>>>(a, D);
        else bn_act_apply_split_pool3d_kernel<GS_F16, 0><<<(int)blocks, 256, 0, s>>>(a, D);
    } else {
        if (z_lo) bn_act_apply_split_pool3d_kernel<GS_BF16, 1><<<(int)blocks, 256, 0, s>>>(a, D);
        else bn_act_apply_split_pool3d_kernel<GS_BF16, 0><<<(int)blocks, 256, 0, s>>>(a, D);
    }
    GS_CHECK_LAUNCH("gs_bn_act_apply_split_pool3d");
    return GS_OK;
}

static int head1x1_split_impl(const void* x_hi, const void* x_lo, const float* scale, const float* shift, int act, const float* w,
                              const float* bias, float* y, int N, int H, int W, int Cin, int Cout, int dtype, void* stream);
// the head on a CONV OUTPUT pair with BatchNorm + activation on the load path: the last activation pair is never stored
extern "C" int gs_head1x1_bn_fwd_split(const void* y_hi, const void* y_lo, const float* scale, const float* shift, int act,
                                       const float* w, const float* bias, float* logits, int N, int H, int W, int Cin, int Cout,
                                       int dtype, void* stream) {
    GS_CHECK_ARG(scale && shift, "gs_head1x1_bn_fwd_split: scale / shift are NULL");
    GS_CHECK_ARG(act == GS_ACT_NONE || act == GS_ACT_RELU || act == GS_ACT_LEAKY02, "gs_head1x1_bn_fwd_split: activation %d", act);
    return head1x1_split_impl(y_hi, y_lo, scale, shift, act, w, bias, logits, N, H, W, Cin, Cout, dtype, stream);
}
extern "C" int gs_head1x1_fwd_split(const void* x_hi, const void* x_lo, const float* w, const float* bias, float* y, int N,
                                    int H, int W, int Cin, int Cout, int dtype, void* stream) {
    return head1x1_split_impl(x_hi, x_lo, nullptr, nullptr, GS_ACT_NONE, w, bias, y, N, H, W, Cin, Cout, dtype, stream);
}
static int head1x1_split_impl(const void* x_hi, const void* x_lo, const float* scale, const float* shift, int act, const float* w,
                              const float* bias, float* y, int N, int H, int W, int Cin, int Cout, int dtype, void* stream) {
    GS_CHECK_ARG(x_hi && x_lo && w && y && N > 0 && H > 0 && W > 0, "gs_head1x1_fwd_split: bad arguments");
    GS_CHECK_ARG(Cin == 64 && Cout >= 1 && Cout <= 4, "gs_head1x1_fwd_split: Cin must be 64 and Cout 1..4");
    GS_CHECK_ARG((int64_t)N * H * W + 256 < 2147483647LL, "gs_head1x1_fwd_split: too many pixels");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_head1x1_fwd_split: bad dtype");
    HeadSArgs a{(const unsigned short*)x_hi, (const unsigned short*)x_lo, w, bias, y, N, H * W, Cout};
    a.scale = scale; a.shift = shift; a.act = act;
    int64_t hb = cdiv64((int64_t)N * H * W, 32);
    if (hb > 8192) hb = 8192;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GS_F16) head1x1_split_kernel<GS_F16><<<(int)hb, 256, 0, s>>>(a);
    else head1x1_split_kernel<GS_BF16><<<(int)hb, 256, 0, s>>>(a);
    GS_CHECK_LAUNCH("gs_head1x1_fwd_split");
    return GS_OK;
}
