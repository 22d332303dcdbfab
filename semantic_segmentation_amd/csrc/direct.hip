// Direct (VALU) convolutions for the 1..4-channel ends of the networks, where the implicit-GEMM K or N
// dimension is too small for MFMA and the kernels are HBM-bound:
//   small-Cin : U-Net inc.0 (1->64, 3x3; unet_parts.py:16), generator outermost down-conv (1->64, 4x4 s2;
//               networks.py:582), PatchGAN first conv (2->64, 4x4 s2; networks.py:640).  Reads the fp32 NCHW
//               image directly (no separate layout pass), writes 16-bit NHWC + BatchNorm partial sums.
//   small-Cout: OutConv 1x1 (64->n_classes; unet_parts.py:74) and the PatchGAN last conv (512->1, 4x4;
//               networks.py:661).  Reads 16-bit NHWC, writes fp32 NCHW logits.
#include <stdlib.h>

#include "common.hpp"

namespace {

constexpr int SC_TILE = 1024;       // output pixels per block / BN tile (small-Cin fwd)
constexpr int SC_GROUP = 4;         // tiles per block of the statistics-only stem pass
constexpr int SC_MAX_W = 8192;      // floats of weights cached in LDS

struct SCArgs {
    const float* x; const float* w; const float* bias; unsigned short* y; float* bnp;
    int N, Cin, IH, IW, Cout, OH, OW, k, stride, pad, act;
    const float* bn_scale; const float* bn_shift;      // smallcin_fwd64_line_kernel MODE 2
    float* sg;                                         // MODE 1: per-block tap sums + Gram entries [block][54] (may be null)
    unsigned short* y_lo = nullptr;                    // MODE 2: lo half of the output pair (pair forward), same stride
    int ys = 64;                                       // MODE 2: pixel stride of y / y_lo
    int lo_q8 = 0;                                     // MODE 2: y_lo is a q plane (FP8 correction chunks, common.hpp) of pixel stride ys
    int rows256 = 0;                                   // MODE 2: y_lo == y + 64 and ys == 128 (one 256-byte row per pixel), staged through LDS
};

template <int DT>
__global__ __launch_bounds__(256) void smallcin_fwd_kernel(const SCArgs a) {
    __shared__ float wl[SC_MAX_W];            // [tap][Cout]
    __shared__ float red[2][256][8];
    const int T = a.Cin * a.k * a.k;
    for (int i = threadIdx.x; i < T * a.Cout; i += 256) {
        const int co = i % a.Cout, tap = i / a.Cout;
        wl[i] = a.w[(int64_t)co * T + tap];   // [co][ci][ky][kx] -> [tap][co], tap = (ci*k+ky)*k+kx
    }
    __syncthreads();
    const int nch = a.Cout >> 3;
    const int lanes = 256 / nch;
    const int ch = threadIdx.x % nch, pl = threadIdx.x / nch;
    const int64_t M = (int64_t)a.N * a.OH * a.OW;
    const int64_t m0 = (int64_t)blockIdx.x * SC_TILE;
    float bv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bv[i] = a.bias ? a.bias[ch * 8 + i] : 0.f;
    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
    if (pl < lanes) {
        for (int p = pl; p < SC_TILE; p += lanes) {
            const int64_t m = m0 + p;
            if (m >= M) break;
            const int ox = (int)(m % a.OW);
            const int64_t r = m / a.OW;
            const int oy = (int)(r % a.OH);
            const int n = (int)(r / a.OH);
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = 0.f;
            int tap = 0;
            for (int ci = 0; ci < a.Cin; ++ci)
                for (int ky = 0; ky < a.k; ++ky)
                    for (int kx = 0; kx < a.k; ++kx, ++tap) {
                        const int iy = oy * a.stride - a.pad + ky, ix = ox * a.stride - a.pad + kx;
                        float xv = 0.f;
                        if ((unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW)
                            xv = a.x[(((int64_t)n * a.Cin + ci) * a.IH + iy) * a.IW + ix];
                        const float* wp = wl + tap * a.Cout + ch * 8;
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc[i] += xv * wp[i];
                    }
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                s1[i] += acc[i]; s2[i] += acc[i] * acc[i];
                o[i] = act_fwd(acc[i] + bv[i], a.act);
            }
            *reinterpret_cast<uint4*>(a.y + m * a.Cout + ch * 8) = pack8<DT>(o);
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

// Cout == 64 specialisation: ONE THREAD PER OUTPUT PIXEL holding all 64 channels.  The [tap][64] weights are
// read from LDS with wave-uniform addresses (broadcast ds_read_b128: 16 reads + 64 FMAs per tap) and every
// lane stores its own full 128-byte output line; the generic kernel above spends 8 lanes (and 8x the index
// arithmetic and image loads) per pixel.  BatchNorm partial sums go through a padded LDS transpose.
constexpr int SC64_PAD = 33;
template <int DT>
__global__ __launch_bounds__(256) void smallcin_fwd64_kernel(const SCArgs a) {
    extern __shared__ float sc64_smem[];
    const int T = a.Cin * a.k * a.k;
    float* wl = sc64_smem;                          // [tap][64]
    float* stage = sc64_smem + T * 64;              // [256 pixels][33]
    for (int i = threadIdx.x; i < T * 64; i += 256) {
        const int co = i & 63, tap = i >> 6;
        wl[i] = a.w[co * T + tap];
    }
    __syncthreads();
    const int M = a.N * a.OH * a.OW;                // host guarantees < 2^31
    for (int sub = 0; sub < SC_TILE / 256; ++sub) {  // a BN tile (SC_TILE pixels) = four passes of 256 pixels
    const int m = blockIdx.x * SC_TILE + sub * 256 + threadIdx.x;
    const bool live = m < M;
    const int mm = live ? m : M - 1;
    const int ox = mm % a.OW;
    const int r = mm / a.OW;
    const int oy = r % a.OH;
    const int n = r / a.OH;
    float acc[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) acc[c] = 0.f;
    const float* xn = a.x + (int64_t)n * a.Cin * a.IH * a.IW;
    int tap = 0;
    for (int ci = 0; ci < a.Cin; ++ci)
        for (int ky = 0; ky < a.k; ++ky) {
            const int iy = oy * a.stride - a.pad + ky;
            for (int kx = 0; kx < a.k; ++kx, ++tap) {
                const int ix = ox * a.stride - a.pad + kx;
                float xv = 0.f;
                if ((unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW)
                    xv = xn[(ci * a.IH + iy) * a.IW + ix];
                const float4* wp = reinterpret_cast<const float4*>(wl + tap * 64);
#pragma unroll
                for (int c4 = 0; c4 < 16; ++c4) {
                    const float4 w4 = wp[c4];
                    acc[4 * c4 + 0] += xv * w4.x;
                    acc[4 * c4 + 1] += xv * w4.y;
                    acc[4 * c4 + 2] += xv * w4.z;
                    acc[4 * c4 + 3] += xv * w4.w;
                }
            }
        }
    if (live) {
        unsigned short* dst = a.y + (int64_t)m * 64;
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                o[i] = act_fwd(acc[c8 * 8 + i] + (a.bias ? a.bias[c8 * 8 + i] : 0.f), a.act);
            *reinterpret_cast<uint4*>(dst + c8 * 8) = pack8<DT>(o);
        }
    }
    if (a.bnp) {
        // two channel halves through a [256 pixels][33] transpose: lane (c, q) sums 64 pixels of channel c
        const int c = threadIdx.x & 31, q = threadIdx.x >> 5;           // 8 pixel groups of 32
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 32; ++i) stage[threadIdx.x * SC64_PAD + i] = live ? acc[hf * 32 + i] : 0.f;
            __syncthreads();
            float t1 = 0.f, t2 = 0.f;
            for (int i = 0; i < 32; ++i) {
                const float v = stage[(q * 32 + i) * SC64_PAD + c];
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
                float* dst = a.bnp + (int64_t)blockIdx.x * 128 + st * 64 + hf * 32 + cc;
                *dst = sub == 0 ? t : *dst + t;      // same thread, same address on every pass
            }
        }
    }
    __syncthreads();
    }
}

// Cout == 64, nine taps (the 3x3 one-channel U-Net stem, unet_parts.py:16 with in_channels = 1), slope-family activation:
// ONE pixel at a time per thread -- all 64 output channels are accumulated, then the pixel's whole 128-byte line is written by
// eight back-to-back 16-byte stores.  (smallcin_fwd64x4_kernel below writes a pixel's line in four 32-byte instalments spread
// over the whole kernel: with every lane holding four lines open, the partial lines are evicted from L2 before they are
// complete -- rocprofv3 WRITE_SIZE 657 MB for a 268 MB output.)  The BatchNorm partial sums do not need the 64 outputs at all:
// y[p][co] = sum_t x_t(p) w[co][t], so  sum_p y = sum_t w[co][t] S[t]  and  sum_p y^2 = sum_{t,t'} w[co][t] w[co][t'] G[t][t']
// with the tap sums S[t] = sum_p x_t(p) and the 9 x 9 tap Gram matrix G (45 accumulators per thread instead of 128; the
// per-channel quadratic forms are evaluated once per block).  Same tile (SC_TILE pixels) and partial layout as the other kernels.
// Measured and dropped (batch 32, 256^2, 151 us as is): two pixels per thread so that every LDS weight read feeds both -- 220 us;
// the wave's 64 lines collected in LDS and written as eight 1 KB stores of complete lines -- 219 us.
// Hides a value from the optimiser (in a __device__ helper: an asm constraint in a __global__ template body loses the host stub).
__device__ __forceinline__ int sc_opaque(int x) { asm volatile("" : "+v"(x)); return x; }

// MODE 0: convolution output y + BatchNorm partials (or bias + activation without them).
// MODE 1: the partials ALONE -- they depend on the image and the weights only (tap sums + Gram matrix), y is not formed.
// MODE 2: y * bn_scale + bn_shift, activation, stored: with MODE 1 in front (and gs_bn_finalize between) the stem's
//         convolution output is never written -- train-mode BatchNorm without the 2 x 268 MB round trip of y at batch 32.
template <int DT, int MODE>
__global__ __launch_bounds__(256) void smallcin_fwd64_line_kernel(const SCArgs a) {
    constexpr int TT = 9, NG = TT * (TT + 1) / 2;
    __shared__ uint4 sc_line_stage[MODE == 2 ? 4 * 64 * 16 : 1];      // MODE 2 with a.rows256: [4 waves][64 rows][256 B]
    __shared__ float wl[TT * 64];                   // [tap][co]
    __shared__ float part[4][TT + NG];              // per-wave tap sums and Gram entries
    __shared__ float bl[64];                        // bias (zeros without one): read per use -- as registers the 64 values spill
    for (int i = threadIdx.x; i < TT * 64; i += 256) {
        const int co = i & 63, tap = i >> 6;
        wl[i] = a.w[co * TT + tap];
    }
    __shared__ float sl[64];                        // MODE 2: BatchNorm scale (bl holds the shift)
    if (threadIdx.x < 64) {
        bl[threadIdx.x] = MODE == 2 ? a.bn_shift[threadIdx.x] : (a.bias ? a.bias[threadIdx.x] : 0.f);
        sl[threadIdx.x] = MODE == 2 ? a.bn_scale[threadIdx.x] : 1.f;
    }
    const int M = a.N * a.OH * a.OW;                // host guarantees < 2^31
    const float slope = a.act == GS_ACT_RELU ? 0.f : (a.act == GS_ACT_LEAKY02 ? 0.2f : 1.f);
    float S[TT], G[NG];
#pragma unroll
    for (int t = 0; t < TT; ++t) S[t] = 0.f;
#pragma unroll
    for (int i = 0; i < NG; ++i) G[i] = 0.f;
    __syncthreads();
    // MODE 1 (statistics only): a block takes SC_GROUP tiles and writes their combined sums into the first tile's row (zeros
    // into the others): the 54 wave reductions + 64 quadratic forms per block are a quarter as many (46 -> 31 us at batch 32)
    constexpr int TPB = MODE == 1 ? SC_GROUP : 1;
#pragma unroll 1
    for (int k = 0; k < TPB * (SC_TILE / 256); ++k) {
        const int m = blockIdx.x * (TPB * SC_TILE) + k * 256 + threadIdx.x;
        const bool live = m < M;
        const int mm = live ? m : 0;
        const int ox = mm % a.OW;
        const int r = mm / a.OW;
        const int oy = r % a.OH;
        const int n = r / a.OH;
        const float* xn = a.x + (int64_t)n * a.IH * a.IW;
        float xv[TT];
#pragma unroll
        for (int tap = 0; tap < TT; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int iy = oy - 1 + ky, ix = ox - 1 + kx;
            const bool ok = live && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
            xv[tap] = ok ? xn[iy * a.IW + ix] : 0.f;
        }
        if (MODE != 2 && a.bnp) {
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                S[t] += xv[t];
#pragma unroll
                for (int u = t; u < TT; ++u) G[t * TT - t * (t - 1) / 2 + (u - t)] += xv[t] * xv[u];      // upper triangle, row-major
            }
        }
        unsigned short* dst = a.y + (int64_t)mm * (MODE == 2 ? a.ys : 64);
        unsigned short* dst_lo = (MODE == 2 && a.y_lo) ? a.y_lo + (int64_t)mm * a.ys : nullptr;
        // MODE 2, [hi(64) | lo(64)] pair rows of 256 contiguous bytes (a.rows256): the wave's 64 rows are assembled in its 16 KB of
        // LDS (16-byte slot s of row r at s ^ (r & 15): conflict-free both ways) and leave as sixteen 1 KB stores of complete lines.
        // Written lane by lane, a row's two 128-byte lines receive sixteen 16-byte pieces spread over the whole channel loop and L2
        // evicts them half written: 1.15 GB of HBM writes for 0.54 GB of pairs at batch 32 (profiles/r04_v2_mixed_pmc_traffic.json).
        const bool rows256 = MODE == 2 && a.rows256;
        unsigned char* const stg = reinterpret_cast<unsigned char*>(sc_line_stage) + (threadIdx.x >> 6) * (64 * 256) + (threadIdx.x & 63) * 256;
        const int swz = threadIdx.x & 15;
#pragma unroll 1
        for (int g = 0; g < (MODE == 1 ? 0 : 4); ++g) {     // 16 channels at a time, a runtime loop: unrolled, the compiler hoists all
            const float* wk = wl + sc_opaque(g * 16);       // 144 weight reads above the FMAs and spills the accumulators
            float acc[16];
            {
                const float4* bp = reinterpret_cast<const float4*>(bl + sc_opaque(g * 16));
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const float4 b = MODE == 2 ? make_float4(0.f, 0.f, 0.f, 0.f) : bp[c4];
                    acc[4 * c4] = b.x; acc[4 * c4 + 1] = b.y; acc[4 * c4 + 2] = b.z; acc[4 * c4 + 3] = b.w;
                }
            }
#pragma unroll
            for (int tap = 0; tap < TT; ++tap) {
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {            // wave-uniform addresses: broadcast reads
                    const float4 w4 = reinterpret_cast<const float4*>(wk + tap * 64)[c4];
                    acc[4 * c4] += xv[tap] * w4.x; acc[4 * c4 + 1] += xv[tap] * w4.y;
                    acc[4 * c4 + 2] += xv[tap] * w4.z; acc[4 * c4 + 3] += xv[tap] * w4.w;
                }
            }
            if (MODE == 2) {                                 // BatchNorm scale / shift of the 16 channels
                const float4* sp = reinterpret_cast<const float4*>(sl + sc_opaque(g * 16));
                const float4* hp = reinterpret_cast<const float4*>(bl + sc_opaque(g * 16));
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const float4 sv = sp[c4], hv = hp[c4];
                    acc[4 * c4] = acc[4 * c4] * sv.x + hv.x; acc[4 * c4 + 1] = acc[4 * c4 + 1] * sv.y + hv.y;
                    acc[4 * c4 + 2] = acc[4 * c4 + 2] * sv.z + hv.z; acc[4 * c4 + 3] = acc[4 * c4 + 3] * sv.w + hv.w;
                }
            }
            if (live) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float o[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const float v = acc[8 * i + c];
                        o[c] = v > 0.f ? v : v * slope;
                    }
                    const uint4 hi = pack8<DT>(o);
                    if (rows256) *reinterpret_cast<uint4*>(stg + (((g * 2 + i) ^ swz) << 4)) = hi;
                    else *reinterpret_cast<uint4*>(dst + g * 16 + 8 * i) = hi;
                    if (MODE == 2 && dst_lo && a.lo_q8) {    // pair forward, "q" consumer: the q-plane bytes of these 8 channels
                        uint2 lo8, hi8;
                        q8_of8<DT>(o, hi, __builtin_ldexpf(1.f, GS_Q8_XH_EXP + Q8Shift<DT>::v), __builtin_ldexpf(1.f, GS_Q8_XH_EXP), lo8, hi8);
                        const int qo = q8_off(g * 16 + 8 * i);
                        if (rows256) {
                            *reinterpret_cast<uint2*>(stg + ((((128 + qo) >> 4) ^ swz) << 4) + (qo & 8)) = lo8;
                            *reinterpret_cast<uint2*>(stg + ((((160 + qo) >> 4) ^ swz) << 4) + (qo & 8)) = hi8;
                        } else {
                            unsigned char* zq = reinterpret_cast<unsigned char*>(dst_lo) + qo;
                            *reinterpret_cast<uint2*>(zq) = lo8;
                            *reinterpret_cast<uint2*>(zq + 32) = hi8;
                        }
                    } else if (MODE == 2 && dst_lo) {        // pair forward: lo = 16-bit(value - hi)
                        float hf[8];
                        unpack8<DT>(hi, hf);
#pragma unroll
                        for (int c = 0; c < 8; ++c) o[c] -= hf[c];
                        if (rows256) *reinterpret_cast<uint4*>(stg + (((8 + g * 2 + i) ^ swz) << 4)) = pack8<DT>(o);
                        else *reinterpret_cast<uint4*>(dst_lo + g * 16 + 8 * i) = pack8<DT>(o);
                    }
                }
            }
        }
        if (rows256) {                                   // (a wave's LDS accesses execute in order: no barrier between its lanes' rows)
            __builtin_amdgcn_wave_barrier();
            const int lane = threadIdx.x & 63;
            const int m_w = m - lane;                    // the wave's first pixel
            const unsigned char* wst = reinterpret_cast<const unsigned char*>(sc_line_stage) + (threadIdx.x >> 6) * (64 * 256);
            unsigned char* gw = reinterpret_cast<unsigned char*>(a.y) + (int64_t)m_w * 256;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row = it * 4 + (lane >> 4);
                const uint4 v = *reinterpret_cast<const uint4*>(wst + row * 256 + (((lane & 15) ^ (row & 15)) << 4));
                if (m_w + row < M) st16(gw + it * 1024 + lane * 16, v);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (MODE != 2 && a.bnp) {
        const int wv = threadIdx.x >> 6;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const float v = wave_sum(S[t]);
            if ((threadIdx.x & 63) == 0) part[wv][t] = v;
        }
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const float v = wave_sum(G[i]);
            if ((threadIdx.x & 63) == 0) part[wv][TT + i] = v;
        }
        __syncthreads();
        if (threadIdx.x < TT + NG)                       // the four waves' sums, in wave order
            part[0][threadIdx.x] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        __syncthreads();
        const int mtiles = (M + SC_TILE - 1) / SC_TILE;
        const int row0 = blockIdx.x * TPB;                  // rows row0+1 .. row0+TPB-1 (inside mtiles) get zeros
        if (MODE == 1 && a.sg != nullptr && threadIdx.x < TT + NG) {
            a.sg[(int64_t)row0 * (TT + NG) + threadIdx.x] = part[0][threadIdx.x];
            for (int r = 1; r < TPB && row0 + r < mtiles; ++r) a.sg[(int64_t)(row0 + r) * (TT + NG) + threadIdx.x] = 0.f;
        }
        if (MODE == 1 && threadIdx.x >= 128)
            for (int r = 1; r < TPB && row0 + r < mtiles; ++r) a.bnp[(int64_t)(row0 + r) * 128 + (threadIdx.x - 128)] = 0.f;
        if (threadIdx.x < 64) {
            const int co = threadIdx.x;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll 1
            for (int t = 0; t < TT; ++t) {
                const float wt = wl[t * 64 + co];
                s1 += wt * part[0][t];
                const int row = TT + t * TT - t * (t - 1) / 2 - t;
#pragma unroll 1
                for (int u = t; u < TT; ++u) s2 += (u == t ? 1.f : 2.f) * wt * wl[u * 64 + co] * part[0][row + u];
            }
            a.bnp[(int64_t)blockIdx.x * TPB * 128 + co] = s1;
            a.bnp[(int64_t)blockIdx.x * TPB * 128 + 64 + co] = s2;
        }
    }
}

// Cout == 64, <= TT taps (the 3x3 one-channel U-Net stem: TT = 9; the generator's 4x4 one-channel stem: TT = 16),
// slope-family activation: FOUR pixels per thread, the 64 output channels in four passes of 16.  The image taps of
// the four pixels are loaded once into registers; each weight vector read from LDS (wave-uniform address) feeds
// 4 x 16 FMAs, so LDS traffic is a quarter of the one-pixel kernel's.
template <int DT, int TT>
__global__ __launch_bounds__(256) void smallcin_fwd64x4_kernel(const SCArgs a) {
    extern __shared__ float sc4_smem[];
    const int T = a.Cin * a.k * a.k, kk = a.k * a.k;
    float* wl = sc4_smem;                           // [TT][64]
    float* stage = sc4_smem + TT * 64;              // [256][17]
    for (int i = threadIdx.x; i < TT * 64; i += 256) {
        const int co = i & 63, tap = i >> 6;
        wl[i] = tap < T ? a.w[co * T + tap] : 0.f;
    }
    const int M = a.N * a.OH * a.OW;                // host guarantees < 2^31
    const int m0 = blockIdx.x * SC_TILE + threadIdx.x;
    float xv[4][TT];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int m = m0 + 256 * k;
        const bool live = m < M;
        const int mm = live ? m : 0;
        const int ox = mm % a.OW;
        const int r = mm / a.OW;
        const int oy = r % a.OH;
        const int n = r / a.OH;
        const float* xn = a.x + (int64_t)n * a.Cin * a.IH * a.IW;
#pragma unroll
        for (int tap = 0; tap < TT; ++tap) {
            const int ci = tap / kk, rr = tap - ci * kk, ky = rr / a.k, kx = rr - ky * a.k;
            const int iy = oy * a.stride - a.pad + ky, ix = ox * a.stride - a.pad + kx;
            const bool ok = live && tap < T && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
            xv[k][tap] = ok ? xn[(ci * a.IH + iy) * a.IW + ix] : 0.f;
        }
    }
    const float slope = a.act == GS_ACT_RELU ? 0.f : (a.act == GS_ACT_LEAKY02 ? 0.2f : 1.f);
    __syncthreads();
    for (int g = 0; g < 4; ++g) {
        float acc[4][16];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[k][c] = 0.f;
#pragma unroll
        for (int tap = 0; tap < TT; ++tap) {
            const float4* wp = reinterpret_cast<const float4*>(wl + tap * 64 + g * 16);
            float w16[16];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const float4 w4 = wp[c4];
                w16[4 * c4] = w4.x; w16[4 * c4 + 1] = w4.y; w16[4 * c4 + 2] = w4.z; w16[4 * c4 + 3] = w4.w;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[k][c] += xv[k][tap] * w16[c];
        }
        float s1[16], s2[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            s1[c] = (acc[0][c] + acc[1][c]) + (acc[2][c] + acc[3][c]);
            s2[c] = (acc[0][c] * acc[0][c] + acc[1][c] * acc[1][c]) + (acc[2][c] * acc[2][c] + acc[3][c] * acc[3][c]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int m = m0 + 256 * k;
            if (m < M) {
                float o[2][8];
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const float v = acc[k][c] + (a.bias ? a.bias[g * 16 + c] : 0.f);
                    o[c >> 3][c & 7] = v > 0.f ? v : v * slope;
                }
                unsigned short* dst = a.y + (int64_t)m * 64 + g * 16;
                st16(dst, pack8<DT>(o[0]));
                st16(dst + 8, pack8<DT>(o[1]));
            }
        }
        if (a.bnp) {
            const int c = threadIdx.x & 15, q = threadIdx.x >> 4;        // 16 groups of 16 threads
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 16; ++i) stage[threadIdx.x * 17 + i] = st == 0 ? s1[i] : s2[i];
                __syncthreads();
                float t = 0.f;
                for (int i = 0; i < 16; ++i) t += stage[(q * 16 + i) * 17 + c];
                __syncthreads();
                stage[q * 16 + c] = t;
                __syncthreads();
                if (threadIdx.x < 16) {
                    float u = 0.f;
#pragma unroll
                    for (int gq = 0; gq < 16; ++gq) u += stage[gq * 16 + threadIdx.x];
                    a.bnp[(int64_t)blockIdx.x * 128 + st * 64 + g * 16 + threadIdx.x] = u;
                }
            }
        }
    }
}

struct SCWArgs {
    const float* x; const unsigned short* dy; float* dw;   // dw: partial slab [nblocks][Cout*T] (stage 1)
    int N, Cin, IH, IW, Cout, OH, OW, k, stride, pad;
    float gscale; int64_t pix_per_block;
};

// TG = taps accumulated per pass (template: 9 keeps the 3x3 / 1-channel stem at ~100 VGPRs)
template <int DT, int TG>
__global__ __launch_bounds__(256) void smallcin_wgrad_kernel(const SCWArgs a) {
    __shared__ float red[256][8];
    const int T = a.Cin * a.k * a.k, kk = a.k * a.k;
    const int nch = a.Cout >> 3, lanes = 256 / nch;
    const int ch = threadIdx.x % nch, pl = threadIdx.x / nch;
    const int M = a.N * a.OH * a.OW;                               // host guarantees < 2^31
    const int m0 = (int)(blockIdx.x * a.pix_per_block);
    const int m1 = m0 + (int)a.pix_per_block < M ? m0 + (int)a.pix_per_block : M;
    const int img = a.Cin * a.IH * a.IW;
    for (int t0 = 0; t0 < T; t0 += TG) {
        float acc[TG][8];
        int tdy[TG], tdx[TG], tco[TG];       // tap decode hoisted out of the pixel loop (integer divisions)
#pragma unroll
        for (int j = 0; j < TG; ++j) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = 0.f;
            const int tap = t0 + j;
            const int ci = tap / kk, rr = tap - ci * kk, ky = rr / a.k, kx = rr - ky * a.k;
            tdy[j] = ky - a.pad; tdx[j] = kx - a.pad; tco[j] = ci * a.IH * a.IW;
        }
        if (pl < lanes) {
            // two pixels per iteration: both gradient rows and all their image taps are in flight together
            for (int mb = m0 + pl; mb < m1; mb += 2 * lanes) {
                const bool two = mb + lanes < m1;
                const int mq[2] = {mb, two ? mb + lanes : mb};
                uint4 rg[2];
                float xv[2][TG];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    rg[q] = *reinterpret_cast<const uint4*>(a.dy + (int64_t)mq[q] * a.Cout + ch * 8);
                    const int ox = mq[q] % a.OW;
                    const int r = mq[q] / a.OW;
                    const int oy = r % a.OH;
                    const int n = r / a.OH;
                    const float* xn = a.x + (int64_t)n * img;
#pragma unroll
                    for (int j = 0; j < TG; ++j) {
                        const int iy = oy * a.stride + tdy[j], ix = ox * a.stride + tdx[j];
                        const bool ok = (t0 + j < T) && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
                        xv[q][j] = ok ? xn[tco[j] + iy * a.IW + ix] : 0.f;
                    }
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (q == 1 && !two) break;
                    float g[8];
                    unpack8<DT>(rg[q], g);
#pragma unroll
                    for (int j = 0; j < TG; ++j)
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc[j][i] += g[i] * xv[q][j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TG; ++j) {
            const int tap = t0 + j;
            if (tap < T) {                           // uniform
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 8; ++i) red[threadIdx.x][i] = acc[j][i];
                __syncthreads();
                // one thread per output channel sums the pixel lanes (64 threads x 32 reads; eight threads doing
                // 256 dependent reads each made this tail as long as the whole pixel loop)
                for (int co = threadIdx.x; co < a.Cout; co += 256) {
                    float s = 0.f;
                    for (int q = 0; q < lanes; ++q) s += red[q * nch + (co >> 3)][co & 7];
                    a.dw[(int64_t)blockIdx.x * a.Cout * T + (int64_t)co * T + tap] = s;
                }
            }
        }
    }
}

// ---- stem backward: BatchNorm/ReLU backward apply + weight gradient of the one-channel 3x3 stem in ONE pass ------------
// (unet_parts.py:16-18 with in_channels = 1, the first stage of `inc`.)  The image needs no gradient, so the gradient
// w.r.t. the stem convolution's output has exactly one consumer -- this weight gradient: it is formed in registers
//   dy = scale * (dz * relu'(scale*y + shift) - c1 - xhat * c2),  xhat = (y - mean) * invstd      (as bn_act_bwd_kernel)
// and multiplied with the nine image taps at once; the 268 MB tensor dy (batch 32, 256^2) is neither written nor read
// back.  The block's strip of the image (its pixels plus one row and one pixel either side) is staged in LDS once; a
// thread owns 8 channels and every 32nd pixel of the block.  Partial slabs [block][64*9] as smallcin_wgrad_kernel.
struct SBWArgs {
    const unsigned short* y; const unsigned short* dz; const float* x;
    const float *scale, *shift, *mean, *invstd, *c1, *c2;
    float* slabs;
    int dz_stride, dz_coff, N, H, W, act;
    int64_t pix_per_block;
    float* partials;         // stem_bwd_onepass_kernel: s1 [block][64]
    int zs = 64;             // stem_bwd_onepass_kernel: pixel stride of the stored activation (128: the hi plane of a pair)
};

// ONE backward pass for the y-free stem.  With g = dz * act'(.), A[c][t] = sum_p g[p][c] x_t(p) and s1_c = sum_p g[p][c]:
//   sum_p g xhat  = invstd_c (sum_t w[c][t] A[c][t] - mean_c s1_c)                      (y = sum_t w_t x_t, exactly)
//   dW[c][t]      = scale_c (A[c][t] - c1_c S[t] - c2_c invstd_c (sum_u w[c][u] G[u][t] - mean_c S[t]))
// with S / G the tap sums / tap Gram matrix of the image (from the forward's statistics pass) and c1 = s1/count,
// c2 = sum g xhat / count.  So while the tensors stream by only the activation's MASK (the sign of the stored z) and dz are
// needed: this kernel accumulates A (slabs [block][576]) and s1 (partials [block][64]); stem_bwd_finalize_kernel does the
// rest -- BatchNorm weight / bias gradients included -- in fp64 on 640 numbers per block.
template <int DT>
__global__ __launch_bounds__(256) void stem_bwd_onepass_kernel(const SBWArgs a) {
    extern __shared__ float sbw_smem[];
    const int M = a.N * a.H * a.W;
    const int ppb = (int)a.pix_per_block;
    const int m0 = blockIdx.x * ppb;
    const int m1 = m0 + ppb < M ? m0 + ppb : M;
    const int nx = ppb + 2 * a.W + 2;
    float* xs = sbw_smem;
    float* red = sbw_smem + ((nx + 3) & ~3);                       // [256][25]
    for (int i = threadIdx.x; i < nx; i += 256) {
        const int idx = m0 - a.W - 1 + i;
        xs[i] = (idx >= 0 && idx < M) ? a.x[idx] : 0.f;
    }
    const int ch = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int c0 = ch * 8;
    const float slope = a.act == GS_ACT_RELU ? 0.f : (a.act == GS_ACT_LEAKY02 ? 0.2f : 1.f);
    float s1[8], acc[9][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s1[i] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[t][i] = 0.f;
    __syncthreads();
    constexpr int UNR = 4;
    for (int mb = m0 + pl; mb < m1; mb += 32 * UNR) {
        uint4 rz[UNR], rg[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const int m = mb + 32 * k;
            const int64_t mm = m < m1 ? m : mb;
            rz[k] = *reinterpret_cast<const uint4*>(a.y + mm * a.zs + c0);          // a.y: the stored activation z here
            rg[k] = *reinterpret_cast<const uint4*>(a.dz + mm * a.dz_stride + a.dz_coff + c0);
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const int m = mb + 32 * k;
            const bool ok = m < m1;
            const int mm = ok ? m : mb;
            const int ox = mm % a.W;
            const int oy = (mm / a.W) % a.H;
            const float* xc = xs + (mm - m0 + a.W + 1);
            float xv[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy_ = t / 3 - 1, dx_ = t % 3 - 1;
                const bool in = (unsigned)(oy + dy_) < (unsigned)a.H && (unsigned)(ox + dx_) < (unsigned)a.W;
                xv[t] = in ? xc[dy_ * a.W + dx_] : 0.f;
            }
            float zv[8], g[8];
            unpack8<DT>(rz[k], zv);
            unpack8<DT>(rg[k], g);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float gh = ok ? g[i] * (zv[i] > 0.f ? 1.f : slope) : 0.f;
                s1[i] += gh;
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[t][i] += gh * xv[t];
            }
        }
    }
    // block sums: s1, then the nine taps three at a time
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) red[threadIdx.x * 25 + i] = s1[i];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int co = threadIdx.x;
        float sum = 0.f;
#pragma unroll 8
        for (int q = 0; q < 32; ++q) sum += red[(q * 8 + (co >> 3)) * 25 + (co & 7)];
        a.partials[(int64_t)blockIdx.x * 64 + co] = sum;
    }
#pragma unroll
    for (int t0 = 0; t0 < 9; t0 += 3) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) red[threadIdx.x * 25 + t * 8 + i] = acc[t0 + t][i];
        __syncthreads();
        if (threadIdx.x < 192) {
            const int t = threadIdx.x >> 6, co = threadIdx.x & 63;
            float sum = 0.f;
#pragma unroll 8
            for (int q = 0; q < 32; ++q) sum += red[(q * 8 + (co >> 3)) * 25 + t * 8 + (co & 7)];
            a.slabs[(int64_t)blockIdx.x * 576 + co * 9 + t0 + t] = sum;
        }
    }
}

// 16 blocks of four channels: each sums the image's 54 tap-sum / Gram totals over the forward tiles (16 lanes per quantity),
// its 36 A entries and 4 s1 entries over the backward blocks (24 lanes per quantity; 8 loads in flight, fixed order, fp64), then
//   s2 = invstd (sum_t w A - mean s1);  c1 = s1/count, c2 = s2/count (0 with eval-mode statistics);
//   dbeta = gscale s1, dgamma = gscale s2 (OVERWRITE);  dW += gscale scale (A - c1 S_t - c2 invstd (sum_u w_u G_ut - mean S_t)).
__global__ __launch_bounds__(1024) void stem_bwd_finalize_kernel(const float* __restrict__ slabs, const float* __restrict__ s1p, int nb,
                                                                 const float* __restrict__ sg, int nsg, const float* __restrict__ w,
                                                                 const float* scale, const float* mean, const float* invstd,
                                                                 double count, int train_stats, float gscale, float* dw,
                                                                 float* dgamma, float* dbeta) {
    constexpr int LS = 16, LA = 24;                             // lanes per tap-sum quantity / per slab quantity (1024 threads)
    __shared__ double sgl[LS][54];
    __shared__ double SG[54];
    __shared__ double red[LA][40];
    __shared__ double tot[40];
    if (threadIdx.x < LS * 54 && train_stats) {
        const int q = threadIdx.x % 54, l = threadIdx.x / 54;
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int b = l;
        for (; b + 7 * LS < nsg; b += 8 * LS) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = sg[(int64_t)(b + LS * u) * 54 + q];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += (double)v[u];
        }
        for (; b < nsg; b += LS) acc[0] += (double)sg[(int64_t)b * 54 + q];
        sgl[l][q] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
    const int cb = blockIdx.x * 4;                              // first channel of the block
    if (threadIdx.x < LA * 40) {
        const int q = threadIdx.x % 40, l = threadIdx.x / 40;
        const float* src = q < 36 ? slabs + cb * 9 + q : s1p + cb + (q - 36);
        const int64_t stride = q < 36 ? 576 : 64;
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int b = l;
        for (; b + 7 * LA < nb; b += 8 * LA) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(b + LA * u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += (double)v[u];
        }
        for (; b < nb; b += LA) acc[0] += (double)src[(int64_t)b * stride];
        red[l][q] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
    __syncthreads();
    if (threadIdx.x < 54) {
        double t = 0.0;
        if (train_stats)
            for (int l = 0; l < LS; ++l) t += sgl[l][threadIdx.x];
        SG[threadIdx.x] = t;
    }
    if (threadIdx.x >= 64 && threadIdx.x < 104) {
        const int q = threadIdx.x - 64;
        double t = 0.0;
        for (int l = 0; l < LA; ++l) t += red[l][q];
        tot[q] = t;
    }
    __syncthreads();
    if (threadIdx.x < 36) {
        const int cl = threadIdx.x / 9, t = threadIdx.x % 9, c = cb + cl;
        const double s1 = tot[36 + cl];
        double wA = 0.0, wg = 0.0;
        for (int u = 0; u < 9; ++u) {
            const double wu = (double)w[c * 9 + u];
            wA += wu * tot[cl * 9 + u];
            const int lo = u < t ? u : t, hi = u < t ? t : u;           // upper triangle, row-major: row lo, column hi
            wg += wu * SG[9 + lo * 9 - lo * (lo - 1) / 2 + (hi - lo)];
        }
        const double is = (double)invstd[c], mu = (double)mean[c];
        const double s2 = is * (wA - mu * s1);
        const double c1 = train_stats ? s1 / count : 0.0, c2 = train_stats ? s2 / count : 0.0;
        const double St = SG[t];
        const double r = tot[cl * 9 + t] - c1 * St - c2 * is * (wg - mu * St);
        dw[c * 9 + t] += (float)((double)gscale * (double)scale[c] * r);
        if (t == 0) {
            if (dbeta) dbeta[c] = (float)(s1 * (double)gscale);
            if (dgamma) dgamma[c] = (float)(s2 * (double)gscale);
        }
    }
}

template <int DT>
__global__ __launch_bounds__(256) void stem_bn_bwd_wgrad_kernel(const SBWArgs a) {
    extern __shared__ float sbw_smem[];
    const int M = a.N * a.H * a.W;                                 // host guarantees < 2^31
    const int ppb = (int)a.pix_per_block;
    const int m0 = blockIdx.x * ppb;
    const int m1 = m0 + ppb < M ? m0 + ppb : M;
    const int nx = ppb + 2 * a.W + 2;
    float* xs = sbw_smem;                                          // xs[i] = image[m0 - W - 1 + i] (flat over N*H*W)
    float* red = sbw_smem + ((nx + 3) & ~3);                       // [256][25]
    for (int i = threadIdx.x; i < nx; i += 256) {
        const int idx = m0 - a.W - 1 + i;
        xs[i] = (idx >= 0 && idx < M) ? a.x[idx] : 0.f;
    }
    const int ch = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int c0 = ch * 8;
    const float slope = a.act == GS_ACT_RELU ? 0.f : (a.act == GS_ACT_LEAKY02 ? 0.2f : 1.f);
    float sc[8], sh[8], mu[8], is[8], k1[8], k2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sc[i] = a.scale[c0 + i]; sh[i] = a.shift[c0 + i]; mu[i] = a.mean[c0 + i];
        is[i] = a.invstd[c0 + i]; k1[i] = a.c1[c0 + i]; k2[i] = a.c2[c0 + i];
    }
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[t][i] = 0.f;
    __syncthreads();
    constexpr int UNR = 4;
    for (int mb = m0 + pl; mb < m1; mb += 32 * UNR) {
        uint4 ry[UNR], rg[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const int m = mb + 32 * k;
            const int64_t mm = m < m1 ? m : mb;
            ry[k] = *reinterpret_cast<const uint4*>(a.y + mm * 64 + c0);
            rg[k] = *reinterpret_cast<const uint4*>(a.dz + mm * a.dz_stride + a.dz_coff + c0);
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const int m = mb + 32 * k;
            const bool ok = m < m1;
            const int mm = ok ? m : mb;
            const int ox = mm % a.W;
            const int oy = (mm / a.W) % a.H;
            const float* xc = xs + (mm - m0 + a.W + 1);
            float xv[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int dy_ = t / 3 - 1, dx_ = t % 3 - 1;
                const bool in = ok && (unsigned)(oy + dy_) < (unsigned)a.H && (unsigned)(ox + dx_) < (unsigned)a.W;
                xv[t] = in ? xc[dy_ * a.W + dx_] : 0.f;
            }
            float yv[8], g[8];
            unpack8<DT>(ry[k], yv);
            unpack8<DT>(rg[k], g);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float v = yv[i] * sc[i] + sh[i];
                const float gh = g[i] * (v > 0.f ? 1.f : slope);
                const float xh = (yv[i] - mu[i]) * is[i];
                const float d = sc[i] * (gh - k1[i] - xh * k2[i]);
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[t][i] += d * xv[t];     // xv = 0 for a masked pixel
            }
        }
    }
    // the 32 pixel lanes of every channel chunk, three taps per pass
#pragma unroll
    for (int t0 = 0; t0 < 9; t0 += 3) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) red[threadIdx.x * 25 + t * 8 + i] = acc[t0 + t][i];
        __syncthreads();
        if (threadIdx.x < 192) {                 // (tap of the pass, channel): sums the 32 pixel lanes
            const int t = threadIdx.x >> 6, co = threadIdx.x & 63;
            float sum = 0.f;
#pragma unroll 8
            for (int q = 0; q < 32; ++q) sum += red[(q * 8 + (co >> 3)) * 25 + t * 8 + (co & 7)];
            a.slabs[(int64_t)blockIdx.x * 576 + co * 9 + t0 + t] = sum;
        }
    }
}

struct SCDArgs {
    const unsigned short* dy; const float* w; float* dx;
    int N, Cin, IH, IW, Cout, OH, OW, k, stride, pad; float gscale;
};

template <int DT>
__global__ __launch_bounds__(256) void smallcin_dgrad_kernel(const SCDArgs a) {
    __shared__ float wl[SC_MAX_W];     // [ci][kk][co]
    const int kk = a.k * a.k, T = a.Cin * kk;
    for (int i = threadIdx.x; i < T * a.Cout; i += 256) {
        const int co = i % a.Cout, tap = i / a.Cout;   // tap = ci*kk + ky*k + kx
        wl[i] = a.w[(int64_t)co * T + tap];
    }
    __syncthreads();
    const int total = a.N * a.IH * a.IW;                 // host guarantees < 2^31
    for (int p = blockIdx.x * 256 + threadIdx.x; p < total; p += gridDim.x * 256) {
        const int ix = p % a.IW;
        const int r = p / a.IW;
        const int iy = r % a.IH;
        const int n = r / a.IH;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < a.k; ++ky) {
            const int ty = iy + a.pad - ky;
            if (ty < 0 || ty % a.stride) continue;
            const int oy = ty / a.stride;
            if (oy >= a.OH) continue;
            for (int kx = 0; kx < a.k; ++kx) {
                const int tx = ix + a.pad - kx;
                if (tx < 0 || tx % a.stride) continue;
                const int ox = tx / a.stride;
                if (ox >= a.OW) continue;
                const unsigned short* dp = a.dy + (((int64_t)n * a.OH + oy) * a.OW + ox) * a.Cout;
                for (int c0 = 0; c0 < a.Cout; c0 += 8) {
                    float g[8];
                    unpack8<DT>(*reinterpret_cast<const uint4*>(dp + c0), g);
                    for (int ci = 0; ci < a.Cin; ++ci) {
                        const float* wp = wl + ((ci * kk) + ky * a.k + kx) * a.Cout + c0;
                        float s = 0.f;
#pragma unroll
                        for (int i = 0; i < 8; ++i) s += g[i] * wp[i];
                        acc[ci] += s;
                    }
                }
            }
        }
        for (int ci = 0; ci < a.Cin; ++ci)
            a.dx[(((int64_t)n * a.Cin + ci) * a.IH + iy) * a.IW + ix] = acc[ci] * a.gscale;
    }
}

// ---- small-Cout head ---------------------------------------------------------------------------------
struct HArgs {
    const unsigned short* x; const float* w; const float* bias; float* y;
    const float* dy; unsigned short* dx; float* dw; float* db;
    int N, IH, IW, Cin, Cout, OH, OW, k, stride, pad; float gscale; int64_t pix_per_block;
    // head1x1 kernels only: x is the convolution OUTPUT of the last stage and z = act(x * bn_scale + bn_shift) is formed on
    // the load path (gs_head1x1_bn_fwd / gs_head1x1_bn_wgrad): the activation tensor is neither written nor read
    const float* bn_scale; const float* bn_shift; float bn_slope;
};

// the last stage's BatchNorm + slope activation on eight loaded channels (no-op without coefficients)
__device__ __forceinline__ void head_bn_act(const HArgs& a, const float* sc, const float* sh, float* v) {
    if (a.bn_scale != nullptr) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float t = v[i] * sc[i] + sh[i];
            v[i] = t > 0.f ? t : t * a.bn_slope;
        }
    }
}

// wl layout [c][tap][Cin]
__device__ __forceinline__ void load_head_weights(const HArgs& a, float* wl) {
    const int kk = a.k * a.k;
    for (int i = threadIdx.x; i < a.Cout * kk * a.Cin; i += 256) {
        const int ci = i % a.Cin;
        const int r = i / a.Cin;
        const int tap = r % kk, c = r / kk;
        wl[i] = a.w[((int64_t)c * a.Cin + ci) * kk + tap];
    }
}

template <int DT>
__global__ __launch_bounds__(256) void smallcout_fwd_kernel(const HArgs a) {
    __shared__ float wl[SC_MAX_W];
    load_head_weights(a, wl);
    __syncthreads();
    const int kk = a.k * a.k;
    const int nch = a.Cin >> 3;
    const int gl = nch < 64 ? nch : 64;           // lanes cooperating on one pixel (power of two)
    const int groups = 256 / gl;
    const int li = threadIdx.x % gl, grp = threadIdx.x / gl;
    const int M = a.N * a.OH * a.OW;                     // host guarantees < 2^31
    const int nit = (M + groups - 1) / groups;
    for (int it = blockIdx.x; it < nit; it += gridDim.x) {
        const int m = it * groups + grp;
        const bool valid = m < M;
        const int mm = valid ? m : 0;
        const int ox = mm % a.OW;
        const int r = mm / a.OW;
        const int oy = r % a.OH;
        const int n = r / a.OH;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < a.k; ++ky)
            for (int kx = 0; kx < a.k; ++kx) {
                const int iy = oy * a.stride - a.pad + ky, ix = ox * a.stride - a.pad + kx;
                if (!valid || (unsigned)iy >= (unsigned)a.IH || (unsigned)ix >= (unsigned)a.IW) continue;
                const unsigned short* xp = a.x + (((int64_t)n * a.IH + iy) * a.IW + ix) * a.Cin;
                for (int ch = li; ch < nch; ch += gl) {
                    float v[8];
                    unpack8<DT>(*reinterpret_cast<const uint4*>(xp + ch * 8), v);
                    for (int c = 0; c < a.Cout; ++c) {
                        const float* wp = wl + (c * kk + ky * a.k + kx) * a.Cin + ch * 8;
                        float s = 0.f;
#pragma unroll
                        for (int i = 0; i < 8; ++i) s += v[i] * wp[i];
                        acc[c] += s;
                    }
                }
            }
        for (int c = 0; c < a.Cout; ++c) {
            float s = acc[c];
            for (int o = gl >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (li == 0 && valid)
                a.y[(((int64_t)n * a.Cout + c) * a.OH + oy) * a.OW + ox] = s + (a.bias ? a.bias[c] : 0.f);
        }
    }
}

template <int DT>
__global__ __launch_bounds__(256) void smallcout_dgrad_kernel(const HArgs a) {
    __shared__ float wl[SC_MAX_W];
    load_head_weights(a, wl);
    __syncthreads();
    const int kk = a.k * a.k;
    const int nch = a.Cin >> 3;
    const int total = a.N * a.IH * a.IW * nch;           // host guarantees < 2^31
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int ch = idx % nch;
        int p = idx / nch;
        const int ix = (int)(p % a.IW); p /= a.IW;
        const int iy = (int)(p % a.IH);
        const int n = (int)(p / a.IH);
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        for (int ky = 0; ky < a.k; ++ky) {
            const int ty = iy + a.pad - ky;
            if (ty < 0 || ty % a.stride) continue;
            const int oy = ty / a.stride;
            if (oy >= a.OH) continue;
            for (int kx = 0; kx < a.k; ++kx) {
                const int tx = ix + a.pad - kx;
                if (tx < 0 || tx % a.stride) continue;
                const int ox = tx / a.stride;
                if (ox >= a.OW) continue;
                for (int c = 0; c < a.Cout; ++c) {
                    const float g = a.dy[(((int64_t)n * a.Cout + c) * a.OH + oy) * a.OW + ox];
                    const float* wp = wl + (c * kk + ky * a.k + kx) * a.Cin + ch * 8;
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] += g * wp[i];
                }
            }
        }
        const int64_t pix = ((int64_t)n * a.IH + iy) * a.IW + ix;
        *reinterpret_cast<uint4*>(a.dx + pix * a.Cin + ch * 8) = pack8<DT>(acc);
    }
}

// ---- pointwise head, Cin == 64 (OutConv of the U-Net, unet_parts.py:74): 8 lanes per pixel (one 16-byte chunk
// each: a wave reads 1 KB contiguous), the <= 4 x 8 weights of a lane's chunk live in registers, four pixels per lane
// are in flight, the 8-lane dot-product reduction is DPP only (quad_perm / row_half_mirror: no LDS round trips).
__device__ __forceinline__ float sum8_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    return v;
}

template <int DT>
__global__ __launch_bounds__(256) void head1x1_fwd_kernel(const HArgs a) {
    const int ch = threadIdx.x & 7, pl = threadIdx.x >> 3;           // 32 pixels per block pass
    float w[4][8], bv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        bv[c] = (c < a.Cout && a.bias) ? a.bias[c] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) w[c][i] = c < a.Cout ? a.w[c * 64 + ch * 8 + i] : 0.f;
    }
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sc[i] = a.bn_scale ? a.bn_scale[ch * 8 + i] : 1.f;
        sh[i] = a.bn_scale ? a.bn_shift[ch * 8 + i] : 0.f;
    }
    const int M = a.N * a.OH * a.OW, ohw = a.OH * a.OW;              // host guarantees < 2^31
    constexpr int UNR = 4;
    for (int mb = blockIdx.x * (32 * UNR) + pl; mb < M; mb += gridDim.x * (32 * UNR)) {
        uint4 r[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int m = mb + 32 * u;
            r[u] = *reinterpret_cast<const uint4*>(a.x + (int64_t)(m < M ? m : mb) * 64 + ch * 8);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int m = mb + 32 * u;
            float v[8], s[4];
            unpack8<DT>(r[u], v);
            head_bn_act(a, sc, sh, v);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) t += v[i] * w[c][i];
                s[c] = sum8_dpp(t);
            }
            if (ch == 0 && m < M) {
                const int n = m / ohw, hw = m - n * ohw;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < a.Cout) a.y[((int64_t)n * a.Cout + c) * ohw + hw] = s[c] + bv[c];
            }
        }
    }
}

template <int DT>
__global__ __launch_bounds__(256) void head1x1_dgrad_kernel(const HArgs a) {
    const int ch = threadIdx.x & 7, pl = threadIdx.x >> 3;
    float w[4][8];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) w[c][i] = c < a.Cout ? a.w[c * 64 + ch * 8 + i] : 0.f;
    const int M = a.N * a.IH * a.IW, hw_n = a.IH * a.IW;
    constexpr int UNR = 4;
    for (int mb = blockIdx.x * (32 * UNR) + pl; mb < M; mb += gridDim.x * (32 * UNR)) {
        float g[UNR][4];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int m = mb + 32 * u < M ? mb + 32 * u : mb;
            const int n = m / hw_n, hw = m - n * hw_n;
#pragma unroll
            for (int c = 0; c < 4; ++c) g[u][c] = c < a.Cout ? a.dy[((int64_t)n * a.Cout + c) * hw_n + hw] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int m = mb + 32 * u;
            if (m >= M) continue;
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = g[u][0] * w[0][i] + g[u][1] * w[1][i] + g[u][2] * w[2][i] + g[u][3] * w[3][i];
            *reinterpret_cast<uint4*>(a.dx + (int64_t)m * 64 + ch * 8) = pack8<DT>(o);
        }
    }
}

// weight + bias gradient of the pointwise head (Cin == 64): same slab layout as smallcout_wgrad_kernel
template <int DT>
__global__ __launch_bounds__(256) void head1x1_wgrad_kernel(const HArgs a) {
    __shared__ float red[256][8];
    __shared__ float redb[4][4];
    const int ch = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int M = a.N * a.OH * a.OW, ohw = a.OH * a.OW;
    const int m0 = (int)(blockIdx.x * a.pix_per_block);
    const int m1 = m0 + (int)a.pix_per_block < M ? m0 + (int)a.pix_per_block : M;
    float acc[4][8], sb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[c][i] = 0.f;
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sc[i] = a.bn_scale ? a.bn_scale[ch * 8 + i] : 1.f;
        sh[i] = a.bn_scale ? a.bn_shift[ch * 8 + i] : 0.f;
    }
    constexpr int UNR = 4;
    for (int mb = m0 + pl; mb < m1; mb += 32 * UNR) {
        uint4 r[UNR];
        float g[UNR][4];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int m = mb + 32 * u;
            const bool ok = m < m1;
            const int mm = ok ? m : mb;
            r[u] = *reinterpret_cast<const uint4*>(a.x + (int64_t)mm * 64 + ch * 8);
            const int n = mm / ohw, hw = mm - n * ohw;
#pragma unroll
            for (int c = 0; c < 4; ++c) g[u][c] = (ok && c < a.Cout) ? a.dy[((int64_t)n * a.Cout + c) * ohw + hw] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float v[8];
            unpack8<DT>(r[u], v);
            head_bn_act(a, sc, sh, v);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                sb[c] += g[u][c];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[c][i] += g[u][c] * v[i];
            }
        }
    }
    if (a.db) {                                   // every 8-lane group saw each pixel once: lanes ch == 0 carry the bias sums
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float s = wave_sum(ch == 0 ? sb[c] : 0.f);
            if ((threadIdx.x & 63) == 0) redb[c][threadIdx.x >> 6] = s;
        }
        __syncthreads();
        if (threadIdx.x < 4) a.db[(int64_t)blockIdx.x * 4 + threadIdx.x] =
            redb[threadIdx.x][0] + redb[threadIdx.x][1] + redb[threadIdx.x][2] + redb[threadIdx.x][3];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c < a.Cout) {                         // uniform
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) red[threadIdx.x][i] = acc[c][i];
            __syncthreads();
            if (threadIdx.x < 64) {                  // one thread per input channel sums the 32 pixel lanes
                const int ci = threadIdx.x;
                float t = 0.f;
                for (int q = 0; q < 32; ++q) t += red[q * 8 + (ci >> 3)][ci & 7];
                a.dw[(int64_t)blockIdx.x * a.Cout * 64 + c * 64 + ci] = t;
            }
        }
    }
}

template <int DT>
__global__ __launch_bounds__(256) void smallcout_wgrad_kernel(const HArgs a) {
    __shared__ float red[256][8];
    __shared__ float redb[4][4];
    const int kk = a.k * a.k;
    const int nch = a.Cin >> 3;
    const int lpu = nch < 256 ? nch : 256, lanes = 256 / lpu;
    const int chl = threadIdx.x % lpu, pl = threadIdx.x / lpu;
    const int M = a.N * a.OH * a.OW;                     // host guarantees < 2^31
    const int m0 = (int)(blockIdx.x * a.pix_per_block);
    const int m1 = m0 + (int)a.pix_per_block < M ? m0 + (int)a.pix_per_block : M;
    const int ohw = a.OH * a.OW;
    // bias gradient: plain block reduction over this block's pixels
    if (a.db) {
        for (int c = 0; c < a.Cout; ++c) {
            float s = 0.f;
            for (int m = m0 + threadIdx.x; m < m1; m += 256) {
                const int hw = m % ohw, n = m / ohw;
                s += a.dy[((int64_t)n * a.Cout + c) * ohw + hw];
            }
            s = wave_sum(s);
            __syncthreads();
            if ((threadIdx.x & 63) == 0) redb[c][threadIdx.x >> 6] = s;
            __syncthreads();
            if (threadIdx.x == 0) a.db[(int64_t)blockIdx.x * 4 + c] = redb[c][0] + redb[c][1] + redb[c][2] + redb[c][3];
        }
    }
    for (int tap = 0; tap < kk; ++tap) {
        const int ky = tap / a.k, kx = tap - ky * a.k;
        for (int ch = chl; ch < nch; ch += lpu) {       // one trip for Cin <= 2048
            float acc[4][8];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[c][i] = 0.f;
            if (pl < lanes) {
                for (int m = m0 + pl; m < m1; m += lanes) {
                    const int ox = m % a.OW;
                    const int r = m / a.OW;
                    const int oy = r % a.OH;
                    const int n = r / a.OH;
                    const int iy = oy * a.stride - a.pad + ky, ix = ox * a.stride - a.pad + kx;
                    if ((unsigned)iy >= (unsigned)a.IH || (unsigned)ix >= (unsigned)a.IW) continue;
                    float v[8];
                    unpack8<DT>(*reinterpret_cast<const uint4*>(a.x + (((int64_t)n * a.IH + iy) * a.IW + ix) * a.Cin + ch * 8), v);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (c < a.Cout) {
                            const float g = a.dy[(((int64_t)n * a.Cout + c) * a.OH + oy) * a.OW + ox];
#pragma unroll
                            for (int i = 0; i < 8; ++i) acc[c][i] += g * v[i];
                        }
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c < a.Cout) {                    // uniform
                    __syncthreads();
#pragma unroll
                    for (int i = 0; i < 8; ++i) red[threadIdx.x][i] = acc[c][i];
                    __syncthreads();
                    for (int xt = threadIdx.x; xt < lpu * 8; xt += 256) {   // one thread per input channel of this trip
                        const int cl = xt >> 3, i = xt & 7;
                        const int chx = ch - chl + cl;         // chunk handled by lane cl in this trip
                        if (chx < nch) {
                            float s = 0.f;
                            for (int q = 0; q < lanes; ++q) s += red[q * lpu + cl][i];
                            a.dw[(int64_t)blockIdx.x * a.Cout * a.Cin * kk + ((int64_t)c * a.Cin + chx * 8 + i) * kk + tap] = s;
                        }
                    }
                }
            }
        }
    }
}

// stage 2 of the direct weight gradients: out[j] += gscale * sum_b slab[b*stride + j]  (fixed order)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, int nb, int64_t stride, int n,
                                                          float gscale, float* out) {
    __shared__ double red[8][32];
    const int jl = threadIdx.x & 31, bl = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + jl;
    double s = 0.0;
    if (j < n) {
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // eight independent loads in flight per lane
        int b = bl;
        for (; b + 56 < nb; b += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slab[(int64_t)(b + 8 * u) * stride + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += (double)v[u];
        }
        for (; b < nb; b += 8) a[0] += (double)slab[(int64_t)b * stride + j];
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    red[bl][jl] = s;
    __syncthreads();
    if (bl == 0 && j < n) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i][jl];
        out[j] += (float)(t * gscale);
    }
}

}  // namespace

// pixels per block of the two slab-writing weight-gradient kernels (stem and head): many blocks, each writes its partial
// slab, a second kernel sums the slabs in order.  GSSEG_DIRECT_WGRAD_BLOCKS overrides the block count.
// (measured at batch 32, 256^2: stem weight gradient 136 / 124 / 145 us and head weight gradient 73 / 64 / 69 us with
// 2048 / 1024 / 512 blocks; the fused stem backward 136 / 127 / 118 us -- GSSEG_STEM_BWD_BLOCKS, default 512.)
static int64_t direct_wgrad_ppb(int64_t M, int64_t min_ppb = 64) {
    static const int blocks = getenv("GSSEG_DIRECT_WGRAD_BLOCKS") ? atoi(getenv("GSSEG_DIRECT_WGRAD_BLOCKS")) : 1024;
    int64_t ppb = cdiv64(M, blocks > 0 ? blocks : 1024);
    // (min_ppb = 8 for the few-output-channel weight gradient: few pixels -- the PatchGAN's last layer at batch 2: 1,800 --
    // still spread over ~250 blocks; with 64 pixels per block 29 blocks walked 16 taps x 16 dependent loads each, 124 us)
    if (ppb < min_ppb) ppb = min_ppb;
    return ppb;
}
static int64_t stem_bwd_ppb(int64_t M) {
    static const int blocks = getenv("GSSEG_STEM_BWD_BLOCKS") ? atoi(getenv("GSSEG_STEM_BWD_BLOCKS")) : 512;
    int64_t ppb = cdiv64(M, blocks > 0 ? blocks : 512);
    if (ppb < 64) ppb = 64;
    return ppb;
}

extern "C" int64_t gs_conv_direct_wgrad_ws_floats(int N, int OH, int OW, int Cin, int Cout, int k) {
    const int64_t M = (int64_t)N * OH * OW;
    const int64_t nb1 = cdiv64(M, direct_wgrad_ppb(M, 8)), nb2 = cdiv64(M, stem_bwd_ppb(M));
    return (nb1 > nb2 ? nb1 : nb2) * ((int64_t)Cout * Cin * k * k + 4);
}

extern "C" int gs_conv_smallcin_mtiles(int N, int OH, int OW) {
    return (int)cdiv64((int64_t)N * OH * OW, SC_TILE);
}

static int check_smallcin(const char* who, int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int k, int stride,
                          int pad, int dtype) {
    GS_CHECK_ARG(N > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0, "%s: bad dims", who);
    GS_CHECK_ARG(Cin >= 1 && Cin <= 4, "%s: Cin=%d must be 1..4", who, Cin);
    GS_CHECK_ARG(Cout % 8 == 0 && Cout >= 8 && 256 % (Cout / 8) == 0, "%s: Cout=%d must be 8*2^j", who, Cout);
    GS_CHECK_ARG(k >= 1 && stride >= 1 && pad >= 0 && Cin * k * k * Cout <= SC_MAX_W, "%s: weights exceed %d floats", who, SC_MAX_W);
    GS_CHECK_ARG(OH == (IH + 2 * pad - k) / stride + 1 && OW == (IW + 2 * pad - k) / stride + 1, "%s: output size mismatch", who);
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "%s: bad dtype", who);
    return GS_OK;
}

extern "C" int gs_conv_smallcin_fwd(const float* x, const float* w, const float* bias, void* y, float* bn_partials,
                                    int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int k, int stride,
                                    int pad, int act, int dtype, void* stream) {
    int rc = check_smallcin("gs_conv_smallcin_fwd", N, Cin, IH, IW, Cout, OH, OW, k, stride, pad, dtype);
    if (rc) return rc;
    GS_CHECK_ARG(x && w && y, "gs_conv_smallcin_fwd: null pointer");
    GS_CHECK_ARG(!(bias && bn_partials), "gs_conv_smallcin_fwd: BatchNorm partials are taken before the bias; pass only one");
    SCArgs a{x, w, bias, (unsigned short*)y, bn_partials, N, Cin, IH, IW, Cout, OH, OW, k, stride, pad, act};
    const int nb = gs_conv_smallcin_mtiles(N, OH, OW);
    hipStream_t s = (hipStream_t)stream;
    GS_CHECK_ARG((int64_t)N * OH * OW < 2147483647LL && (int64_t)Cin * IH * IW < 2147483647LL, "gs_conv_smallcin_fwd: too many pixels");
    const size_t lds = ((size_t)Cin * k * k * 64 + 256 * SC64_PAD) * sizeof(float);
    const int T = Cin * k * k;
    static const int line_env = getenv("GSSEG_STEM_LINE") ? atoi(getenv("GSSEG_STEM_LINE")) : 1;
    if (line_env && Cout == 64 && Cin == 1 && k == 3 && stride == 1 && pad == 1 && act != GS_ACT_TANH) {
        if (dtype == GS_F16) smallcin_fwd64_line_kernel<GS_F16, 0><<<nb, 256, 0, s>>>(a);
        else smallcin_fwd64_line_kernel<GS_BF16, 0><<<nb, 256, 0, s>>>(a);
    } else if (Cout == 64 && T <= 16 && act != GS_ACT_TANH) {
        const int tt = T <= 9 ? 9 : 16;
        const size_t lds4 = ((size_t)tt * 64 + 256 * 17) * sizeof(float);
        if (dtype == GS_F16) {
            if (tt == 9) smallcin_fwd64x4_kernel<GS_F16, 9><<<nb, 256, lds4, s>>>(a);
            else smallcin_fwd64x4_kernel<GS_F16, 16><<<nb, 256, lds4, s>>>(a);
        } else {
            if (tt == 9) smallcin_fwd64x4_kernel<GS_BF16, 9><<<nb, 256, lds4, s>>>(a);
            else smallcin_fwd64x4_kernel<GS_BF16, 16><<<nb, 256, lds4, s>>>(a);
        }
    } else if (Cout == 64 && lds <= 64 * 1024) {
        if (dtype == GS_F16) smallcin_fwd64_kernel<GS_F16><<<nb, 256, lds, s>>>(a);
        else smallcin_fwd64_kernel<GS_BF16><<<nb, 256, lds, s>>>(a);
    } else {
        if (dtype == GS_F16) smallcin_fwd_kernel<GS_F16><<<nb, 256, 0, s>>>(a);
        else smallcin_fwd_kernel<GS_BF16><<<nb, 256, 0, s>>>(a);
    }
    GS_CHECK_LAUNCH("gs_conv_smallcin_fwd");
    return GS_OK;
}

extern "C" int gs_conv_smallcin_wgrad(const float* x, const void* dy, float* dw, float* ws, int N, int Cin, int IH,
                                      int IW, int Cout, int OH, int OW, int k, int stride, int pad, float gscale,
                                      int dtype, void* stream) {
    int rc = check_smallcin("gs_conv_smallcin_wgrad", N, Cin, IH, IW, Cout, OH, OW, k, stride, pad, dtype);
    if (rc) return rc;
    GS_CHECK_ARG(x && dy && dw && ws, "gs_conv_smallcin_wgrad: null pointer");
    const int64_t M = (int64_t)N * OH * OW;
    const int64_t ppb = direct_wgrad_ppb(M);
    SCWArgs a{x, (const unsigned short*)dy, ws, N, Cin, IH, IW, Cout, OH, OW, k, stride, pad, gscale, ppb};
    const int nb = (int)cdiv64(M, ppb);
    hipStream_t s = (hipStream_t)stream;
    GS_CHECK_ARG(M < 2147483647LL && (int64_t)Cin * IH * IW < 2147483647LL, "gs_conv_smallcin_wgrad: too many pixels");
    const int T = Cin * k * k;
    if (T <= 9) {
        if (dtype == GS_F16) smallcin_wgrad_kernel<GS_F16, 9><<<nb, 256, 0, s>>>(a);
        else smallcin_wgrad_kernel<GS_BF16, 9><<<nb, 256, 0, s>>>(a);
    } else {
        if (dtype == GS_F16) smallcin_wgrad_kernel<GS_F16, 16><<<nb, 256, 0, s>>>(a);
        else smallcin_wgrad_kernel<GS_BF16, 16><<<nb, 256, 0, s>>>(a);
    }
    const int n = Cout * Cin * k * k;
    slab_reduce_kernel<<<cdiv(n, 32), 256, 0, s>>>(ws, nb, n, n, gscale, dw);
    GS_CHECK_LAUNCH("gs_conv_smallcin_wgrad");
    return GS_OK;
}

extern "C" int gs_stem_bn_bwd_wgrad(const void* y, const void* dz, int dz_stride, int dz_coff, const float* x,
                                    const float* scale, const float* shift, const float* mean, const float* invstd,
                                    const float* c1, const float* c2, int act, float* dw, float* ws, int N, int H, int W,
                                    float gscale, int dtype, void* stream) {
    GS_CHECK_ARG(y && dz && x && scale && shift && mean && invstd && c1 && c2 && dw && ws, "gs_stem_bn_bwd_wgrad: null pointer");
    GS_CHECK_ARG(N > 0 && H > 0 && W > 0 && (int64_t)N * H * W < 2147483647LL / 64, "gs_stem_bn_bwd_wgrad: bad dims");
    GS_CHECK_ARG(dz_stride % 8 == 0 && dz_coff % 8 == 0 && dz_stride >= dz_coff + 64, "gs_stem_bn_bwd_wgrad: bad gradient layout");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_stem_bn_bwd_wgrad: bad dtype");
    GS_CHECK_ARG(act == GS_ACT_NONE || act == GS_ACT_RELU || act == GS_ACT_LEAKY02, "gs_stem_bn_bwd_wgrad: bad activation");
    const int64_t M = (int64_t)N * H * W;
    const int64_t ppb = stem_bwd_ppb(M);
    const size_t lds = ((size_t)((ppb + 2 * W + 2 + 3) & ~(int64_t)3) + 256 * 25) * sizeof(float);
    if (lds > 64 * 1024) return GS_EUNSUPPORTED;        // very wide images: the caller runs the two-kernel path
    SBWArgs a{(const unsigned short*)y, (const unsigned short*)dz, x, scale, shift, mean, invstd, c1, c2, ws,
              dz_stride, dz_coff, N, H, W, act, ppb};
    const int nb = (int)cdiv64(M, ppb);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GS_F16) stem_bn_bwd_wgrad_kernel<GS_F16><<<nb, 256, lds, s>>>(a);
    else stem_bn_bwd_wgrad_kernel<GS_BF16><<<nb, 256, lds, s>>>(a);
    slab_reduce_kernel<<<cdiv(576, 32), 256, 0, s>>>(ws, nb, 576, 576, gscale, dw);
    GS_CHECK_LAUNCH("gs_stem_bn_bwd_wgrad");
    return GS_OK;
}

// ---- the stem without its convolution output in memory (train-mode BatchNorm, one input channel, 64 outputs) ----------
static int stem_check(const char* who, const float* x, const float* w, int N, int H, int W, int dtype) {
    GS_CHECK_ARG(x && w && N > 0 && H > 0 && W > 0 && (int64_t)N * H * W < 2147483647LL / 64, "%s: bad arguments", who);
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "%s: bad dtype", who);
    return GS_OK;
}

extern "C" int gs_stem_stats(const float* x, const float* w, float* bn_partials, float* tap_sums, int N, int H, int W,
                             void* stream) {
    int rc = stem_check("gs_stem_stats", x, w, N, H, W, GS_F16);
    if (rc) return rc;
    GS_CHECK_ARG(bn_partials != nullptr, "gs_stem_stats: null partials");
    SCArgs a{x, w, nullptr, nullptr, bn_partials, N, 1, H, W, 64, H, W, 3, 1, 1, GS_ACT_NONE};
    a.sg = tap_sums;
    smallcin_fwd64_line_kernel<GS_F16, 1><<<cdiv(gs_conv_smallcin_mtiles(N, H, W), SC_GROUP), 256, 0, (hipStream_t)stream>>>(a);
    GS_CHECK_LAUNCH("gs_stem_stats");
    return GS_OK;
}

extern "C" int gs_stem_fwd_bn(const float* x, const float* w, const float* bn_scale, const float* bn_shift, int act, void* z,
                              int N, int H, int W, int dtype, void* stream) {
    int rc = stem_check("gs_stem_fwd_bn", x, w, N, H, W, dtype);
    if (rc) return rc;
    GS_CHECK_ARG(bn_scale && bn_shift && z, "gs_stem_fwd_bn: null pointer");
    GS_CHECK_ARG(act == GS_ACT_NONE || act == GS_ACT_RELU || act == GS_ACT_LEAKY02, "gs_stem_fwd_bn: activation %d not supported", act);
    SCArgs a{x, w, nullptr, (unsigned short*)z, nullptr, N, 1, H, W, 64, H, W, 3, 1, 1, act};
    a.bn_scale = bn_scale; a.bn_shift = bn_shift;
    const int nb = gs_conv_smallcin_mtiles(N, H, W);
    if (dtype == GS_F16) smallcin_fwd64_line_kernel<GS_F16, 2><<<nb, 256, 0, (hipStream_t)stream>>>(a);
    else smallcin_fwd64_line_kernel<GS_BF16, 2><<<nb, 256, 0, (hipStream_t)stream>>>(a);
    GS_CHECK_LAUNCH("gs_stem_fwd_bn");
    return GS_OK;
}

// Pair form (UNet(precise=...)): z_hi / z_lo with pixel stride z_pix_stride (e.g. the two planes of a [hi | lo] buffer); z_lo may be NULL.
static int stem_fwd_bn_pair_impl(const float* x, const float* w, const float* bn_scale, const float* bn_shift, int act,
                                 void* z_hi, void* z_lo, int z_pix_stride, int N, int H, int W, int dtype, void* stream, int lo_q8);
extern "C" int gs_stem_fwd_bn_pair(const float* x, const float* w, const float* bn_scale, const float* bn_shift, int act,
                                   void* z_hi, void* z_lo, int z_pix_stride, int N, int H, int W, int dtype, void* stream) {
    return stem_fwd_bn_pair_impl(x, w, bn_scale, bn_shift, act, z_hi, z_lo, z_pix_stride, N, H, W, dtype, stream, 0);
}
// the same with the lo plane as a Q PLANE (z_q: byte 0 = channel 0; the consumer runs a "q" stage)
extern "C" int gs_stem_fwd_bn_pair_q8(const float* x, const float* w, const float* bn_scale, const float* bn_shift, int act,
                                      void* z_hi, void* z_q, int z_pix_stride, int N, int H, int W, int dtype, void* stream) {
    GS_CHECK_ARG(z_q != nullptr, "gs_stem_fwd_bn_pair_q8: z_q is NULL");
    return stem_fwd_bn_pair_impl(x, w, bn_scale, bn_shift, act, z_hi, z_q, z_pix_stride, N, H, W, dtype, stream, 1);
}
static int stem_fwd_bn_pair_impl(const float* x, const float* w, const float* bn_scale, const float* bn_shift, int act,
                                 void* z_hi, void* z_lo, int z_pix_stride, int N, int H, int W, int dtype, void* stream, int lo_q8) {
    int rc = stem_check("gs_stem_fwd_bn_pair", x, w, N, H, W, dtype);
    if (rc) return rc;
    GS_CHECK_ARG(bn_scale && bn_shift && z_hi && z_pix_stride >= 64 && z_pix_stride % 8 == 0, "gs_stem_fwd_bn_pair: bad arguments");
    GS_CHECK_ARG(act == GS_ACT_NONE || act == GS_ACT_RELU || act == GS_ACT_LEAKY02, "gs_stem_fwd_bn_pair: activation %d not supported", act);
    SCArgs a{x, w, nullptr, (unsigned short*)z_hi, nullptr, N, 1, H, W, 64, H, W, 3, 1, 1, act};
    a.bn_scale = bn_scale; a.bn_shift = bn_shift; a.y_lo = (unsigned short*)z_lo; a.ys = z_pix_stride; a.lo_q8 = lo_q8;
    const int nb = gs_conv_smallcin_mtiles(N, H, W);
    const bool rows256_off = getenv("GSSEG_STEM_ROWS256_OFF") != nullptr;   // diagnostics: the lane-by-lane stores (read per call: tools/experiments/stem_pair_time.py switches in one process)
    a.rows256 = z_lo != nullptr && z_pix_stride == 128 && (unsigned short*)z_lo == (unsigned short*)z_hi + 64 &&
                ((uintptr_t)z_hi & 15) == 0 && !rows256_off;
    if (dtype == GS_F16) smallcin_fwd64_line_kernel<GS_F16, 2><<<nb, 256, 0, (hipStream_t)stream>>>(a);
    else smallcin_fwd64_line_kernel<GS_BF16, 2><<<nb, 256, 0, (hipStream_t)stream>>>(a);
    GS_CHECK_LAUNCH("gs_stem_fwd_bn_pair");
    return GS_OK;
}

extern "C" int gs_stem_bwd_tiles(int N, int H, int W) {
    const int64_t M = (int64_t)N * H * W;
    return (N > 0 && H > 0 && W > 0) ? (int)cdiv64(M, stem_bwd_ppb(M)) : 0;
}

static int stem_bwd_onepass_impl(const float* x, const void* z, int z_stride, const void* dz, int dz_stride, int dz_coff, int act,
                                 float* s1_partials, float* ws, int N, int H, int W, int dtype, void* stream);
extern "C" int gs_stem_bwd_onepass(const float* x, const void* z, const void* dz, int dz_stride, int dz_coff, int act,
                                   float* s1_partials, float* ws, int N, int H, int W, int dtype, void* stream) {
    return stem_bwd_onepass_impl(x, z, 64, dz, dz_stride, dz_coff, act, s1_partials, ws, N, H, W, dtype, stream);
}
// the same with the stored activation at a pixel stride (the hi plane of a [hi | lo] pair buffer)
extern "C" int gs_stem_bwd_onepass_strided(const float* x, const void* z, int z_pix_stride, const void* dz, int dz_stride,
                                           int dz_coff, int act, float* s1_partials, float* ws, int N, int H, int W, int dtype,
                                           void* stream) {
    GS_CHECK_ARG(z_pix_stride >= 64 && z_pix_stride % 8 == 0, "gs_stem_bwd_onepass_strided: bad z stride");
    return stem_bwd_onepass_impl(x, z, z_pix_stride, dz, dz_stride, dz_coff, act, s1_partials, ws, N, H, W, dtype, stream);
}
static int stem_bwd_onepass_impl(const float* x, const void* z, int z_stride, const void* dz, int dz_stride, int dz_coff, int act,
                                 float* s1_partials, float* ws, int N, int H, int W, int dtype, void* stream) {
    GS_CHECK_ARG(x && z && dz && s1_partials && ws && N > 0 && H > 0 && W > 0 && (int64_t)N * H * W < 2147483647LL / z_stride,
                 "gs_stem_bwd_onepass: bad arguments");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_stem_bwd_onepass: bad dtype");
    GS_CHECK_ARG(dz_stride % 8 == 0 && dz_coff % 8 == 0 && dz_stride >= dz_coff + 64, "gs_stem_bwd_onepass: bad gradient layout");
    GS_CHECK_ARG(act == GS_ACT_NONE || act == GS_ACT_RELU || act == GS_ACT_LEAKY02, "gs_stem_bwd_onepass: bad activation");
    const int64_t M = (int64_t)N * H * W;
    const int64_t ppb = stem_bwd_ppb(M);
    const size_t lds = ((size_t)((ppb + 2 * W + 2 + 3) & ~(int64_t)3) + 256 * 25) * sizeof(float);
    if (lds > 64 * 1024) return GS_EUNSUPPORTED;
    SBWArgs a{};
    a.y = (const unsigned short*)z; a.dz = (const unsigned short*)dz; a.x = x; a.slabs = ws; a.partials = s1_partials;
    a.dz_stride = dz_stride; a.dz_coff = dz_coff; a.N = N; a.H = H; a.W = W; a.act = act; a.pix_per_block = ppb;
    a.zs = z_stride;
    const int nb = (int)cdiv64(M, ppb);
    if (dtype == GS_F16) stem_bwd_onepass_kernel<GS_F16><<<nb, 256, lds, (hipStream_t)stream>>>(a);
    else stem_bwd_onepass_kernel<GS_BF16><<<nb, 256, lds, (hipStream_t)stream>>>(a);
    GS_CHECK_LAUNCH("gs_stem_bwd_onepass");
    return GS_OK;
}

extern "C" int gs_stem_bwd_finalize(const float* ws, const float* s1_partials, const float* tap_sums, const float* w,
                                    const float* scale, const float* mean, const float* invstd, int train_stats, float gscale,
                                    float* dw, float* dgamma, float* dbeta, int N, int H, int W, void* stream) {
    GS_CHECK_ARG(ws && s1_partials && w && scale && mean && invstd && dw && N > 0 && H > 0 && W > 0,
                 "gs_stem_bwd_finalize: bad arguments");
    GS_CHECK_ARG(!train_stats || tap_sums, "gs_stem_bwd_finalize: train-mode statistics need the tap sums of gs_stem_stats");
    stem_bwd_finalize_kernel<<<16, 1024, 0, (hipStream_t)stream>>>(ws, s1_partials, gs_stem_bwd_tiles(N, H, W), tap_sums,
                                                                gs_conv_smallcin_mtiles(N, H, W), w, scale, mean, invstd,
                                                                (double)N * H * W, train_stats, gscale, dw, dgamma, dbeta);
    GS_CHECK_LAUNCH("gs_stem_bwd_finalize");
    return GS_OK;
}

extern "C" int gs_conv_smallcin_dgrad(const void* dy, const float* w, float* dx, int N, int Cin, int IH, int IW,
                                      int Cout, int OH, int OW, int k, int stride, int pad, float gscale, int dtype,
                                      void* stream) {
    int rc = check_smallcin("gs_conv_smallcin_dgrad", N, Cin, IH, IW, Cout, OH, OW, k, stride, pad, dtype);
    if (rc) return rc;
    GS_CHECK_ARG(dy && w && dx, "gs_conv_smallcin_dgrad: null pointer");
    GS_CHECK_ARG((int64_t)N * IH * IW + 4096 * 256 < 2147483647LL, "gs_conv_smallcin_dgrad: too many pixels");
    SCDArgs a{(const unsigned short*)dy, w, dx, N, Cin, IH, IW, Cout, OH, OW, k, stride, pad, gscale};
    int64_t nb = cdiv64((int64_t)N * IH * IW, 256);
    if (nb > 4096) nb = 4096;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GS_F16) smallcin_dgrad_kernel<GS_F16><<<(int)nb, 256, 0, s>>>(a);
    else smallcin_dgrad_kernel<GS_BF16><<<(int)nb, 256, 0, s>>>(a);
    GS_CHECK_LAUNCH("gs_conv_smallcin_dgrad");
    return GS_OK;
}

static int check_head(const char* who, int N, int IH, int IW, int Cin, int Cout, int OH, int OW, int k, int stride,
                      int pad, int dtype) {
    GS_CHECK_ARG(N > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0, "%s: bad dims", who);
    GS_CHECK_ARG(Cout >= 1 && Cout <= 4, "%s: Cout=%d must be 1..4", who, Cout);
    const int nch = Cin / 8;
    GS_CHECK_ARG(Cin % 8 == 0 && nch >= 1 && (nch & (nch - 1)) == 0 && nch <= 256, "%s: Cin=%d must be 8*2^j <= 2048", who, Cin);
    GS_CHECK_ARG(k >= 1 && stride >= 1 && pad >= 0 && Cout * k * k * Cin <= SC_MAX_W, "%s: weights exceed %d floats", who, SC_MAX_W);
    GS_CHECK_ARG(OH == (IH + 2 * pad - k) / stride + 1 && OW == (IW + 2 * pad - k) / stride + 1, "%s: output size mismatch", who);
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "%s: bad dtype", who);
    return GS_OK;
}

static int smallcout_fwd_impl(const void* x, const float* w, const float* bias, float* y, int N, int IH, int IW,
                              int Cin, int Cout, int OH, int OW, int k, int stride, int pad, int dtype,
                              void* stream, const float* bn_scale, const float* bn_shift, float bn_slope) {
    int rc = check_head("gs_conv_smallcout_fwd", N, IH, IW, Cin, Cout, OH, OW, k, stride, pad, dtype);
    if (rc) return rc;
    GS_CHECK_ARG(x && w && y, "gs_conv_smallcout_fwd: null pointer");
    GS_CHECK_ARG((int64_t)N * OH * OW + 256 < 2147483647LL, "gs_conv_smallcout_fwd: too many pixels");
    HArgs a{};
    a.bn_scale = bn_scale; a.bn_shift = bn_shift; a.bn_slope = bn_slope;
    a.x = (const unsigned short*)x; a.w = w; a.bias = bias; a.y = y;
    a.N = N; a.IH = IH; a.IW = IW; a.Cin = Cin; a.Cout = Cout; a.OH = OH; a.OW = OW; a.k = k; a.stride = stride; a.pad = pad;
    const int nch = Cin / 8, gl = nch < 64 ? nch : 64, groups = 256 / gl;
    int64_t nb = cdiv64((int64_t)N * OH * OW, groups);
    if (nb > 8192) nb = 8192;
    hipStream_t s = (hipStream_t)stream;
    if (k == 1 && stride == 1 && pad == 0 && Cin == 64) {
        int64_t hb = cdiv64((int64_t)N * OH * OW, 128);
        if (hb > 4096) hb = 4096;
        if (dtype == GS_F16) head1x1_fwd_kernel<GS_F16><<<(int)hb, 256, 0, s>>>(a);
        else head1x1_fwd_kernel<GS_BF16><<<(int)hb, 256, 0, s>>>(a);
    } else if (dtype == GS_F16) smallcout_fwd_kernel<GS_F16><<<(int)nb, 256, 0, s>>>(a);
    else smallcout_fwd_kernel<GS_BF16><<<(int)nb, 256, 0, s>>>(a);
    GS_CHECK_LAUNCH("gs_conv_smallcout_fwd");
    return GS_OK;
}

extern "C" int gs_conv_smallcout_fwd(const void* x, const float* w, const float* bias, float* y, int N, int IH, int IW,
                                     int Cin, int Cout, int OH, int OW, int k, int stride, int pad, int dtype,
                                     void* stream) {
    return smallcout_fwd_impl(x, w, bias, y, N, IH, IW, Cin, Cout, OH, OW, k, stride, pad, dtype, stream, nullptr, nullptr, 0.f);
}

static float head_bn_slope(int act) { return act == GS_ACT_RELU ? 0.f : (act == GS_ACT_LEAKY02 ? 0.2f : 1.f); }

extern "C" int gs_head1x1_bn_fwd(const void* y_conv, const float* bn_scale, const float* bn_shift, int act, const float* w,
                                 const float* bias, float* logits, int N, int H, int W, int Cout, int dtype, void* stream) {
    GS_CHECK_ARG(bn_scale && bn_shift, "gs_head1x1_bn_fwd: needs the BatchNorm scale / shift");
    GS_CHECK_ARG(act == GS_ACT_NONE || act == GS_ACT_RELU || act == GS_ACT_LEAKY02, "gs_head1x1_bn_fwd: activation %d not supported", act);
    return smallcout_fwd_impl(y_conv, w, bias, logits, N, H, W, 64, Cout, H, W, 1, 1, 0, dtype, stream, bn_scale, bn_shift,
                              head_bn_slope(act));
}

static int smallcout_bwd_impl(const void* x, const float* w, const float* dy, void* dx, float* dw, float* db,
                              float* ws, int N, int IH, int IW, int Cin, int Cout, int OH, int OW, int k,
                              int stride, int pad, float gscale, int dtype, void* stream, const float* bn_scale,
                              const float* bn_shift, float bn_slope) {
    int rc = check_head("gs_conv_smallcout_bwd", N, IH, IW, Cin, Cout, OH, OW, k, stride, pad, dtype);
    if (rc) return rc;
    GS_CHECK_ARG(w && dy, "gs_conv_smallcout_bwd: null pointer");
    GS_CHECK_ARG((int64_t)N * IH * IW * (Cin / 8) < 2147483647LL && (int64_t)N * OH * OW < 2147483647LL,
                 "gs_conv_smallcout_bwd: too many pixels");
    HArgs a{};
    a.bn_scale = bn_scale; a.bn_shift = bn_shift; a.bn_slope = bn_slope;
    a.x = (const unsigned short*)x; a.w = w; a.dy = dy; a.dx = (unsigned short*)dx; a.dw = dw; a.db = db;
    a.N = N; a.IH = IH; a.IW = IW; a.Cin = Cin; a.Cout = Cout; a.OH = OH; a.OW = OW; a.k = k; a.stride = stride; a.pad = pad;
    a.gscale = gscale;
    hipStream_t s = (hipStream_t)stream;
    if (dx) {
        int64_t nb = cdiv64((int64_t)N * IH * IW * (Cin / 8), 256);
        if (nb > 8192) nb = 8192;
        if (k == 1 && stride == 1 && pad == 0 && Cin == 64) {
            int64_t hb = cdiv64((int64_t)N * IH * IW, 128);
            if (hb > 4096) hb = 4096;
            if (dtype == GS_F16) head1x1_dgrad_kernel<GS_F16><<<(int)hb, 256, 0, s>>>(a);
            else head1x1_dgrad_kernel<GS_BF16><<<(int)hb, 256, 0, s>>>(a);
        } else if (dtype == GS_F16) smallcout_dgrad_kernel<GS_F16><<<(int)nb, 256, 0, s>>>(a);
        else smallcout_dgrad_kernel<GS_BF16><<<(int)nb, 256, 0, s>>>(a);
    }
    if (dw) {
        GS_CHECK_ARG(x != nullptr && ws != nullptr, "gs_conv_smallcout_bwd: dw needs x and a workspace");
        const int64_t M = (int64_t)N * OH * OW;
        const int64_t ppb = direct_wgrad_ppb(M, (k == 1 && Cin == 64) ? 64 : 8);
        a.pix_per_block = ppb;
        const int nb = (int)cdiv64(M, ppb);
        const int n = Cout * Cin * k * k;
        HArgs b = a;
        b.dw = ws;                                   // partial slabs [nb][n], then bias partials [nb][4]
        b.db = db ? ws + (int64_t)nb * n : nullptr;
        if (k == 1 && stride == 1 && pad == 0 && Cin == 64) {
            if (dtype == GS_F16) head1x1_wgrad_kernel<GS_F16><<<nb, 256, 0, s>>>(b);
            else head1x1_wgrad_kernel<GS_BF16><<<nb, 256, 0, s>>>(b);
        } else if (dtype == GS_F16) smallcout_wgrad_kernel<GS_F16><<<nb, 256, 0, s>>>(b);
        else smallcout_wgrad_kernel<GS_BF16><<<nb, 256, 0, s>>>(b);
        slab_reduce_kernel<<<cdiv(n, 32), 256, 0, s>>>(ws, nb, n, n, gscale, dw);
        if (db) slab_reduce_kernel<<<1, 256, 0, s>>>(ws + (int64_t)nb * n, nb, 4, Cout, gscale, db);
    }
    GS_CHECK_LAUNCH("gs_conv_smallcout_bwd");
    return GS_OK;
}

extern "C" int gs_conv_smallcout_bwd(const void* x, const float* w, const float* dy, void* dx, float* dw, float* db,
                                     float* ws, int N, int IH, int IW, int Cin, int Cout, int OH, int OW, int k,
                                     int stride, int pad, float gscale, int dtype, void* stream) {
    return smallcout_bwd_impl(x, w, dy, dx, dw, db, ws, N, IH, IW, Cin, Cout, OH, OW, k, stride, pad, gscale, dtype, stream,
                              nullptr, nullptr, 0.f);
}

extern "C" int gs_head1x1_bn_wgrad(const void* y_conv, const float* bn_scale, const float* bn_shift, int act, const float* w,
                                   const float* dl, float* dw, float* db, float* ws, int N, int H, int W, int Cout,
                                   float gscale, int dtype, void* stream) {
    GS_CHECK_ARG(bn_scale && bn_shift && dw, "gs_head1x1_bn_wgrad: needs the BatchNorm scale / shift and dw");
    GS_CHECK_ARG(act == GS_ACT_NONE || act == GS_ACT_RELU || act == GS_ACT_LEAKY02, "gs_head1x1_bn_wgrad: activation %d not supported", act);
    return smallcout_bwd_impl(y_conv, w, dl, nullptr, dw, db, ws, N, H, W, 64, Cout, H, W, 1, 1, 0, gscale, dtype, stream,
                              bn_scale, bn_shift, head_bn_slope(act));
}
