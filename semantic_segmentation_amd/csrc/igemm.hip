// MFMA implicit-GEMM convolution engine for gfx950 (MI355X): forward / data-gradient
// (gs_conv_igemm) and weight-gradient (gs_conv_wgrad) of any Conv2d / ConvTranspose2d that the
// GsConvGeom tap-list form can express.
//
// Forward:  Y[m][co] = sum_{tap,ci} X[inpix(m,tap)][ci] * W[tap][co][ci]
//   GEMM view M = N*OHg*OWg pixels, N = Cout, K = ntaps*Cin.  Block tile 128(M) x BN(64|128) x 64(K),
//   4 waves (2x2), each wave 64 x BN/2 as 32x32x16 MFMA tiles, fp32 accumulate.  A (gathered NHWC rows,
//   zero outside the image) and B (packed weights) are register-staged into padded LDS rows
//   (144 B stride: conflict-free ds_read_b128 for the 32x32x16 operand), double buffered, one barrier per
//   K-step.  Epilogue: bias / activation, 16-bit store, and per-M-tile BatchNorm partial sums.
// Weight gradient: dW[tap][co][ci] += sum_m dY[m][co] * X[inpix(m,tap)][ci]
//   GEMM view M = Cout, N = (tap, ci), K = pixels.  Both operands are pixel-major in memory but the
//   MFMA wants K(=pixel)-minor fragments: tiles are staged as [pixel][channel] and read with the
//   gfx950 transposing LDS read (ds_read_b64_tr_b16).  Split-K over pixels, fp32 atomics into dW.
#include <stdlib.h>

#include "common.hpp"
#include "skinny.hpp"

namespace {

// Compact geometry for batched launches (several GEMMs -- the four sub-pixel classes of a stride-2 transposed conv or of a
// stride-2 conv's data gradient -- in ONE grid): same field names as GsConvGeom with 16-tap arrays, so that four of them fit
// in the 4 KB kernel-argument segment.
constexpr int GEOMC_MAX_TAPS = 16;
struct GeomC {
    int32_t N, IH, IW, Cin, in_pix_stride, in_coff;
    int32_t OHg, OWg, Cout;
    int32_t OH, OW, out_pix_stride, out_coff;
    int32_t isy, isx, osy, osx, ooy, oox;
    int32_t ntaps;
    int32_t tap_dy[GEOMC_MAX_TAPS], tap_dx[GEOMC_MAX_TAPS], tap_w[GEOMC_MAX_TAPS];
    int32_t Dg, Din, Dout, isz, osz, ooz;
    int32_t tap_dz[GEOMC_MAX_TAPS];
};

template <typename GEOM>
struct IgemmArgsT {
    GEOM g;
    const unsigned short* x;
    const unsigned short* w;
    unsigned short* y;
    const float* bias;
    float* bnp;
    int act;
    int M;        // N*OHg*OWg
    int kchunks;  // ceil(Cin/64)
    int ntn;      // number of N tiles
    int nblocks;
    int vec_store;     // 1: epilogue stages the tile through LDS and stores 16-byte channel chunks
    int shuffle_cout;  // > 0: merged stride-2 transposed conv (kernel 2): GEMM column c' = cls * shuffle_cout + co is
    int shuffle_cls;   //      written to output pixel (2z+cz, 2y+cy, 2x+cx), cls = (cz, cy, cx) bits; 4 or 8 classes
    // split-K (skinny GEMMs: few M/N tiles, long K): blockIdx = tile * ksplit + part; every part stores its fp32
    // accumulators in its slab ws_acc[tile][part][256 threads][regs]; the LAST part to arrive (ticket in ws_cnt[tile])
    // sums the slabs, runs the normal epilogue and resets the ticket counter.
    int ksplit;
    float* ws_acc;
    unsigned int* ws_cnt;
    // split-K runs as TWO launches on the stream (no in-kernel hand-off): phase 1 = K parts store their fp32 tiles in slabs,
    // phase 2 = one block per tile sums the slabs in part order and runs the epilogue.  (The former one-launch form -- every
    // block a device-scope release fence, a ticket, the last arriver reducing -- made a launch of 512-1024 blocks cost
    // 180-350 us: the fences write back the whole L2 and serialise.)  phase 0: no split.
    int phase;
    // precise mode (igemm_fwd_kernel<.., PREC>): K = g.Cin is a concatenation of segments over `in_wrap` input channels
    // (K channel ci reads input channel ci >= in_wrap ? ci - in_wrap : ci); the result is stored as a 16-bit hi/lo pair
    // (hi at y, lo = 16-bit(value - hi) at y_lo, same stride / offset).
    unsigned short* y_lo = nullptr;
    int in_wrap = 0;
};
typedef IgemmArgsT<GsConvGeom> IgemmArgs;
typedef IgemmArgsT<GeomC> IgemmArgsC;
constexpr int IGEMM_BATCH_MAX = 4;
struct IgemmBatchArgs {
    IgemmArgsC c[IGEMM_BATCH_MAX];
    int start[IGEMM_BATCH_MAX + 1];     // first block of every GEMM; start[n] = grid size
    int n;
};
static_assert(sizeof(IgemmBatchArgs) <= 3800, "batched igemm arguments must fit the kernel-argument segment");

// bijective XCD-aware remap: blocks that share an XCD (bid % 8) get a contiguous range of logical ids
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

constexpr int FW_BM = 128, FW_BK = 64, FW_LDR = 72;   // LDS row = 64 + 8 pad elements (144 B)

template <int DT, int BN, bool PACKED, bool PREC, typename ARGS>
__device__ __forceinline__ void igemm_fwd_body(const ARGS& a, const int block_id) {
    static_assert(!(PACKED && PREC), "no precise form of the packed-tap kernel");
    typedef typename Elem<DT>::V8 V8;
    constexpr int NT = BN / 64;            // 32-wide N tiles per wave
    constexpr int BROWS = BN / 32;         // B rows staged per thread
    constexpr int A_EL = FW_BM * FW_LDR, B_EL = BN * FW_LDR;
    constexpr int BUF_EL = A_EL + B_EL;
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * BUF_EL];

    const auto& g = a.g;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int ksplit = a.ksplit;
    const bool reduce_phase = ksplit > 1 && a.phase == 2;       // one block per tile: slabs -> epilogue
    const int kpart = (ksplit > 1 && !reduce_phase) ? (int)(block_id % ksplit) : 0;
    const int lid = (ksplit > 1 && !reduce_phase) ? (int)(block_id / ksplit) : xcd_remap(block_id, a.nblocks);
    const int ntile = lid % a.ntn, mtile = lid / a.ntn;
    const int m0 = mtile * FW_BM, n0 = ntile * BN;

    // ---- per-thread staging rows ----
    const int chunk = t & 7, rbase = t >> 3;
    int a_iy0[4], a_ix0[4], a_n[4], a_d0[4];      // a_n: input image index before the depth tap; a_d0: depth part
    const int ohw = g.OHg * g.OWg;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + rbase + 32 * j;
        if (m < a.M) {
            const int n = m / ohw, rem = m - n * ohw;
            const int oy = rem / g.OWg, ox = rem - oy * g.OWg;
            const int nb = n / g.Dg, d = n - nb * g.Dg;
            a_d0[j] = d * g.isz; a_n[j] = nb * g.Din + d * g.isz; a_iy0[j] = oy * g.isy; a_ix0[j] = ox * g.isx;
        } else {
            a_n[j] = -1; a_d0[j] = 0; a_iy0[j] = 0; a_ix0[j] = 0;
        }
    }
    uint4 ra[4], rb[BROWS];
    // Taps none of this tile's 128 rows can see (out of the image for every row) are dropped from the K loop: at the
    // 1x1 .. 8x8 bottom of the Pix2Pix generator 75-94 % of the 4x4 / 8x8 taps fall outside the feature map, and with
    // M = a few pixels the kernel otherwise streams (and multiplies by zero) the whole multi-megabyte weight pack.
    __shared__ int act_taps[GS_MAX_TAPS + 1];
    __shared__ int4 tap_info[GS_MAX_TAPS];          // (dy, dx, dz, weight slot) of the active taps, in list order
    {
        unsigned long long mask = 0ull;              // ntaps <= 64
        for (int tp = 0; tp < g.ntaps; ++tp) {
            const int dy = g.tap_dy[tp], dx = g.tap_dx[tp], dz = g.tap_dz[tp];
            bool any = false;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                any = any || (a_n[j] >= 0 && (unsigned)(a_iy0[j] + dy) < (unsigned)g.IH &&
                              (unsigned)(a_ix0[j] + dx) < (unsigned)g.IW && (unsigned)(a_d0[j] + dz) < (unsigned)g.Din);
            if (__any(any)) mask |= 1ull << tp;
        }
        // OR over the four waves, then a compact list in tap order
        unsigned long long* wmask = reinterpret_cast<unsigned long long*>(smem);
        if (lane == 0) wmask[wave] = mask;
        __syncthreads();
        if (t == 0) {
            const unsigned long long m4 = wmask[0] | wmask[1] | wmask[2] | wmask[3];
            int n = 0;
            for (int tp = 0; tp < g.ntaps; ++tp)
                if ((m4 >> tp) & 1ull) {
                    // the tap descriptors move to LDS with the list: indexing the kernel-argument arrays with a run-time
                    // tap number inside the K loop is a scalar memory load per K-step per wave
                    tap_info[n] = make_int4(g.tap_dy[tp], g.tap_dx[tp], g.tap_dz[tp], g.tap_w[tp]);
                    act_taps[1 + n++] = tp;
                }
            act_taps[0] = n;
        }
        __syncthreads();
    }
    // PACKED (Cin == 8: the 1-channel image end of the Pix2Pix generator, padded to one 16-byte chunk): a K step carries
    // EIGHT taps -- chunk c of the 64-wide K slice = (tap 8*step + c, channels 0..7) -- instead of one tap with 8 of its 64
    // channels in use.  A separate instantiation: the per-chunk tap makes the tap offsets vector values.
    const int nk_all = PACKED ? (act_taps[0] + 7) / 8 : act_taps[0] * a.kchunks;
    const int ks_begin = ksplit > 1 ? (int)((int64_t)nk_all * kpart / ksplit) : 0;
    const int ks_end = ksplit > 1 ? (int)((int64_t)nk_all * (kpart + 1) / ksplit) : nk_all;
    const int nk = reduce_phase ? 0 : ks_end - ks_begin;

    auto load_tile = [&](int ks) {
        int ti = (ks + ks_begin) / a.kchunks;
        const int cc = (ks + ks_begin) - ti * a.kchunks;
        int ci = cc * FW_BK + chunk * 8;
        bool cok = ci < g.Cin;
        if constexpr (PACKED) {
            ti = (ks + ks_begin) * 8 + chunk;
            cok = ti < act_taps[0];
            if (!cok) ti = 0;
            ci = 0;
        }
        const int4 ti4 = tap_info[ti];
        const int dy = ti4.x, dx = ti4.y, dz = ti4.z, tapw = ti4.w;
        const int cix = (PREC && a.in_wrap > 0 && ci >= a.in_wrap) ? ci - a.in_wrap : ci;     // input channel of K channel ci
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iy = a_iy0[j] + dy, ix = a_ix0[j] + dx;
            const bool ok = cok && a_n[j] >= 0 && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW &&
                            (unsigned)(a_d0[j] + dz) < (unsigned)g.Din;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) {
                const int64_t pix = ((int64_t)(a_n[j] + dz) * g.IH + iy) * g.IW + ix;
                v = *reinterpret_cast<const uint4*>(a.x + pix * g.in_pix_stride + g.in_coff + cix);
            }
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < BROWS; ++j) {
            const int co = n0 + rbase + 32 * j;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (cok && co < g.Cout)
                v = *reinterpret_cast<const uint4*>(a.w + ((int64_t)tapw * g.Cout + co) * g.Cin + ci);
            rb[j] = v;
        }
    };
    auto store_tile = [&](int buf) {
        unsigned short* A = smem + buf * BUF_EL;
        unsigned short* B = A + A_EL;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<uint4*>(A + (rbase + 32 * j) * FW_LDR + chunk * 8) = ra[j];
#pragma unroll
        for (int j = 0; j < BROWS; ++j)
            *reinterpret_cast<uint4*>(B + (rbase + 32 * j) * FW_LDR + chunk * 8) = rb[j];
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nk > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < nk) load_tile(ks + 1);
        const unsigned short* A = smem + cur * BUF_EL + (wm * 64 + l31) * FW_LDR + h * 8;
        const unsigned short* B = smem + cur * BUF_EL + A_EL + (wn * (BN / 2) + l31) * FW_LDR + h * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            V8 af[2], bf[NT];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const V8*>(A + i * 32 * FW_LDR + kk * 16);
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const V8*>(B + j * 32 * FW_LDR + kk * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = Elem<DT>::mfma32(af[i], bf[j], acc[i][j]);
        }
        if (ks + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    if (ksplit > 1) {
        // phase 1: every part stores its partial tile in its own slab (plain 16-byte stores) and is done; phase 2 (the next
        // launch on the stream) sums the slabs in part order -- deterministic, no atomics, no fences
        constexpr int REGS = 2 * NT * 16;
        // a 32-row block of the tile that lies wholly beyond M (skinny GEMMs fill 8..32 of the 128 rows) is neither
        // stored nor summed: its accumulators are never written to y
        bool rows_ok[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) rows_ok[i] = m0 + wm * 64 + i * 32 < a.M;
        if (!reduce_phase) {
            float* slab = a.ws_acc + (((int64_t)lid * ksplit + kpart) * 256 + t) * REGS;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (!rows_ok[i]) continue;
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4)
                        *reinterpret_cast<float4*>(slab + (i * NT + j) * 16 + r4 * 4) =
                            make_float4(acc[i][j][r4 * 4], acc[i][j][r4 * 4 + 1], acc[i][j][r4 * 4 + 2], acc[i][j][r4 * 4 + 3]);
            }
            return;
        }
        for (int pp = 0; pp < ksplit; ++pp) {                     // acc is still zero: no K step ran in this phase
            const float* q = a.ws_acc + (((int64_t)lid * ksplit + pp) * 256 + t) * REGS;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (!rows_ok[i]) continue;
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const float4 v = *reinterpret_cast<const float4*>(q + (i * NT + j) * 16 + r4 * 4);
                        acc[i][j][r4 * 4] += v.x; acc[i][j][r4 * 4 + 1] += v.y;
                        acc[i][j][r4 * 4 + 2] += v.z; acc[i][j][r4 * 4 + 3] += v.w;
                    }
            }
        }
    }

    // ---- epilogue: row table (output pixel index per tile row), stores, BN partial sums ----
    int* rowpix = reinterpret_cast<int*>(smem);                       // 128 ints
    float* red = reinterpret_cast<float*>(smem) + 128;                // [2 wm][2 stat][BN] floats
    if (t < FW_BM) {
        const int m = m0 + t;
        int p = -1;
        if (m < a.M) {
            const int n = m / ohw, rem = m - n * ohw;
            const int oy = rem / g.OWg, ox = rem - oy * g.OWg;
            const int nb = n / g.Dg, d = n - nb * g.Dg;
            p = ((nb * g.Dout + d * g.osz + g.ooz) * g.OH + oy * g.osy + g.ooy) * g.OW + ox * g.osx + g.oox;
        }
        rowpix[t] = p;
    }
    __syncthreads();
    const float slope = a.act == GS_ACT_RELU ? 0.f : (a.act == GS_ACT_LEAKY02 ? 0.2f : 1.f);
    const bool is_tanh = a.act == GS_ACT_TANH;
    const int bmod = a.shuffle_cout > 0 ? a.shuffle_cout : g.Cout;    // bias / output channel = column mod bmod
    float bvj[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int co = n0 + wn * (BN / 2) + j * 32 + l31;
        bvj[j] = (a.bias != nullptr && co < g.Cout) ? a.bias[co % bmod] : 0.f;
    }
    if (a.bnp != nullptr) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[i][j][r];
                    s1 += v;
                    s2 += v * v;
                }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (h == 0) {
                const int c = wn * (BN / 2) + j * 32 + l31;
                red[(wm * 2 + 0) * BN + c] = s1;
                red[(wm * 2 + 1) * BN + c] = s2;
            }
        }
    }
    // bias + activation in place (slope family branch-free; tanh behind one uniform branch)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[i][j][r] + bvj[j];
                acc[i][j][r] = v > 0.f ? v : v * slope;
            }
    if (is_tanh) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = tanhf(acc[i][j][r]);
    }
    if (a.vec_store) {
        // stage 32 rows x BN/2 columns per wave in LDS, then every lane stores 16-byte channel chunks
        constexpr int WCOLS = BN / 2, STG_LD = WCOLS + 8, CH = WCOLS / 8;
        unsigned short* stg = smem + 2048 + wave * (32 * STG_LD);     // behind rowpix / red
        const int ohw_out = g.OH * g.OW;
#pragma unroll
        for (int pass = 0; pass < (PREC ? 2 : 1); ++pass) {           // PREC pass 1: the lo halves, acc -= hi first
        unsigned short* ydst = (PREC && pass == 1) ? a.y_lo : a.y;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const unsigned short q16 = Elem<DT>::from_f(acc[i][j][r]);
                    stg[row * STG_LD + j * 32 + l31] = q16;
                    if (PREC && pass == 0) acc[i][j][r] -= Elem<DT>::to_f(q16);
                }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < (32 * CH) / 64; ++q) {
                const int idx = q * 64 + lane;
                const int rrow = idx / CH, ch = idx - rrow * CH;
                const uint4 v = *reinterpret_cast<const uint4*>(stg + rrow * STG_LD + ch * 8);
                int p = rowpix[wm * 64 + i * 32 + rrow];
                int co = n0 + wn * WCOLS + ch * 8;
                if (p >= 0 && co < g.Cout) {
                    if (a.shuffle_cout > 0) {
                        const int cls = co / a.shuffle_cout;
                        co -= cls * a.shuffle_cout;
                        p += (cls & 1) + ((cls >> 1) & 1) * g.OW + (a.shuffle_cls == 8 ? (cls >> 2) * ohw_out : 0);
                    }
                    *reinterpret_cast<uint4*>(ydst + (int64_t)p * g.out_pix_stride + g.out_coff + co) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int co = n0 + wn * (BN / 2) + j * 32 + l31;
            const bool cok = co < g.Cout;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int p = rowpix[row];
                    if (cok && p >= 0)
                        a.y[(int64_t)p * g.out_pix_stride + g.out_coff + co] = Elem<DT>::from_f(acc[i][j][r]);
                }
        }
    }
    if (a.bnp != nullptr) {
        __syncthreads();
        if (t < BN && n0 + t < g.Cout) {
            float* dst = a.bnp + (int64_t)mtile * 2 * g.Cout + n0 + t;
            dst[0] = red[(0 * 2 + 0) * BN + t] + red[(1 * 2 + 0) * BN + t];
            dst[g.Cout] = red[(0 * 2 + 1) * BN + t] + red[(1 * 2 + 1) * BN + t];
        }
    }
}

template <int DT, int BN, bool PACKED, bool PREC = false>
__global__ __launch_bounds__(256) void igemm_fwd_kernel(const IgemmArgs a) {
    igemm_fwd_body<DT, BN, PACKED, PREC>(a, (int)blockIdx.x);
}

// Several independent GEMMs (<= 4, <= 16 taps each) in one grid: block b belongs to GEMM c with start[c] <= b < start[c+1].
// The sub-pixel classes of a layer are latency bound at the script's batch size (M = a few hundred pixels, one launch
// fills a fraction of the chip and costs a launch gap): together they fill it and pay one gap.
template <int DT, int BN>
__global__ __launch_bounds__(256) void igemm_fwd_batch_kernel(const IgemmBatchArgs b) {
    int c = 0;
#pragma unroll
    for (int i = 1; i < IGEMM_BATCH_MAX; ++i)
        if (i < b.n && (int)blockIdx.x >= b.start[i]) c = i;
    igemm_fwd_body<DT, BN, false, false>(b.c[c], (int)blockIdx.x - b.start[c]);
}

// ------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------
template <typename GEOM>
struct WgradArgsT {
    GEOM g;
    const unsigned short* x;
    const unsigned short* dy;
    float* dw;
    int M;         // logical pixels N*OHg*OWg
    int kchunks;   // ceil(Cin/64)
    int ncb;       // ntaps*kchunks column blocks
    int n_cotiles, n_cbgroups, ksplit;
    int kper;      // pixels per K split (multiple of 64)
    int d_n, d_oy, d_ox;   // mixed-radix decomposition of 64 pixels
    int assign;            // plain stores instead of atomic adds: ksplit == 1 (dw need not be zeroed) or slab mode
    int64_t slab_stride;   // > 0: K part ks writes its own slab dw + ks*slab_stride (ordered reduction afterwards)
};
typedef WgradArgsT<GsConvGeom> WgradArgs;
typedef WgradArgsT<GeomC> WgradArgsC;
struct WgradBatchArgs {
    WgradArgsC c[IGEMM_BATCH_MAX];
    int start[IGEMM_BATCH_MAX + 1];
    int n;
};
static_assert(sizeof(WgradBatchArgs) <= 3800, "batched wgrad arguments must fit the kernel-argument segment");

constexpr int WG_KP = 64;   // pixels per K step

// WR = wave rows: 2 -> block tile 128 co x 2 column blocks; 1 -> 64 co x 4 column blocks
template <int DT, int WR, typename ARGS>
__device__ __forceinline__ void igemm_wgrad_body(const ARGS& a, const int block_id) {
    typedef typename Elem<DT>::V8 V8;
    constexpr int WC = 4 / WR;                 // column blocks per block (= wave columns)
    constexpr int BCO = 64 * WR;
    constexpr int DY_LDR = BCO + 32;           // 320 B / 192 B rows: conflict-free transposed reads
    constexpr int X_LDR = 64 + 32;
    constexpr int DY_EL = WG_KP * DY_LDR, X_EL = WG_KP * X_LDR;
    constexpr int DY_CH = BCO / 8;             // 16-byte chunks per dY row
    constexpr int DY_PER_T = WG_KP * DY_CH / 256;
    constexpr int DY_ROWS_STEP = 256 / DY_CH;
    __shared__ __attribute__((aligned(16))) unsigned short smem[DY_EL + WC * X_EL];

    const auto& g = a.g;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (WR == 2) ? (wave >> 1) : 0;
    const int wn = (WR == 2) ? (wave & 1) : wave;
    int bid = block_id;
    const int cot = bid % a.n_cotiles; bid /= a.n_cotiles;
    const int cbg = bid % a.n_cbgroups; bid /= a.n_cbgroups;
    const int ks_id = bid;
    const int co0 = cot * BCO;
    const int kbeg = ks_id * a.kper;
    const int kend = min(a.M, kbeg + a.kper);
    const int ohw = g.OHg * g.OWg;

    // column blocks of this block
    int cb_tap[WC], cb_ci0[WC];
#pragma unroll
    for (int c = 0; c < WC; ++c) {
        const int cb = cbg * WC + c;
        if (cb < a.ncb) { cb_tap[c] = cb / a.kchunks; cb_ci0[c] = (cb - cb_tap[c] * a.kchunks) * 64; }
        else { cb_tap[c] = -1; cb_ci0[c] = 0; }
    }

    // ---- staging rows of this thread: dY rows and X rows, tracked as (n, oy, ox) and advanced by 64 ----
    const int dy_chunk = t % DY_CH, dy_row0 = t / DY_CH;
    const int x_chunk = t & 7, x_row0 = t >> 3;
    int dn[DY_PER_T], doy[DY_PER_T], dox[DY_PER_T], dm[DY_PER_T];
    int xn[2], xoy[2], xox[2], xm[2];
    auto decode = [&](int m, int& n, int& oy, int& ox) {
        n = m / ohw; const int rem = m - n * ohw; oy = rem / g.OWg; ox = rem - oy * g.OWg;
    };
#pragma unroll
    for (int j = 0; j < DY_PER_T; ++j) { dm[j] = kbeg + dy_row0 + DY_ROWS_STEP * j; decode(dm[j], dn[j], doy[j], dox[j]); }
#pragma unroll
    for (int j = 0; j < 2; ++j) { xm[j] = kbeg + x_row0 + 32 * j; decode(xm[j], xn[j], xoy[j], xox[j]); }
    auto advance = [&](int& m, int& n, int& oy, int& ox) {
        m += WG_KP;
        ox += a.d_ox; if (ox >= g.OWg) { ox -= g.OWg; oy += 1; }
        oy += a.d_oy; if (oy >= g.OHg) { oy -= g.OHg; n += 1; }
        n += a.d_n;
    };

    uint4 rdy[DY_PER_T], rx[WC][2];
    auto load_tile = [&]() {
#pragma unroll
        for (int j = 0; j < DY_PER_T; ++j) {
            uint4 v = make_uint4(0, 0, 0, 0);
            const int co = co0 + dy_chunk * 8;
            if (dm[j] < kend && co < g.Cout) {
                const int nb = dn[j] / g.Dg, d = dn[j] - nb * g.Dg;
                const int64_t pix = ((int64_t)(nb * g.Dout + d * g.osz + g.ooz) * g.OH + doy[j] * g.osy + g.ooy) * g.OW +
                                    dox[j] * g.osx + g.oox;
                v = *reinterpret_cast<const uint4*>(a.dy + pix * g.out_pix_stride + g.out_coff + co);
            }
            rdy[j] = v;
        }
#pragma unroll
        for (int c = 0; c < WC; ++c) {
            const int tap = cb_tap[c];
            const int ci = cb_ci0[c] + x_chunk * 8;
            const int tdy = tap >= 0 ? g.tap_dy[tap] : 0, tdx = tap >= 0 ? g.tap_dx[tap] : 0;
            const int tdz = tap >= 0 ? g.tap_dz[tap] : 0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint4 v = make_uint4(0, 0, 0, 0);
                const int iy = xoy[j] * g.isy + tdy, ix = xox[j] * g.isx + tdx;
                const int nb = xn[j] / g.Dg, d = xn[j] - nb * g.Dg;
                const int iz = d * g.isz + tdz;
                if (tap >= 0 && xm[j] < kend && ci < g.Cin && (unsigned)iy < (unsigned)g.IH &&
                    (unsigned)ix < (unsigned)g.IW && (unsigned)iz < (unsigned)g.Din) {
                    const int64_t pix = ((int64_t)(nb * g.Din + iz) * g.IH + iy) * g.IW + ix;
                    v = *reinterpret_cast<const uint4*>(a.x + pix * g.in_pix_stride + g.in_coff + ci);
                }
                rx[c][j] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < DY_PER_T; ++j) advance(dm[j], dn[j], doy[j], dox[j]);
#pragma unroll
        for (int j = 0; j < 2; ++j) advance(xm[j], xn[j], xoy[j], xox[j]);
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < DY_PER_T; ++j)
            *reinterpret_cast<uint4*>(smem + (dy_row0 + DY_ROWS_STEP * j) * DY_LDR + dy_chunk * 8) = rdy[j];
#pragma unroll
        for (int c = 0; c < WC; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<uint4*>(smem + DY_EL + c * X_EL + (x_row0 + 32 * j) * X_LDR + x_chunk * 8) = rx[c][j];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposed-read lane addressing (ds_read_b64_tr_b16): 16-lane group G = lane>>4 covers channels
    // 16*(G&1)..+15 and k (pixel) rows 8*(G>>1) + 4*r + q;  lane 4q+p supplies row q, channels 4p..4p+3.
    const int G = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int krow = 8 * (G >> 1) + q;
    const int chn = 16 * (G & 1) + 4 * p;
    const LDS_AS unsigned short* lds = (const LDS_AS unsigned short*)smem;
    const LDS_AS unsigned short* a_base = lds + krow * DY_LDR + wm * 64 + chn;
    const LDS_AS unsigned short* b_base = lds + DY_EL + wn * X_EL + krow * X_LDR + chn;

    const int nsteps = (kend - kbeg + WG_KP - 1) / WG_KP;
    if (nsteps > 0) load_tile();
    for (int s = 0; s < nsteps; ++s) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (s + 1 < nsteps) load_tile();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            V8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[i] = tr_read8<DT>(a_base + (kk * 16) * DY_LDR + i * 32, a_base + (kk * 16 + 4) * DY_LDR + i * 32);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bf[j] = tr_read8<DT>(b_base + (kk * 16) * X_LDR + j * 32, b_base + (kk * 16 + 4) * X_LDR + j * 32);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = Elem<DT>::mfma32(af[i], bf[j], acc[i][j]);
        }
    }

    // ---- epilogue: fp32 atomics into dW[tap][co][ci] ----
    const int tap = cb_tap[wn];
    if (tap < 0 || nsteps <= 0) return;
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ci = cb_ci0[wn] + j * 32 + l31;
            if (ci >= g.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (co < g.Cout) {
                    float* q = a.dw + (int64_t)ks_id * a.slab_stride + ((int64_t)g.tap_w[tap] * g.Cout + co) * g.Cin + ci;
                    if (a.assign) *q = acc[i][j][r];
                    else atomicAdd(q, acc[i][j][r]);
                }
            }
        }
}

template <int DT, int WR>
__global__ __launch_bounds__(256) void igemm_wgrad_kernel(const WgradArgs a) {
    igemm_wgrad_body<DT, WR>(a, (int)blockIdx.x);
}

// the weight gradients of several GEMMs (the four sub-pixel classes of a merged transposed conv) in one grid
template <int DT, int WR>
__global__ __launch_bounds__(256) void igemm_wgrad_batch_kernel(const WgradBatchArgs b) {
    int c = 0;
#pragma unroll
    for (int i = 1; i < IGEMM_BATCH_MAX; ++i)
        if (i < b.n && (int)blockIdx.x >= b.start[i]) c = i;
    igemm_wgrad_body<DT, WR>(b.c[c], (int)blockIdx.x - b.start[c]);
}

int check_geom(const GsConvGeom* g, const char* who) {
    GS_CHECK_ARG(g != nullptr, "%s: null geometry", who);
    GS_CHECK_ARG(g->N > 0 && g->IH > 0 && g->IW > 0 && g->OHg > 0 && g->OWg > 0 && g->OH > 0 && g->OW > 0,
                 "%s: non-positive dims", who);
    GS_CHECK_ARG(g->Dg > 0 && g->Din > 0 && g->Dout > 0 && g->isz > 0 && g->osz > 0 && g->ooz >= 0 &&
                     (int64_t)(g->Dg - 1) * g->osz + g->ooz < g->Dout,
                 "%s: bad depth geometry (Dg=%d Din=%d Dout=%d isz=%d osz=%d ooz=%d); 2-D launches use 1,1,1,1,1,0", who,
                 g->Dg, g->Din, g->Dout, g->isz, g->osz, g->ooz);
    GS_CHECK_ARG(g->Cin > 0 && g->Cin % 8 == 0, "%s: Cin=%d must be a positive multiple of 8", who, g->Cin);
    GS_CHECK_ARG(g->Cout > 0, "%s: Cout=%d", who, g->Cout);
    GS_CHECK_ARG(g->ntaps > 0 && g->ntaps <= GS_MAX_TAPS, "%s: ntaps=%d out of range", who, g->ntaps);
    GS_CHECK_ARG(g->in_pix_stride >= g->in_coff + g->Cin && g->in_pix_stride % 8 == 0 && g->in_coff % 8 == 0,
                 "%s: input stride/offset (%d,%d) must cover Cin=%d and be multiples of 8", who, g->in_pix_stride,
                 g->in_coff, g->Cin);
    GS_CHECK_ARG(g->out_pix_stride >= g->out_coff + g->Cout, "%s: output stride %d < coff %d + Cout %d", who,
                 g->out_pix_stride, g->out_coff, g->Cout);
    GS_CHECK_ARG(g->isy > 0 && g->isx > 0 && g->osy > 0 && g->osx > 0 && g->ooy >= 0 && g->oox >= 0,
                 "%s: bad steps/offsets", who);
    GS_CHECK_ARG((int64_t)(g->OHg - 1) * g->osy + g->ooy < g->OH && (int64_t)(g->OWg - 1) * g->osx + g->oox < g->OW,
                 "%s: logical output grid exceeds the physical output tensor", who);
    GS_CHECK_ARG((int64_t)g->N * g->Dg * g->OHg * g->OWg < (int64_t)2147483000 &&
                     (int64_t)g->N * g->Dout * g->OH * g->OW < (int64_t)2147483000 &&
                     (int64_t)g->N * g->Din * g->IH * g->IW < (int64_t)2147483000,
                 "%s: pixel count exceeds int32", who);
    return GS_OK;
}

}  // namespace

constexpr int64_t SPLITK_TILE_FLOATS = 256 * 64;          // 256 threads x (2 x 2 x 16) accumulators of a 128x128 tile

constexpr int SPLITK_MAX_PARTS = 16, SPLITK_MAX_SLABS = 512;
constexpr int64_t SPLITK_CNT_SLOTS = 1024;                // tile ticket counters per GEMM (split launches have <= 128 tiles)

// ksplit for a launch with `tiles` output tiles and up to `ksteps` K steps.  Skinny GEMMs (few M/N tiles, long K: the
// batch-2 Pix2Pix layers) are latency bound -- one block walks 128 K steps of ~0.5 us with 64 blocks on 256 CUs --
// so K is cut until ~512 blocks exist, at most 16 parts.  Row blocks of a tile beyond M are not stored in the slabs.
static int choose_ksplit(int tiles, int ksteps, int64_t ws_floats, int64_t cnt_slots) {
    static const int max_tiles = getenv("GSSEG_SPLITK_TILES") ? atoi(getenv("GSSEG_SPLITK_TILES")) : 128;
    static const int target = getenv("GSSEG_SPLITK_BLOCKS") ? atoi(getenv("GSSEG_SPLITK_BLOCKS")) : 512;
    if (tiles > max_tiles || ksteps < 32) return 1;
    int k = target / tiles;
    static const int min_steps = getenv("GSSEG_SPLITK_MINSTEPS") ? atoi(getenv("GSSEG_SPLITK_MINSTEPS")) : 32;
    // K steps per part: 8 for the really skinny launches, 32 once whole 128-row tiles travel through the slabs (measured
    // on the Pix2Pix step trio: 16 tiles/8 steps 73.2 | 412 img/s at batch 2 | 32; 128 tiles/8 steps 69.3 | 425;
    // 128 tiles/32 steps 74.2 | 436)
    const int per = tiles > 16 ? min_steps : 8;
    if (k > ksteps / per) k = ksteps / per;
    if (k > SPLITK_MAX_PARTS) k = SPLITK_MAX_PARTS;
    while (k > 1 && (int64_t)tiles * k > SPLITK_MAX_SLABS) --k;
    if ((int64_t)tiles * k * SPLITK_TILE_FLOATS > ws_floats || tiles > cnt_slots) return 1;
    return k < 2 ? 1 : k;
}

// taps some logical output pixel can see inside the input (the kernel drops the others from its K loop per tile: at the
// 1x1 .. 4x4 bottom of the generator most of the 16 taps of a class fall outside the feature map)
static int visible_taps(const GsConvGeom& g) {
    auto axis_ok = [](int n_out, int step, int off, int n_in) {
        if (n_out <= 0) return false;
        int o = off < 0 ? (-off + step - 1) / step : 0;             // first output index whose input index is >= 0
        return o < n_out && o * step + off < n_in;
    };
    int n = 0;
    for (int t = 0; t < g.ntaps; ++t)
        if (axis_ok(g.OHg, g.isy, g.tap_dy[t], g.IH) && axis_ok(g.OWg, g.isx, g.tap_dx[t], g.IW) &&
            axis_ok(g.Dg, g.isz, g.tap_dz[t], g.Din)) ++n;
    return n < 1 ? 1 : n;
}

// tile shape, N tiles, block count and K split of one GEMM; ws (may be NULL): [cnt_slots tile counters][slabs]
static int plan_igemm(IgemmArgs& a, float* ws, int64_t ws_floats, int64_t cnt_slots) {
    const int mt = cdiv(a.M, FW_BM);
    // few-tile (weight-streaming) launches take the 64-wide N tile: twice the blocks for the same 16-part split-K limit
    static const int skinny_tiles = getenv("GSSEG_IGEMM_SKINNY") ? atoi(getenv("GSSEG_IGEMM_SKINNY")) : 16;
    const int bn = (a.g.Cout <= 64 || mt * cdiv(a.g.Cout, 128) <= skinny_tiles) ? 64 : 128;
    a.ntn = cdiv(a.g.Cout, bn);
    a.nblocks = mt * a.ntn;
    a.ksplit = 1; a.ws_acc = nullptr; a.ws_cnt = nullptr;
    const bool packed = a.g.Cin == 8 && a.g.ntaps >= 8;
    if (ws != nullptr) {
        // layout of the caller's (zero-initialised, self-cleaning) workspace: [tile counters][tile accumulators]
        const int vt = visible_taps(a.g);
        const int k = choose_ksplit(a.nblocks, packed ? cdiv(vt, 8) : vt * a.kchunks, ws_floats - cnt_slots, cnt_slots);
        if (k > 1) {
            a.ksplit = k;
            a.ws_cnt = reinterpret_cast<unsigned int*>(ws);
            a.ws_acc = ws + cnt_slots;
        }
    }
    return bn;
}

static int launch_igemm(IgemmArgs& a, int dtype, hipStream_t s, const char* who, float* ws = nullptr,
                        int64_t ws_floats = 0) {
    // a single launch uses the first quarter of the workspace (the batched form gives each of its <= 4 GEMMs a quarter)
    const int bn = plan_igemm(a, ws, ws ? ws_floats / IGEMM_BATCH_MAX : 0, SPLITK_CNT_SLOTS);
    const bool packed = a.g.Cin == 8 && a.g.ntaps >= 8;
    a.phase = a.ksplit > 1 ? 1 : 0;
    dim3 grid(a.nblocks * a.ksplit), block(256);
    if (a.y_lo != nullptr) {                      // precise mode: hi/lo output pair, K segments over wrapped input channels
        GS_CHECK_ARG(!packed && a.ksplit == 1 && a.vec_store && a.bnp == nullptr, "%s: precise mode needs 16-byte stores, no split-K", who);
        if (dtype == GS_F16) {
            if (bn == 64) igemm_fwd_kernel<GS_F16, 64, false, true><<<grid, block, 0, s>>>(a);
            else igemm_fwd_kernel<GS_F16, 128, false, true><<<grid, block, 0, s>>>(a);
        } else {
            if (bn == 64) igemm_fwd_kernel<GS_BF16, 64, false, true><<<grid, block, 0, s>>>(a);
            else igemm_fwd_kernel<GS_BF16, 128, false, true><<<grid, block, 0, s>>>(a);
        }
        GS_CHECK_LAUNCH(who);
        return GS_OK;
    }
    auto run = [&](const IgemmArgs& aa, dim3 gr) {
        if (packed) {
            if (dtype == GS_F16) {
                if (bn == 64) igemm_fwd_kernel<GS_F16, 64, true><<<gr, block, 0, s>>>(aa);
                else igemm_fwd_kernel<GS_F16, 128, true><<<gr, block, 0, s>>>(aa);
            } else {
                if (bn == 64) igemm_fwd_kernel<GS_BF16, 64, true><<<gr, block, 0, s>>>(aa);
                else igemm_fwd_kernel<GS_BF16, 128, true><<<gr, block, 0, s>>>(aa);
            }
        } else if (dtype == GS_F16) {
            if (bn == 64) igemm_fwd_kernel<GS_F16, 64, false><<<gr, block, 0, s>>>(aa);
            else igemm_fwd_kernel<GS_F16, 128, false><<<gr, block, 0, s>>>(aa);
        } else {
            if (bn == 64) igemm_fwd_kernel<GS_BF16, 64, false><<<gr, block, 0, s>>>(aa);
            else igemm_fwd_kernel<GS_BF16, 128, false><<<gr, block, 0, s>>>(aa);
        }
    };
    run(a, grid);
    GS_CHECK_LAUNCH(who);
    if (a.ksplit > 1) {                            // phase 2: one block per tile sums the parts and runs the epilogue
        a.phase = 2;
        run(a, dim3(a.nblocks));
        GS_CHECK_LAUNCH(who);
    }
    return GS_OK;
}

// rows of BatchNorm partial sums a launch of this geometry writes: one per 128-row tile, or -- for the geometries the
// weight-streaming form (skinny.hip) covers -- one per 16 output rows
extern "C" int gs_conv_igemm_mtiles(const GsConvGeom* g) {
    if (!g) return GS_EINVAL;
    const int sk = gs_skinny_stat_rows(g);
    if (sk > 0) return sk;
    return (int)cdiv64((int64_t)g->N * (g->Dg > 0 ? g->Dg : 1) * g->OHg * g->OWg, FW_BM);
}

// a covered geometry that ran on the register-staged engine after all (no workspace, mismatched batch): it wrote one row per
// 128-row tile; the rows gs_conv_igemm_mtiles() promised beyond those carry zeros
static int pad_stat_rows(const GsConvGeom* g, float* bnp, hipStream_t s) {
    const int sk = gs_skinny_stat_rows(g);
    if (sk <= 0 || bnp == nullptr) return GS_OK;
    const int wrote = (int)cdiv64((int64_t)g->N * g->OHg * g->OWg, FW_BM);
    if (sk > wrote && hipMemsetAsync(bnp + (int64_t)wrote * 2 * g->Cout, 0, (size_t)(sk - wrote) * 2 * g->Cout * 4, s) != hipSuccess)
        return GS_ELAUNCH;
    return GS_OK;
}

extern "C" int gs_conv_igemm(const GsConvGeom* g, const void* x, const void* w, void* y, const float* bias,
                             float* bn_partials, int act, int dtype, float* splitk_ws, int64_t splitk_ws_floats,
                             void* stream) {
    int rc = check_geom(g, "gs_conv_igemm");
    if (rc) return rc;
    GS_CHECK_ARG(x && w && y, "gs_conv_igemm: null pointer");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_conv_igemm: bad dtype %d", dtype);
    IgemmArgs a;
    a.g = *g;
    a.x = (const unsigned short*)x; a.w = (const unsigned short*)w; a.y = (unsigned short*)y;
    a.bias = bias; a.bnp = bn_partials; a.act = act;
    a.M = g->N * g->Dg * g->OHg * g->OWg;
    a.kchunks = cdiv(g->Cin, FW_BK);
    a.vec_store = (g->Cout % 8 == 0 && g->out_pix_stride % 8 == 0 && g->out_coff % 8 == 0) ? 1 : 0;
    a.shuffle_cout = 0; a.shuffle_cls = 0;
    GS_CHECK_ARG(splitk_ws == nullptr || splitk_ws_floats >= IGEMM_BATCH_MAX * (SPLITK_CNT_SLOTS + SPLITK_TILE_FLOATS),
                 "gs_conv_igemm: split-K workspace too small (gs_conv_igemm_workspace_floats())");
    if (a.vec_store && !(g->Cin == 8 && g->ntaps >= 8)) {        // <= 128 output pixels under a long K: the weight-streaming form
        float* const bnp1[1] = {bn_partials};
        rc = gs_skinny_try(1, &g, x, &w, y, bias, bn_partials ? bnp1 : nullptr, act, dtype, splitk_ws, splitk_ws_floats,
                           (hipStream_t)stream);
        if (rc != GS_EUNSUPPORTED) return rc;
    }
    rc = launch_igemm(a, dtype, (hipStream_t)stream, "gs_conv_igemm", splitk_ws, splitk_ws ? splitk_ws_floats : 0);
    return rc ? rc : pad_stat_rows(g, bn_partials, (hipStream_t)stream);
}

// Split-K workspace of gs_conv_igemm (skinny GEMMs: the 1x1 .. 16x16 levels of the Pix2Pix generator at the script's
// batch size 2 run 4-8 blocks over multi-megabyte weight packs otherwise): an explicit argument of every call, owned by
// the caller (fp32, zero-initialised once; the kernels leave the ticket counters zeroed), NULL = no split.  The library
// keeps no pointer: launches that share one workspace must be ordered on one stream, so the host keeps one per
// (device, stream).
extern "C" int64_t gs_conv_igemm_workspace_floats(void) {
    return IGEMM_BATCH_MAX * (SPLITK_CNT_SLOTS + (int64_t)SPLITK_MAX_SLABS * SPLITK_TILE_FLOATS);
}

// Up to four GEMMs over the same x / y / bias in ONE grid (the sub-pixel classes of a stride-2 transposed convolution, or
// of the data gradient of a stride-2 convolution): g[i], w[i], bn_partials[i] per GEMM, <= 16 taps each, equal Cin / Cout /
// logical pixel counts.  GEMM i splits K through the i-th quarter of the workspace.
extern "C" int gs_conv_igemm_batch(int n, const GsConvGeom* const* g, const void* x, const void* const* w, void* y,
                                   const float* bias, float* const* bn_partials, int act, int dtype, float* splitk_ws,
                                   int64_t splitk_ws_floats, void* stream) {
    GS_CHECK_ARG(n >= 1 && n <= IGEMM_BATCH_MAX && g && w && x && y, "gs_conv_igemm_batch: bad arguments (1..4 GEMMs)");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_conv_igemm_batch: bad dtype %d", dtype);
    GS_CHECK_ARG(splitk_ws == nullptr || splitk_ws_floats >= IGEMM_BATCH_MAX * (SPLITK_CNT_SLOTS + SPLITK_TILE_FLOATS),
                 "gs_conv_igemm_batch: split-K workspace too small (gs_conv_igemm_workspace_floats())");
    {
        bool vec = true;
        for (int i = 0; i < n && vec; ++i)
            vec = g[i] && g[i]->Cout % 8 == 0 && g[i]->out_pix_stride % 8 == 0 && g[i]->out_coff % 8 == 0 && g[i]->Cin % 64 == 0;
        if (vec) {
            for (int i = 0; i < n; ++i) {
                int rc = check_geom(g[i], "gs_conv_igemm_batch");
                if (rc) return rc;
                GS_CHECK_ARG(w[i] != nullptr, "gs_conv_igemm_batch: null weight pointer");
            }
            const int rc = gs_skinny_try(n, g, x, w, y, bias, bn_partials, act, dtype, splitk_ws, splitk_ws_floats, (hipStream_t)stream);
            if (rc != GS_EUNSUPPORTED) return rc;
        }
    }
    IgemmBatchArgs b;
    b.n = n;
    int bn0 = 0, total = 0;
    const int64_t quarter = splitk_ws ? splitk_ws_floats / IGEMM_BATCH_MAX : 0;
    for (int i = 0; i < n; ++i) {
        int rc = check_geom(g[i], "gs_conv_igemm_batch");
        if (rc) return rc;
        GS_CHECK_ARG(w[i] != nullptr, "gs_conv_igemm_batch: null weight pointer");
        GS_CHECK_ARG(g[i]->ntaps <= GEOMC_MAX_TAPS, "gs_conv_igemm_batch: at most %d taps per GEMM", GEOMC_MAX_TAPS);
        GS_CHECK_ARG(!(g[i]->Cin == 8 && g[i]->ntaps >= 8), "gs_conv_igemm_batch: packed-tap (Cin == 8) layers are not batched");
        IgemmArgs a;
        a.g = *g[i];
        a.x = (const unsigned short*)x; a.w = (const unsigned short*)w[i]; a.y = (unsigned short*)y;
        a.bias = bias; a.bnp = bn_partials ? bn_partials[i] : nullptr; a.act = act;
        a.M = g[i]->N * g[i]->Dg * g[i]->OHg * g[i]->OWg;
        a.kchunks = cdiv(g[i]->Cin, FW_BK);
        a.vec_store = (g[i]->Cout % 8 == 0 && g[i]->out_pix_stride % 8 == 0 && g[i]->out_coff % 8 == 0) ? 1 : 0;
        a.shuffle_cout = 0; a.shuffle_cls = 0;
        const int bn = plan_igemm(a, splitk_ws ? splitk_ws + i * quarter : nullptr, quarter, SPLITK_CNT_SLOTS);
        if (i == 0) bn0 = bn;
        GS_CHECK_ARG(bn == bn0, "gs_conv_igemm_batch: the GEMMs of a batch must share one tile shape");
        IgemmArgsC& c = b.c[i];
        const GsConvGeom& s0 = a.g;
        c.g.N = s0.N; c.g.IH = s0.IH; c.g.IW = s0.IW; c.g.Cin = s0.Cin; c.g.in_pix_stride = s0.in_pix_stride; c.g.in_coff = s0.in_coff;
        c.g.OHg = s0.OHg; c.g.OWg = s0.OWg; c.g.Cout = s0.Cout; c.g.OH = s0.OH; c.g.OW = s0.OW;
        c.g.out_pix_stride = s0.out_pix_stride; c.g.out_coff = s0.out_coff;
        c.g.isy = s0.isy; c.g.isx = s0.isx; c.g.osy = s0.osy; c.g.osx = s0.osx; c.g.ooy = s0.ooy; c.g.oox = s0.oox;
        c.g.ntaps = s0.ntaps;
        for (int t = 0; t < GEOMC_MAX_TAPS; ++t) {
            c.g.tap_dy[t] = s0.tap_dy[t]; c.g.tap_dx[t] = s0.tap_dx[t]; c.g.tap_w[t] = s0.tap_w[t]; c.g.tap_dz[t] = s0.tap_dz[t];
        }
        c.g.Dg = s0.Dg; c.g.Din = s0.Din; c.g.Dout = s0.Dout; c.g.isz = s0.isz; c.g.osz = s0.osz; c.g.ooz = s0.ooz;
        c.x = a.x; c.w = a.w; c.y = a.y; c.bias = a.bias; c.bnp = a.bnp; c.act = a.act; c.M = a.M; c.kchunks = a.kchunks;
        c.ntn = a.ntn; c.nblocks = a.nblocks; c.vec_store = a.vec_store; c.shuffle_cout = 0; c.shuffle_cls = 0;
        c.ksplit = a.ksplit; c.ws_acc = a.ws_acc; c.ws_cnt = a.ws_cnt; c.y_lo = nullptr; c.in_wrap = 0;
        c.phase = a.ksplit > 1 ? 1 : 0;
        b.start[i] = total;
        total += a.nblocks * a.ksplit;
    }
    for (int i = n; i <= IGEMM_BATCH_MAX; ++i) b.start[i] = total;
    hipStream_t s = (hipStream_t)stream;
    auto run = [&](const IgemmBatchArgs& bb, int blocks) {
        if (dtype == GS_F16) {
            if (bn0 == 64) igemm_fwd_batch_kernel<GS_F16, 64><<<blocks, 256, 0, s>>>(bb);
            else igemm_fwd_batch_kernel<GS_F16, 128><<<blocks, 256, 0, s>>>(bb);
        } else {
            if (bn0 == 64) igemm_fwd_batch_kernel<GS_BF16, 64><<<blocks, 256, 0, s>>>(bb);
            else igemm_fwd_batch_kernel<GS_BF16, 128><<<blocks, 256, 0, s>>>(bb);
        }
    };
    run(b, total);
    GS_CHECK_LAUNCH("gs_conv_igemm_batch");
    // phase 2 for the GEMMs that split K: one block per tile sums the parts and runs the epilogue (the others are done)
    int total2 = 0;
    for (int i = 0; i < n; ++i) {
        b.start[i] = total2;
        if (b.c[i].ksplit > 1) { b.c[i].phase = 2; total2 += b.c[i].nblocks; }
    }
    for (int i = n; i <= IGEMM_BATCH_MAX; ++i) b.start[i] = total2;
    if (total2 > 0) {
        run(b, total2);
        GS_CHECK_LAUNCH("gs_conv_igemm_batch");
    }
    for (int i = 0; i < n && bn_partials; ++i) {
        const int rc = pad_stat_rows(g[i], bn_partials[i], s);
        if (rc) return rc;
    }
    return GS_OK;
}

// Merged stride-2 / kernel-2 transposed convolution (unet_parts.py:51 ConvTranspose2d(C, C/2, 2, 2);
// GenSeg-3D/UNet3D/unet3d.py:68 ConvTranspose3d(k=2, s=2)): every input voxel produces a 2x2(x2) output patch,
// i.e. ONE pointwise GEMM  [pixels x Cin] . [Cin x ncls*Cout]  with a sub-pixel scatter in the epilogue --
// the input is read once instead of once per sub-pixel class.
//   x  [N*D, IH, IW, Cin] 16-bit NHWC (strided);  w = gs_pack_weight slots [ncls][Cout][Cin], slot = (kz*2+ky)*2+kx
//   y  [N*Dout, OH, OW, *] : voxel (2z+kz+ooz, 2y+ky+ooy, 2x+kx+oox), channels out_coff..out_coff+Cout
static int upconv2x2_impl(const void* x, const void* w, const float* bias, void* y, void* y_lo, int in_wrap, int N, int D,
                          int IH, int IW, int Cin, int in_pix_stride, int in_coff, int Cout, int Dout, int OH, int OW,
                          int out_pix_stride, int out_coff, int ooz, int ooy, int oox, int act, int dtype, void* stream);
int pw_upconv2x2_fwd_fast(const void* x, const void* w, const float* bias, void* y, int N, int IH, int IW, int Cin,
                          int in_pix_stride, int in_coff, int Cout, int OH, int OW, int out_pix_stride, int out_coff, int ooy,
                          int oox, int act, int dtype, void* stream, int D, int Dout, int ooz);

extern "C" int gs_upconv2x2_fwd(const void* x, const void* w, const float* bias, void* y, int N, int D, int IH, int IW,
                                int Cin, int in_pix_stride, int in_coff, int Cout, int Dout, int OH, int OW,
                                int out_pix_stride, int out_coff, int ooz, int ooy, int oox, int act, int dtype,
                                void* stream) {
    return upconv2x2_impl(x, w, bias, y, nullptr, 0, N, D, IH, IW, Cin, in_pix_stride, in_coff, Cout, Dout, OH, OW,
                          out_pix_stride, out_coff, ooz, ooy, oox, act, dtype, stream);
}

// Precise-mode form: x holds `in_wrap` channels per pixel (the [hi | lo] planes of the activation), w is
// the [ncls][Cout][K] pack of matching segments ([w_hi | w_hi | w_lo]), K channel ci reads input channel ci mod in_wrap;
// the up-sampled tensor leaves as the pair y_hi / y_lo (same stride and offset).
extern "C" int gs_upconv2x2_fwd_precise(const void* x, const void* w, const float* bias, void* y_hi, void* y_lo, int N,
                                        int IH, int IW, int K, int in_pix_stride, int in_coff, int in_wrap, int Cout,
                                        int OH, int OW, int out_pix_stride, int out_coff, int ooy, int oox, int dtype,
                                        void* stream) {
    GS_CHECK_ARG(y_lo != nullptr && in_wrap > 0 && in_wrap % 8 == 0 && K >= in_wrap && K <= 2 * in_wrap,
                 "gs_upconv2x2_fwd_precise: need y_lo and in_wrap <= K <= 2*in_wrap");
    GS_CHECK_ARG(in_pix_stride >= in_coff + in_wrap, "gs_upconv2x2_fwd_precise: input stride smaller than the wrapped planes");
    return upconv2x2_impl(x, w, bias, y_hi, y_lo, in_wrap, N, 1, IH, IW, K, in_pix_stride, in_coff, Cout, 1, OH, OW,
                          out_pix_stride, out_coff, 0, ooy, oox, GS_ACT_NONE, dtype, stream);
}

static int upconv2x2_impl(const void* x, const void* w, const float* bias, void* y, void* y_lo, int in_wrap, int N, int D,
                          int IH, int IW, int Cin, int in_pix_stride, int in_coff, int Cout, int Dout, int OH, int OW,
                          int out_pix_stride, int out_coff, int ooz, int ooy, int oox, int act, int dtype, void* stream) {
    GS_CHECK_ARG(x && w && y, "gs_upconv2x2_fwd: null pointer");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_upconv2x2_fwd: bad dtype %d", dtype);
    GS_CHECK_ARG(D >= 1 && Dout >= 1, "gs_upconv2x2_fwd: bad depth");
    const bool is3d = D > 1 || Dout > 1;
    const int ncls = is3d ? 8 : 4;
    GS_CHECK_ARG(Cout > 0 && Cout % 8 == 0 && out_pix_stride % 8 == 0 && out_coff % 8 == 0 &&
                     out_pix_stride >= out_coff + Cout,
                 "gs_upconv2x2_fwd: Cout / output stride / offset must be multiples of 8");
    GS_CHECK_ARG(2 * IH - 1 + ooy < OH && 2 * IW - 1 + oox < OW && (is3d ? 2 * D - 1 + ooz < Dout : (Dout == 1 && ooz == 0)),
                 "gs_upconv2x2_fwd: output patch exceeds the output tensor");
    if (y_lo == nullptr && in_wrap == 0 && in_pix_stride >= in_coff + Cin && ooy >= 0 && oox >= 0) {
        // LDS-DMA pointwise GEMM (pwgemm.hip) for the shapes it covers (2-D and 3-D); everything else: the generic engine below
        const int rc = pw_upconv2x2_fwd_fast(x, w, bias, y, N, IH, IW, Cin, in_pix_stride, in_coff, Cout, OH, OW, out_pix_stride,
                                             out_coff, ooy, oox, act, dtype, stream, D, Dout, ooz);
        if (rc == 1) return GS_OK;
        if (rc != 0) return GS_ELAUNCH;
    }
    GsConvGeom g{};
    g.N = N; g.IH = IH; g.IW = IW; g.Cin = Cin; g.in_pix_stride = in_pix_stride; g.in_coff = in_coff;
    g.OHg = IH; g.OWg = IW; g.Cout = ncls * Cout; g.OH = OH; g.OW = OW;
    g.out_pix_stride = ncls * Cout > out_pix_stride ? ncls * Cout : out_pix_stride;   // checked below with the real one
    g.out_coff = 0;
    g.isy = 1; g.isx = 1; g.osy = 2; g.osx = 2; g.ooy = ooy; g.oox = oox;
    g.ntaps = 1; g.tap_dy[0] = 0; g.tap_dx[0] = 0; g.tap_w[0] = 0; g.tap_dz[0] = 0;
    g.Dg = D; g.Din = D; g.Dout = Dout; g.isz = 1; g.osz = is3d ? 2 : 1; g.ooz = ooz;
    // geometry check on the logical (widened) GEMM, then the physical output layout is put back
    {
        GsConvGeom c = g;
        c.OH = OH + 1; c.OW = OW + 1; c.Dout = Dout + 1;      // the +1 sub-pixel is checked explicitly above
        if (in_wrap > 0) c.Cin = in_wrap;                     // precise mode: the input holds in_wrap channels, K = Cin wraps
        int rc = check_geom(&c, "gs_upconv2x2_fwd");
        if (rc) return rc;
    }
    g.out_pix_stride = out_pix_stride; g.out_coff = out_coff;
    IgemmArgs a;
    a.g = g;
    a.x = (const unsigned short*)x; a.w = (const unsigned short*)w; a.y = (unsigned short*)y;
    a.bias = bias; a.bnp = nullptr; a.act = act;
    a.M = N * D * IH * IW;
    a.kchunks = cdiv(Cin, FW_BK);
    a.vec_store = 1; a.shuffle_cout = Cout; a.shuffle_cls = ncls;
    a.y_lo = (unsigned short*)y_lo; a.in_wrap = in_wrap;
    return launch_igemm(a, dtype, (hipStream_t)stream, "gs_upconv2x2_fwd");
}

// plan only (x == nullptr): returns the K split of this geometry through *ksplit_out
static int conv_wgrad_launch(const GsConvGeom* g, const void* x, const void* dy, float* dw, int dtype, void* stream,
                             int assign, int* ksplit_out, int64_t slab_stride = 0, WgradArgs* batch_out = nullptr) {
    int rc = check_geom(g, "gs_conv_wgrad");
    if (rc) return rc;
    const bool plan = ksplit_out != nullptr;
    GS_CHECK_ARG(plan || (x && dy && dw), "gs_conv_wgrad: null pointer");
    GS_CHECK_ARG(g->Cout % 8 == 0 && g->out_pix_stride % 8 == 0 && g->out_coff % 8 == 0,
                 "gs_conv_wgrad: Cout/out stride/offset must be multiples of 8");
    GS_CHECK_ARG(plan || dtype == GS_F16 || dtype == GS_BF16, "gs_conv_wgrad: bad dtype %d", dtype);
    WgradArgs a;
    a.g = *g;
    a.x = (const unsigned short*)x; a.dy = (const unsigned short*)dy; a.dw = dw;
    a.M = g->N * g->Dg * g->OHg * g->OWg;
    a.kchunks = cdiv(g->Cin, 64);
    a.ncb = g->ntaps * a.kchunks;
    const int wr = (g->Cout <= 64) ? 1 : 2;
    const int wc = 4 / wr;
    a.n_cotiles = cdiv(g->Cout, 64 * wr);
    a.n_cbgroups = cdiv(a.ncb, wc);
    const int base_blocks = a.n_cotiles * a.n_cbgroups;
    const int ksteps = cdiv(a.M, WG_KP);
    static const int wg_target = getenv("GSSEG_WG_BLOCKS") ? atoi(getenv("GSSEG_WG_BLOCKS")) : 512;
    // ~512 blocks (2 per CU): every K part adds its tile to dw with fp32 atomics, and with 1024 blocks the same-address
    // adds cost more than the occupancy gains (up-conv wgrad 0.64 -> 0.50 ms/step, Pix2Pix trio 75.6 -> 80 img/s at batch 2)
    int ksplit = cdiv(wg_target, base_blocks);
    if (ksplit > ksteps) ksplit = ksteps;
    if (ksplit < 1) ksplit = 1;
    a.kper = cdiv(ksteps, ksplit) * WG_KP;
    a.ksplit = cdiv(a.M, a.kper);
    if (plan) { *ksplit_out = a.ksplit; return GS_OK; }
    GS_CHECK_ARG(!assign || slab_stride > 0 || a.ksplit == 1,
                 "gs_conv_wgrad_assign: this geometry splits K (%d parts): zero dw and use gs_conv_wgrad", a.ksplit);
    a.assign = assign;
    a.slab_stride = slab_stride;
    const int ohw = g->OHg * g->OWg;
    a.d_n = WG_KP / ohw;
    const int rem = WG_KP - a.d_n * ohw;
    a.d_oy = rem / g->OWg;
    a.d_ox = rem - a.d_oy * g->OWg;
    if (batch_out != nullptr) {                 // batched launch: hand the planned arguments back
        *batch_out = a;
        return GS_OK;
    }
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(base_blocks * a.ksplit), block(256);
    if (dtype == GS_F16) {
        if (wr == 1) igemm_wgrad_kernel<GS_F16, 1><<<grid, block, 0, s>>>(a);
        else igemm_wgrad_kernel<GS_F16, 2><<<grid, block, 0, s>>>(a);
    } else {
        if (wr == 1) igemm_wgrad_kernel<GS_BF16, 1><<<grid, block, 0, s>>>(a);
        else igemm_wgrad_kernel<GS_BF16, 2><<<grid, block, 0, s>>>(a);
    }
    GS_CHECK_LAUNCH("gs_conv_wgrad");
    return GS_OK;
}

extern "C" int gs_conv_wgrad(const GsConvGeom* g, const void* x, const void* dy, float* dw, int dtype, void* stream) {
    return conv_wgrad_launch(g, x, dy, dw, dtype, stream, 0, nullptr);
}

extern "C" int gs_conv_wgrad_single_pass(const GsConvGeom* g) {
    int ks = 0;
    if (conv_wgrad_launch(g, nullptr, nullptr, nullptr, 0, nullptr, 0, &ks) != GS_OK) return 0;
    return ks == 1 ? 1 : 0;
}

// Deterministic form: the K parts of the launch store their partial gradients in slabs ws[part][tap][Cout][Cin] (no
// zero fill, no atomics); gs_wgrad_reduce_unpack sums them in part order into the reference layout.
static int64_t wgrad_slab_floats(const GsConvGeom* g) {
    int tw = 0;
    for (int t = 0; t < g->ntaps; ++t) tw = g->tap_w[t] > tw ? g->tap_w[t] : tw;
    return (int64_t)(tw + 1) * g->Cout * g->Cin;
}

extern "C" int gs_conv_wgrad_parts(const GsConvGeom* g) {
    int ks = 0;
    if (conv_wgrad_launch(g, nullptr, nullptr, nullptr, 0, nullptr, 0, &ks) != GS_OK) return 0;
    return ks;
}

extern "C" int64_t gs_conv_wgrad_ws_floats(const GsConvGeom* g) {
    const int parts = gs_conv_wgrad_parts(g);
    return parts > 0 ? (int64_t)parts * wgrad_slab_floats(g) : 0;
}

extern "C" int gs_conv_wgrad_slabs(const GsConvGeom* g, const void* x, const void* dy, float* ws, int dtype, void* stream) {
    GS_CHECK_ARG(g != nullptr, "gs_conv_wgrad_slabs: null geometry");
    return conv_wgrad_launch(g, x, dy, ws, dtype, stream, 1, nullptr, wgrad_slab_floats(g));
}

extern "C" int gs_conv_wgrad_assign(const GsConvGeom* g, const void* x, const void* dy, float* dw, int dtype, void* stream) {
    return conv_wgrad_launch(g, x, dy, dw, dtype, stream, 1, nullptr);
}

// The weight gradients of up to four GEMMs over the same x / dy in ONE grid (the sub-pixel classes of a merged transposed
// convolution): GEMM i writes its K parts to ws + part * (n * slab) + i * slab, slab = gs_conv_wgrad_ws_floats(g[i]) /
// gs_conv_wgrad_parts(g[i]) floats (equal for all i) -- i.e. one slab of all n gradients per part, so that ONE
// gs_wgrad_reduce_unpack over [n * taps * Cout][Cin] sums every class; with one part the "workspace" is the gradient itself.
extern "C" int gs_conv_wgrad_slabs_batch(int n, const GsConvGeom* const* g, const void* x, const void* dy, float* ws, int dtype,
                                         void* stream) {
    GS_CHECK_ARG(n >= 1 && n <= IGEMM_BATCH_MAX && g && x && dy && ws, "gs_conv_wgrad_slabs_batch: bad arguments (1..4 GEMMs)");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_conv_wgrad_slabs_batch: bad dtype %d", dtype);
    WgradBatchArgs b;
    b.n = n;
    int total = 0, wr0 = 0, parts0 = 0;
    int64_t slab0 = 0;
    for (int i = 0; i < n; ++i) {
        GS_CHECK_ARG(g[i] != nullptr && g[i]->ntaps <= GEOMC_MAX_TAPS, "gs_conv_wgrad_slabs_batch: at most %d taps per GEMM", GEOMC_MAX_TAPS);
        const int64_t slab = wgrad_slab_floats(g[i]);
        if (i == 0) slab0 = slab;
        GS_CHECK_ARG(slab == slab0, "gs_conv_wgrad_slabs_batch: the GEMMs of a batch must have equal gradient sizes");
        WgradArgs a;
        int rc = conv_wgrad_launch(g[i], x, dy, ws + (int64_t)i * slab0, dtype, stream, 1, nullptr, (int64_t)n * slab0, &a);
        if (rc) return rc;
        const int wr = (g[i]->Cout <= 64) ? 1 : 2;
        if (i == 0) { wr0 = wr; parts0 = a.ksplit; }
        GS_CHECK_ARG(wr == wr0 && a.ksplit == parts0, "gs_conv_wgrad_slabs_batch: the GEMMs of a batch must share tile shape and K split");
        WgradArgsC& c = b.c[i];
        const GsConvGeom& s0 = a.g;
        c.g.N = s0.N; c.g.IH = s0.IH; c.g.IW = s0.IW; c.g.Cin = s0.Cin; c.g.in_pix_stride = s0.in_pix_stride; c.g.in_coff = s0.in_coff;
        c.g.OHg = s0.OHg; c.g.OWg = s0.OWg; c.g.Cout = s0.Cout; c.g.OH = s0.OH; c.g.OW = s0.OW;
        c.g.out_pix_stride = s0.out_pix_stride; c.g.out_coff = s0.out_coff;
        c.g.isy = s0.isy; c.g.isx = s0.isx; c.g.osy = s0.osy; c.g.osx = s0.osx; c.g.ooy = s0.ooy; c.g.oox = s0.oox;
        c.g.ntaps = s0.ntaps;
        for (int t = 0; t < GEOMC_MAX_TAPS; ++t) {
            c.g.tap_dy[t] = s0.tap_dy[t]; c.g.tap_dx[t] = s0.tap_dx[t]; c.g.tap_w[t] = s0.tap_w[t]; c.g.tap_dz[t] = s0.tap_dz[t];
        }
        c.g.Dg = s0.Dg; c.g.Din = s0.Din; c.g.Dout = s0.Dout; c.g.isz = s0.isz; c.g.osz = s0.osz; c.g.ooz = s0.ooz;
        c.x = a.x; c.dy = a.dy; c.dw = a.dw; c.M = a.M; c.kchunks = a.kchunks; c.ncb = a.ncb;
        c.n_cotiles = a.n_cotiles; c.n_cbgroups = a.n_cbgroups; c.ksplit = a.ksplit; c.kper = a.kper;
        c.d_n = a.d_n; c.d_oy = a.d_oy; c.d_ox = a.d_ox; c.assign = 1; c.slab_stride = a.slab_stride;
        b.start[i] = total;
        total += a.n_cotiles * a.n_cbgroups * a.ksplit;
    }
    for (int i = n; i <= IGEMM_BATCH_MAX; ++i) b.start[i] = total;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GS_F16) {
        if (wr0 == 1) igemm_wgrad_batch_kernel<GS_F16, 1><<<total, 256, 0, s>>>(b);
        else igemm_wgrad_batch_kernel<GS_F16, 2><<<total, 256, 0, s>>>(b);
    } else {
        if (wr0 == 1) igemm_wgrad_batch_kernel<GS_BF16, 1><<<total, 256, 0, s>>>(b);
        else igemm_wgrad_batch_kernel<GS_BF16, 2><<<total, 256, 0, s>>>(b);
    }
    GS_CHECK_LAUNCH("gs_conv_wgrad_slabs_batch");
    return GS_OK;
}
