// Weight-streaming form of the implicit-GEMM convolution for launches with FEW output pixels and LONG K: the 1x1 .. 8x8 levels of
// the Pix2Pix U-Net generator and the deep PatchGAN layers at the script's batch size 2 (running_files/train_end2end_jsrt.py:
// batch_size 2; models_pix2pix/networks.py:582 Conv2d(k4,s2), :486-511 the merged k8/s2 transposed convolution and their data
// gradients).  There M = N*OH*OW is 2 .. 128 rows while the layer's weights are 8 .. 67 MB: the GEMM is a stream over the
// weight pack, and the register-staged engine (igemm.hip) -- one K step of 8-16 KB in flight per block, a barrier per step --
// moved it at 0.1-0.2 TB/s (80-120 us per layer, profiles/r03_cfg_g2_timeline.txt).
//
//   Y[m][co] = sum_{tap visible, ci} X[inpix(m, tap)][ci] * W[slot(tap)][co][ci]          (same GsConvGeom tap-list form)
//
//   block   : 4 waves = 128 couts (32 per wave) x one tile of <= 128 rows (MT = 1 / 2 / 4 accumulator tiles per wave) x one K part
//             (M <= 128: ONE row tile, the weights are streamed once; larger M -- up to 8192 rows, the 16x16 .. 64x64 levels --
//             adds row tiles that re-read the chunk from L2)
//   K chunk : 64 channels of one visible tap: A = the M gathered pixel rows (128 B each, zeros outside the image via the
//             buffer bounds check), B = 128 weight rows of 128 B.  Both land in LDS by `buffer_load_dwordx4 ... lds` pieces
//             (8 rows x 128 B, fully coalesced) into a FOUR-deep ring: three chunks (60-96 KB per CU) are in flight while one
//             is multiplied -- Little's law for 6 TB/s at ~1.5 us wants ~35 KB per CU.
//   LDS     : 128-byte rows, 16-byte slot s of row r holds the logical slot s ^ (r & 7) (source side of the DMA): the
//             ds_read_b128 of the 32x32x16 operands is conflict free.
//   split K : the visible (tap, chunk) list is cut into `parts` ranges so that ~2 blocks per CU exist; part p stores its fp32
//             rows in slab [gemm][p][M][Cout]; skinny_reduce_kernel (16 rows x 64 channels per block) sums the slabs in part order
//             (deterministic, no atomics), adds bias, applies the activation, stores 16-bit NHWC and writes one BatchNorm partial
//             row per 16 output rows (sums before bias and rounding, as gs_conv_igemm does): gs_conv_igemm_mtiles() reports
//             that row count for the launches this form covers.
// Taps that no output pixel can see (75-94 % of them at the 1x1 .. 2x2 levels) are dropped on the host: their weights are never
// read.  Up to four GEMMs (the sub-pixel classes of a layer) share one launch of each kernel.
#include <stdlib.h>

#include "common.hpp"
#include "skinny.hpp"

namespace {

constexpr int SK_MAX_GEMMS = 4;
constexpr int SK_RING = 4;
constexpr unsigned SK_VOOB = 0x80000000u;

struct SkGemm {
    const unsigned short* w;      // pack [slot][Cout][Cin]
    float* bnp;                   // one row [2][Cout] or NULL
    float* slab;                  // [parts][M][Cout] fp32
    unsigned w_bytes;
    int M;
    int IH, IW, OHg, OWg, OH, OW, isy, isx, osy, osx, ooy, oox;
    int nq, parts, qpp;           // K chunks (visible taps x Cin / 64), K parts, chunks per part
    int mtiles;                   // row tiles of 128
    int blk0;                     // first block of this GEMM in the stream grid
    int nvis;
    unsigned tap[GS_MAX_TAPS];    // visible taps: (dy + 128) | (dx + 128) << 8 | slot << 16
};
struct SkArgs {
    SkGemm g[SK_MAX_GEMMS];
    int n;
    const unsigned short* x;
    unsigned short* y;
    const float* bias;
    unsigned x_bytes;
    int N, Cin, in_stride, in_coff, Cout, out_stride, out_coff, act, ntn, kch;
};
static_assert(sizeof(SkArgs) <= 3900, "kernel-argument segment");

__device__ __forceinline__ void sk_dma_piece16(const __amdgpu_buffer_rsrc_t& rs, unsigned char* dst, unsigned voff, unsigned soff) {
    const unsigned lds_addr = (unsigned)(size_t)(LDS_AS void*)dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}

template <int DT, int MT>
__global__ __launch_bounds__(256) void skinny_stream_kernel(const SkArgs a) {
    typedef typename Elem<DT>::V8 V8;
    constexpr int A_B = MT * 4096, B_B = 128 * 128, SLOT_B = A_B + B_B;
    constexpr int PPW = MT + 4;                        // DMA pieces per wave per chunk
    static_assert(SK_RING * SLOT_B <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SK_RING * SLOT_B];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    int c = 0;
#pragma unroll
    for (int i = 1; i < SK_MAX_GEMMS; ++i)
        if (i < a.n && (int)blockIdx.x >= a.g[i].blk0) c = i;
    const SkGemm& g = a.g[c];
    const int local = (int)blockIdx.x - g.blk0;
    const int part = local % g.parts, rest = local / g.parts;
    const int mtile = rest % g.mtiles, ntile = rest / g.mtiles;
    const int n0 = ntile * 128, m0 = mtile * 128;
    const int q0 = part * g.qpp;
    const int q1 = min(g.nq, q0 + g.qpp);
    const int nchunk = q1 - q0;

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g.w, 0, g.w_bytes, 0x00020000);

    // ---- DMA side: lane l of a piece = row l >> 3, physical 16-byte slot l & 7 <- logical slot (l & 7) ^ (l >> 3) ----
    const int prow = lane >> 3, lslot = (lane & 7) ^ prow;
    // A pieces of this wave: piece pa = wave + 4 j covers rows 8 pa .. 8 pa + 7
    int a_iy0[MT], a_ix0[MT], a_n[MT];
    const int ohw = g.OHg * g.OWg;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = m0 + (wave + 4 * j) * 8 + prow;
        if (m < g.M) {
            const int n = m / ohw, rem = m - n * ohw;
            const int oy = rem / g.OWg, ox = rem - oy * g.OWg;
            a_n[j] = n; a_iy0[j] = oy * g.isy; a_ix0[j] = ox * g.isx;
        } else {
            a_n[j] = 0; a_iy0[j] = -(1 << 20); a_ix0[j] = 0;
        }
    }
    unsigned b_voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co = n0 + wave * 32 + j * 8 + prow;
        b_voff[j] = co < a.Cout ? (unsigned)((co * a.Cin + lslot * 8) * 2) : SK_VOOB;       // Cout % 128 != 0: the tail waves multiply zeros
    }
    const unsigned tap_stride = (unsigned)a.Cout * (unsigned)a.Cin * 2u;

    // running (tap, channel chunk) of the chunk being ISSUED and the tap's per-lane pixel offsets
    int iq = q0;
    int itap = q0 / a.kch, icc = q0 - itap * a.kch;
    unsigned a_voff[MT], wslot_off = 0u;
    auto set_tap = [&](int tp) __attribute__((always_inline)) {
        const unsigned tv = g.tap[tp < g.nvis ? tp : 0];
        const int dy = (int)(tv & 0xffu) - 128, dx = (int)((tv >> 8) & 0xffu) - 128;
        wslot_off = (tv >> 16) * tap_stride;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int iy = a_iy0[j] + dy, ix = a_ix0[j] + dx;
            const bool ok = (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW;
            a_voff[j] = ok ? (unsigned)((((a_n[j] * g.IH + iy) * g.IW + ix) * a.in_stride + a.in_coff + lslot * 8) * 2) : SK_VOOB;
        }
    };
    set_tap(itap);
    auto issue_chunk = [&](unsigned slot) __attribute__((always_inline)) {
        const unsigned kill = iq < q1 ? 0u : SK_VOOB;
        const unsigned cso = (unsigned)icc * 128u;
        unsigned char* base = smem + slot * SLOT_B;
#pragma unroll
        for (int j = 0; j < MT; ++j) sk_dma_piece16(x_rsrc, base + (wave + 4 * j) * 1024, a_voff[j] | kill, cso);
#pragma unroll
        for (int j = 0; j < 4; ++j) sk_dma_piece16(w_rsrc, base + A_B + (wave * 4 + j) * 1024, b_voff[j] | kill, wslot_off + cso);
        ++iq;
        if (++icc == a.kch) { icc = 0; ++itap; set_tap(itap); }
    };

    // ---- MFMA side ----
    const unsigned hs = (unsigned)(h ^ (l31 & 7)) << 4;
    const unsigned a_rd = (unsigned)(l31 * 128) ^ hs;                             // + i * 4096 + (kk << 5) (xor: bits 5..6)
    const unsigned b_rd = (unsigned)(A_B + (wave * 32 + l31) * 128) ^ hs;
    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    issue_chunk(0u); issue_chunk(1u); issue_chunk(2u);
    for (int i = 0; i < nchunk; ++i) {
        // chunk i has landed when at most the pieces of chunks i+1, i+2 are outstanding (completion counts in issue order)
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PPW) : "memory");
        __builtin_amdgcn_s_barrier();                  // ... everybody's pieces; slot (i + 3) % 4 was read by everyone (chunk i - 1)
        asm volatile("" ::: "memory");
        issue_chunk((unsigned)((i + 3) % SK_RING));
        const unsigned char* sb = smem + (i % SK_RING) * SLOT_B;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const V8 bf = *reinterpret_cast<const V8*>(sb + (b_rd ^ (unsigned)(kk << 5)));
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                const V8 af = *reinterpret_cast<const V8*>(sb + mi * 4096 + (a_rd ^ (unsigned)(kk << 5)));
                acc[mi] = Elem<DT>::mfma32(af, bf, acc[mi]);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dead pieces of the tail

    // ---- the part's rows: slab[part][m][co], lanes along co (two 128-byte runs per store) ----
    const int co = n0 + wave * 32 + l31;
    float* slab = g.slab + (int64_t)part * g.M * a.Cout + co;
    if (co < a.Cout) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < g.M) slab[(int64_t)m * a.Cout] = acc[mi][r];
            }
    }
}

// one block = 16 output rows x 64 channels (16 lanes x 4) of one GEMM
template <int DT>
__global__ __launch_bounds__(256) void skinny_reduce_kernel(const SkArgs a) {
    __shared__ float red[2][16][65];
    const SkGemm& g = a.g[blockIdx.z];
    const int t = threadIdx.x, cq = t & 15, rl = t >> 4;
    const int c0 = blockIdx.x * 64 + cq * 4;
    const int m = blockIdx.y * 16 + rl;
    if (blockIdx.y * 16 >= g.M) return;                // (GEMMs of a batch share M; kept for safety)
    const bool live = m < g.M && c0 < a.Cout;
    float f[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const int64_t pstride = (int64_t)g.M * a.Cout;
        const float* q = g.slab + (int64_t)m * a.Cout + c0;
        float4 v = *reinterpret_cast<const float4*>(q);
        constexpr int U = 8;
        int p = 1;
        for (; p + U <= g.parts; p += U) {             // U loads in flight, added in part order
            float4 u[U];
#pragma unroll
            for (int k = 0; k < U; ++k) u[k] = *reinterpret_cast<const float4*>(q + (int64_t)(p + k) * pstride);
#pragma unroll
            for (int k = 0; k < U; ++k) { v.x += u[k].x; v.y += u[k].y; v.z += u[k].z; v.w += u[k].w; }
        }
        for (; p < g.parts; ++p) {
            const float4 u = *reinterpret_cast<const float4*>(q + (int64_t)p * pstride);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    }
    if (g.bnp != nullptr) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { red[0][rl][cq * 4 + i] = f[i]; red[1][rl][cq * 4 + i] = f[i] * f[i]; }
    }
    if (live) {
        const float slope = a.act == GS_ACT_RELU ? 0.f : (a.act == GS_ACT_LEAKY02 ? 0.2f : 1.f);
        const bool is_tanh = a.act == GS_ACT_TANH;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float o = f[i] + (a.bias ? a.bias[c0 + i] : 0.f);
            o = o > 0.f ? o : o * slope;
            if (is_tanh) o = tanhf(o);
            f[i] = o;
        }
        const int ohw = g.OHg * g.OWg;
        const int n = m / ohw, rem = m - n * ohw;
        const int oy = rem / g.OWg, ox = rem - oy * g.OWg;
        const int64_t p = ((int64_t)n * g.OH + oy * g.osy + g.ooy) * g.OW + ox * g.osx + g.oox;
        uint2 o2;
        o2.x = Elem<DT>::pack2(f[0], f[1]);
        o2.y = Elem<DT>::pack2(f[2], f[3]);
        *reinterpret_cast<uint2*>(a.y + p * a.out_stride + a.out_coff + c0) = o2;
    }
    if (g.bnp != nullptr) {                            // fixed-order sum over the 16 rows -> partial row blockIdx.y
        __syncthreads();
        if (t < 128) {
            const int st = t >> 6, ch = t & 63;
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += red[st][r][ch];
            if (blockIdx.x * 64 + ch < a.Cout) g.bnp[((int64_t)blockIdx.y * 2 + st) * a.Cout + blockIdx.x * 64 + ch] = s;
        }
    }
}

bool sk_axis_ok(int n_out, int step, int off, int n_in) {
    if (n_out <= 0) return false;
    const int o = off < 0 ? (-off + step - 1) / step : 0;             // first output index whose input index is >= 0
    return o < n_out && o * step + off < n_in;
}

// geometry part of the eligibility (a pure function of the geometry and the process environment: gs_conv_igemm_mtiles relies on it)
bool sk_geom_ok(const GsConvGeom& g, int* nvis_out) {
    static const int env = getenv("GSSEG_SKINNY") ? atoi(getenv("GSSEG_SKINNY")) : 1;
    static const int max_m = getenv("GSSEG_SKINNY_MAXM") ? atoi(getenv("GSSEG_SKINNY_MAXM")) : 8192;
    if (!env) return false;
    if (g.Dg != 1 || g.Din != 1 || g.Dout != 1) return false;
    if (g.Cin % 64 || g.Cout % 32 || g.out_pix_stride % 8 || g.out_coff % 8 || g.in_pix_stride % 8 || g.in_coff % 8) return false;
    if (g.ntaps < 1 || g.ntaps > GS_MAX_TAPS) return false;
    const int64_t M = (int64_t)g.N * g.OHg * g.OWg;
    if (M < 1 || M > max_m) return false;
    if ((int64_t)g.N * g.IH * g.IW * g.in_pix_stride * 2 >= 2147483000LL) return false;
    int nv = 0, max_slot = 0;
    for (int t = 0; t < g.ntaps; ++t) {
        if (g.tap_dy[t] < -127 || g.tap_dy[t] > 127 || g.tap_dx[t] < -127 || g.tap_dx[t] > 127 || g.tap_w[t] < 0 || g.tap_w[t] > 0xffff)
            return false;
        if (sk_axis_ok(g.OHg, g.isy, g.tap_dy[t], g.IH) && sk_axis_ok(g.OWg, g.isx, g.tap_dx[t], g.IW)) {
            ++nv;
            if (g.tap_w[t] > max_slot) max_slot = g.tap_w[t];
        }
    }
    if (nv * (g.Cin / 64) < 4) return false;                         // a K of < 4 chunks: nothing to pipeline
    if ((int64_t)(max_slot + 1) * g.Cout * g.Cin * 2 >= 4294967000LL) return false;
    if (nvis_out) *nvis_out = nv;
    return true;
}

}  // namespace

// BatchNorm partial rows a gs_conv_igemm / gs_conv_igemm_batch launch of this geometry writes when the weight-streaming form
// takes it (one per 16 output rows); 0: the geometry is not covered
int gs_skinny_stat_rows(const GsConvGeom* g) {
    if (!g || !sk_geom_ok(*g, nullptr)) return 0;
    return cdiv(g->N * g->OHg * g->OWg, 16);
}

// GS_OK: handled; GS_EUNSUPPORTED: not covered (the caller runs the register-staged engine; nothing was launched)
int gs_skinny_try(int n, const GsConvGeom* const* gg, const void* x, const void* const* w, void* y, const float* bias,
                  float* const* bn_partials, int act, int dtype, float* ws, int64_t ws_floats, hipStream_t s) {
    if (ws == nullptr || n < 1 || n > SK_MAX_GEMMS) return GS_EUNSUPPORTED;
    const GsConvGeom& g0 = *gg[0];
    SkArgs a;
    a.n = n; a.x = (const unsigned short*)x; a.y = (unsigned short*)y; a.bias = bias;
    a.x_bytes = (unsigned)((int64_t)g0.N * g0.IH * g0.IW * g0.in_pix_stride * 2);
    a.N = g0.N; a.Cin = g0.Cin; a.in_stride = g0.in_pix_stride; a.in_coff = g0.in_coff; a.Cout = g0.Cout;
    a.out_stride = g0.out_pix_stride; a.out_coff = g0.out_coff; a.act = act; a.ntn = cdiv(g0.Cout, 128); a.kch = g0.Cin / 64;
    const int M = g0.N * g0.OHg * g0.OWg;
    for (int i = 0; i < n; ++i) {
        const GsConvGeom& g = *gg[i];
        int nv = 0;
        if (!sk_geom_ok(g, &nv)) return GS_EUNSUPPORTED;
        if (g.N != g0.N || g.IH != g0.IH || g.IW != g0.IW || g.Cin != g0.Cin || g.in_pix_stride != g0.in_pix_stride ||
            g.in_coff != g0.in_coff || g.Cout != g0.Cout || g.out_pix_stride != g0.out_pix_stride || g.out_coff != g0.out_coff ||
            g.N * g.OHg * g.OWg != M)
            return GS_EUNSUPPORTED;
        SkGemm& k = a.g[i];
        k.w = (const unsigned short*)w[i]; k.bnp = bn_partials ? bn_partials[i] : nullptr;
        k.M = M; k.IH = g.IH; k.IW = g.IW; k.OHg = g.OHg; k.OWg = g.OWg; k.OH = g.OH; k.OW = g.OW;
        k.isy = g.isy; k.isx = g.isx; k.osy = g.osy; k.osx = g.osx; k.ooy = g.ooy; k.oox = g.oox;
        int q = 0, max_slot = 0;
        for (int t = 0; t < g.ntaps; ++t)
            if (sk_axis_ok(g.OHg, g.isy, g.tap_dy[t], g.IH) && sk_axis_ok(g.OWg, g.isx, g.tap_dx[t], g.IW)) {
                k.tap[q++] = (unsigned)(g.tap_dy[t] + 128) | ((unsigned)(g.tap_dx[t] + 128) << 8) | ((unsigned)g.tap_w[t] << 16);
                if (g.tap_w[t] > max_slot) max_slot = g.tap_w[t];
            }
        for (int t = q; t < GS_MAX_TAPS; ++t) k.tap[t] = k.tap[0];
        k.w_bytes = (unsigned)((int64_t)(max_slot + 1) * g.Cout * g.Cin * 2);
        k.nvis = nv;
        k.nq = nv * a.kch;
        k.mtiles = cdiv(M, 128);
    }
    // K parts: one block per CU (row tiles of 128: 128 KB of LDS) or two (<= 64 rows) over the whole launch, >= 4 chunks per part
    const int mt = M > 64 ? 4 : (M > 32 ? 2 : 1);
    static const int parts_max = getenv("GSSEG_SKINNY_PARTS") ? atoi(getenv("GSSEG_SKINNY_PARTS")) : 32;
    const int target = (mt == 4 ? 1 : 2) * gs_get_persistent_grid();
    int total = 0;
    int64_t used = 0;
    for (int i = 0; i < n; ++i) {
        SkGemm& k = a.g[i];
        int parts = target / (n * a.ntn * k.mtiles);
        if (parts > k.nq / 4) parts = k.nq / 4;
        if (parts > parts_max) parts = parts_max;
        if (parts < 1) parts = 1;
        k.qpp = cdiv(k.nq, parts);
        k.parts = cdiv(k.nq, k.qpp);
        k.blk0 = total;
        total += k.parts * a.ntn * k.mtiles;
        k.slab = ws + used;
        used += (int64_t)k.parts * k.M * a.Cout;
    }
    if (used > ws_floats) return GS_EUNSUPPORTED;
    for (int i = n; i < SK_MAX_GEMMS; ++i) { a.g[i] = a.g[0]; a.g[i].blk0 = total; }
#define SK_LAUNCH(DT)                                                                                  \
    do {                                                                                               \
        if (mt == 1) skinny_stream_kernel<DT, 1><<<total, 256, 0, s>>>(a);                             \
        else if (mt == 2) skinny_stream_kernel<DT, 2><<<total, 256, 0, s>>>(a);                        \
        else skinny_stream_kernel<DT, 4><<<total, 256, 0, s>>>(a);                                     \
        skinny_reduce_kernel<DT><<<dim3(cdiv(a.Cout, 64), cdiv(M, 16), n), 256, 0, s>>>(a);            \
    } while (0)
    if (dtype == GS_F16) SK_LAUNCH(GS_F16);
    else SK_LAUNCH(GS_BF16);
#undef SK_LAUNCH
    GS_CHECK_LAUNCH("gs_conv_igemm (weight-streaming form)");
    return GS_OK;
}
