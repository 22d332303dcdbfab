// Weight gradient of the kernel-2 / stride-2 transposed convolution of the U-Net `Up` block (unet/unet_parts.py:51,57:
// nn.ConvTranspose2d(in, in // 2, kernel_size=2, stride=2)) as a pointwise GEMM with K = pixels and LDS-DMA operands:
//
//   dW[ci][co][a][b] = sum over input pixels p = (n, y, x) of  X[p][ci] * dU[n, 2y + a + ooy, 2x + b + oox][co]
//
// The generic weight-gradient engine (igemm.hip: igemm_wgrad_kernel) ran these four launches at 0.13 of the MFMA peak with
// 2.4x their algorithmic HBM traffic (rocprofv3, profiles/r03_v6_*: FETCH 424 MB + WRITE 34 MB per launch): it gathers the dU
// tile once per sub-pixel class and per 64-channel ci tile.  Here a block owns (128 ci) x (64 co) x (all four classes) and
// walks K tiles of 64 input pixels: per tile ONE X slab [64 px][128 ci] and the four class slabs [64 px][64 co] of dU are
// written into LDS by `buffer_load_dwordx4 ... lds` into a three-deep ring while the previous tile is multiplied -- X is read
// Cout / 64 times and dU Cin / 128 times in total (once each at the full-resolution level: 402 MB, the byte bound there).
//   LDS image : six images of [64 px][128 B] per ring slot (X channels 0..63, 64..127; dU classes 0..3), 128-byte rows, the two
//               64-byte halves of row r swapped when (r >> 1) & 1 (source side of the DMA): conflict-free for
//               ds_read_b64_tr_b16, as in wgrad3x3_dma_kernel.
//   waves     : 8 = 4 (ci tiles of 32) x 2 (co halves of 32); each wave keeps 4 accumulator tiles (one per class), per K tile 4
//               steps of 16 pixels: one X fragment + four dU fragments (transposing reads) feed 4 MFMAs.
//   split K   : every (ci tile, co tile) pair is cut into `parts` ranges of K tiles; part k stores its [4][128][64] fp32 tile into
//               slab k of the workspace ([parts][4][Cin][Cout]); gs_wgrad_reduce_unpack sums the slabs in order, scales and
//               writes the reference layout [Cin][Cout][2][2] -- deterministic, no atomics (as gs_conv_wgrad_slabs).
// Covers power-of-two maps with IW >= 16, Cin % 128 == 0, Cout % 64 == 0 (every `Up` of the U-Net at the sizes it trains on);
// GS_EUNSUPPORTED otherwise (the caller runs the generic engine).
#include <stdlib.h>

#include "common.hpp"

namespace {

struct UwArgs {
    const unsigned short* x;      // [N][IH][IW][x_stride]
    const unsigned short* dy;     // [N][OH][OW][dy_stride]
    float* ws;                    // [parts][4][Cin][Cout]
    int N, IH, IW, Cin, x_stride, x_coff, Cout, OH, OW, dy_stride, dy_coff, ooy, oox;
    int iw_shift, tw_shift;       // log2(IW), log2(tile width)
    int ncit, ncot, parts, tpp, ntiles;      // ci tiles of 128, co tiles of 64, K parts, K tiles per part, K tiles in all
};

// one LDS-DMA piece (8 rows x 128 B): lane l's 16 bytes at buffer offset voff + soff land at dst + 16 l.  Inline assembly as in
// wgrad3x3.hip: with the builtin hipcc puts `s_waitcnt vmcnt(0)` in front of every transposing LDS read that follows a piece.
__device__ __forceinline__ void uw_dma_piece16(const __amdgpu_buffer_rsrc_t& rs, unsigned char* dst, unsigned voff, unsigned soff) {
    const unsigned lds_addr = (unsigned)(size_t)(LDS_AS void*)dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void uw_opaque(unsigned& x) { asm volatile("" : "+v"(x)); }

template <int DT>
__global__ __launch_bounds__(512, 1) void upwgrad_dma_kernel(const UwArgs a) {
    typedef typename Elem<DT>::V8 V8;
    constexpr int KT = 64;                         // input pixels per K tile
    constexpr int IMG_B = KT * 128;                // one image: 8 KB = 8 pieces
    constexpr int NIMG = 6, SLOT_B = NIMG * IMG_B; // 48 KB per ring slot
    constexpr int RING = 3;
    constexpr int PPW = NIMG * 8 / 8;              // pieces per wave per K tile: 48 / 8 = 6 (piece = wave + 8 k -> image k, rows 8 * wave ..)
    static_assert(RING * SLOT_B <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[RING * SLOT_B];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;       // ci tile of 32 (0..3), co half of 32
    int bid = blockIdx.x;
    const int cot = bid % a.ncot; bid /= a.ncot;
    const int cit = bid % a.ncit; bid /= a.ncit;
    const int part = bid;
    const int ci0 = cit * 128, co0 = cot * 64;
    const int t_begin = part * a.tpp;
    const int t_end = min(a.ntiles, t_begin + a.tpp);

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, (unsigned)a.N * (unsigned)a.IH * (unsigned)a.IW * (unsigned)a.x_stride * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t dy_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.dy, 0, (unsigned)a.N * (unsigned)a.OH * (unsigned)a.OW * (unsigned)a.dy_stride * 2u, 0x00020000);

    // ---- DMA side: piece (image k, rows 8 * wave .. 8 * wave + 7); lane l -> row l >> 3, physical slot l & 7 ----
    // tile-local pixel row r = 8 * wave + (lane >> 3): input pixel (ty, tx) = (r >> tw_shift, r & (tw - 1)) of the tile
    const int r_loc = 8 * wave + (lane >> 3);
    const int dls = (lane & 7) ^ (((lane >> 4) & 1) << 2);          // logical 16-byte slot: halves swapped when (row >> 1) & 1
    const int ty = r_loc >> a.tw_shift, tx = r_loc & ((1 << a.tw_shift) - 1);
    // per-lane offsets relative to the tile origin (input pixel (n, y0, x0)); the origin rides in the scalar offset
    const unsigned rel_x = (unsigned)(((ty * a.IW + tx) * a.x_stride + a.x_coff + ci0 + dls * 8) * 2);
    const unsigned rel_d = (unsigned)((((2 * ty + a.ooy) * a.OW + 2 * tx + a.oox) * a.dy_stride + a.dy_coff + co0 + dls * 8) * 2);
    const int tiles_x = a.IW >> a.tw_shift;                         // K tiles per image row band
    const int th = KT >> a.tw_shift;                                // rows of a tile
    const int tiles_per_img = (a.IH / th) * tiles_x;
    auto issue_tile = [&](int tile, unsigned slot, bool live) __attribute__((always_inline)) {
        const int n = tile / tiles_per_img, rr = tile - n * tiles_per_img;
        const int by = rr / tiles_x, bx = rr - by * tiles_x;
        const int y0 = by * th, x0 = bx << a.tw_shift;
        const unsigned so_x = (unsigned)((((n * a.IH + y0) * a.IW + x0) * a.x_stride) * 2);
        const unsigned so_d = (unsigned)((((n * a.OH + 2 * y0) * a.OW + 2 * x0) * a.dy_stride) * 2);
        const unsigned kill = live ? 0u : 0x80000000u;
        unsigned char* base = smem + slot * SLOT_B + wave * 1024;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            if (k < 2) {                                            // X channels [64 k, 64 k + 64) of the block's 128
                uw_dma_piece16(x_rsrc, base + k * IMG_B, (rel_x + (unsigned)k * 128u) | kill, so_x);
            } else {                                                // dU class (a, b) = ((k - 2) >> 1, (k - 2) & 1)
                const int cls = k - 2;
                const unsigned cls_off = (unsigned)((((cls >> 1) * a.OW + (cls & 1)) * a.dy_stride) * 2);
                uw_dma_piece16(dy_rsrc, base + k * IMG_B, (rel_d + cls_off) | kill, so_d);
            }
        }
    };

    // ---- MFMA side: transposed-read lane addressing (as wgrad3x3_dma_kernel) on the swizzled images ----
    const int G = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int krow = 8 * (G >> 1) + q;
    const int chn = 16 * (G & 1) + 4 * pp;
    // X: image (wm >> 1), 32-channel half (wm & 1); dU: image 2 + cls, 32-channel half wn
    unsigned a_off = (unsigned)((wm >> 1) * IMG_B + krow * 128 + ((((wm & 1) ^ ((q >> 1) & 1)) * 32 + chn) * 2));
    unsigned b_off = (unsigned)(2 * IMG_B + krow * 128 + (((wn ^ ((q >> 1) & 1)) * 32 + chn) * 2));
    uw_opaque(a_off); uw_opaque(b_off);

    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    const int ntile = t_end - t_begin;
    // prologue: two tiles in flight
    issue_tile(t_begin, 0u, ntile > 0);
    issue_tile(t_begin + 1, 1u, ntile > 1);
    const LDS_AS unsigned char* lds = (const LDS_AS unsigned char*)smem;
    for (int i = 0; i < ntile; ++i) {
        // tile i has landed when at most the 6 pieces of tile i+1 are outstanding (completion is counted in issue order)
        if (i + 1 < ntile) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // ... everybody's pieces; slot (i + 2) % 3 is free (tile i - 1 was read)
        asm volatile("" ::: "memory");
        const unsigned slot = (unsigned)(i % RING);
        issue_tile(t_begin + i + 2, (unsigned)((i + 2) % RING), i + 2 < ntile);
        const unsigned sb = slot * SLOT_B;
        V8 af[2], bf[2][4];
        auto load_step = [&](int ks, V8& fa, V8 (&fb)[4]) __attribute__((always_inline)) {
            const unsigned ra = sb + a_off + (unsigned)(ks * 16 * 128);
            fa = tr_read8<DT>((const LDS_AS unsigned short*)(lds + ra), (const LDS_AS unsigned short*)(lds + ra + 4 * 128));
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned rb = sb + b_off + (unsigned)(c * IMG_B + ks * 16 * 128);
                fb[c] = tr_read8<DT>((const LDS_AS unsigned short*)(lds + rb), (const LDS_AS unsigned short*)(lds + rb + 4 * 128));
            }
        };
        load_step(0, af[0], bf[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) load_step(ks + 1, af[(ks + 1) & 1], bf[(ks + 1) & 1]);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = Elem<DT>::mfma32(af[ks & 1], bf[ks & 1][c], acc[c]);
        }
    }
    // ---- store the part's tile: slab[part][cls][ci][co], lanes along co ----
    if (ntile <= 0) {                                               // (cannot happen: the host sizes parts so that every part has tiles)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    }
    const int l31 = lane & 31, h = lane >> 5;
    const int co = co0 + wn * 32 + l31;
    float* slab = a.ws + (int64_t)part * 4 * a.Cin * a.Cout;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            slab[((int64_t)c * a.Cin + ci) * a.Cout + co] = acc[c][r];
        }
}

static int ilog2_exact_u(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

struct UwPlan { bool ok; int tw_shift, iw_shift, ncit, ncot, ntiles, parts, tpp; };

static UwPlan uw_plan(int N, int IH, int IW, int Cin, int Cout) {
    UwPlan p{};
    p.iw_shift = ilog2_exact_u(IW);
    const int ihs = ilog2_exact_u(IH);
    static const int env = getenv("GSSEG_UPWGRAD_DMA") ? atoi(getenv("GSSEG_UPWGRAD_DMA")) : 1;
    p.ok = env != 0 && p.iw_shift >= 4 && ihs >= 0 && Cin % 128 == 0 && Cout % 64 == 0 && N > 0;
    if (!p.ok) return p;
    const int tw = IW < 32 ? IW : 32;
    p.tw_shift = ilog2_exact_u(tw);
    const int th = 64 / tw;
    if (IH % th != 0) { p.ok = false; return p; }
    p.ncit = Cin / 128; p.ncot = Cout / 64;
    p.ntiles = N * (IH / th) * (IW / tw);
    const int pairs = p.ncit * p.ncot;
    const int target = gs_get_persistent_grid();
    int parts = target / pairs;                                     // floor: one block per CU, no second round for a remainder
    if (parts > p.ntiles / 4) parts = p.ntiles / 4;                 // a part covers a few tiles: its slab costs 128 KB of traffic
    if (parts < 1) parts = 1;
    p.tpp = (p.ntiles + parts - 1) / parts;
    p.parts = (p.ntiles + p.tpp - 1) / p.tpp;
    return p;
}

}  // namespace

// K parts / workspace floats of gs_upconv2x2_wgrad_slabs for these dimensions; 0: the shape is not covered (use gs_conv_wgrad_slabs)
extern "C" int gs_upconv2x2_wgrad_parts(int N, int IH, int IW, int Cin, int Cout) {
    const UwPlan p = uw_plan(N, IH, IW, Cin, Cout);
    return p.ok ? p.parts : 0;
}
extern "C" int64_t gs_upconv2x2_wgrad_ws_floats(int N, int IH, int IW, int Cin, int Cout) {
    const UwPlan p = uw_plan(N, IH, IW, Cin, Cout);
    return p.ok ? (int64_t)p.parts * 4 * Cin * Cout : 0;
}

// x: the transposed convolution's input [N][IH][IW][x_pix_stride] (channels x_coff .. x_coff + Cin); dy: the gradient of its
// output inside [N][OH][OW][dy_pix_stride] (channels dy_coff .. dy_coff + Cout; output pixel (2y + a + ooy, 2x + b + oox));
// ws: gs_upconv2x2_wgrad_ws_floats() floats, receives gs_upconv2x2_wgrad_parts() slabs [4][Cin][Cout] (class = 2a + b), to be
// summed by gs_wgrad_reduce_unpack(ws, parts, grad, Cin, Cout, 4, 0, gscale) into the reference layout [Cin][Cout][2][2].
extern "C" int gs_upconv2x2_wgrad_slabs(const void* x, const void* dy, float* ws, int N, int IH, int IW, int Cin, int x_pix_stride,
                                        int x_coff, int Cout, int OH, int OW, int dy_pix_stride, int dy_coff, int ooy, int oox,
                                        int dtype, void* stream) {
    GS_CHECK_ARG(x && dy && ws && N > 0 && IH > 0 && IW > 0 && Cin > 0 && Cout > 0, "gs_upconv2x2_wgrad_slabs: bad arguments");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_upconv2x2_wgrad_slabs: bad dtype %d", dtype);
    GS_CHECK_ARG(ooy >= 0 && oox >= 0 && 2 * IH - 1 + ooy < OH && 2 * IW - 1 + oox < OW, "gs_upconv2x2_wgrad_slabs: the patch exceeds dy");
    GS_CHECK_ARG(x_pix_stride >= x_coff + Cin && dy_pix_stride >= dy_coff + Cout, "gs_upconv2x2_wgrad_slabs: bad strides");
    const UwPlan p = uw_plan(N, IH, IW, Cin, Cout);
    if (!p.ok || x_pix_stride % 8 || x_coff % 8 || dy_pix_stride % 8 || dy_coff % 8 ||
        (int64_t)N * IH * IW * x_pix_stride * 2 >= 2147483000LL || (int64_t)N * OH * OW * dy_pix_stride * 2 >= 2147483000LL)
        return GS_EUNSUPPORTED;
    UwArgs a;
    a.x = (const unsigned short*)x; a.dy = (const unsigned short*)dy; a.ws = ws;
    a.N = N; a.IH = IH; a.IW = IW; a.Cin = Cin; a.x_stride = x_pix_stride; a.x_coff = x_coff; a.Cout = Cout;
    a.OH = OH; a.OW = OW; a.dy_stride = dy_pix_stride; a.dy_coff = dy_coff; a.ooy = ooy; a.oox = oox;
    a.iw_shift = p.iw_shift; a.tw_shift = p.tw_shift;
    a.ncit = p.ncit; a.ncot = p.ncot; a.parts = p.parts; a.tpp = p.tpp; a.ntiles = p.ntiles;
    const int blocks = p.ncit * p.ncot * p.parts;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GS_F16) upwgrad_dma_kernel<GS_F16><<<blocks, 512, 0, s>>>(a);
    else upwgrad_dma_kernel<GS_BF16><<<blocks, 512, 0, s>>>(a);
    GS_CHECK_LAUNCH("gs_upconv2x2_wgrad_slabs");
    return GS_OK;
}
