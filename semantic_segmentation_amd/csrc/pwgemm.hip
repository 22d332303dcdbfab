// Pointwise MFMA GEMM with LDS-DMA operands for the kernel-2 / stride-2 transposed convolution of the U-Net `Up` block
// (unet/unet_parts.py:51,57: nn.ConvTranspose2d(in, in // 2, kernel_size=2, stride=2)).
//
//   forward : Y[p][(cls, co)] = bias[co] + sum_ci X[p][ci] * W[(cls, co)][ci]      M = N*IH*IW pixels, K = Cin, N' = 4*Cout
//             column (cls, co) of input pixel p = (n, iy, ix) is output pixel (n, 2*iy + (cls >> 1), 2*ix + (cls & 1)), channel co
//
// The generic gather engine (igemm.hip) runs these GEMMs at 180-470 TFLOP/s: its 128x128 tiles hold one or two K steps at the
// full-resolution end (K = 128), so every tile is prologue + epilogue.  Here one 8-wave block per CU walks (256 pixels x 256
// columns) items; every K stage (64 channels: a [256][64] slab of X and of W) is written into LDS by `buffer_load ... lds`
// while the previous stage is multiplied -- no staging registers, one barrier per stage, per-lane DMA offsets constant for
// the whole launch (the item and the stage ride in the scalar offset).
//   LDS image: 128-byte rows, 16-byte slot s of row r at s ^ ((r >> 1) & 7): the 16 rows of a ds_read_b128 lane group hit 16
//   distinct 16-byte bank groups; applied on the source side of the DMA (the destination is lane-linear).
//   waves    : 4 (pixels) x 2 (columns), 64 x 128 per wave = 2 x 4 tiles of mfma_f32_32x32x16, 32 MFMAs per stage.
//   epilogue : + bias, 16-bit pack, wave-local LDS transpose, one 128-byte store per (pixel, 64-channel chunk): each chunk of
//              64 columns belongs to one sub-pixel class (Cout % 64 == 0).
#include <stdlib.h>
#include <type_traits>

#include "common.hpp"

namespace {

struct PwArgs {
    const unsigned short* x;      // [M][in_stride]
    const unsigned short* w;      // [4*Cout][K]
    const float* bias;            // [Cout] or null
    unsigned short* y;            // [N][OH][OW][out_stride]
    int M, K, Cout, in_stride, in_coff;
    int iw_shift, ih_shift, IH, IW, OH, OW, out_stride, out_coff, ooy, oox;
    int ntn, nitems, xcd_order;
    int D, Dout, ooz, ncls;       // 3-D (ConvTranspose3d k2 s2): images are depth slices n' = nb * D + d, eight sub-voxel classes, the
                                  // output slice is nb * Dout + 2 d + cz + ooz; 2-D: D = Dout = 1, ooz = 0, ncls = 4
};

constexpr int PW_LDR = 72;        // staging row (64 + 8 elements): conflict-free transposes

__device__ __forceinline__ unsigned int pw_dpp_xor1(unsigned int v) {
    return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false);
}
__device__ __forceinline__ void pw_opaque(unsigned& x) { asm volatile("" : "+v"(x)); }
// one LDS-DMA piece (8 rows x 128 B): lane l's 16 bytes at voff + soff land at dst + 16 l
__device__ __forceinline__ void pw_dma_piece16(const __amdgpu_buffer_rsrc_t& rs, unsigned char* dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)dst, 16, voff, soff, 0, 0);
}

// DGRAD: the data gradient of the same layer as a GEMM over K = (sub-pixel class, co):
//   dX[p][ci] = sum_{cls, co} dY[n, 2*iy + (cls >> 1) + ooy, 2*ix + (cls & 1) + oox][co] * W[(cls, ci)][co]
// x = dY [N][OH][OW][in_stride] (gathered: per-lane pixel offsets per item, the class and the channel chunk in the scalar
// offset), w = the data-gradient pack [4][Cin][Cout] (K = Cout per class), y = dX [M][out_stride], a.Cout = Cin of the layer
// (the GEMM's column count), a.K = its Cout; K stages run over (class, 64-channel chunk).
template <int DT, bool DGRAD>
__global__ __launch_bounds__(512, 1) void upconv2x2_dma_kernel(const PwArgs a) {
    typedef typename Elem<DT>::V8 V8;
    constexpr int BM = 256, BN = 256, ROWB = 128;
    constexpr int A_B = BM * ROWB, B_B = BN * ROWB, BUF_B = A_B + B_B;      // [A0 | B0 | A1 | B1]
    constexpr int STG_EL = 32 * PW_LDR;
    constexpr unsigned VOOB = 0x80000000u;
    static_assert(8 * STG_EL * 2 <= BUF_B, "epilogue staging overlays the second stage buffer");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * BUF_B];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int kchunks = a.K >> 6;
    const int nstage = DGRAD ? 4 * kchunks : kchunks;              // even (host: forward K % 128 == 0)
    const unsigned nimg = (unsigned)(a.M >> (a.iw_shift + a.ih_shift));
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.x, 0, DGRAD ? nimg * (unsigned)a.OH * (unsigned)a.OW * (unsigned)a.in_stride * 2u : (unsigned)a.M * (unsigned)a.in_stride * 2u,
        0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (unsigned)a.ncls * (unsigned)a.Cout * (unsigned)a.K * 2u, 0x00020000);

    // ---- DMA side: wave w fills pieces w, w+8, w+16, w+24 of both slabs; lane l -> row l >> 3, physical slot l & 7 ----
    const int drow = lane >> 3;
    const int dls = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);  // logical slot: piece parity == wave parity
    unsigned va[4], vb[4], vb0[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (wave + 8 * j) * 8 + drow;
        va[j] = (unsigned)((r * a.in_stride + a.in_coff + dls * 8) * 2);       // forward: contiguous pixels (DGRAD: per item)
        vb0[j] = (unsigned)((r * a.K + dls * 8) * 2);
        vb[j] = vb0[j];
    }
    // DGRAD: the dY pixel of class (0,0) of every piece row, and the weight rows beyond Cin (a partial last column tile)
    auto setup_item = [&](int m0, int n0) __attribute__((always_inline)) {
        if (!DGRAD) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = (wave + 8 * j) * 8 + drow;
            const int p = m0 + r;
            const int ix = p & (a.IW - 1), iy = (p >> a.iw_shift) & (a.IH - 1), n = p >> (a.iw_shift + a.ih_shift);
            va[j] = (unsigned)((((n * a.OH + 2 * iy + a.ooy) * a.OW + 2 * ix + a.oox) * a.in_stride + a.in_coff + dls * 8) * 2);
            vb[j] = (n0 + r < a.Cout) ? vb0[j] : VOOB;
        }
    };
    // scalar offsets of stage s: forward (chunk s of the pixel / of the weight row); DGRAD (class s / kchunks, chunk s % kchunks)
    auto stage_soff = [&](int s, unsigned sa, unsigned sb, unsigned& oa, unsigned& ob) __attribute__((always_inline)) {
        if (!DGRAD) { oa = sa + (unsigned)s * ROWB; ob = sb + (unsigned)s * ROWB; return; }
        const int cls = s / kchunks, ch = s - cls * kchunks;
        oa = (unsigned)((((cls >> 1) * a.OW + (cls & 1)) * a.in_stride) * 2) + (unsigned)ch * ROWB;
        ob = sb + (unsigned)cls * (unsigned)a.Cout * (unsigned)a.K * 2u + (unsigned)ch * ROWB;
    };
    // piece k (compile-time): 0..3 the X slab, 4..7 the W slab
    auto issue_piece = [&](int k, unsigned soff_a, unsigned soff_b, unsigned bb, unsigned kill) __attribute__((always_inline)) {
        if (k < 4) pw_dma_piece16(x_rsrc, smem + bb * BUF_B + (unsigned)(wave + 8 * k) * 1024u, va[k < 4 ? k : 0] | kill, soff_a);
        else pw_dma_piece16(w_rsrc, smem + bb * BUF_B + A_B + (unsigned)(wave + 8 * (k - 4)) * 1024u, vb[k >= 4 ? k - 4 : 0] | kill, soff_b);
    };

    // ---- MFMA side ----
    unsigned abase[2][4], bbase[2][4];                              // [buffer][k step]; + i * 4096 / + j * 4096
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned sl = (unsigned)((((ks << 1) | h) ^ ((l31 >> 1) & 7)) << 4);
            abase[b][ks] = (unsigned)(b * BUF_B + (wm * 64 + l31) * ROWB) + sl;
            bbase[b][ks] = (unsigned)(b * BUF_B + A_B + (wn * 128 + l31) * ROWB) + sl;
            pw_opaque(abase[b][ks]); pw_opaque(bbase[b][ks]);
        }
    f32x16 acc[2][4];

    auto run_stage = [&](auto buf_tag, auto first_tag, unsigned soff_a, unsigned soff_b, unsigned kill) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        V8 af[2][2], bf[2][4];
        auto frag_load = [&](int ks, V8 (&fa)[2], V8 (&fb)[4]) __attribute__((always_inline)) {
            fa[0] = *reinterpret_cast<const V8*>(smem + abase[BUF][ks]);
            fb[0] = *reinterpret_cast<const V8*>(smem + bbase[BUF][ks]);
            fa[1] = *reinterpret_cast<const V8*>(smem + abase[BUF][ks] + 4096);
            fb[1] = *reinterpret_cast<const V8*>(smem + bbase[BUF][ks] + 4096);
            fb[2] = *reinterpret_cast<const V8*>(smem + bbase[BUF][ks] + 8192);
            fb[3] = *reinterpret_cast<const V8*>(smem + bbase[BUF][ks] + 12288);
        };
        frag_load(0, af[0], bf[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1;
            if (ks + 1 < 4) frag_load(ks + 1, af[cur ^ 1], bf[cur ^ 1]);
            issue_piece(2 * ks, soff_a, soff_b, 1 - BUF, kill);
            issue_piece(2 * ks + 1, soff_a, soff_b, 1 - BUF, kill);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (FIRST && ks == 0) {
                        f32x16 z;
#pragma unroll
                        for (int r = 0; r < 16; ++r) z[r] = 0.f;
                        acc[i][j] = Elem<DT>::mfma32(af[cur][i], bf[cur][j], z);
                    } else {
                        acc[i][j] = Elem<DT>::mfma32(af[cur][i], bf[cur][j], acc[i][j]);
                    }
                }
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // one LDS fragment read in its shadow
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);           // the two DMA pieces of this step
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto stage_sync = [&]() __attribute__((always_inline)) {        // this wave's pieces landed; barrier: everybody's did
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- epilogue ----
    unsigned short* stg = reinterpret_cast<unsigned short*>(smem + BUF_B) + wave * STG_EL;
    const bool odd = lane & 1;
    const unsigned int psel = odd ? 0x03020706u : 0x05040100u;
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.y, 0, DGRAD ? (unsigned)a.M * (unsigned)a.out_stride * 2u
                             : (nimg / (unsigned)a.D) * (unsigned)a.Dout * (unsigned)a.OH * (unsigned)a.OW * (unsigned)a.out_stride * 2u,
        0x00020000);
    auto epilogue = [&](int m0, int n0) __attribute__((always_inline)) {
        int e_m0 = m0, e_n0 = n0;
        asm volatile("" : "+s"(e_m0), "+s"(e_n0));
        // byte offset of the class-(0,0) output pixel of each of this lane's 8 store rows
        unsigned pixoff[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = e_m0 + wm * 64 + i * 32 + q * 8 + (lane >> 3);
                const int ix = p & (a.IW - 1), iy = (p >> a.iw_shift) & (a.IH - 1), n = p >> (a.iw_shift + a.ih_shift);
                const int nb = n / a.D, no = nb * a.Dout + 2 * (n - nb * a.D) + a.ooz;      // 2-D: no = n
                pixoff[i][q] = DGRAD ? (unsigned)(p * a.out_stride * 2)
                                     : (unsigned)((((no * a.OH + 2 * iy + a.ooy) * a.OW + 2 * ix + a.oox) * a.out_stride) * 2);
            }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int col0 = e_n0 + wn * 128 + jj * 64;            // 64 columns: of one sub-pixel class / of dX
            const int cls = DGRAD ? 0 : col0 / a.Cout, co0 = col0 - cls * a.Cout;
            const bool dead = DGRAD && col0 >= a.Cout;             // partial last column tile
            // class bits (cz, cy, cx): one output slice / row / column further
            const unsigned clsoff = dead ? VOOB : (unsigned)(((((cls >> 2) * a.OH + ((cls >> 1) & 1)) * a.OW + (cls & 1)) * (DGRAD ? 0 : a.out_stride) + a.out_coff + co0 + (lane & 7) * 8) * 2);
            float bv[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = (!DGRAD && a.bias != nullptr) ? a.bias[co0 + j * 32 + l31] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int r0 = 2 * m;
                    const int rowa = (r0 & 3) + 8 * (r0 >> 2) + 4 * h;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float v0 = acc[i][2 * jj + j][r0] + bv[j], v1 = acc[i][2 * jj + j][r0 + 1] + bv[j];
                        const unsigned int own = Elem<DT>::pack2(v0, v1);
                        const unsigned int oth = pw_dpp_xor1(own);
                        const unsigned int pk = __builtin_amdgcn_perm(oth, own, psel);
                        const int row = rowa + (odd ? 1 : 0);
                        *reinterpret_cast<unsigned int*>(stg + row * PW_LDR + j * 32 + (l31 & ~1)) = pk;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                uint4 sv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    sv[q] = *reinterpret_cast<const uint4*>(stg + (q * 8 + (lane >> 3)) * PW_LDR + (lane & 7) * 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    u32x4 d;
                    d[0] = sv[q].x; d[1] = sv[q].y; d[2] = sv[q].z; d[3] = sv[q].w;
                    __builtin_amdgcn_raw_buffer_store_b128(d, y_rsrc, dead ? VOOB : pixoff[i][q] + clsoff, 0, GS_OUT_AUX);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    };

    // ---- items: column tile fastest (the tiles of one pixel range share its X slab in L2) ----
    int it = a.xcd_order ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    if (it >= a.nitems) return;
    auto item_soff = [&](int item, unsigned& sa, unsigned& sb, int& m0, int& n0) __attribute__((always_inline)) {
        const int nt = item % a.ntn, mt = item / a.ntn;
        m0 = mt * BM; n0 = nt * BN;
        sa = (unsigned)m0 * (unsigned)a.in_stride * 2u;             // (forward only: DGRAD carries the pixel in the lane offsets)
        sb = (unsigned)n0 * (unsigned)a.K * 2u;
    };
    unsigned sa, sb;
    int m0, n0;
    item_soff(it, sa, sb, m0, n0);
    setup_item(m0, n0);
    {
        unsigned oa, ob;
        stage_soff(0, sa, sb, oa, ob);
#pragma unroll
        for (int k = 0; k < 8; ++k) issue_piece(k, oa, ob, 0u, 0u); // stage 0 -> buffer 0
    }
    for (;;) {
        const int nit = it + gridDim.x;
        const bool more_items = nit < a.nitems;
        unsigned sa_n = sa, sb_n = sb;
        int m0_n = m0, n0_n = n0;
        if (more_items) item_soff(nit, sa_n, sb_n, m0_n, n0_n);
        for (int sp = 0; sp < nstage; sp += 2) {
            const bool last = sp + 2 >= nstage;
            unsigned oa, ob;
            stage_sync();
            stage_soff(sp + 1, sa, sb, oa, ob);
            if (sp == 0) run_stage(std::integral_constant<int, 0>{}, std::true_type{}, oa, ob, 0u);
            else run_stage(std::integral_constant<int, 0>{}, std::false_type{}, oa, ob, 0u);
            stage_sync();
            if (last) {
                if (more_items) setup_item(m0_n, n0_n);            // from here on the pieces belong to the next item
                stage_soff(0, sa_n, sb_n, oa, ob);
            } else {
                stage_soff(sp + 2, sa, sb, oa, ob);
            }
            run_stage(std::integral_constant<int, 1>{}, std::false_type{}, oa, ob, (last && !more_items) ? VOOB : 0u);
        }
        __builtin_amdgcn_s_barrier();              // every wave has left the second buffer: staging may overlay it
        asm volatile("" ::: "memory");
        epilogue(m0, n0);
        if (!more_items) break;
        it = nit; sa = sa_n; sb = sb_n; m0 = m0_n; n0 = n0_n;
    }
}

int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1)) != 0) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

}  // namespace

// Fast path of gs_upconv2x2_fwd (igemm.hip): returns 1 when the launch was taken, 0 when the shape is not covered (the caller
// then uses the generic engine), a negative GS_E* code on a launch error.  GSSEG_UP_DMA=0 switches it off.
int pw_upconv2x2_fwd_fast(const void* x, const void* w, const float* bias, void* y, int N, int IH, int IW, int Cin,
                          int in_pix_stride, int in_coff, int Cout, int OH, int OW, int out_pix_stride, int out_coff, int ooy,
                          int oox, int act, int dtype, void* stream, int D, int Dout, int ooz) {
    static const int env = getenv("GSSEG_UP_DMA") ? atoi(getenv("GSSEG_UP_DMA")) : 1;
    const bool is3d = D > 1 || Dout > 1;
    const int ncls = is3d ? 8 : 4;
    const int64_t M = (int64_t)N * D * IH * IW;
    const int iws = ilog2_exact(IW), ihs = ilog2_exact(IH);
    if (env == 0 || act != GS_ACT_NONE || Cin % 128 != 0 || Cout % 64 != 0 || M % 256 != 0 || iws < 0 || ihs < 0 ||
        in_pix_stride % 8 != 0 || in_coff % 8 != 0 || out_pix_stride % 8 != 0 || out_coff % 8 != 0 ||
        M * in_pix_stride * 2 >= 2147483000LL || (int64_t)N * Dout * OH * OW * out_pix_stride * 2 >= 2147483000LL ||
        (int64_t)ncls * Cout * Cin * 2 >= 2147483000LL || ooz < 0)
        return 0;
    PwArgs a;
    a.x = (const unsigned short*)x; a.w = (const unsigned short*)w; a.bias = bias; a.y = (unsigned short*)y;
    a.M = (int)M; a.K = Cin; a.Cout = Cout; a.in_stride = in_pix_stride; a.in_coff = in_coff;
    a.iw_shift = iws; a.ih_shift = ihs; a.IH = IH; a.IW = IW; a.OH = OH; a.OW = OW;
    a.out_stride = out_pix_stride; a.out_coff = out_coff; a.ooy = ooy; a.oox = oox;
    a.D = D; a.Dout = Dout; a.ooz = ooz; a.ncls = ncls;
    a.ntn = ncls * Cout / 256;
    a.nitems = (int)(M / 256) * a.ntn;
    const int pg = gs_get_persistent_grid();              // 256, or fewer when CUs are left to RCCL
    const int blocks = a.nitems < pg ? a.nitems : pg;
    a.xcd_order = (blocks % 8 == 0 && a.ntn > 1) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GS_F16) upconv2x2_dma_kernel<GS_F16, false><<<blocks, 512, 0, s>>>(a);
    else upconv2x2_dma_kernel<GS_BF16, false><<<blocks, 512, 0, s>>>(a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        gs_set_error("gs_upconv2x2_fwd: launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 1;
}

// Data gradient of ConvTranspose2d(kernel 2, stride 2) (unet/unet_parts.py:51,57): dx[n, iy, ix, ci] = sum over the four
// sub-pixel classes and co of dy[n, 2*iy + a + ooy, 2*ix + b + oox, co] * w[ci][co][a][b]; wd = the data-gradient pack
// [4][Cin][Cout] of gs_pack_weight(transposed).  Covers the shapes of the LDS-DMA GEMM (power-of-two IH / IW, N*IH*IW % 256 == 0,
// Cin % 128 == 0, Cout % 64 == 0); GS_EUNSUPPORTED otherwise (the caller then runs gs_conv_igemm on the 4-tap stride-2 geometry).
extern "C" int gs_upconv2x2_dgrad(const void* dy, const void* wd, void* dx, int N, int IH, int IW, int Cin, int Cout, int OH,
                                  int OW, int dy_pix_stride, int dy_coff, int ooy, int oox, int dx_pix_stride, int dx_coff,
                                  int dtype, void* stream) {
    GS_CHECK_ARG(dy && wd && dx && N > 0 && IH > 0 && IW > 0 && Cin > 0 && Cout > 0, "gs_upconv2x2_dgrad: bad arguments");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_upconv2x2_dgrad: bad dtype %d", dtype);
    GS_CHECK_ARG(ooy >= 0 && oox >= 0 && 2 * IH - 1 + ooy < OH && 2 * IW - 1 + oox < OW, "gs_upconv2x2_dgrad: the patch exceeds dy");
    GS_CHECK_ARG(dy_pix_stride >= dy_coff + Cout && dx_pix_stride >= dx_coff + Cin, "gs_upconv2x2_dgrad: bad strides");
    static const int env = getenv("GSSEG_UP_DMA") ? atoi(getenv("GSSEG_UP_DMA")) : 1;
    const int64_t M = (int64_t)N * IH * IW;
    const int iws = ilog2_exact(IW), ihs = ilog2_exact(IH);
    if (env == 0 || Cin % 128 != 0 || Cout % 64 != 0 || M % 256 != 0 || iws < 0 || ihs < 0 || dy_pix_stride % 8 != 0 ||
        dy_coff % 8 != 0 || dx_pix_stride % 8 != 0 || dx_coff % 8 != 0 || M * dx_pix_stride * 2 >= 2147483000LL ||
        (int64_t)N * OH * OW * dy_pix_stride * 2 >= 2147483000LL || (int64_t)4 * Cout * Cin * 2 >= 2147483000LL)
        return GS_EUNSUPPORTED;
    PwArgs a;
    a.x = (const unsigned short*)dy; a.w = (const unsigned short*)wd; a.bias = nullptr; a.y = (unsigned short*)dx;
    a.M = (int)M; a.K = Cout; a.Cout = Cin; a.in_stride = dy_pix_stride; a.in_coff = dy_coff;
    a.iw_shift = iws; a.ih_shift = ihs; a.IH = IH; a.IW = IW; a.OH = OH; a.OW = OW;
    a.out_stride = dx_pix_stride; a.out_coff = dx_coff; a.ooy = ooy; a.oox = oox;
    a.D = 1; a.Dout = 1; a.ooz = 0; a.ncls = 4;
    a.ntn = cdiv(Cin, 256);
    a.nitems = (int)(M / 256) * a.ntn;
    const int pg = gs_get_persistent_grid();              // 256, or fewer when CUs are left to RCCL
    const int blocks = a.nitems < pg ? a.nitems : pg;
    a.xcd_order = (blocks % 8 == 0 && a.ntn > 1) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GS_F16) upconv2x2_dma_kernel<GS_F16, true><<<blocks, 512, 0, s>>>(a);
    else upconv2x2_dma_kernel<GS_BF16, true><<<blocks, 512, 0, s>>>(a);
    GS_CHECK_LAUNCH("gs_upconv2x2_dgrad");
    return GS_OK;
}
