// 3x3 / stride-1 / pad-1 convolution (and its data gradient) as an MFMA implicit GEMM with HALO REUSE.
//
// The generic engine (igemm.hip) gathers the A operand once per tap, i.e. every input pixel travels
// L2 -> LDS nine times; at the 256x256 / 64-channel end of the U-Net that makes the kernel L2-bandwidth
// bound.  Here a block owns a spatial patch TH x TW of one image: the (TH+2) x (TW+2) x 64-channel input
// halo is staged in LDS ONCE per channel chunk and all nine taps read it at shifted row offsets, so the
// activation traffic drops ~7x and only the (L2-resident) weight tile is re-staged per tap.
//
//   block  : 256 threads = 4 waves; each wave owns 64 pixels x 64 couts (2x2 tiles of mfma_f32_32x32x16)
//   BN=128 : waves 2(M) x 2(N), patch 128 pixels (4x32 or 8x16)
//   BN=64  : waves 4(M) x 1(N), patch 256 pixels (8x32 or 16x16)   -- the Cout = 64 layers
//   LDS    : halo [(TH+2)(TW+2)][72] + 2 x weights [BN][72]   (144-B padded rows: conflict-free b128 reads)
//   K loop : for each 64-channel chunk { halo resident; 9 taps x { weights double-buffered, 16 MFMAs/wave } }
//            the next chunk's halo is fetched into registers while the 9 taps run (issue early, write late).
//   epilogue as igemm.hip: bias/act, 16-bit NHWC store (strided), per-patch BatchNorm partial sums.
#include <stdlib.h>
#include <atomic>
#include <type_traits>

#include "common.hpp"
#include "conv3x3_args.hpp"

// K loop of the big kernel: one LDS fragment read in the shadow of every MFMA (sched_group_barrier pattern) instead of the
// four reads + prefetch load as a block in front of four MFMAs: +1.3 % over the 13 layer shapes (5.23 -> 5.16 ms), same
// arithmetic per accumulator.  -DGS_C3_NO_INTERLEAVE restores the blocked form.
#ifndef GS_C3_NO_INTERLEAVE
#define GS_C3_INTERLEAVE 1
#endif

namespace {

__device__ __forceinline__ int xcd_remap3(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}


template <int DT, int BN, int TW>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const C3Args a) {
    typedef typename Elem<DT>::V8 V8;
    constexpr int WM = (BN == 128) ? 2 : 4;       // waves along M
    constexpr int BM = WM * 64;                   // pixels per patch
    constexpr int TH = BM / TW;
    constexpr int HWD = TW + 2, HHT = TH + 2, HP = HWD * HHT;
    constexpr int HALO_EL = HP * C3_LDR, B_EL = BN * C3_LDR;
    constexpr int HCH = (HP * 8 + 255) / 256;     // 16-byte halo chunks per thread
    constexpr int BROWS = BN / 32;
    constexpr int TWS = (TW == 32) ? 5 : 4;
    __shared__ __attribute__((aligned(16))) unsigned short smem[HALO_EL + 2 * B_EL];
    unsigned short* halo = smem;
    unsigned short* Bs = smem + HALO_EL;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (BN == 128) ? (wave >> 1) : wave;
    const int wn = (BN == 128) ? (wave & 1) : 0;
    const int l31 = lane & 31, h = lane >> 5;
    const int lid = xcd_remap3(blockIdx.x, a.nblocks);
    const int ntile = lid % a.ntn;
    int patch = lid / a.ntn;
    const int tx = patch % a.tiles_x; patch /= a.tiles_x;
    const int ty = patch % a.tiles_y;
    const int n = patch / a.tiles_y;
    const int y0 = ty * TH, x0 = tx * TW, n0 = ntile * BN;
    const int mtile = lid / a.ntn;

    // ---- staging assignments ----
    int h_off[HCH];           // element offset into the image of the halo pixel (or -1), channel part added later
    int h_lds[HCH];           // LDS element offset
#pragma unroll
    for (int j = 0; j < HCH; ++j) {
        const int id = t + 256 * j;
        const int hp = id >> 3, c = id & 7;
        h_lds[j] = -1; h_off[j] = -1;
        if (hp < HP) {
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const int gy = y0 + hy - 1, gx = x0 + hx - 1;
            h_lds[j] = hp * C3_LDR + c * 8;
            if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) h_off[j] = (n * a.H + gy) * a.W + gx;
        }
    }
    const int chunk = t & 7, rbase = t >> 3;
    uint4 rh[HCH], rb[BROWS];
    const int nchunks = (a.Cin + 63) >> 6;

    auto load_halo = [&](int cc) {
        const int ci = cc * 64 + (t & 7) * 8;
#pragma unroll
        for (int j = 0; j < HCH; ++j) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (h_off[j] >= 0 && ci < a.Cin)
                v = *reinterpret_cast<const uint4*>(a.x + (int64_t)h_off[j] * a.in_stride + a.in_coff + ci);
            rh[j] = v;
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int j = 0; j < HCH; ++j)
            if (h_lds[j] >= 0) *reinterpret_cast<uint4*>(halo + h_lds[j]) = rh[j];
    };
    auto load_b = [&](int cc, int tap) {
        const int ci = cc * 64 + chunk * 8;
#pragma unroll
        for (int j = 0; j < BROWS; ++j) {
            const int co = n0 + rbase + 32 * j;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ci < a.Cin && co < a.Cout)
                v = *reinterpret_cast<const uint4*>(a.w + ((int64_t)tap * a.Cout + co) * a.Cin + ci);
            rb[j] = v;
        }
    };
    auto store_b = [&](int buf) {
        unsigned short* B = Bs + buf * B_EL;
#pragma unroll
        for (int j = 0; j < BROWS; ++j)
            *reinterpret_cast<uint4*>(B + (rbase + 32 * j) * C3_LDR + chunk * 8) = rb[j];
    };

    // A fragment base offsets (centre tap) of the wave's two 32-pixel M tiles
    int a_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = (wm * 2 + i) * 32 + l31;
        const int py = p >> TWS, px = p & (TW - 1);
        a_off[i] = ((py + 1) * HWD + px + 1) * C3_LDR + h * 8;
    }
    const int b_off = (wn * 64 + l31) * C3_LDR + h * 8;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_halo(0);
    load_b(0, 0);
    store_halo();
    store_b(0);
    __syncthreads();
    int ks = 0;
    const int dbg = a.act >> 8;
    for (int cc = 0; cc < ((dbg & 2) ? 0 : nchunks); ++cc) {
        const bool more = cc + 1 < nchunks;
        if (more) load_halo(cc + 1);
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap, ++ks) {
            const bool last = (tap == 8) && !more;
            if (!last) {
                if (tap < 8) load_b(cc, tap + 1);
                else load_b(cc + 1, 0);
            }
            const int toff = (a.tap_dy[tap] * HWD + a.tap_dx[tap]) * C3_LDR;
            const unsigned short* B = Bs + (ks & 1) * B_EL + b_off;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                V8 af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const V8*>(halo + a_off[i] + toff + kk * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const V8*>(B + j * 32 * C3_LDR + kk * 16);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = Elem<DT>::mfma32(af[i], bf[j], acc[i][j]);
            }
            if (!last) store_b((ks + 1) & 1);
            if (tap == 8 && more) {
                __syncthreads();          // every wave is done with this chunk's halo
                store_halo();
            }
            __syncthreads();
        }
    }

    // ---- epilogue ----
    int* rowpix = reinterpret_cast<int*>(smem);                 // BM ints
    float* red = reinterpret_cast<float*>(smem) + BM;           // [WM][2][BN] floats
    if (t < BM) {
        const int py = t >> TWS, px = t & (TW - 1);
        const int gy = y0 + py, gx = x0 + px;
        rowpix[t] = (gy < a.H && gx < a.W) ? (n * a.H + gy) * a.W + gx : -1;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int cl = wn * 64 + j * 32 + l31;
        const int co = n0 + cl;
        const bool cok = co < a.Cout;
        const float bv = (a.bias != nullptr && cok) ? a.bias[co] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int p = rowpix[row];
                const float v = p >= 0 ? acc[i][j][r] : 0.f;     // pixels outside the image carry no statistics
                s1 += v;
                s2 += v * v;
                if (cok && p >= 0 && !(dbg & 1))
                    a.y[(int64_t)p * a.out_stride + a.out_coff + co] = Elem<DT>::from_f(act_fwd(v + bv, a.act & 0xff));
            }
        }
        if (a.bnp != nullptr) {
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (h == 0) {
                red[(wm * 2 + 0) * BN + cl] = s1;
                red[(wm * 2 + 1) * BN + cl] = s2;
            }
        }
    }
    if (a.bnp != nullptr) {
        __syncthreads();
        if (t < BN && n0 + t < a.Cout) {
            float v1 = 0.f, v2 = 0.f;
#pragma unroll
            for (int m = 0; m < WM; ++m) { v1 += red[(m * 2 + 0) * BN + t]; v2 += red[(m * 2 + 1) * BN + t]; }
            float* dst = a.bnp + (int64_t)mtile * 2 * a.Cout + n0 + t;
            dst[0] = v1;
            dst[a.Cout] = v2;
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// v2: PERSISTENT blocks.  A block walks work items (patch, N-tile) it, it+grid, ... ; the halo of the next
// (item, chunk) is fetched into registers while the current nine taps run, so HBM reads, MFMA work and the
// output stores of consecutive patches overlap inside one block.  Epilogue: the wave transposes its 64x64
// tile through a small LDS staging area (packed pairs via DPP + v_perm, ds_write_b32 / ds_read_b128) and
// writes whole 128-byte channel rows with 16-byte stores instead of 64 two-byte stores per lane.
// ---------------------------------------------------------------------------------------------------

template <int DT, int BN, int TW>
__global__ __launch_bounds__(256, 2) void conv3x3_persist_kernel(const C3Args a) {
    typedef typename Elem<DT>::V8 V8;
    constexpr int WM = (BN == 128) ? 2 : 4;
    constexpr int BM = WM * 64;
    constexpr int TH = BM / TW;
    constexpr int HWD = TW + 2, HHT = TH + 2, HP = HWD * HHT;
    constexpr int HALO_EL = HP * C3_LDR, B_EL = BN * C3_LDR;
    constexpr int HCH = (HP * 8 + 255) / 256;
    constexpr int BROWS = BN / 32;
    constexpr int TWS = (TW == 32) ? 5 : 4;
    constexpr int STG_EL = 32 * C3_LDR;            // per-wave staging: 32 rows x (64+8) elements
    constexpr unsigned OOB = 0xFFFFFFFFu;          // buffer loads beyond num_records return 0
    static_assert(4 * STG_EL + WM * 2 * BN * 2 <= HALO_EL, "epilogue staging must fit in the halo region");
    __shared__ __attribute__((aligned(16))) unsigned short smem[HALO_EL + 2 * B_EL];
    unsigned short* halo = smem;
    unsigned short* Bs = smem + HALO_EL;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (BN == 128) ? (wave >> 1) : wave;
    const int wn = (BN == 128) ? (wave & 1) : 0;
    const int l31 = lane & 31, h = lane >> 5;
    const int nitems = a.nblocks;
    const int dbg = a.act >> 8;
    const int act = a.act & 0xff;
    const int chunk = t & 7, rbase = t >> 3;
    const int nchunks = (a.Cin + 63) >> 6;
    const unsigned img_bytes = (unsigned)a.H * a.W * a.in_stride * 2u;   // host guarantees < 4 GiB

    struct Item { int n, y0, x0, n0, mtile; };
    auto decode = [&](int it) __attribute__((always_inline)) {
        Item r;
        const int ntile = it % a.ntn;
        int patch = it / a.ntn;
        r.mtile = patch;
        const int tx = patch % a.tiles_x; patch /= a.tiles_x;
        const int ty = patch % a.tiles_y;
        r.n = patch / a.tiles_y;
        r.y0 = ty * TH; r.x0 = tx * TW; r.n0 = ntile * BN;
        return r;
    };
    auto x_rsrc = [&](int n) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (int64_t)n * a.H * a.W * a.in_stride), 0, img_bytes, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (unsigned)(9u * a.ndz * a.Cout * a.Cin * 2u), 0x00020000);

    // ---- halo staging: byte offsets inside the image (OOB when the halo pixel is outside) ----
    unsigned h_off[HCH];
    const int h_lds0 = (t >> 3) * C3_LDR + chunk * 8;     // chunk j lives 32 halo pixels further
    auto setup_halo = [&](const Item& itn) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < HCH; ++j) {
            const int hp = (t + 256 * j) >> 3;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const int gy = itn.y0 + hy - 1, gx = itn.x0 + hx - 1;
            const bool ok = hp < HP && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            h_off[j] = ok ? (unsigned)(((gy * a.W + gx) * a.in_stride + a.in_coff + chunk * 8) * 2) : OOB;
        }
    };
    uint4 rh[HCH];
    auto load_halo = [&](const __amdgpu_buffer_rsrc_t& rs, int cc) __attribute__((always_inline)) {
        const bool cok = cc * 64 + chunk * 8 < a.Cin;
#pragma unroll
        for (int j = 0; j < HCH; ++j) {
            const unsigned off = (cok && h_off[j] != OOB) ? h_off[j] + (unsigned)cc * 128u : OOB;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
            rh[j] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    };
    auto store_halo = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < HCH; ++j)
            if ((t >> 3) + 32 * j < HP) *reinterpret_cast<uint4*>(halo + h_lds0 + j * 32 * C3_LDR) = rh[j];
    };

    // ---- weight tiles: cursor runs two K-steps ahead of the compute cursor ----
    uint4 rb3[1][BROWS];
    int bl_it = blockIdx.x, bl_cc = 0, bl_tap = 0, bl_n0 = 0;
    bool bl_valid = true;
    auto bl_advance = [&]() __attribute__((always_inline)) {
        if (++bl_tap == 9) {
            bl_tap = 0;
            if (++bl_cc == nchunks) {
                bl_cc = 0;
                bl_it += gridDim.x;
                bl_valid = bl_it < nitems;
                bl_n0 = bl_valid ? (bl_it % a.ntn) * BN : 0;
            }
        }
    };
    auto load_b2 = [&](uint4 (&dst)[BROWS]) {
        const int ci = bl_cc * 64 + chunk * 8;
#pragma unroll
        for (int j = 0; j < BROWS; ++j) {
            const int co = bl_n0 + rbase + 32 * j;
            const bool ok = bl_valid && ci < a.Cin && co < a.Cout;
            const unsigned off = ok ? (unsigned)((((bl_tap * a.Cout) + co) * a.Cin + ci) * 2) : OOB;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, off, 0, 0);
            dst[j] = make_uint4(v[0], v[1], v[2], v[3]);
        }
        bl_advance();
    };
    auto store_b2 = [&](const uint4 (&src)[BROWS], int buf) {
        unsigned short* B = Bs + buf * B_EL;
#pragma unroll
        for (int j = 0; j < BROWS; ++j)
            *reinterpret_cast<uint4*>(B + (rbase + 32 * j) * C3_LDR + chunk * 8) = src[j];
    };

    int a_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = (wm * 2 + i) * 32 + l31;
        a_off[i] = (((p >> TWS) + 1) * HWD + (p & (TW - 1)) + 1) * C3_LDR + h * 8;
    }
    const int b_off = (wn * 64 + l31) * C3_LDR + h * 8;

    f32x16 acc[2][2];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    auto mma_tap = [&](int tap, int buf) __attribute__((always_inline)) {
        const int toff = (a.tap_dy[tap] * HWD + a.tap_dx[tap]) * C3_LDR;
        const unsigned short* B = Bs + buf * B_EL + b_off;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            V8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const V8*>(halo + a_off[i] + toff + kk * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const V8*>(B + j * 32 * C3_LDR + kk * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = Elem<DT>::mfma32(af[i], bf[j], acc[i][j]);
        }
    };

    // ---- epilogue of one finished item (called between two block barriers) ----
    unsigned short* stg = halo + wave * STG_EL;
    float* red = reinterpret_cast<float*>(halo + 4 * STG_EL);          // [WM][2][BN]
    const bool odd = lane & 1;
    const unsigned int psel = odd ? 0x03020706u : 0x05040100u;
    const float neg_slope = act == GS_ACT_RELU ? 0.f : (act == GS_ACT_LEAKY02 ? 0.2f : 1.f);
    auto epilogue = [&](const Item& itc) __attribute__((always_inline)) {
        float bv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = itc.n0 + wn * 64 + j * 32 + l31;
            bv[j] = (a.bias != nullptr && co < a.Cout) ? a.bias[co] : 0.f;
        }
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
        const bool want_stats = a.bnp != nullptr;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int prow0 = (wm * 2 + i) * 32;       // first patch pixel of this M tile
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int r0 = 2 * m;
                const int rowa = (r0 & 3) + 8 * (r0 >> 2) + 4 * h;      // row of reg r0; reg r0+1 is rowa+1
                // BatchNorm statistics weight of the two rows: 1 inside the image, 0 outside (edge patches);
                // arithmetic instead of compares so no lane masks are kept live
                float w0 = 1.f, w1 = 1.f;
                if (want_stats) {
                    const int p0 = prow0 + rowa;
                    const int gy0 = itc.y0 + (p0 >> TWS), gx0 = itc.x0 + (p0 & (TW - 1));
                    const int gy1 = itc.y0 + ((p0 + 1) >> TWS), gx1 = itc.x0 + ((p0 + 1) & (TW - 1));
                    w0 = (float)((unsigned)((gy0 - a.H) & (gx0 - a.W)) >> 31);
                    w1 = (float)((unsigned)((gy1 - a.H) & (gx1 - a.W)) >> 31);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v0 = acc[i][j][r0], v1 = acc[i][j][r0 + 1];
                    if (want_stats) {
                        const float u0 = v0 * w0, u1 = v1 * w1;
                        s1[j] += u0 + u1;
                        s2[j] += u0 * u0 + u1 * u1;
                    }
                    v0 += bv[j];
                    v1 += bv[j];
                    v0 = v0 > 0.f ? v0 : v0 * neg_slope;      // NONE: slope 1, RELU: 0, LEAKY: 0.2 (branch free)
                    v1 = v1 > 0.f ? v1 : v1 * neg_slope;
                    const unsigned int own = Elem<DT>::pack2(v0, v1);
                    const unsigned int oth = dpp_xor1(own);
                    const unsigned int pk = __builtin_amdgcn_perm(oth, own, psel);
                    const int row = rowa + (odd ? 1 : 0);
                    *reinterpret_cast<unsigned int*>(stg + row * C3_LDR + j * 32 + (l31 & ~1)) = pk;
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rrow = q * 8 + (lane >> 3), ch = lane & 7;
                const uint4 v = *reinterpret_cast<const uint4*>(stg + rrow * C3_LDR + ch * 8);
                const int p = prow0 + rrow;
                const int gy = itc.y0 + (p >> TWS), gx = itc.x0 + (p & (TW - 1));
                const int co = itc.n0 + wn * 64 + ch * 8;
                // host guarantees Cout % 8 == 0 and 16-byte aligned output rows: whole 16-byte stores only
                if (gy < a.H && gx < a.W && co < a.Cout && !(dbg & 1))
                    *reinterpret_cast<uint4*>(a.y + (int64_t)((itc.n * a.H + gy) * a.W + gx) * a.out_stride +
                                              a.out_coff + co) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (a.bnp != nullptr) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                s1[j] += __shfl_xor(s1[j], 32, 64);
                s2[j] += __shfl_xor(s2[j], 32, 64);
                if (h == 0) {
                    const int cl = wn * 64 + j * 32 + l31;
                    red[(wm * 2 + 0) * BN + cl] = s1[j];
                    red[(wm * 2 + 1) * BN + cl] = s2[j];
                }
            }
        }
    };
    auto finish_stats = [&](const Item& itc) __attribute__((always_inline)) {       // after a block barrier
        if (a.bnp != nullptr && t < BN && itc.n0 + t < a.Cout) {
            float v1 = 0.f, v2 = 0.f;
#pragma unroll
            for (int m = 0; m < WM; ++m) { v1 += red[(m * 2 + 0) * BN + t]; v2 += red[(m * 2 + 1) * BN + t]; }
            float* dst = a.bnp + (int64_t)itc.mtile * 2 * a.Cout + itc.n0 + t;
            dst[0] = v1;
            dst[a.Cout] = v2;
        }
    };

    int it = blockIdx.x;
    if (it >= nitems) return;
    Item cur = decode(it);
    bl_n0 = cur.n0;

    // Weights are prefetched ONE K-step ahead (registers -> the other LDS buffer), the halo one chunk ahead.
    // (A two-step-ahead variant with rotating register sets kept counted vmcnt waits in the loop but did not
    // pay: the two co-resident blocks run their load/LDS/barrier and MFMA phases in lockstep -- see DESIGN.md.)
    setup_halo(cur);
    load_halo(x_rsrc(cur.n), 0);
    load_b2(rb3[0]);                 // K-step 0
    zero_acc();
    store_halo();
    store_b2(rb3[0], 0);
    __syncthreads();
    int ks = 0;
    for (;;) {
        const int nit = it + gridDim.x;
        const bool more_items = nit < nitems;
        Item nxt = cur;
        if (more_items) nxt = decode(nit);
        for (int cc = 0; cc < nchunks; ++cc) {
            const bool more_cc = cc + 1 < nchunks;
            const bool have_next = more_cc || more_items;
            if (have_next) {                                      // next halo: one chunk ahead
                if (!more_cc) setup_halo(nxt);
                load_halo(x_rsrc(more_cc ? cur.n : nxt.n), more_cc ? cc + 1 : 0);
            }
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap, ++ks) {
                load_b2(rb3[0]);                                  // weights of K-step ks+1 (zeros past the end)
                if (!(dbg & 2)) mma_tap(tap, ks & 1);
                store_b2(rb3[0], (ks + 1) & 1);
                if (tap < 8) __syncthreads();
            }
            if (!more_cc) {
                __syncthreads();               // all waves finished reading the halo
                epilogue(cur);
                zero_acc();
                __syncthreads();
                finish_stats(cur);
            } else {
                __syncthreads();
            }
            if (have_next) store_halo();
            __syncthreads();
        }
        if (!more_items) break;
        it = nit;
        cur = nxt;
    }
}


// ---------------------------------------------------------------------------------------------------
// v3 "big K-step": one 4-wave block per CU (full register file, 132 KB LDS).  Per (patch of 256 pixels,
// 64-channel chunk) the halo AND the weights of all nine taps (9 x 64 co x 64 ci) are staged ONCE; each wave
// then issues 9 x 4 x 4 = 144 MFMAs with no barrier in between, so LDS reads and MFMAs pipeline freely inside
// the wave while the NEXT (patch, chunk) is prefetched into registers (29 x 16 B per lane).  N tile = 64 couts.
// Modelled on wgrad3x3_kernel, which reaches ~1 PFLOP/s with 72 MFMAs per barrier pair.
// ---------------------------------------------------------------------------------------------------
template <int DT, int TW, bool WRES, bool PREC = false>
__global__ __launch_bounds__(256, 1) void conv3x3_big_kernel(const C3Args a) {
    static_assert(!(WRES && PREC), "the precise mode has at least two K segments: weights cannot stay resident");
    typedef typename Elem<DT>::V8 V8;
    constexpr int BN = 64, BM = 256;
    constexpr int TH = BM / TW;
    constexpr int HWD = TW + 2, HHT = TH + 2, HP = HWD * HHT;
    constexpr int HALO_EL = HP * C3_LDR, W_EL = 9 * BN * C3_LDR;
    constexpr int HCH = (HP * 8 + 255) / 256;
    constexpr int WCH = 9 * BN * 8 / 256;           // 18 weight chunks per lane
    constexpr int TWS = (TW == 32) ? 5 : 4;
    constexpr int STG_EL = 32 * C3_LDR;
    __shared__ __attribute__((aligned(16))) unsigned short smem[HALO_EL + W_EL];
    unsigned short* halo = smem;
    unsigned short* Ws = smem + HALO_EL;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave;
    const int l31 = lane & 31, h = lane >> 5;
    const int nitems = a.nblocks;
    const int dbg = a.act >> 8;
    const int act = a.act & 0xff;
    const int chunk = t & 7, rbase = t >> 3;
    const int nchunks_c = (a.Cin + 63) >> 6;               // 64-channel chunks of one slice
    const int nchunks = a.ndz * nchunks_c;                  // K stages per item: (depth tap, channel chunk)
    // stage cc of an item that lives in slice n: source slice, weight-slot offset, validity (zero padding in depth)
    struct StageSrc { int n; unsigned sc, wsc; unsigned kill; unsigned ckill; };   // ckill: this lane's 8 channels >= Cin
    const unsigned img_bytes = (unsigned)a.H * a.W * a.in_stride * 2u;

#ifdef GS_C3_PHASE_TIMING
    long long ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tq = clock64();
#define PH(i) do { __builtin_amdgcn_sched_barrier(0); const long long t_ = clock64(); ph[i] += t_ - tq; tq = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PH(i) do {} while (0)
#endif
    struct Item { int n, y0, x0, n0, mtile; };
    auto decode = [&](int it) __attribute__((always_inline)) {
        Item r;
        const int ntile = it % a.ntn;
        int patch = it / a.ntn;
        r.mtile = patch;
        const int tx = patch % a.tiles_x; patch /= a.tiles_x;
        const int ty = patch % a.tiles_y;
        r.n = patch / a.tiles_y;
        r.y0 = ty * TH; r.x0 = tx * TW; r.n0 = ntile * BN;
        return r;
    };
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (unsigned)(9u * a.ndz * a.Cout * a.Cin * 2u), 0x00020000);

    // Staging addresses are strength reduced: per ITEM each lane keeps the byte offsets of its two weight rows
    // and of its halo pixels (voffset; 0x80000000 = out of range -> the buffer load returns zeros), the per-tap
    // and per-chunk parts are wave-uniform and ride in the scalar soffset operand: no vector ALU per load.
    constexpr unsigned VOOB = 0x80000000u;
    uint4 rh[HCH], rw[WCH];
    unsigned wv[2], hv[HCH];
    const unsigned tap_stride = (unsigned)a.Cout * a.Cin * 2u;
    auto setup_w = [&](int n0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = n0 + rbase + 32 * j;
            wv[j] = co < a.Cout ? (unsigned)((co * a.Cin + chunk * 8) * 2) : VOOB;
        }
    };
    auto setup_h = [&](int y0, int x0, int j) __attribute__((always_inline)) {      // j compile-time
        const int hp = rbase + 32 * j;
        const int hy = hp / HWD, hx = hp - hy * HWD;
        const int gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = hp < HP && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        hv[j] = ok ? (unsigned)(((gy * a.W + gx) * a.in_stride + a.in_coff + chunk * 8) * 2) : VOOB;
    };
    auto setup_item = [&](const Item& itn) __attribute__((always_inline)) {
        setup_w(itn.n0);
#pragma unroll
        for (int j = 0; j < HCH; ++j) setup_h(itn.y0, itn.x0, j);
    };
    // WRES (one channel chunk and one N tile: the 64->64 layers): the 9-tap weight slab is the same for every
    // item of the block -- it is staged once by the first stage and stays resident in LDS.
    const unsigned tap_stride_c = (unsigned)a.Cout * a.Cin * 2u;
    auto stage_src = [&](int n, int cc) __attribute__((always_inline)) {
        StageSrc r;
        int dzi = 0, c = cc;
        if (a.ndz > 1) { dzi = cc / nchunks_c; c = cc - dzi * nchunks_c; }
        const int dz = a.tap_dz[dzi];
        const int d = a.D > 1 ? n % a.D : 0;
        r.n = n + dz;
        r.kill = ((unsigned)(d + dz) < (unsigned)a.D) ? 0u : VOOB;
        int cx = c;                                               // input chunk of K chunk c
        if (PREC && a.in_wrap > 0 && c >= a.in_wrap) cx = c - a.in_wrap + a.in_wrap_to;
        r.sc = (unsigned)cx * 128u;                               // 64 channels x 2 bytes per chunk
        r.wsc = (unsigned)c * 128u + (unsigned)(dzi * 9) * tap_stride_c;
        r.ckill = (c * 64 + chunk * 8 < a.Cin) ? 0u : VOOB;       // Cin % 64 != 0: the tail chunk is zero padded
        return r;
    };
    auto load_stage = [&](const Item& itn, int cc, auto with_w) __attribute__((always_inline)) {
        const StageSrc ss = stage_src(itn.n, cc);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.x + (int64_t)(ss.kill ? itn.n : ss.n) * a.H * a.W * a.in_stride), 0, img_bytes, 0x00020000);
        if (decltype(with_w)::value) {
#pragma unroll
            for (int j = 0; j < WCH; ++j) {                       // row = rbase + 32 j = (j>>1)*64 + (rbase + 32 (j&1))
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wv[j & 1] | ss.ckill, ss.wsc + (unsigned)(j >> 1) * tap_stride, 0);
                rw[j] = make_uint4(v[0], v[1], v[2], v[3]);
            }
        }
#pragma unroll
        for (int j = 0; j < HCH; ++j) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rx, hv[j] | ss.kill | ss.ckill, ss.sc, 0);
            rh[j] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    };
    auto store_stage = [&](auto with_w) __attribute__((always_inline)) {
        if (decltype(with_w)::value) {
#pragma unroll
            for (int j = 0; j < WCH; ++j)
                *reinterpret_cast<uint4*>(Ws + (rbase + 32 * j) * C3_LDR + chunk * 8) = rw[j];
        }
#pragma unroll
        for (int j = 0; j < HCH; ++j)
            if (rbase + 32 * j < HP) *reinterpret_cast<uint4*>(halo + (rbase + 32 * j) * C3_LDR + chunk * 8) = rh[j];
    };

    int a_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = (wm * 2 + i) * 32 + l31;
        a_off[i] = (((p >> TWS) + 1) * HWD + (p & (TW - 1)) + 1) * C3_LDR + h * 8;
    }
    const int b_off = l31 * C3_LDR + h * 8;

    f32x16 acc[2][2];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };

    // ---- epilogue (same scheme as conv3x3_persist_kernel, N tile = 64) ----
    static_assert((PREC ? 8 : 4) * STG_EL + 4 * 2 * BN * 2 <= HALO_EL, "epilogue staging must fit in the halo region");
    unsigned short* stg = halo + wave * STG_EL;                   // staging overlays the consumed halo (the weights
    float* red = reinterpret_cast<float*>(halo + 4 * STG_EL);     // may be resident); red = [4 waves][2][64]
    unsigned short* stg_lo = halo + 4 * STG_EL + 4 * 2 * BN * 2 + wave * STG_EL;      // PREC: the lo halves
    const bool odd = lane & 1;
    const unsigned int psel = odd ? 0x03020706u : 0x05040100u;
    const float neg_slope = act == GS_ACT_RELU ? 0.f : (act == GS_ACT_LEAKY02 ? 0.2f : 1.f);
    auto epilogue_t = [&](const Item& itc, auto plain_tag, auto full_tag) __attribute__((always_inline)) {
        constexpr bool PLAIN = decltype(plain_tag)::value;        // no bias, no activation (every U-Net conv)
        constexpr bool FULL = decltype(full_tag)::value;          // patch completely inside the image
        // opaque copies: hipcc otherwise hoists the per-item edge weights and store addresses of BOTH
        // instantiations into the item loop header (~600 VALU instructions per item, measured 23% of the
        // 64->64 layers) although only one path runs
        int e_y0 = itc.y0, e_x0 = itc.x0, e_n = itc.n, e_n0 = itc.n0;
        asm volatile("" : "+s"(e_y0), "+s"(e_x0), "+s"(e_n), "+s"(e_n0));
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.y + (int64_t)e_n * a.H * a.W * a.out_stride), 0, (unsigned)a.H * a.W * a.out_stride * 2u, 0x00020000);
        const __amdgpu_buffer_rsrc_t ry_lo = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((PREC ? a.y_lo : a.y) + (int64_t)e_n * a.H * a.W * a.out_stride), 0,
            (unsigned)a.H * a.W * a.out_stride * 2u, 0x00020000);
        float bv[2] = {0.f, 0.f};
        if (!PLAIN) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int co = e_n0 + j * 32 + l31;
                bv[j] = (a.bias != nullptr && co < a.Cout) ? a.bias[co] : 0.f;
            }
        }
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
        const bool want_stats = a.bnp != nullptr;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int prow0 = (wm * 2 + i) * 32;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int r0 = 2 * m;
                const int rowa = (r0 & 3) + 8 * (r0 >> 2) + 4 * h;
                float w0 = 1.f, w1 = 1.f;
                if (!FULL && want_stats) {
                    const int p0 = prow0 + rowa;
                    const int gy0 = e_y0 + (p0 >> TWS), gx0 = e_x0 + (p0 & (TW - 1));
                    const int gy1 = e_y0 + ((p0 + 1) >> TWS), gx1 = e_x0 + ((p0 + 1) & (TW - 1));
                    w0 = (float)((unsigned)((gy0 - a.H) & (gx0 - a.W)) >> 31);
                    w1 = (float)((unsigned)((gy1 - a.H) & (gx1 - a.W)) >> 31);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v0 = acc[i][j][r0], v1 = acc[i][j][r0 + 1];
                    if (want_stats) {
                        if (FULL) {
                            s1[j] += v0 + v1;
                            s2[j] += v0 * v0 + v1 * v1;
                        } else {
                            const float u0 = v0 * w0, u1 = v1 * w1;
                            s1[j] += u0 + u1;
                            s2[j] += u0 * u0 + u1 * u1;
                        }
                    }
                    if (!PLAIN) {
                        v0 += bv[j];
                        v1 += bv[j];
                        v0 = v0 > 0.f ? v0 : v0 * neg_slope;
                        v1 = v1 > 0.f ? v1 : v1 * neg_slope;
                    }
                    const unsigned int own = Elem<DT>::pack2(v0, v1);
                    const unsigned int oth = dpp_xor1(own);
                    const unsigned int pk = __builtin_amdgcn_perm(oth, own, psel);
                    const int row = rowa + (odd ? 1 : 0);
                    *reinterpret_cast<unsigned int*>(stg + row * C3_LDR + j * 32 + (l31 & ~1)) = pk;
                    if (PREC) {                               // lo = 16-bit(value - hi): the pair carries ~22 bits
                        const float l0 = v0 - Elem<DT>::to_f((unsigned short)(own & 0xffffu));
                        const float l1 = v1 - Elem<DT>::to_f((unsigned short)(own >> 16));
                        const unsigned int own_l = Elem<DT>::pack2(l0, l1);
                        const unsigned int oth_l = dpp_xor1(own_l);
                        const unsigned int pk_l = __builtin_amdgcn_perm(oth_l, own_l, psel);
                        *reinterpret_cast<unsigned int*>(stg_lo + row * C3_LDR + j * 32 + (l31 & ~1)) = pk_l;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            PH(8 + 2 * i);
            // buffer stores: 32-bit offsets inside the image, out-of-range lanes get an out-of-range offset (the
            // hardware drops them) -- no exec-mask branch per store, so the four LDS reads are issued back to back
            uint4 sv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                sv[q] = *reinterpret_cast<const uint4*>(stg + (q * 8 + (lane >> 3)) * C3_LDR + (lane & 7) * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = prow0 + q * 8 + (lane >> 3);
                const int gy = e_y0 + (p >> TWS), gx = e_x0 + (p & (TW - 1));
                const int co = e_n0 + (lane & 7) * 8;
                const bool ok = (FULL || (gy < a.H && gx < a.W)) && co < a.Cout && !(dbg & 1);
                const unsigned off = ok ? (unsigned)(((gy * a.W + gx) * a.out_stride + a.out_coff + co) * 2) : VOOB;
                u32x4 d;
                d[0] = sv[q].x; d[1] = sv[q].y; d[2] = sv[q].z; d[3] = sv[q].w;
                __builtin_amdgcn_raw_buffer_store_b128(d, ry, off, 0, 0);
                if (PREC) {
                    const uint4 lv = *reinterpret_cast<const uint4*>(stg_lo + (q * 8 + (lane >> 3)) * C3_LDR + (lane & 7) * 8);
                    u32x4 dl;
                    dl[0] = lv.x; dl[1] = lv.y; dl[2] = lv.z; dl[3] = lv.w;
                    __builtin_amdgcn_raw_buffer_store_b128(dl, ry_lo, off, 0, 0);
                }
            }
            __builtin_amdgcn_wave_barrier();
            PH(9 + 2 * i);
        }
        if (want_stats) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                s1[j] += __shfl_xor(s1[j], 32, 64);
                s2[j] += __shfl_xor(s2[j], 32, 64);
                if (h == 0) {
                    red[(wm * 2 + 0) * BN + j * 32 + l31] = s1[j];
                    red[(wm * 2 + 1) * BN + j * 32 + l31] = s2[j];
                }
            }
        }
    };
    const bool plain = (a.bias == nullptr && act == GS_ACT_NONE);
    auto epilogue = [&](const Item& itc) __attribute__((always_inline)) {
        const bool full = (itc.y0 + TH <= a.H) && (itc.x0 + TW <= a.W);
        if (plain && full) epilogue_t(itc, std::true_type{}, std::true_type{});
        else epilogue_t(itc, std::false_type{}, std::false_type{});
    };
    auto finish_stats = [&](const Item& itc) __attribute__((always_inline)) {
        if (a.bnp != nullptr && t < BN && itc.n0 + t < a.Cout) {
            float v1 = 0.f, v2 = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) { v1 += red[(m * 2 + 0) * BN + t]; v2 += red[(m * 2 + 1) * BN + t]; }
            float* dst = a.bnp + (int64_t)itc.mtile * 2 * a.Cout + itc.n0 + t;
            dst[0] = v1;
            dst[a.Cout] = v2;
        }
    };

    // XCD-aware item order (a.xcd_order, full 256-block grids): workgroups are dealt round-robin over the 8 XCDs, so block
    // b and b + 8 share an L2.  Items are numbered ntile-fastest (the Cout/64 tiles of one patch are consecutive): block b
    // starts at item (b % 8) * 32 + b / 8, so that in every round the 32 blocks of an XCD hold 32 CONSECUTIVE items = all
    // cout tiles of the same few patches -- the patch's input halo is fetched into that L2 once instead of once per XCD
    // (it was re-read from HBM / Infinity Cache once per cout tile: 1.8x the algorithmic traffic).
    int it = a.xcd_order ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    if (it >= nitems) return;
    Item cur = decode(it);
    // items advance by gridDim.x: the (ntile, tx, ty, n) digits are stepped with carries instead of being
    // re-derived with six integer divisions per item (measured: ~3.5k cycles per item, 23% of the 64->64 layers)
    int dg0 = it % a.ntn, dg1, dg2, dg3;
    {
        int r = it / a.ntn;
        dg1 = r % a.tiles_x; r /= a.tiles_x;
        dg2 = r % a.tiles_y; dg3 = r / a.tiles_y;
    }
    int st0, st1, st2, st3;
    {
        int r = gridDim.x;
        st0 = r % a.ntn; r /= a.ntn;
        st1 = r % a.tiles_x; r /= a.tiles_x;
        st2 = r % a.tiles_y; st3 = r / a.tiles_y;
    }
    auto advance_item = [&]() __attribute__((always_inline)) {
        Item r;
        dg0 += st0; int c = dg0 >= a.ntn ? 1 : 0; dg0 -= c * a.ntn;
        dg1 += st1 + c; c = dg1 >= a.tiles_x ? 1 : 0; dg1 -= c * a.tiles_x;
        dg2 += st2 + c; c = dg2 >= a.tiles_y ? 1 : 0; dg2 -= c * a.tiles_y;
        dg3 += st3 + c;
        r.n = dg3; r.y0 = dg2 * TH; r.x0 = dg1 * TW; r.n0 = dg0 * BN;
        r.mtile = (dg3 * a.tiles_y + dg2) * a.tiles_x + dg1;
        return r;
    };
    setup_item(cur);
    load_stage(cur, 0, std::true_type{});
    zero_acc();
    bool first = true;
    for (;;) {
        const int nit = it + gridDim.x;
        const bool more_items = nit < nitems;
        Item nxt = cur;
        if (more_items) nxt = advance_item();
        for (int cc = 0; cc < nchunks; ++cc) {
            const bool more_cc = cc + 1 < nchunks;
#ifdef GS_C3_PHASE_TIMING
            PH(6);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PH(7);
#endif
            __syncthreads();                       // previous stage fully consumed (and epilogue staging done)
            PH(0);
            if (WRES) {
                if (first) store_stage(std::true_type{});         // only the very first stage carries the weights
                else store_stage(std::false_type{});
                first = false;
            } else {
                store_stage(std::true_type{});
            }
            __syncthreads();
            PH(1);
            // The prefetch of the next stage is spread over the K loop (one 16-byte load per lane per MFMA
            // step): issued in one burst in front of the loop, its 29 loads per lane keep the wave in the
            // memory-issue queue (~2k cycles per CU) before the first MFMA can start.  No next stage: the
            // voffsets are forced out of range and the loads return zeros that nobody reads.
            // With one wave per SIMD nothing hides ALU latency: in the single-chunk (WRES) kernel, where a new
            // item starts every stage, the ~250 address instructions of the next item's halo offsets are
            // interleaved with the first SH MFMA steps as well and the loads follow.  (Multi-chunk layers change
            // item once per nchunks stages; recomputing the offsets every stage costs more than it hides.)
            const bool have_next = more_cc || more_items;
            const Item& ldi = more_cc ? cur : nxt;
            constexpr int SH = WRES ? (HCH + 1) / 2 : 0;
            if (!WRES && !more_cc && more_items) setup_item(nxt);
            const StageSrc ssn = stage_src(ldi.n, more_cc ? cc + 1 : 0);
            const __amdgpu_buffer_rsrc_t rx_n = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(a.x + (int64_t)(ssn.kill ? ldi.n : ssn.n) * a.H * a.W * a.in_stride), 0, img_bytes, 0x00020000);
            const unsigned sc_n = ssn.sc, wsc_n = ssn.wsc;
            const unsigned kill = (have_next ? 0u : VOOB) | ssn.ckill;
            const unsigned killh = kill | ssn.kill;              // depth tap outside the volume: zero slice
            auto issue_load = [&](int j) __attribute__((always_inline)) {       // j is a compile-time constant
                constexpr int NW = WRES ? 0 : WCH;
                if (j < NW) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wv[j & 1] | kill,
                                                                          wsc_n + (unsigned)(j >> 1) * tap_stride, 0);
                    rw[j < NW ? j : 0] = make_uint4(v[0], v[1], v[2], v[3]);
                } else if (j - NW < HCH) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rx_n, hv[j - NW < HCH ? j - NW : 0] | killh, sc_n, 0);
                    rh[j - NW < HCH ? j - NW : 0] = make_uint4(v[0], v[1], v[2], v[3]);
                }
            };
            static_assert(SH + (WRES ? 0 : WCH) + HCH <= 36, "address setup + prefetch loads must fit the 36 MFMA steps");
            auto setup_step = [&](int step) __attribute__((always_inline)) {    // step compile-time, < SH
                int py0 = ldi.y0, px0 = ldi.x0, pn0 = ldi.n0;
                asm volatile("" : "+s"(py0), "+s"(px0), "+s"(pn0));              // keep this arithmetic in the K loop
                if (step == 0) setup_w(pn0);
                if (2 * step < HCH) setup_h(py0, px0, 2 * step < HCH ? 2 * step : 0);
                if (2 * step + 1 < HCH) setup_h(py0, px0, 2 * step + 1 < HCH ? 2 * step + 1 : 0);
            };
            if (dbg & 2) {
                if (WRES) setup_item(ldi);
#pragma unroll
                for (int j = 0; j < 36; ++j) issue_load(j);
            }
            if (!(dbg & 2)) {
                // 36 steps (tap, kk) of 4 MFMAs; the four fragment reads of step s+1 are issued BEFORE the MFMAs
                // of step s (hipcc otherwise schedules them just-in-time behind lgkmcnt(0) and the LDS latency
                // is exposed three times per four MFMAs -- fatal with one wave per SIMD)
                V8 af[2][2], bf[2][2];
                auto frag_load = [&](int step, V8 (&fa)[2], V8 (&fb)[2]) __attribute__((always_inline)) {
                    const int tap = step >> 2, kk = step & 3;
                    const int toff = (a.tap_dy[tap] * HWD + a.tap_dx[tap]) * C3_LDR;
                    const unsigned short* B = Ws + tap * BN * C3_LDR + b_off;
#ifdef GS_C3_INTERLEAVE
                    // read order A0, B0, A1, B1 = the order the MFMAs (0,0), (1,0), (0,1), (1,1) of the NEXT step need them:
                    // every fragment is issued at least three MFMAs before its first use
                    fa[0] = *reinterpret_cast<const V8*>(halo + a_off[0] + toff + kk * 16);
                    fb[0] = *reinterpret_cast<const V8*>(B + kk * 16);
                    fa[1] = *reinterpret_cast<const V8*>(halo + a_off[1] + toff + kk * 16);
                    fb[1] = *reinterpret_cast<const V8*>(B + 32 * C3_LDR + kk * 16);
#else
#pragma unroll
                    for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const V8*>(halo + a_off[i] + toff + kk * 16);
#pragma unroll
                    for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const V8*>(B + j * 32 * C3_LDR + kk * 16);
#endif
                };
                frag_load(0, af[0], bf[0]);
#pragma unroll
                for (int step = 0; step < 36; ++step) {
                    const int cur = step & 1;
                    if (step + 1 < 36) frag_load(step + 1, af[cur ^ 1], bf[cur ^ 1]);
                    if (step < SH) setup_step(step);
                    else issue_load(step - SH);
#ifndef GS_C3_INTERLEAVE
                    __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef GS_C3_INTERLEAVE
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int i = 0; i < 2; ++i) acc[i][j] = Elem<DT>::mfma32(af[cur][i], bf[cur][j], acc[i][j]);
#else
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = Elem<DT>::mfma32(af[cur][i], bf[cur][j], acc[i][j]);
#endif
#ifdef GS_C3_INTERLEAVE
                    // one LDS fragment read in the shadow of every MFMA (an MFMA holds the issue port for 8 of its 32 cycles):
                    // issued as a block in front of the four MFMAs, the 4 reads + the prefetch load (~45 issue cycles) left
                    // the matrix pipe idle for ~20 cycles per step
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // one DS read
                    }
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);           // the prefetch buffer load
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            PH(2);
            if (!more_cc) {
                __syncthreads();                   // every wave is done reading the halo: staging may overlay it
                PH(3);
                if (!(dbg & 4)) epilogue(cur);
                zero_acc();
                PH(4);
                __syncthreads();
                finish_stats(cur);
                PH(5);
            }
        }
        if (!more_items) break;
        it = nit;
        cur = nxt;
    }
#ifdef GS_C3_PHASE_TIMING
    if (blockIdx.x == 0 && lane == 0 && (a.bias != nullptr || a.bnp != nullptr)) {   // debug build only
        float* sink = a.bias != nullptr ? const_cast<float*>(a.bias) : a.bnp;      // bias / partials double as sink
        for (int i = 0; i < 16; ++i) sink[wave * 16 + i] = (float)ph[i];
    }
#endif
#undef PH
}

struct C3Plan { int bn, tw, th, tiles_x, tiles_y; };

int c3_variant_get();

C3Plan c3_plan(int H, int W, int Cout) {
    C3Plan p;
    // variant 2 (big K-step kernel) tiles every layer as 256-pixel patches x 64 couts; its fallbacks keep the same
    // tiling so gs_conv3x3_mtiles() never depends on which kernel finally runs
    p.bn = (Cout <= 64 || (c3_variant_get() == 2 && Cout % 8 == 0)) ? 64 : 128;
    p.tw = (W >= 24) ? 32 : 16;
    const int bm = (p.bn == 128) ? 128 : 256;
    p.th = bm / p.tw;
    p.tiles_x = cdiv(W, p.tw);
    p.tiles_y = cdiv(H, p.th);
    return p;
}

}  // namespace

// form of the LDS-DMA kernel (conv3x3_dma.hip): -1 = by CU fill (default), 0 = off (big K-step kernel), 4 / 8 = waves per block.
// GSSEG_C3_DMA presets it; tests and tools switch it through gs_conv3x3_set_kernel_form() to compare the forms on the same
// operands.  (Round 2 also carried a two-blocks-per-CU form and two 128-cout forms; all three measured slower -- numbers in
// DESIGN.md section 4 -- and were removed in round 3.)
static std::atomic<int> c3_dma_form{getenv("GSSEG_C3_DMA") ? atoi(getenv("GSSEG_C3_DMA")) : -1};

extern "C" int gs_conv3x3_set_kernel_form(int form) {
    GS_CHECK_ARG(form == -1 || form == 0 || form == 4 || form == 8 || form == 44,
                 "gs_conv3x3_set_kernel_form: form must be -1 (automatic), 0 (big K-step kernel), 4 / 8 (waves of the LDS-DMA kernel) or 44 (its quad form)");
    c3_dma_form.store(form, std::memory_order_relaxed);
    return GS_OK;
}

static int c3_variant() {          // 0 = v1 one patch per block, 1 = persistent (v2), 2 = big K-step (v3)
    static const int v = getenv("GSSEG_C3") ? atoi(getenv("GSSEG_C3")) : 2;
    return v;
}
namespace { int c3_variant_get() { return c3_variant(); } }

static bool c3_use_big(int Cout, int out_pix_stride, int out_coff) {
    return c3_variant() == 2 && (Cout % 8) == 0 && (out_pix_stride % 8) == 0 && (out_coff % 8) == 0;
}
static bool c3_big_ok(int H, int W, int Cin, int in_pix_stride, int Cout, int out_pix_stride) {
    // 32-bit buffer offsets with bit 31 as the "out of range" marker: one image and the weights stay below 2 GiB
    return (Cin % 8) == 0 && (int64_t)H * W * in_pix_stride * 2 < 2147483000LL &&
           (int64_t)H * W * out_pix_stride * 2 < 2147483000LL && (int64_t)9 * Cout * Cin * 2 < 2147483000LL;
}

extern "C" int gs_conv3x3_mtiles(int N, int H, int W, int Cout) {
    const C3Plan p = c3_plan(H, W, Cout);
    return N * p.tiles_x * p.tiles_y;
}

// ---- the LDS-DMA kernel's launch plan: shared by the launch and by gs_conv3x3_stat_rows() ----
static int c3_max_blocks() {
    static const int v = getenv("GSSEG_C3_GRID") ? atoi(getenv("GSSEG_C3_GRID")) : 256;
    return v;
}
// persistent-grid cap at run time (parallel.py leaves CUs to RCCL when world > 1): 0 = the default above
static std::atomic<int> c3_grid_cap{0};
static int c3_blocks_now() {
    const int cap = c3_grid_cap.load(std::memory_order_relaxed);
    return cap > 0 ? cap : c3_max_blocks();
}
extern "C" int gs_set_persistent_grid(int blocks) {
    GS_CHECK_ARG(blocks == 0 || (blocks >= 8 && blocks <= 1024), "gs_set_persistent_grid: 0 (default) or 8..1024 blocks");
    c3_grid_cap.store(blocks, std::memory_order_relaxed);
    return GS_OK;
}
extern "C" int gs_get_persistent_grid(void) { return c3_blocks_now(); }

struct C3DmaPlan { int waves, tiles_x, tiles_y, ntn, nitems, grid; };
// shape part of the eligibility (the launch also needs the standard / flipped tap table and 16-byte aligned strides)
static bool c3_dma_shape_ok(int W, int Cin, int Cout) {
    return c3_variant() == 2 && c3_dma_form.load(std::memory_order_relaxed) != 0 && W >= 24 && Cin % 64 == 0 && Cout % 8 == 0;
}
// q8_cin: the layer's channels when the launch is a "q" stage (0: not one).  A 64-channel "q" stage -- two 16-bit and two FP8 K stages
// per item -- runs faster in the 4-wave form: the 8-wave Q8 kernel spills (its epilogue reloads ~90 dwords per lane and item),
// which the four stages of such an item do not amortise (measured 64->64 @256^2: 409 vs 459 us; from 128 channels on the 8-wave form wins)
static C3DmaPlan c3_dma_plan(int N, int H, int W, int Cout, bool per_block_rows, int q8_cin = 0, bool pair_form = false) {
    C3DmaPlan p;
    const int blocks = c3_blocks_now();
    p.tiles_x = cdiv(W, 32);
    p.ntn = cdiv(Cout, 64);
    // the 8-wave form (16x32-pixel items, -11..13 % against the big K-step kernel over the U-Net layer shapes) unless its
    // items -- twice the work each -- fill the CUs worse than the 4-wave form's (-2 %) do
    const int64_t items4 = (int64_t)N * p.tiles_x * cdiv(H, 8) * p.ntn, items8 = (int64_t)N * p.tiles_x * cdiv(H, 16) * p.ntn;
    const double cost4 = 0.98 * (double)((items4 + blocks - 1) / blocks);
    const double cost8 = 0.87 * 2.0 * (double)((items8 + blocks - 1) / blocks);
    const int forced = c3_dma_form.load(std::memory_order_relaxed);
    p.waves = (forced == 4 || forced == 8) ? forced : (q8_cin == 64 ? 4 : (cost8 <= cost4 ? 8 : 4));
    if (forced == 44 && !pair_form) p.waves = 44;          // the quad form: 4 waves x 4 pixel rows (16-bit forward / data gradient only)
    p.tiles_y = cdiv(H, p.waves == 4 ? 8 : 16);
    p.nitems = N * p.tiles_x * p.tiles_y * p.ntn;
    p.grid = per_block_rows ? c3_dma_grid(p.nitems, p.ntn, blocks) : (p.nitems < blocks ? p.nitems : blocks);
    return p;
}

// Rows of BatchNorm partial sums ([rows][2][Cout] fp32) a gs_conv3x3 / gs_conv3d_3x3x3 / gs_conv3x3_precise launch with these
// dimensions writes -- what gs_bn_finalize / gs_bn_partials_colsum must be told.  The LDS-DMA kernel keeps its sums in
// registers across the items of a block and writes ONE row per block and cout-tile group (the pair forward too: `pair` tells
// which K extent the shape test sees); the other kernels write one row per 8x32 / 16x16 patch (= gs_conv3x3_mtiles).  Never more than gs_conv3x3_mtiles(): a buffer of
// that many rows always suffices.  Assumes what every caller in this package does: the forward or the flipped tap table and
// channel strides / offsets that are multiples of 8 (a launch that would fall off that path with statistics fails loudly).
extern "C" int gs_conv3x3_stat_rows(int N, int H, int W, int Cin, int Cout, int pair) {
    static const int prec_dma = getenv("GSSEG_C3_PREC_DMA") ? atoi(getenv("GSSEG_C3_PREC_DMA")) : 1;
    if ((!pair || prec_dma != 0) && c3_dma_shape_ok(W, Cin, Cout)) {
        const C3DmaPlan p = c3_dma_plan(N, H, W, Cout, true, pair == 2 ? Cin / 2 : 0, pair != 0);      // pair == 2: a "q" stage, Cin = its K = 2 * channels
        return p.grid / p.ntn;
    }
    return gs_conv3x3_mtiles(N, H, W, Cout);
}

static int conv3x3_launch(const void* x, const void* w, void* y, const float* bias, float* bn_partials, int N, int H,
                          int W, int Cin, int in_pix_stride, int in_coff, int Cout, int out_pix_stride, int out_coff,
                          const int32_t* tap_dy, const int32_t* tap_dx, int act, int dtype, void* stream, int D, int ndz,
                          const int32_t* tap_dz, void* y_lo = nullptr, int in_wrap = 0, const int* wexp = nullptr, int in_wrap_to = 0) {
    GS_CHECK_ARG(x && w && y && tap_dy && tap_dx, "gs_conv3x3: null pointer");
    GS_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 8 == 0 && Cout > 0, "gs_conv3x3: bad dims");
    const bool prec = y_lo != nullptr;
    // precise mode: Cin is the K extent (segments x channels); the input holds in_wrap channels
    GS_CHECK_ARG(!prec || (in_wrap > 0 && in_wrap % 64 == 0 && Cin % 64 == 0 && Cin >= in_wrap && Cin <= 2 * in_wrap),
                 "gs_conv3x3_precise: K extent %d / wrap %d must be multiples of 64 with wrap <= K <= 2*wrap", Cin, in_wrap);
    GS_CHECK_ARG(in_pix_stride >= in_coff + (prec ? in_wrap : Cin) && in_pix_stride % 8 == 0 && in_coff % 8 == 0, "gs_conv3x3: bad input stride");
    GS_CHECK_ARG(out_pix_stride >= out_coff + Cout, "gs_conv3x3: bad output stride");
    GS_CHECK_ARG((int64_t)N * H * W < 2147483000LL, "gs_conv3x3: pixel count exceeds int32");
    GS_CHECK_ARG((int64_t)H * W * in_pix_stride * 2 < 4294967000LL && (int64_t)9 * Cout * Cin * 2 < 4294967000LL,
                 "gs_conv3x3: one image / the weight pack must stay below 4 GiB (32-bit buffer offsets)");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_conv3x3: bad dtype");
    C3Args a;
    a.x = (const unsigned short*)x; a.w = (const unsigned short*)w; a.y = (unsigned short*)y;
    // bits 8.. of `act` are ablation switches of the big kernel (0x100 skip stores, 0x200 skip the K loop, 0x400 skip the
    // epilogue: tools/bench_layers.py); they are honoured only when GSSEG_C3_DEBUG=1 is set in the environment
    static const bool dbg_env = getenv("GSSEG_C3_DEBUG") && atoi(getenv("GSSEG_C3_DEBUG")) != 0;
    if (!dbg_env) act &= 0xff;
    a.bias = bias; a.bnp = bn_partials; a.act = act;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.in_stride = in_pix_stride; a.in_coff = in_coff;
    a.Cout = Cout; a.out_stride = out_pix_stride; a.out_coff = out_coff;
    for (int i = 0; i < 9; ++i) {
        GS_CHECK_ARG(tap_dy[i] >= -1 && tap_dy[i] <= 1 && tap_dx[i] >= -1 && tap_dx[i] <= 1, "gs_conv3x3: tap offsets must be in [-1,1]");
        a.tap_dy[i] = tap_dy[i]; a.tap_dx[i] = tap_dx[i];
    }
    a.D = D; a.ndz = ndz;
    GS_CHECK_ARG(in_wrap_to == 0 || (prec && wexp == nullptr && in_wrap_to % 64 == 0 && in_wrap_to > 0 && in_wrap_to + (Cin - in_wrap) <= in_wrap),
                 "gs_conv3x3_precise: the wrapped K chunks must stay inside the input (wrap_to %d + K %d - wrap %d <= wrap)", in_wrap_to, Cin, in_wrap);
    a.y_lo = (unsigned short*)y_lo; a.in_wrap = in_wrap / 64; a.in_wrap_to = in_wrap_to / 64; a.xcd_order = 0;
    // "q" stages (gs_conv3x3_q8): Cin = K = 2 * (the layer's channels): the first half 16-bit stages, the second FP8 correction stages
    a.q8_c0 = wexp ? Cin / 64 : 0; a.wexp = wexp;
    for (int i = 0; i < 3; ++i) a.tap_dz[i] = (tap_dz && i < ndz) ? tap_dz[i] : 0;
    const C3Plan p = c3_plan(H, W, Cout);
    a.tiles_x = p.tiles_x; a.tiles_y = p.tiles_y;
    a.ntn = cdiv(Cout, p.bn);
    a.nblocks = N * p.tiles_x * p.tiles_y * a.ntn;
    hipStream_t s = (hipStream_t)stream;
    GS_CHECK_ARG((act & 0xff) == GS_ACT_NONE || (act & 0xff) == GS_ACT_RELU || (act & 0xff) == GS_ACT_LEAKY02,
                 "gs_conv3x3: activation %d not supported (use gs_conv_igemm)", act & 0xff);
    const bool big = c3_use_big(Cout, out_pix_stride, out_coff) && c3_big_ok(H, W, Cin, in_pix_stride, Cout, out_pix_stride) &&
                     (int64_t)9 * ndz * Cout * Cin * 2 < 2147483000LL;
    GS_CHECK_ARG(big || ndz == 1, "gs_conv3d_3x3x3: needs Cin %% 8 == 0, Cout %% 8 == 0 and 16-byte aligned output channels");
    if (big) {
        const int tw = (W >= 24) ? 32 : 16, th = 256 / tw;
        a.tiles_x = cdiv(W, tw); a.tiles_y = cdiv(H, th);
        a.ntn = cdiv(Cout, 64);
        a.nblocks = N * a.tiles_x * a.tiles_y * a.ntn;
        const int big_blocks = c3_blocks_now();
        dim3 bgrid(a.nblocks < big_blocks ? a.nblocks : big_blocks);
        static const int xcd_env = getenv("GSSEG_C3_XCD") ? atoi(getenv("GSSEG_C3_XCD")) : 1;
        // activation-heavy launches only: where the weights outweigh the input (the 16x16 level at batch 32: 18.9 MB of
        // weights vs 16.8 MB of input) the plain order -- XCD x sees the N tiles {x, x+8} -- fetches each weight slab into one
        // or two L2s instead of all eight (measured FETCH 144 vs 223 MB per launch there, 338 vs 185 MB on the big layers)
        a.xcd_order = (xcd_env && (bgrid.x % 8) == 0 && a.ntn > 1 && (int64_t)N * H * W > (int64_t)18 * Cout) ? 1 : 0;
        hipStream_t bs = (hipStream_t)stream;
        const bool wres = (Cin <= 64 && a.ntn == 1 && ndz == 1);   // one stage, one N tile: weights stay resident in LDS
        // LDS-DMA kernel (conv3x3_dma.hip) for the layers it covers (2-D and Conv3d: depth taps = stages); GSSEG_C3_DMA = 0
        // (off) / 4 / 8 forces a form.  The pair forward (prec) keeps the forward tap table and per-patch statistics rows.
        bool std_taps = true, flip_taps = true;
        for (int i = 0; i < 9; ++i) {
            std_taps = std_taps && tap_dy[i] == i / 3 - 1 && tap_dx[i] == i % 3 - 1;
            flip_taps = flip_taps && tap_dy[i] == 1 - i / 3 && tap_dx[i] == 1 - i % 3;
        }
        static const int prec_dma_env = getenv("GSSEG_C3_PREC_DMA") ? atoi(getenv("GSSEG_C3_PREC_DMA")) : 1;
        const bool dma = c3_dma_shape_ok(W, Cin, Cout) && (prec ? (std_taps && prec_dma_env != 0) : (std_taps || flip_taps));
        GS_CHECK_ARG(wexp == nullptr || (dma && dtype == GS_F16 && Cin % 128 == 0),
                     "gs_conv3x3_q8: needs the LDS-DMA kernel (W >= 24, channels %% 64 == 0, Cout %% 8 == 0, forward taps) and fp16");
        if (dma) {
            const C3DmaPlan dp = c3_dma_plan(N, H, W, Cout, true, wexp ? Cin / 2 : 0, prec);
            a.tiles_x = dp.tiles_x; a.tiles_y = dp.tiles_y; a.ntn = dp.ntn; a.nblocks = dp.nitems;
            a.xcd_order = (xcd_env && (dp.grid % 8) == 0 && a.ntn > 1 && (int64_t)N * H * W > (int64_t)18 * Cout) ? 1 : 0;
            c3_dma_launch(a, dp.waves, prec, dtype, dp.grid, bs);
            GS_CHECK_LAUNCH(prec ? "gs_conv3x3_precise" : "gs_conv3x3");
            return GS_OK;
        }
        // statistics rows: gs_conv3x3_stat_rows() promised the per-block rows of the LDS-DMA kernel for this shape
        GS_CHECK_ARG(bn_partials == nullptr || !c3_dma_shape_ok(W, Cin, Cout) || (prec && prec_dma_env == 0),
                     "gs_conv3x3: BatchNorm partials with a tap table / layout outside the LDS-DMA kernel (gs_conv3x3_stat_rows would be wrong)");
        if (prec) {
            if (dtype == GS_F16) {
                if (tw == 32) conv3x3_big_kernel<GS_F16, 32, false, true><<<bgrid, 256, 0, bs>>>(a);
                else conv3x3_big_kernel<GS_F16, 16, false, true><<<bgrid, 256, 0, bs>>>(a);
            } else {
                if (tw == 32) conv3x3_big_kernel<GS_BF16, 32, false, true><<<bgrid, 256, 0, bs>>>(a);
                else conv3x3_big_kernel<GS_BF16, 16, false, true><<<bgrid, 256, 0, bs>>>(a);
            }
            GS_CHECK_LAUNCH("gs_conv3x3_precise");
            return GS_OK;
        }
        if (dtype == GS_F16) {
            if (wres) {
                if (tw == 32) conv3x3_big_kernel<GS_F16, 32, true><<<bgrid, 256, 0, bs>>>(a);
                else conv3x3_big_kernel<GS_F16, 16, true><<<bgrid, 256, 0, bs>>>(a);
            } else {
                if (tw == 32) conv3x3_big_kernel<GS_F16, 32, false><<<bgrid, 256, 0, bs>>>(a);
                else conv3x3_big_kernel<GS_F16, 16, false><<<bgrid, 256, 0, bs>>>(a);
            }
        } else {
            if (wres) {
                if (tw == 32) conv3x3_big_kernel<GS_BF16, 32, true><<<bgrid, 256, 0, bs>>>(a);
                else conv3x3_big_kernel<GS_BF16, 16, true><<<bgrid, 256, 0, bs>>>(a);
            } else {
                if (tw == 32) conv3x3_big_kernel<GS_BF16, 32, false><<<bgrid, 256, 0, bs>>>(a);
                else conv3x3_big_kernel<GS_BF16, 16, false><<<bgrid, 256, 0, bs>>>(a);
            }
        }
        GS_CHECK_LAUNCH("gs_conv3x3");
        return GS_OK;
    }
    GS_CHECK_ARG(bn_partials == nullptr || !c3_dma_shape_ok(W, Cin, Cout),
                 "gs_conv3x3: BatchNorm partials with channel strides / offsets that are not multiples of 8 (gs_conv3x3_stat_rows would be wrong)");
    GS_CHECK_ARG(!prec, "gs_conv3x3_precise: needs the big-K-step kernel (GSSEG_C3=2, Cout %% 8 == 0, 16-byte aligned output channels)");
    GS_CHECK_ARG(wexp == nullptr, "gs_conv3x3_q8: shape outside the LDS-DMA kernel");
    static const bool force_v1 = c3_variant() == 0;
    // the persistent kernel stores whole 16-byte channel groups; odd shapes go to the one-patch-per-block kernel
    const bool use_v1 = force_v1 || (Cout % 8) != 0 || (out_pix_stride % 8) != 0 || (out_coff % 8) != 0;
    static const int persist_blocks = getenv("GSSEG_C3_GRID") ? atoi(getenv("GSSEG_C3_GRID")) : 512;
    dim3 grid(a.nblocks), block(256);
    if (!use_v1) {
        dim3 pgrid(a.nblocks < persist_blocks ? a.nblocks : persist_blocks);
#define C3P_LAUNCH(DT)                                                                        \
    do {                                                                                      \
        if (p.bn == 128) {                                                                    \
            if (p.tw == 32) conv3x3_persist_kernel<DT, 128, 32><<<pgrid, block, 0, s>>>(a);   \
            else conv3x3_persist_kernel<DT, 128, 16><<<pgrid, block, 0, s>>>(a);              \
        } else {                                                                              \
            if (p.tw == 32) conv3x3_persist_kernel<DT, 64, 32><<<pgrid, block, 0, s>>>(a);    \
            else conv3x3_persist_kernel<DT, 64, 16><<<pgrid, block, 0, s>>>(a);               \
        }                                                                                     \
    } while (0)
        if (dtype == GS_F16) C3P_LAUNCH(GS_F16);
        else C3P_LAUNCH(GS_BF16);
#undef C3P_LAUNCH
        GS_CHECK_LAUNCH("gs_conv3x3");
        return GS_OK;
    }
#define C3_LAUNCH(DT)                                                                    \
    do {                                                                                 \
        if (p.bn == 128) {                                                               \
            if (p.tw == 32) conv3x3_halo_kernel<DT, 128, 32><<<grid, block, 0, s>>>(a);  \
            else conv3x3_halo_kernel<DT, 128, 16><<<grid, block, 0, s>>>(a);             \
        } else {                                                                         \
            if (p.tw == 32) conv3x3_halo_kernel<DT, 64, 32><<<grid, block, 0, s>>>(a);   \
            else conv3x3_halo_kernel<DT, 64, 16><<<grid, block, 0, s>>>(a);              \
        }                                                                                \
    } while (0)
    if (dtype == GS_F16) C3_LAUNCH(GS_F16);
    else C3_LAUNCH(GS_BF16);
#undef C3_LAUNCH
    GS_CHECK_LAUNCH("gs_conv3x3");
    return GS_OK;
}

extern "C" int gs_conv3x3(const void* x, const void* w, void* y, const float* bias, float* bn_partials, int N, int H,
                          int W, int Cin, int in_pix_stride, int in_coff, int Cout, int out_pix_stride, int out_coff,
                          const int32_t* tap_dy, const int32_t* tap_dx, int act, int dtype, void* stream) {
    return conv3x3_launch(x, w, y, bias, bn_partials, N, H, W, Cin, in_pix_stride, in_coff, Cout, out_pix_stride, out_coff,
                          tap_dy, tap_dx, act, dtype, stream, 1, 1, nullptr);
}

// Precise-mode 3x3 convolution (U-Net forward with hi/lo 16-bit pairs): x holds `in_wrap` channels per
// pixel -- the [hi | lo] planes of the activation, or one plane -- and the weight pack [9][Cout][K] is the matching
// concatenation of segments ([w_hi | w_hi | w_lo] against [x_hi | x_lo | x_hi]): K chunk c reads input chunk c mod in_wrap.
// All products are accumulated in the same fp32 MFMA accumulators; the result leaves as a pair y_hi = 16-bit(v),
// y_lo = 16-bit(v - y_hi) (same stride / offset for both).  BatchNorm partials as gs_conv3x3.
extern "C" int gs_conv3x3_precise(const void* x, const void* w, void* y_hi, void* y_lo, const float* bias, float* bn_partials,
                                  int N, int H, int W, int K, int in_pix_stride, int in_coff, int in_wrap, int Cout,
                                  int out_pix_stride, int out_coff, const int32_t* tap_dy, const int32_t* tap_dx, int act,
                                  int dtype, void* stream) {
    GS_CHECK_ARG(y_lo != nullptr, "gs_conv3x3_precise: y_lo is NULL");
    return conv3x3_launch(x, w, y_hi, bias, bn_partials, N, H, W, K, in_pix_stride, in_coff, Cout, out_pix_stride, out_coff,
                          tap_dy, tap_dx, act, dtype, stream, 1, 1, nullptr, y_lo, in_wrap);
}

// 3x3x3 / stride 1 / pad 1 Conv3d (GenSeg-3D/UNet3D/unet3d.py:28-31,69-71) and its data gradient on the same halo-reuse
// kernel: the volumes are NB*D depth slices [NB*D, H, W, *]; per output patch the K loop runs over (depth tap, channel
// chunk) stages, stage (dz, c) staging the halo of slice d + tap_dz[dz] (zeros outside the volume) and the nine
// weights of slot dz*9 .. dz*9+8 of the [27][Cout][Cin] pack.  Requires Cin % 8 == 0, Cout % 8 == 0 (a tail channel chunk is zero padded).
extern "C" int gs_conv3d_3x3x3_mtiles(int NB, int D, int H, int W, int Cout) {
    return gs_conv3x3_mtiles(NB * D, H, W, Cout);
}

// "q" stage of the pair forward (common.hpp): x holds the hi plane (Cin 16-bit channels) followed by the q plane
// (per 32 channels 64 bytes: lo8 | hi8) at in_coff of a pixel of in_pix_stride 16-bit elements; w = gs_pack_weight_q8 pack
// [taps][Cout][4*Cin bytes], wexp = its per-cout exponents.  The kernel runs x_hi.w_hi on the 16-bit MFMA and x_lo.w_hi + x_hi.w_lo as
// one block-scaled e4m3 MFMA segment (2x the MFMA work of the plain conv instead of 3x).  LDS-DMA kernel only: gs_conv3x3_q8_ok().
extern "C" int gs_conv3x3_q8_ok(int W, int Cin, int Cout) {
    return (c3_dma_shape_ok(W, 2 * Cin, Cout) && Cin % 64 == 0) ? 1 : 0;
}
extern "C" int gs_conv3x3_q8(const void* x, const void* w, const int32_t* wexp, void* y_hi, void* y_lo, float* bn_partials, int N,
                             int H, int W, int Cin, int in_pix_stride, int in_coff, int Cout, int out_pix_stride, int out_coff,
                             int dtype, void* stream) {
    GS_CHECK_ARG(y_lo != nullptr && wexp != nullptr, "gs_conv3x3_q8: y_lo / wexp is NULL");
    GS_CHECK_ARG(gs_conv3x3_q8_ok(W, Cin, Cout), "gs_conv3x3_q8: shape outside the LDS-DMA kernel (gs_conv3x3_q8_ok)");
    int32_t dy[9], dx[9];
    for (int i = 0; i < 9; ++i) { dy[i] = i / 3 - 1; dx[i] = i % 3 - 1; }
    return conv3x3_launch(x, w, y_hi, nullptr, bn_partials, N, H, W, 2 * Cin, in_pix_stride, in_coff, Cout, out_pix_stride, out_coff,
                          dy, dx, GS_ACT_NONE, dtype, stream, 1, 1, nullptr, y_lo, 2 * Cin, wexp);
}
extern "C" int gs_conv3d_3x3x3_q8(const void* x, const void* w, const int32_t* wexp, void* y_hi, void* y_lo, float* bn_partials,
                                  int NB, int D, int H, int W, int Cin, int in_pix_stride, int in_coff, int Cout, int out_pix_stride,
                                  int out_coff, int dtype, void* stream) {
    GS_CHECK_ARG(y_lo != nullptr && wexp != nullptr && NB > 0 && D > 0, "gs_conv3d_3x3x3_q8: bad arguments");
    GS_CHECK_ARG(gs_conv3x3_q8_ok(W, Cin, Cout), "gs_conv3d_3x3x3_q8: shape outside the LDS-DMA kernel (gs_conv3x3_q8_ok)");
    int32_t dy[9], dx[9], dz[3] = {-1, 0, 1};
    for (int i = 0; i < 9; ++i) { dy[i] = i / 3 - 1; dx[i] = i % 3 - 1; }
    return conv3x3_launch(x, w, y_hi, nullptr, bn_partials, NB * D, H, W, 2 * Cin, in_pix_stride, in_coff, Cout, out_pix_stride,
                          out_coff, dy, dx, GS_ACT_NONE, dtype, stream, D, 3, dz, y_lo, 2 * Cin, wexp);
}

// Conv3d 3x3x3 of the pair forward (UNet3D, BASELINE config 5): as gs_conv3x3_precise -- K = the stage's concatenation of segments
// over the `in_wrap` channels of the [hi | lo] input, w = [27][Cout][K] segment pack, result as the pair y_hi / y_lo.
extern "C" int gs_conv3d_3x3x3_precise(const void* x, const void* w, void* y_hi, void* y_lo, const float* bias, float* bn_partials,
                                       int NB, int D, int H, int W, int K, int in_pix_stride, int in_coff, int in_wrap, int Cout,
                                       int out_pix_stride, int out_coff, const int32_t* tap_dz, const int32_t* tap_dy,
                                       const int32_t* tap_dx, int act, int dtype, void* stream) {
    GS_CHECK_ARG(y_lo != nullptr, "gs_conv3d_3x3x3_precise: y_lo is NULL");
    GS_CHECK_ARG(NB > 0 && D > 0 && tap_dz, "gs_conv3d_3x3x3_precise: bad depth arguments");
    for (int i = 0; i < 3; ++i) GS_CHECK_ARG(tap_dz[i] >= -1 && tap_dz[i] <= 1, "gs_conv3d_3x3x3_precise: depth tap offsets must be in [-1,1]");
    GS_CHECK_ARG(c3_variant_get() == 2, "gs_conv3d_3x3x3_precise: needs the big-K-step kernel (GSSEG_C3=2)");
    return conv3x3_launch(x, w, y_hi, bias, bn_partials, NB * D, H, W, K, in_pix_stride, in_coff, Cout, out_pix_stride, out_coff,
                          tap_dy, tap_dx, act, dtype, stream, D, 3, tap_dz, y_lo, in_wrap);
}

// The same with the wrapped part of K continuing at input channel in_wrap_to (a multiple of 64) instead of channel 0: a decoder-entry conv
// whose concat input is [up_h res_h | res_l] runs its w_lo segment over the residual channels only -- K = [hi (all) | res_l] then res_h.
extern "C" int gs_conv3d_3x3x3_precise_to(const void* x, const void* w, void* y_hi, void* y_lo, const float* bias, float* bn_partials,
                                          int NB, int D, int H, int W, int K, int in_pix_stride, int in_coff, int in_wrap, int in_wrap_to,
                                          int Cout, int out_pix_stride, int out_coff, const int32_t* tap_dz, const int32_t* tap_dy,
                                          const int32_t* tap_dx, int act, int dtype, void* stream) {
    GS_CHECK_ARG(y_lo != nullptr, "gs_conv3d_3x3x3_precise_to: y_lo is NULL");
    GS_CHECK_ARG(NB > 0 && D > 0 && tap_dz, "gs_conv3d_3x3x3_precise_to: bad depth arguments");
    for (int i = 0; i < 3; ++i) GS_CHECK_ARG(tap_dz[i] >= -1 && tap_dz[i] <= 1, "gs_conv3d_3x3x3_precise_to: depth tap offsets must be in [-1,1]");
    GS_CHECK_ARG(c3_variant_get() == 2, "gs_conv3d_3x3x3_precise_to: needs the big-K-step kernel (GSSEG_C3=2)");
    return conv3x3_launch(x, w, y_hi, bias, bn_partials, NB * D, H, W, K, in_pix_stride, in_coff, Cout, out_pix_stride, out_coff,
                          tap_dy, tap_dx, act, dtype, stream, D, 3, tap_dz, y_lo, in_wrap, nullptr, in_wrap_to);
}

extern "C" int gs_conv3d_3x3x3(const void* x, const void* w, void* y, const float* bias, float* bn_partials, int NB, int D,
                               int H, int W, int Cin, int in_pix_stride, int in_coff, int Cout, int out_pix_stride,
                               int out_coff, const int32_t* tap_dz, const int32_t* tap_dy, const int32_t* tap_dx, int act,
                               int dtype, void* stream) {
    GS_CHECK_ARG(NB > 0 && D > 0 && tap_dz, "gs_conv3d_3x3x3: bad depth arguments");
    for (int i = 0; i < 3; ++i) GS_CHECK_ARG(tap_dz[i] >= -1 && tap_dz[i] <= 1, "gs_conv3d_3x3x3: depth tap offsets must be in [-1,1]");
    GS_CHECK_ARG(c3_variant_get() == 2, "gs_conv3d_3x3x3: needs the big-K-step kernel (GSSEG_C3=2)");
    return conv3x3_launch(x, w, y, bias, bn_partials, NB * D, H, W, Cin, in_pix_stride, in_coff, Cout, out_pix_stride,
                          out_coff, tap_dy, tap_dx, act, dtype, stream, D, 3, tap_dz);
}
