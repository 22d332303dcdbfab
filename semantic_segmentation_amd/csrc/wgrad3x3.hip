// Weight gradient of the 3x3 / stride-1 / pad-1 convolution with LDS HALO REUSE.
//
//   dW[tap][co][ci] += sum_pixels dY[p][co] * X[p + tap][ci]
//
// The generic wgrad (igemm.hip) stages X once per tap (and dY once per group of taps): at the 256x256 /
// 64-channel end of the U-Net it fetches 2.8 GB per launch for 0.54 GB of algorithmic traffic (rocprofv3 PMC,
// profiles/r01_v2_pmc_traffic.json).  Here a block owns ONE (64 co x 64 ci) channel-block pair and a range of
// spatial patches (split-K over patches): per patch the dY tile [128 px][64 co] and the X halo
// [(TH+2)(TW+2) px][64 ci] are staged in LDS once and ALL NINE taps are accumulated from them
// (9 taps x 4 MFMA 32x32 tiles = 36 accumulator tiles per block, 9 per wave: waves split co-half x ci-half).
// Both MFMA operands are K(=pixel)-minor, read with the transposing LDS read ds_read_b64_tr_b16 from
// [pixel][channel] images (192-B rows: conflict free).  The next patch is prefetched into registers while the
// current one is multiplied.  One fp32 atomic pass per block at the end.
#include <stdlib.h>

#include "common.hpp"

namespace {

struct W3Args {
    const unsigned short* x;
    const unsigned short* dy;
    float* dw;                  // [9][Cout][Cin]
    int N, H, W, Cin, in_stride, in_coff, Cout, out_stride, out_coff;
    int tiles_x, tiles_y, npatches, ncob, ncib, ksplit, pps;   // pps = patches per split
    int D, dz;                  // 3-D: images are depth slices; X is read from slice n + dz (zeros outside the volume)
    int64_t slab_stride;        // > 0: split-K part ks STORES its tile to dw + ks*slab_stride (no atomics; reduced later)
};

constexpr int W3_LDR = 96;      // 64 channels + 32 pad elements = 192-byte rows
constexpr int W3_BM = 128;      // pixels per patch

template <int DT, int TW>
__global__ __launch_bounds__(256, 2) void wgrad3x3_kernel(const W3Args a) {
    typedef typename Elem<DT>::V8 V8;
    constexpr int TH = W3_BM / TW;
    constexpr int HWD = TW + 2, HHT = TH + 2, HP = HWD * HHT;
    constexpr int DY_EL = W3_BM * W3_LDR, H_EL = HP * W3_LDR;
    constexpr int HCH = (HP * 8 + 255) / 256;
    constexpr int TWS = (TW == 32) ? 5 : 4;
    constexpr unsigned OOB = 0xFFFFFFFFu;
    __shared__ __attribute__((aligned(16))) unsigned short smem[DY_EL + H_EL];
    unsigned short* dyt = smem;
    unsigned short* halo = smem + DY_EL;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware: the (co, ci) channel-block pairs that share one patch range read the same dY / X tiles, so
    // they get consecutive logical ids inside ONE XCD (blocks with equal id % 8 share an XCD and its L2)
    int bid;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int cob = bid % a.ncob; bid /= a.ncob;
    const int cib = bid % a.ncib; bid /= a.ncib;
    const int ks = bid;
    const int co0 = cob * 64, ci0 = cib * 64;
    const int p_begin = ks * a.pps;
    const int p_end = min(a.npatches, p_begin + a.pps);
    if (p_begin >= p_end) return;

    const unsigned x_img_bytes = (unsigned)a.H * a.W * a.in_stride * 2u;
    const unsigned dy_img_bytes = (unsigned)a.H * a.W * a.out_stride * 2u;
    const int chunk = t & 7, row0 = t >> 3;
    const bool ci_ok = ci0 + chunk * 8 < a.Cin, co_ok = co0 + chunk * 8 < a.Cout;

    uint4 rd[4], rh[HCH];
    // prefetch of the next patch, split so that the loads can be spread over the MFMA steps of the current one
    struct Pf { int y0, x0; __amdgpu_buffer_rsrc_t rx, rdy; unsigned kill, killx; };
    auto prep_patch = [&](int patch, bool live) __attribute__((always_inline)) {
        Pf f;
        const int tx = patch % a.tiles_x;
        const int r = patch / a.tiles_x;
        const int ty = r % a.tiles_y, n = r / a.tiles_y;
        f.y0 = ty * TH; f.x0 = tx * TW;
        const int d = a.D > 1 ? n % a.D : 0;
        const bool xin = (unsigned)(d + a.dz) < (unsigned)a.D;
        f.rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (int64_t)(xin ? n + a.dz : n) * a.H * a.W * a.in_stride), 0, x_img_bytes, 0x00020000);
        f.rdy = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dy + (int64_t)n * a.H * a.W * a.out_stride), 0, dy_img_bytes, 0x00020000);
        f.kill = live ? 0u : OOB;                  // no next patch: every offset out of range (loads return zeros)
        f.killx = (live && xin) ? 0u : OOB;
        return f;
    };
    auto issue_load = [&](const Pf& f, int j) __attribute__((always_inline)) {      // j compile-time, 0 .. 4+HCH-1
        if (j < 4) {
            const int p = row0 + 32 * j;
            const int gy = f.y0 + (p >> TWS), gx = f.x0 + (p & (TW - 1));
            const bool ok = co_ok && gy < a.H && gx < a.W;
            const unsigned off = ok ? (unsigned)(((gy * a.W + gx) * a.out_stride + a.out_coff + co0 + chunk * 8) * 2) : OOB;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(f.rdy, off | f.kill, 0, 0);
            rd[j < 4 ? j : 0] = make_uint4(v[0], v[1], v[2], v[3]);
        } else if (j - 4 < HCH) {
            const int jj = j - 4 < HCH ? j - 4 : 0;
            const int hp = row0 + 32 * jj;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const int gy = f.y0 + hy - 1, gx = f.x0 + hx - 1;
            const bool ok = ci_ok && hp < HP && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            const unsigned off = ok ? (unsigned)(((gy * a.W + gx) * a.in_stride + a.in_coff + ci0 + chunk * 8) * 2) : OOB;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(f.rx, off | f.killx, 0, 0);
            rh[jj] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    };
    auto load_patch = [&](int patch) __attribute__((always_inline)) {
        const Pf f = prep_patch(patch, true);
#pragma unroll
        for (int j = 0; j < 4 + HCH; ++j) issue_load(f, j);
    };
    auto store_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<uint4*>(dyt + (row0 + 32 * j) * W3_LDR + chunk * 8) = rd[j];
#pragma unroll
        for (int j = 0; j < HCH; ++j)
            if (row0 + 32 * j < HP) *reinterpret_cast<uint4*>(halo + (row0 + 32 * j) * W3_LDR + chunk * 8) = rh[j];
    };

    f32x16 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // transposed-read lane addressing: 16-lane group G covers channels 16*(G&1).. and k rows 8*(G>>1)+4r+q
    const int G = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int krow = 8 * (G >> 1) + q;
    const int chn = 16 * (G & 1) + 4 * pp;
    const LDS_AS unsigned short* lds = (const LDS_AS unsigned short*)smem;
    const LDS_AS unsigned short* a_base = lds + krow * W3_LDR + wm * 32 + chn;
    const LDS_AS unsigned short* b_base = lds + DY_EL + krow * W3_LDR + wn * 32 + chn;

    load_patch(p_begin);
    for (int patch = p_begin; patch < p_end; ++patch) {
        __syncthreads();                 // previous patch fully consumed
        store_patch();
        __syncthreads();
        const bool more = patch + 1 < p_end;
        const Pf pf = prep_patch(more ? patch + 1 : patch, more);
        static_assert(4 + HCH <= 2 * (W3_BM / 16), "prefetch loads must fit two per k16 step");
#pragma unroll
        for (int k16 = 0; k16 < W3_BM / 16; ++k16) {
            const int pb = k16 * 16;
            const int py = pb >> TWS, px0 = pb & (TW - 1);
            const V8 af = tr_read8<DT>(a_base + pb * W3_LDR, a_base + (pb + 4) * W3_LDR);
            issue_load(pf, 2 * k16);               // two 16-byte loads of the next patch per 9 MFMAs
            issue_load(pf, 2 * k16 + 1);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int hrow = (py + tap / 3) * HWD + px0 + tap % 3;      // (py+1+dy)*HWD + px0+1+dx
                const V8 bf = tr_read8<DT>(b_base + hrow * W3_LDR, b_base + (hrow + 4) * W3_LDR);
                acc[tap] = Elem<DT>::mfma32(af, bf, acc[tap]);
            }
        }
    }

    const int l31 = lane & 31, h = lane >> 5;
    const int ci = ci0 + wn * 32 + l31;
    if (ci < a.Cin) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (co < a.Cout) {
                    float* q = a.dw + ((int64_t)tap * a.Cout + co) * a.Cin + ci;
                    if (a.slab_stride > 0) q[(int64_t)ks * a.slab_stride] = acc[tap][r];
                    else atomicAdd(q, acc[tap][r]);
                }
            }
    }
}


// ---------------------------------------------------------------------------------------------------
// LDS-DMA form (W >= 24, 2-D): one 8-wave block per CU, 256-pixel patches (8 x 32).  The dY tile [256 px][64 co] and the X
// halo [10 x 34 px][64 ci] of the NEXT patch are written into the other LDS buffer by `buffer_load_dwordx4 ... lds` (8 rows
// x 128 B per wave-instruction) while the current patch is multiplied: no staging registers, no ds_write pass, one barrier
// per patch.  Waves 0-3 / 4-7 (split as co-half x ci-half like the 4-wave kernel) take the left / right 16 pixels of every
// patch row -- a split of K inside the block; the two partial tiles are added in LDS in a fixed order at the end.
//   LDS image: 128-byte rows without padding (the DMA destination is lane-linear); the two 64-byte halves of row r are
//   swapped when (r >> 1) & 1: the four rows of a transposing read's 32-lane group then cover four distinct 64-byte bank
//   windows for every base row.  The swap is applied on the source side (lane l of a piece fetches slot (l&7) ^ ((l>>4&1)<<2)).
// ---------------------------------------------------------------------------------------------------
// One piece: lane l's 16 bytes at buffer offset voff land at dst + 16 l (dst wave-uniform).  Issued as inline assembly: with
// the builtin hipcc (ROCm 7.2) puts an `s_waitcnt vmcnt(0)` in front of every transposing LDS read that follows a piece (it
// cannot tell the read from the DMA's destination), which serialises the K loop.  The pieces are therefore invisible to the
// compiler's counters: the kernel waits for them itself (`s_waitcnt vmcnt(0)` + barrier before a buffer is read).  Extra
// outstanding operations only ever make a compiler-computed vmcnt(N) wait longer (completion is counted in issue order).
__device__ __forceinline__ void w3_dma_piece16(const __amdgpu_buffer_rsrc_t& rs, unsigned char* dst, unsigned voff) {
    const unsigned lds_addr = (unsigned)(size_t)(LDS_AS void*)dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rs) : "memory");
}
__device__ __forceinline__ void w3_opaque(unsigned& x) { asm volatile("" : "+v"(x)); }

template <int DT>
__global__ __launch_bounds__(512, 1) void wgrad3x3_dma_kernel(const W3Args a) {
    typedef typename Elem<DT>::V8 V8;
    constexpr int TW = 32, TH = 8, TWS = 5, BM = 256;
    constexpr int HWD = TW + 2, HHT = TH + 2, HP = HWD * HHT;       // 340 halo pixels
    static_assert(BM / 8 == 4 * 8, "32 dY pieces: four per wave");
    constexpr int HJ = 6;                                            // halo piece slots per wave: 48 slots, 43 real + spare
    constexpr int HPC = (HP + 7) / 8;                                // 43
    constexpr int DY_B = BM * 128, HALO_B = (HPC + 1) * 1024;        // one spare piece takes the surplus slots
    constexpr int HALO_OFF = 2 * DY_B;
    constexpr unsigned VOOB = 0x80000000u;
    static_assert(HPC + 1 <= HJ * 8 && 2 * (DY_B + HALO_B) <= 160 * 1024, "LDS budget");
    static_assert(4 * 9 * 16 * 64 * 4 <= 2 * (DY_B + HALO_B), "final in-block reduction overlays the stage buffers");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * (DY_B + HALO_B)];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = wave >> 2;                 // K group: pixel columns 16*grp .. 16*grp+15 of every patch row
    const int wm = (wave >> 1) & 1, wn = wave & 1;
    int bid;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int cob = bid % a.ncob; bid /= a.ncob;
    const int cib = bid % a.ncib; bid /= a.ncib;
    const int ks = bid;
    const int co0 = cob * 64, ci0 = cib * 64;
    const int p_begin = ks * a.pps;
    const int p_end = min(a.npatches, p_begin + a.pps);
    if (p_begin >= p_end) return;

    const unsigned x_img_bytes = (unsigned)a.H * a.W * a.in_stride * 2u;
    const unsigned dy_img_bytes = (unsigned)a.H * a.W * a.out_stride * 2u;
    // DMA side: lane l fills physical 16-byte slot l & 7 of row l >> 3 of its piece
    const int drow = lane >> 3;
    const int dls = (lane & 7) ^ (((lane >> 4) & 1) << 2);
    const bool ci_ok = ci0 + dls * 8 < a.Cin, co_ok = co0 + dls * 8 < a.Cout;
    unsigned dv[4], hv[HJ];
    // Per patch only the origin changes: each slot keeps its byte offset RELATIVE to the patch origin (computed once; the
    // per-patch part is a scalar base + two compares + a select per slot -- recomputing (gy * W + gx) * stride per slot per
    // patch cost ~150 instructions with exec-masked branches at the top of every patch, with no MFMA in flight)
    unsigned rel_d[4], rel_h[HJ];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = (wave + 8 * j) * 8 + drow;
        rel_d[j] = (unsigned)((((p >> TWS) * a.W + (p & (TW - 1))) * a.out_stride + a.out_coff + co0 + dls * 8) * 2);
    }
#pragma unroll
    for (int j = 0; j < HJ; ++j) {
        const int hp = (wave + 8 * j) * 8 + drow;
        const int hy = hp / HWD, hx = hp - hy * HWD;
        rel_h[j] = (unsigned)((((hy - 1) * a.W + (hx - 1)) * a.in_stride + a.in_coff + ci0 + dls * 8) * 2);   // may wrap: added mod 2^32
    }
    struct Pf { __amdgpu_buffer_rsrc_t rx, rdy; };
    // patch position, stepped with carries (tx fastest)
    int pt_x, pt_y, pt_n;
    {
        pt_x = p_begin % a.tiles_x;
        const int r = p_begin / a.tiles_x;
        pt_y = r % a.tiles_y; pt_n = r / a.tiles_y;
    }
    auto prep_patch = [&](bool live) __attribute__((always_inline)) {      // prepares (pt_x, pt_y, pt_n), then steps to the next patch
        Pf f;
        const int n = pt_n, y0 = pt_y * TH, x0 = pt_x * TW;
        const int d = a.D > 1 ? n % a.D : 0;                       // Conv3d: X comes from slice n + dz (zeros outside the volume)
        const bool xin = (unsigned)(d + a.dz) < (unsigned)a.D;
        f.rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (int64_t)(xin ? n + a.dz : n) * a.H * a.W * a.in_stride), 0, x_img_bytes, 0x00020000);
        f.rdy = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dy + (int64_t)n * a.H * a.W * a.out_stride), 0, dy_img_bytes, 0x00020000);
        const unsigned base_d = (unsigned)((y0 * a.W + x0) * a.out_stride * 2), base_h = (unsigned)((y0 * a.W + x0) * a.in_stride * 2);
        const int ylim = a.H - y0, xlim = a.W - x0;                 // rows / columns of the patch inside the image
        const bool dlive = live && co_ok, hlive = live && xin && ci_ok;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = (wave + 8 * j) * 8 + drow;
            const bool ok = dlive && (p >> TWS) < ylim && (p & (TW - 1)) < xlim;
            const unsigned v = base_d + rel_d[j];
            dv[j] = ok ? v : VOOB;
        }
#pragma unroll
        for (int j = 0; j < HJ; ++j) {
            const int hp = (wave + 8 * j) * 8 + drow;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const bool ok = hlive && hp < HP && (unsigned)(hy + y0 - 1) < (unsigned)a.H && (unsigned)(hx + x0 - 1) < (unsigned)a.W;
            const unsigned v = base_h + rel_h[j];
            hv[j] = ok ? v : VOOB;
        }
        int c = ++pt_x == a.tiles_x ? 1 : 0;
        pt_x -= c * a.tiles_x;
        pt_y += c; c = pt_y == a.tiles_y ? 1 : 0;
        pt_y -= c * a.tiles_y;
        pt_n += c;
        return f;
    };
    // piece slot k (compile-time, 0 .. 4+HJ-1) of the prepared patch into buffer nb
    auto issue_piece = [&](const Pf& f, int k, unsigned nb) __attribute__((always_inline)) {
        if (k < 4) {
            w3_dma_piece16(f.rdy, smem + nb * DY_B + (unsigned)(wave + 8 * k) * 1024u, dv[k < 4 ? k : 0]);
        } else if (k - 4 < HJ) {
            const int j = k - 4 < HJ ? k - 4 : 0;
            const int q = wave + 8 * j;
            w3_dma_piece16(f.rx, smem + HALO_OFF + nb * HALO_B + (unsigned)(q < HPC ? q : HPC) * 1024u, hv[j]);
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // transposed-read lane addressing (as wgrad3x3_kernel) on the swizzled image, in bytes
    const int G = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int krow = 8 * (G >> 1) + q;
    const int chn = 16 * (G & 1) + 4 * pp;
    unsigned a_base0 = (unsigned)((grp * 16 + krow) * 128 + (((wm ^ ((q >> 1) & 1)) * 32 + chn) * 2));
    unsigned b_base0[4];                       // by (constant row offset & 3)
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4)
        b_base0[c4] = (unsigned)(HALO_OFF + (grp * 16 + krow) * 128 + (((wn ^ (((c4 + q) >> 1) & 1)) * 32 + chn) * 2));

    {
        const Pf f0 = prep_patch(true);
#pragma unroll
        for (int k = 0; k < 4 + HJ; ++k) issue_piece(f0, k, 0u);
    }
    for (int patch = p_begin; patch < p_end; ++patch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of the patch have landed
        __builtin_amdgcn_s_barrier();                              // ... everybody's; the other buffer is free
        asm volatile("" ::: "memory");
        const unsigned buf = (unsigned)(patch - p_begin) & 1u;
        const bool more = patch + 1 < p_end;
        const Pf pf = prep_patch(more);        // (behind the range: empty pieces, the position is not used again)
        unsigned ab = a_base0 + buf * DY_B, bb[4];
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) { bb[c4] = b_base0[c4] + buf * HALO_B; w3_opaque(bb[c4]); }
        w3_opaque(ab);
        const LDS_AS unsigned char* lds = (const LDS_AS unsigned char*)smem;
        // 72 steps (patch row s = this group's 16 pixels, tap); the X fragment of step i+PD (and the dY fragment of the next
        // row) is read while the MFMA of step i runs: left to itself hipcc reads each fragment right in front of its MFMA
        // behind lgkmcnt(0) and the LDS latency is exposed 72 times per patch
        constexpr int PD = 3;                                      // X fragments in flight ahead of their MFMA
        V8 afr[2], bfr[PD + 1];
        auto load_a = [&](int s, V8& dst) __attribute__((always_inline)) {
            dst = tr_read8<DT>((const LDS_AS unsigned short*)(lds + ab + s * 32 * 128),
                               (const LDS_AS unsigned short*)(lds + ab + (s * 32 + 4) * 128));
        };
        auto load_b = [&](int step, V8& dst) __attribute__((always_inline)) {
            const int s = step / 9, tap = step - 9 * s;
            const int C = (s + tap / 3) * HWD + tap % 3;           // halo row of pixel column 0 of this group, before + 16*grp
            dst = tr_read8<DT>((const LDS_AS unsigned short*)(lds + bb[C & 3] + C * 128),
                               (const LDS_AS unsigned short*)(lds + bb[C & 3] + (C + 4) * 128));
        };
        load_a(0, afr[0]);
#pragma unroll
        for (int i = 0; i < PD; ++i) load_b(i, bfr[i]);
#pragma unroll
        for (int step = 0; step < 72; ++step) {
            const int s = step / 9, tap = step - 9 * s;
            if (step + PD < 72) load_b(step + PD, bfr[(step + PD) % (PD + 1)]);
            if (tap == 4 && s + 1 < 8) load_a(s + 1, afr[(s + 1) & 1]);
#if !defined(GS_W3_ABLATE) || GS_W3_ABLATE != 1          // ablation 1: no DMA traffic (stale LDS), 2: no MFMAs / fragment reads
            // the ten pieces of the next patch go out in the first rows: spread over the whole patch (one per row) the last
            // ones had a fraction of the patch time to land before the hand-over -- the full-resolution layers stream
            // both operands from HBM (64->64 @256^2: 178 -> 163 us, 128->64 @256^2: 300 -> 284 us; all shapes 2.51 -> 2.45 ms)
            // (all ten in the first two rows: the same)
            if (s < 4 && (tap == 1 || tap == 4 || tap == 7) && 3 * s + tap / 3 < 4 + HJ) issue_piece(pf, 3 * s + tap / 3, buf ^ 1u);
#endif
#if !defined(GS_W3_ABLATE) || GS_W3_ABLATE != 2
            acc[tap] = Elem<DT>::mfma32(afr[s & 1], bfr[step % (PD + 1)], acc[tap]);
#endif
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // the MFMA, then the reads of the next step in its shadow
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // in-block K reduction: group 1 hands its nine tiles over through LDS, group 0 adds (fixed order) and stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the (empty) pieces behind the last patch
    __syncthreads();
    float* xch = reinterpret_cast<float*>(smem) + (wave & 3) * (9 * 16 * 64) + lane;
    if (grp == 1) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int r = 0; r < 16; ++r) xch[(tap * 16 + r) * 64] = acc[tap][r];
    }
    __syncthreads();
    if (grp == 1) return;
    const int l31 = lane & 31, h = lane >> 5;
    const int ci = ci0 + wn * 32 + l31;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[tap][r] + xch[(tap * 16 + r) * 64];
            const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (ci < a.Cin && co < a.Cout) {
                float* qd = a.dw + ((int64_t)tap * a.Cout + co) * a.Cin + ci;
                if (a.slab_stride > 0) qd[(int64_t)ks * a.slab_stride] = v;
                else atomicAdd(qd, v);
            }
        }
}

// ---------------------------------------------------------------------------------------------------
// 128-cout form of the LDS-DMA kernel (layers with Cout % 128 == 0): a block owns (128 co x 64 ci) and walks 128-pixel patches
// (4 x 32).  The 256-pixel / 64-cout form above moves 75 KB through the CU's DMA path per 576 block-MFMAs (130 B per MFMA) and
// is bound by that stream and its overlap with the matrix pipe (ablation in DESIGN.md: MFMAs alone 1.87 ms, DMA alone 1.55 ms,
// both 2.45 ms per step); here a patch is 32 KB of dY (two 64-cout images) + 26 KB of X halo = 58 KB for the same 576 MFMAs
// (101 B per MFMA), the X halo is read once per 128 couts instead of once per 64, and there is no K split inside the block
// (waves = 4 cout tiles x 2 ci halves, every wave sees all 128 pixels: 4 rows x 2 column halves x 9 taps = 72 MFMAs per
// patch), so the final in-block reduction through LDS is gone as well.
// ---------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(512, 1) void wgrad3x3_dma128_kernel(const W3Args a) {
    typedef typename Elem<DT>::V8 V8;
    constexpr int TW = 32, TH = 4, TWS = 5, BM = 128;
    constexpr int HWD = TW + 2, HHT = TH + 2, HP = HWD * HHT;       // 204 halo pixels
    constexpr int DJ = 4;                                            // dY piece slots per wave: 2 images x 16 pieces = 32
    constexpr int HPC = (HP + 7) / 8;                                // 26 halo pieces
    constexpr int HJ = 4;                                            // halo piece slots per wave: 32 slots, 26 real + spare
    constexpr int IMG_B = BM * 128;                                  // one 64-cout dY image: 16 KB
    constexpr int DY_B = 2 * IMG_B, HALO_B = (HPC + 1) * 1024;       // one spare piece takes the surplus slots
    constexpr int HALO_OFF = 2 * DY_B;
    constexpr unsigned VOOB = 0x80000000u;
    static_assert(HPC + 1 <= HJ * 8 && 2 * (DY_B + HALO_B) <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * (DY_B + HALO_B)];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wmm = wave >> 1, wn = wave & 1;  // cout tile of 32 (0..3), ci half of 32
    int bid;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int cob = bid % a.ncob; bid /= a.ncob;
    const int cib = bid % a.ncib; bid /= a.ncib;
    const int ks = bid;
    const int co0 = cob * 128, ci0 = cib * 64;
    const int p_begin = ks * a.pps;
    const int p_end = min(a.npatches, p_begin + a.pps);
    if (p_begin >= p_end) return;

    const unsigned x_img_bytes = (unsigned)a.H * a.W * a.in_stride * 2u;
    const unsigned dy_img_bytes = (unsigned)a.H * a.W * a.out_stride * 2u;
    // DMA side: lane l fills physical 16-byte slot l & 7 of row l >> 3 of its piece
    const int drow = lane >> 3;
    const int dls = (lane & 7) ^ (((lane >> 4) & 1) << 2);
    const bool ci_ok = ci0 + dls * 8 < a.Cin;
    unsigned dv[DJ], hv[HJ];
    unsigned rel_d[DJ], rel_h[HJ];
    bool d_ok[DJ];
#pragma unroll
    for (int j = 0; j < DJ; ++j) {
        const int q = wave + 8 * j;                                  // piece: image q >> 4, rows 8 * (q & 15) ..
        const int p = (q & 15) * 8 + drow;
        const int co = co0 + (q >> 4) * 64 + dls * 8;
        d_ok[j] = co < a.Cout;
        rel_d[j] = (unsigned)((((p >> TWS) * a.W + (p & (TW - 1))) * a.out_stride + a.out_coff + co) * 2);
    }
#pragma unroll
    for (int j = 0; j < HJ; ++j) {
        const int hp = (wave + 8 * j) * 8 + drow;
        const int hy = hp / HWD, hx = hp - hy * HWD;
        rel_h[j] = (unsigned)((((hy - 1) * a.W + (hx - 1)) * a.in_stride + a.in_coff + ci0 + dls * 8) * 2);   // may wrap: added mod 2^32
    }
    struct Pf { __amdgpu_buffer_rsrc_t rx, rdy; };
    int pt_x, pt_y, pt_n;
    {
        pt_x = p_begin % a.tiles_x;
        const int r = p_begin / a.tiles_x;
        pt_y = r % a.tiles_y; pt_n = r / a.tiles_y;
    }
    auto prep_patch = [&](bool live) __attribute__((always_inline)) {      // prepares (pt_x, pt_y, pt_n), then steps to the next patch
        Pf f;
        const int n = pt_n, y0 = pt_y * TH, x0 = pt_x * TW;
        const int d = a.D > 1 ? n % a.D : 0;                       // Conv3d: X comes from slice n + dz (zeros outside the volume)
        const bool xin = (unsigned)(d + a.dz) < (unsigned)a.D;
        f.rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (int64_t)(xin ? n + a.dz : n) * a.H * a.W * a.in_stride), 0, x_img_bytes, 0x00020000);
        f.rdy = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dy + (int64_t)n * a.H * a.W * a.out_stride), 0, dy_img_bytes, 0x00020000);
        const unsigned base_d = (unsigned)((y0 * a.W + x0) * a.out_stride * 2), base_h = (unsigned)((y0 * a.W + x0) * a.in_stride * 2);
        const int ylim = a.H - y0, xlim = a.W - x0;                 // rows / columns of the patch inside the image
        const bool hlive = live && xin && ci_ok;
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            const int p = ((wave + 8 * j) & 15) * 8 + drow;
            const bool ok = live && d_ok[j] && (p >> TWS) < ylim && (p & (TW - 1)) < xlim;
            const unsigned v = base_d + rel_d[j];
            dv[j] = ok ? v : VOOB;
        }
#pragma unroll
        for (int j = 0; j < HJ; ++j) {
            const int hp = (wave + 8 * j) * 8 + drow;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const bool ok = hlive && hp < HP && (unsigned)(hy + y0 - 1) < (unsigned)a.H && (unsigned)(hx + x0 - 1) < (unsigned)a.W;
            const unsigned v = base_h + rel_h[j];
            hv[j] = ok ? v : VOOB;
        }
        int c = ++pt_x == a.tiles_x ? 1 : 0;
        pt_x -= c * a.tiles_x;
        pt_y += c; c = pt_y == a.tiles_y ? 1 : 0;
        pt_y -= c * a.tiles_y;
        pt_n += c;
        return f;
    };
    // piece slot k (compile-time, 0 .. DJ+HJ-1) of the prepared patch into buffer nb
    auto issue_piece = [&](const Pf& f, int k, unsigned nb) __attribute__((always_inline)) {
        if (k < DJ) {
            w3_dma_piece16(f.rdy, smem + nb * DY_B + (unsigned)(wave + 8 * k) * 1024u, dv[k < DJ ? k : 0]);
        } else if (k - DJ < HJ) {
            const int j = k - DJ < HJ ? k - DJ : 0;
            const int q = wave + 8 * j;
            w3_dma_piece16(f.rx, smem + HALO_OFF + nb * HALO_B + (unsigned)(q < HPC ? q : HPC) * 1024u, hv[j]);
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // transposed-read lane addressing on the swizzled images, in bytes
    const int G = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int krow = 8 * (G >> 1) + q;
    const int chn = 16 * (G & 1) + 4 * pp;
    unsigned a_base0 = (unsigned)((wmm >> 1) * IMG_B + krow * 128 + ((((wmm & 1) ^ ((q >> 1) & 1)) * 32 + chn) * 2));
    unsigned b_base0[4];                       // by (constant row offset & 3)
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4)
        b_base0[c4] = (unsigned)(HALO_OFF + krow * 128 + (((wn ^ (((c4 + q) >> 1) & 1)) * 32 + chn) * 2));

    {
        const Pf f0 = prep_patch(true);
#pragma unroll
        for (int k = 0; k < DJ + HJ; ++k) issue_piece(f0, k, 0u);
    }
    for (int patch = p_begin; patch < p_end; ++patch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of the patch have landed
        __builtin_amdgcn_s_barrier();                              // ... everybody's; the other buffer is free
        asm volatile("" ::: "memory");
        const unsigned buf = (unsigned)(patch - p_begin) & 1u;
        const bool more = patch + 1 < p_end;
        const Pf pf = prep_patch(more);        // (behind the range: empty pieces, the position is not used again)
        unsigned ab = a_base0 + buf * DY_B, bb[4];
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) { bb[c4] = b_base0[c4] + buf * HALO_B; w3_opaque(bb[c4]); }
        w3_opaque(ab);
        const LDS_AS unsigned char* lds = (const LDS_AS unsigned char*)smem;
        // 72 steps: (patch row s, column half g, tap); the X fragment of step i+PD (and the dY fragment of the next (s, g)) is
        // read while the MFMA of step i runs
        constexpr int PD = 3;
        V8 afr[2], bfr[PD + 1];
        auto load_a = [&](int sg, V8& dst) __attribute__((always_inline)) {       // sg = 2 * s + g: pixels 32 s + 16 g ..
            dst = tr_read8<DT>((const LDS_AS unsigned short*)(lds + ab + sg * 16 * 128),
                               (const LDS_AS unsigned short*)(lds + ab + (sg * 16 + 4) * 128));
        };
        auto load_b = [&](int step, V8& dst) __attribute__((always_inline)) {
            const int sg = step / 9, tap = step - 9 * sg;
            const int s = sg >> 1, g = sg & 1;
            const int C = (s + tap / 3) * HWD + tap % 3 + 16 * g;   // halo row of this step's pixel column 0
            dst = tr_read8<DT>((const LDS_AS unsigned short*)(lds + bb[C & 3] + C * 128),
                               (const LDS_AS unsigned short*)(lds + bb[C & 3] + (C + 4) * 128));
        };
        load_a(0, afr[0]);
#pragma unroll
        for (int i = 0; i < PD; ++i) load_b(i, bfr[i]);
#pragma unroll
        for (int step = 0; step < 72; ++step) {
            const int sg = step / 9, tap = step - 9 * sg;
            if (step + PD < 72) load_b(step + PD, bfr[(step + PD) % (PD + 1)]);
            if (tap == 4 && sg + 1 < 8) load_a(sg + 1, afr[(sg + 1) & 1]);
            // the eight pieces of the next patch go out in the first steps
            if (sg < 3 && (tap == 1 || tap == 4 || tap == 7) && 3 * sg + tap / 3 < DJ + HJ) issue_piece(pf, 3 * sg + tap / 3, buf ^ 1u);
            acc[tap] = Elem<DT>::mfma32(afr[sg & 1], bfr[step % (PD + 1)], acc[tap]);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // the MFMA, then the reads of the next step in its shadow
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the (empty) pieces behind the last patch
    const int l31 = lane & 31, h = lane >> 5;
    const int ci = ci0 + wn * 32 + l31;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wmm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (ci < a.Cin && co < a.Cout) {
                float* qd = a.dw + ((int64_t)tap * a.Cout + co) * a.Cin + ci;
                if (a.slab_stride > 0) qd[(int64_t)ks * a.slab_stride] = acc[tap][r];
                else atomicAdd(qd, acc[tap][r]);
            }
        }
}
}  // namespace

// the LDS-DMA kernel covers the layers with 32-wide patches, 2-D and Conv3d (GSSEG_W3_DMA=0 switches it off)
static bool w3_use_dma(int W) {
    static const int env = getenv("GSSEG_W3_DMA") ? atoi(getenv("GSSEG_W3_DMA")) : 1;
    return env != 0 && W >= 24;
}

// the 128-cout form of the LDS-DMA kernel (GSSEG_W3_CO128=0 switches it off)
static bool w3_use_co128(int W, int Cout) {
    static const int env = getenv("GSSEG_W3_CO128") ? atoi(getenv("GSSEG_W3_CO128")) : 1;
    return env != 0 && w3_use_dma(W) && Cout % 128 == 0;
}

static int w3_ksplit(int N, int H, int W, int Cin, int Cout, int* pps_out, int* npatches_out, bool dma) {
    const bool co128 = dma && w3_use_co128(W, Cout);
    const int tw = (W >= 24) ? 32 : 16, th = co128 ? 4 : (dma ? 8 : W3_BM / tw);
    const int npatches = N * cdiv(W, tw) * cdiv(H, th);
    const int pairs = cdiv(Cout, co128 ? 128 : 64) * cdiv(Cin, 64);
    static const int target_env = getenv("GSSEG_W3_GRID") ? atoi(getenv("GSSEG_W3_GRID")) : 0;
    // one 8-wave / two 4-wave blocks per CU (fewer when gs_set_persistent_grid() leaves CUs to RCCL)
    const int target = target_env > 0 ? target_env : (dma ? gs_get_persistent_grid() : 2 * gs_get_persistent_grid());
    // every part writes a 9 x 64 x 64 fp32 tile per (co, ci) pair (147 KB) that the ordered reduction reads back: a part
    // must cover a few patches for that to be worth it (batch 2: 512 one-patch parts cost more than the MFMAs)
    static const int min_pps_env = getenv("GSSEG_W3_MINPPS") ? atoi(getenv("GSSEG_W3_MINPPS")) : 4;
    const int min_pps = (dma && !co128) ? (min_pps_env + 1) / 2 : min_pps_env;            // in patches of this kernel (256 / 128 pixels)
    // (floor, not ceil: the blocks hold a CU each, so 3 tile pairs x ceil(256 / 3) = 258 blocks ran a second round for two
    // stragglers -- the 192 -> 64 / 384 -> 128 / 768 -> 256 decoder convs of the 3-D net at 0.67-0.76 PFLOP/s against 1.05 for
    // their 128 / 256-channel neighbours)
    int ksplit = target / pairs;
    if (ksplit > npatches / min_pps) ksplit = npatches / min_pps;
    if (ksplit < 1) ksplit = 1;
    const int pps = cdiv(npatches, ksplit);
    if (pps_out) *pps_out = pps;
    if (npatches_out) *npatches_out = npatches;
    return cdiv(npatches, pps);
}

static int wgrad3x3_launch(const void* x, const void* dy, float* dw, int N, int H, int W, int Cin,
                           int in_pix_stride, int in_coff, int Cout, int out_pix_stride, int out_coff, int dtype,
                           void* stream, int D, int dz, int64_t slab_stride = 0) {
    GS_CHECK_ARG(x && dy && dw, "gs_conv3x3_wgrad: null pointer");
    GS_CHECK_ARG(N > 0 && H > 0 && W > 0 && Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout % 8 == 0, "gs_conv3x3_wgrad: bad dims");
    GS_CHECK_ARG(in_pix_stride >= in_coff + Cin && in_pix_stride % 8 == 0 && in_coff % 8 == 0, "gs_conv3x3_wgrad: bad x stride");
    GS_CHECK_ARG(out_pix_stride >= out_coff + Cout && out_pix_stride % 8 == 0 && out_coff % 8 == 0, "gs_conv3x3_wgrad: bad dy stride");
    GS_CHECK_ARG((int64_t)H * W * in_pix_stride * 2 < 4294967000LL && (int64_t)H * W * out_pix_stride * 2 < 4294967000LL,
                 "gs_conv3x3_wgrad: one image must stay below 4 GiB (32-bit buffer offsets)");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_conv3x3_wgrad: bad dtype");
    W3Args a;
    a.x = (const unsigned short*)x; a.dy = (const unsigned short*)dy; a.dw = dw;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.in_stride = in_pix_stride; a.in_coff = in_coff;
    a.Cout = Cout; a.out_stride = out_pix_stride; a.out_coff = out_coff;
    a.D = D; a.dz = dz; a.slab_stride = slab_stride;
    const bool dma = w3_use_dma(W);
    const bool co128 = dma && w3_use_co128(W, Cout);
    const int tw = (W >= 24) ? 32 : 16, th = co128 ? 4 : (dma ? 8 : W3_BM / tw);
    a.tiles_x = cdiv(W, tw); a.tiles_y = cdiv(H, th);
    a.npatches = N * a.tiles_x * a.tiles_y;
    a.ncob = cdiv(Cout, co128 ? 128 : 64); a.ncib = cdiv(Cin, 64);
    const int pairs = a.ncob * a.ncib;
    {
        int pps = 0, np = 0;
        a.ksplit = w3_ksplit(N, H, W, Cin, Cout, &pps, &np, dma);
        a.pps = pps;
    }
    hipStream_t s = (hipStream_t)stream;
    if (dma) {
        dim3 dgrid(pairs * a.ksplit);
        if (co128) {
            if (dtype == GS_F16) wgrad3x3_dma128_kernel<GS_F16><<<dgrid, 512, 0, s>>>(a);
            else wgrad3x3_dma128_kernel<GS_BF16><<<dgrid, 512, 0, s>>>(a);
        } else if (dtype == GS_F16) wgrad3x3_dma_kernel<GS_F16><<<dgrid, 512, 0, s>>>(a);
        else wgrad3x3_dma_kernel<GS_BF16><<<dgrid, 512, 0, s>>>(a);
        GS_CHECK_LAUNCH("gs_conv3x3_wgrad");
        return GS_OK;
    }
    dim3 grid(pairs * a.ksplit), block(256);
    if (dtype == GS_F16) {
        if (tw == 32) wgrad3x3_kernel<GS_F16, 32><<<grid, block, 0, s>>>(a);
        else wgrad3x3_kernel<GS_F16, 16><<<grid, block, 0, s>>>(a);
    } else {
        if (tw == 32) wgrad3x3_kernel<GS_BF16, 32><<<grid, block, 0, s>>>(a);
        else wgrad3x3_kernel<GS_BF16, 16><<<grid, block, 0, s>>>(a);
    }
    GS_CHECK_LAUNCH("gs_conv3x3_wgrad");
    return GS_OK;
}

extern "C" int gs_conv3x3_wgrad(const void* x, const void* dy, float* dw, int N, int H, int W, int Cin,
                                int in_pix_stride, int in_coff, int Cout, int out_pix_stride, int out_coff, int dtype,
                                void* stream) {
    return wgrad3x3_launch(x, dy, dw, N, H, W, Cin, in_pix_stride, in_coff, Cout, out_pix_stride, out_coff, dtype, stream, 1, 0);
}

// weight gradient of the 3x3x3 Conv3d: dw [27][Cout][Cin] (fp32, caller zeroes); depth tap kd = 0..2 pairs dY slice d
// with X slice d + kd - 1 -- three launches of the 2-D halo kernel, each into its own nine weight slots.
extern "C" int gs_conv3d_3x3x3_wgrad(const void* x, const void* dy, float* dw, int NB, int D, int H, int W, int Cin,
                                     int in_pix_stride, int in_coff, int Cout, int out_pix_stride, int out_coff, int dtype,
                                     void* stream) {
    GS_CHECK_ARG(NB > 0 && D > 0 && dw, "gs_conv3d_3x3x3_wgrad: bad arguments");
    for (int kd = 0; kd < 3; ++kd) {
        if (D == 1 && kd != 1) continue;               // a one-slice volume only sees the centre depth tap
        int rc = wgrad3x3_launch(x, dy, dw + (int64_t)kd * 9 * Cout * Cin, NB * D, H, W, Cin, in_pix_stride, in_coff, Cout,
                                 out_pix_stride, out_coff, dtype, stream, D, kd - 1);
        if (rc) return rc;
    }
    return GS_OK;
}

// ---- deterministic split-K without atomics -----------------------------------------------------------------
// gs_conv3x3_wgrad_slabs: every split-K part stores its (64 co x 64 ci x 9 taps) tile into its own slab
//   ws[part][9][Cout][Cin]  (plain coalesced stores; nothing has to be zeroed), returns the number of parts;
// gs_wgrad_reduce_unpack: grad[Cout][Cin][taps] (reference layout; [Cin][Cout][taps] when transposed) =
//   gscale * sum over the parts IN ORDER -- replaces the fp32 atomics (18.9 M per launch, 15-30 % of the kernel and
//   order dependent), the zero fill in front of them and the separate gs_unpack_wgrad pass.
extern "C" int64_t gs_conv3x3_wgrad_ws_floats(int N, int H, int W, int Cin, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
    return (int64_t)w3_ksplit(N, H, W, Cin, Cout, nullptr, nullptr, w3_use_dma(W)) * 9 * Cout * Cin;
}

extern "C" int gs_conv3x3_wgrad_slabs(const void* x, const void* dy, float* ws, int N, int H, int W, int Cin,
                                      int in_pix_stride, int in_coff, int Cout, int out_pix_stride, int out_coff,
                                      int dtype, void* stream) {
    GS_CHECK_ARG(ws != nullptr, "gs_conv3x3_wgrad_slabs: null workspace");
    int rc = wgrad3x3_launch(x, dy, ws, N, H, W, Cin, in_pix_stride, in_coff, Cout, out_pix_stride, out_coff, dtype, stream,
                             1, 0, (int64_t)9 * Cout * Cin);
    if (rc) return rc;
    return GS_OK;
}

namespace {
// block = 128 outputs (32 lanes x float4) x 8 part lanes; fixed summation order: the parts of a lane ascending
// (eight 16-byte loads in flight), then lanes 0..7
__global__ __launch_bounds__(256) void wgrad_reduce_unpack_kernel(const float* __restrict__ ws, int nparts, int64_t stride,
                                                                  float* __restrict__ grad, int A, int B, int T,
                                                                  int transposed, float gscale) {
    __shared__ float4 red[8][33];
    const int jl = threadIdx.x & 31, bl = threadIdx.x >> 5;
    const int64_t n = (int64_t)T * A * B;                     // multiple of 4 (host check)
    const int64_t j = ((int64_t)blockIdx.x * 32 + jl) * 4;   // index in the accumulation layout [t][a][b]
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < n) {
        float4 acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        int p = bl;
        for (; p + 56 < nparts; p += 64) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(ws + (int64_t)(p + 8 * u) * stride + j);
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc[u].x += v[u].x; acc[u].y += v[u].y; acc[u].z += v[u].z; acc[u].w += v[u].w; }
        }
        for (; p < nparts; p += 8) {
            const float4 v = *reinterpret_cast<const float4*>(ws + (int64_t)p * stride + j);
            acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += acc[u].x; s.y += acc[u].y; s.z += acc[u].z; s.w += acc[u].w; }
    }
    red[bl][jl] = s;
    __syncthreads();
    if (bl == 0 && j < n) {
        float4 tsum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float4 v = red[i][jl]; tsum.x += v.x; tsum.y += v.y; tsum.z += v.z; tsum.w += v.w; }
        const float o[4] = {tsum.x, tsum.y, tsum.z, tsum.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t je = j + e;
            const int b = (int)(je % B);
            const int64_t r = je / B;
            const int aa = (int)(r % A), t = (int)(r / A);
            const int64_t dst = transposed ? ((int64_t)b * A + aa) * T + t : ((int64_t)aa * B + b) * T + t;
            grad[dst] = o[e] * gscale;
        }
    }
}
// Few parts, many weights (the deep layers: 4..32 slabs of 0.6..9.4 MB): one thread per (a, b) sums its T taps over the
// parts -- every load instruction of a wave reads 256 contiguous bytes of one slab row, 4 x T loads in flight -- and the
// block writes its 256 x T results, contiguous in the reference layout [a][b][t], through LDS in whole rows.  (The kernel
// above spreads 128 floats x 8 part lanes over a block and scatters 4-byte stores T floats apart: launch- and store-bound
// here, 17 us for 37.7 MB.)
template <int T>
__global__ __launch_bounds__(256) void wgrad_reduce_rows_kernel(const float* __restrict__ ws, int nparts, int64_t stride,
                                                                float* __restrict__ grad, int AB, float gscale) {
    constexpr int TP = T | 1;                                  // odd row stride: conflict-free both ways
    __shared__ float sm[256 * TP];
    const int r0 = blockIdx.x * 256, r = r0 + threadIdx.x;
    float acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = 0.f;
    if (r < AB) {
        int p = 0;
        for (; p + 3 < nparts; p += 4) {
            float v[4][T];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < T; ++t) v[u][t] = ws[(int64_t)(p + u) * stride + (int64_t)t * AB + r];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < T; ++t) acc[t] += v[u][t];
        }
        for (; p < nparts; ++p)
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t] += ws[(int64_t)p * stride + (int64_t)t * AB + r];
    }
#pragma unroll
    for (int t = 0; t < T; ++t) sm[threadIdx.x * TP + t] = acc[t] * gscale;
    __syncthreads();
    const int64_t n = (int64_t)AB * T, o0 = (int64_t)r0 * T;
#pragma unroll
    for (int k = 0; k < T; ++k) {
        const int e = k * 256 + threadIdx.x;                  // element of the block's output: row e / T, tap e % T
        const int64_t o = o0 + e;
        if (o < n) grad[o] = sm[(e / T) * TP + e % T];
    }
}
}  // namespace

extern "C" int gs_wgrad_reduce_unpack(const float* ws, int nparts, float* grad, int A, int B, int taps, int transposed,
                                      float gscale, void* stream) {
    GS_CHECK_ARG(ws && grad && nparts > 0 && A > 0 && B > 0 && taps > 0, "gs_wgrad_reduce_unpack: bad arguments");
    const int64_t n = (int64_t)taps * A * B;
    GS_CHECK_ARG(n % 4 == 0 && (((uintptr_t)ws) & 15) == 0, "gs_wgrad_reduce_unpack: taps*A*B must be a multiple of 4 and ws 16-byte aligned");
    static const int rows_parts = getenv("GSSEG_WGRAD_REDUCE_ROWS") ? atoi(getenv("GSSEG_WGRAD_REDUCE_ROWS")) : 32;
    const int64_t AB = (int64_t)A * B;
    if (!transposed && nparts <= rows_parts && AB >= 256 * 64 && AB < 2147483647LL / 64 && (taps == 9 || taps == 16 || taps == 4)) {
        const int nb = (int)cdiv64(AB, 256);
        hipStream_t s = (hipStream_t)stream;
        if (taps == 9) wgrad_reduce_rows_kernel<9><<<nb, 256, 0, s>>>(ws, nparts, n, grad, (int)AB, gscale);
        else if (taps == 16) wgrad_reduce_rows_kernel<16><<<nb, 256, 0, s>>>(ws, nparts, n, grad, (int)AB, gscale);
        else wgrad_reduce_rows_kernel<4><<<nb, 256, 0, s>>>(ws, nparts, n, grad, (int)AB, gscale);
        GS_CHECK_LAUNCH("gs_wgrad_reduce_unpack");
        return GS_OK;
    }
    wgrad_reduce_unpack_kernel<<<(int)cdiv64(n, 128), 256, 0, (hipStream_t)stream>>>(ws, nparts, n, grad, A, B, taps, transposed, gscale);
    GS_CHECK_LAUNCH("gs_wgrad_reduce_unpack");
    return GS_OK;
}

extern "C" int gs_conv3x3_wgrad_parts(int N, int H, int W, int Cin, int Cout) {
    return w3_ksplit(N, H, W, Cin, Cout, nullptr, nullptr, w3_use_dma(W));
}

// 3-D form of the deterministic weight gradient: the three depth-tap launches store their parts into one set of slabs
// ws[part][27][Cout][Cin]; gs_wgrad_reduce_unpack(taps = 27) then writes [Cout][Cin][3][3][3].
extern "C" int64_t gs_conv3d_3x3x3_wgrad_ws_floats(int NB, int D, int H, int W, int Cin, int Cout) {
    if (NB <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
    return (int64_t)w3_ksplit(NB * D, H, W, Cin, Cout, nullptr, nullptr, w3_use_dma(W)) * 27 * Cout * Cin;
}

extern "C" int gs_conv3d_3x3x3_wgrad_parts(int NB, int D, int H, int W, int Cin, int Cout) {
    return w3_ksplit(NB * D, H, W, Cin, Cout, nullptr, nullptr, w3_use_dma(W));
}

extern "C" int gs_conv3d_3x3x3_wgrad_slabs(const void* x, const void* dy, float* ws, int NB, int D, int H, int W, int Cin,
                                           int in_pix_stride, int in_coff, int Cout, int out_pix_stride, int out_coff,
                                           int dtype, void* stream) {
    GS_CHECK_ARG(NB > 0 && D > 0 && ws, "gs_conv3d_3x3x3_wgrad_slabs: bad arguments");
    const int64_t slab = (int64_t)27 * Cout * Cin;
    for (int kd = 0; kd < 3; ++kd) {
        if (D == 1 && kd != 1) {
            // a one-slice volume never pairs with the outer depth taps: their slots must still be defined (zeros)
            const int parts = w3_ksplit(NB * D, H, W, Cin, Cout, nullptr, nullptr, w3_use_dma(W));
            for (int p = 0; p < parts; ++p)
                if (hipMemsetAsync(ws + p * slab + (int64_t)kd * 9 * Cout * Cin, 0, (size_t)9 * Cout * Cin * 4, (hipStream_t)stream) != hipSuccess)
                    return GS_ELAUNCH;
            continue;
        }
        int rc = wgrad3x3_launch(x, dy, ws + (int64_t)kd * 9 * Cout * Cin, NB * D, H, W, Cin, in_pix_stride, in_coff, Cout,
                                 out_pix_stride, out_coff, dtype, stream, D, kd - 1, slab);
        if (rc) return rc;
    }
    return GS_OK;
}

