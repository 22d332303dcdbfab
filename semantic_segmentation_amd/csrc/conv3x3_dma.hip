// 3x3 / stride-1 / pad-1 convolution (forward and data gradient, Conv3d 3x3x3 through depth-tap stages) on the LDS-DMA
// kernel: the same (patch x 64 couts) items as the big K-step kernel of conv3x3.hip, but the operands never pass through
// registers: every K stage (32 input channels: the halo + the nine 64x32 weight slabs) is written into LDS by
// `buffer_load_dwordx4 ... lds` (1 KiB = 16 rows x 64 B per wave-instruction) into the buffer the previous stage has just
// left, while the current stage's 72 MFMAs per wave run -- no staging registers, no ds_write phase, one barrier per stage.
//   LDS image : rows of 64 B (32 channels), no padding (the DMA destination is lane-linear); the 16-byte slot s of row r
//               lives at slot s ^ ((r >> 2) & 3): any 16 rows {b..b+3, b+12..b+15, b+20..b+27} (one ds_read_b128 lane
//               group of a 32-row fragment) hit 16 distinct 16-byte bank groups for every base b.  The swizzle is
//               applied on the SOURCE side: lane l of a piece fetches logical slot (l & 3) ^ ((l >> 4) & 3).
//   zero pad  : out-of-image halo pixels / couts >= Cout carry an out-of-range buffer offset: the DMA writes zeros.
//   stage s   : wait vmcnt(0) + barrier (stage s landed everywhere, stage s-1 fully read) -> issue the pieces of stage
//               s+1 interleaved with the first MFMA steps -> 18 steps of 4 MFMAs (fragment reads one step ahead).
//   NWV       : 4 waves = 8x32 pixels (one wave per SIMD) or 8 waves = 16x32 pixels (two per SIMD, weights fetched
//               once per 512 pixels).
// Requires Cin % 64 == 0 (an even number of stages keeps the buffer parity fixed per item), W >= 24, the forward or the
// data-gradient (flipped) tap table.  Conv3d 3x3x3: the stages run over (depth tap, channel chunk) -- stage (dz, c) fetches the
// halo of slice n + dz (empty pieces outside the volume) and the nine weight slots of that depth tap.  Everything else (the
// 16x16 level, Cin % 64 != 0) stays on conv3x3_big_kernel.
//
// Two epilogue forms (template flag DEFER):
//   DEFER = false (the pair / precise forward, PREC): as conv3x3_big_kernel -- at the end of an item every wave packs its
//     64 x 64 tile (hi and lo), transposes it through an LDS staging area that overlays the second stage buffer and stores
//     128-byte rows.
//   DEFER = true (everything else): the epilogue of item k runs UNDER THE MFMAs OF ITEM k+1.  Measured on the round-2
//     kernel (tools/ablate_conv_epilogue.py): without its epilogue the 64->64 layer at 256^2 runs 28 % faster (128->64:
//     19 %, 128->128 @128^2: 12 %), while the global stores themselves cost 2-3 % -- the time is the pack / DPP / LDS
//     transposition work with no MFMA in flight, because the two waves of a SIMD reach it together.  Now, at the item
//     boundary a wave only converts its fp32 accumulators to 32 registers of packed 16-bit pairs (and adds them to its
//     running statistics); the transposition (8 passes of 8 pixel rows through a PRIVATE 1 KB LDS buffer per wave: 4
//     ds_write_b32, one ds_read_b128, one 16-byte store each) is spread over the steps of the next item's first stage, two
//     steps per pass.  Items with an activation (inference) or a partial patch take the immediate path.
// Both forms keep the BatchNorm partial sums in registers across ALL items of the block (a block keeps its cout tile: the grid
// is a multiple of the tile count) and write them once at the end: one row per block instead of one per half-item
// (c3_dma_grid() rows; no LDS hand-over and barrier per item, no first-stage reduction launch afterwards).
#include <stdlib.h>
#include <type_traits>

#include "conv3x3_args.hpp"

namespace {

// (a "v" constraint inside the kernel body itself would make the host-side instantiation of the launch stub invalid)
__device__ __forceinline__ void opaque_vgpr(unsigned& x) { asm volatile("" : "+v"(x)); }
// one LDS-DMA piece: lane l's 16 bytes at buffer offset voff + soff land at dst + 16 l (dst wave-uniform).  A device
// function of its own: the address-space cast inside the kernel body invalidates the host-side launch stub.
__device__ __forceinline__ void dma_piece16(const __amdgpu_buffer_rsrc_t& rs, unsigned char* dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)dst, 16, voff, soff, 0, 0);
}

typedef __attribute__((ext_vector_type(8))) int i32x8;

// MI = 32-pixel rows per wave: 2 (the 4- and 8-wave forms) or 4 -- the "quad" form: 4 waves, ONE per SIMD, each with a 128-pixel x 64-cout
// tile (eight 32x32 MFMA tiles, 128 accumulator registers): the 16x32-pixel items of the 8-wave form -- the same DMA bytes per MFMA --
// with half the fragment reads per MFMA (6 per 8 instead of 4 per 4) and the whole 512-register file of the SIMD for one wave
// (double-buffered fragments, the deferred epilogue's packed tile, no spills).
template <int DT, int NWV, bool STATS, bool PREC, bool DEFER, bool Q8 = false, int MI = 2>
__global__ __launch_bounds__(64 * NWV, 1) void conv3x3_dma_kernel(const C3Args a) {
    typedef typename Elem<DT>::V8 V8;
    // the pair epilogue is deferred only in the 4-wave form: one wave per SIMD has the 512-register file to itself, so the 64
    // registers of packed hi + lo results can ride through the next item's first stage (the 8-wave form has 256 per wave)
    static_assert(!(PREC && DEFER) || NWV == 4, "the deferred pair epilogue needs the 4-wave form's registers");
    static_assert(MI == 2 || (MI == 4 && NWV == 4 && DEFER), "the quad form: 4 waves, deferred epilogue");
    static_assert(!Q8 || PREC, "FP8 correction stages belong to the pair forward");
    constexpr int BN = 64, TW = 32, TH = MI * NWV, TWS = 5, KC = 32;
    constexpr int HWD = TW + 2, HHT = TH + 2, HP = HWD * HHT;
    constexpr int ROWB = KC * 2;                           // bytes per LDS row (one pixel / one cout, KC channels)
    constexpr int SPR = ROWB / 16, RPP = 64 / SPR;         // 16-byte slots per row, rows per 1-KiB piece
    constexpr int KSTEPS = KC / 16, NSTEP = 9 * KSTEPS;    // MFMA steps (tap, k half) per stage
    constexpr int HI = (HP + RPP - 1) / RPP;               // halo pieces that hold real rows
    constexpr int HJ = (HI + NWV - 1) / NWV;               // halo pieces per wave (the tail ones are spare)
    constexpr int WG = BN / RPP;                           // weight pieces (row groups) per tap
    constexpr int WP = 9 * WG;                             // weight pieces per stage
    constexpr int NWP = (WP + NWV - 1) / NWV;              // ... per wave
    constexpr int HALO_B = HJ * NWV * 1024, W_B = 9 * BN * ROWB, STAGE_B = HALO_B + W_B;
    constexpr int STG_EL = 32 * C3_LDR;
    constexpr unsigned VOOB = 0x80000000u;
    // LDS: [weights 0 | halo 0 | halo 1 | weights 1 | private transposition buffers (DEFER)]; the immediate epilogue's
    // staging overlays halo 1 + weights 1 (the last stage of an item always sits in buffer 1, the next item's first stage
    // is on its way into buffer 0)
    constexpr int H0_OFF = W_B, W1_OFF = W_B + 2 * HALO_B;
    constexpr int PRIV_OFF = 2 * STAGE_B, PRIV_B = DEFER ? (PREC ? 2 : 1) * NWV * 1024 : 0;       // (PREC: a second buffer per wave for the lo halves)
    constexpr int LDS_B = 2 * STAGE_B + PRIV_B;
    static_assert(LDS_B <= 160 * 1024, "the stage buffers must fit in LDS");
    // PREC (precise mode): K is a concatenation of segments over the same input channels (stage c reads
    // input chunk c mod wrap) and the result leaves as a hi / lo pair (a second staging area and store stream)
    static_assert((PREC ? 2 : 1) * NWV * STG_EL * 2 + NWV * 2 * BN * 4 <= HALO_B + W_B, "epilogue staging overlays the second stage buffer");
    static_assert(NWP + HJ <= NSTEP + 1 && HJ + 1 <= NSTEP, "one DMA piece (two in step 0 of the quad form) / one next-item offset per MFMA step");
    static_assert(NWV % WG == 0, "a wave's weight pieces all belong to one row group");
    static_assert(WP % NWV == 0 || HI < HJ * NWV, "a surplus weight slot parks its (empty) piece in the spare halo piece");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_B];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int nitems = a.nblocks;
    const int act = a.act & 0xff;
    const int dbg = a.act >> 8;                            // ablation bits (GSSEG_C3_DEBUG=1): 2 no MFMAs, 4 no epilogue, 8 / 16 no
                                                           // weight / halo traffic, 64 no global stores (immediate epilogue only)
    const int nstage = a.ndz * (a.Cin / KC);               // even: Cin % 64 == 0
    const unsigned img_bytes = (unsigned)a.H * a.W * a.in_stride * 2u;
    const unsigned tap_stride = (unsigned)a.Cout * a.Cin * 2u;
    const bool flip = a.tap_dy[0] > 0;                     // data-gradient table: geometric tap g uses weight slot 8 - g
    const int tiles_y8 = (a.H + 7) >> 3;

#ifdef GS_C3_PHASE_TIMING          // diagnostic build (tools/conv_dma_phase.py): cycles per phase of this wave, summed over its items
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = clock64();
#define PH(i) do { __builtin_amdgcn_sched_barrier(0); const long long t_ = clock64(); ph[i] += t_ - tq; tq = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PH(i) do {} while (0)
#endif
    // (Two waves share a SIMD's matrix pipe and the OLDER one wins every arbitration: in-kernel stamps (tools/conv_dma_phase.py,
    // profiles/r03_conv_phase_stamps.txt) show wave 0 finishing a stage after ~60 % of the stage time and waiting at the barrier
    // while its partner, wave 5, spends 91 % of its time inside stage bodies.  Evening them out does not help: a static
    // s_setprio for waves 4..7 (+0.5 %) and a priority hand-over inside every stage at step 7 / 9 / 11 / 13 (+1.2 / +0.4 / +0.2 /
    // +0.2 %, profiles/r03_ab_setprio.txt) both measured no gain -- the kernel delivers what the chip grants this instruction
    // mix at the 1.8 GHz it holds under it, however the two waves split the pipe.)
    struct Item { int n, y0, x0, n0, mt; };
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (unsigned)(9u * a.ndz * a.Cout * a.Cin * 2u), 0x00020000);

    // ---- DMA side: lane l of a piece fills physical slot l % SPR of row l / SPR; the logical slot (8 channels) that belongs
    // there is the physical one xor-ed with the row's swizzle key (row >> 2) & 3 ----
    const int drow = lane / SPR;
    const int dls = (lane % SPR) ^ ((lane >> 4) & (SPR - 1));
    unsigned hv[HJ], wv;         // this item's buffer offsets per piece slot
    unsigned hvn[HJ], wvn;       // the next item's: computed one slot per MFMA step of every even stage (nearly free
                                 // there; at the item boundary the same ~70 instructions ran with nothing to hide behind)
    auto halo_voff = [&](int y0, int x0, int j) __attribute__((always_inline)) {      // j compile-time
        const int r = (wave + j * NWV) * RPP + drow;
        const int hy = r / HWD, hx = r - hy * HWD;
        const int gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = r < HP && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        return ok ? (unsigned)(((gy * a.W + gx) * a.in_stride + a.in_coff + dls * 8) * 2) : VOOB;
    };
    auto weight_voff = [&](int n0) __attribute__((always_inline)) {
        const int co = n0 + (wave % WG) * RPP + drow;
        return co < a.Cout ? (unsigned)((co * a.Cin + dls * 8) * 2) : VOOB;
    };
    auto setup_item = [&](const Item& itn) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < HJ; ++j) hv[j] = halo_voff(itn.y0, itn.x0, j);
        wv = weight_voff(itn.n0);
    };
    // piece k (compile-time) of stage (image rx, channel byte offset sc) into buffer bb
    // (sc / wsc: scalar byte offsets of the stage's channel chunk in the image / of its chunk and depth-tap group in the pack;
    // hkill: the depth tap points outside the volume -- the halo pieces carry zeros)
    // (issuing the halo pieces before the weight pieces measured the same)
    auto issue_piece = [&](int k, const __amdgpu_buffer_rsrc_t& rx, unsigned sc, unsigned wsc, unsigned bb, unsigned hkill,
                           unsigned kill, bool wskip = false) __attribute__((always_inline)) {
        if (k < NWP) {
            if (wskip) return;                             // resident weights (WRES below): the slabs are already there
            const int pc = wave + NWV * k;                 // piece = tap * WG + row group; the group is wave % WG for every k
            const int tap = pc / WG;
            const bool real = pc < WP;                     // surplus slots of the last round carry an empty piece
            const int slot = flip ? 8 - tap : tap;
            const unsigned dst = real ? (unsigned)(pc * 1024) + bb * W1_OFF
                                      : (unsigned)(H0_OFF + HALO_B - 1024) + bb * HALO_B;
            dma_piece16(w_rsrc, smem + dst, (real && !(dbg & 8)) ? (wv | kill) : VOOB, (unsigned)(real ? slot : 0) * tap_stride + wsc);
        } else if (k - NWP < HJ) {
            const int j = k - NWP < HJ ? k - NWP : 0;
            const unsigned dst = H0_OFF + bb * HALO_B + (unsigned)(wave + j * NWV) * 1024u;
            dma_piece16(rx, smem + dst, (dbg & 16) ? VOOB : (hv[j] | kill | hkill), sc);
        }
    };
    // stage c of an item in slice n: 2-D: channel chunk c; Conv3d: (depth tap c / nchunk, chunk c % nchunk) reads slice n + dz
    // and the nine weight slots of that depth tap
    struct Src { int n; unsigned sc, wsc, hkill; };
    const int nchunk = a.Cin / KC;
    auto stage_src = [&](int n, int c) __attribute__((always_inline)) {
        Src r;
        r.n = n; r.sc = (unsigned)c * ROWB; r.wsc = r.sc; r.hkill = 0u;
        if (PREC && a.in_wrap > 0 && c >= 2 * a.in_wrap) r.sc = (unsigned)(c - 2 * a.in_wrap + 2 * a.in_wrap_to) * ROWB;    // in_wrap / in_wrap_to count 64-channel chunks
        if (a.ndz > 1) {
            const int dzi = c / nchunk, cc = c - dzi * nchunk;
            const int dz = a.tap_dz[dzi];
            const int d = n % a.D;
            const bool inside = (unsigned)(d + dz) < (unsigned)a.D;
            r.n = inside ? n + dz : n;
            r.hkill = inside ? 0u : VOOB;
            // (pair forward: K chunk cc of a depth tap reads input chunk cc, wrapping once to the start of the input)
            r.sc = (unsigned)((PREC && a.in_wrap > 0 && cc >= 2 * a.in_wrap) ? cc - 2 * a.in_wrap + 2 * a.in_wrap_to : cc) * ROWB;
            r.wsc = (unsigned)(dzi * 9) * tap_stride + (unsigned)cc * ROWB;
        }
        return r;
    };

    // ---- MFMA side: fragment byte addresses inside a stage buffer (item independent) ----
    // (kept opaque: hipcc otherwise materialises every base + constant combination of both buffers in its own register)
    // (DEFER is short of registers -- 32 of packed results ride through the next item's first stage: the second k half's
    // address is formed at the read, one v_xor each)
    constexpr int AK = (DEFER || Q8) ? 1 : KSTEPS;          // (Q8: the FP8 stage's 48 fragment registers leave no room for both sets)
    unsigned aaddr[AK][MI + 2][3];                              // [k half][halo row 2*wave + e, e = i + dy + 1][column l31 + dx + 1]
#pragma unroll
    for (int e = 0; e < MI + 2; ++e)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int r = (MI * wave + e) * HWD + l31 + d;
            const int key = (r >> 2) & 3;
            aaddr[0][e][d] = (unsigned)(H0_OFF + r * ROWB + ((h ^ key) << 4));
            opaque_vgpr(aaddr[0][e][d]);
            if (AK == 2) {
                aaddr[AK - 1][e][d] = aaddr[0][e][d] ^ 32u;
                opaque_vgpr(aaddr[AK - 1][e][d]);
            }
        }
    unsigned baddr[2][KSTEPS];                             // [buffer][k half]
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int key = (l31 >> 2) & 3;
        baddr[b][0] = (unsigned)(b * W1_OFF + l31 * ROWB + ((h ^ key) << 4));
        opaque_vgpr(baddr[b][0]);
        baddr[b][1] = baddr[b][0] ^ 32u;
        opaque_vgpr(baddr[b][1]);
    }

    f32x16 acc[MI][2];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };

    // ---- deferred epilogue state (DEFER): the previous item's tile as packed 16-bit pairs, its store base, its image ----
    constexpr int NPK = DEFER ? 8 : 1;
    unsigned pk[MI][2][NPK];                                // [i][j][m]: rows 2m, 2m+1 of the lane's cout in tile (i, j)
    constexpr int NPKL = (DEFER && PREC) ? 8 : 1;
    unsigned pkl[MI][2][NPKL];                              // PREC: the lo halves of the same pairs
    unsigned pend_voff = MI == 4 ? 0x80000000u : 0u;       // lane part of the store offset of the pending tile (quad form: out of range = nothing pending)
    int pend_n = 0;                                        // its image
    bool pend = false, pend2 = false;                      // (pend2: the quad form's second eight passes are still to come)
    const bool odd = lane & 1;
    const unsigned int psel = odd ? 0x03020706u : 0x05040100u;
    // private transposition buffer of the wave: 8 pixel rows x 128 B (64 couts), 16-byte slot s of row r at s ^ r
    unsigned char* const priv = smem + PRIV_OFF + wave * 1024;
    const unsigned row_stride_b = (unsigned)a.W * a.out_stride * 2u;          // one image row of the output, bytes
    // pass p (0..7) of the pending tile, three phases: W (4 ds_write_b32), R (ds_read_b128), S (the 16-byte store)
    u32x4 dsv, dsvl;                                       // the read-back values in flight (PREC: hi and lo)
    auto defer_write = [&](int p) __attribute__((always_inline)) {          // p compile-time
        const int i = p >> 2, m0 = (p & 3) * 2;
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
            const int m = m0 + mm;                         // rows 2m, 2m+1 (+4h) of the 32-row tile: local rows 2*mm + odd + 4h
            const int lr = 2 * mm + (odd ? 1 : 0) + 4 * h;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned own = pk[i][j][DEFER ? m : 0];
                const unsigned oth = (unsigned)__builtin_amdgcn_mov_dpp((int)own, 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true);
                const unsigned v = __builtin_amdgcn_perm(oth, own, psel);
                const int slot = (j * 4 + (l31 >> 3)) ^ lr;                  // 16-byte slot (8 couts) of cout pair l31 & ~1
                *reinterpret_cast<unsigned*>(priv + lr * 128 + slot * 16 + (l31 & 6) * 2) = v;
                if (PREC) {
                    const unsigned ownl = pkl[i][j][(DEFER && PREC) ? m : 0];
                    const unsigned othl = (unsigned)__builtin_amdgcn_mov_dpp((int)ownl, 0xB1, 0xf, 0xf, true);
                    *reinterpret_cast<unsigned*>(priv + NWV * 1024 + lr * 128 + slot * 16 + (l31 & 6) * 2) = __builtin_amdgcn_perm(othl, ownl, psel);
                }
            }
        }
    };
    // The read-back is inline assembly: hipcc cannot tell a plain LDS read from the destination of the LDS-DMA pieces in flight
    // and puts `s_waitcnt vmcnt(0)` in front of it -- every pass then waited for the whole next stage to land.  The wait for
    // the value (defer_wait) sits at the top of the following step, where the step's own fragment reads are due anyway.
    const unsigned priv_rd = (unsigned)(PRIV_OFF + wave * 1024 + (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4));
    auto defer_read = [&](int p) __attribute__((always_inline)) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(dsv) : "v"(priv_rd) : "memory");
        if (PREC) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dsvl) : "v"(priv_rd), "n"(NWV * 1024) : "memory");
    };
    auto defer_wait = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    auto defer_store = [&](int p) __attribute__((always_inline)) {
        const int i = p >> 2;
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.y + (int64_t)pend_n * a.H * a.W * a.out_stride), 0, (unsigned)a.H * a.W * a.out_stride * 2u, 0x00020000);
        const unsigned soff = (unsigned)i * row_stride_b + (unsigned)((p & 3) * 8) * (unsigned)a.out_stride * 2u;
        __builtin_amdgcn_raw_buffer_store_b128(dsv, ry, pend_voff, soff, GS_OUT_AUX);
        if (PREC) {
            const __amdgpu_buffer_rsrc_t ryl = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(a.y_lo + (int64_t)pend_n * a.H * a.W * a.out_stride), 0, (unsigned)a.H * a.W * a.out_stride * 2u, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(dsvl, ryl, pend_voff, soff, GS_OUT_AUX);
        }
    };
    // the deferred work of MFMA step `step` of the first stage: pass p = step / 2 writes at step 2p, reads at 2p + 1, stores at 2p + 2
    // (in front of the writes of pass p + 1; LDS operations of a wave execute in order)
    // (quad form: 16 passes -- the first eight on the item's first stage, pb = 0, the other eight on its second stage, pb = 8)
    auto defer_step = [&](int step, int pb) __attribute__((always_inline)) {
        if (step >= 2 && step % 2 == 0 && (step - 2) / 2 < 8) defer_store(pb + (step - 2) / 2);
        if (step % 2 == 0 && step / 2 < 8) defer_write(pb + step / 2);
        if (step % 2 == 1 && step / 2 < 8) defer_read(pb + step / 2);
    };
    auto defer_step_head = [&](int step) __attribute__((always_inline)) {     // top of a step that stores: the read-back has landed
        if (step >= 2 && step % 2 == 0 && (step - 2) / 2 < 8) defer_wait();
    };
    static_assert(NSTEP >= 17, "eight passes need steps 0..16 of one stage");
    auto defer_flush = [&]() __attribute__((always_inline)) {               // everything at once (after the last item)
#pragma unroll
        for (int p = 0; p < 4 * MI; ++p) {
            defer_write(p);
            defer_read(p);
            defer_wait();
            defer_store(p);
        }
    };

    // one stage: NSTEP steps (tap, k half) of 4 MFMAs out of buffer BUF; the pieces of the next stage (image rx_n, channel
    // offset sc_n) go into the other buffer, one per step
    // FIRST: the first stage of an item starts its accumulators from the MFMA's zero C operand (no clearing pass)
    // PEND : the previous item's deferred epilogue passes ride on this stage's steps
    auto run_stage = [&](auto buf_tag, auto first_tag, auto pend_tag, const __amdgpu_buffer_rsrc_t& rx_n, const Src& sn, unsigned kill,
                         const Item& itn, bool wskip = false) __attribute__((always_inline)) {
        const unsigned sc_n = sn.sc, wsc_n = sn.wsc, hkill_n = sn.hkill;
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int PENDV = (int)decltype(pend_tag)::value;      // 0: nothing pending; 1: deferred passes 0..7 ride here; 2: passes 8..15
        constexpr bool PEND = PENDV != 0;
        constexpr unsigned OBUF = 1 - BUF;           // the buffer the next stage is fetched into
        if (dbg & 2) {                                     // ablation: DMA traffic only
#pragma unroll
            for (int k = 0; k < NWP + HJ; ++k) issue_piece(k, rx_n, sc_n, wsc_n, OBUF, hkill_n, kill, wskip);
            if (BUF == 0) {
#pragma unroll
                for (int j = 0; j < HJ; ++j) hvn[j] = halo_voff(itn.y0, itn.x0, j);
                wvn = weight_voff(itn.n0);
            }
            if (FIRST) zero_acc();
            if (PEND) defer_flush();
            return;
        }
        V8 af[2][MI], bf[2][2];
        auto frag_load = [&](int step, V8 (&fa)[MI], V8 (&fb)[2]) __attribute__((always_inline)) {
            const int tap = step / KSTEPS, kh = step % KSTEPS;
            const int dyi = tap / 3, dxi = tap - 3 * dyi;
            auto aof = [&](int i) __attribute__((always_inline)) {
                return (AK == 1 && kh == 1) ? (aaddr[0][dyi + i][dxi] ^ 32u) : aaddr[AK == 1 ? 0 : kh][dyi + i][dxi];
            };
            fa[0] = *reinterpret_cast<const V8*>(smem + aof(0) + BUF * HALO_B);
            fb[0] = *reinterpret_cast<const V8*>(smem + baddr[BUF][kh] + tap * (BN * ROWB));
            fa[1] = *reinterpret_cast<const V8*>(smem + aof(1) + BUF * HALO_B);
            fb[1] = *reinterpret_cast<const V8*>(smem + baddr[BUF][kh] + tap * (BN * ROWB) + 32 * ROWB);
#pragma unroll
            for (int i = 2; i < MI; ++i) fa[i] = *reinterpret_cast<const V8*>(smem + aof(i) + BUF * HALO_B);
        };
        frag_load(0, af[0], bf[0]);
#pragma unroll
        for (int step = 0; step < NSTEP; ++step) {
            const int cur = step & 1;
            if (PEND) defer_step_head(step);
            if (step + 1 < NSTEP) frag_load(step + 1, af[cur ^ 1], bf[cur ^ 1]);
            // (two pieces per step, i.e. everything issued in the first half of the stage, measured the same or 1 % slower
            // on the deep layers: the pieces are not late)
            issue_piece(step, rx_n, sc_n, wsc_n, OBUF, hkill_n, kill, wskip);
            if (NWP + HJ > NSTEP && step == 0) issue_piece(NSTEP, rx_n, sc_n, wsc_n, OBUF, hkill_n, kill, wskip);      // quad form: 19 pieces
            if (BUF == 0 && step <= HJ) {                  // the next item's piece offsets, one slot per step
                int py0 = itn.y0, px0 = itn.x0, pn0 = itn.n0;
                asm volatile("" : "+s"(py0), "+s"(px0), "+s"(pn0));      // keeps this arithmetic in the step (else hoisted to the item boundary)
                if (step < HJ) hvn[step < HJ ? step : 0] = halo_voff(py0, px0, step < HJ ? step : 0);
                else wvn = weight_voff(pn0);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    if (FIRST && step == 0) {
                        f32x16 z;
#pragma unroll
                        for (int r = 0; r < 16; ++r) z[r] = 0.f;
                        acc[i][j] = Elem<DT>::mfma32(af[cur][i], bf[cur][j], z);
                    } else {
                        acc[i][j] = Elem<DT>::mfma32(af[cur][i], bf[cur][j], acc[i][j]);
                    }
                }
            if (PEND) defer_step(step, PENDV == 2 ? 8 : 0);
            if (MI == 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // one LDS fragment read in its shadow
                }
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);           // the DMA piece of this step
            } else {                                       // quad form: eight MFMAs, six fragment reads, one or two DMA pieces
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (step == 0) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                else __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // ---- FP8 correction stage (Q8): the same 64-byte rows, now [lo8 (32 ch) | hi8 (32 ch)] pixels against [w_hi8 | w_lo8] couts.
    // One block-scaled e4m3 MFMA per (tap, 32x32 tile) covers BOTH correction terms of 32 channels: a lane's operand is the 16-byte
    // slots {h, 2 + h} of its row -- the two addresses of the 16-bit steps -- whose first 16 bytes form K block 0 (x_lo8 . w_hi8,
    // scaled by the E8M0 bytes of lanes 0..31) and last 16 K block 1 (x_hi8 . w_lo8, lanes 32..63).  9 steps of 4 MFMAs at 64
    // cycles each = the time of a 16-bit stage; the same eight fragment reads per tap, two DMA pieces per step.
    int q8_sa = 0, q8_sb[2] = {0, 0};                       // E8M0 scale bytes of this lane's K block: activations / couts j = 0, 1
    auto run_stage8 = [&](auto buf_tag, const __amdgpu_buffer_rsrc_t& rx_n, const Src& sn, unsigned kill, const Item& itn)
                          __attribute__((always_inline)) {
        const unsigned sc_n = sn.sc, wsc_n = sn.wsc, hkill_n = sn.hkill;
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr unsigned OBUF = 1 - BUF;
        // registers: ONE set of fragments (pixel tiles 2 x 8, cout tiles 2 x 8 = 32, what a 16-bit stage holds): a fragment's next
        // tap is fetched right behind the last MFMA that reads it -- the pixel fragments behind the step's last MFMA, i.e. their
        // latency is covered by the SIMD's other wave, not by this wave's own MFMAs.  (Double-buffered pixel fragments -- 48
        // registers -- pushed the 8-wave kernel over its 256 registers: 87 dwords per lane spilled, +6..8 us per item.)
        typedef __attribute__((ext_vector_type(4))) int i32x4;
        i32x4 af8[2][2], bf8[2][2];                         // [tile][16-byte half]
        auto load_a8 = [&](int tap, int i, i32x4 (&fa)[2]) __attribute__((always_inline)) {
            const int dyi = tap / 3, dxi = tap - 3 * dyi;
            fa[0] = *reinterpret_cast<const i32x4*>(smem + aaddr[0][dyi + i][dxi] + BUF * HALO_B);
            fa[1] = *reinterpret_cast<const i32x4*>(smem + (aaddr[0][dyi + i][dxi] ^ 32u) + BUF * HALO_B);
        };
        auto load_b8 = [&](int tap, int j, i32x4 (&fb)[2]) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
                fb[q] = *reinterpret_cast<const i32x4*>(smem + baddr[BUF][q] + tap * (BN * ROWB) + j * 32 * ROWB);
        };
        auto mfma8 = [&](int i, int j) __attribute__((always_inline)) {
            const i32x8 va = __builtin_shufflevector(af8[i][0], af8[i][1], 0, 1, 2, 3, 4, 5, 6, 7);
            const i32x8 vb = __builtin_shufflevector(bf8[j][0], bf8[j][1], 0, 1, 2, 3, 4, 5, 6, 7);
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc[i][j], 0, 0, 0, q8_sa, 0, q8_sb[j]);
        };
        load_a8(0, 0, af8[0]);
        load_b8(0, 0, bf8[0]);
        load_a8(0, 1, af8[1]);
        load_b8(0, 1, bf8[1]);
#pragma unroll
        for (int step = 0; step < 9; ++step) {
            const bool more = step + 1 < 9;
            issue_piece(2 * step, rx_n, sc_n, wsc_n, OBUF, hkill_n, kill);
            issue_piece(2 * step + 1, rx_n, sc_n, wsc_n, OBUF, hkill_n, kill);
            // (the next item's piece offsets are formed in the item's first stage, which is always a 16-bit one)
            // order (i, j): (0,0) (1,0) (1,1) (0,1): after the second MFMA cout fragment 0 is free, after the third pixel fragment 1,
            // after the fourth pixel fragment 0 and cout fragment 1
            mfma8(0, 0);
            mfma8(1, 0);
            if (more) load_b8(step + 1, 0, bf8[0]);
            mfma8(1, 1);
            if (more) load_a8(step + 1, 1, af8[1]);
            mfma8(0, 1);
            if (more) load_a8(step + 1, 0, af8[0]);
            if (more) load_b8(step + 1, 1, bf8[1]);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);           // MFMA (0,0), (1,0)
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);           //   next cout fragment 0
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           // MFMA (1,1)
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);           //   next pixel fragment 1
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           // MFMA (0,1)
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);           //   next pixel fragment 0, cout fragment 1
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);           // the two DMA pieces of this step
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    static_assert(!Q8 || NWP + HJ <= 18, "two DMA pieces per FP8 step");

    // stage hand-over: this wave's pieces have landed; after the barrier everybody's have, and nobody reads the other
    // buffer any more.  (Stores count in vmcnt too: the epilogue's are drained here as well.)
    auto stage_sync = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- BatchNorm partial sums: (even row, odd row) sums in packed fp32 math.  DEFER: kept across the items of the block ----
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 s1v[2] = {{0.f, 0.f}, {0.f, 0.f}}, s2v[2] = {{0.f, 0.f}, {0.f, 0.f}};

    // ---- immediate epilogue: as conv3x3_big_kernel (staging overlays the second stage buffer, which the last stage has just left) ----
    unsigned short* stg = reinterpret_cast<unsigned short*>(smem + H0_OFF + HALO_B) + wave * STG_EL;
    unsigned short* stg_lo = stg + NWV * STG_EL;                                        // PREC: the lo halves
    const float neg_slope = act == GS_ACT_RELU ? 0.f : (act == GS_ACT_LEAKY02 ? 0.2f : 1.f);
    auto epilogue_t = [&](const Item& itc, auto plain_tag, auto full_tag) __attribute__((always_inline)) {
        constexpr bool PLAIN = decltype(plain_tag)::value;
        constexpr bool FULL = decltype(full_tag)::value;
        int e_y0 = itc.y0, e_x0 = itc.x0, e_n = itc.n, e_n0 = itc.n0;
        asm volatile("" : "+s"(e_y0), "+s"(e_x0), "+s"(e_n), "+s"(e_n0));
        // (Q8: the staging addresses are re-formed per item from an opaque lane offset.  Left alone, hipcc hoists the ~40 per-lane
        // write / read addresses of both staging areas out of the item loop -- loop invariants -- and, with the FP8 stage's 48 fragment
        // registers in the same kernel, spills them: 87 dwords per lane, reloaded in every item's epilogue: +6..8 us per item measured)
        unsigned stg_b = (unsigned)(wave * STG_EL * 2 + ((l31 & ~1) * 2)), stg_rb = (unsigned)(wave * STG_EL * 2 + ((lane >> 3) * C3_LDR + (lane & 7) * 8) * 2);
        if (Q8) { opaque_vgpr(stg_b); opaque_vgpr(stg_rb); }
        unsigned char* const stg_base = smem + H0_OFF + HALO_B;
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.y + (int64_t)e_n * a.H * a.W * a.out_stride), 0, (unsigned)a.H * a.W * a.out_stride * 2u, 0x00020000);
        const __amdgpu_buffer_rsrc_t ry_lo = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((PREC ? a.y_lo : a.y) + (int64_t)e_n * a.H * a.W * a.out_stride), 0, (unsigned)a.H * a.W * a.out_stride * 2u, 0x00020000);
        constexpr bool want_stats = STATS;              // BatchNorm partial sums: a template flag (no per-element selects)
        float bv[2] = {0.f, 0.f};
        if (!PLAIN) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int co = e_n0 + j * 32 + l31;
                bv[j] = (a.bias != nullptr && co < a.Cout) ? a.bias[co] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int prow0 = (wave * MI + i) * 32;
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
                        f32x2 vv = {v0, v1};
                        if (!FULL) vv *= f32x2{w0, w1};
                        s1v[j] += vv;
                        s2v[j] += vv * vv;
                    }
                    if (!PLAIN) {
                        v0 += bv[j];
                        v1 += bv[j];
                        v0 = v0 > 0.f ? v0 : v0 * neg_slope;
                        v1 = v1 > 0.f ? v1 : v1 * neg_slope;
                    }
                    const unsigned int own = Elem<DT>::pack2(v0, v1);
                    const unsigned int oth = dpp_xor1(own);
                    const unsigned int pkv = __builtin_amdgcn_perm(oth, own, psel);
                    const int row = rowa + (odd ? 1 : 0);
                    if (Q8) *reinterpret_cast<unsigned int*>(stg_base + stg_b + (unsigned)((row * C3_LDR + j * 32) * 2)) = pkv;
                    else *reinterpret_cast<unsigned int*>(stg + row * C3_LDR + j * 32 + (l31 & ~1)) = pkv;
                    if (PREC) {                               // lo = 16-bit(value - hi): the pair carries ~22 bits
                        const float l0 = v0 - Elem<DT>::to_f((unsigned short)(own & 0xffffu));
                        const float l1 = v1 - Elem<DT>::to_f((unsigned short)(own >> 16));
                        const unsigned int own_l = Elem<DT>::pack2(l0, l1);
                        const unsigned int oth_l = dpp_xor1(own_l);
                        const unsigned int pk_l = __builtin_amdgcn_perm(oth_l, own_l, psel);
                        if (Q8) *reinterpret_cast<unsigned int*>(stg_base + stg_b + (unsigned)((NWV * STG_EL + row * C3_LDR + j * 32) * 2)) = pk_l;
                        else *reinterpret_cast<unsigned int*>(stg_lo + row * C3_LDR + j * 32 + (l31 & ~1)) = pk_l;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            uint4 sv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                sv[q] = Q8 ? *reinterpret_cast<const uint4*>(stg_base + stg_rb + (unsigned)(q * 8 * C3_LDR * 2))
                           : *reinterpret_cast<const uint4*>(stg + (q * 8 + (lane >> 3)) * C3_LDR + (lane & 7) * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = prow0 + q * 8 + (lane >> 3);
                const int gy = e_y0 + (p >> TWS), gx = e_x0 + (p & (TW - 1));
                const int co = e_n0 + (lane & 7) * 8;
                const bool ok = (FULL || (gy < a.H && gx < a.W)) && co < a.Cout;
                const unsigned off = ok ? (unsigned)(((gy * a.W + gx) * a.out_stride + a.out_coff + co) * 2) : VOOB;
                u32x4 d;
                d[0] = sv[q].x; d[1] = sv[q].y; d[2] = sv[q].z; d[3] = sv[q].w;
                if (dbg & 64) asm volatile("" :: "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(off));      // ablation: no global stores
                else __builtin_amdgcn_raw_buffer_store_b128(d, ry, off, 0, GS_OUT_AUX);
                if (PREC) {
                    const uint4 lv = Q8 ? *reinterpret_cast<const uint4*>(stg_base + stg_rb + (unsigned)((NWV * STG_EL + q * 8 * C3_LDR) * 2))
                                        : *reinterpret_cast<const uint4*>(stg_lo + (q * 8 + (lane >> 3)) * C3_LDR + (lane & 7) * 8);
                    u32x4 dl;
                    dl[0] = lv.x; dl[1] = lv.y; dl[2] = lv.z; dl[3] = lv.w;
                    __builtin_amdgcn_raw_buffer_store_b128(dl, ry_lo, off, 0, GS_OUT_AUX);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    };
    const bool plain = (a.bias == nullptr && act == GS_ACT_NONE);
    const bool deferable = DEFER && act == GS_ACT_NONE;    // a bias rides along (bvb), an activation takes the immediate path
    // deferred form, item boundary: accumulators -> packed pairs + running statistics (nothing leaves the registers).  A bias
    // (Conv3d of the 3-D U-Net, unet3d.py:28-31) is added after the statistics, as on the immediate path; the block keeps its
    // cout tile, so the lane's two bias values are loaded once.
    float bvb[2] = {0.f, 0.f};
    auto convert_item = [&](const Item& itc) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v0 = acc[i][j][2 * m], v1 = acc[i][j][2 * m + 1];
                    if (STATS) {
                        const f32x2 vv = {v0, v1};
                        s1v[j] += vv;
                        s2v[j] += vv * vv;
                    }
                    v0 += bvb[j];
                    v1 += bvb[j];
                    unsigned own = Elem<DT>::pack2(v0, v1);
                    // (quad form: pinned here -- hipcc otherwise sinks the conversions to their uses in the next item's stages and
                    // carries the 128 fp32 values there in a second accumulator set: 560 registers)
                    if (MI == 4) opaque_vgpr(own);
                    pk[i][j][DEFER ? m : 0] = own;
                    if (PREC)                                 // lo = 16-bit(value - hi): the pair carries ~22 bits
                        pkl[i][j][(DEFER && PREC) ? m : 0] = Elem<DT>::pack2(v0 - Elem<DT>::to_f((unsigned short)(own & 0xffffu)),
                                                                             v1 - Elem<DT>::to_f((unsigned short)(own >> 16)));
                }
        int e_y0 = itc.y0, e_x0 = itc.x0;
        asm volatile("" : "+s"(e_y0), "+s"(e_x0));
        // pixel (patch row 2*wave + i, column 8*(p & 3) + lane / 8), couts n0 + 8*(lane & 7): i and p ride in the scalar offset
        // (a partial cout tile: the lanes beyond Cout carry the out-of-range offset, their stores are dropped)
        pend_voff = (itc.n0 + (lane & 7) * 8 < a.Cout)
                        ? (unsigned)((((e_y0 + MI * wave) * a.W + e_x0 + (lane >> 3)) * a.out_stride + a.out_coff + itc.n0 + (lane & 7) * 8) * 2)
                        : VOOB;
        pend_n = itc.n;
        pend = true;
    };

    // ---- items: same numbering and XCD-aware start as conv3x3_big_kernel ----
    int it = a.xcd_order ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    if (it >= nitems) return;
    const int it_first = it;
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
    auto make_item = [&]() __attribute__((always_inline)) {
        Item r;
        r.n = dg3; r.y0 = dg2 * TH; r.x0 = dg1 * TW; r.n0 = dg0 * BN;
        r.mt = (dg3 * tiles_y8 + (r.y0 >> 3)) * a.tiles_x + dg1;
        return r;
    };
    auto advance_item = [&]() __attribute__((always_inline)) {
        dg0 += st0; int c = dg0 >= a.ntn ? 1 : 0; dg0 -= c * a.ntn;
        dg1 += st1 + c; c = dg1 >= a.tiles_x ? 1 : 0; dg1 -= c * a.tiles_x;
        dg2 += st2 + c; c = dg2 >= a.tiles_y ? 1 : 0; dg2 -= c * a.tiles_y;
        dg3 += st3 + c;
        return make_item();
    };
    auto image_rsrc = [&](int n) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (int64_t)n * a.H * a.W * a.in_stride), 0, img_bytes, 0x00020000);
    };
    Item cur = make_item();
    const int block_n0 = cur.n0;                           // DEFER: the block's cout tile (st0 == 0: host-guaranteed)
    if (DEFER && a.bias != nullptr) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = block_n0 + j * 32 + l31;
            bvb[j] = co < a.Cout ? a.bias[co] : 0.f;
        }
    }
    if (Q8) {
        constexpr int LS = Q8Shift<DT>::v;
        q8_sa = (127 - GS_Q8_XH_EXP - (h == 0 ? LS : 0)) & 0xff;               // lanes 0..31: x_lo8's block, lanes 32..63: x_hi8's
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = block_n0 + j * 32 + l31;
            const int e = co < a.Cout ? a.wexp[co] : 0;
            q8_sb[j] = (127 - e - (h == 0 ? 0 : LS)) & 0xff;                  // lanes 0..31: w_hi8's block, lanes 32..63: w_lo8's
        }
    }
    // WRES: a 64-channel input is two stages -- the weight slabs of stage 0 / stage 1 of the block's cout tile sit in buffer 0 /
    // buffer 1, and every item would fetch the same 2 x 36 KB again.  The block keeps its cout tile, so they are fetched ONCE:
    // 73 of the 215 KB an item moves through the CU's memory pipe (halo 78 KB in, 64 KB out).  With the immediate epilogue this
    // measured +-2 % (the epilogue was the longer pole: 194 vs 198 us on 64->64 @256^2); with the epilogue out of the way the
    // weight traffic is what these layers wait for (without epilogue: 171 us -> 138 us with resident weights).
    // (The immediate epilogue's staging overlays halo 1 exactly -- 8 x 4.5 KB + 4 KB of partial sums = 40 KB -- so a partial
    // patch on that path does not disturb the slabs; the pair form overlays weights 1 and never takes this path.)
    const bool wres = DEFER && !PREC && nstage == 2 && a.ndz == 1 && st0 == 0 && !(dbg & 32);      // (PREC: the immediate path of partial patches overlays weights 1)
    bool first_item = true;
    setup_item(cur);
    {
        const Src s0 = stage_src(cur.n, 0);
        const __amdgpu_buffer_rsrc_t rx0 = image_rsrc(s0.n);
#pragma unroll
        for (int k = 0; k < NWP + HJ; ++k) issue_piece(k, rx0, s0.sc, s0.wsc, 0u, s0.hkill, 0u);   // stage 0 -> buffer 0
    }
    for (;;) {
        const int nit = it + gridDim.x;
        const bool more_items = nit < nitems;
        Item nxt = cur;
        if (more_items) nxt = advance_item();
        PH(5);
        if (MI == 4) {
            // Quad form: the item's first stage pair is peeled and ALWAYS carries the deferred passes (with nothing pending the
            // stores go to the out-of-range offset and are dropped): one straight line of stage bodies and one plain loop -- with
            // the 2-tile forms' runtime choice between pending / not-pending bodies hipcc assigned the 128 accumulator registers
            // differently per body and reconciled them with 128 moves at every join (a second accumulator set: 560 registers).
            const bool last0 = nstage <= 2;
            stage_sync();
            {
                const Src s1 = stage_src(cur.n, 1);
                run_stage(std::integral_constant<int, 0>{}, std::true_type{}, std::integral_constant<int, 1>{}, image_rsrc(s1.n), s1, 0u, nxt,
                          wres && !first_item);
            }
            stage_sync();
            if (last0) {
#pragma unroll
                for (int j = 0; j < HJ; ++j) hv[j] = hvn[j];
                wv = wvn;
            }
            {
                const Src s2 = stage_src(last0 ? nxt.n : cur.n, last0 ? 0 : 2);
                run_stage(std::integral_constant<int, 1>{}, std::false_type{}, std::integral_constant<int, 2>{}, image_rsrc(s2.n), s2,
                          (last0 && !more_items) ? VOOB : 0u, nxt, wres && last0);
            }
            pend = false;
            pend_voff = VOOB;
            for (int sp = 2; sp < nstage; sp += 2) {
                const bool last = sp + 2 >= nstage;
                stage_sync();
                {
                    const Src s1 = stage_src(cur.n, sp + 1);
                    run_stage(std::integral_constant<int, 0>{}, std::false_type{}, std::integral_constant<int, 0>{}, image_rsrc(s1.n), s1, 0u, nxt);
                }
                stage_sync();
                if (last) {
#pragma unroll
                    for (int j = 0; j < HJ; ++j) hv[j] = hvn[j];
                    wv = wvn;
                }
                const Src s2 = stage_src(last ? nxt.n : cur.n, last ? 0 : sp + 2);
                run_stage(std::integral_constant<int, 1>{}, std::false_type{}, std::integral_constant<int, 0>{}, image_rsrc(s2.n), s2,
                          (last && !more_items) ? VOOB : 0u, nxt);
            }
        } else
        for (int sp = 0; sp < nstage; sp += 2) {
            const bool last = sp + 2 >= nstage;
            stage_sync();
            PH(0);
            const Src s1 = stage_src(cur.n, sp + 1);
            if (sp == 0) {
                const bool ws0 = wres && !first_item;      // stage 1's slabs are in buffer 1 since the first item
                if (DEFER && pend) {
                    run_stage(std::integral_constant<int, 0>{}, std::true_type{}, std::integral_constant<int, 1>{}, image_rsrc(s1.n), s1, 0u, nxt, ws0);
                    pend = false;
                    pend2 = MI == 4;                       // quad form: passes 8..15 ride on the item's second stage
                } else {
                    run_stage(std::integral_constant<int, 0>{}, std::true_type{}, std::integral_constant<int, 0>{}, image_rsrc(s1.n), s1, 0u, nxt, ws0);
                }
            } else if (Q8 && (sp % nchunk) >= a.q8_c0) {
                run_stage8(std::integral_constant<int, 0>{}, image_rsrc(s1.n), s1, 0u, nxt);
                PH(6);
            } else {
                run_stage(std::integral_constant<int, 0>{}, std::false_type{}, std::integral_constant<int, 0>{}, image_rsrc(s1.n), s1, 0u, nxt);
            }
            PH(1);
            stage_sync();
            PH(2);
            if (last) {                                    // from here on the pieces belong to the next item
#pragma unroll
                for (int j = 0; j < HJ; ++j) hv[j] = hvn[j];
                wv = wvn;
            }
            const Src s2 = stage_src(last ? nxt.n : cur.n, last ? 0 : sp + 2);
            if (Q8 && ((sp + 1) % nchunk) >= a.q8_c0) {
                run_stage8(std::integral_constant<int, 1>{}, image_rsrc(s2.n), s2, (last && !more_items) ? VOOB : 0u, nxt);
                PH(7);
            } else if (MI == 4 && pend2) {
                run_stage(std::integral_constant<int, 1>{}, std::false_type{}, std::integral_constant<int, 2>{}, image_rsrc(s2.n), s2,
                          (last && !more_items) ? VOOB : 0u, nxt, wres && last);
                pend2 = false;
            } else
            run_stage(std::integral_constant<int, 1>{}, std::false_type{}, std::integral_constant<int, 0>{}, image_rsrc(s2.n), s2,
                      (last && !more_items) ? VOOB : 0u, nxt, wres && last);      // the next item's stage-0 slabs are in buffer 0
            PH(3);
        }
        const bool full = (cur.y0 + TH <= a.H) && (cur.x0 + TW <= a.W);
        if (deferable && full && !(dbg & 4)) {
            convert_item(cur);                             // registers only: no barrier, no LDS
        } else {
            if (DEFER) {                                   // (nothing is pending here; tells the register allocator so)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int m = 0; m < NPK; ++m) { pk[i][j][m] = 0u; if (PREC) pkl[i][j][m < NPKL ? m : 0] = 0u; }
            }
            __builtin_amdgcn_s_barrier();              // every wave has left the second buffer: staging may overlay it
            asm volatile("" ::: "memory");
            if (!(dbg & 4)) {
                if (Q8) {                                  // (a "q" stage has neither bias nor activation: no code for them in this kernel)
                    if (full) epilogue_t(cur, std::true_type{}, std::true_type{});
                    else epilogue_t(cur, std::true_type{}, std::false_type{});
                } else if (plain && full) epilogue_t(cur, std::true_type{}, std::true_type{});
                else epilogue_t(cur, std::false_type{}, std::false_type{});
            }
            // (no barrier behind it: the staging area is overwritten by DMA pieces only after the next stage hand-over)
        }
        PH(4);
        if (!more_items) break;
        it = nit;
        cur = nxt;
        first_item = false;
    }
#ifdef GS_C3_PHASE_TIMING
    if (blockIdx.x == 0 && lane == 0 && a.bnp != nullptr) {          // the partials buffer doubles as sink (rows 300..)
        for (int i = 0; i < 8; ++i) a.bnp[300 * 2 * a.Cout + wave * 8 + i] = (float)ph[i];
    }
#endif
#undef PH
    {
        if (DEFER && pend) defer_flush();
        if (STATS) {
            // the block's partial sums: one row per block, written once (rows = grid / ntn; block b with cout tile b % ntn
            // writes columns n0 .. n0 + 63 of row it_first / ntn)
            __syncthreads();                               // the stage buffers are free: every wave has left the K loop
            float* redb = reinterpret_cast<float*>(smem);  // [NWV][2][BN]
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float t1 = s1v[j].x + s1v[j].y, t2 = s2v[j].x + s2v[j].y;
                t1 += __shfl_xor(t1, 32, 64);
                t2 += __shfl_xor(t2, 32, 64);
                if (h == 0) {
                    redb[(wave * 2 + 0) * BN + j * 32 + l31] = t1;
                    redb[(wave * 2 + 1) * BN + j * 32 + l31] = t2;
                }
            }
            __syncthreads();
            if (t < 2 * BN) {
                const int stat = t / BN, c = t % BN;
                if (block_n0 + c < a.Cout) {
                    float v = 0.f;
#pragma unroll
                    for (int m = 0; m < NWV; ++m) v += redb[(m * 2 + stat) * BN + c];     // wave order: deterministic
                    a.bnp[(int64_t)(it_first / a.ntn) * 2 * a.Cout + stat * a.Cout + block_n0 + c] = v;
                }
            }
        }
    }
}

}  // namespace

// grid of the deferred form: at most max_blocks, a multiple of the cout-tile count (a block keeps its tile), at least one
// item per block.  It is also the number of BatchNorm partial rows times ntn.
int c3_dma_grid(int nitems, int ntn, int max_blocks) {
    int g = nitems < max_blocks ? nitems : max_blocks;
    g -= g % ntn;
    return g < ntn ? ntn : g;
}

int c3_dma_launch(C3Args& a, int waves, bool prec, int dtype, int grid_blocks, hipStream_t s) {
    dim3 grid(grid_blocks);
    const bool stats = a.bnp != nullptr;
    if (waves == 44) {                                     // the quad form (4 waves x 4 pixel rows: 16x32-pixel items): 16-bit forward / data gradient
        if (prec || a.q8_c0 > 0) return -1;
        if (dtype == GS_F16) {
            if (stats) conv3x3_dma_kernel<GS_F16, 4, true, false, true, false, 4><<<grid, 256, 0, s>>>(a);
            else conv3x3_dma_kernel<GS_F16, 4, false, false, true, false, 4><<<grid, 256, 0, s>>>(a);
        } else {
            if (stats) conv3x3_dma_kernel<GS_BF16, 4, true, false, true, false, 4><<<grid, 256, 0, s>>>(a);
            else conv3x3_dma_kernel<GS_BF16, 4, false, false, true, false, 4><<<grid, 256, 0, s>>>(a);
        }
        return 0;
    }
    if (a.q8_c0 > 0) {                                     // FP8 correction stages: pair forward, fp16 (the plans use "q" for fp16 only)
        if (!prec || dtype != GS_F16 || a.wexp == nullptr) return -1;
        if (waves == 8) {
            if (stats) conv3x3_dma_kernel<GS_F16, 8, true, true, false, true><<<grid, 512, 0, s>>>(a);
            else conv3x3_dma_kernel<GS_F16, 8, false, true, false, true><<<grid, 512, 0, s>>>(a);
        } else {                                           // 4 waves: the deferred pair epilogue
            if (stats) conv3x3_dma_kernel<GS_F16, 4, true, true, true, true><<<grid, 256, 0, s>>>(a);
            else conv3x3_dma_kernel<GS_F16, 4, false, true, true, true><<<grid, 256, 0, s>>>(a);
        }
        return 0;
    }
#define C3_DMA_GO(DT, NWV)                                                                                    \
    do {                                                                                                      \
        if (prec) {                                            /* 4 waves: the deferred pair epilogue */      \
            if (stats) conv3x3_dma_kernel<DT, NWV, true, true, NWV == 4><<<grid, 64 * NWV, 0, s>>>(a);        \
            else conv3x3_dma_kernel<DT, NWV, false, true, NWV == 4><<<grid, 64 * NWV, 0, s>>>(a);             \
        } else {                                                                                              \
            if (stats) conv3x3_dma_kernel<DT, NWV, true, false, true><<<grid, 64 * NWV, 0, s>>>(a);           \
            else conv3x3_dma_kernel<DT, NWV, false, false, true><<<grid, 64 * NWV, 0, s>>>(a);                \
        }                                                                                                     \
    } while (0)
    if (dtype == GS_F16) {
        if (waves == 8) C3_DMA_GO(GS_F16, 8);
        else C3_DMA_GO(GS_F16, 4);
    } else {
        if (waves == 8) C3_DMA_GO(GS_BF16, 8);
        else C3_DMA_GO(GS_BF16, 4);
    }
#undef C3_DMA_GO
    return 0;
}
