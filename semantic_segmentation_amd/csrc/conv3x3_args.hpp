// Shared by conv3x3.hip (halo / persistent / big K-step kernels, launch logic) and conv3x3_dma.hip (the LDS-DMA kernel).
#pragma once
#include "common.hpp"

struct C3Args {
    const unsigned short* x;
    const unsigned short* w;      // [9][Cout][Cin]
    unsigned short* y;
    const float* bias;
    float* bnp;                   // [npatches][2][Cout]
    int act;
    int N, H, W, Cin, in_stride, in_coff, Cout, out_stride, out_coff;
    int tiles_x, tiles_y, ntn, nblocks;
    int tap_dy[9], tap_dx[9];
    // 3-D (3x3x3) convolution: images are the N = volumes*D depth slices; stage (dz index, channel chunk) reads slice
    // n + tap_dz and the weight slots [dzi*9 .. dzi*9+8].  2-D: D = 1, ndz = 1, tap_dz = {0}.
    int D, ndz, tap_dz[3];
    // precise mode (conv3x3_big_kernel<.., PREC>): the K extent Cin is a concatenation of segments over the same input
    // channels -- K chunk c reads input chunk (c >= in_wrap ? c - in_wrap : c) -- and the result is stored as a 16-bit
    // hi/lo pair: hi at y, lo = 16-bit(value - hi) at y_lo (same stride / offset).
    unsigned short* y_lo;
    int in_wrap;                  // in 64-channel chunks; 0 = no wrap
    int in_wrap_to = 0;           // the wrapped K chunks continue at this input chunk (64-channel chunks) instead of chunk 0
    int xcd_order;                // big kernel: XCD-aware item order (grid must be a multiple of 8)
    // "q" stages of the pair forward (conv3x3_dma_kernel<.., Q8>): per depth tap the first q8_c0 32-channel K stages are 16-bit
    // (x_hi . w_hi), the remaining ones FP8 correction stages reading the q planes of the input / of the pack (common.hpp); wexp[co]
    // = the power-of-two exponent of cout row co's e4m3 weight planes (gs_pack_weight_q8).  q8_c0 = 0: none.
    int q8_c0 = 0;
    const int* wexp = nullptr;
};

constexpr int C3_LDR = 72;

__device__ __forceinline__ unsigned int dpp_xor1(unsigned int v) {
    return (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, false);
}

// conv3x3_dma.hip: launches conv3x3_dma_kernel.  waves: 4 / 8; prec: pair output (y_lo, in_wrap); per_block_stats: see
// c3_dma_stat_rows().  The caller has filled every field of `a` except nblocks / tiles_y for the 8-wave form.
int c3_dma_launch(C3Args& a, int waves, bool prec, int dtype, int grid_blocks, hipStream_t s);      // (a.q8_c0 > 0: the Q8 form, f16 only)
// number of BatchNorm partial rows the non-pair DMA kernel writes: one per block and cout tile group (grid / ntn)
int c3_dma_grid(int nitems, int ntn, int max_blocks);
