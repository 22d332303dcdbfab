// Weight packing / gradient unpacking and fp32-NCHW <-> 16-bit-NHWC layout converters (all tiny or
// HBM-bound elementwise kernels).  The nn.Parameters stay fp32 in the reference's own layouts
// ([Cout][Cin][kh][kw] for Conv2d, [Cin][Cout][kh][kw] for ConvTranspose2d) so checkpoints, optimisers
// and Betty's in-place perturbations keep working; the MFMA engine consumes K-major 16-bit packs.
#include "common.hpp"

namespace {

template <int DT>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, unsigned short* wf,
                                                          unsigned short* wd, int Cout, int Cin, int T, int transposed) {
    const int64_t total = (int64_t)Cout * Cin * T;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        // i enumerates the fwd pack [t][co][ci]
        const int ci = (int)(i % Cin);
        int64_t r = i / Cin;
        const int co = (int)(r % Cout);
        const int t = (int)(r / Cout);
        const int64_t src = transposed ? (((int64_t)ci * Cout + co) * T + t) : (((int64_t)co * Cin + ci) * T + t);
        const unsigned short v = Elem<DT>::from_f(w[src]);
        if (wf) wf[i] = v;
        if (wd) wd[((int64_t)t * Cin + ci) * Cout + co] = v;
    }
}

// Row-wise variant for the common tap counts: a lane owns one (a, b) pair of the source [A][B][T] and reads its T
// contiguous floats back to back (T loads in flight, the wave covers one contiguous run when b is the fast index), then
// writes one pack with the fast index along the lanes (128-byte runs).  blockIdx.y picks the pack: 0 -> P_ab [T][A][B]
// (lanes along b: perfectly coalesced source), 1 -> P_ba [T][B][A] (lanes along a: 36-byte segments of the source).
// The element-wise kernel above reads the source with a stride of T elements: ~T x over-fetch on the big layers.
template <int DT, int T>
__global__ __launch_bounds__(256) void pack_weight_rows_kernel(const float* __restrict__ w, unsigned short* p_ab,
                                                               unsigned short* p_ba, int A, int B) {
    const int64_t n = (int64_t)A * B;
    const bool ba = blockIdx.y == 1;
    unsigned short* dst = ba ? p_ba : p_ab;
    if (dst == nullptr) return;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int a, b;
        if (!ba) { b = (int)(i % B); a = (int)(i / B); }
        else { a = (int)(i % A); b = (int)(i / A); }
        const float* src = w + ((int64_t)a * B + b) * T;
        float v[T];
#pragma unroll
        for (int t = 0; t < T; ++t) v[t] = src[t];
#pragma unroll
        for (int t = 0; t < T; ++t) dst[(int64_t)t * n + i] = Elem<DT>::from_f(v[t]);
    }
}

// All conv weights of a network in ONE launch (a U-Net step re-packs 21 tensors after every optimiser step: 21 launches of
// 5-8 us each were launch-bound).  Descriptor d owns blocks [first[d], first[d+1]); inside, the row-wise scheme above.
constexpr int PACK_MULTI_MAX = 32;
struct PackMultiArgs {
    const float* w[PACK_MULTI_MAX];
    unsigned short* p_ab[PACK_MULTI_MAX];
    unsigned short* p_ba[PACK_MULTI_MAX];
    int A[PACK_MULTI_MAX], B[PACK_MULTI_MAX], T[PACK_MULTI_MAX], first[PACK_MULTI_MAX + 1];
    int n;
};

template <int DT, int T>
__device__ __forceinline__ void pack_rows_body(const float* __restrict__ w, unsigned short* dst, int A, int B, bool ba,
                                               int block, int nblocks) {
    const int64_t n = (int64_t)A * B;
    for (int64_t i = (int64_t)block * 256 + threadIdx.x; i < n; i += (int64_t)nblocks * 256) {
        int a, b;
        if (!ba) { b = (int)(i % B); a = (int)(i / B); }
        else { a = (int)(i % A); b = (int)(i / A); }
        const float* src = w + ((int64_t)a * B + b) * T;
        float v[T];
#pragma unroll
        for (int t = 0; t < T; ++t) v[t] = src[t];
#pragma unroll
        for (int t = 0; t < T; ++t) dst[(int64_t)t * n + i] = Elem<DT>::from_f(v[t]);
    }
}

template <int DT>
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const PackMultiArgs a) {
    int d = 0;
    while (d + 1 < a.n && (int)blockIdx.x >= a.first[d + 1]) ++d;
    const bool ba = blockIdx.y == 1;
    unsigned short* dst = ba ? a.p_ba[d] : a.p_ab[d];
    if (dst == nullptr) return;
    const int block = blockIdx.x - a.first[d], nblocks = a.first[d + 1] - a.first[d];
    if (a.T[d] == 9) pack_rows_body<DT, 9>(a.w[d], dst, a.A[d], a.B[d], ba, block, nblocks);
    else if (a.T[d] == 27) pack_rows_body<DT, 27>(a.w[d], dst, a.A[d], a.B[d], ba, block, nblocks);      // Conv3d 3x3x3
    else if (a.T[d] == 8) pack_rows_body<DT, 8>(a.w[d], dst, a.A[d], a.B[d], ba, block, nblocks);        // ConvTranspose3d k2 s2
    else pack_rows_body<DT, 4>(a.w[d], dst, a.A[d], a.B[d], ba, block, nblocks);
}

__global__ __launch_bounds__(256) void unpack_wgrad_kernel(const float* __restrict__ dw, float* __restrict__ grad,
                                                           int A, int B, int T, int transposed, float gscale) {
    const int64_t total = (int64_t)A * B * T;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        // i enumerates the destination
        const int t = (int)(i % T);
        int64_t r = i / T;
        int a, b;
        if (!transposed) { b = (int)(r % B); a = (int)(r / B); }
        else { a = (int)(r % A); b = (int)(r / A); }
        grad[i] = dw[((int64_t)t * A + a) * B + b] * gscale;
    }
}

template <int DT>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, unsigned short* dst, int N,
                                                           int C, int64_t HW, int ds, int dc) {
    const int64_t total = (int64_t)N * HW * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const int64_t n = pix / HW, hw = pix - n * HW;
        dst[pix * ds + dc + c] = Elem<DT>::from_f(src[(n * C + c) * HW + hw]);
    }
}

template <int DT>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const unsigned short* __restrict__ src, int ss, int sc,
                                                           float* __restrict__ dst, int N, int C, int64_t HW, float gscale) {
    const int64_t total = (int64_t)N * HW * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        // i enumerates the destination [n][c][hw]
        const int64_t hw = i % HW;
        int64_t r = i / HW;
        const int c = (int)(r % C);
        const int64_t n = r / C;
        dst[i] = Elem<DT>::to_f(src[(n * HW + hw) * ss + sc + c]) * gscale;
    }
}

inline int ew_blocks(int64_t n) {
    int64_t b = cdiv64(n, 256);
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int gs_pack_weight(const float* w, void* w_fwd, void* w_dgrad, int Cout, int Cin, int taps, int transposed,
                              int dtype, void* stream) {
    GS_CHECK_ARG(w && (w_fwd || w_dgrad) && Cout > 0 && Cin > 0 && taps > 0, "gs_pack_weight: bad arguments");
    const int64_t total = (int64_t)Cout * Cin * taps;
    hipStream_t s = (hipStream_t)stream;
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_pack_weight: bad dtype");
    if (taps == 1 || taps == 4 || taps == 9 || taps == 16 || taps == 27) {
        // source [A][B][T]: Conv2d A=Cout,B=Cin (P_ab = fwd pack, P_ba = dgrad pack); ConvTranspose2d A=Cin,B=Cout
        // (P_ab = dgrad pack [T][Cin][Cout], P_ba = fwd pack [T][Cout][Cin])
        const int A = transposed ? Cin : Cout, B = transposed ? Cout : Cin;
        unsigned short* p_ab = (unsigned short*)(transposed ? w_dgrad : w_fwd);
        unsigned short* p_ba = (unsigned short*)(transposed ? w_fwd : w_dgrad);
        dim3 grid(ew_blocks((int64_t)A * B), 2);
#define GS_PACK_ROWS(DT, TT) pack_weight_rows_kernel<DT, TT><<<grid, 256, 0, s>>>(w, p_ab, p_ba, A, B)
#define GS_PACK_T(DT)                                    \
        switch (taps) {                                  \
            case 1: GS_PACK_ROWS(DT, 1); break;          \
            case 4: GS_PACK_ROWS(DT, 4); break;          \
            case 9: GS_PACK_ROWS(DT, 9); break;          \
            case 16: GS_PACK_ROWS(DT, 16); break;        \
            default: GS_PACK_ROWS(DT, 27); break;        \
        }
        if (dtype == GS_F16) { GS_PACK_T(GS_F16) } else { GS_PACK_T(GS_BF16) }
#undef GS_PACK_T
#undef GS_PACK_ROWS
        GS_CHECK_LAUNCH("gs_pack_weight");
        return GS_OK;
    }
    if (dtype == GS_F16)
        pack_weight_kernel<GS_F16><<<ew_blocks(total), 256, 0, s>>>(w, (unsigned short*)w_fwd, (unsigned short*)w_dgrad,
                                                                    Cout, Cin, taps, transposed);
    else if (dtype == GS_BF16)
        pack_weight_kernel<GS_BF16><<<ew_blocks(total), 256, 0, s>>>(w, (unsigned short*)w_fwd, (unsigned short*)w_dgrad,
                                                                     Cout, Cin, taps, transposed);
    else GS_CHECK_ARG(false, "gs_pack_weight: bad dtype");
    GS_CHECK_LAUNCH("gs_pack_weight");
    return GS_OK;
}

extern "C" int gs_pack_weight_multi(int n, const GsPackDesc* descs, int dtype, void* stream) {
    GS_CHECK_ARG(n > 0 && descs != nullptr, "gs_pack_weight_multi: no descriptors");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_pack_weight_multi: bad dtype");
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < n; base += PACK_MULTI_MAX) {
        PackMultiArgs a;
        a.n = n - base < PACK_MULTI_MAX ? n - base : PACK_MULTI_MAX;
        int blocks = 0;
        for (int i = 0; i < a.n; ++i) {
            const GsPackDesc& d = descs[base + i];
            GS_CHECK_ARG(d.w && (d.w_fwd || d.w_dgrad) && d.Cout > 0 && d.Cin > 0 &&
                         (d.taps == 9 || d.taps == 4 || d.taps == 27 || d.taps == 8),
                         "gs_pack_weight_multi: descriptor %d: needs w, a pack, positive dims and 9 / 4 (2-D) or 27 / 8 (3-D) taps", base + i);
            // source [A][B][T]: Conv2d A=Cout,B=Cin (P_ab = fwd pack, P_ba = dgrad pack); ConvTranspose2d A=Cin,B=Cout
            a.w[i] = d.w;
            a.A[i] = d.transposed ? d.Cin : d.Cout;
            a.B[i] = d.transposed ? d.Cout : d.Cin;
            a.T[i] = d.taps;
            a.p_ab[i] = (unsigned short*)(d.transposed ? d.w_dgrad : d.w_fwd);
            a.p_ba[i] = (unsigned short*)(d.transposed ? d.w_fwd : d.w_dgrad);
            a.first[i] = blocks;
            int b = ew_blocks((int64_t)a.A[i] * a.B[i]);
            if (b > 512) b = 512;
            blocks += b;
        }
        a.first[a.n] = blocks;
        dim3 grid(blocks, 2);
        if (dtype == GS_F16) pack_weight_multi_kernel<GS_F16><<<grid, 256, 0, s>>>(a);
        else pack_weight_multi_kernel<GS_BF16><<<grid, 256, 0, s>>>(a);
        GS_CHECK_LAUNCH("gs_pack_weight_multi");
    }
    return GS_OK;
}

extern "C" int gs_unpack_wgrad(const float* dw, float* grad, int A, int B, int taps, int transposed, float gscale,
                               void* stream) {
    GS_CHECK_ARG(dw && grad && A > 0 && B > 0 && taps > 0, "gs_unpack_wgrad: bad arguments");
    const int64_t total = (int64_t)A * B * taps;
    unpack_wgrad_kernel<<<ew_blocks(total), 256, 0, (hipStream_t)stream>>>(dw, grad, A, B, taps, transposed, gscale);
    GS_CHECK_LAUNCH("gs_unpack_wgrad");
    return GS_OK;
}

extern "C" int gs_nchw_to_nhwc(const float* src, void* dst, int N, int C, int H, int W, int dst_pix_stride,
                               int dst_coff, int dtype, void* stream) {
    GS_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && dst_pix_stride >= dst_coff + C,
                 "gs_nchw_to_nhwc: bad arguments");
    const int64_t total = (int64_t)N * C * H * W;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GS_F16)
        nchw_to_nhwc_kernel<GS_F16><<<ew_blocks(total), 256, 0, s>>>(src, (unsigned short*)dst, N, C, (int64_t)H * W,
                                                                     dst_pix_stride, dst_coff);
    else if (dtype == GS_BF16)
        nchw_to_nhwc_kernel<GS_BF16><<<ew_blocks(total), 256, 0, s>>>(src, (unsigned short*)dst, N, C, (int64_t)H * W,
                                                                      dst_pix_stride, dst_coff);
    else GS_CHECK_ARG(false, "gs_nchw_to_nhwc: bad dtype");
    GS_CHECK_LAUNCH("gs_nchw_to_nhwc");
    return GS_OK;
}

extern "C" int gs_nhwc_to_nchw(const void* src, int src_pix_stride, int src_coff, float* dst, int N, int C, int H,
                               int W, float gscale, int dtype, void* stream) {
    GS_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && src_pix_stride >= src_coff + C,
                 "gs_nhwc_to_nchw: bad arguments");
    const int64_t total = (int64_t)N * C * H * W;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GS_F16)
        nhwc_to_nchw_kernel<GS_F16><<<ew_blocks(total), 256, 0, s>>>((const unsigned short*)src, src_pix_stride, src_coff,
                                                                     dst, N, C, (int64_t)H * W, gscale);
    else if (dtype == GS_BF16)
        nhwc_to_nchw_kernel<GS_BF16><<<ew_blocks(total), 256, 0, s>>>((const unsigned short*)src, src_pix_stride, src_coff,
                                                                      dst, N, C, (int64_t)H * W, gscale);
    else GS_CHECK_ARG(false, "gs_nhwc_to_nchw: bad dtype");
    GS_CHECK_LAUNCH("gs_nhwc_to_nchw");
    return GS_OK;
}
