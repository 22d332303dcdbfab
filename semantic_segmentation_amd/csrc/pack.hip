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
