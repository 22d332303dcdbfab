// Bilinear x2 up-sampling with align_corners=True (nn.Upsample of the `bilinear=True` U-Net, unet/unet_parts.py:49-50)
// on 16-bit NHWC tensors, forward and backward, written straight into / read from the channel slice of the concat
// buffer that torch.cat would build (unet_parts.py:58-67), including the F.pad offset.  HBM-bound.
//   src = dst * (in - 1) / (out - 1);  i0 = floor(src);  i1 = min(i0 + 1, in - 1);  l = src - i0    (ATen upsample_bilinear2d)
// Backward is the transposed interpolation in gather form: the weight of input i for output o is the hat function
// max(0, 1 - |o*r - i|), so each input pixel sums its <= 6 x 6 candidate outputs (deterministic, no atomics).
#include "common.hpp"

namespace {

struct UpArgs {
    const unsigned short* x; unsigned short* y;      // fwd: x -> y;  bwd: x = dY (read), y = dX (written)
    int N, IH, IW, C, xs, xc;                        // low-resolution tensor dims; strides/offsets of the tensor in `x`
    int OH, OW, ys, yc;                              // OH/OW: up-sampled size (2*IH, 2*IW); strides/offsets of `y`
    int PH, PW, oy0, ox0;                            // physical size of the high-resolution buffer and the pad offset
    float ry, rx;
};

template <int DT>
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const UpArgs a) {
    const int nch = a.C >> 3;
    const int total = a.N * a.OH * a.OW * nch;       // host guarantees < 2^31
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int ch = idx % nch;
        int p = idx / nch;
        const int ox = p % a.OW; p /= a.OW;
        const int oy = p % a.OH;
        const int n = p / a.OH;
        const float sy = oy * a.ry, sx = ox * a.rx;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < a.IH - 1), x1 = x0 + (x0 < a.IW - 1);
        const float ly = sy - y0, lx = sx - x0;
        const unsigned short* base = a.x + (int64_t)n * a.IH * a.IW * a.xs + a.xc + ch * 8;
        float v00[8], v01[8], v10[8], v11[8], o[8];
        unpack8<DT>(*reinterpret_cast<const uint4*>(base + (int64_t)(y0 * a.IW + x0) * a.xs), v00);
        unpack8<DT>(*reinterpret_cast<const uint4*>(base + (int64_t)(y0 * a.IW + x1) * a.xs), v01);
        unpack8<DT>(*reinterpret_cast<const uint4*>(base + (int64_t)(y1 * a.IW + x0) * a.xs), v10);
        unpack8<DT>(*reinterpret_cast<const uint4*>(base + (int64_t)(y1 * a.IW + x1) * a.xs), v11);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            o[i] = (1.f - ly) * ((1.f - lx) * v00[i] + lx * v01[i]) + ly * ((1.f - lx) * v10[i] + lx * v11[i]);
        const int64_t pix = ((int64_t)n * a.PH + oy + a.oy0) * a.PW + ox + a.ox0;
        *reinterpret_cast<uint4*>(a.y + pix * a.ys + a.yc + ch * 8) = pack8<DT>(o);
    }
}

// Forward on hi/lo pairs (the pair forward of the bilinear U-Net): the interpolation of the pair VALUES, stored as a pair again
// (x_lo / y_lo: the lo planes, same strides and offsets as the hi planes; y_lo may be NULL when no consumer reads it).
template <int DT>
__global__ __launch_bounds__(256) void upsample2x_fwd_pair_kernel(const UpArgs a, const unsigned short* __restrict__ x_lo,
                                                                  unsigned short* __restrict__ y_lo) {
    const int nch = a.C >> 3;
    const int total = a.N * a.OH * a.OW * nch;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int ch = idx % nch;
        int p = idx / nch;
        const int ox = p % a.OW; p /= a.OW;
        const int oy = p % a.OH;
        const int n = p / a.OH;
        const float sy = oy * a.ry, sx = ox * a.rx;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < a.IH - 1), x1 = x0 + (x0 < a.IW - 1);
        const float ly = sy - y0, lx = sx - x0;
        const int64_t boff = (int64_t)n * a.IH * a.IW * a.xs + a.xc + ch * 8;
        const int64_t o00 = boff + (int64_t)(y0 * a.IW + x0) * a.xs, o01 = boff + (int64_t)(y0 * a.IW + x1) * a.xs;
        const int64_t o10 = boff + (int64_t)(y1 * a.IW + x0) * a.xs, o11 = boff + (int64_t)(y1 * a.IW + x1) * a.xs;
        float v00[8], v01[8], v10[8], v11[8], o[8];
        join8<DT>(*reinterpret_cast<const uint4*>(a.x + o00), *reinterpret_cast<const uint4*>(x_lo + o00), v00);
        join8<DT>(*reinterpret_cast<const uint4*>(a.x + o01), *reinterpret_cast<const uint4*>(x_lo + o01), v01);
        join8<DT>(*reinterpret_cast<const uint4*>(a.x + o10), *reinterpret_cast<const uint4*>(x_lo + o10), v10);
        join8<DT>(*reinterpret_cast<const uint4*>(a.x + o11), *reinterpret_cast<const uint4*>(x_lo + o11), v11);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            o[i] = (1.f - ly) * ((1.f - lx) * v00[i] + lx * v01[i]) + ly * ((1.f - lx) * v10[i] + lx * v11[i]);
        const int64_t pix = ((int64_t)n * a.PH + oy + a.oy0) * a.PW + ox + a.ox0;
        uint4 hi, lo;
        split8<DT>(o, hi, lo);
        *reinterpret_cast<uint4*>(a.y + pix * a.ys + a.yc + ch * 8) = hi;
        if (y_lo) *reinterpret_cast<uint4*>(y_lo + pix * a.ys + a.yc + ch * 8) = lo;
    }
}

template <int DT>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const UpArgs a) {
    const int nch = a.C >> 3;
    const int total = a.N * a.IH * a.IW * nch;
    const float iry = a.ry > 0.f ? 1.f / a.ry : 0.f, irx = a.rx > 0.f ? 1.f / a.rx : 0.f;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int ch = idx % nch;
        int p = idx / nch;
        const int ix = p % a.IW; p /= a.IW;
        const int iy = p % a.IH;
        const int n = p / a.IH;
        // candidate outputs: |o*r - i| < 1  <=>  (i-1)/r < o < (i+1)/r   (r == 0: a 1-pixel input feeds every output)
        int oya = 0, oyb = a.OH - 1, oxa = 0, oxb = a.OW - 1;
        if (a.ry > 0.f) { oya = max(0, (int)floorf((iy - 1) * iry)); oyb = min(a.OH - 1, (int)ceilf((iy + 1) * iry)); }
        if (a.rx > 0.f) { oxa = max(0, (int)floorf((ix - 1) * irx)); oxb = min(a.OW - 1, (int)ceilf((ix + 1) * irx)); }
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        for (int oy = oya; oy <= oyb; ++oy) {
            // the same fp32 arithmetic as the forward pass, so forward and backward weights agree exactly
            const float sy = oy * a.ry;
            const int y0 = (int)sy, y1 = y0 + (y0 < a.IH - 1);
            const float ly = sy - y0;
            const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int ox = oxa; ox <= oxb; ++ox) {
                const float sx = ox * a.rx;
                const int x0 = (int)sx, x1 = x0 + (x0 < a.IW - 1);
                const float lx = sx - x0;
                const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
                if (wx == 0.f) continue;
                const int64_t pix = ((int64_t)n * a.PH + oy + a.oy0) * a.PW + ox + a.ox0;
                float g[8];
                unpack8<DT>(*reinterpret_cast<const uint4*>(a.x + pix * a.xs + a.xc + ch * 8), g);
                const float w = wy * wx;
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += w * g[i];
            }
        }
        const int64_t ip = ((int64_t)n * a.IH + iy) * a.IW + ix;
        *reinterpret_cast<uint4*>(a.y + ip * a.ys + a.yc + ch * 8) = pack8<DT>(acc);
    }
}

int check_up(const char* who, const void* x, const void* y, int N, int IH, int IW, int C, int lo_stride, int lo_coff,
             int PH, int PW, int hi_stride, int hi_coff, int oy0, int ox0, int dtype) {
    GS_CHECK_ARG(x && y, "%s: null pointer", who);
    GS_CHECK_ARG(N > 0 && IH > 0 && IW > 0 && C > 0 && C % 8 == 0, "%s: bad dims (C must be a multiple of 8)", who);
    GS_CHECK_ARG(lo_stride >= lo_coff + C && lo_stride % 8 == 0 && lo_coff % 8 == 0, "%s: bad low-resolution stride/offset", who);
    GS_CHECK_ARG(hi_stride >= hi_coff + C && hi_stride % 8 == 0 && hi_coff % 8 == 0, "%s: bad high-resolution stride/offset", who);
    GS_CHECK_ARG(oy0 >= 0 && ox0 >= 0 && 2 * IH + oy0 <= PH && 2 * IW + ox0 <= PW, "%s: up-sampled patch exceeds the buffer", who);
    GS_CHECK_ARG((int64_t)N * 4 * IH * IW * (C / 8) < 2147483647LL, "%s: too many elements", who);
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "%s: bad dtype", who);
    return GS_OK;
}

UpArgs make_args(int N, int IH, int IW, int C, int PH, int PW, int oy0, int ox0) {
    UpArgs a{};
    a.N = N; a.IH = IH; a.IW = IW; a.C = C; a.OH = 2 * IH; a.OW = 2 * IW; a.PH = PH; a.PW = PW; a.oy0 = oy0; a.ox0 = ox0;
    a.ry = a.OH > 1 ? (float)(IH - 1) / (float)(a.OH - 1) : 0.f;
    a.rx = a.OW > 1 ? (float)(IW - 1) / (float)(a.OW - 1) : 0.f;
    return a;
}

int grid_for(int64_t total) {
    int64_t b = cdiv64(total, 256);
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int gs_upsample2x_bilinear_fwd(const void* x, void* y, int N, int IH, int IW, int C, int in_pix_stride,
                                          int in_coff, int OH, int OW, int out_pix_stride, int out_coff, int ooy, int oox,
                                          int dtype, void* stream) {
    int rc = check_up("gs_upsample2x_bilinear_fwd", x, y, N, IH, IW, C, in_pix_stride, in_coff, OH, OW, out_pix_stride,
                      out_coff, ooy, oox, dtype);
    if (rc) return rc;
    UpArgs a = make_args(N, IH, IW, C, OH, OW, ooy, oox);
    a.x = (const unsigned short*)x; a.y = (unsigned short*)y;
    a.xs = in_pix_stride; a.xc = in_coff; a.ys = out_pix_stride; a.yc = out_coff;
    const int g = grid_for((int64_t)N * a.OH * a.OW * (C / 8));
    if (dtype == GS_F16) upsample2x_fwd_kernel<GS_F16><<<g, 256, 0, (hipStream_t)stream>>>(a);
    else upsample2x_fwd_kernel<GS_BF16><<<g, 256, 0, (hipStream_t)stream>>>(a);
    GS_CHECK_LAUNCH("gs_upsample2x_bilinear_fwd");
    return GS_OK;
}

// the same interpolation on a hi/lo pair: x_hi / x_lo share (in_pix_stride, in_coff), y_hi / y_lo share (out_pix_stride,
// out_coff); y_lo may be NULL (no consumer reads the lo plane)
extern "C" int gs_upsample2x_bilinear_fwd_pair(const void* x_hi, const void* x_lo, void* y_hi, void* y_lo, int N, int IH, int IW,
                                               int C, int in_pix_stride, int in_coff, int OH, int OW, int out_pix_stride,
                                               int out_coff, int ooy, int oox, int dtype, void* stream) {
    int rc = check_up("gs_upsample2x_bilinear_fwd_pair", x_hi, y_hi, N, IH, IW, C, in_pix_stride, in_coff, OH, OW, out_pix_stride,
                      out_coff, ooy, oox, dtype);
    if (rc) return rc;
    GS_CHECK_ARG(x_lo != nullptr, "gs_upsample2x_bilinear_fwd_pair: x_lo is NULL");
    UpArgs a = make_args(N, IH, IW, C, OH, OW, ooy, oox);
    a.x = (const unsigned short*)x_hi; a.y = (unsigned short*)y_hi;
    a.xs = in_pix_stride; a.xc = in_coff; a.ys = out_pix_stride; a.yc = out_coff;
    const int g = grid_for((int64_t)N * a.OH * a.OW * (C / 8));
    if (dtype == GS_F16) upsample2x_fwd_pair_kernel<GS_F16><<<g, 256, 0, (hipStream_t)stream>>>(a, (const unsigned short*)x_lo, (unsigned short*)y_lo);
    else upsample2x_fwd_pair_kernel<GS_BF16><<<g, 256, 0, (hipStream_t)stream>>>(a, (const unsigned short*)x_lo, (unsigned short*)y_lo);
    GS_CHECK_LAUNCH("gs_upsample2x_bilinear_fwd_pair");
    return GS_OK;
}

extern "C" int gs_upsample2x_bilinear_bwd(const void* dy, void* dx, int N, int IH, int IW, int C, int dy_pix_stride,
                                          int dy_coff, int OH, int OW, int dx_pix_stride, int dx_coff, int ooy, int oox,
                                          int dtype, void* stream) {
    int rc = check_up("gs_upsample2x_bilinear_bwd", dy, dx, N, IH, IW, C, dx_pix_stride, dx_coff, OH, OW, dy_pix_stride,
                      dy_coff, ooy, oox, dtype);
    if (rc) return rc;
    UpArgs a = make_args(N, IH, IW, C, OH, OW, ooy, oox);
    a.x = (const unsigned short*)dy; a.y = (unsigned short*)dx;
    a.xs = dy_pix_stride; a.xc = dy_coff; a.ys = dx_pix_stride; a.yc = dx_coff;
    const int g = grid_for((int64_t)N * IH * IW * (C / 8));
    if (dtype == GS_F16) upsample2x_bwd_kernel<GS_F16><<<g, 256, 0, (hipStream_t)stream>>>(a);
    else upsample2x_bwd_kernel<GS_BF16><<<g, 256, 0, (hipStream_t)stream>>>(a);
    GS_CHECK_LAUNCH("gs_upsample2x_bilinear_bwd");
    return GS_OK;
}

// ---- on-device mask augmentation -------------------------------------------------
// The reference augments the generator's input masks on the HOST with imgaug (Fliplr, CropAndPad, Affine scale /
// translate / rotate / shear in random order: running_files/train_end2end_jsrt.py:99-112), which costs a
// GPU -> CPU -> GPU round trip per Unet step (:186-190).  Every one of those operations is an affine map of the image
// plane, so their composition is ONE affine map per sample: this kernel applies it (dst pixel centre -> src
// coordinates, bilinear taps, zeros outside) and re-binarises (:191-193, threshold 0.1) in a single pass.
namespace {
__global__ __launch_bounds__(256) void affine_warp_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                          const float* __restrict__ mats, int N, int C, int H, int W,
                                                          float thresh) {
    const int total = N * C * H * W;                      // host guarantees < 2^31
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int x = idx % W;
        int r = idx / W;
        const int y = r % H; r /= H;
        const int n = r / C;
        const float* m = mats + n * 6;                   // [a b c; d e f]: (sx, sy) = M * (x + .5, y + .5, 1) - .5
        const float fx = x + 0.5f, fy = y + 0.5f;
        const float sx = m[0] * fx + m[1] * fy + m[2] - 0.5f;
        const float sy = m[3] * fx + m[4] * fy + m[5] - 0.5f;
        const float x0f = floorf(sx), y0f = floorf(sy);
        const int x0 = (int)x0f, y0 = (int)y0f;
        const float lx = sx - x0f, ly = sy - y0f;
        const float* img = src + (int64_t)(idx / (H * W)) * H * W;
        auto tap = [&](int yy, int xx) { return ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? img[yy * W + xx] : 0.f; };
        const float v = (1.f - ly) * ((1.f - lx) * tap(y0, x0) + lx * tap(y0, x0 + 1)) +
                        ly * ((1.f - lx) * tap(y0 + 1, x0) + lx * tap(y0 + 1, x0 + 1));
        dst[idx] = thresh >= 0.f ? (v > thresh ? 1.f : 0.f) : v;
    }
}
}  // namespace

extern "C" int gs_affine_warp(const float* src, float* dst, const float* mats, int N, int C, int H, int W, float thresh,
                              void* stream) {
    GS_CHECK_ARG(src && dst && mats && src != dst && N > 0 && C > 0 && H > 0 && W > 0, "gs_affine_warp: bad arguments");
    GS_CHECK_ARG((int64_t)N * C * H * W < 2147483647LL, "gs_affine_warp: too many elements");
    affine_warp_kernel<<<grid_for((int64_t)N * C * H * W), 256, 0, (hipStream_t)stream>>>(src, dst, mats, N, C, H, W, thresh);
    GS_CHECK_LAUNCH("gs_affine_warp");
    return GS_OK;
}
