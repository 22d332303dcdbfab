// MaxPool3d(kernel 2, stride 2) forward and backward on NDHWC 16-bit volumes
// (GenSeg-3D/UNet3D/unet3d.py:37,44: nn.MaxPool3d((2,2,2), stride=2)).  HBM-bound, 16-byte accesses.
// Backward follows ATen: the whole window gradient goes to the FIRST maximum in (d,h,w) scan order; the skip
// ("residual") gradient of the un-pooled tensor is added in the same pass.
#include "common.hpp"

namespace {

struct P3Args {
    const unsigned short* z; unsigned short* zp;
    const unsigned short* dzp; const unsigned short* dres; unsigned short* dz;
    int rs, rc, zs, zc;
    int NB, D, H, W, C;
};

template <int DT, bool BWD>
__global__ __launch_bounds__(256) void maxpool3d_kernel(const P3Args a) {
    const int nch = a.C >> 3;
    const int PD = (a.D + 1) / 2, PH = (a.H + 1) / 2, PW = (a.W + 1) / 2;     // ceil grid: covers the leftovers
    const int64_t total = (int64_t)a.NB * PD * PH * PW * nch;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int ch = (int)(idx % nch);
        int64_t r = idx / nch;
        const int px = (int)(r % PW); r /= PW;
        const int py = (int)(r % PH); r /= PH;
        const int pd = (int)(r % PD);
        const int nb = (int)(r / PD);
        const int c0 = ch * 8;
        const bool pooled = pd < a.D / 2 && py < a.H / 2 && px < a.W / 2;
        float v[8][8];
        bool ok[8];
        int64_t pix[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int dd = 2 * pd + (k >> 2), yy = 2 * py + ((k >> 1) & 1), xx = 2 * px + (k & 1);
            ok[k] = dd < a.D && yy < a.H && xx < a.W;
            pix[k] = (((int64_t)nb * a.D + dd) * a.H + yy) * a.W + xx;
            if (ok[k]) unpack8<DT>(*reinterpret_cast<const uint4*>(a.z + pix[k] * a.zs + a.zc + c0), v[k]);
            else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[k][i] = -INFINITY;
            }
        }
        if (!BWD) {
            if (pooled) {
                float m[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    m[i] = v[0][i];
#pragma unroll
                    for (int k = 1; k < 8; ++k) m[i] = fmaxf(m[i], v[k][i]);
                }
                const int64_t pp = (((int64_t)nb * (a.D / 2) + pd) * (a.H / 2) + py) * (a.W / 2) + px;
                *reinterpret_cast<uint4*>(a.zp + pp * a.C + c0) = pack8<DT>(m);
            }
        } else {
            float gp[8];
            if (pooled) {
                const int64_t pp = (((int64_t)nb * (a.D / 2) + pd) * (a.H / 2) + py) * (a.W / 2) + px;
                unpack8<DT>(*reinterpret_cast<const uint4*>(a.dzp + pp * a.C + c0), gp);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) gp[i] = 0.f;
            }
            int amax[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int best = 0; float bv = v[0][i];
#pragma unroll
                for (int k = 1; k < 8; ++k) if (v[k][i] > bv) { bv = v[k][i]; best = k; }
                amax[i] = best;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (!ok[k]) continue;
                float g[8];
                if (a.dres) unpack8<DT>(*reinterpret_cast<const uint4*>(a.dres + pix[k] * a.rs + a.rc + c0), g);
                else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) g[i] = 0.f;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) g[i] += (amax[i] == k) ? gp[i] : 0.f;
                *reinterpret_cast<uint4*>(a.dz + pix[k] * a.C + c0) = pack8<DT>(g);
            }
        }
    }
}

// Forward on hi/lo pairs (the pair forward of UNet3D): the maximum of the pair VALUES hi + lo over the 2x2x2 window, stored as a
// pair again (the lo plane only when a consumer reads it).  z_hi / z_lo: the two planes of the un-pooled tensor (each with its own
// channel offset inside a buffer of pixel stride zs); zp_hi / zp_lo: pooled planes with pixel stride zps.
struct P3PairArgs {
    const unsigned short* z_hi; const unsigned short* z_lo; unsigned short* zp_hi; unsigned short* zp_lo;
    int zs, zps, NB, D, H, W, C;
    int z_q8, zq_coff, zp_q8;      // z_lo / zp_lo are Q PLANES (common.hpp): z_lo points at byte 0 of the buffer's q plane and the pooled
                                   // tensor's channels start at channel zq_coff of it; zp_lo at byte 0 of the pooled buffer's q plane
};

template <int DT>
__global__ __launch_bounds__(256) void maxpool3d_pair_kernel(const P3PairArgs a) {
    const int nch = a.C >> 3;
    const int PD = a.D / 2, PH = a.H / 2, PW = a.W / 2;
    const int64_t total = (int64_t)a.NB * PD * PH * PW * nch;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int ch = (int)(idx % nch);
        int64_t r = idx / nch;
        const int px = (int)(r % PW); r /= PW;
        const int py = (int)(r % PH); r /= PH;
        const int pd = (int)(r % PD);
        const int nb = (int)(r / PD);
        const int c0 = ch * 8;
        uint4 vh[8], vl[8];
        uint2 vq[8];
        const float inv_slo = __builtin_ldexpf(1.f, -(GS_Q8_XH_EXP + Q8Shift<DT>::v));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int dd = 2 * pd + (k >> 2), yy = 2 * py + ((k >> 1) & 1), xx = 2 * px + (k & 1);
            const int64_t pix = (((int64_t)nb * a.D + dd) * a.H + yy) * a.W + xx;
            vh[k] = *reinterpret_cast<const uint4*>(a.z_hi + pix * a.zs + c0);
            if (a.z_q8) vq[k] = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(a.z_lo) + pix * a.zs * 2 + q8_off(a.zq_coff + c0));
            else vl[k] = *reinterpret_cast<const uint4*>(a.z_lo + pix * a.zs + c0);
        }
        float m[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) m[i] = -INFINITY;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v[8];
            if (a.z_q8) {                                  // the pair value as the q plane holds it: hi + e4m3(lo * 2^s) / 2^s
                unpack8<DT>(vh[k], v);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += e4m3_to_f(((i < 4 ? vq[k].x : vq[k].y) >> (8 * (i & 3))) & 0xffu) * inv_slo;
            } else {
                join8<DT>(vh[k], vl[k], v);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) m[i] = fmaxf(m[i], v[i]);
        }
        const int64_t pp = (((int64_t)nb * PD + pd) * PH + py) * PW + px;
        uint4 hi, lo;
        split8<DT>(m, hi, lo);
        *reinterpret_cast<uint4*>(a.zp_hi + pp * a.zps + c0) = hi;
        if (a.zp_lo && !a.zp_q8) *reinterpret_cast<uint4*>(a.zp_lo + pp * a.zps + c0) = lo;
        if (a.zp_lo && a.zp_q8) {
            uint2 lo8, hi8;
            q8_of8<DT>(m, hi, __builtin_ldexpf(1.f, GS_Q8_XH_EXP + Q8Shift<DT>::v), __builtin_ldexpf(1.f, GS_Q8_XH_EXP), lo8, hi8);
            unsigned char* zq = reinterpret_cast<unsigned char*>(a.zp_lo) + pp * a.zps * 2 + q8_off(c0);
            *reinterpret_cast<uint2*>(zq) = lo8;
            *reinterpret_cast<uint2*>(zq + 32) = hi8;
        }
    }
}

int launch(const P3Args& a, bool bwd, int dtype, hipStream_t s) {
    const int64_t total = (int64_t)a.NB * ((a.D + 1) / 2) * ((a.H + 1) / 2) * ((a.W + 1) / 2) * (a.C / 8);
    int64_t nb = cdiv64(total, 256);
    if (nb > 8192) nb = 8192;
    if (dtype == GS_F16) {
        if (bwd) maxpool3d_kernel<GS_F16, true><<<(int)nb, 256, 0, s>>>(a);
        else maxpool3d_kernel<GS_F16, false><<<(int)nb, 256, 0, s>>>(a);
    } else {
        if (bwd) maxpool3d_kernel<GS_BF16, true><<<(int)nb, 256, 0, s>>>(a);
        else maxpool3d_kernel<GS_BF16, false><<<(int)nb, 256, 0, s>>>(a);
    }
    return 0;
}

}  // namespace

namespace {
// MaxPool2d(2) forward alone (unet_parts.py:34) for the inference path, where conv + folded BatchNorm + ReLU is one
// kernel and nothing else touches the activation: z [N,H,W,*] (strided: a concat buffer) -> zp [N,H/2,W/2,C] dense.
template <int DT>
__global__ __launch_bounds__(256) void maxpool2x2_kernel(const unsigned short* __restrict__ z, int zs, int zc,
                                                         unsigned short* __restrict__ zp, int N, int H, int W, int C) {
    const int nch = C >> 3, PH = H / 2, PW = W / 2;
    const int total = N * PH * PW * nch;                  // host guarantees < 2^31
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int ch = idx % nch;
        int r = idx / nch;
        const int px = r % PW; r /= PW;
        const int py = r % PH;
        const int n = r / PH;
        const int64_t p00 = ((int64_t)n * H + 2 * py) * W + 2 * px;
        const unsigned short* src = z + p00 * zs + zc + ch * 8;
        const uint4 r0 = *reinterpret_cast<const uint4*>(src);
        const uint4 r1 = *reinterpret_cast<const uint4*>(src + zs);
        const uint4 r2 = *reinterpret_cast<const uint4*>(src + (int64_t)W * zs);
        const uint4 r3 = *reinterpret_cast<const uint4*>(src + (int64_t)(W + 1) * zs);
        float a[8], b[8], c[8], d[8], m[8];
        unpack8<DT>(r0, a); unpack8<DT>(r1, b); unpack8<DT>(r2, c); unpack8<DT>(r3, d);
#pragma unroll
        for (int i = 0; i < 8; ++i) m[i] = fmaxf(fmaxf(a[i], b[i]), fmaxf(c[i], d[i]));
        *reinterpret_cast<uint4*>(zp + (int64_t)(idx / nch) * C + ch * 8) = pack8<DT>(m);
    }
}
// The same on PAIRS (folded-BatchNorm inference of the pair forward): the pooled pair is the maximum of the pair VALUES hi + lo of
// the window, split again; z_hi / z_lo share the pixel stride zs (two planes of one buffer), zp_hi / zp_lo the stride zps.
template <int DT>
__global__ __launch_bounds__(256) void maxpool2x2_pair_kernel(const unsigned short* __restrict__ z_hi, const unsigned short* __restrict__ z_lo,
                                                              int zs, unsigned short* __restrict__ zp_hi, unsigned short* __restrict__ zp_lo,
                                                              int zps, int N, int H, int W, int C) {
    const int nch = C >> 3, PH = H / 2, PW = W / 2;
    const int total = N * PH * PW * nch;                  // host guarantees < 2^31
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int ch = idx % nch;
        int r = idx / nch;
        const int px = r % PW; r /= PW;
        const int py = r % PH;
        const int n = r / PH;
        const int64_t p00 = ((int64_t)n * H + 2 * py) * W + 2 * px;
        const int64_t off[4] = {p00 * zs, (p00 + 1) * zs, (p00 + W) * zs, (p00 + W + 1) * zs};
        uint4 vh[4], vl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            vh[k] = *reinterpret_cast<const uint4*>(z_hi + off[k] + ch * 8);
            vl[k] = *reinterpret_cast<const uint4*>(z_lo + off[k] + ch * 8);
        }
        float m[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) m[i] = -INFINITY;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v[8];
            join8<DT>(vh[k], vl[k], v);
#pragma unroll
            for (int i = 0; i < 8; ++i) m[i] = fmaxf(m[i], v[i]);
        }
        uint4 hi, lo;
        split8<DT>(m, hi, lo);
        const int64_t pp = (int64_t)(idx / nch) * zps + ch * 8;
        *reinterpret_cast<uint4*>(zp_hi + pp) = hi;
        if (zp_lo) *reinterpret_cast<uint4*>(zp_lo + pp) = lo;
    }
}
}  // namespace

extern "C" int gs_maxpool2x2_fwd(const void* z, int z_pix_stride, int z_coff, void* zp, int N, int H, int W, int C,
                                 int dtype, void* stream) {
    GS_CHECK_ARG(z && zp && N > 0 && H > 1 && W > 1 && C > 0 && C % 8 == 0, "gs_maxpool2x2_fwd: bad arguments");
    GS_CHECK_ARG(z_pix_stride >= z_coff + C && z_pix_stride % 8 == 0 && z_coff % 8 == 0, "gs_maxpool2x2_fwd: bad z stride");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_maxpool2x2_fwd: bad dtype");
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 8);
    GS_CHECK_ARG(total < 2147483000LL, "gs_maxpool2x2_fwd: output too large for 32-bit indexing");
    int64_t nb = cdiv64(total, 256);
    if (nb > 16384) nb = 16384;
    if (dtype == GS_F16)
        maxpool2x2_kernel<GS_F16><<<(int)nb, 256, 0, (hipStream_t)stream>>>((const unsigned short*)z, z_pix_stride, z_coff,
                                                                            (unsigned short*)zp, N, H, W, C);
    else
        maxpool2x2_kernel<GS_BF16><<<(int)nb, 256, 0, (hipStream_t)stream>>>((const unsigned short*)z, z_pix_stride, z_coff,
                                                                             (unsigned short*)zp, N, H, W, C);
    GS_CHECK_LAUNCH("gs_maxpool2x2_fwd");
    return GS_OK;
}

extern "C" int gs_maxpool2x2_fwd_pair(const void* z_hi, const void* z_lo, int z_pix_stride, void* zp_hi, void* zp_lo, int zp_pix_stride,
                                      int N, int H, int W, int C, int dtype, void* stream) {
    GS_CHECK_ARG(z_hi && z_lo && zp_hi && N > 0 && H > 1 && W > 1 && C > 0 && C % 8 == 0, "gs_maxpool2x2_fwd_pair: bad arguments");
    GS_CHECK_ARG(z_pix_stride >= C && z_pix_stride % 8 == 0 && zp_pix_stride >= C && zp_pix_stride % 8 == 0, "gs_maxpool2x2_fwd_pair: bad strides");
    GS_CHECK_ARG(((uintptr_t)z_hi | (uintptr_t)z_lo | (uintptr_t)zp_hi | (uintptr_t)zp_lo) % 16 == 0, "gs_maxpool2x2_fwd_pair: planes must be 16-byte aligned");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_maxpool2x2_fwd_pair: bad dtype");
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 8);
    GS_CHECK_ARG(total < 2147483000LL, "gs_maxpool2x2_fwd_pair: output too large for 32-bit indexing");
    int64_t nb = cdiv64(total, 256);
    if (nb > 16384) nb = 16384;
    if (dtype == GS_F16)
        maxpool2x2_pair_kernel<GS_F16><<<(int)nb, 256, 0, (hipStream_t)stream>>>((const unsigned short*)z_hi, (const unsigned short*)z_lo, z_pix_stride,
                                                                                 (unsigned short*)zp_hi, (unsigned short*)zp_lo, zp_pix_stride, N, H, W, C);
    else
        maxpool2x2_pair_kernel<GS_BF16><<<(int)nb, 256, 0, (hipStream_t)stream>>>((const unsigned short*)z_hi, (const unsigned short*)z_lo, z_pix_stride,
                                                                                  (unsigned short*)zp_hi, (unsigned short*)zp_lo, zp_pix_stride, N, H, W, C);
    GS_CHECK_LAUNCH("gs_maxpool2x2_fwd_pair");
    return GS_OK;
}

extern "C" int gs_maxpool3d_fwd(const void* z, int z_pix_stride, int z_coff, void* zp, int NB, int D, int H, int W, int C,
                                int dtype, void* stream) {
    GS_CHECK_ARG(z && zp && NB > 0 && D > 1 && H > 1 && W > 1 && C > 0 && C % 8 == 0, "gs_maxpool3d_fwd: bad arguments");
    GS_CHECK_ARG(z_pix_stride >= z_coff + C && z_pix_stride % 8 == 0 && z_coff % 8 == 0, "gs_maxpool3d_fwd: bad z stride");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_maxpool3d_fwd: bad dtype");
    P3Args a{(const unsigned short*)z, (unsigned short*)zp, nullptr, nullptr, nullptr, 0, 0, z_pix_stride, z_coff, NB, D, H, W, C};
    launch(a, false, dtype, (hipStream_t)stream);
    GS_CHECK_LAUNCH("gs_maxpool3d_fwd");
    return GS_OK;
}

static int maxpool3d_pair_impl(const void* z_hi, const void* z_lo, int z_pix_stride, void* zp_hi, void* zp_lo, int zp_pix_stride, int NB,
                               int D, int H, int W, int C, int dtype, void* stream, int z_q8, int zq_coff, int zp_q8);
extern "C" int gs_maxpool3d_fwd_pair(const void* z_hi, const void* z_lo, int z_pix_stride, void* zp_hi, void* zp_lo,
                                     int zp_pix_stride, int NB, int D, int H, int W, int C, int dtype, void* stream) {
    return maxpool3d_pair_impl(z_hi, z_lo, z_pix_stride, zp_hi, zp_lo, zp_pix_stride, NB, D, H, W, C, dtype, stream, 0, 0, 0);
}
// the same with q planes (FP8 correction chunks): z_q8: z_lo = byte 0 of the input buffer's q plane, the pooled channels start at its
// channel zq_coff; zp_q8: zp_lo = byte 0 of the pooled buffer's q plane
extern "C" int gs_maxpool3d_fwd_pair_q8(const void* z_hi, const void* z_lo, int z_q8, int zq_coff, int z_pix_stride, void* zp_hi, void* zp_lo,
                                        int zp_q8, int zp_pix_stride, int NB, int D, int H, int W, int C, int dtype, void* stream) {
    GS_CHECK_ARG((!z_q8 || (zq_coff % 32 == 0 && C % 32 == 0)) && (!zp_q8 || C % 32 == 0), "gs_maxpool3d_fwd_pair_q8: q planes need multiples of 32 channels");
    return maxpool3d_pair_impl(z_hi, z_lo, z_pix_stride, zp_hi, zp_lo, zp_pix_stride, NB, D, H, W, C, dtype, stream, z_q8 ? 1 : 0, zq_coff, zp_q8 ? 1 : 0);
}
static int maxpool3d_pair_impl(const void* z_hi, const void* z_lo, int z_pix_stride, void* zp_hi, void* zp_lo, int zp_pix_stride, int NB,
                               int D, int H, int W, int C, int dtype, void* stream, int z_q8, int zq_coff, int zp_q8) {
    GS_CHECK_ARG(z_hi && z_lo && zp_hi && NB > 0 && D > 1 && H > 1 && W > 1 && C > 0 && C % 8 == 0, "gs_maxpool3d_fwd_pair: bad arguments");
    GS_CHECK_ARG(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "gs_maxpool3d_fwd_pair: even volume dims only");
    GS_CHECK_ARG(z_pix_stride >= C && z_pix_stride % 8 == 0 && zp_pix_stride >= C && zp_pix_stride % 8 == 0,
                 "gs_maxpool3d_fwd_pair: bad strides");
    GS_CHECK_ARG(((uintptr_t)z_hi | (uintptr_t)z_lo | (uintptr_t)zp_hi | (uintptr_t)zp_lo) % 16 == 0, "gs_maxpool3d_fwd_pair: planes must be 16-byte aligned");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_maxpool3d_fwd_pair: bad dtype");
    P3PairArgs a{(const unsigned short*)z_hi, (const unsigned short*)z_lo, (unsigned short*)zp_hi, (unsigned short*)zp_lo,
                 z_pix_stride, zp_pix_stride, NB, D, H, W, C, z_q8, zq_coff, zp_q8};
    int64_t nb = cdiv64((int64_t)NB * (D / 2) * (H / 2) * (W / 2) * (C / 8), 256);
    if (nb > 8192) nb = 8192;
    if (dtype == GS_F16) maxpool3d_pair_kernel<GS_F16><<<(int)nb, 256, 0, (hipStream_t)stream>>>(a);
    else maxpool3d_pair_kernel<GS_BF16><<<(int)nb, 256, 0, (hipStream_t)stream>>>(a);
    GS_CHECK_LAUNCH("gs_maxpool3d_fwd_pair");
    return GS_OK;
}

extern "C" int gs_maxpool3d_bwd(const void* z, int z_pix_stride, int z_coff, const void* dzp, const void* dres,
                                int res_pix_stride, int res_coff, void* dz, int NB, int D, int H, int W, int C, int dtype,
                                void* stream) {
    GS_CHECK_ARG(z_pix_stride >= z_coff + C && z_pix_stride % 8 == 0 && z_coff % 8 == 0, "gs_maxpool3d_bwd: bad z stride");
    GS_CHECK_ARG(z && dzp && dz && NB > 0 && D > 1 && H > 1 && W > 1 && C > 0 && C % 8 == 0, "gs_maxpool3d_bwd: bad arguments");
    GS_CHECK_ARG(!dres || (res_pix_stride >= res_coff + C && res_pix_stride % 8 == 0 && res_coff % 8 == 0),
                 "gs_maxpool3d_bwd: bad residual stride");
    GS_CHECK_ARG(dtype == GS_F16 || dtype == GS_BF16, "gs_maxpool3d_bwd: bad dtype");
    P3Args a{(const unsigned short*)z, nullptr, (const unsigned short*)dzp, (const unsigned short*)dres,
             (unsigned short*)dz, res_pix_stride, res_coff, z_pix_stride, z_coff, NB, D, H, W, C};
    launch(a, true, dtype, (hipStream_t)stream);
    GS_CHECK_LAUNCH("gs_maxpool3d_bwd");
    return GS_OK;
}
