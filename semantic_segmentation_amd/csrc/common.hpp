// Shared device/host helpers for the gfx950 kernels.  gfx950 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "gsseg.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short i16x4;
typedef __attribute__((ext_vector_type(8))) short i16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// Output stores of the streaming kernels.  GS_OUT_AUX = 16 (`sc1`, write-through) on the buffer stores of the conv / up-conv
// epilogues: the line leaves the XCD's L2 at once instead of staying dirty until the kernel boundary writes the L2 back
// (MI355X_MICROARCH.md, "boundary").  Measured on the bs 32 step, three alternating rounds on one box: plain 15.29 ms, conv /
// up-conv stores sc1 15.25 ms, these AND the 16-byte stores of the BatchNorm / stem passes (st16 as inline-asm
// `global_store_dwordx4 ... sc1`, GS_ST16_WT = 1) 15.28 ms: kept for the buffer stores, not for st16.
#ifndef GS_OUT_AUX
#define GS_OUT_AUX 16
#endif
#ifndef GS_ST16_WT
#define GS_ST16_WT 0
#endif
__device__ __forceinline__ void st16(void* p, const uint4& v) {
#if GS_ST16_WT
    const u32x4 t = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(t) : "memory");
#else
    *reinterpret_cast<uint4*>(p) = v;
#endif
}

#define LDS_AS __attribute__((address_space(3)))

void gs_set_error(const char* fmt, ...);

#define GS_CHECK_ARG(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            gs_set_error(__VA_ARGS__);     \
            return GS_EINVAL;              \
        }                                  \
    } while (0)

#define GS_CHECK_LAUNCH(name)                                                         \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            gs_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));       \
            return GS_ELAUNCH;                                                        \
        }                                                                             \
    } while (0)

// ---- 16-bit element traits --------------------------------------------------------------------
template <int DT>
struct Elem;

template <>
struct Elem<GS_F16> {
    typedef _Float16 T;
    typedef f16x8 V8;
    typedef f16x4 V4;
    static __device__ __forceinline__ float to_f(unsigned short u) {
        return (float)__builtin_bit_cast(_Float16, u);
    }
    static __device__ __forceinline__ unsigned short from_f(float f) {
        return __builtin_bit_cast(unsigned short, (_Float16)f);
    }
    // two values -> one dword (lo = a, hi = b): a single v_cvt_pk_f16_f32 (RNE, same results as two scalar converts)
    static __device__ __forceinline__ unsigned int pack2(float a, float b) {
        typedef float f32x2_ __attribute__((ext_vector_type(2)));
        typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
        const f32x2_ v = {a, b};
        return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, f16x2_));
    }
    static __device__ __forceinline__ f32x16 mfma32(V8 a, V8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }

};

template <>
struct Elem<GS_BF16> {
    typedef __bf16 T;
    typedef bf16x8 V8;
    typedef bf16x4 V4;
    static __device__ __forceinline__ float to_f(unsigned short u) {
        return __builtin_bit_cast(float, ((unsigned int)u) << 16);
    }
    static __device__ __forceinline__ unsigned short from_f(float f) {
        return __builtin_bit_cast(unsigned short, (__bf16)f);   // v_cvt_pk_bf16_f32: RNE, NaN-safe
    }
    static __device__ __forceinline__ unsigned int pack2(float a, float b) {
        typedef float f32x2_ __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
        const f32x2_ v = {a, b};
        return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_));
    }
    static __device__ __forceinline__ f32x16 mfma32(V8 a, V8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }

};

// gfx950 transposing LDS read (ds_read_b64_tr_b16): two reads give one 8-element K-minor MFMA fragment
__device__ __forceinline__ i16x4 tr_read4(const LDS_AS unsigned short* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS i16x4*)p);
}
template <int DT>
__device__ __forceinline__ typename Elem<DT>::V8 tr_read8(const LDS_AS unsigned short* lo, const LDS_AS unsigned short* hi) {
    const i16x4 a = tr_read4(lo), b = tr_read4(hi);
    const i16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(typename Elem<DT>::V8, v);
}

// unpack / pack 8 consecutive 16-bit elements held in a uint4
template <int DT>
__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
    const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = Elem<DT>::to_f((unsigned short)(w[i] & 0xffffu));
        f[2 * i + 1] = Elem<DT>::to_f((unsigned short)(w[i] >> 16));
    }
}
template <int DT>
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    unsigned int w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        w[i] = Elem<DT>::pack2(f[2 * i], f[2 * i + 1]);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// hi/lo pairs of 16-bit values (the pair forward): v = hi + lo, hi = 16-bit(v), lo = 16-bit(v - hi)
template <int DT>
__device__ __forceinline__ void split8(const float (&v)[8], uint4& hi, uint4& lo) {
    hi = pack8<DT>(v);
    float h[8], l[8];
    unpack8<DT>(hi, h);
#pragma unroll
    for (int i = 0; i < 8; ++i) l[i] = v[i] - h[i];
    lo = pack8<DT>(l);
}
template <int DT>
__device__ __forceinline__ void join8(const uint4& hi, const uint4& lo, float (&v)[8]) {
    float l[8];
    unpack8<DT>(hi, v);
    unpack8<DT>(lo, l);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += l[i];
}

// ---- FP8 correction planes of the pair forward ("q" stages) ----------------------------------------
// A "q" conv stage computes x_hi.w_hi on the 16-bit MFMA and the two correction terms x_lo.w_hi + x_hi.w_lo as ONE block-scaled
// e4m3 MFMA segment (v_mfma_scale_f32_32x32x64_f8f6f4: twice the 16-bit rate).  Its operands travel in "q planes": per 32 channels
// one 64-byte chunk [lo8 (32 x e4m3 of lo * 2^XL) | hi8 (32 x e4m3 of hi * 2^XH)] for activations and [w_hi8 | w_lo8] for weights
// -- the same bytes per pixel as the 16-bit lo plane they replace.  In the MFMA the first 16 bytes of a lane's operand belong to
// K block 0 (scaled by the E8M0 byte of lanes 0..31) and the last 16 to K block 1 (lanes 32..63) [probed: tools/probes/mx_probe2.hip],
// so a lane that reads the 16-byte slots {h, 2 + h} of a chunk row -- exactly the two reads of the 16-bit stages -- pairs lo8 with
// w_hi8 in block 0 and hi8 with w_lo8 in block 1.
// Activation scales are static powers of two: hi * 2^-2 keeps |x| < 1792 in range (e4m3 saturates at 448; a saturated value only
// degrades the correction term), lo = x - hi is at most 2^-11 |x| in fp16, so lo * 2^9 has the same range.
constexpr int GS_Q8_XH_EXP = -2;
constexpr int GS_Q8_LO_SHIFT_F16 = 11, GS_Q8_LO_SHIFT_BF16 = 8;
template <int DT> struct Q8Shift { static constexpr int v = DT == GS_F16 ? GS_Q8_LO_SHIFT_F16 : GS_Q8_LO_SHIFT_BF16; };

// four floats -> four saturating e4m3 bytes (a in byte 0)
__device__ __forceinline__ unsigned int cvt4_e4m3(float a, float b, float c, float d) {
    a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
    c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f); d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
    int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (unsigned int)v;
}
// the q-plane bytes of 8 consecutive channels whose exact values are v and whose stored hi halves are `hi`: lo8 / hi8 (8 bytes each)
template <int DT>
__device__ __forceinline__ void q8_of8(const float (&v)[8], const uint4& hi, float s_lo, float s_hi, uint2& lo8, uint2& hi8) {
    float h[8];
    unpack8<DT>(hi, h);
    lo8.x = cvt4_e4m3((v[0] - h[0]) * s_lo, (v[1] - h[1]) * s_lo, (v[2] - h[2]) * s_lo, (v[3] - h[3]) * s_lo);
    lo8.y = cvt4_e4m3((v[4] - h[4]) * s_lo, (v[5] - h[5]) * s_lo, (v[6] - h[6]) * s_lo, (v[7] - h[7]) * s_lo);
    hi8.x = cvt4_e4m3(h[0] * s_hi, h[1] * s_hi, h[2] * s_hi, h[3] * s_hi);
    hi8.y = cvt4_e4m3(h[4] * s_hi, h[5] * s_hi, h[6] * s_hi, h[7] * s_hi);
}
// byte offset, inside a q plane, of the lo8 bytes of channel c (a multiple of 8); the hi8 bytes sit 32 bytes further
__device__ __forceinline__ int q8_off(int c) { return (c >> 5) * 64 + (c & 31); }
// e4m3 byte -> float (for the pair max-pool, which reads a q plane back)
__device__ __forceinline__ float e4m3_to_f(unsigned int b) {
    const unsigned int e = (b >> 3) & 15u, m = b & 7u;
    const float f = e == 0 ? (float)m * 0.001953125f : __builtin_bit_cast(float, ((e + 120u) << 23) | (m << 20));
    return (b & 0x80u) ? -f : f;
}

__device__ __forceinline__ float act_fwd(float v, int act) {
    switch (act) {
        case GS_ACT_RELU: return v > 0.f ? v : 0.f;
        case GS_ACT_LEAKY02: return v > 0.f ? v : 0.2f * v;
        case GS_ACT_TANH: return tanhf(v);
        default: return v;
    }
}
// derivative of act w.r.t. its input, given the pre-activation value v
__device__ __forceinline__ float act_grad(float v, int act) {
    switch (act) {
        case GS_ACT_RELU: return v > 0.f ? 1.f : 0.f;
        case GS_ACT_LEAKY02: return v > 0.f ? 1.f : 0.2f;
        case GS_ACT_TANH: { float t = tanhf(v); return 1.f - t * t; }
        default: return 1.f;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread
__device__ __forceinline__ float block_sum_256(float v, float* red /* >= 4 floats of LDS */) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
