// Multi-tensor optimiser steps: ONE launch updates every parameter of a model.
//   RMSprop(momentum, weight_decay)  -- the U-Net optimiser of running_files/train_end2end_jsrt.py:69-70
//   Adam(betas, weight_decay)        -- the Pix2Pix optimisers of models_pix2pix/pix2pix_model.py:69-72 and the
//                                        architecture optimiser of train_end2end_jsrt.py:318
// Arithmetic follows torch.optim's single-tensor reference implementations (torch/optim/rmsprop.py, adam.py) in
// fp32, element by element, so the results agree with torch to the last couple of ulps.  HBM-bound: RMSprop with
// momentum moves 4 reads + 3 writes of 4 bytes per parameter, Adam 4 + 3.
//
// The caller passes device-resident tables (built once per parameter set by optim.py): per tensor the four
// pointers and the element count, per 64K-element chunk the owning tensor and the chunk's start.
#include "common.hpp"

namespace {

constexpr int OPT_CHUNK = 65536;

struct OptTable {
    float* const* p;
    const float* const* g;
    float* const* s1;
    float* const* s2;
    const int64_t* n;
    const int32_t* chunk_tensor;
    const int64_t* chunk_start;
};

struct RmsHyper { float lr, alpha, eps, wd, momentum, gscale; int centered; };

__global__ __launch_bounds__(256) void rmsprop_kernel(const OptTable tb, const RmsHyper h) {
    const int t = tb.chunk_tensor[blockIdx.x];
    const int64_t start = tb.chunk_start[blockIdx.x];
    const int64_t n = tb.n[t];
    const int64_t end = start + OPT_CHUNK < n ? start + OPT_CHUNK : n;
    float* __restrict__ p = tb.p[t];
    const float* __restrict__ g = tb.g[t];
    float* __restrict__ sq = tb.s1[t];
    float* __restrict__ buf = tb.s2[t];
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)sq | (uintptr_t)buf) & 15) == 0) && ((start & 3) == 0);
    auto upd = [&](float& pv, float gv, float& sv, float& bv) __attribute__((always_inline)) {
        gv *= h.gscale;
        if (h.wd != 0.f) gv = gv + h.wd * pv;                       // grad.add(param, alpha=weight_decay)
        sv = sv * h.alpha + (1.f - h.alpha) * gv * gv;              // square_avg.mul_(alpha).addcmul_(g, g, 1-alpha)
        const float avg = sqrtf(sv) + h.eps;                        // square_avg.sqrt().add_(eps)
        if (h.momentum > 0.f) {
            bv = bv * h.momentum + gv / avg;                        // buf.mul_(momentum).addcdiv_(grad, avg)
            pv = pv - h.lr * bv;                                    // param.add_(buf, alpha=-lr)
        } else {
            pv = pv - h.lr * (gv / avg);                            // param.addcdiv_(grad, avg, value=-lr)
        }
    };
    if (vec) {
        const int64_t end4 = start + ((end - start) & ~(int64_t)3);
        for (int64_t i = start + (int64_t)threadIdx.x * 4; i < end4; i += 256 * 4) {
            float4 pv = *reinterpret_cast<float4*>(p + i);
            const float4 gv = *reinterpret_cast<const float4*>(g + i);
            float4 sv = *reinterpret_cast<float4*>(sq + i);
            float4 bv = buf ? *reinterpret_cast<float4*>(buf + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            upd(pv.x, gv.x, sv.x, bv.x); upd(pv.y, gv.y, sv.y, bv.y); upd(pv.z, gv.z, sv.z, bv.z); upd(pv.w, gv.w, sv.w, bv.w);
            *reinterpret_cast<float4*>(p + i) = pv;
            *reinterpret_cast<float4*>(sq + i) = sv;
            if (buf) *reinterpret_cast<float4*>(buf + i) = bv;
        }
        for (int64_t i = end4 + threadIdx.x; i < end; i += 256) {
            float bv = buf ? buf[i] : 0.f;
            upd(p[i], g[i], sq[i], bv);
            if (buf) buf[i] = bv;
        }
    } else {
        for (int64_t i = start + threadIdx.x; i < end; i += 256) {
            float bv = buf ? buf[i] : 0.f;
            upd(p[i], g[i], sq[i], bv);
            if (buf) buf[i] = bv;
        }
    }
}

struct AdamHyper { float beta1, beta2, eps, wd, gscale; };

// per-tensor step scalars: sc[2t] = lr / (1 - beta1^step), sc[2t+1] = sqrt(1 - beta2^step)
__global__ __launch_bounds__(256) void adam_kernel(const OptTable tb, const float* __restrict__ sc, const AdamHyper h) {
    const int t = tb.chunk_tensor[blockIdx.x];
    const int64_t start = tb.chunk_start[blockIdx.x];
    const int64_t n = tb.n[t];
    const int64_t end = start + OPT_CHUNK < n ? start + OPT_CHUNK : n;
    float* __restrict__ p = tb.p[t];
    const float* __restrict__ g = tb.g[t];
    float* __restrict__ m = tb.s1[t];
    float* __restrict__ v = tb.s2[t];
    const float step_size = sc[2 * t], bc2_sqrt = sc[2 * t + 1];
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) && ((start & 3) == 0);
    auto upd = [&](float& pv, float gv, float& mv, float& vv) __attribute__((always_inline)) {
        gv *= h.gscale;
        if (h.wd != 0.f) gv = gv + h.wd * pv;                       // grad.add(param, alpha=weight_decay)
        mv = mv + (gv - mv) * (1.f - h.beta1);                      // exp_avg.lerp_(grad, 1 - beta1)
        vv = vv * h.beta2 + (1.f - h.beta2) * gv * gv;              // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
        const float denom = sqrtf(vv) / bc2_sqrt + h.eps;           // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
        pv = pv - step_size * (mv / denom);                         // param.addcdiv_(exp_avg, denom, value=-step_size)
    };
    if (vec) {
        const int64_t end4 = start + ((end - start) & ~(int64_t)3);
        int64_t i = start + (int64_t)threadIdx.x * 4;
        // two float4 groups per lane per trip: eight 16-byte loads in flight (the 1.09 GB generator: 3.8 -> TB/s of 28 B / parameter)
        for (; i + 256 * 4 < end4; i += 2 * 256 * 4) {
            const int64_t j = i + 256 * 4;
            float4 pa = *reinterpret_cast<float4*>(p + i), pb = *reinterpret_cast<float4*>(p + j);
            const float4 ga = *reinterpret_cast<const float4*>(g + i), gb = *reinterpret_cast<const float4*>(g + j);
            float4 ma = *reinterpret_cast<float4*>(m + i), mb = *reinterpret_cast<float4*>(m + j);
            float4 va = *reinterpret_cast<float4*>(v + i), vb = *reinterpret_cast<float4*>(v + j);
            upd(pa.x, ga.x, ma.x, va.x); upd(pa.y, ga.y, ma.y, va.y); upd(pa.z, ga.z, ma.z, va.z); upd(pa.w, ga.w, ma.w, va.w);
            upd(pb.x, gb.x, mb.x, vb.x); upd(pb.y, gb.y, mb.y, vb.y); upd(pb.z, gb.z, mb.z, vb.z); upd(pb.w, gb.w, mb.w, vb.w);
            *reinterpret_cast<float4*>(p + i) = pa; *reinterpret_cast<float4*>(p + j) = pb;
            *reinterpret_cast<float4*>(m + i) = ma; *reinterpret_cast<float4*>(m + j) = mb;
            *reinterpret_cast<float4*>(v + i) = va; *reinterpret_cast<float4*>(v + j) = vb;
        }
        for (; i < end4; i += 256 * 4) {
            float4 pv = *reinterpret_cast<float4*>(p + i);
            const float4 gv = *reinterpret_cast<const float4*>(g + i);
            float4 mv = *reinterpret_cast<float4*>(m + i);
            float4 vv = *reinterpret_cast<float4*>(v + i);
            upd(pv.x, gv.x, mv.x, vv.x); upd(pv.y, gv.y, mv.y, vv.y); upd(pv.z, gv.z, mv.z, vv.z); upd(pv.w, gv.w, mv.w, vv.w);
            *reinterpret_cast<float4*>(p + i) = pv;
            *reinterpret_cast<float4*>(m + i) = mv;
            *reinterpret_cast<float4*>(v + i) = vv;
        }
        for (int64_t i = end4 + threadIdx.x; i < end; i += 256) upd(p[i], g[i], m[i], v[i]);
    } else {
        for (int64_t i = start + threadIdx.x; i < end; i += 256) upd(p[i], g[i], m[i], v[i]);
    }
}

}  // namespace

extern "C" int gs_optim_chunk_elems(void) { return OPT_CHUNK; }

extern "C" int gs_optim_rmsprop(float* const* params, const float* const* grads, float* const* square_avg,
                                float* const* momentum_buf, const int64_t* sizes, const int32_t* chunk_tensor,
                                const int64_t* chunk_start, int nchunks, float lr, float alpha, float eps,
                                float weight_decay, float momentum, float grad_scale, void* stream) {
    GS_CHECK_ARG(params && grads && square_avg && momentum_buf && sizes && chunk_tensor && chunk_start,
                 "gs_optim_rmsprop: null table");
    GS_CHECK_ARG(nchunks >= 0 && lr >= 0.f && eps >= 0.f && alpha >= 0.f && momentum >= 0.f && weight_decay >= 0.f,
                 "gs_optim_rmsprop: bad hyper-parameters");
    if (nchunks == 0) return GS_OK;
    const OptTable tb{params, grads, square_avg, momentum_buf, sizes, chunk_tensor, chunk_start};
    const RmsHyper h{lr, alpha, eps, weight_decay, momentum, grad_scale, 0};
    rmsprop_kernel<<<nchunks, 256, 0, (hipStream_t)stream>>>(tb, h);
    GS_CHECK_LAUNCH("gs_optim_rmsprop");
    return GS_OK;
}

extern "C" int gs_optim_adam(float* const* params, const float* const* grads, float* const* exp_avg,
                             float* const* exp_avg_sq, const int64_t* sizes, const int32_t* chunk_tensor,
                             const int64_t* chunk_start, int nchunks, const float* step_scalars, float beta1,
                             float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
    GS_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && sizes && chunk_tensor && chunk_start && step_scalars,
                 "gs_optim_adam: null table");
    GS_CHECK_ARG(nchunks >= 0 && eps >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && weight_decay >= 0.f,
                 "gs_optim_adam: bad hyper-parameters");
    if (nchunks == 0) return GS_OK;
    const OptTable tb{params, grads, exp_avg, exp_avg_sq, sizes, chunk_tensor, chunk_start};
    const AdamHyper h{beta1, beta2, eps, weight_decay, grad_scale};
    adam_kernel<<<nchunks, 256, 0, (hipStream_t)stream>>>(tb, step_scalars, h);
    GS_CHECK_LAUNCH("gs_optim_adam");
    return GS_OK;
}
