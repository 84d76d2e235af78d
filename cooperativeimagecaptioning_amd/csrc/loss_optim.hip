// Sequence-level loss assembly (self-critical / REINFORCE / masked NLL) and the fused
// clamp + Adam update over a flat parameter buffer.
#include "cic_common.h"

namespace {

struct LossTerms {
    const float* term[CIC_LOSS_MAX_TERMS];
    float weight[CIC_LOSS_MAX_TERMS];
    int count;
};

// loss = sum_{b, t<L} slp[b,t] * coef[b] * m[b,t] / sum m,   m[b,0] = 1, m[b,t] = seq[b,t-1] > 0
//   (gen_masks[:, 1:] of models/AlternatingJointModel.py:353-355; :292-297,:321-325,:421-428)
// dslp (+)= weight * coef[b] * m[b,t] / sum m
__global__ __launch_bounds__(1024) void seq_loss_kernel(const float* __restrict__ slp, const int32_t* __restrict__ seq,
                                                        const int32_t* __restrict__ Lp, const float* __restrict__ coef,
                                                        float coef_sign, float weight, int B, int T,
                                                        float* __restrict__ loss_out, float* __restrict__ dslp,
                                                        int accumulate, LossTerms lt, float w_self,
                                                        float* __restrict__ total) {
    // total (optional): the step's loss sum_i lt.weight[i] * lt.term[i][0] + w_self * (this term), added in that order - what
    // loss_combine_kernel computes when this term is the last one (AlternatingJointModel.py:470-503), without its launch
    // one workgroup of 16 waves: at B x T = 2048 every thread owns two elements and the kernel is two round trips (loads,
    // stores) around one barrier - with 256 threads it was eight dependent trips
    constexpr int NT = 1024;
    __shared__ float sh[2][NT / 64];
    const int L = min(*Lp, T);
    float num = 0.f, den = 0.f;
    for (int i = threadIdx.x; i < B * T; i += NT) {
        const int b = i / T, t = i % T;
        const bool m = t < L && (t == 0 || seq[(size_t)b * T + t - 1] > 0);
        if (m) {
            num += slp[i] * (coef_sign * coef[b]);
            den += 1.f;
        }
    }
    num = wave_sum(num);
    den = wave_sum(den);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = num; sh[1][threadIdx.x >> 6] = den; }
    __syncthreads();
    num = 0.f; den = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { num += sh[0][w]; den += sh[1][w]; }
    if (threadIdx.x == 0 && loss_out) *loss_out = num / den;
    if (threadIdx.x == 0 && total) {
        float t = 0.f;
        for (int i = 0; i < lt.count; ++i) t += lt.weight[i] * lt.term[i][0];
        *total = t + w_self * (num / den);
    }
    if (dslp) {
        const float k = weight / den;
        for (int i = threadIdx.x; i < B * T; i += NT) {
            const int b = i / T, t = i % T;
            const bool m = t < L && (t == 0 || seq[(size_t)b * T + t - 1] > 0);
            const float g = m ? k * coef_sign * coef[b] : 0.f;
            dslp[i] = accumulate ? dslp[i] + g : g;
        }
    }
}

// LanguageModelCriterion (misc/utils.py:49-58): loss = -sum slp*mask / sum mask; dslp = -weight*mask/sum mask
__global__ __launch_bounds__(256) void masked_nll_kernel(const float* __restrict__ slp, const float* __restrict__ mask,
                                                         int mask_ld, float weight, int B, int T,
                                                         float* __restrict__ loss_out, float* __restrict__ dslp) {
    __shared__ float sh[2][4];
    float num = 0.f, den = 0.f;
    for (int i = threadIdx.x; i < B * T; i += 256) {
        const int b = i / T, t = i % T;
        const float m = mask[(size_t)b * mask_ld + t];
        num -= slp[i] * m;
        den += m;
    }
    num = wave_sum(num);
    den = wave_sum(den);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = num; sh[1][threadIdx.x >> 6] = den; }
    __syncthreads();
    num = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    den = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    if (threadIdx.x == 0 && loss_out) *loss_out = num / den;
    if (dslp)
        for (int i = threadIdx.x; i < B * T; i += 256) {
            const int b = i / T, t = i % T;
            dslp[i] = -weight * mask[(size_t)b * mask_ld + t] / den;
        }
}

// torch's clamp_ (misc/utils.py:65-69) keeps a NaN a NaN; fminf(fmaxf(NaN, -c), c) would hand Adam a finite -c instead
__device__ __forceinline__ float clamp_nan(float x, float clip) { return x != x ? x : fminf(fmaxf(x, -clip), clip); }

// clip_gradient (elementwise clamp, misc/utils.py:65-69) + torch.optim.Adam.step (optimizer.py:25-27)
// ZERO: the gradient is cleared on the way out (the next step's zero_grad(), optimizer.py:224-230, without a pass of its own)
// status: the caller's sticky status word (cic.h) or null.  Non-zero = a one-launch recurrence of this or an earlier step gave up
// on a hand-off and poisoned its gradients: nothing is touched (the word is read by every thread: one cached line).
template <bool ZERO>
__global__ __launch_bounds__(256) void clamp_adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                         float clip, float wd, float b1, float b2, float eps,
                                                         float step_size, float bc2_sqrt, float gscale,
                                                         unsigned* __restrict__ status) {
    const int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = i4 * 4;
    if (status) {
        const unsigned st = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (st & ~(unsigned)CIC_STATUS_UPDATE_SKIPPED) {
            if (i4 == 0) __hip_atomic_fetch_or(status, (unsigned)CIC_STATUS_UPDATE_SKIPPED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
    if (i >= n) return;
    if (i + 3 < n) {
        f32x4 pp = *reinterpret_cast<f32x4*>(p + i);
        const f32x4 gg = *reinterpret_cast<const f32x4*>(g + i);
        f32x4 mm = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = clamp_nan(gg[e] * gscale, clip);
            if (wd != 0.f) x += wd * pp[e];
            mm[e] = mm[e] * b1 + (1.0f - b1) * x;
            vv[e] = vv[e] * b2 + (1.0f - b2) * x * x;
            const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
            pp[e] = pp[e] - step_size * (mm[e] / denom);
        }
        *reinterpret_cast<f32x4*>(p + i) = pp;
        *reinterpret_cast<f32x4*>(m + i) = mm;
        *reinterpret_cast<f32x4*>(v + i) = vv;
        if (ZERO) *reinterpret_cast<f32x4*>(g + i) = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
        for (int64_t j = i; j < n; ++j) {
            float x = clamp_nan(g[j] * gscale, clip);
            if (wd != 0.f) x += wd * p[j];
            m[j] = m[j] * b1 + (1.0f - b1) * x;
            v[j] = v[j] * b2 + (1.0f - b2) * x * x;
            const float denom = sqrtf(v[j]) / bc2_sqrt + eps;
            p[j] = p[j] - step_size * (m[j] / denom);
            if (ZERO) g[j] = 0.f;
        }
    }
}

__global__ void loss_combine_kernel(LossTerms lt, float* __restrict__ total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < lt.count; ++i) t += lt.weight[i] * lt.term[i][0];
        total[0] = t;
    }
}

}  // namespace

extern "C" int cic_loss_combine(const float* const* term, const float* weight, int count, float* total, cic_stream_t s) {
    CIC_REQUIRE(term && weight && total && count >= 1 && count <= CIC_LOSS_MAX_TERMS);
    LossTerms lt = {};
    lt.count = count;
    for (int i = 0; i < count; ++i) {
        CIC_REQUIRE(term[i]);
        lt.term[i] = term[i];
        lt.weight[i] = weight[i];
    }
    hipLaunchKernelGGL(loss_combine_kernel, dim3(1), dim3(64), 0, cic_s(s), lt, total);
    CIC_LAUNCH_CHECK();
    return 0;
}

extern "C" int cic_seq_loss(const float* slp, const int32_t* seq, const int32_t* L, const float* coef, float coef_sign,
                            float weight, int B, int T, float* loss_out, float* dslp, int accumulate, cic_stream_t s) {
    return cic_seq_loss_total(slp, seq, L, coef, coef_sign, weight, B, T, loss_out, dslp, accumulate, nullptr, nullptr, 0, 0.f,
                              nullptr, s);
}

extern "C" int cic_seq_loss_total(const float* slp, const int32_t* seq, const int32_t* L, const float* coef, float coef_sign,
                                  float weight, int B, int T, float* loss_out, float* dslp, int accumulate,
                                  const float* const* term, const float* term_weight, int count, float self_weight,
                                  float* total, cic_stream_t s) {
    CIC_REQUIRE(slp && seq && L && coef && B > 0 && T > 0);
    CIC_REQUIRE(count >= 0 && count < CIC_LOSS_MAX_TERMS && (count == 0 || (term && term_weight)));
    LossTerms lt = {};
    lt.count = total ? count : 0;
    for (int i = 0; i < lt.count; ++i) {
        CIC_REQUIRE(term[i]);
        lt.term[i] = term[i];
        lt.weight[i] = term_weight[i];
    }
    hipLaunchKernelGGL(seq_loss_kernel, dim3(1), dim3(1024), 0, cic_s(s), slp, seq, L, coef, coef_sign, weight, B, T,
                       loss_out, dslp, accumulate, lt, self_weight, total);
    CIC_LAUNCH_CHECK();
    return 0;
}

extern "C" int cic_masked_nll(const float* slp, const float* mask, int mask_ld, float weight, int B, int T,
                              float* loss_out, float* dslp, cic_stream_t s) {
    CIC_REQUIRE(slp && mask && B > 0 && T > 0 && mask_ld >= T);
    hipLaunchKernelGGL(masked_nll_kernel, dim3(1), dim3(256), 0, cic_s(s), slp, mask, mask_ld, weight, B, T, loss_out,
                       dslp);
    CIC_LAUNCH_CHECK();
    return 0;
}

extern "C" int cic_clamp_adam(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                              double beta2, double eps, double weight_decay, double grad_clip, int step,
                              double grad_scale, cic_stream_t s) {
    return cic_clamp_adam_zero(p, const_cast<float*>(g), m, v, n, lr, beta1, beta2, eps, weight_decay, grad_clip, step,
                               grad_scale, 0, s);
}

extern "C" int cic_clamp_adam_zero(float* p, float* g, float* m, float* v, int64_t n, double lr, double beta1,
                                   double beta2, double eps, double weight_decay, double grad_clip, int step,
                                   double grad_scale, int zero_grad, cic_stream_t s) {
    return cic_clamp_adam_guarded(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, grad_clip, step, grad_scale, zero_grad,
                                  nullptr, s);
}

extern "C" int cic_clamp_adam_guarded(float* p, float* g, float* m, float* v, int64_t n, double lr, double beta1,
                                      double beta2, double eps, double weight_decay, double grad_clip, int step,
                                      double grad_scale, int zero_grad, uint32_t* status, cic_stream_t s) {
    CIC_REQUIRE(p && g && m && v && n > 0 && step >= 1 && grad_clip > 0.0);
    CIC_REQUIRE(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                  reinterpret_cast<uintptr_t>(v)) & 15) == 0);
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    const float step_size = (float)(lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    if (zero_grad)
        hipLaunchKernelGGL(clamp_adam_kernel<true>, dim3(cic_cdiv((n + 3) / 4, 256)), dim3(256), 0, cic_s(s), p, g, m, v, n,
                           (float)grad_clip, (float)weight_decay, (float)beta1, (float)beta2, (float)eps, step_size, bc2_sqrt,
                           (float)grad_scale, status);
    else
        hipLaunchKernelGGL(clamp_adam_kernel<false>, dim3(cic_cdiv((n + 3) / 4, 256)), dim3(256), 0, cic_s(s), p, g, m, v, n,
                           (float)grad_clip, (float)weight_decay, (float)beta1, (float)beta2, (float)eps, step_size, bc2_sqrt,
                           (float)grad_scale, status);
    CIC_LAUNCH_CHECK();
    return 0;
}
