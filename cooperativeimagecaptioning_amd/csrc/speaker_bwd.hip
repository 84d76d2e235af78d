// Speaker (att2in2) backward kernels: hand-written reverse of speaker_fwd.hip.
//   sampler_bwd   d(one-hot rows, sampled log-probs) -> d logits          (gumbel.py:17-30, multinomial.py:4-27)
//   cell_bwd      maxout-LSTM cell pointwise part                          (AttModel.py:515-529)
//   attn_bwd      per-step attention: d att_res -> d att_h, d dot          (AttModel.py:465-489)
//   attn_bwd_feats  after the time loop: d att, d p_att, d alpha_net       (sum over all steps in one pass)
//   embed_bwd, relu_keep_bwd
// The time loop only carries what is truly recurrent (dh, dc); every weight gradient and the
// two [B,K,H] feature gradients are formed once after the loop from saved per-step slabs.
#include "cic_common.h"
#include "engine_util.h"

namespace {

CIC_SWITCH(g_bptt_early_stop, 1);   // development build: cic_debug_bptt_early_stop(0) = the products of every BPTT step run (A/B)
CIC_SWITCH(g_bptt_seq, 1);           // development build: cic_debug_bptt_seq(0) = the BPTT loop as four launches per step (A/B, parity)


constexpr int SNW = 16;   // waves per row workgroup of sampler_bwd_kernel
__device__ __forceinline__ float block_sum4(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int w = 0; w < SNW; ++w) r += sh[w];
    return r;
}
__device__ __forceinline__ float block_max4(float v, float* sh) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sh[0];
#pragma unroll
    for (int w = 1; w < SNW; ++w) r = fmaxf(r, sh[w]);
    return r;
}

// One workgroup per (t, b) row of the vocabulary.
//   y = softmax(z), z = (logp + g)/tau (gumbel-ST) or logp/tau (multinomial-ST)
//   one_hot = (y_hard - y).detach() + y  =>  d y = G (masked by `unfinished`; finished rows were
//   overwritten by the constant EOS one-hot, AttModel.py:416-420)
//   d logits_j = unf * y_j (G_j - sum_i y_i G_i) / tau + dslp * ([j == it] - exp(logp_j))
template <int RV>
__global__ __launch_bounds__(SNW * 64) __attribute__((amdgpu_waves_per_eu(RV <= 3 ? 8 : 4))) void sampler_bwd_kernel(const float* __restrict__ logp_all,   // [T,B,V1]
                                                          const float* __restrict__ U,          // [T+1,B,V1] or null
                                                          const float* __restrict__ G,          // [T,B,V1] or null
                                                          const int32_t* __restrict__ it_all,   // [T+1,B]
                                                          const int64_t* __restrict__ target,   // [T+1,B] teacher targets or null
                                                          const int32_t* __restrict__ seq,      // [B,T] or null
                                                          const float* __restrict__ dslp,       // [B,T] or null
                                                          const int32_t* __restrict__ Lp, int mode, float tau,
                                                          float* __restrict__ dlogits,          // [T,B,V1] (may alias G)
                                                          int T, int B, int V1,
                                                          // logp_all holds RAW logits and lse_all their [T,B] log-sum-exp
                                                          // (row-wise decodes), or log-probs and NULL
                                                          const float* __restrict__ lse_all,
                                                          // U == NULL: the uniforms are drawn here, element i of the
                                                          // [T+1,B,V1] slab = element u_elem0 + i of the Philox stream
                                                          int u_philox, uint64_t u_seed, uint64_t u_elem0,
                                                          // != 0: the forward masked column seq[b, t-1] of row (t >= 1, b)
                                                          // to -inf before the log-softmax (AttModel.py:438-442)
                                                          int decoding_constraint,
                                                          // [1] factor on dslp (the upstream gradient of the loss) or null
                                                          const float* __restrict__ dslp_scale,
                                                          // hand-off counters of spk_bptt_seq_kernel, cleared here (this
                                                          // launch precedes it on the stream) or null
                                                          unsigned* __restrict__ zsync, int nzsync,
                                                          // [T*B, zrow_n] cleared here for the K-sliced product d out = d logits W
                                                          // that follows (its partial tiles are added into it), or null
                                                          float* __restrict__ zrow, int zrow_n) {
    constexpr int NT = SNW * 64;
    __shared__ float sh[SNW];
    const int row = blockIdx.x, t = row / B, b = row % B, tid = threadIdx.x;
    if (blockIdx.x == 0 && zsync)
        for (int i = tid; i < nzsync; i += NT) zsync[i] = 0u;
    if (zrow)
        for (int i = tid; i < zrow_n; i += NT) zrow[(size_t)row * zrow_n + i] = 0.f;
    // the constrained column carries log p = -inf: y = 0, p = 0, no gradient (lse_all was taken without it; a row kernel
    // that stored log-probs wrote -inf there itself)
    const int cons = (decoding_constraint && seq && t >= 1) ? seq[(size_t)b * T + (t - 1)] : -1;
    const float lse = lse_all ? lse_all[row] : 0.f;
    const int L = Lp ? *Lp : T;
    const float* lp = logp_all + (size_t)row * V1;
    float* out = dlogits + (size_t)row * V1;
    const float* g = G ? G + (size_t)row * V1 : nullptr;
    // the log-prob that was gathered: the fed token, or the label under teacher forcing (scheduled sampling
    // may feed a different token than the target)
    const int it = target ? (int)target[(size_t)(t + 1) * B + b] : it_all[(size_t)(t + 1) * B + b];
    const float ds = (dslp && t < L) ? dslp[(size_t)b * T + t] * (dslp_scale ? *dslp_scale : 1.0f) : 0.f;
    const bool st_mode = (mode == CIC_SAMPLE_GUMBEL_ST || mode == CIC_SAMPLE_MULTINOMIAL_ST);
    const bool unf = st_mode && g && seq && t < L && seq[(size_t)b * T + t] > 0;
    const int nq = (V1 + 3) >> 2;
    // rows of whole, 16-byte aligned float4 (the vocabulary rows of the engines): float4 traffic, and every load of the
    // row goes out before the first use (element loads under column conditions compile to one load + full wait each:
    // 36 serial round trips per thread at RV = 3)
    const bool fast = (V1 & 3) == 0 && ((reinterpret_cast<uintptr_t>(logp_all) | reinterpret_cast<uintptr_t>(dlogits) |
                                         reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(U)) & 15) == 0 &&
                      (U || !u_philox || (u_elem0 & 3) == 0);
    const uint64_t urow_elem = u_elem0 + ((uint64_t)(t + 1) * (uint64_t)B + (uint64_t)b) * (uint64_t)V1;
    if (!unf && ds == 0.f) {   // block-uniform: nothing flows into this row
        if (fast) {
            for (int q = tid; q < nq; q += NT) reinterpret_cast<f32x4*>(out)[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            for (int c = tid; c < V1; c += NT) out[c] = 0.f;
        }
        return;
    }
    const float inv_t = 1.0f / tau;
    const float* urow = U ? U + ((size_t)(t + 1) * B + b) * V1 : nullptr;
    float x[RV][4], y[RV][4], gg[RV][4];
    float zm = -INFINITY;
    if (fast) {
        const bool gum = unf && mode == CIC_SAMPLE_GUMBEL_ST;
        f32x4 xv[RV], gv[RV], uv[RV];
#pragma unroll
        for (int r = 0; r < RV; ++r) {
            const int q = tid + NT * r;
            xv[r] = reinterpret_cast<const f32x4*>(lp)[q < nq ? q : nq - 1];
        }
        if (unf) {
#pragma unroll
            for (int r = 0; r < RV; ++r) {
                const int q = tid + NT * r;
                gv[r] = reinterpret_cast<const f32x4*>(g)[q < nq ? q : nq - 1];
            }
        }
        if (gum) {
#pragma unroll
            for (int r = 0; r < RV; ++r) {
                const int q = tid + NT * r;
                const int qc = q < nq ? q : nq - 1;
                if (urow) uv[r] = reinterpret_cast<const f32x4*>(urow)[qc];
                else uv[r] = philox_uniform4(u_seed, (urow_elem >> 2) + (uint64_t)qc);   // one call = the float4's four uniforms
            }
        }
#pragma unroll
        for (int r = 0; r < RV; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = (tid + NT * r) < nq && 4 * (tid + NT * r) + e != cons;
                x[r][e] = ok ? xv[r][e] - lse : -INFINITY;
                gg[r][e] = (ok && unf) ? gv[r][e] : 0.f;
                float z = -INFINITY;
                if (ok && unf) z = gum ? (x[r][e] + gumbel_from_u(uv[r][e])) * inv_t : x[r][e] * inv_t;
                y[r][e] = z;
                zm = fmaxf(zm, z);
            }
    } else {
#pragma unroll
        for (int r = 0; r < RV; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 4 * (tid + NT * r) + e;
                const bool ok = (tid + NT * r) < nq && c < V1 && c != cons;
                x[r][e] = ok ? lp[c] - lse : -INFINITY;
                gg[r][e] = (ok && unf) ? g[c] : 0.f;
                float z = -INFINITY;
                if (ok && unf) {
                    if (mode == CIC_SAMPLE_GUMBEL_ST) {
                        float u;
                        if (urow) u = urow[c];
                        else { const uint64_t el = urow_elem + (uint64_t)c; u = philox_uniform4(u_seed, el >> 2)[el & 3]; }
                        z = (x[r][e] + gumbel_from_u(u)) * inv_t;
                    } else {
                        z = x[r][e] * inv_t;
                    }
                }
                y[r][e] = z;
                zm = fmaxf(zm, z);
            }
    }
    float cdot = 0.f, ysum = 1.f;
    if (unf) {
        zm = block_max4(zm, sh);
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < RV; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[r][e] = __expf(y[r][e] - zm);
                s += y[r][e];
            }
        ysum = block_sum4(s, sh);
        const float inv = 1.0f / ysum;
        float c = 0.f;
#pragma unroll
        for (int r = 0; r < RV; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[r][e] *= inv;
                c += y[r][e] * gg[r][e];
            }
        cdot = block_sum4(c, sh);
    }
#pragma unroll
    for (int r = 0; r < RV; ++r) {
        f32x4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * (tid + NT * r) + e;
            float v = 0.f;
            if (unf) v = y[r][e] * (gg[r][e] - cdot) * inv_t;
            if (ds != 0.f) v += ds * ((c == it ? 1.f : 0.f) - __expf(x[r][e]));
            o4[e] = v;
            if (!fast && (tid + NT * r) < nq && c < V1) out[c] = v;
        }
        if (fast && (tid + NT * r) < nq) reinterpret_cast<f32x4*>(out)[tid + NT * r] = o4;
    }
}


// Partial-sampling modes, one core step per launch (one workgroup per image row b of step t).
//   soft row  o = (m y_hard - m y).detach() + y  =>  d y = G    (gumbel_softmax.py:30-41, multinomial_soft.py:23-33)
//   G = [unfinished] * d soft_out[t,b,:]  (finished rows were replaced by the constant EOS one-hot, AttModel.py:428-432)
//       + d soft_raw[t,b,:] from the next step's input xt = relu_dropout(soft_raw @ embed)   (:395-397; already in dl)
//   gumbel_ps      y = softmax((logp+g)/tau):  d logp_j = y_j (G_j - sum_i y_i G_i) / tau      (zero-sum)
//   multinomial_ps y = exp(logp/tau):          d logp_j = y_j G_j / tau                        (not zero-sum)
//   d logits_j = d logp_j - exp(logp_j) sum_i d logp_i  +  dslp ([j == it] - exp(logp_j))
template <int RV>
__global__ __launch_bounds__(SNW * 64) void sampler_ps_bwd_kernel(const float* __restrict__ logp,     // [B,V1] step t
                                                             const float* __restrict__ U,        // [B,V1] row t+1 or null
                                                             const float* __restrict__ G,        // [B,V1] d soft_out[t] or null
                                                             float* __restrict__ dl,             // [B,V1] in: recurrent part (rec), out: d logits
                                                             int rec, const int32_t* __restrict__ it_next,   // [B] it_all[t+1]
                                                             const int32_t* __restrict__ seq,    // [B,T]
                                                             const float* __restrict__ dslp,     // [B,T] or null
                                                             const int32_t* __restrict__ Lp, int t, int mode, float tau,
                                                             int T, int V1) {
    constexpr int NT = SNW * 64;
    __shared__ float sh[SNW];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int L = *Lp;
    const float* lp = logp + (size_t)b * V1;
    float* out = dl + (size_t)b * V1;
    if (t >= L) {   // block-uniform: the reference never ran this step's sampler with gradient (break, AttModel.py:407)
        for (int c = tid; c < V1; c += NT) out[c] = 0.f;
        return;
    }
    const float* g = G ? G + (size_t)b * V1 : nullptr;
    const float* urow = U ? U + (size_t)b * V1 : nullptr;
    const int it = it_next[b];
    const float ds = dslp ? dslp[(size_t)b * T + t] : 0.f;
    const bool unf = g && seq[(size_t)b * T + t] > 0;
    const bool gum = mode == CIC_SAMPLE_GUMBEL_PS;
    const int nq = (V1 + 3) >> 2;
    const float inv_t = 1.0f / tau;
    float x[RV][4], y[RV][4], gg[RV][4];
    float zm = -INFINITY;
#pragma unroll
    for (int r = 0; r < RV; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * (tid + NT * r) + e;
            const bool ok = (tid + NT * r) < nq && c < V1;
            x[r][e] = ok ? lp[c] : -INFINITY;
            float gv = 0.f;
            if (ok && unf) gv = g[c];
            if (ok && rec) gv += out[c];
            gg[r][e] = gv;
            float z = -INFINITY;
            if (ok) z = gum ? (x[r][e] + gumbel_from_u(urow[c])) * inv_t : x[r][e] * inv_t;
            y[r][e] = z;
            zm = fmaxf(zm, z);
        }
    float corr = 0.f;   // sum_i d logp_i (multinomial_ps) or sum_i y_i G_i (gumbel_ps)
    if (gum) {
        zm = block_max4(zm, sh);
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < RV; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[r][e] = __expf(y[r][e] - zm);
                s += y[r][e];
            }
        const float inv = 1.0f / block_sum4(s, sh);
        float c = 0.f;
#pragma unroll
        for (int r = 0; r < RV; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[r][e] *= inv;
                c += y[r][e] * gg[r][e];
            }
        corr = block_sum4(c, sh);
    } else {
        float c = 0.f;
#pragma unroll
        for (int r = 0; r < RV; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y[r][e] = __expf(y[r][e]);          // exp(-inf) = 0 for the padding lanes
                c += y[r][e] * gg[r][e] * inv_t;
            }
        corr = block_sum4(c, sh);
    }
#pragma unroll
    for (int r = 0; r < RV; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * (tid + NT * r) + e;
            if ((tid + NT * r) < nq && c < V1) {
                const float pj = __expf(x[r][e]);
                float v = gum ? y[r][e] * (gg[r][e] - corr) * inv_t : y[r][e] * gg[r][e] * inv_t - pj * corr;
                if (ds != 0.f) v += ds * ((c == it ? 1.f : 0.f) - pj);
                out[c] = v;
            }
        }
}

// ---- cell backward -----------------------------------------------------------------------
__global__ __launch_bounds__(256) void cell_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ c_prev,
                                                       const float* __restrict__ c_new, const float* __restrict__ d_out,
                                                       const float* __restrict__ dh_carry, float* __restrict__ dc_carry,
                                                       const uint8_t* __restrict__ keep, float scale,
                                                       float* __restrict__ dpre, int B, int H, int first,
                                                       float* __restrict__ zero_a, float* __restrict__ zero_b,
                                                       int state_dropped) {
    // state_dropped: the recurrent state is the dropped-out h (LSTMCore, models/FCModel.py:38-42), so the carried
    // dh passes through the dropout mask as well
    // first != 0: dh_carry / dc_carry hold nothing yet (last time step)
    // zero_a / zero_b: [B,H] buffers cleared here for the K-split products that follow in this time step (their
    // partial tiles are added into them), or null
    const int H4 = H >> 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * H4) return;
    const int b = idx / H4, j = idx % H4;
    const f32x4* p = reinterpret_cast<const f32x4*>(pre + (size_t)b * 5 * H);
    const f32x4 pi = p[j], pf = p[H4 + j], po = p[2 * H4 + j], pa = p[3 * H4 + j], pb = p[4 * H4 + j];
    const f32x4 cp = reinterpret_cast<const f32x4*>(c_prev)[idx];
    const f32x4 cn = reinterpret_cast<const f32x4*>(c_new)[idx];
    const f32x4 dout = reinterpret_cast<const f32x4*>(d_out)[idx];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 dhc = first ? zero : reinterpret_cast<const f32x4*>(dh_carry)[idx];
    const f32x4 dcc = first ? zero : reinterpret_cast<const f32x4*>(dc_carry)[idx];
    uint32_t kp = 0x01010101u;
    if (keep) kp = *reinterpret_cast<const uint32_t*>(keep + (size_t)idx * 4);
    f32x4 gi, gf, go, ga, gb, dcp;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float ig = fast_sigmoid(pi[e]), fg = fast_sigmoid(pf[e]), og = fast_sigmoid(po[e]);
        const float g = fmaxf(pa[e], pb[e]);
        const float tc = fast_tanh(cn[e]);
        const float kf = keep ? (float)((kp >> (8 * e)) & 0xffu) * scale : 1.0f;
        const float dh = state_dropped ? (dout[e] + dhc[e]) * kf : dout[e] * kf + dhc[e];
        const float dc = dcc[e] + dh * og * (1.0f - tc * tc);
        gi[e] = dc * g * ig * (1.0f - ig);
        gf[e] = dc * cp[e] * fg * (1.0f - fg);
        go[e] = dh * tc * og * (1.0f - og);
        const float dg = dc * ig;
        // torch.max(a, b) backward: larger gets it, exact tie splits it evenly
        ga[e] = pa[e] > pb[e] ? dg : (pa[e] == pb[e] ? 0.5f * dg : 0.f);
        gb[e] = pb[e] > pa[e] ? dg : (pa[e] == pb[e] ? 0.5f * dg : 0.f);
        dcp[e] = dc * fg;
    }
    f32x4* o = reinterpret_cast<f32x4*>(dpre + (size_t)b * 5 * H);
    o[j] = gi; o[H4 + j] = gf; o[2 * H4 + j] = go; o[3 * H4 + j] = ga; o[4 * H4 + j] = gb;
    reinterpret_cast<f32x4*>(dc_carry)[idx] = dcp;
    if (zero_a) reinterpret_cast<f32x4*>(zero_a)[idx] = zero;
    if (zero_b) reinterpret_cast<f32x4*>(zero_b)[idx] = zero;
}

// ---- per-step attention backward ---------------------------------------------------------
// in : d_att_res[b,:], alpha[b,:], att_h[b,:], p_att[b], att[b]
// out: ddot[b,k] = alpha_k (dalpha_k - sum_j alpha_j dalpha_j),  dalpha_k = d_att_res . att[b,k,:]
//      d_att_h[b,a] = w_a * sum_k ddot_k (1 - tanh^2(p_att[b,k,a] + att_h[b,a]))
// (a masked, renormalised softmax is alpha_k ~ m_k e^{dot_k}: same Jacobian in terms of the final alpha)
template <int NI, int KPW, int NW, bool HOLD>
__global__ __launch_bounds__(NW * 64) void attn_bwd_kernel(const float* __restrict__ d_att_res, const float* __restrict__ alpha,
                                                       const float* __restrict__ att_h, const float* __restrict__ p_att,
                                                       const float* __restrict__ att, const float* __restrict__ w_alpha,
                                                       float* __restrict__ d_att_h, float* __restrict__ ddot_out, int K,
                                                       int A, int H) {
    __shared__ float sd[64];
    __shared__ __attribute__((aligned(16))) float sacc[NW * NI * 256];
    __shared__ __attribute__((aligned(16))) float sacc2[4 * NI * 256];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int A4 = A >> 2, H4 = H >> 2;
    const f32x4* pa4 = reinterpret_cast<const f32x4*>(p_att + (size_t)b * K * A);
    const f32x4* at4 = reinterpret_cast<const f32x4*>(att + (size_t)b * K * H);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 dr[NI], ah[NI], wa[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
        dr[i] = c < H4 ? reinterpret_cast<const f32x4*>(d_att_res + (size_t)b * H)[c] : z4;
        ah[i] = c < A4 ? reinterpret_cast<const f32x4*>(att_h + (size_t)b * A)[c] : z4;
        wa[i] = c < A4 ? reinterpret_cast<const f32x4*>(w_alpha)[c] : z4;
    }
    f32x4 pv[HOLD ? KPW : 1][NI];
#pragma unroll
    for (int j = 0; j < KPW; ++j) {
        const int k = w + NW * j;
        if (k < K) {
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int c = lane + 64 * i;
                const f32x4 a = c < H4 ? at4[(size_t)k * H4 + c] : z4;
                if (HOLD) pv[j][i] = c < A4 ? pa4[(size_t)k * A4 + c] : z4;
#pragma unroll
                for (int e = 0; e < 4; ++e) part += dr[i][e] * a[e];
            }
            part = wave_sum(part);
            if (lane == 0) sd[k] = part;
        }
    }
    __syncthreads();
    float cs = 0.f;
    for (int k = 0; k < K; ++k) cs += alpha[(size_t)b * K + k] * sd[k];
    f32x4 acc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] = z4;
#pragma unroll
    for (int j = 0; j < KPW; ++j) {
        const int k = w + NW * j;
        if (k < K) {
            const float dd = alpha[(size_t)b * K + k] * (sd[k] - cs);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int c = lane + 64 * i;
                f32x4 p;
                if (HOLD) p = pv[j][i];
                else p = c < A4 ? pa4[(size_t)k * A4 + c] : z4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float th = fast_tanh(p[e] + ah[i][e]);
                    acc[i][e] += dd * (1.0f - th * th);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][e] *= wa[i][e];
        *reinterpret_cast<f32x4*>(&sacc[(w * NI * 64 + i * 64 + lane) * 4]) = acc[i];
    }
    __syncthreads();
    for (int t = tid; t < 4 * A4; t += NW * 64) {
        const int part = t / A4, c = t % A4;
        const int i = c >> 6, l = c & 63;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NW / 4; ++q)
            s += *reinterpret_cast<f32x4*>(&sacc[(((part * (NW / 4) + q) * NI + i) * 64 + l) * 4]);
        *reinterpret_cast<f32x4*>(&sacc2[((part * NI + i) * 64 + l) * 4]) = s;
    }
    __syncthreads();
    for (int c = tid; c < A4; c += NW * 64) {
        const int i = c >> 6, l = c & 63;
        f32x4 s = *reinterpret_cast<f32x4*>(&sacc2[((0 * NI + i) * 64 + l) * 4]);
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) s += *reinterpret_cast<f32x4*>(&sacc2[((ww * NI + i) * 64 + l) * 4]);
        reinterpret_cast<f32x4*>(d_att_h + (size_t)b * A)[c] = s;
    }
    if (tid < K) ddot_out[(size_t)b * K + tid] = alpha[(size_t)b * K + tid] * (sd[tid] - cs);
}


// column-owner layout of attn_bwd (see attn_fwd_cols_kernel in speaker_fwd.hip): A == H, H % 32 == 0
template <int JMAX, bool TWIN>
__global__ __launch_bounds__(1024) void attn_bwd_cols_kernel(const float* __restrict__ d_att_res, const float* __restrict__ alpha,
                                                             const float* __restrict__ att_h, const float* __restrict__ p_att,
                                                             const float* __restrict__ att, const float* __restrict__ w_alpha,
                                                             float* __restrict__ d_att_h, float* __restrict__ ddot_out, int K,
                                                             int H) {
    __shared__ float sp[16 * 64];
    const int b = TWIN ? blockIdx.x >> 1 : blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int NW = blockDim.x >> 6;
    const int half = TWIN ? (blockIdx.x & 1) : 0;
    const bool owner = !TWIN || ((2 * w) / NW == half);
    const int c = lane & 7, rg = lane >> 3;
    const int H4 = H >> 2;
    const int col4 = 8 * w + c;
    const f32x4* pa4 = reinterpret_cast<const f32x4*>(p_att + (size_t)b * K * H);
    const f32x4* at4 = reinterpret_cast<const f32x4*>(att + (size_t)b * K * H);
    const f32x4 dr = reinterpret_cast<const f32x4*>(d_att_res + (size_t)b * H)[col4];
    const f32x4 ah = reinterpret_cast<const f32x4*>(att_h + (size_t)b * H)[col4];
    const f32x4 wa = reinterpret_cast<const f32x4*>(w_alpha)[col4];
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 pv[JMAX], av[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const int k = 8 * j + rg;
        av[j] = k < K ? at4[(size_t)k * H4 + col4] : z4;
        pv[j] = (owner && k < K) ? pa4[(size_t)k * H4 + col4] : z4;
    }
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        float part = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) part += dr[e] * av[j][e];
        part = sum8_dpp(part);
        const int k = 8 * j + rg;
        if (c == 0 && k < K) sp[w * 64 + k] = part;
    }
    __syncthreads();
    float dal = 0.f, al = 0.f;
    if (lane < K) {
        for (int q = 0; q < NW; ++q) dal += sp[q * 64 + lane];
        al = alpha[(size_t)b * K + lane];
    }
    const float cs = wave_sum_fast(al * dal);
    const float dd = al * (dal - cs);                  // 0 for lanes >= K
    if (w == 0 && half == 0 && lane < K) ddot_out[(size_t)b * K + lane] = dd;
    if (!owner) return;
    f32x4 acc = z4;
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const float d = __shfl(dd, 8 * j + rg, 64);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float th = fast_tanh(pv[j][e] + ah[e]);
            acc[e] += d * (1.0f - th * th);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        acc[e] = sum_over_rg(acc[e]) * wa[e];
    }
    if (rg == 0) reinterpret_cast<f32x4*>(d_att_h + (size_t)b * H)[col4] = acc;
}

// ---- the speaker's BPTT loop in ONE launch ------------------------------------------------------------------------------
// Per time step the plain form launches four kernels (cell backward 5 us, d att_res = d in_transform a2c.W 7 us, attention
// backward 7 us, dh = dpre h2h.W + d_att_h h2att.W 11 us; x T), each re-streaming its weights into a few fat workgroups.
// Here 256 workgroups - a 16-row strip of the batch x a 16-column tile of the H = 512 hidden units, one per CU, all
// resident - walk the whole loop (same scheme as the listener's gru_seq_bwd_kernel):
//   * the COLUMNS jt*16 .. +16 of a2c.W [2H,H], h2h.W [5H,H] and h2att.W [A,H] stay in the workgroup as MFMA B fragments
//     (v_mfma_f32_16x16x4_f32), the contraction index split over its 8 waves: h2h in registers (80 VGPRs per lane), a2c and
//     h2att in LDS in fragment order (96 KB) - read from memory once instead of T times;
//   * dh and dc never go to memory: after the cross-wave sum a lane of waves 0-3 holds dh of ONE (row, unit) and runs the
//     cell backward of the previous step on it in place (five dpre values);
//   * three hand-offs per step inside a strip (32 workgroups, kept on ONE XCD at B = 128: blockIdx % 8), each the
//     write-through-store / drain / barrier / counter-add + one-poller / barrier / sc1-load form of gru_seq_kernel:
//       1. dpre_t rows of the strip (A operand of both products),
//       2. d att_res_t rows (what the attention backward of an image needs),
//       3. d att_h_t rows (A operand of the h2att part);
//     the slabs they travel in are the workspace slabs the batched gradient products read after the loop anyway;
//   * a wave's K slice covers the SAME columns 3H + 128 ks .. of dpre for a2c.W and for the (a, b) rows of h2h.W, so those A
//     fragments are loaded once for both products; the (i, f, o) columns of the dh product wait for nothing but dpre_t and
//     run while the attention results are on their way, only the h2att part follows the third hand-off;
//   * the attention backward of the strip's 16 images runs on the 16 even-numbered workgroups of the strip (8 waves x 64
//     columns, two passes of 36 registers: att for d alpha - requested before the second hand-off is waited for -, then p_att);
//     the odd workgroups do not wait for the second hand-off at all.
// Measured (tools/bptt_stamps.py): 15.3 us per step against 31 us for the four launches; what is left are three dependent
// fabric round trips per step (store - drain - counter - poll - load: ~3 us each).  Moving more of the dh product in front of
// the attention (prefetching its fragments across it) cost registers and made every phase slower: 16.1 us.
// Steps at or beyond the decode's length L carry no gradient (d out = 0, no carry): zeros are stored, no hand-off runs.
// Every spin is bounded (1 s): a workgroup that gives up raises *err and poisons what it produces with NaN.
struct BpttArgs {
    const float *pre_all, *c_all, *alpha_all, *att_h_all, *p_att, *att;     // forward state [T,B,5H] [T+1,B,H] [T,B,K] [T,B,H] [B,K,H] x2
    const uint8_t* out_keep;                                                // [T,B,H] or null
    const float *a2c_w, *h2h_w, *h2att_w, *alpha_w;
    const float* d_out_all;                                                 // [T,B,H]
    float *dpre_all, *d_att_res_all, *d_att_h_all, *ddot_all;               // [T,B,5H] [T,B,H] [T,B,H] [T,B,K]
    float* zero_tbh;                                                        // [T,B,H] cleared on the way (d x of the batched product after the loop) or null
    unsigned* cnt;                                                          // [strips][T][3] counters (zeroed by the launcher)
    HandoffGuard hg;                                                        // error word behind them, status word, spin bound (cic_common.h)
    const int32_t* L;                                                       // the decode's length on the device, or null
    float scale;
    int B, K, T;
    int row0, row_end;                                                      // rows [row0, row_end) of the batch: one launch per row block
};
constexpr int BPTT_GA = 8, BPTT_GI = 12, BPTT_GC = 4;      // k groups of 16 per wave: (a, b) gate columns / (i, f, o) / d_att_h
constexpr size_t BPTT_LDS_BYTES = sizeof(float) * ((size_t)8 * (BPTT_GA + BPTT_GC) * 64 * 4 + 8 * 4 * 64 + 8 * 64);
#ifdef CIC_DEVTOOLS
__device__ unsigned long long* g_bptt_stamps = nullptr;    // development build: [workgroup][step][8] s_memrealtime stamps
#endif
typedef float f32x4acc_b __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_b __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_b __attribute__((ext_vector_type(8)));
typedef float f32x8_b __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8_b to_bf16x8_b(const f32x4 lo, const f32x4 hi) {
    const f32x8_b v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_convertvector(v, bf16x8_b);        // round to nearest even; a plain cast keeps a NaN a NaN
}
// BF (r4, compute_dtype bf16): the stationary weight columns are bf16 MFMA fragments (v_mfma_f32_16x16x32_bf16; all of them in
// 64 VGPRs, none in LDS), the handed-over gradient rows (dpre, d att_h) are rounded to bf16 as they are loaded, accumulation f32:
// 16 bf16 MFMAs per step instead of 128 f32 ones.  The cell backward, the attention backward, dh and dc stay f32.
template <int KS, bool BF>
__global__ __launch_bounds__(KS * 64) void spk_bptt_seq_kernel(BpttArgs a) {
    static_assert(KS == 8, "K slices of 128 / 192 / 64 columns per wave");
    constexpr int H = 512, H5 = 5 * H, TJ = H / 16, GA = BPTT_GA, GI = BPTT_GI, GC = BPTT_GC;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x4* wl = reinterpret_cast<f32x4*>(lds);            // [KS][GA + GC][64]: a2c fragments, then h2att fragments
    float* red = lds + (size_t)KS * (GA + GC) * 64 * 4;   // [KS][4][64]
    float* sp = red + KS * 4 * 64;                        // [KS][64] per-wave partial d alpha
    __shared__ int ok_s;
    const int tid = threadIdx.x, lane = tid & 63, ks = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int B = a.B, K = a.K, T = a.T;
    const int strips = gridDim.x / TJ;
    int strip = blockIdx.x / TJ, jt = blockIdx.x % TJ;
    if (strips <= 8 && (8 % strips) == 0 && (TJ % (8 / strips)) == 0) {     // speed only: a strip on as few XCDs as possible
        const int xs = 8 / strips, xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        strip = xcd / xs;
        jt = (xcd % xs) * (TJ / xs) + local;
    }
    const int RE = a.row_end;
    const int m0 = a.row0 + strip * 16;
    const int col = jt * 16 + li;
    const bool owner = ks < 4;                            // waves 0-3 finish the 16 x 16 outputs: register ks of the tile
    const int orow = m0 + 4 * lq + (ks & 3);
    const int orc = orow < RE ? orow : RE - 1;
    const int mc = min(m0 + li, RE - 1);                  // A rows; rows past the block repeat its last row: their sums are never stored
    unsigned* cnt = a.cnt + (size_t)(a.row0 / 16 + strip) * T * 3;
    // ---- the weight tiles, once --------------------------------------------------------------------------------------
    f32x4 wh_ab[BF ? 1 : GA], wh_ifo[BF ? 1 : GI];
    // BF: k-steps of 32 inside a wave's K slice: [0, GA/2) h2h (a, b) rows, then a2c, then h2h (i, f, o) rows, then h2att
    constexpr int SA = GA / 2, SI = GI / 2, SC = GC / 2;
    bf16x8_b wb[BF ? 2 * SA + SI + SC : 1];
    if (BF) {
        auto gather8 = [&](const float* W, int k0) {      // rows k0 .. k0 + 7 of column col: one fragment
            f32x4 lo, hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) { lo[e] = W[(size_t)(k0 + e) * H + col]; hi[e] = W[(size_t)(k0 + 4 + e) * H + col]; }
            return to_bf16x8_b(lo, hi);
        };
#pragma unroll
        for (int j = 0; j < SA; ++j) {
            wb[BF ? j : 0] = gather8(a.h2h_w, 3 * H + 128 * ks + 32 * j + 8 * lq);
            wb[BF ? SA + j : 0] = gather8(a.a2c_w, 128 * ks + 32 * j + 8 * lq);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < SI; ++j) { wb[BF ? 2 * SA + j : 0] = gather8(a.h2h_w, 192 * ks + 32 * j + 8 * lq); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int j = 0; j < SC; ++j) { wb[BF ? 2 * SA + SI + j : 0] = gather8(a.h2att_w, 64 * ks + 32 * j + 8 * lq); __builtin_amdgcn_sched_barrier(0); }
    } else {
#pragma unroll
    for (int i = 0; i < GA; ++i) {
        const int k = 128 * ks + 16 * i + 4 * lq;
        f32x4 v;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            wh_ab[BF ? 0 : i][s] = a.h2h_w[(size_t)(3 * H + k + s) * H + col];
            v[s] = a.a2c_w[(size_t)(k + s) * H + col];
        }
        wl[(ks * (GA + GC) + i) * 64 + lane] = v;
        __builtin_amdgcn_sched_barrier(0);                // group by group: keeps the live addresses of this one-time gather few
    }
#pragma unroll
    for (int i = 0; i < GI; ++i) {
        const int k = 192 * ks + 16 * i + 4 * lq;
#pragma unroll
        for (int s = 0; s < 4; ++s) wh_ifo[BF ? 0 : i][s] = a.h2h_w[(size_t)(k + s) * H + col];
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < GC; ++i) {
        const int k = 64 * ks + 16 * i + 4 * lq;
        f32x4 v;
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = a.h2att_w[(size_t)(k + s) * H + col];
        wl[(ks * (GA + GC) + GA + i) * 64 + lane] = v;
        __builtin_amdgcn_sched_barrier(0);
    }
    }
    // attention backward: this wave's 64 columns (float4 column col4), lane -> (c = column quad, rg = region group)
    const int ac = lane & 15, rg = lane >> 4;
    const int col4 = 16 * ks + ac;
    const f32x4 wa = reinterpret_cast<const f32x4*>(a.alpha_w)[col4];
    const int img = m0 + (jt >> 1);
    const bool att_wg = (jt & 1) == 0;
    if (tid == 0) ok_s = 1;                               // sticky: a workgroup that has given up once does not wait again
    __syncthreads();
    const int Lv = __builtin_amdgcn_readfirstlane(a.L ? *a.L : T);      // wave-uniform: the step loop stays scalar control flow
    float dh = 0.f, dc = 0.f;                             // carried gradients of this lane's (row, unit) (owners)
    float poison = 0.f;
    unsigned long long* stamps = CIC_STAMP_BUF(g_bptt_stamps);
#define BPTT_STAMP(i) if (stamps && tid == 0) stamps[((size_t)blockIdx.x * T + t) * 8 + (i)] = __builtin_amdgcn_s_memrealtime()
    // hand-off, producer side: what this workgroup stored becomes visible, then ONE lane counts the workgroup in (add != 0)
    auto publish = [&](unsigned* c, bool add) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // EVERY storing wave drains before the signal
        __syncthreads();
        if (tid == 0 && add) handoff_arrive(c, a.hg);
    };
    // ... consumer side: ONE lane polls until `target` workgroups of the strip have published; loads of handed-off bytes
    // (all sc1) come after it
    auto wait_for = [&](unsigned* c, unsigned target) {
        if (tid == 0 && ok_s) ok_s = handoff_poll(c, target, a.hg);
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    // compiler only: no load of handed-off bytes above the poll
        if (!ok_s) poison = __builtin_nanf("");
    };
    // the three hand-off slabs as buffer resources over all T steps, built once from scalar registers; a step's slab is
    // the scalar offset of its loads and stores
    auto uniform_rsrc = [](float* ptr, size_t bytes) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo), 0, (int)bytes, 0x00020000);
    };
    const auto r_dpre = uniform_rsrc(a.dpre_all, (size_t)T * B * H5 * sizeof(float));
    const auto r_dres = uniform_rsrc(a.d_att_res_all, (size_t)T * B * H * sizeof(float));
    const auto r_dah = uniform_rsrc(a.d_att_h_all, (size_t)T * B * H * sizeof(float));
    const int imc = img < RE ? img : RE - 1;
    const auto r_att = uniform_rsrc(const_cast<float*>(a.att) + (size_t)imc * K * H, (size_t)K * H * sizeof(float));
    const auto r_patt = uniform_rsrc(const_cast<float*>(a.p_att) + (size_t)imc * K * H, (size_t)K * H * sizeof(float));
    for (int t = T - 1; t >= 0; --t) {
        BPTT_STAMP(0);
        // per-lane index values made opaque once per step: the address arithmetic that hangs on them is then redone next to
        // each access instead of being hoisted out of the loop into ~100 registers held across all phases (that cost 160
        // spilled VGPRs, most of them reloaded inside the MFMA phase of every step)
        int lq_t = lq, mc_t = mc, orc_t = orc, orow_t = orow, col_t = col, col4_t = col4, rg_t = rg, lane_t = lane;
        asm volatile("" : "+v"(lq_t), "+v"(mc_t), "+v"(orc_t), "+v"(orow_t), "+v"(col_t), "+v"(col4_t), "+v"(rg_t), "+v"(lane_t));
        const size_t rowH = (size_t)t * B * H;
        float* dres = a.d_att_res_all + rowH;
        float* dah = a.d_att_h_all + rowH;
        const int so5 = __builtin_amdgcn_readfirstlane((int)((size_t)t * B * H5 * sizeof(float)));   // slab t of dpre, of the [B,H] slabs
        const int so1 = __builtin_amdgcn_readfirstlane((int)(rowH * sizeof(float)));
        const bool live = t < Lv;
        // ---- 1. cell backward of step t at (orow_t, col_t): cell_bwd_kernel on one element (waves 0-3) ----------------------
        if (owner) {
            const size_t e = (size_t)orc_t * H + col_t;
            const float* pre = a.pre_all + (size_t)t * B * H5 + (size_t)orc_t * H5 + col_t;
            const float pi = pre[0], pf = pre[H], po = pre[2 * H], pa = pre[3 * H], pb = pre[4 * H];
            const float cp = a.c_all[rowH + e], cn = a.c_all[rowH + (size_t)B * H + e];
            const float dout = a.d_out_all[rowH + e];
            const float kf = a.out_keep ? (float)a.out_keep[rowH + e] * a.scale : 1.0f;
            const float ig = fast_sigmoid(pi), fg = fast_sigmoid(pf), og = fast_sigmoid(po);
            const float g = fmaxf(pa, pb);
            const float tc = fast_tanh(cn);
            const float dhh = dout * kf + dh;
            const float dcc = dc + dhh * og * (1.0f - tc * tc);
            float gi = dcc * g * ig * (1.0f - ig);
            float gf = dcc * cp * fg * (1.0f - fg);
            float go = dhh * tc * og * (1.0f - og);
            const float dg = dcc * ig;
            // torch.max(a, b) backward: larger gets it, exact tie splits it evenly
            float ga = pa > pb ? dg : (pa == pb ? 0.5f * dg : 0.f);
            float gb = pb > pa ? dg : (pa == pb ? 0.5f * dg : 0.f);
            dc = dcc * fg;
            if (poison != 0.f) { gi = poison; gf = poison; go = poison; ga = poison; gb = poison; }
            if (orow_t < RE) {
                const int o = (int)(((size_t)orow_t * H5 + col_t) * 4);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gi), r_dpre, o, so5, 16);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gf), r_dpre, o + 4 * H, so5, 16);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, go), r_dpre, o + 8 * H, so5, 16);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ga), r_dpre, o + 12 * H, so5, 16);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gb), r_dpre, o + 16 * H, so5, 16);
                if (!live) { dres[(size_t)orow_t * H + col_t] = 0.f; dah[(size_t)orow_t * H + col_t] = 0.f; }
                if (a.zero_tbh) a.zero_tbh[rowH + (size_t)orow_t * H + col_t] = 0.f;
            }
        }
        if (!live) {            // grid-uniform: nothing but zeros flows through this step (dpre above came out as zeros)
            if (jt == 0)
                for (int i = tid; i < 16 * K; i += KS * 64)
                    if (m0 + i / K < RE) a.ddot_all[((size_t)t * B + m0 + i / K) * K + i % K] = 0.f;
            dh = 0.f;
            continue;
        }
        BPTT_STAMP(1);
        publish(cnt + t * 3 + 0, true);
        wait_for(cnt + t * 3 + 0, TJ);
        BPTT_STAMP(2);
        // ---- 2. the (a, b) gate columns of dpre_t: d att_res = dpre[:, 3H:5H] a2c.W and their part of dh = dpre h2h.W -----------
        f32x4acc_b acc_res = {0.f, 0.f, 0.f, 0.f}, acc_dh = {0.f, 0.f, 0.f, 0.f};
        {
            f32x4 af[GA];
#pragma unroll
            // (BF: the same loads as runs of eight consecutive k: 32 (i >> 1) + 8 lq + 4 (i & 1))
            for (int i = 0; i < GA; ++i)
                af[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    r_dpre, (int)(((size_t)mc_t * H5 + 3 * H + 128 * ks + (BF ? 32 * (i >> 1) + 8 * lq_t + 4 * (i & 1) : 16 * i + 4 * lq_t)) * 4),
                    so5, 16));
            if (BF) {
#pragma unroll
                for (int j = 0; j < SA; ++j) {
                    const bf16x8_b ab = to_bf16x8_b(af[2 * j], af[2 * j + 1]);
                    acc_res = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, wb[BF ? SA + j : 0], acc_res, 0, 0, 0);
                    acc_dh = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, wb[BF ? j : 0], acc_dh, 0, 0, 0);
                }
            } else {
#pragma unroll
            for (int i = 0; i < GA; ++i) {
                const f32x4 bw = wl[(ks * (GA + GC) + i) * 64 + lane_t];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc_res = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bw[s], acc_res, 0, 0, 0);
                    acc_dh = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], wh_ab[BF ? 0 : i][s], acc_dh, 0, 0, 0);
                }
            }
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) red[(ks * 4 + v) * 64 + lane_t] = acc_res[v];
        __syncthreads();
        if (owner) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < KS; ++w) v += red[(w * 4 + ks) * 64 + lane_t];
            if (poison != 0.f) v = poison;
            if (orow_t < RE) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r_dres, (int)(((size_t)orow_t * H + col_t) * 4), so1, 16);
        }
        BPTT_STAMP(3);
        publish(cnt + t * 3 + 1, true);
        // ---- 3. attention backward of image img (attn_bwd_cols_kernel, 8 waves x 64 columns) on the even workgroups -------------
        // Two passes over the image's regions, 36 registers each: att for d alpha (requested before the second hand-off is
        // waited for: it depends on nothing), then p_att for d att_h (requested as soon as the att rows are consumed).  Buffer
        // loads over the image's [K,H] block: one offset register per lane, the region group in the scalar offset, regions
        // beyond K read as zeros (out of the resource's range).  The odd workgroups go straight on to part 4.
        const bool do_att = att_wg && img < RE;
        if (att_wg) {
            constexpr int JMAX = 9;                          // regions 4 j + rg, K <= 36
            const int vo = (rg_t * H + 4 * col4_t) * 4;
            f32x4 rv[JMAX];
#pragma unroll
            for (int j = 0; j < JMAX; ++j)
                rv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_att, vo, 4 * j * H * 4, 0));
            const f32x4 ah4 = reinterpret_cast<const f32x4*>(a.att_h_all + rowH + (size_t)imc * H)[col4_t];
            wait_for(cnt + t * 3 + 1, TJ);
            BPTT_STAMP(4);
            const f32x4 dr = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_dres, (int)(((size_t)imc * H + 4 * col4_t) * 4), so1, 16));
#pragma unroll
            for (int j = 0; j < JMAX; ++j) {
                float part = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) part += dr[e] * rv[j][e];
                part = sum8_dpp(part);
                part += dpp_f32<DPP_ROW_MIRROR>(part);
                const int k = 4 * j + rg_t;
                if (ac == 0 && k < K) sp[ks * 64 + k] = part;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < JMAX; ++j)
                rv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_patt, vo, 4 * j * H * 4, 0));
            __syncthreads();                                 // (workgroup-uniform branch)
            float dal = 0.f, al = 0.f;
            if (lane < K) {
#pragma unroll
                for (int q = 0; q < KS; ++q) dal += sp[q * 64 + lane];
                al = a.alpha_all[((size_t)t * B + imc) * K + lane];
            }
            const float cs = wave_sum_fast(al * dal);
            float dd = al * (dal - cs);                      // 0 for lanes >= K
            if (poison != 0.f) dd = poison;
            if (ks == 0 && lane < K && do_att) a.ddot_all[((size_t)t * B + img) * K + lane] = dd;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < JMAX; ++j) {
                const float dk = __shfl(dd, 4 * j + rg_t, 64);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float th = fast_tanh(rv[j][e] + ah4[e]);
                    acc[e] += dk * (1.0f - th * th);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[e];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                acc[e] = v * wa[e];
            }
            if (rg_t == 0 && do_att)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_b, acc), r_dah, (int)(((size_t)img * H + 4 * col4_t) * 4), so1, 16);
            BPTT_STAMP(5);
            publish(cnt + t * 3 + 2, true);
        }
        if (t == 0) break;                                   // h_{-1} is the constant zero state: nothing flows further
        // ---- 4. the (i, f, o) gate columns of dpre_t: the rest of dh = dpre h2h.W (nothing of it waits for the attention: it runs
        //         while the strip's attention results are on their way) ----------------------------------------------------------
        {
            f32x4 af[GI];
#pragma unroll
            for (int i = 0; i < GI; ++i)
                af[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    r_dpre, (int)(((size_t)mc_t * H5 + 192 * ks + (BF ? 32 * (i >> 1) + 8 * lq_t + 4 * (i & 1) : 16 * i + 4 * lq_t)) * 4), so5, 16));
            if (BF) {
#pragma unroll
                for (int j = 0; j < SI; ++j)
                    acc_dh = __builtin_amdgcn_mfma_f32_16x16x32_bf16(to_bf16x8_b(af[2 * j], af[2 * j + 1]), wb[BF ? 2 * SA + j : 0], acc_dh, 0, 0, 0);
            } else {
#pragma unroll
            for (int i = 0; i < GI; ++i)
#pragma unroll
                for (int s = 0; s < 4; ++s) acc_dh = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], wh_ifo[BF ? 0 : i][s], acc_dh, 0, 0, 0);
            }
        }
        BPTT_STAMP(6);
        wait_for(cnt + t * 3 + 2, TJ / 2);
        // ---- 5. dh += d_att_h h2att.W, cross-wave sum -> this lane's dh of step t - 1 --------------------------------------------
        {
            f32x4 ac4[GC];
#pragma unroll
            for (int i = 0; i < GC; ++i)
                ac4[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    r_dah, (int)(((size_t)mc_t * H + 64 * ks + (BF ? 32 * (i >> 1) + 8 * lq_t + 4 * (i & 1) : 16 * i + 4 * lq_t)) * 4), so1, 16));
            if (BF) {
#pragma unroll
                for (int j = 0; j < SC; ++j)
                    acc_dh = __builtin_amdgcn_mfma_f32_16x16x32_bf16(to_bf16x8_b(ac4[2 * j], ac4[2 * j + 1]), wb[BF ? 2 * SA + SI + j : 0], acc_dh, 0, 0, 0);
            } else {
#pragma unroll
            for (int i = 0; i < GC; ++i) {
                const f32x4 bw = wl[(ks * (GA + GC) + GA + i) * 64 + lane_t];
#pragma unroll
                for (int s = 0; s < 4; ++s) acc_dh = __builtin_amdgcn_mfma_f32_16x16x4f32(ac4[i][s], bw[s], acc_dh, 0, 0, 0);
            }
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) red[(ks * 4 + v) * 64 + lane_t] = acc_dh[v];
        __syncthreads();
        if (owner) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < KS; ++w) v += red[(w * 4 + ks) * 64 + lane_t];
            dh = poison != 0.f ? poison : v;
        }
        BPTT_STAMP(7);
    }
#undef BPTT_STAMP
}

// ---- feature gradients after the time loop ----------------------------------------------------
//   d_att[b,k,:]   = sum_t alpha_t[b,k] * d_att_res_t[b,:]
//   d_p_att[b,k,a] = w_a * sum_t ddot_t[b,k] * (1 - tanh^2(p_att[b,k,a] + att_h_t[b,a]))
//   d w_alpha[a]  += sum_{t,b,k} ddot_t[b,k] * tanh(...),   d b_alpha += sum ddot
// NSPLIT workgroups per image (each takes every NSPLIT-th group of NW regions); the per-step row vectors (att_h_t,
// d_att_res_t) and scalars of the image are staged once in LDS (dynamic) and re-used by all its regions.
// The kernel is bound by its transcendentals (T x K x A = 295 k tanh per image), so the exponential is taken OUT of the
// (region, step) loop: with x = e^{-2(p + h)} = e^{-2p} e^{-2h},
//     1 - tanh^2(p + h) = 4 / (x + 2 + 1/x),      tanh(p + h) = (1/x - x) / (x + 2 + 1/x),
// and e^{-2p}, e^{2p} are made once per (region, column), e^{-2h}, e^{2h} once per (step, column) while the rows are staged:
// the inner loop is two products, two sums and ONE reciprocal per element (before: an exponential and a reciprocal).  The
// exponents are clamped to +-40, so no factor overflows and no 0 x inf arises; where the clamp acts the true derivative
// and the computed one are both below 1e-15.
__device__ __forceinline__ float clamp40(float x) { return fminf(fmaxf(x, -40.0f), 40.0f); }
template <int NI, int NW, int NSPLIT, bool FACT>   // FACT: the factorised exponentials (needs 2 T A floats of LDS); else tanh per element
__global__ __launch_bounds__(NW * 64) void attn_bwd_feats_kernel(const float* __restrict__ p_att, const float* __restrict__ att_h_all,
                                                             const float* __restrict__ d_att_res_all,
                                                             const float* __restrict__ alpha_all,
                                                             const float* __restrict__ ddot_all,
                                                             const float* __restrict__ w_alpha, float* __restrict__ d_att,
                                                             float* __restrict__ d_p_att, float* __restrict__ dw_alpha,
                                                             float* __restrict__ db_alpha, int T, int B, int K, int A,
                                                             int H) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* s_eh = lds;                     // [T][A] e^{-2 att_h}  (FACT; else att_h itself)
    float* s_ehi = s_eh + (size_t)T * A;   // [T][A] e^{+2 att_h}  (FACT only)
    float* s_dr = s_ehi + (FACT ? (size_t)T * A : 0);   // [T][H]
    float* s_al = s_dr + (size_t)T * H;    // [T][K]
    float* s_dd = s_al + (size_t)T * K;    // [T][K]
    float* s_dw = s_dd + (size_t)T * K;    // [NW][NI*256] cross-wave reduce of dw_alpha
    const int b = blockIdx.x / NSPLIT, part = blockIdx.x % NSPLIT, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int A4 = A >> 2, H4 = H >> 2;
    for (int i = tid; i < T * A4; i += NW * 64) {
        const int t = i / A4, c = i % A4;
        const f32x4 ah = reinterpret_cast<const f32x4*>(att_h_all + ((size_t)t * B + b) * A)[c];
        if (!FACT) { reinterpret_cast<f32x4*>(s_eh)[i] = ah; continue; }
        f32x4 e, ei;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float z = clamp40(2.0f * ah[q]);
            e[q] = __expf(-z);
            ei[q] = __expf(z);
        }
        reinterpret_cast<f32x4*>(s_eh)[i] = e;
        reinterpret_cast<f32x4*>(s_ehi)[i] = ei;
    }
    for (int i = tid; i < T * H4; i += NW * 64) {
        const int t = i / H4, c = i % H4;
        reinterpret_cast<f32x4*>(s_dr)[i] = reinterpret_cast<const f32x4*>(d_att_res_all + ((size_t)t * B + b) * H)[c];
    }
    float bsum = 0.f;
    for (int i = tid; i < T * K; i += NW * 64) {
        const int t = i / K, k = i % K;
        s_al[i] = alpha_all[((size_t)t * B + b) * K + k];
        const float dd = ddot_all[((size_t)t * B + b) * K + k];
        s_dd[i] = dd;
        bsum += dd;
    }
    __syncthreads();
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 wa[NI], dw[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
        wa[i] = c < A4 ? reinterpret_cast<const f32x4*>(w_alpha)[c] : z4;
        dw[i] = z4;
    }
    for (int k = w + NW * part; k < K; k += NW * NSPLIT) {
        f32x4 ep[NI], epi[NI], dp[NI], da[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = lane + 64 * i;
            const f32x4 p = c < A4 ? reinterpret_cast<const f32x4*>(p_att + ((size_t)b * K + k) * A)[c] : z4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float z = clamp40(2.0f * p[q]);
                ep[i][q] = FACT ? __expf(-z) : p[q];
                epi[i][q] = FACT ? __expf(z) : 0.f;
            }
            dp[i] = z4;
            da[i] = z4;
        }
        for (int t = 0; t < T; ++t) {
            const float al = s_al[t * K + k], dd = s_dd[t * K + k];
            const float dd4 = 4.0f * dd;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int c = lane + 64 * i;
                if (c < A4) {
                    const f32x4 eh = reinterpret_cast<const f32x4*>(s_eh + (size_t)t * A)[c];
                    if (FACT) {
                        const f32x4 ehi = reinterpret_cast<const f32x4*>(s_ehi + (size_t)t * A)[c];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float x = ep[i][e] * eh[e], xi = epi[i][e] * ehi[e];
                            const float r = __builtin_amdgcn_rcpf((x + 2.0f) + xi);
                            dp[i][e] += dd4 * r;                 // dd * (1 - tanh^2)
                            dw[i][e] += dd * ((xi - x) * r);     // dd * tanh
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float th = fast_tanh(ep[i][e] + eh[e]);
                            dp[i][e] += dd * (1.0f - th * th);
                            dw[i][e] += dd * th;
                        }
                    }
                }
                if (c < H4) da[i] += al * reinterpret_cast<const f32x4*>(s_dr + (size_t)t * H)[c];
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = lane + 64 * i;
            if (c < A4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) dp[i][e] *= wa[i][e];
                reinterpret_cast<f32x4*>(d_p_att + ((size_t)b * K + k) * A)[c] = dp[i];
            }
            if (c < H4) reinterpret_cast<f32x4*>(d_att + ((size_t)b * K + k) * H)[c] = da[i];
        }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) *reinterpret_cast<f32x4*>(&s_dw[(w * NI * 64 + i * 64 + lane) * 4]) = dw[i];
    __syncthreads();
    for (int a = tid; a < A; a += NW * 64) {
        const int c = a >> 2, e = a & 3, i = c >> 6, l = c & 63;
        float s = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) s += s_dw[((ww * NI + i) * 64 + l) * 4 + e];
        atomicAdd(dw_alpha + a, s);
    }
    if (part == 0) {
        bsum = wave_sum(bsum);
        if (lane == 0) atomicAdd(db_alpha, bsum);
    }
}

// dE[it[t,b], :] += dx[t,b,:] * [E[it] > 0] * keep * scale        (AttModel.py:74-76 reversed)
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ E, const int32_t* __restrict__ it_all,
                                                        const uint8_t* __restrict__ keep, float scale,
                                                        const float* __restrict__ dx, float* __restrict__ dE, int TB,
                                                        int Ed, int plain) {
    // plain: bare embedding rows (FCModel): no ReLU gate
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)TB * Ed) return;
    const int r = (int)(i / Ed), j = (int)(i % Ed);
    const int tok = it_all[r];
    if (!plain && E[(size_t)tok * Ed + j] <= 0.f) return;
    float g = dx[i];
    if (keep) g *= (float)keep[i] * scale;
    if (g != 0.f) atomicAdd(dE + (size_t)tok * Ed + j, g);
}

// d_pre = d_att * keep * scale * [att_pre > 0]                     (AttModel.py:82-85 reversed)
__global__ __launch_bounds__(256) void relu_keep_bwd_kernel(const float* __restrict__ d_att, const float* __restrict__ att_pre,
                                                            const uint8_t* __restrict__ keep, float scale,
                                                            float* __restrict__ d_pre, int64_t n4) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n4) return;
    f32x4 g = reinterpret_cast<const f32x4*>(d_att)[idx];
    const f32x4 a = reinterpret_cast<const f32x4*>(att_pre)[idx];
    uint32_t kp = 0x01010101u;
    if (keep) kp = *reinterpret_cast<const uint32_t*>(keep + idx * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float kf = keep ? (float)((kp >> (8 * e)) & 0xffu) * scale : 1.0f;
        g[e] = a[e] > 0.f ? g[e] * kf : 0.f;
    }
    reinterpret_cast<f32x4*>(d_pre)[idx] = g;
}

struct SpkBws {
    float *dlogits, *d_out_all, *dpre_all, *d_att_h_all, *d_att_res_all, *ddot_all, *dh_a, *dh_b, *dc, *dx_all;
    float *d_att, *d_p_att, *d_attpre;
    float* dpre_img;   // [B,5H] image step of an FCModel decode
    unsigned* sync;    // spk_bptt_seq_kernel: [strips of 16 rows][T][3] hand-off counters + error word (cleared by the launcher)
    size_t nsync;
    size_t bytes;
};
SpkBws spk_bcarve(const cic_speaker_dims& d, void* base, bool own_dlogits) {
    SpkBws w;
    Carver c(base);
    const size_t B = d.B, K = d.K, H = d.H, E = d.E, A = d.A, T = d.T, V1 = d.V + 1;
    w.dlogits = own_dlogits ? c.f32(T * B * V1) : nullptr;
    w.d_out_all = c.f32(T * B * H);
    w.dpre_all = c.f32(T * B * 5 * H);
    w.d_att_h_all = c.f32(T * B * A);
    w.d_att_res_all = c.f32(T * B * H);
    w.ddot_all = c.f32(T * B * K);
    w.dh_a = c.f32(B * H);
    w.dh_b = c.f32(B * H);
    w.dc = c.f32(B * H);
    w.dx_all = c.f32(T * B * E);
    w.d_att = c.f32(B * K * H);
    w.d_p_att = c.f32(B * K * A);
    w.d_attpre = c.f32(B * K * H);
    w.dpre_img = c.f32(B * 5 * H);
    w.nsync = (((B + 15) / 16) * T * 3 + 1 + 3) / 4 * 4;
    w.sync = reinterpret_cast<unsigned*>(c.i32(w.nsync));
    w.bytes = c.used();
    return w;
}

}  // namespace

#ifdef CIC_DEVTOOLS
extern "C" int cic_debug_bptt_early_stop(int on) { g_bptt_early_stop = on; return 0; }
extern "C" int cic_debug_bptt_seq(int on) { g_bptt_seq = on; return 0; }
extern "C" int cic_debug_set_bptt_stamps(unsigned long long* buf) {
    CIC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_bptt_stamps), &buf, sizeof(buf)));
    return 0;
}
#endif

extern "C" size_t cic_speaker_decode_bwd_ws_bytes(const cic_speaker_dims* d) {
    if (!d) return 0;
    return spk_bcarve(*d, nullptr, true).bytes;
}

static int decode_bwd_impl(const cic_speaker_dims* dp, const cic_speaker_params* p, const cic_decode_io* io,
                           const cic_decode_bwd_io* bio, void* ws_fwd, size_t ws_fwd_bytes, void* ws_bwd,
                           size_t ws_bwd_bytes, cic_stream_t s);

extern "C" int cic_speaker_decode_bwd(const cic_speaker_dims* dp, const cic_speaker_params* p, const cic_decode_io* io,
                                      const cic_decode_bwd_io* bio, void* ws_fwd, size_t ws_fwd_bytes, void* ws_bwd,
                                      size_t ws_bwd_bytes, cic_stream_t s) {
    CIC_REQUIRE(dp && p && io && bio && ws_fwd && ws_bwd && bio->grads && (bio->att_raw || io->fc_mode));
    return decode_bwd_impl(dp, p, io, bio, ws_fwd, ws_fwd_bytes, ws_bwd, ws_bwd_bytes, s);
}

static int decode_bwd_impl(const cic_speaker_dims* dp, const cic_speaker_params* p, const cic_decode_io* io,
                           const cic_decode_bwd_io* bio, void* ws_fwd, size_t ws_fwd_bytes, void* ws_bwd,
                           size_t ws_bwd_bytes, cic_stream_t s) {
    CIC_REQUIRE(dp && p && io && bio && ws_fwd && ws_bwd && bio->grads && (bio->att_raw || io->fc_mode));
    const cic_speaker_dims& d = *dp;
    const bool fc = io->fc_mode != 0;
    CIC_REQUIRE(!fc || (d.K == 0 && io->x0 && bio->d_x0));
    SpkWs w = spk_carve(d, ws_fwd);
    CIC_REQUIRE(ws_fwd_bytes >= w.bytes);
    SpkBws g = spk_bcarve(d, ws_bwd, true);
    CIC_REQUIRE(ws_bwd_bytes >= g.bytes);
    const bool ps = io->mode == CIC_SAMPLE_GUMBEL_PS || io->mode == CIC_SAMPLE_MULTINOMIAL_PS;
    CIC_REQUIRE(!(bio->d_onehot && io->mode == CIC_SAMPLE_GUMBEL_ST) || io->U || io->u_philox);
    CIC_REQUIRE(!bio->d_onehot || io->seq);
    CIC_REQUIRE(!ps || (io->soft_raw && io->xpre && io->seq && (io->mode != CIC_SAMPLE_GUMBEL_PS || io->U)));
    const int phase = bio->phase;
    CIC_REQUIRE(phase == CIC_BWD_ALL || ((phase == CIC_BWD_LOGIT || phase == CIC_BWD_REST) && !ps) ||
                ((phase == CIC_BWD_LOOP || phase == CIC_BWD_TAIL) && !ps && !fc));
    CIC_REQUIRE(!bio->dslp_scale || !ps);     // the in-loop sampler backward of partial sampling takes dslp as it is
    const bool do_logit = phase == CIC_BWD_ALL || phase == CIC_BWD_LOGIT;
    const bool do_loop = phase == CIC_BWD_ALL || phase == CIC_BWD_REST || phase == CIC_BWD_LOOP;
    const bool do_tail = phase == CIC_BWD_ALL || phase == CIC_BWD_REST || phase == CIC_BWD_TAIL;
    GemmCtx st(cic_s(s), d.compute_dtype == CIC_DTYPE_BF16 ? CIC_PRECISION_BF16 : CIC_PRECISION_F32);
    const int B = d.B, K = d.K, H = d.H, E = d.E, A = d.A, T = d.T, V1 = d.V + 1, D = d.D;
    const float scale = 1.0f / (1.0f - d.p_drop);
    const cic_speaker_params* gr = bio->grads;
    int rc;
#define RUN(x) if ((rc = (x)) != 0) return rc

    // 1. d logits for every step at once (rows are independent of the recurrence; not so for partial sampling,
    //    whose soft row is the next step's input: there steps 1-2 run inside the time loop)
    if (!ps && do_logit) {
        dim3 grid(T * B), blk(1024);
        const int64_t* tgt = io->mode == CIC_SAMPLE_TEACHER ? io->pick : nullptr;
#define GO(RV) hipLaunchKernelGGL((sampler_bwd_kernel<RV>), grid, blk, 0, st, w.logp_all, io->U, bio->d_onehot, w.it_all, \
                                  tgt, io->seq, bio->dslp, io->L, io->mode, io->temp, g.dlogits, T, B, V1, w.lse_all,        \
                                  io->u_philox, io->u_seed, io->u_offset * 4ull, io->decoding_constraint, bio->dslp_scale,   \
                                  g.sync, (int)g.nsync, g.d_out_all, H)
        if (V1 <= 4096) GO(1); else if (V1 <= 12288) GO(3); else if (V1 <= 32768) GO(8);
        else { cic_set_error("vocabulary too large"); return 1; }
#undef GO
        CIC_LAUNCH_CHECK();
    }
    // 2. logit layer, batched over time: d_out = dlogits W,  dW += dlogits^T out,  db += colsum
    if (!ps && do_logit) RUN(gemm_nn(g.dlogits, V1, p->logit_w, H, g.d_out_all, H, T * B, H, V1, false, st, true, true));   // (cleared by the sampler backward)

    // 2b. the logit layer's weight gradient needs only d logits and the saved outputs (a side stream for it beside the
    //     latency-bound BPTT loop measured slower than one stream - its workgroups hold the CUs the loop's short kernels
    //     need; not for partial sampling, whose d logits are made inside the loop)
    if (!ps && do_logit) RUN(gemm_tn(g.dlogits, V1, w.out_all, H, gr->logit_w, H, V1, H, T * B, true, st, gr->logit_b));
    if (!do_loop && !do_tail) return 0;        // CIC_BWD_LOGIT: the logit layer's gradient is final; d out waits in ws_bwd
    // 3. BPTT over the cell + attention (only dh, dc are carried)
    float* dh_in = g.dh_a;
    float* dh_out = g.dh_b;
    // the flagship widths walk the loop in ONE launch (spk_bptt_seq_kernel): every workgroup resident at once, one per CU
    bool seq_kernel = false;
    int seq_rows = 0;
    if (g_bptt_seq && !ps && !fc && H == 512 && A == 512 && K >= 1 && K <= 36 && !bio->device_shared && !io->device_shared) {
        static DeviceOnce attr_set;
        if (attr_set.first())
        {
            CIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&spk_bptt_seq_kernel<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)BPTT_LDS_BYTES));
            CIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&spk_bptt_seq_kernel<8, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)BPTT_LDS_BYTES));
        }
        // every workgroup of a launch resident at once: one per CU, where the occupancy query admits one
        const bool bfk = d.compute_dtype == CIC_DTYPE_BF16;
        const int cus = bfk ? cic_resident_cus(reinterpret_cast<const void*>(&spk_bptt_seq_kernel<8, true>), 512, BPTT_LDS_BYTES)
                            : cic_resident_cus(reinterpret_cast<const void*>(&spk_bptt_seq_kernel<8, false>), 512, BPTT_LDS_BYTES);
        seq_rows = (cus / (H / 16)) * 16;                     // rows one launch can walk with every workgroup resident
        seq_kernel = seq_rows >= 16;
    }
    if (seq_kernel && do_loop) {
        // The hand-off counters and the error word: cleared by sampler_bwd_kernel when the logit phase ran in this call; a call
        // that enters at the loop (CIC_BWD_LOOP / CIC_BWD_REST, or a repeated one on the same ws_bwd) clears them itself - with
        // counters left at their targets every wait would pass at once and the workgroups would race
        if (!do_logit) CIC_HIP(hipMemsetAsync(g.sync, 0, sizeof(unsigned) * g.nsync, st));
        BpttArgs ba = {};
        ba.pre_all = w.pre_all; ba.c_all = w.c_all; ba.alpha_all = w.alpha_all; ba.att_h_all = w.att_h_all;
        ba.p_att = w.p_att; ba.att = w.att; ba.out_keep = io->out_keep;
        ba.a2c_w = p->a2c_w; ba.h2h_w = p->h2h_w; ba.h2att_w = p->h2att_w; ba.alpha_w = p->alpha_w;
        ba.d_out_all = g.d_out_all; ba.dpre_all = g.dpre_all; ba.d_att_res_all = g.d_att_res_all;
        ba.d_att_h_all = g.d_att_h_all; ba.ddot_all = g.ddot_all;
        ba.cnt = g.sync;
        ba.hg = handoff_guard(g.sync + (size_t)cic_cdiv(B, 16) * T * 3, io->status, CIC_STATUS_BPTT);
        ba.L = g_bptt_early_stop ? io->L : nullptr;
        ba.zero_tbh = (E == H && !ps) ? g.dx_all : nullptr;       // the K-sliced d x product after the loop adds into it
        ba.scale = scale; ba.B = B; ba.K = K; ba.T = T;
        for (int row0 = 0; row0 < B; row0 += seq_rows) {          // (B = 128: one launch; B = 256: two row blocks)
            ba.row0 = row0;
            ba.row_end = row0 + seq_rows < B ? row0 + seq_rows : B;
            if (d.compute_dtype == CIC_DTYPE_BF16)
                hipLaunchKernelGGL((spk_bptt_seq_kernel<8, true>), dim3(cic_cdiv(ba.row_end - row0, 16) * (H / 16)), dim3(512), BPTT_LDS_BYTES, st, ba);
            else
                hipLaunchKernelGGL((spk_bptt_seq_kernel<8, false>), dim3(cic_cdiv(ba.row_end - row0, 16) * (H / 16)), dim3(512), BPTT_LDS_BYTES, st, ba);
            CIC_LAUNCH_CHECK();
        }
    }
    for (int t = T - 1; t >= 0 && !seq_kernel && do_loop; --t) {
        const uint8_t* ok = io->out_keep ? io->out_keep + (size_t)(t + (fc ? 1 : 0)) * B * H : nullptr;
        float* dpre = g.dpre_all + (size_t)t * B * 5 * H;
        if (ps) {
            float* dl = g.dlogits + (size_t)t * B * V1;
            const int rec = t + 1 < T ? 1 : 0;
            // d soft_raw[t] from the next step's input: d xpre[t+1] embed[0:V+1]^T         (AttModel.py:395-397)
            if (rec) RUN(gemm_nt(g.dx_all + (size_t)(t + 1) * B * E, E, p->embed_w, E, dl, V1, B, V1, E, nullptr, false, false, st));
            dim3 grid(B), blk(1024);
#define GO(RV) hipLaunchKernelGGL((sampler_ps_bwd_kernel<RV>), grid, blk, 0, st, w.logp_all + (size_t)t * B * V1,          \
                                  io->U ? io->U + (size_t)(t + 1) * B * V1 : nullptr,                                      \
                                  bio->d_onehot ? bio->d_onehot + (size_t)t * B * V1 : nullptr, dl, rec,                   \
                                  w.it_all + (size_t)(t + 1) * B, io->seq, bio->dslp, io->L, t, io->mode, io->temp, T, V1)
            if (V1 <= 4096) GO(1); else if (V1 <= 12288) GO(3); else if (V1 <= 32768) GO(8);
            else { cic_set_error("vocabulary too large"); return 1; }
#undef GO
            CIC_LAUNCH_CHECK();
            RUN(gemm_nn(dl, V1, p->logit_w, H, g.d_out_all + (size_t)t * B * H, H, B, H, V1, false, st));
        }
        hipLaunchKernelGGL(cell_bwd_kernel, dim3(cic_cdiv(B * (H / 4), 256)), dim3(256), 0, st,
                           w.pre_all + (size_t)t * B * 5 * H, w.c_all + (size_t)t * B * H,
                           w.c_all + (size_t)(t + 1) * B * H, g.d_out_all + (size_t)t * B * H, dh_in, g.dc, ok, scale,
                           dpre, B, H, t == T - 1 ? 1 : 0, fc ? nullptr : g.d_att_res_all + (size_t)t * B * H,
                           (t > 0 || fc) ? dh_out : nullptr, fc ? 1 : 0);
        CIC_LAUNCH_CHECK();
        // d att_res = d in_transform a2c.W            [B,2H] x [2H,H]   (into the slab the cell kernel cleared)
        float* dres = g.d_att_res_all + (size_t)t * B * H;
        // steps at or beyond the decode's length L carry no gradient (d out = 0, no carry): their two products would add
        // zeros into the buffers the cell kernel has just cleared, so they return at once (device-side L, no host sync)
        GemmCtx stl = st;
        if (!ps && !fc && io->L && g_bptt_early_stop) { stl.live = io->L; stl.live_min = t + 1; }
        if (!fc) RUN(gemm_nn(dpre + 3 * H, 5 * H, p->a2c_w, H, dres, H, B, H, 2 * H, false, stl, true, true));
        if (!fc) {
            dim3 grid(B), blk(1024);
            const int mx = A > H ? A : H;
            float* dah = g.d_att_h_all + (size_t)t * B * A;
            float* ddot = g.ddot_all + (size_t)t * B * K;
            const float* al = w.alpha_all + (size_t)t * B * K;
            const float* ah = w.att_h_all + (size_t)t * B * A;
#define GO(NI, KPW, HOLD) hipLaunchKernelGGL((attn_bwd_kernel<NI, KPW, 16, HOLD>), grid, blk, 0, st, dres, al, ah, \
                                             w.p_att, w.att, p->alpha_w, dah, ddot, K, A, H)
            void* ph = cic_timer_begin(io->timer, CIC_TIMED_ATTN_BWD, st);
            if (A == H && (H & 31) == 0 && H <= 512 && K <= 64) {
                dim3 blkc((H / 32) * 64);
                const bool twin = false;   // measured: 7.5 us vs 6.9 us single-WG at B = 128 (latency-, not bandwidth-bound)
#define GOC(J)                                                                                                          \
    do {                                                                                                                \
        if (twin) hipLaunchKernelGGL((attn_bwd_cols_kernel<J, true>), dim3(2 * B), blkc, 0, st, dres, al, ah, w.p_att, w.att, \
                                     p->alpha_w, dah, ddot, K, H);                                                      \
        else hipLaunchKernelGGL((attn_bwd_cols_kernel<J, false>), grid, blkc, 0, st, dres, al, ah, w.p_att, w.att,        \
                                p->alpha_w, dah, ddot, K, H);                                                           \
    } while (0)
                if (K <= 8) GOC(1); else if (K <= 16) GOC(2); else if (K <= 24) GOC(3); else if (K <= 32) GOC(4);
                else if (K <= 40) GOC(5); else if (K <= 48) GOC(6); else GOC(8);
#undef GOC
            } else
            if (mx <= 256) { if (K <= 48) GO(1, 3, true); else GO(1, 4, true); }
            else if (mx <= 512) { if (K <= 48) GO(2, 3, true); else GO(2, 4, true); }
            else GO(4, 4, false);
#undef GO
            cic_timer_end(ph, st);
            CIC_LAUNCH_CHECK();
        }
        if (fc) {
            // dh_t = dpre h2h.W  (no attention); also at t = 0: the state came from the image step
            RUN(gemm_nn(dpre, 5 * H, p->h2h_w, H, dh_out, H, B, H, 5 * H, false, st, true, true));
            float* tmp = dh_in; dh_in = dh_out; dh_out = tmp;
        } else if (t > 0) {
            // dh_t = dpre h2h.W + d_att_h h2att.W       [B,5H]x[5H,H] + [B,A]x[A,H]
            RUN(gemm_nn2(dpre, 5 * H, p->h2h_w, H, 5 * H, g.d_att_h_all + (size_t)t * B * A, A, p->h2att_w, H, A,
                         dh_out, H, B, H, false, stl, true));
            float* tmp = dh_in; dh_in = dh_out; dh_out = tmp;
        }
        if (ps) {
            // d xt = dpre i2h.W, then back through relu_dropout to d xpre (in place)          (AttModel.py:396-397)
            float* dx = g.dx_all + (size_t)t * B * E;
            RUN(gemm_nn(dpre, 5 * H, p->i2h_w, E, dx, E, B, E, 5 * H, false, st));
            if (t >= 1) {
                const int64_t n4 = (int64_t)B * E / 4;
                hipLaunchKernelGGL(relu_keep_bwd_kernel, dim3(cic_cdiv(n4, 256)), dim3(256), 0, st, dx,
                                   io->xpre + (size_t)t * B * E, io->x_keep ? io->x_keep + (size_t)t * B * E : nullptr,
                                   scale, dx, n4);
                CIC_LAUNCH_CHECK();
            }
        }
    }
    if (!do_tail) return 0;        // CIC_BWD_LOOP: dpre / d att_res / d att_h / ddot of every step wait in ws_bwd
    if (ps) {
        RUN(gemm_tn(g.dlogits, V1, w.out_all, H, gr->logit_w, H, V1, H, T * B, true, st, gr->logit_b));
    }
    // 4. weight gradients of the recurrent part, batched over time
    // (i2h.bias and h2h.bias receive the same column sums of dpre: one by-product, two targets)
    RUN(gemm_tn(g.dpre_all, 5 * H, w.x_all, E, gr->i2h_w, E, 5 * H, E, T * B, true, st, gr->i2h_b, gr->h2h_b));
    RUN(gemm_tn(g.dpre_all, 5 * H, w.h_all, H, gr->h2h_w, H, 5 * H, H, T * B, true, st));
    if (!fc) {
        RUN(gemm_tn(g.dpre_all + 3 * H, 5 * H, w.att_res_all, H, gr->a2c_w, H, 2 * H, H, T * B, true, st, gr->a2c_b));
        RUN(gemm_tn(g.d_att_h_all, A, w.h_all, H, gr->h2att_w, H, A, H, T * B, true, st, gr->h2att_b));
    } else {
        // image step (FCModel.py:97-99,121): through the cell with a zero previous state, then d x0 = dpre i2h.W
        const uint8_t* ok_img = io->out_keep;
        hipLaunchKernelGGL(cell_bwd_kernel, dim3(cic_cdiv(B * (H / 4), 256)), dim3(256), 0, st, w.pre_img, w.zeros, w.c_all,
                           w.zeros, dh_in, g.dc, ok_img, scale, g.dpre_img, B, H, 0, nullptr, nullptr, 1);
        CIC_LAUNCH_CHECK();
        RUN(gemm_nn(g.dpre_img, 5 * H, p->i2h_w, E, bio->d_x0, E, B, E, 5 * H, false, st));
        RUN(gemm_tn(g.dpre_img, 5 * H, io->x0, E, gr->i2h_w, E, 5 * H, E, B, true, st));
        RUN(cic_colsum_f32(g.dpre_img, B, 5 * H, 5 * H, gr->i2h_b, 1, s));
        RUN(cic_colsum_f32(g.dpre_img, B, 5 * H, 5 * H, gr->h2h_b, 1, s));
    }
    // token embedding: dx = dpre i2h.W, scattered into the embedding rows
    // (grads->embed_w == NULL: the table is frozen - share_embed = 1 in phase 2, AlternatingJointModel.py:86-88 - and takes no gradient)
    if (!ps && gr->embed_w) RUN(gemm_nn(g.dpre_all, 5 * H, p->i2h_w, E, g.dx_all, E, T * B, E, 5 * H, false, st, true, seq_kernel && E == H));
    if (gr->embed_w) {
        // partial sampling: only step 0 reads an embedding row (<bos>); steps >= 1 used soft_raw[t-1] @ embed
        const int rows = ps ? B : T * B;
        const int64_t n = (int64_t)rows * E;
        hipLaunchKernelGGL(embed_bwd_kernel, dim3(cic_cdiv(n, 256)), dim3(256), 0, st, p->embed_w, w.it_all,
                           fc ? nullptr : io->x_keep, scale, g.dx_all, gr->embed_w, rows, E, fc ? 1 : 0);
        CIC_LAUNCH_CHECK();
        if (ps && T > 1)
            RUN(gemm_tn(io->soft_raw, V1, g.dx_all + (size_t)B * E, E, gr->embed_w, E, V1, E, (T - 1) * B, true, st));
    }
    if (fc) return 0;      // no region features
    // 5. attention features: d att, d p_att, d alpha_net in one pass over p_att
    {
        const int mx = A > H ? A : H;
        const int NIv = mx <= 256 ? 1 : (mx <= 512 ? 2 : 4);
        // two workgroups of 6 waves per image (3 regions per wave at K = 36): 256 workgroups at B = 128, one per CU
#define GO(NI, NWF, NSP, FACT)                                                                                       \
    do {                                                                                                             \
        const size_t shm = sizeof(float) * ((size_t)T * ((FACT ? 2 : 1) * A + H + 2 * K) + (size_t)NWF * NI * 256);  \
        CIC_REQUIRE(shm <= 160 * 1024);                                                                              \
        CIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_feats_kernel<NI, NWF, NSP, FACT>),       \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));                          \
        hipLaunchKernelGGL((attn_bwd_feats_kernel<NI, NWF, NSP, FACT>), dim3(B * NSP), dim3(NWF * 64), shm, st, w.p_att, w.att_h_all, \
                           g.d_att_res_all, w.alpha_all, g.ddot_all, p->alpha_w, g.d_att, g.d_p_att, gr->alpha_w,    \
                           gr->alpha_b, T, B, K, A, H);                                                              \
    } while (0)
        const bool fact = sizeof(float) * ((size_t)T * (2 * A + H + 2 * K) + (size_t)6 * 2 * 256) <= 160 * 1024;
        if (NIv == 1) { if (fact) GO(1, 6, 2, true); else GO(1, 6, 2, false); }
        else if (NIv == 2) { if (fact) GO(2, 6, 2, true); else GO(2, 6, 2, false); }
        else GO(4, 4, 1, false);
#undef GO
        CIC_LAUNCH_CHECK();
    }
    // ctx2att: d att += d p_att W,  dW += d p_att^T att,  db += colsum
    RUN(gemm_nn(g.d_p_att, A, p->ctx2att_w, H, g.d_att, H, B * K, H, A, true, st));
    RUN(gemm_tn(g.d_p_att, A, w.att, H, gr->ctx2att_w, H, A, H, B * K, true, st, gr->ctx2att_b));
    // att_embed: through dropout and ReLU, then dW += d_pre^T att_raw
    {
        const int64_t n4 = (int64_t)B * K * H / 4;
        hipLaunchKernelGGL(relu_keep_bwd_kernel, dim3(cic_cdiv(n4, 256)), dim3(256), 0, st, g.d_att, io->att_pre,
                           io->att_keep, scale, g.d_attpre, n4);
        CIC_LAUNCH_CHECK();
    }
    RUN(gemm_tn(g.d_attpre, H, bio->att_raw, D, gr->att_embed_w, D, H, D, B * K, true, st, gr->att_embed_b));
#undef RUN
    return 0;
}
