// Speaker sequence engines: the host loops of AttModel.sample / AttModel.forward
// (models/AttModel.py:103-148,291-452) as back-to-back kernel launches on ONE HIP stream with
// no host synchronisation.  The reference's per-step `unfinished.sum() == 0` sync + break is
// replaced by device-side flags and a device-side length L; every activation the backward pass
// needs is written once into a caller-provided workspace.
#include "cic_common.h"
#include "engine_util.h"

namespace {

__global__ void fill_i32_kernel(int32_t* p, int n, int32_t v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void i64_to_i32_kernel(const int64_t* a, int32_t* o, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = (int32_t)a[i];
}
__global__ void add_vec_kernel(const float* a, const float* b, float* o, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[i] + b[i];
}

}  // namespace

int cic_fill_i32(int32_t* p, int n, int32_t v, hipStream_t st) {
    hipLaunchKernelGGL(fill_i32_kernel, dim3(cic_cdiv(n, 256)), dim3(256), 0, st, p, n, v);
    CIC_LAUNCH_CHECK();
    return 0;
}
int cic_add_vec(const float* a, const float* b, float* o, int n, hipStream_t st) {
    hipLaunchKernelGGL(add_vec_kernel, dim3(cic_cdiv(n, 256)), dim3(256), 0, st, a, b, o, n);
    CIC_LAUNCH_CHECK();
    return 0;
}

SpkWs spk_carve(const cic_speaker_dims& d, void* base) {
    SpkWs w;
    Carver c(base);
    const size_t B = d.B, K = d.K, H = d.H, E = d.E, A = d.A, T = d.T, V1 = d.V + 1;
    w.att = c.f32(B * K * H);
    w.p_att = c.f32(B * K * A);
    w.x_all = c.f32(T * B * E);
    w.h_all = c.f32((T + 1) * B * H);
    w.c_all = c.f32((T + 1) * B * H);
    w.att_h_all = c.f32(T * B * A);
    w.att_res_all = c.f32(T * B * H);
    w.alpha_all = c.f32(T * B * K);
    w.dot_all = c.f32(T * B * K);
    w.pre_all = c.f32(T * B * 5 * H);
    w.out_all = c.f32(T * B * H);
    w.logp_all = c.f32(T * B * V1);
    w.bias_ih = c.f32(5 * H);
    w.it_all = c.i32((T + 1) * B);
    w.unfinished = c.i32(B);
    w.any_unf = c.i32(T + 1);
    w.bytes = c.used();
    return w;
}

extern "C" size_t cic_speaker_decode_ws_bytes(const cic_speaker_dims* d) {
    if (!d) return 0;
    return spk_carve(*d, nullptr).bytes;
}

static int check_dims(const cic_speaker_dims& d) {
    CIC_REQUIRE(d.B > 0 && d.K > 0 && d.K <= 64 && d.T > 0 && d.T <= 64);
    CIC_REQUIRE((d.H & 3) == 0 && (d.E & 3) == 0 && (d.A & 3) == 0 && d.D > 0 && d.V > 0);
    CIC_REQUIRE(d.p_drop >= 0.f && d.p_drop < 1.f);
    return 0;
}

extern "C" int cic_speaker_att_embed_fwd(const cic_speaker_dims* dp, const cic_speaker_params* p,
                                         const float* att_raw, float* att_pre, cic_stream_t s) {
    CIC_REQUIRE(dp && p && att_raw && att_pre);
    const cic_speaker_dims& d = *dp;
    if (int rc = check_dims(d)) return rc;
    // relu(att_raw W^T + b): [B*K, D] x [H, D]^T   (models/AttModel.py:82-85 without the dropout)
    return gemm_nt(att_raw, d.D, p->att_embed_w, d.D, att_pre, d.H, d.B * d.K, d.H, d.D, p->att_embed_b, false,
                   true, cic_s(s));
}

static int decode_fwd_impl(const cic_speaker_dims* dp, const cic_speaker_params* p, const cic_decode_io* io, void* ws,
                           size_t ws_bytes, cic_stream_t s);

extern "C" int cic_speaker_decode_fwd(const cic_speaker_dims* dp, const cic_speaker_params* p,
                                      const cic_decode_io* io, void* ws, size_t ws_bytes, cic_stream_t s) {
    CIC_REQUIRE(dp && p && io && ws);
    uint64_t key = cic_hash_bytes("decode_fwd", 10, 1469598103934665603ull);
    key = cic_hash_bytes(dp, sizeof(*dp), key);
    key = cic_hash_bytes(p, sizeof(*p), key);
    key = cic_hash_bytes(io, sizeof(*io), key);
    key = cic_hash_bytes(&ws, sizeof(ws), key);
    CicGraphScope gs(cic_s(s), key);
    if (gs.replayed) return 0;
    return gs.finish(decode_fwd_impl(dp, p, io, ws, ws_bytes, s));
}

static int decode_fwd_impl(const cic_speaker_dims* dp, const cic_speaker_params* p, const cic_decode_io* io, void* ws,
                           size_t ws_bytes, cic_stream_t s) {
    CIC_REQUIRE(dp && p && io && ws);
    const cic_speaker_dims& d = *dp;
    if (int rc = check_dims(d)) return rc;
    CIC_REQUIRE(io->mode >= CIC_SAMPLE_GREEDY && io->mode <= CIC_SAMPLE_MULTINOMIAL_PS);
    const bool ps = io->mode == CIC_SAMPLE_GUMBEL_PS || io->mode == CIC_SAMPLE_MULTINOMIAL_PS;
    CIC_REQUIRE(!ps || (io->soft_raw && io->xpre && io->soft_out));
    CIC_REQUIRE(io->att_pre && io->seq && io->slp && io->L);
    SpkWs w = spk_carve(d, ws);
    CIC_REQUIRE(ws_bytes >= w.bytes);
    hipStream_t st = cic_s(s);
    const int B = d.B, K = d.K, H = d.H, E = d.E, A = d.A, T = d.T, V1 = d.V + 1;
    const float p_drop = d.p_drop;
    int rc;
#define RUN(x) if ((rc = (x)) != 0) return rc

    // att = dropout(relu(att_embed(att_raw)));  p_att = ctx2att(att)      (AttModel.py:315,319)
    RUN(cic_apply_keep(io->att_pre, io->att_keep, io->att_keep ? p_drop : 0.f, w.att, (int64_t)B * K * H, s));
    RUN(gemm_nt(w.att, H, p->ctx2att_w, H, w.p_att, A, B * K, A, H, p->ctx2att_b, false, false, st));
    RUN(cic_add_vec(p->i2h_b, p->h2h_b, w.bias_ih, 5 * H, st));
    CIC_HIP(hipMemsetAsync(w.h_all, 0, sizeof(float) * B * H, st));          // init_hidden (:311)
    CIC_HIP(hipMemsetAsync(w.c_all, 0, sizeof(float) * B * H, st));
    CIC_HIP(hipMemsetAsync(w.any_unf, 0, sizeof(int32_t) * (T + 1), st));
    RUN(cic_fill_i32(w.unfinished, B, 1, st));
    if (io->first_token) {                                                    // AttModel.forward: seq[:, 0]  (:131)
        hipLaunchKernelGGL(i64_to_i32_kernel, dim3(cic_cdiv(B, 256)), dim3(256), 0, st, io->first_token, w.it_all, B);
        CIC_LAUNCH_CHECK();
    } else {
        RUN(cic_fill_i32(w.it_all, B, d.V + 1, st));                          // <bos> = vocab_size + 1 (:324-326)
    }

    for (int t = 0; t < T; ++t) {
        float* x = w.x_all + (size_t)t * B * E;
        float* h = w.h_all + (size_t)t * B * H;
        float* c = w.c_all + (size_t)t * B * H;
        float* att_h = w.att_h_all + (size_t)t * B * A;
        float* att_res = w.att_res_all + (size_t)t * B * H;
        float* pre = w.pre_all + (size_t)t * B * 5 * H;
        float* out = w.out_all + (size_t)t * B * H;
        float* logp = w.logp_all + (size_t)t * B * V1;
        const uint8_t* xk = io->x_keep ? io->x_keep + (size_t)t * B * E : nullptr;
        const uint8_t* ok = io->out_keep ? io->out_keep + (size_t)t * B * H : nullptr;
        if (ps && t >= 1) {
            // xt = relu_dropout(soft_vec @ embed.weight)                   (:395-397), soft_vec un-masked
            float* xp = io->xpre + (size_t)t * B * E;
            RUN(gemm_nn(io->soft_raw + (size_t)(t - 1) * B * V1, V1, p->embed_w, E, xp, E, B, E, V1, false, st));
            RUN(cic_relu_keep_fwd(xp, xk, xk ? p_drop : 0.f, x, (int64_t)B * E, st));
        } else {
            // xt = embed(it)                                               (:399)
            RUN(cic_embed_fwd(p->embed_w, w.it_all + (size_t)t * B, xk, xk ? p_drop : 0.f, x, B, E, s));
        }
        // attention                                                        (:465-489)
        RUN(gemm_nt(h, H, p->h2att_w, H, att_h, A, B, A, H, p->h2att_b, false, false, st));
        CIC_PROF(CIC_PROF_ATTN_FWD, st,
                 rc = cic_attn_fwd(att_h, w.p_att, w.att, p->alpha_w, p->alpha_b, io->att_masks, att_res,
                                   w.alpha_all + (size_t)t * B * K, w.dot_all + (size_t)t * B * K, B, K, A, H, s));
        if (rc) return rc;
        // all_input_sums = i2h(xt) + h2h(h);  in_transform += a2c(att_res)   (:514,521-522)
        RUN(gemm_nt2(x, E, p->i2h_w, E, E, h, H, p->h2h_w, H, H, pre, 5 * H, B, 5 * H, w.bias_ih, st));
        RUN(gemm_nt(att_res, H, p->a2c_w, H, pre + 3 * H, 5 * H, B, 2 * H, H, p->a2c_b, true, false, st));
        RUN(cic_cell_fwd(pre, c, ok, ok ? p_drop : 0.f, h + (size_t)B * H, c + (size_t)B * H, out, B, H, s));
        // logprobs = log_softmax(logit(output)); choose the input of step t+1   (:328-365,444)
        CIC_PROF(CIC_PROF_LOGIT_GEMM, st,
                 rc = gemm_nt(out, H, p->logit_w, H, logp, V1, B, V1, H, p->logit_b, false, false, st));
        if (rc) return rc;
        cic_sampler_args a;
        a.logits = logp; a.B = B; a.V1 = V1; a.ld = V1;
        a.mode = io->mode; a.temp = io->temp;
        a.U = io->U ? io->U + (size_t)(t + 1) * B * V1 : nullptr; a.ldu = V1;
        a.pick = io->pick ? io->pick + (size_t)(t + 1) * B : nullptr;
        a.soft = ps ? io->soft_raw + (size_t)t * B * V1 : nullptr;
        a.ld_soft = V1;
        a.ps_u = (ps && io->ps_u) ? io->ps_u + (size_t)(t + 1) * B : nullptr;
        a.ps_prob = io->ps_prob;
        a.ss_u = io->ss_u ? io->ss_u + (size_t)(t + 1) * B : nullptr;
        a.ss_prob = io->ss_prob;
        a.ss_pick = io->ss_pick ? io->ss_pick + (size_t)(t + 1) * B : nullptr;
        a.decoding_constraint = io->decoding_constraint;
        a.step = t + 1;
        a.unfinished = w.unfinished;
        a.it_next = w.it_all + (size_t)(t + 1) * B;
        a.seq = io->seq; a.slp = io->slp; a.stv = io->stv; a.seq_ld = T;
        a.any_unfinished = w.any_unf;
        CIC_PROF(CIC_PROF_SAMPLER, st, rc = cic_logsoftmax_sample(&a, s));
        if (rc) return rc;
    }
    if (io->first_token) {
        RUN(cic_fill_i32(io->L, 1, T, st));      // teacher forcing: every step carries a target
    } else {
        RUN(cic_finalize_len(w.any_unf, T, io->L, s));
    }
    if (ps) RUN(cic_soft_mask(io->soft_raw, io->seq, io->L, io->soft_out, T, B, V1, st));
#undef RUN
    return 0;
}
