// Speaker sequence engines: the host loops of AttModel.sample / AttModel.forward
// (models/AttModel.py:103-148,291-452) as back-to-back kernel launches on ONE HIP stream with
// no host synchronisation.  The reference's per-step `unfinished.sum() == 0` sync + break is
// replaced by device-side flags and a device-side length L; every activation the backward pass
// needs is written once into a caller-provided workspace.
#include "cic_common.h"
#include "engine_util.h"

namespace {

CIC_SWITCH(g_early_stop, 1);         // development build: cic_debug_early_stop(0) = every step of a decode runs in full (A/B)
CIC_SWITCH(g_gates_att_fused, 1);   // development build: cic_debug_gates_att_fused(0) = separate h2att launch (A/B timing)
CIC_SWITCH(g_teacher_seq, 1);       // development build: cic_debug_teacher_seq(0) = the teacher-forced recurrence as three launches per step (A/B, parity)

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

// Start-of-decode state of up to two decodes in ONE launch (was: three memsets, two fills and a vector add per decode,
// ~5 us each): zero (h_0, c_0) (init_hidden, AttModel.py:311), clear the any-unfinished flags, unfinished = 1,
// first token = <bos> (= vocab_size + 1, :324-326) or the caller's column (AttModel.forward: seq[:, 0], :131), and the
// summed bias of the i2h / h2h products.
struct DecodeInit {
    float* h;                  // [B,H] or null
    float* c;                  // [B,H] or null
    int32_t* any_unf;          // [T+1]
    int32_t* unfinished;       // [B]
    int32_t* it;               // [B]
    const int64_t* first_token;   // [B] or null
    float* bias_ih;            // [5H]
    float* x0;                 // [B,E] or null: (r4) the input of core step 0, x = dropout(relu(E[first token])) (AttModel.py:74-76,
    const uint8_t* keep0;      //       399; embed_fwd_kernel's arithmetic) - one launch less at the head of every decode
    unsigned* zero_words;      // or null: the hand-off counters + error word of the decode's in-launch hand-offs (attn_a2c_cell_kernel)
    int n_zero;
};
__global__ __launch_bounds__(256) void decode_init_kernel(DecodeInit a, DecodeInit b, const float* __restrict__ i2h_b,
                                                          const float* __restrict__ h2h_b, int BH4, int B, int T1, int H5,
                                                          int bos, const float* __restrict__ Emb, int Ed, float scale, int plain) {
    const DecodeInit d = blockIdx.y ? b : a;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    if (i < BH4) {
        if (d.h) reinterpret_cast<f32x4*>(d.h)[i] = z4;
        if (d.c) reinterpret_cast<f32x4*>(d.c)[i] = z4;
    }
    if (i < T1) d.any_unf[i] = 0;
    if (d.zero_words && i < d.n_zero) d.zero_words[i] = 0u;
    if (i < B) {
        d.unfinished[i] = 1;
        d.it[i] = d.first_token ? (int32_t)d.first_token[i] : bos;
    }
    if (i < H5) d.bias_ih[i] = i2h_b[i] + h2h_b[i];
    const int E4 = Ed >> 2;
    if (d.x0 && i < B * E4) {
        const int r = i / E4, j = i % E4;
        const int tok = d.first_token ? (int32_t)d.first_token[r] : bos;
        f32x4 v = reinterpret_cast<const f32x4*>(Emb + (size_t)tok * Ed)[j];
        uint32_t kp = 0x01010101u;
        if (d.keep0) kp = *reinterpret_cast<const uint32_t*>(d.keep0 + (size_t)i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float kf = (float)((kp >> (8 * e)) & 0xffu);
            const float q = plain ? v[e] : fmaxf(v[e], 0.f);
            v[e] = d.keep0 ? q * (kf * scale) : q;
        }
        reinterpret_cast<f32x4*>(d.x0)[i] = v;
    }
}

}  // namespace

#ifdef CIC_DEVTOOLS
extern "C" int cic_debug_gates_att_fused(int on) { g_gates_att_fused = on; return 0; }
extern "C" int cic_debug_early_stop(int on) { g_early_stop = on; return 0; }
extern "C" int cic_debug_teacher_seq(int on) { g_teacher_seq = on; return 0; }
#endif

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
    w.pre_img = c.f32(B * 5 * H);
    w.zeros = c.f32(B * H);
    w.it_all = c.i32((T + 1) * B);
    w.unfinished = c.i32(B);
    w.any_unf = c.i32(T + 1);
    w.att_bf = c.u16(B * K * H);
    w.p_att_bf = c.u16(B * K * A);
    w.part = c.f32((size_t)CIC_PART_PLANES * CIC_PART_MAX_ENTRIES);
    w.lse_all = c.f32(T * B);
    w.logit_parts = c.u16(3 * V1 * H);
    w.gate_parts = c.u16(5 * H * E + 5 * H * H + A * H);
    w.tsync = reinterpret_cast<unsigned*>(c.i32((((B + 15) / 16) * T * 3 + 1 + 3) / 4 * 4));
    w.bytes = c.used();
    return w;
}

extern "C" size_t cic_speaker_decode_ws_bytes(const cic_speaker_dims* d) {
    if (!d) return 0;
    return spk_carve(*d, nullptr).bytes;
}

static int check_dims(const cic_speaker_dims& d) {
    CIC_REQUIRE(d.B > 0 && d.K >= 0 && d.K <= 64 && d.T > 0 && d.T <= 64);
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
                   true, GemmCtx(cic_s(s), d.compute_dtype == CIC_DTYPE_BF16 ? CIC_PRECISION_BF16 : CIC_PRECISION_F32));
}

static int decode_fwd_impl(const cic_speaker_dims* dp, const cic_speaker_params* p, const cic_decode_io* const* ios,
                           void* const* wss, const size_t* ws_bytes, int nb, cic_stream_t s);

extern "C" int cic_speaker_decode_fwd(const cic_speaker_dims* dp, const cic_speaker_params* p,
                                      const cic_decode_io* io, void* ws, size_t ws_bytes, cic_stream_t s) {
    CIC_REQUIRE(dp && p && io && ws);
    return decode_fwd_impl(dp, p, &io, &ws, &ws_bytes, 1, s);
}

static bool pair_ok(const cic_speaker_dims& d, const cic_decode_io* a, const cic_decode_io* b) {
    auto ps = [](int m) { return m == CIC_SAMPLE_GUMBEL_PS || m == CIC_SAMPLE_MULTINOMIAL_PS; };
    if (ps(a->mode) || ps(b->mode)) return false;                 // soft-input steps are not row-wise launches
    if (a->fc_mode || b->fc_mode || d.K == 0) return false;
    if ((a->first_token != nullptr) != (b->first_token != nullptr)) return false;
    if ((d.B & 31) || 2 * d.B > 256) return false;                // row blocks of the per-step products
    if ((d.H & 7) || (d.E & 7) || (d.A & 7)) return false;        // register-streaming GEMM operands
    return cic_attn_pair_ok(d.K, d.A, d.H);
}

extern "C" int cic_speaker_decode_pair_fused(const cic_speaker_dims* d, const cic_decode_io* io_a, const cic_decode_io* io_b) {
    return (d && io_a && io_b && pair_ok(*d, io_a, io_b)) ? 1 : 0;
}

extern "C" int cic_speaker_decode_fwd_pair(const cic_speaker_dims* dp, const cic_speaker_params* p,
                                           const cic_decode_io* io_a, void* ws_a, size_t ws_a_bytes,
                                           const cic_decode_io* io_b, void* ws_b, size_t ws_b_bytes, cic_stream_t s) {
    CIC_REQUIRE(dp && p && io_a && ws_a && io_b && ws_b && ws_a != ws_b);
    const cic_decode_io* ios[2] = {io_a, io_b};
    void* wss[2] = {ws_a, ws_b};
    const size_t wsb[2] = {ws_a_bytes, ws_b_bytes};
    if (pair_ok(*dp, io_a, io_b)) return decode_fwd_impl(dp, p, ios, wss, wsb, 2, s);
    int rc = decode_fwd_impl(dp, p, ios, wss, wsb, 1, s);          // shapes outside the paired kernels: one after the other
    if (!rc) rc = decode_fwd_impl(dp, p, ios + 1, wss + 1, wsb + 1, 1, s);
    return rc;
}

// nb = 1: one decode.  nb = 2: two decodes of the same images (e.g. the sampled and the greedy decode of a joint
// step) advance in lock step: every per-timestep kernel is launched ONCE over 2B rows, rows [0,B) reading and
// writing decode a's workspace / noise / outputs and rows [B,2B) decode b's.  Each row's arithmetic is exactly
// that of the single-decode launch (same kernels, same K order), so the results are bit-identical to two
// sequential cic_speaker_decode_fwd calls; what changes is 2x fewer launches and twice the rows per launch
// (B = 128 alone fills only half of the 256 CUs in the one-workgroup-per-image kernels).
static int decode_fwd_impl(const cic_speaker_dims* dp, const cic_speaker_params* p, const cic_decode_io* const* ios,
                           void* const* wss, const size_t* ws_bytes, int nb, cic_stream_t s) {
    CIC_REQUIRE(dp && p && ios && wss && (nb == 1 || nb == 2));
    const cic_speaker_dims& d = *dp;
    if (int rc = check_dims(d)) return rc;
    const bool bf = d.compute_dtype == CIC_DTYPE_BF16;
    GemmCtx st(cic_s(s), bf ? CIC_PRECISION_BF16 : CIC_PRECISION_F32);
    const int B = d.B, K = d.K, H = d.H, E = d.E, A = d.A, T = d.T, V1 = d.V + 1;
    const float p_drop = d.p_drop;
    int rc;
#define RUN(x) if ((rc = (x)) != 0) return rc
    SpkWs w[2];
    const cic_decode_io* io[2] = {ios[0], nb == 2 ? ios[1] : nullptr};
    const bool fc = ios[0]->fc_mode != 0;               // FCModel: no attention, image step first, dropped state
    CIC_REQUIRE(fc ? (nb == 1 && d.K == 0 && ios[0]->x0 != nullptr) : d.K > 0);
    bool ps = false;
    for (int q = 0; q < nb; ++q) {
        CIC_REQUIRE(io[q] && wss[q]);
        CIC_REQUIRE(io[q]->mode >= CIC_SAMPLE_GREEDY && io[q]->mode <= CIC_SAMPLE_MULTINOMIAL_PS);
        const bool psq = io[q]->mode == CIC_SAMPLE_GUMBEL_PS || io[q]->mode == CIC_SAMPLE_MULTINOMIAL_PS;
        CIC_REQUIRE(!psq || (nb == 1 && io[q]->soft_raw && io[q]->xpre && io[q]->soft_out));
        ps = ps || psq;
        w[q] = spk_carve(d, wss[q]);
        CIC_REQUIRE(ws_bytes[q] >= w[q].bytes);
        CIC_REQUIRE((fc || io[q]->att_pre) && io[q]->seq && io[q]->slp && io[q]->L && !(fc && psq));
    }
    // teacher forcing without scheduled sampling embeds all its steps in one launch of its own (below)
    const bool init_embeds = !(nb == 1 && !fc && ios[0]->mode == CIC_SAMPLE_TEACHER && ios[0]->pick &&
                               !(ios[0]->ss_u && ios[0]->ss_prob > 0.f));
    // attention + att2ctx + cell of every sampling step as one launch with an in-launch hand-off: needs all its workgroups resident
    // and this process alone on the GPU.  Counters: [T][nb * ceil(B / 32)] words at the head of the (otherwise unused) hand-off
    // region of the teacher-forced kernel, the error word behind that region as there; zeroed with the decode's other state.
    bool attn_cell = !fc && cic_attn_cell_fused_ok(B, nb, K, A, H, bf, 0);
    for (int q = 0; q < nb; ++q) attn_cell = attn_cell && !io[q]->device_shared;
    const int n_tsync = cic_cdiv(B, 16) * T * 3 + 1;
    {
        DecodeInit di[2] = {};
        for (int q = 0; q < nb; ++q) {
            // fc: the image step below produces (h_0, c_0) from a zero state held in `zeros`
            di[q].h = fc ? w[q].zeros : w[q].h_all;
            di[q].c = fc ? nullptr : w[q].c_all;
            di[q].any_unf = w[q].any_unf; di[q].unfinished = w[q].unfinished; di[q].it = w[q].it_all;
            di[q].first_token = io[q]->first_token; di[q].bias_ih = w[q].bias_ih;
            if (init_embeds) {
                di[q].x0 = w[q].x_all;
                di[q].keep0 = fc ? nullptr : io[q]->x_keep;     // (row 0 of the [T+1,B,E] masks; the fc speaker embeds plain rows)
            }
        }
        if (attn_cell) { di[0].zero_words = w[0].tsync; di[0].n_zero = n_tsync; }
        const int BH4 = B * H / 4;
        int span = BH4 > 5 * H ? BH4 : 5 * H;
        if (span < T + 1) span = T + 1;
        if (attn_cell && span < n_tsync) span = n_tsync;
        if (span < B) span = B;
        if (init_embeds && span < B * E / 4) span = B * E / 4;
        hipLaunchKernelGGL(decode_init_kernel, dim3(cic_cdiv(span, 256), nb), dim3(256), 0, st, di[0], di[1], p->i2h_b, p->h2h_b,
                           BH4, B, T + 1, 5 * H, d.V + 1, p->embed_w, E, 1.0f / (1.0f - p_drop), fc ? 1 : 0);
        CIC_LAUNCH_CHECK();
    }
    // a pair over the same embedded regions, no ragged masks: both dropouts in one launch
    const bool keep2 = nb == 2 && !fc && !io[0]->att_masks && !io[1]->att_masks && io[0]->att_pre == io[1]->att_pre &&
                       (io[0]->att_keep != nullptr) == (io[1]->att_keep != nullptr);
    if (keep2)
        RUN(cic_apply_keep2(io[0]->att_pre, Dual<const uint8_t>{io[0]->att_keep, io[1]->att_keep}, io[0]->att_keep ? p_drop : 0.f,
                            Dual<float>{w[0].att, w[1].att}, (int64_t)B * K * H, st));
    for (int q = 0; q < nb; ++q) {
        if (!fc) {
            // att = dropout(relu(att_embed(att_raw)));  p_att = ctx2att(att)      (AttModel.py:315,319)
            // with att_masks: only an image's own region rows are embedded, the padded rows are 0   (:44-51)
            if (keep2) {
            } else if (io[q]->att_masks) {
                RUN(cic_att_keep_rows(io[q]->att_pre, io[q]->att_keep, io[q]->att_keep ? p_drop : 0.f, io[q]->att_masks, w[q].att,
                                      B, K, H, st));
            } else {
                RUN(cic_apply_keep(io[q]->att_pre, io[q]->att_keep, io[q]->att_keep ? p_drop : 0.f, w[q].att, (int64_t)B * K * H, s));
            }
            RUN(gemm_nt(w[q].att, H, p->ctx2att_w, H, w[q].p_att, A, B * K, A, H, p->ctx2att_b, false, false, st));
            if (bf) {
                // bf16 storage: att and p_att are rounded to bf16 (the f32 buffers keep the rounded values for the backward
                // pass); the per-timestep attention streams the bf16 copies
                RUN(cic_round_pack_bf16(w[q].att, w[q].att_bf, (int64_t)B * K * H, st));
                RUN(cic_round_pack_bf16(w[q].p_att, w[q].p_att_bf, (int64_t)B * K * A, st));
            }
        } else {
            // image step: (h0, c0) = LSTMCore(img_embed(fc_feats), zero state)          (FCModel.py:97-99,121,274-276,315)
            // h = 0, so h2h contributes its bias only (already in bias_ih)
            RUN(gemm_nt(io[q]->x0, E, p->i2h_w, E, w[q].pre_img, 5 * H, B, 5 * H, E, w[q].bias_ih, false, false, st));
            RUN(cic_cell_fwd2(dual1((const float*)w[q].pre_img), dual1((const float*)w[q].zeros), dual1(io[q]->out_keep),
                              io[q]->out_keep ? p_drop : 0.f, dual1(w[q].h_all), dual1(w[q].c_all), dual1(w[q].out_all), B, 1,
                              H, st, 1));
        }
    }
    if (nb == 1) w[1] = SpkWs{};                      // all-null second set
    const int M = nb * B;
    // Dual{a, b}: slab t of buffer `f` in both workspaces (b is null for a single decode)
#define SLAB(f, n) Dual<float>{w[0].f + (size_t)t * (n), nb == 2 ? w[1].f + (size_t)t * (n) : nullptr}
#define CSLAB(f, n) Dual<const float>{w[0].f + (size_t)t * (n), nb == 2 ? w[1].f + (size_t)t * (n) : nullptr}
    auto keep_at = [&](const uint8_t* base, size_t off) -> const uint8_t* { return base ? base + off : nullptr; };
    // per-step products over the rows of both decodes: second row block at the b pointers
    auto pair_gemm = [&](cic_gemm_args& g, const float* A_b, const float* A2_b, float* C_b) {
        if (nb == 2) { g.rows_blk = B; g.A_b = A_b; g.A2_b = A2_b; g.C_b = C_b; }
        return cic_gemm_f32(&g, st);
    };

    // Teacher forcing without scheduled sampling (AttModel.forward :103-148 with ss_prob = 0): the fed tokens are the
    // targets, so nothing in the recurrence waits for a logit.  The token columns and every step's input embedding are
    // written before the loop, and the logit product + log-sum-exp + gathered log-probabilities run once over all T*B
    // rows after it (one LDS-tiled GEMM instead of T walker launches and T sampler launches).
    const bool teacher_batched = nb == 1 && !fc && !ps && io[0]->mode == CIC_SAMPLE_TEACHER && io[0]->pick &&
                                 !(io[0]->ss_u && io[0]->ss_prob > 0.f) && (int64_t)T * B <= CIC_PART_MAX_ENTRIES;
    // ... and at the flagship widths the recurrence itself is ONE launch (spk_teacher_seq_kernel): x_t i2h^T + bias of all steps as
    // one batched product, then h2h / h2att / a2c tiles stationary in 16-row x 16-unit workgroups with in-strip hand-offs
    // (bf16 variant: the f32 copies of att / p_att hold the bf16-rounded values, so the kernel's attention computes what the packed
    // copies would give; the batched x i2h^T product follows the variant's one-part precision like every batched product)
    const bool teacher_seq = teacher_batched && g_teacher_seq && !io[0]->device_shared && cic_teacher_seq_ok(B, K, H, A, E);
    if (teacher_batched) {
        RUN(cic_teacher_tokens(io[0]->pick, w[0].it_all, w[0].unfinished, w[0].any_unf, io[0]->seq, T, B, st, w[0].tsync,
                               cic_cdiv(B, 16) * T * 3 + 1));
        const int t_first = teacher_seq ? 0 : 1;          // (the per-step loop embeds step 0 itself)
        if (T > t_first) {
            const uint8_t* xk1 = io[0]->x_keep ? io[0]->x_keep + (size_t)t_first * B * E : nullptr;
            RUN(cic_embed_fwd2(p->embed_w, Dual<const int32_t>{w[0].it_all + (size_t)t_first * B, nullptr}, Dual<const uint8_t>{xk1, nullptr},
                               xk1 ? p_drop : 0.f, Dual<float>{w[0].x_all + (size_t)t_first * B * E, nullptr}, (T - t_first) * B, 1, E, st, 0));
        }
    }
    if (teacher_seq) {
        RUN(gemm_nt(w[0].x_all, E, p->i2h_w, E, w[0].pre_all, 5 * H, T * B, 5 * H, E, w[0].bias_ih, false, false, st));
        TeacherSeqLaunch L = {};
        L.h2h_w = p->h2h_w; L.h2att_w = p->h2att_w; L.h2att_b = p->h2att_b; L.a2c_w = p->a2c_w; L.a2c_b = p->a2c_b;
        L.alpha_w = p->alpha_w; L.alpha_b = p->alpha_b; L.p_att = w[0].p_att; L.att = w[0].att; L.masks = io[0]->att_masks;
        L.out_keep = io[0]->out_keep;
        L.pre_all = w[0].pre_all; L.h_all = w[0].h_all; L.c_all = w[0].c_all; L.att_h_all = w[0].att_h_all;
        L.att_res_all = w[0].att_res_all; L.alpha_all = w[0].alpha_all; L.dot_all = w[0].dot_all; L.out_all = w[0].out_all;
        L.sync = w[0].tsync; L.status = io[0]->status; L.bf16 = bf ? 1 : 0; L.scale = 1.0f / (1.0f - p_drop); L.B = B; L.K = K; L.T = T;
        RUN(cic_teacher_seq(L, st));
    }
    // The logit weights are read unchanged by every step's logit product: cut them into their three bf16 parts once per
    // decode (8 us) instead of once per weight tile and workgroup inside the walker (cic_gemm_args.B_parts; bit-identical)
    const bool presplit_logit = !teacher_batched && !ps && H == 512 && V1 >= 2048 && ((size_t)V1 * H) % 4 == 0;   // the walker's shapes
    // compute_dtype bf16: ONE part - the logit weights and the three gate matrices as packed bf16 images, made once per decode;
    // the per-timestep products then stream 2 bytes per weight and issue one MFMA per k-step
    if (presplit_logit) RUN(bf ? cic_round_bf16(p->logit_w, (int64_t)V1 * H, w[0].logit_parts, s)
                               : cic_split_bf16x3(p->logit_w, (int64_t)V1 * H, w[0].logit_parts, s));
    const bool gate_img = bf && !fc && !teacher_seq && (((size_t)5 * H * E) % 8) == 0 && (((size_t)5 * H * H) % 8) == 0 && (((size_t)A * H) % 4) == 0 &&
                          (E % 8) == 0 && (H % 8) == 0;
    uint16_t* img_i2h = w[0].gate_parts;
    uint16_t* img_h2h = img_i2h + (size_t)5 * H * E;
    uint16_t* img_h2att = img_h2h + (size_t)5 * H * H;
    if (gate_img) {
        RUN(cic_round_bf16(p->i2h_w, (int64_t)5 * H * E, img_i2h, s));
        RUN(cic_round_bf16(p->h2h_w, (int64_t)5 * H * H, img_h2h, s));
        RUN(cic_round_bf16(p->h2att_w, (int64_t)A * H, img_h2att, s));
    }
    bool early_stop = g_early_stop && !fc && !ps && !teacher_batched;
    for (int q = 0; q < nb; ++q) early_stop = early_stop && !io[q]->first_token;
    bool len_in_sampler = !ps && !teacher_batched && !teacher_seq;
    for (int q = 0; q < nb; ++q) len_in_sampler = len_in_sampler && !io[q]->first_token;
    for (int t = 0; t < T && !teacher_seq; ++t) {
        const Dual<float> x = SLAB(x_all, B * E), att_h = SLAB(att_h_all, B * A), att_res = SLAB(att_res_all, B * H),
                          pre = SLAB(pre_all, B * 5 * H), out = SLAB(out_all, B * H), logp = SLAB(logp_all, B * V1);
        const Dual<const float> h = CSLAB(h_all, B * H), c = CSLAB(c_all, B * H);
        const Dual<float> h_new{w[0].h_all + (size_t)(t + 1) * B * H, nb == 2 ? w[1].h_all + (size_t)(t + 1) * B * H : nullptr};
        const Dual<float> c_new{w[0].c_all + (size_t)(t + 1) * B * H, nb == 2 ? w[1].c_all + (size_t)(t + 1) * B * H : nullptr};
        const Dual<const uint8_t> xk{keep_at(io[0]->x_keep, (size_t)t * B * E), nb == 2 ? keep_at(io[1]->x_keep, (size_t)t * B * E) : nullptr};
        // FCModel: out_keep row 0 belongs to the image step
        const Dual<const uint8_t> ok{keep_at(io[0]->out_keep, (size_t)(t + (fc ? 1 : 0)) * B * H),
                                     nb == 2 ? keep_at(io[1]->out_keep, (size_t)t * B * H) : nullptr};
        // dropout is on or off for the whole model (same p_drop); a decode without masks passes NULL
        CIC_REQUIRE(nb == 1 || ((io[0]->x_keep != nullptr) == (io[1]->x_keep != nullptr) &&
                                (io[0]->out_keep != nullptr) == (io[1]->out_keep != nullptr)));
        // Early stop (AttModel.py:401-408: the reference leaves the loop once every caption has ended).  No host sync here:
        // the sampler of step t-1 left "some caption is still open" in any_unf[t]; the heavy kernels of step t read it and
        // return at once when it is 0 for every decode of the launch.  Their outputs then keep what an earlier call left
        // (finite: the workspace is zero-filled when it is allocated); everything past a decode's length L is masked
        // downstream, and the sampler still runs and keeps tokens / flags / L exact.  Decodes that feed given tokens
        // (teacher forcing) run all their steps.
        Dual<const int32_t> live{nullptr, nullptr};
        if (early_stop && t >= 1) live = Dual<const int32_t>{w[0].any_unf + t, nb == 2 ? w[1].any_unf + t : nullptr};
        if (ps && t >= 1) {
            // xt = relu_dropout(soft_vec @ embed.weight)                   (:395-397), soft_vec un-masked
            float* xp = io[0]->xpre + (size_t)t * B * E;
            RUN(gemm_nn_fwd(io[0]->soft_raw + (size_t)(t - 1) * B * V1, V1, p->embed_w, E, xp, E, B, E, V1, false, st));
            RUN(cic_relu_keep_fwd(xp, xk.a, xk.a ? p_drop : 0.f, x.a, (int64_t)B * E, st));
        } else if ((t == 0 && !init_embeds) || (ps && t > 0)) {
            // xt = embed(it)                                               (:399); for t >= 1 the sampler launch of the
            // previous step wrote x[t] (fused embedding)
            RUN(cic_embed_fwd2(p->embed_w,
                               Dual<const int32_t>{w[0].it_all + (size_t)t * B, nb == 2 ? w[1].it_all + (size_t)t * B : nullptr},
                               fc ? Dual<const uint8_t>{nullptr, nullptr} : xk, (xk.a && !fc) ? p_drop : 0.f, x, B, nb, E, st,
                               fc ? 1 : 0));
        }
        // all_input_sums = i2h(xt) + h2h(h)  (:514) and the attention query att_h = h2att(h)  (:470) read the same
        // (x_t, h_{t-1}): one launch where the GEMM implements the column split (flagship widths), else two
        bool gates_done = false;
        if (!fc) {
            cic_gemm_args g = {};
            g.M = M; g.N = 5 * H + A; g.K = E; g.A = x.a; g.lda = E; g.a_kc = 1; g.B = p->i2h_w; g.ldb = E; g.b_kc = 1;
            g.K2 = H; g.A2 = h.a; g.lda2 = H; g.B2 = p->h2h_w; g.ldb2 = H;
            g.C = pre.a; g.ldc = 5 * H; g.bias = w[0].bias_ih;
            g.n_split = 5 * H; g.B2_tail = p->h2att_w; g.ldb2_tail = H; g.bias_tail = p->h2att_b;
            g.C_tail = att_h.a; g.C_tail_b = att_h.b; g.ldc_tail = A;
            if (bf) g.precision = CIC_PRECISION_BF16;
            if (gate_img) { g.B_parts = img_i2h; g.B2_parts = img_h2h; g.B2_tail_parts = img_h2att; }
            if (nb == 2) { g.rows_blk = B; g.A_b = x.b; g.A2_b = h.b; g.C_b = pre.b; }
            g.live = live.a; g.live_b = live.b;
            if (g_gates_att_fused && cic_gemm_split_ok(&g)) {
                RUN(cic_gemm_f32(&g, st));
                gates_done = true;
            }

        }
        // attention                                                        (:465-489)
        if (!fc && !gates_done) {
            cic_gemm_args g = {};
            g.M = M; g.N = A; g.K = H; g.A = h.a; g.lda = H; g.a_kc = 1; g.B = p->h2att_w; g.ldb = H; g.b_kc = 1;
            g.C = att_h.a; g.ldc = A; g.bias = p->h2att_b;
            RUN(pair_gemm(g, h.b, nullptr, att_h.b));
        }
        // att_masks are an input of the step (the same images in both decodes)
        CIC_REQUIRE(nb == 1 || io[0]->att_masks == io[1]->att_masks);
        const bool step_fused = attn_cell && gates_done;
        if (step_fused) {
            AttnCellLaunch L = {};
            L.att_h = Dual<const float>{att_h.a, att_h.b};
            L.p_att = Dual<const float>{w[0].p_att, w[1].p_att}; L.att = Dual<const float>{w[0].att, w[1].att};
            if (bf) {
                L.p_att_bf = Dual<const uint16_t>{w[0].p_att_bf, w[1].p_att_bf};
                L.att_bf = Dual<const uint16_t>{w[0].att_bf, w[1].att_bf};
            }
            L.w_alpha = p->alpha_w; L.b_alpha = p->alpha_b; L.masks = io[0]->att_masks;
            L.att_res = att_res; L.alpha = SLAB(alpha_all, B * K); L.dot = SLAB(dot_all, B * K);
            L.Wa = p->a2c_w; L.ba = p->a2c_b; L.pre = pre; L.c_prev = c; L.keep = ok; L.p_drop = ok.a ? p_drop : 0.f;
            L.h_new = h_new; L.c_new = c_new; L.out = out; L.live = live;
            L.cnt = w[0].tsync + (size_t)t * nb * cic_cdiv(B, 32); L.err = w[0].tsync + (n_tsync - 1); L.status = io[0]->status;
            L.B = B; L.nb = nb; L.K = K;
            CIC_TIMED(io[0]->timer, CIC_TIMED_ATTN_FWD, st, rc = cic_attn_a2c_cell(L, st));
            if (rc) return rc;
        }
        if (!fc && !step_fused) {
            CIC_TIMED(io[0]->timer, CIC_TIMED_ATTN_FWD, st,
                     rc = cic_attn_fwd2(Dual<const float>{att_h.a, att_h.b}, Dual<const float>{w[0].p_att, w[1].p_att},
                                        Dual<const float>{w[0].att, w[1].att}, p->alpha_w, p->alpha_b, io[0]->att_masks, att_res,
                                        SLAB(alpha_all, B * K), SLAB(dot_all, B * K), B, nb, K, A, H, st, 1,
                                        Dual<const uint16_t>{bf ? w[0].p_att_bf : nullptr, bf ? w[1].p_att_bf : nullptr},
                                        Dual<const uint16_t>{bf ? w[0].att_bf : nullptr, bf ? w[1].att_bf : nullptr}, live));
            if (rc) return rc;
        }
        // all_input_sums = i2h(xt) + h2h(h);  in_transform += a2c(att_res)   (:514,521-522)
        if (!gates_done) {
            cic_gemm_args g = {};
            g.M = M; g.N = 5 * H; g.K = E; g.A = x.a; g.lda = E; g.a_kc = 1; g.B = p->i2h_w; g.ldb = E; g.b_kc = 1;
            g.K2 = H; g.A2 = h.a; g.lda2 = H; g.B2 = p->h2h_w; g.ldb2 = H;
            g.C = pre.a; g.ldc = 5 * H; g.bias = w[0].bias_ih;
            RUN(pair_gemm(g, x.b, h.b, pre.b));
        }
        if (step_fused) {
        } else if (!fc && cic_a2c_cell_fused_ok(H)) {
            // a2c product + cell in one launch (flagship width)
            RUN(cic_a2c_cell_fused(Dual<const float>{att_res.a, att_res.b}, p->a2c_w, p->a2c_b, pre, c, ok, ok.a ? p_drop : 0.f,
                                   h_new, c_new, out, B, nb, H, st, live));
        } else {
            if (!fc) {
                cic_gemm_args g = {};
                g.M = M; g.N = 2 * H; g.K = H; g.A = att_res.a; g.lda = H; g.a_kc = 1; g.B = p->a2c_w; g.ldb = H; g.b_kc = 1;
                g.C = pre.a + 3 * H; g.ldc = 5 * H; g.bias = p->a2c_b; g.accumulate = 1;
                RUN(pair_gemm(g, att_res.b, nullptr, nb == 2 ? pre.b + 3 * H : nullptr));
            }
            RUN(cic_cell_fwd2(Dual<const float>{pre.a, pre.b}, c, ok, ok.a ? p_drop : 0.f, h_new, c_new, out, B, nb, H, st,
                              fc ? 1 : 0));
        }
        if (teacher_batched) continue;                  // logits of all steps at once, after the loop
        // logprobs = log_softmax(logit(output)); choose the input of step t+1   (:328-365,444)
        // Row-wise modes: the logits stay RAW in the workspace (+ their log-sum-exp per row); the vocabulary is reduced
        // to row partials by the logit product's own epilogue (or, for shapes that kernel does not take, by a pass over
        // the logits) and the sampler works on the partials.  Partial-sampling modes need whole soft rows: the row
        // kernel normalises in place as before.
        cic_gemm_args lg = {};
        lg.M = M; lg.N = V1; lg.K = H; lg.A = out.a; lg.lda = H; lg.a_kc = 1; lg.B = p->logit_w; lg.ldb = H; lg.b_kc = 1;
        lg.C = logp.a; lg.ldc = V1; lg.bias = p->logit_b;
        if (presplit_logit) lg.B_parts = w[0].logit_parts;
        if (bf) lg.precision = CIC_PRECISION_BF16;
        if (nb == 2) { lg.rows_blk = B; lg.A_b = out.b; lg.C_b = logp.b; }
        lg.live = live.a; lg.live_b = live.b;
        cic_logit_epilogue epi = {};
        int np = 0;
        if (!ps) {
            for (int q = 0; q < nb; ++q) {
                cic_logit_epi_rows& e = epi.blk[q];
                const int mode = io[q]->mode;
                e.mode = mode; e.inv_temp = 1.0f / io[q]->temp;
                const bool gum = mode == CIC_SAMPLE_GUMBEL_ST;
                const bool multi = mode == CIC_SAMPLE_MULTINOMIAL || mode == CIC_SAMPLE_MULTINOMIAL_ST;
                const bool ss = mode == CIC_SAMPLE_TEACHER && io[q]->ss_u && io[q]->ss_prob > 0.f && !io[q]->ss_pick;
                e.noise = (gum || (multi && !io[q]->pick) || ss) ? 1 : 0;
                CIC_REQUIRE(!e.noise || io[q]->U || io[q]->u_philox);
                e.U = (e.noise && io[q]->U) ? io[q]->U + (size_t)(t + 1) * B * V1 : nullptr; e.ldu = V1;
                e.philox = io[q]->u_philox; e.seed = io[q]->u_seed;
                e.elem0 = io[q]->u_offset * 4ull + (uint64_t)(t + 1) * (uint64_t)B * (uint64_t)V1;
                if (io[q]->decoding_constraint && t + 1 >= 2) { e.cons_seq = io[q]->seq; e.cons_ld = T; e.cons_col = t - 1; }
                e.part = w[q].part; e.part_rows = B;
            }
            np = cic_gemm_logit_parts(&lg);
            // the walker's Philox draw covers whole aligned groups of four uniforms
            for (int q = 0; q < nb; ++q)
                if (epi.blk[q].noise && !epi.blk[q].U && ((V1 & 3) || (epi.blk[q].elem0 & 3))) np = 0;
            if (np > 0) lg.epi = &epi;
        }
        CIC_TIMED(io[0]->timer, CIC_TIMED_LOGIT_GEMM, st, rc = cic_gemm_f32(&lg, st));
        if (rc) return rc;
        if (!ps && np == 0) {
            np = CIC_PART_MAX_ENTRIES / B < 32 ? CIC_PART_MAX_ENTRIES / B : 32;
            CIC_REQUIRE(np >= 1);
            for (int q = 0; q < nb; ++q) RUN(cic_logit_partials(q ? logp.b : logp.a, B, V1, V1, &epi.blk[q], np, s));
        }
        cic_sampler_args sa[2];
        for (int q = 0; q < nb; ++q) {
            cic_sampler_args& a = sa[q];
            a = cic_sampler_args{};
            a.logits = q ? logp.b : logp.a; a.B = B; a.V1 = V1; a.ld = V1;
            a.mode = io[q]->mode; a.temp = io[q]->temp;
            a.U = io[q]->U ? io[q]->U + (size_t)(t + 1) * B * V1 : nullptr; a.ldu = V1;
            a.pick = io[q]->pick ? io[q]->pick + (size_t)(t + 1) * B : nullptr;
            a.soft = ps ? io[q]->soft_raw + (size_t)t * B * V1 : nullptr;
            a.ld_soft = V1;
            a.ps_u = (ps && io[q]->ps_u) ? io[q]->ps_u + (size_t)(t + 1) * B : nullptr;
            a.ps_prob = io[q]->ps_prob;
            a.ss_u = io[q]->ss_u ? io[q]->ss_u + (size_t)(t + 1) * B : nullptr;
            a.ss_prob = io[q]->ss_prob;
            a.ss_pick = io[q]->ss_pick ? io[q]->ss_pick + (size_t)(t + 1) * B : nullptr;
            a.decoding_constraint = io[q]->decoding_constraint;
            a.step = t + 1;
            a.unfinished = w[q].unfinished;
            a.it_next = w[q].it_all + (size_t)(t + 1) * B;
            a.seq = io[q]->seq; a.slp = io[q]->slp; a.stv = io[q]->stv; a.seq_ld = T;
            a.any_unfinished = w[q].any_unf;
            if (!ps && t + 1 < T) {                  // the next step's input row, embedded by the sampler itself
                const uint8_t* xkn = (!fc && io[q]->x_keep) ? io[q]->x_keep + (size_t)(t + 1) * B * E : nullptr;
                a.emb_w = p->embed_w; a.emb_x = w[q].x_all + (size_t)(t + 1) * B * E; a.emb_keep = xkn;
                a.emb_scale = 1.0f / (1.0f - p_drop); a.emb_dim = E; a.emb_plain = fc ? 1 : 0;
            }
        }
        if (ps) {
            CIC_TIMED(io[0]->timer, CIC_TIMED_SAMPLER, st, rc = cic_logsoftmax_sample2(&sa[0], nb == 2 ? &sa[1] : nullptr, st));
        } else {
            // the last sampler launch of decodes that choose their own tokens also writes their lengths L
            const bool last = t == T - 1 && len_in_sampler;
            CIC_TIMED(io[0]->timer, CIC_TIMED_SAMPLER, st,
                     rc = cic_sample_finish2(&sa[0], w[0].part, B, w[0].lse_all + (size_t)t * B, nb == 2 ? &sa[1] : nullptr,
                                             w[1].part, B, nb == 2 ? w[1].lse_all + (size_t)t * B : nullptr, np, st,
                                             last ? io[0]->L : nullptr, last && nb == 2 ? io[1]->L : nullptr, T));
        }
        if (rc) return rc;
    }
#undef SLAB
#undef CSLAB
    if (teacher_batched) {
        const int R = T * B;
        RUN(gemm_nt(w[0].out_all, H, p->logit_w, H, w[0].logp_all, V1, R, V1, H, p->logit_b, false, false, st));
        int np = CIC_PART_MAX_ENTRIES / R < 32 ? CIC_PART_MAX_ENTRIES / R : 32;
        cic_logit_epi_rows e = {};
        e.mode = CIC_SAMPLE_NONE; e.inv_temp = 1.0f; e.ldu = V1; e.part = w[0].part; e.part_rows = R;
        RUN(cic_logit_partials(w[0].logp_all, R, V1, V1, &e, np, s));
        RUN(cic_teacher_finish_all(w[0].part, np, R, w[0].logp_all, V1, io[0]->pick, w[0].lse_all, io[0]->slp, T, B, st));
    }
    if (len_in_sampler) {
        // (written by the last sampler launch)
    } else if (nb == 2 && !io[0]->first_token && !io[1]->first_token) {
        RUN(cic_finalize_len2(Dual<const int>{w[0].any_unf, w[1].any_unf}, T, Dual<int>{io[0]->L, io[1]->L}, 2, st));
    } else {
        for (int q = 0; q < nb; ++q) {
            if (io[q]->first_token) {
                RUN(cic_fill_i32(io[q]->L, 1, T, st));      // teacher forcing: every step carries a target
            } else {
                RUN(cic_finalize_len(w[q].any_unf, T, io[q]->L, s));
            }
        }
    }
    if (ps) RUN(cic_soft_mask(io[0]->soft_raw, io[0]->seq, io[0]->L, io[0]->soft_out, T, B, V1, st));
#undef RUN
    return 0;
}
