// Beam-search decode of the speaker (evaluation path): AttModel.sample_beam, models/AttModel.py:150-289.
// The reference decodes one image at a time, moves the log-probs to the host every step and merges beams in
// nested Python loops; here all B x beam rows advance together through the same per-timestep kernels as the
// training decode (attention rows index their image's regions through att_div), and the per-image beam merge,
// the state re-ordering and the done-beam bookkeeping are device kernels: no host synchronisation at all.
//
// Reference behaviours reproduced (they define parity):
//   * a beam that emits <eos> = 0 is recorded as done but keeps decoding (nothing removes it, :256-263);
//   * the recorded score is a 0-dim VIEW of the running sums (`beam_logprobs_sum[vix]`, :262), so the final sort
//     (:281-282) ranks a recorded beam by the FINAL running sum of the slot it was recorded from; seq / logps are
//     snapshots.  The winner is the first recorded entry among those with the largest such score, so only the
//     first entry of every slot has to be kept;
//   * candidates: the top beam_size words of every beam, word-rank major / beam minor, stably sorted by the fp32
//     cumulative log-prob (:209-224); at t = 1 only beam 0 is active (:211-213).
#include "cic_common.h"
#include "engine_util.h"

namespace {

constexpr int MAXB = 16;     // beam_size limit (candidate table 16 x 16)

// top-`beam` of one log-prob row (value descending, lowest index among equals); one workgroup per row.
// The decoding constraint (:201-204) masks the word the beam emitted at the previous step.
__global__ __launch_bounds__(256) void beam_topk_kernel(float* __restrict__ logp, int V1, int beam, const int32_t* __restrict__ seq_prev_tok,
                                                        float* __restrict__ ys, int32_t* __restrict__ ix) {
    __shared__ float shv[4];
    __shared__ int shi[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    float* lp = logp + (size_t)row * V1;
    if (seq_prev_tok && tid == 0) lp[seq_prev_tok[row]] = -INFINITY;
    __syncthreads();
    for (int c = 0; c < beam; ++c) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int j = tid; j < V1; j += 256) {
            const float v = lp[j];
            if (v > bv || (v == bv && j < bi)) { bv = v; bi = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { shv[tid >> 6] = bv; shi[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (shv[w] > bv || (shv[w] == bv && shi[w] < bi)) { bv = shv[w]; bi = shi[w]; }
            ys[(size_t)row * beam + c] = bv;
            ix[(size_t)row * beam + c] = bi;
            if (bi != 0x7fffffff) lp[bi] = -INFINITY;      // taken: the next pass finds the next rank
        }
        __syncthreads();
    }
}

// One wave per image: candidate table -> the beam_size survivors -> new histories, sums, parents, next tokens,
// first-done snapshots.  (beam <= 16: at most 256 candidates; the selection is a serial scan by lane 0, the
// history copies use all lanes.)
__global__ __launch_bounds__(64) void beam_merge_kernel(const float* __restrict__ ys, const int32_t* __restrict__ ix, int beam, int t,
                                                        int T, const int32_t* __restrict__ seq_old, const float* __restrict__ lp_old,
                                                        int32_t* __restrict__ seq_new, float* __restrict__ lp_new,
                                                        float* __restrict__ beam_sum, int32_t* __restrict__ parent,
                                                        int32_t* __restrict__ it, int32_t* __restrict__ done_seq,
                                                        float* __restrict__ done_lp, int32_t* __restrict__ done_order,
                                                        int32_t* __restrict__ done_count) {
    __shared__ float cp[MAXB * MAXB], cr[MAXB * MAXB];
    __shared__ int cw[MAXB * MAXB], cq[MAXB * MAXB], sel[MAXB];
    __shared__ float newsum[MAXB];
    __shared__ int fresh[MAXB];
    const int img = blockIdx.x, lane = threadIdx.x;
    const int rows = t == 1 ? 1 : beam;                                  // :211-213
    const int nc = beam * rows;
    for (int i = lane; i < nc; i += 64) {
        const int c = i / rows, q = i % rows;                            // word-rank major, beam minor (:214-222)
        const float r = ys[((size_t)img * beam + q) * beam + c];
        cr[i] = r;
        cp[i] = beam_sum[img * beam + q] + r;                            // fp32 add, as the FloatTensor sum
        cw[i] = ix[((size_t)img * beam + q) * beam + c];
        cq[i] = q;
    }
    __syncthreads();
    if (lane == 0) {
        unsigned long long used[4] = {0, 0, 0, 0};
        for (int v = 0; v < beam; ++v) {                                 // stable descending sort, first beam_size (:224)
            int best = -1;
            for (int i = 0; i < nc; ++i) {
                if ((used[i >> 6] >> (i & 63)) & 1ull) continue;
                if (best < 0 || cp[i] > cp[best]) best = i;
            }
            used[best >> 6] |= 1ull << (best & 63);
            sel[v] = best;
            newsum[v] = cp[best];
        }
    }
    __syncthreads();
    for (int v = 0; v < beam; ++v) {
        const int s = sel[v], q = cq[s];
        const size_t dst = ((size_t)img * beam + v) * T, src = ((size_t)img * beam + q) * T;
        for (int pos = lane; pos < T; pos += 64) {
            int32_t tok = 0;
            float lpv = 0.f;
            if (pos < t - 1) { tok = seq_old[src + pos]; lpv = lp_old[src + pos]; }     // fork beam q into slot v (:231-234)
            else if (pos == t - 1) { tok = cw[s]; lpv = cr[s]; }                        // :249-251
            seq_new[dst + pos] = tok;
            lp_new[dst + pos] = lpv;
        }
    }
    __syncthreads();
    if (lane == 0) {
        int count = done_count[img];
        for (int v = 0; v < beam; ++v) {
            const int s = sel[v];
            beam_sum[img * beam + v] = newsum[v];                        // :253
            parent[img * beam + v] = img * beam + cq[s];
            it[img * beam + v] = cw[s];
            fresh[v] = 0;
            if (cw[s] == 0 || t == T) {                                  // :256-263: recorded as done (and decoding on)
                if (done_order[img * beam + v] < 0) {                    // only a slot's FIRST entry can win the final sort
                    done_order[img * beam + v] = count;
                    fresh[v] = 1;
                }
                ++count;
            }
        }
        done_count[img] = count;
    }
    __syncthreads();
    for (int v = 0; v < beam; ++v) {
        if (!fresh[v]) continue;                                         // block-uniform
        const size_t row = ((size_t)img * beam + v) * T;
        for (int pos = lane; pos < T; pos += 64) {                       // clone of the slot's history (:259-261)
            done_seq[row + pos] = seq_new[row + pos];
            done_lp[row + pos] = lp_new[row + pos];
        }
    }
}

// h2[r] = h[parent[r]], c2[r] = c[parent[r]]                                   (:236-246)
__global__ __launch_bounds__(256) void beam_gather_kernel(const float* __restrict__ h, const float* __restrict__ c,
                                                          const int32_t* __restrict__ parent, float* __restrict__ h2,
                                                          float* __restrict__ c2, int R, int H) {
    const int H4 = H >> 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * H4) return;
    const int r = i / H4, j = i % H4;
    const int p = parent[r];
    reinterpret_cast<f32x4*>(h2)[i] = reinterpret_cast<const f32x4*>(h + (size_t)p * H)[j];
    reinterpret_cast<f32x4*>(c2)[i] = reinterpret_cast<const f32x4*>(c + (size_t)p * H)[j];
}

// the recorded beam with the largest FINAL slot sum, first recorded among equals           (:281-285)
__global__ __launch_bounds__(64) void beam_final_kernel(const float* __restrict__ beam_sum, const int32_t* __restrict__ done_order,
                                                        const int32_t* __restrict__ done_seq, const float* __restrict__ done_lp,
                                                        int beam, int T, int32_t* __restrict__ seq, float* __restrict__ logps,
                                                        float* __restrict__ score) {
    const int img = blockIdx.x, lane = threadIdx.x;
    int best = -1;
    for (int v = 0; v < beam; ++v) {
        if (done_order[img * beam + v] < 0) continue;
        if (best < 0 || beam_sum[img * beam + v] > beam_sum[img * beam + best] ||
            (beam_sum[img * beam + v] == beam_sum[img * beam + best] && done_order[img * beam + v] < done_order[img * beam + best]))
            best = v;
    }
    for (int pos = lane; pos < T; pos += 64) {
        seq[(size_t)img * T + pos] = best >= 0 ? done_seq[((size_t)img * beam + best) * T + pos] : 0;
        logps[(size_t)img * T + pos] = best >= 0 ? done_lp[((size_t)img * beam + best) * T + pos] : 0.f;
    }
    if (lane == 0) score[img] = best >= 0 ? beam_sum[img * beam + best] : 0.f;
}

struct BeamWs {
    float *att_m, *p_att, *x, *h[2], *c[2], *att_h, *att_res, *alpha, *dot, *pre, *out, *logp, *bias_ih;
    float *ys, *lp[2], *beam_sum, *done_lp;
    int32_t *it, *ix, *parent, *seq[2], *done_seq, *done_order, *done_count;
    size_t bytes;
};
BeamWs beam_carve(const cic_speaker_dims& d, int beam, void* base) {
    BeamWs w;
    Carver c(base);
    const size_t B = d.B, K = d.K, H = d.H, E = d.E, A = d.A, T = d.T, V1 = d.V + 1, R = B * beam;
    w.att_m = c.f32(B * K * H);      // embedded regions with the padded rows zeroed (att_masks only)
    w.p_att = c.f32(B * K * A);
    w.x = c.f32(R * E);
    for (int i = 0; i < 2; ++i) { w.h[i] = c.f32(R * H); w.c[i] = c.f32(R * H); }
    w.att_h = c.f32(R * A);
    w.att_res = c.f32(R * H);
    w.alpha = c.f32(R * K);
    w.dot = c.f32(R * K);
    w.pre = c.f32(R * 5 * H);
    w.out = c.f32(R * H);
    w.logp = c.f32(R * V1);
    w.bias_ih = c.f32(5 * H);
    w.ys = c.f32(R * beam);
    for (int i = 0; i < 2; ++i) { w.lp[i] = c.f32(R * T); w.seq[i] = c.i32(R * T); }
    w.beam_sum = c.f32(R);
    w.done_lp = c.f32(R * T);
    w.it = c.i32(R);
    w.ix = c.i32(R * beam);
    w.parent = c.i32(R);
    w.done_seq = c.i32(R * T);
    w.done_order = c.i32(R);
    w.done_count = c.i32(B);
    w.bytes = c.used();
    return w;
}

__global__ void fill_f32_kernel(float* p, int n, float v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace

extern "C" size_t cic_speaker_beam_ws_bytes(const cic_speaker_dims* d, int beam) {
    if (!d || beam < 1 || beam > MAXB) return 0;
    return beam_carve(*d, beam, nullptr).bytes;
}

extern "C" int cic_speaker_beam_search(const cic_speaker_dims* dp, const cic_speaker_params* p, const cic_beam_io* io,
                                       void* ws, size_t ws_bytes, cic_stream_t s) {
    CIC_REQUIRE(dp && p && io && ws && io->att_pre && io->seq && io->logps && io->score);
    const cic_speaker_dims& d = *dp;
    const int beam = io->beam;
    CIC_REQUIRE(beam >= 1 && beam <= MAXB && beam <= d.V + 1);            // the reference's assert (:164-166)
    CIC_REQUIRE(d.B > 0 && d.K > 0 && d.K <= 64 && d.T > 0 && d.T <= 64);
    CIC_REQUIRE((d.H & 3) == 0 && (d.E & 3) == 0 && (d.A & 3) == 0);
    BeamWs w = beam_carve(d, beam, ws);
    CIC_REQUIRE(ws_bytes >= w.bytes);
    hipStream_t st = cic_s(s);
    const int B = d.B, K = d.K, H = d.H, E = d.E, A = d.A, T = d.T, V1 = d.V + 1, R = B * beam;
    int rc;
#define RUN(x) if ((rc = (x)) != 0) return rc
    // evaluation mode: att = relu(att_embed(att_raw)) without dropout; p_att = ctx2att(att)      (:158-162)
    const float* att = io->att_pre;
    if (io->att_masks) {          // ragged region counts: padded rows of the embedded regions are 0 (pack_wrapper :44-51)
        RUN(cic_att_keep_rows(io->att_pre, nullptr, 0.f, io->att_masks, w.att_m, B, K, H, st));
        att = w.att_m;
    }
    RUN(gemm_nt(att, H, p->ctx2att_w, H, w.p_att, A, B * K, A, H, p->ctx2att_b, false, false, st));
    RUN(cic_add_vec(p->i2h_b, p->h2h_b, w.bias_ih, 5 * H, st));
    CIC_HIP(hipMemsetAsync(w.h[0], 0, sizeof(float) * R * H, st));
    CIC_HIP(hipMemsetAsync(w.c[0], 0, sizeof(float) * R * H, st));
    CIC_HIP(hipMemsetAsync(w.seq[0], 0, sizeof(int32_t) * R * T, st));
    CIC_HIP(hipMemsetAsync(w.lp[0], 0, sizeof(float) * R * T, st));
    CIC_HIP(hipMemsetAsync(w.beam_sum, 0, sizeof(float) * R, st));
    CIC_HIP(hipMemsetAsync(w.done_count, 0, sizeof(int32_t) * B, st));
    RUN(cic_fill_i32(w.done_order, R, -1, st));
    RUN(cic_fill_i32(w.it, R, d.V + 1, st));                               // <bos> (:190-193)
    int cur = 0;     // state buffers holding (h, c) that the next core step reads
    int hist = 0;    // history buffers (beam_seq / beam_seq_logprobs) that are current
    for (int t = 0; t <= T; ++t) {
        if (t >= 1) {
            const int32_t* prev = nullptr;
            // :201-204: the word a beam emitted at the previous step (beam_seq[t-2][row], still in it[row]) -> -inf
            if (io->decoding_constraint && t > 1) prev = w.it;
            hipLaunchKernelGGL(beam_topk_kernel, dim3(R), dim3(256), 0, st, w.logp, V1, beam, prev, w.ys, w.ix);
            CIC_LAUNCH_CHECK();
            hipLaunchKernelGGL(beam_merge_kernel, dim3(B), dim3(64), 0, st, w.ys, w.ix, beam, t, T, w.seq[hist], w.lp[hist],
                               w.seq[hist ^ 1], w.lp[hist ^ 1], w.beam_sum, w.parent, w.it, w.done_seq, w.done_lp,
                               w.done_order, w.done_count);
            CIC_LAUNCH_CHECK();
            hist ^= 1;
            if (t == T) break;                                             // the reference's last core call is unused
            hipLaunchKernelGGL(beam_gather_kernel, dim3(cic_cdiv(R * (H / 4), 256)), dim3(256), 0, st, w.h[cur], w.c[cur],
                               w.parent, w.h[cur ^ 1], w.c[cur ^ 1], R, H);
            CIC_LAUNCH_CHECK();
            cur ^= 1;
        }
        // one core step for all B x beam rows                              (:268-270)
        RUN(cic_embed_fwd2(p->embed_w, dual1((const int32_t*)w.it), Dual<const uint8_t>{nullptr, nullptr}, 0.f, dual1(w.x), R, 1,
                           E, st));
        RUN(gemm_nt(w.h[cur], H, p->h2att_w, H, w.att_h, A, R, A, H, p->h2att_b, false, false, st));
        RUN(cic_attn_fwd2(dual1((const float*)w.att_h), dual1((const float*)w.p_att), dual1(att), p->alpha_w, p->alpha_b,
                          io->att_masks, dual1(w.att_res), dual1(w.alpha), dual1(w.dot), R, 1, K, A, H, st, beam));
        RUN(gemm_nt2(w.x, E, p->i2h_w, E, E, w.h[cur], H, p->h2h_w, H, H, w.pre, 5 * H, R, 5 * H, w.bias_ih, st));
        RUN(gemm_nt(w.att_res, H, p->a2c_w, H, w.pre + 3 * H, 5 * H, R, 2 * H, H, p->a2c_b, true, false, st));
        RUN(cic_cell_fwd2(dual1((const float*)w.pre), dual1((const float*)w.c[cur]), Dual<const uint8_t>{nullptr, nullptr}, 0.f,
                          dual1(w.h[cur ^ 1]), dual1(w.c[cur ^ 1]), dual1(w.out), R, 1, H, st));
        cur ^= 1;
        RUN(gemm_nt(w.out, H, p->logit_w, H, w.logp, V1, R, V1, H, p->logit_b, false, false, st));
        cic_sampler_args a = {};
        a.logits = w.logp; a.B = R; a.V1 = V1; a.ld = V1; a.mode = CIC_SAMPLE_NONE;
        RUN(cic_logsoftmax_sample2(&a, nullptr, st));
    }
    hipLaunchKernelGGL(beam_final_kernel, dim3(B), dim3(64), 0, st, w.beam_sum, w.done_order, w.done_seq, w.done_lp, beam, T,
                       io->seq, io->logps, io->score);
    CIC_LAUNCH_CHECK();
#undef RUN
    return 0;
}
