// Host-side helpers for the sequence engines: workspace carving and GEMM call shorthands.
#pragma once
#include "cic_common.h"

struct Carver {
    char* base;
    size_t off;
    explicit Carver(void* b) : base(static_cast<char*>(b)), off(0) {}
    void* take(size_t bytes) {
        off = (off + 255) & ~size_t(255);   // 256-B aligned slices (float4 / dwordx4 safe)
        void* p = base ? base + off : nullptr;
        off += bytes;
        return p;
    }
    float* f32(size_t n) { return static_cast<float*>(take(n * sizeof(float))); }
    int32_t* i32(size_t n) { return static_cast<int32_t*>(take(n * sizeof(int32_t))); }
    double* f64(size_t n) { return static_cast<double*>(take(n * sizeof(double))); }
    uint64_t* u64(size_t n) { return static_cast<uint64_t*>(take(n * sizeof(uint64_t))); }
    uint16_t* u16(size_t n) { return static_cast<uint16_t*>(take(n * sizeof(uint16_t))); }
    size_t used() const { return (off + 255) & ~size_t(255); }
};

// Activations of one speaker decode, in the order the backward pass walks them.
struct SpkWs {
    float *att, *p_att;                       // [B,K,H], [B,K,A]
    float *x_all, *h_all, *c_all;             // [T,B,E], [T+1,B,H], [T+1,B,H]
    float *att_h_all, *att_res_all;           // [T,B,A], [T,B,H]
    float *alpha_all, *dot_all;               // [T,B,K]
    float *pre_all, *out_all, *logp_all;      // [T,B,5H], [T,B,H], [T,B,V+1]
    float* bias_ih;                           // [5H] = i2h.bias + h2h.bias
    float *pre_img, *zeros;                   // [B,5H], [B,H]: the image step of an FCModel decode; a zero state
    int32_t *it_all, *unfinished, *any_unf;   // [T+1,B], [B], [T+1]
    uint16_t *att_bf, *p_att_bf;              // bf16 copies of att / p_att (compute_dtype bf16 only)
    float *part, *lse_all;                    // row partials of one step's logits [6][B][nparts]; [T,B] log-sum-exp rows
    uint16_t* logit_parts;                    // [3][V+1][H] bf16: the logit weights cut into their parts once per decode
    uint16_t* gate_parts;                     // compute_dtype bf16: bf16 images of i2h [5H,E], h2h [5H,H], h2att [A,H], once per decode
    unsigned* tsync;                          // spk_teacher_seq_kernel: [strips of 16 rows][T][3] hand-off counters + error word
    size_t bytes;
};
SpkWs spk_carve(const cic_speaker_dims& d, void* base);

int cic_fill_i32(int32_t* p, int n, int32_t v, hipStream_t st);
int cic_add_vec(const float* a, const float* b, float* o, int n, hipStream_t st);
// pair-capable launchers (nb = 1: one decode, the .b pointers are unused; nb = 2: a pair in lock step)
bool cic_attn_pair_ok(int K, int A, int H);
int cic_attn_fwd2(Dual<const float> att_h, Dual<const float> p_att, Dual<const float> att, const float* w_alpha,
                  const float* b_alpha, const float* masks, Dual<float> att_res, Dual<float> alpha, Dual<float> dot, int B,
                  int nb, int K, int A, int H, hipStream_t st, int att_div = 1,
                  Dual<const uint16_t> p_att_bf = Dual<const uint16_t>{nullptr, nullptr},
                  Dual<const uint16_t> att_bf = Dual<const uint16_t>{nullptr, nullptr},
                  Dual<const int32_t> live = Dual<const int32_t>{nullptr, nullptr});
// x <- bf16(x) (round to nearest even, in place as f32) and its packed bf16 copy
int cic_round_pack_bf16(float* x, uint16_t* packed, int64_t n, hipStream_t st);
int cic_cell_fwd2(Dual<const float> pre, Dual<const float> c_prev, Dual<const uint8_t> keep, float p_drop, Dual<float> h_new,
                  Dual<float> c_new, Dual<float> out, int B, int nb, int H, hipStream_t st, int state_dropped = 0);
int cic_embed_fwd2(const float* E, Dual<const int32_t> it, Dual<const uint8_t> keep, float p_drop, Dual<float> x, int B,
                   int nb, int Ed, hipStream_t st, int plain = 0);
bool cic_a2c_cell_fused_ok(int H);
// attention + att2ctx product + cell of a decode step as ONE launch (speaker_fwd.hip: attn_a2c_cell_kernel)
struct AttnCellLaunch {
    Dual<const float> att_h, p_att, att;
    Dual<const uint16_t> p_att_bf, att_bf;       // compute_dtype bf16: packed bf16 region features instead of p_att / att
    const float *w_alpha, *b_alpha, *masks;
    Dual<float> att_res, alpha, dot;
    const float *Wa, *ba;
    Dual<float> pre;
    Dual<const float> c_prev;
    Dual<const uint8_t> keep;
    float p_drop;
    Dual<float> h_new, c_new, out;
    Dual<const int32_t> live;
    unsigned* cnt;                               // [nb * ceil(B / 32)] arrival counters of this step, zero before the launch
    unsigned* err;                               // the decode's error word
    uint32_t* status;                            // the caller's sticky status word or null (cic.h)
    int B, nb, K;
};
bool cic_attn_cell_fused_ok(int B, int nb, int K, int A, int H, bool bf, int device_shared);
int cic_attn_a2c_cell(const AttnCellLaunch& L, hipStream_t st);
int cic_a2c_cell_fused(Dual<const float> att_res, const float* Wa, const float* ba, Dual<float> pre, Dual<const float> c_prev,
                       Dual<const uint8_t> keep, float p_drop, Dual<float> h_new, Dual<float> c_new, Dual<float> out, int B,
                       int nb, int H, hipStream_t st, Dual<const int32_t> live = Dual<const int32_t>{nullptr, nullptr});
int cic_logsoftmax_sample2(const cic_sampler_args* a, const cic_sampler_args* b, hipStream_t st);
// the sampler on the row partials of the step's logits (no pass over the vocabulary); writes the rows' lse
int cic_teacher_tokens(const int64_t* pick, int32_t* it_all, int32_t* unfinished, int32_t* any_unf, int32_t* seq, int T, int B,
                       hipStream_t st, unsigned* zsync = nullptr, int nzsync = 0);
int cic_teacher_finish_all(const float* part, int np, int part_rows, const float* logits, int ld, const int64_t* pick,
                           float* lse_all, float* slp, int T, int B, hipStream_t st);
int cic_finalize_len2(Dual<const int> any_unfinished, int T, Dual<int> L, int nb, hipStream_t st);
int cic_sample_finish2(const cic_sampler_args* a, const float* part_a, int part_rows_a, float* lse_a,
                       const cic_sampler_args* b, const float* part_b, int part_rows_b, float* lse_b, int np, hipStream_t st,
                       int32_t* L_a = nullptr, int32_t* L_b = nullptr, int T = 0);
// dropout of the embedded regions with ragged region counts: rows beyond an image's own regions become 0
// the teacher-forced recurrence of a decode as ONE launch (speaker_fwd.hip: spk_teacher_seq_kernel)
struct TeacherSeqLaunch {
    const float *h2h_w, *h2att_w, *h2att_b, *a2c_w, *a2c_b, *alpha_w, *alpha_b, *p_att, *att, *masks;
    const uint8_t* out_keep;
    float *pre_all, *h_all, *c_all, *att_h_all, *att_res_all, *alpha_all, *dot_all, *out_all;
    unsigned* sync;           // cic_cdiv(B, 16) * T * 3 + 1 words
    uint32_t* status;         // the caller's sticky status word (cic.h) or null
    int bf16;                 // != 0: compute_dtype bf16 - weight tiles and handed-over rows as bf16 MFMA operands, f32 accumulation
    float scale;
    int B, K, T;
};
bool cic_teacher_seq_ok(int B, int K, int H, int A, int E);
int cic_resident_cus(const void* kernel, int threads, size_t lds_bytes);   // CUs that each admit one such workgroup, or 0
int cic_teacher_seq(const TeacherSeqLaunch& L, hipStream_t st);
int cic_apply_keep2(const float* x, Dual<const uint8_t> keep, float p_drop, Dual<float> y, int64_t n, hipStream_t st);
int cic_att_keep_rows(const float* x, const uint8_t* keep, float p_drop, const float* masks, float* y, int B, int K, int H,
                      hipStream_t st);
int cic_relu_keep_fwd(const float* xpre, const uint8_t* keep, float p_drop, float* x, int64_t n, hipStream_t st);
int cic_soft_mask(const float* soft_raw, const int32_t* seq, const int32_t* L, float* soft_out, int T, int B, int V1,
                  hipStream_t st);

// The stream of an engine call together with the arithmetic of its batched products (cic_gemm_args.precision): the GEMM
// shorthands below take it where they used to take the stream, and it still converts to the stream for launches.
struct GemmCtx {
    hipStream_t st;
    int precision;
    const int32_t* live = nullptr;      // cic_gemm_args.live / live_min of the products launched through this context
    int live_min = 0;
    GemmCtx(hipStream_t s, int p = CIC_PRECISION_F32) : st(s), precision(p) {}
    operator hipStream_t() const { return st; }
};

// C[M,N] = A[M,K] W[N,K]^T (+bias) (+C)           — nn.Linear forward
static inline int gemm_nt(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                          const float* bias, bool accumulate, bool relu, GemmCtx st) {
    cic_gemm_args g = {};
    g.M = M; g.N = N; g.K = K;
    g.A = A; g.lda = lda; g.a_kc = 1;
    g.B = W; g.ldb = ldw; g.b_kc = 1;
    g.C = C; g.ldc = ldc; g.bias = bias; g.accumulate = accumulate; g.relu = relu;
    g.precision = st.precision;
    return cic_gemm_f32(&g, st.st);
}
// C = A1 W1^T + A2 W2^T + bias
static inline int gemm_nt2(const float* A1, int lda1, const float* W1, int ldw1, int K1, const float* A2, int lda2,
                           const float* W2, int ldw2, int K2, float* C, int ldc, int M, int N, const float* bias,
                           GemmCtx st) {
    cic_gemm_args g = {};
    g.M = M; g.N = N; g.K = K1;
    g.A = A1; g.lda = lda1; g.a_kc = 1;
    g.B = W1; g.ldb = ldw1; g.b_kc = 1;
    g.K2 = K2; g.A2 = A2; g.lda2 = lda2; g.B2 = W2; g.ldb2 = ldw2;
    g.C = C; g.ldc = ldc; g.bias = bias;
    g.precision = st.precision;
    return cic_gemm_f32(&g, st.st);
}
// C[M,N] = A[M,K] Bm[K,N] (+C)                    — dX = dY W   (W stored [K=out, N=in]); a gradient product:
// the summation order is free
static inline int gemm_nn(const float* A, int lda, const float* Bm, int ldb, float* C, int ldc, int M, int N, int K,
                          bool accumulate, GemmCtx st, bool order_free = true, bool c_is_zero = false) {
    cic_gemm_args g = {};
    g.M = M; g.N = N; g.K = K;
    g.A = A; g.lda = lda; g.a_kc = 1;
    g.B = Bm; g.ldb = ldb; g.b_kc = 0;
    g.C = C; g.ldc = ldc; g.accumulate = accumulate; g.sum_order_free = order_free; g.c_is_zero = c_is_zero;
    g.precision = st.precision;
    if (c_is_zero) { g.live = st.live; g.live_min = st.live_min; }      // leaving the launch out leaves the zeros
    return cic_gemm_f32(&g, st.st);
}
// the same product in a FORWARD pass (soft caption rows @ embedding): fixed summation order
static inline int gemm_nn_fwd(const float* A, int lda, const float* Bm, int ldb, float* C, int ldc, int M, int N, int K,
                              bool accumulate, GemmCtx st) {
    return gemm_nn(A, lda, Bm, ldb, C, ldc, M, N, K, accumulate, st, false);
}
// C = A1 B1 + A2 B2 (+C), all B stored [K,N]
static inline int gemm_nn2(const float* A1, int lda1, const float* B1, int ldb1, int K1, const float* A2, int lda2,
                           const float* B2, int ldb2, int K2, float* C, int ldc, int M, int N, bool accumulate,
                           GemmCtx st, bool c_is_zero = false) {
    cic_gemm_args g = {};
    g.M = M; g.N = N; g.K = K1;
    g.A = A1; g.lda = lda1; g.a_kc = 1;
    g.B = B1; g.ldb = ldb1; g.b_kc = 0;
    g.K2 = K2; g.A2 = A2; g.lda2 = lda2; g.B2 = B2; g.ldb2 = ldb2;
    g.C = C; g.ldc = ldc; g.accumulate = accumulate; g.sum_order_free = 1; g.c_is_zero = c_is_zero;
    g.precision = st.precision;
    if (c_is_zero) { g.live = st.live; g.live_min = st.live_min; }      // leaving the launch out leaves the zeros
    return cic_gemm_f32(&g, st.st);
}
// C[M,N] = At[K,M]^T Bm[K,N] (+C)                 — dW = dY^T X
static inline int gemm_tn(const float* At, int lda, const float* Bm, int ldb, float* C, int ldc, int M, int N, int K,
                          bool accumulate, GemmCtx st, float* bias_grad = nullptr, float* bias_grad2 = nullptr) {
    cic_gemm_args g = {};
    g.M = M; g.N = N; g.K = K;
    g.A = At; g.lda = lda; g.a_kc = 0;
    g.B = Bm; g.ldb = ldb; g.b_kc = 0;
    g.C = C; g.ldc = ldc; g.accumulate = accumulate; g.sum_order_free = 1;   // weight gradients
    g.colsum_A = bias_grad; g.colsum_A2 = bias_grad2;                        // db += colsum(dY), a by-product of the A tiles
    g.precision = st.precision;
    return cic_gemm_f32(&g, st.st);
}
