/*
 * cic.h — C ABI of libcic_hip.so, the MI355X (gfx950) hot path of the cooperative
 * image-captioning joint training step (speaker <-> listener).
 *
 * Boundary: the reference has no FFI; its boundary is the Python module
 * API (models.*, misc.rewards).  The re-implemented Python host mirrors that API and calls
 * the entry points below through ctypes.  Every function cites the reference code whose
 * computation it replaces (paths relative to the reference repo root).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch types.
 *   - every pointer is a DEVICE pointer to a contiguous row-major buffer owned by the
 *     caller, unless the parameter name ends in _host.
 *   - explicit hipStream_t (passed as void*); nothing synchronises the device.
 *   - no hidden allocation: scratch comes from a caller workspace (see *_ws_bytes).
 *   - return 0 on success, non-zero error code otherwise; cic_last_error() has the text.
 *     Nothing throws across the ABI.
 *   - f32 everywhere (the reference computes in fp32); token ids are int64 at the API
 *     surface (torch LongTensor) and int32 inside workspaces.
 */
#ifndef CIC_H
#define CIC_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* cic_stream_t; /* hipStream_t */

/* ---- library ---------------------------------------------------------------------- */
int cic_version(void);
const char* cic_last_error(void);

/* ---- status word: failures of the one-launch recurrences, reported without a host synchronisation -------------------
 * Four loops of a step run as ONE launch whose workgroups hand state to each other inside the launch (the listener's GRU pass
 * and its BPTT loop, the speaker's teacher-forced recurrence and its BPTT loop), and so does the attention + att2ctx + cell
 * launch of every sampling decode step (one hand-off).  They need every workgroup RESIDENT at once:
 * one per CU.  When something else holds CUs (a second process on the GPU that did not say so through device_shared, a
 * CU-masked queue) a workgroup waits for partners that never start; every such wait is bounded (1 s), the workgroup that gives
 * up poisons what it produces with NaN, and - new in round 4 - sets its loop's bit in the caller-owned device word
 * `status` of the io struct (cic_decode_io.status, cic_listener_io.status; NULL = not reported):
 *     CIC_STATUS_GRU_FWD | CIC_STATUS_GRU_BWD | CIC_STATUS_TEACHER | CIC_STATUS_BPTT | CIC_STATUS_DECODE_STEP.
 * The word is STICKY: the library only ever ORs into it; the caller zeroes it once and reads it when it likes (the trainer
 * copies it to pinned memory beside the step's loss and raises; no extra synchronisation).  cic_clamp_adam_guarded reads the
 * same word on the device: while it is non-zero the update is SKIPPED - parameters, moments and the gradient stay as they
 * were - and CIC_STATUS_UPDATE_SKIPPED is set, so a poisoned gradient never reaches the weights.  A process that shares its
 * GPU must set device_shared (CIC_SHARED_DEVICE=1 on the Python host): the loops then run as per-step launches. */
enum {
    CIC_STATUS_GRU_FWD = 1,          /* gru_seq_kernel         (VSEFCModel.py:95-140) */
    CIC_STATUS_GRU_BWD = 2,          /* gru_seq_bwd_kernel */
    CIC_STATUS_TEACHER = 4,          /* spk_teacher_seq_kernel (AttModel.py:103-148) */
    CIC_STATUS_BPTT = 8,             /* spk_bptt_seq_kernel    (AttModel.py:465-531 reversed) */
    CIC_STATUS_DECODE_STEP = 16,     /* attn_a2c_cell_kernel   (one decode step's attention -> att2ctx + cell, AttModel.py:465-529) */
    CIC_STATUS_UPDATE_SKIPPED = 256  /* cic_clamp_adam_guarded found the word set and left the weights alone */
};

/* ---- in-situ kernel timing: a caller-owned object, no library state -------------------------------------
 * A decode whose cic_decode_io.timer is set brackets every launch of the kernels listed below with a pair of HIP
 * events on its own stream (forward and backward); the events live in the cic_timer.  cic_timer_collect
 * synchronises them and returns the summed elapsed time and the launch count of one kernel id since the last
 * reset.  One timer per stream / host thread; NULL timer = nothing is recorded. */
typedef struct cic_timer cic_timer;
enum { CIC_TIMED_ATTN_FWD = 0, CIC_TIMED_LOGIT_GEMM = 1, CIC_TIMED_ATTN_BWD = 2, CIC_TIMED_SAMPLER = 3, CIC_TIMED_COUNT = 4 };
cic_timer* cic_timer_create(void);
void cic_timer_destroy(cic_timer* t);
int cic_timer_reset(cic_timer* t);
int cic_timer_collect(cic_timer* t, int id, double* total_ms, int* launches);
/* average elapsed time (us) of `pairs` back-to-back event pairs with nothing between them on stream s: what the bracket
 * itself adds to every timed launch (synchronises s) */
int cic_timer_bracket_overhead(cic_timer* t, int pairs, double* avg_us, cic_stream_t s);

/* ---- RNG (replaces torch.rand / nn.Dropout's bernoulli_ draws) ---------------------- */
/* Philox4x32-10 counter RNG.  u[i] = (r >> 8) * 2^-24 in [0,1), as torch.rand does
 * (models/gumbel.py:6-11).  Element i uses counter (offset + i/4), lane i%4. */
int cic_uniform_f32(float* out, int64_t n, uint64_t seed, uint64_t offset, cic_stream_t s);
/* keep[i] = u[i] >= p  (1 = keep), nn.Dropout in training mode (models/AttModel.py:74-85,506). */
int cic_dropout_keep_u8(uint8_t* keep, int64_t n, float p, uint64_t seed, uint64_t offset,
                        cic_stream_t s);
/* The same draws for up to CIC_KEEP_MAX_SEGMENTS tensors in ONE launch (the three masks of a decode: att_embed, the
 * token embeddings, the LSTM outputs).  Segment i is exactly what cic_dropout_keep_u8(keep[i], n[i], p, seed,
 * offset[i]) writes; the arrays are host arrays. */
#define CIC_KEEP_MAX_SEGMENTS 8
int cic_dropout_keep_u8_multi(uint8_t* const* keep, const int64_t* n, const uint64_t* offset, int count, float p,
                              uint64_t seed, cic_stream_t s);

/* ---- dense contraction on the f32 MFMA (v_mfma_f32_32x32x2_f32) --------------------- */
/* C[M,N] = op(A)[M,K] * op(B)[K,N] (+ A2*B2 over K2) (+ bias[N]) (+ C if accumulate), then
 * optional ReLU.  Row-major with leading dimensions.
 *   a_kc != 0: op(A)[m,k] = A[m*lda + k]   (K contiguous: activations x[M,K])
 *   a_kc == 0: op(A)[m,k] = A[k*lda + m]   (A stored [K,M]:  dW = dY^T X)
 *   b_kc != 0: op(B)[k,n] = B[n*ldb + k]   (nn.Linear weight W[N,K]:  y = x W^T)
 *   b_kc == 0: op(B)[k,n] = B[k*ldb + n]   (B stored [K,N]:  dX = dY W)
 * Replaces every nn.Linear / mm / matmul on the path: models/AttModel.py:82-88,140,444,
 * 470,477,503-505,514,522; models/VSEFCModel.py:28,42,74-76,104,146. */
typedef struct {
    int M, N, K;
    const float* A; int lda; int a_kc;
    const float* B; int ldb; int b_kc;
    int K2;                       /* 0 = no second operand pair */
    const float* A2; int lda2;    /* same a_kc / b_kc as the first pair */
    const float* B2; int ldb2;
    float* C; int ldc;
    const float* bias;            /* [N] or NULL */
    int accumulate;               /* C += ... */
    int relu;
    /* Row blocks (optional, per-timestep products of a PAIR of decodes advancing in lock step): when
     * rows_blk > 0, rows [rows_blk, M) of op(A), op(A2) and C live at A_b, A2_b and C_b (row 0 of those
     * = row rows_blk of the product).  Requires a_kc != 0, rows_blk % 32 == 0, M <= 2*rows_blk <= 256. */
    int sum_order_free;           /* != 0: the caller accepts a summation order that varies run to run (partial tiles
                                     combined with float atomics).  The engines set it for gradient products only;
                                     forward activations keep a fixed order. */
    int c_is_zero;                /* != 0 (with sum_order_free, accumulate == 0): C already holds zeros, so partial products may
                                     be added into it */
    /* Bias-gradient by-product of a weight-gradient product dW = dY^T X (a_kc == 0, A = dY stored [K, M]):
     * colsum_A[m] += sum_k A[k*lda + m] (and the same into colsum_A2 if not NULL), added with float atomics by the
     * workgroups of the first tile column while their A tiles are in LDS - no separate column-sum pass over dY. */
    float* colsum_A;
    float* colsum_A2;
    int rows_blk;
    const float* A_b;
    const float* A2_b;
    float* C_b;
    /* Column split (optional; two products of one decode step that read the same state in ONE launch:
     * [i2h(x) + h2h(h) | h2att(h)], AttModel.py:470,514): when n_split > 0, columns [n_split, N) are the product of the
     * SECOND operand pair only, with weight rows B2_tail[(n - n_split)*ldb2_tail + k], bias bias_tail[n - n_split], and
     * go to C_tail[m*ldc_tail + (n - n_split)] (rows at or beyond rows_blk: C_tail_b).  Supported where
     * cic_gemm_split_ok() says so (K-contiguous operands, K = K2 = 512, n_split % 16 == 0); else leave n_split = 0. */
    int n_split;
    const float* B2_tail; int ldb2_tail;
    const float* bias_tail;
    float* C_tail; float* C_tail_b; int ldc_tail;
    /* Fused vocabulary epilogue (optional; the logit product of a decode step, AttModel.py:444 + :328-365): while a
     * workgroup still holds the logits it made, it reduces them to ROW PARTIALS (below) so that the log-softmax and the
     * sampler need no second pass over the [M, N] logits.  C still receives the raw logits (the backward pass reads
     * them).  Only where cic_gemm_logit_parts() > 0; else leave NULL and run cic_logit_partials on C. */
    const struct cic_logit_epilogue* epi;
    /* Arithmetic of the LDS-tiled (batched) products; the per-timestep kernels always use the f32-input MFMA.
     *   CIC_PRECISION_F32 (0, default): f32 in, f32 out, f32 accuracy - the implementation may cut the operands into three
     *     bf16 parts and sum six part products on the bf16 matrix cores (error ~1e-7 relative, as an f32 fma chain);
     *   CIC_PRECISION_F32_MFMA: the f32-input MFMA only (bitwise a k-ordered f32 fma chain per output);
     *   CIC_PRECISION_BF16: operands rounded to bf16 once, f32 accumulation (reduced precision, ~3e-3 relative). */
    int precision;
    /* Optional: B already cut into its three bf16 parts (cic_split_bf16x3: uint16 images [3][N][K], K-contiguous), for a B
     * that many launches read unchanged (the logit weights: T launches per decode).  Used by the logit walker only; the
     * results are those of the launch without it, bit for bit.  NULL elsewhere. */
    const uint16_t* B_parts;
    /* Optional: device flags of a decode loop, "some caption of the decode (pair) is still being written"
     * (AttModel.py:401-408: the reference breaks out of the loop once every caption has ended).  When live is not NULL and
     * *live < live_min (and live_b is NULL or *live_b < live_min) the launch returns at once and leaves C untouched;
     * live_min <= 0 counts as 1.  The forward loops pass the flag of their step (0 / 1); the BPTT loop passes the decode's
     * length L with live_min = t + 1 for its step t (steps at or beyond L carry no gradient; their products add into
     * buffers cleared beforehand, so leaving them out leaves exact zeros).  Honoured by the per-timestep kernels of the
     * decode engines' flagship shapes; other kernels compute as if it were NULL (same results, the engines mask
     * everything past a decode's length anyway). */
    const int32_t* live;
    const int32_t* live_b;
    int live_min;
    /* CIC_PRECISION_BF16 only (optional): B2 and B2_tail as packed bf16 images (cic_round_bf16), beside B_parts holding ONE image
     * of B - the per-timestep gate product of a decode then streams 2 bytes per weight.  Same leading dimensions as the f32
     * matrices.  NULL: the f32 matrices are read and rounded on the fly. */
    const uint16_t* B2_parts;
    const uint16_t* B2_tail_parts;
} cic_gemm_args;
enum { CIC_PRECISION_F32 = 0, CIC_PRECISION_F32_MFMA = 1, CIC_PRECISION_BF16 = 2 };
/* Row partials of the vocabulary: the columns of a row are dealt to `nparts` parts; a part reduces its columns
 * (after the decoding constraint, AttModel.py:438-442: one column per row set to -inf) to six numbers, stored as
 * planes  part[plane][row][p],  plane stride = part_rows * nparts floats:
 *   0  m1     max x                         1  s1   sum exp(x - m1)          (log-softmax: lse = M + log S)
 *   2  kbest  best sampling key             3  xbest the logit at that column  4  kidx  its column (int32 bits)
 *   5  s2     GUMBEL_ST: sum exp(k - kbest) (the straight-through value y = softmax(k)[it], gumbel.py:28);
 *             multinomial / teacher modes: sum exp((x - m1) * inv_temp)       (y = softmax(logp/tau), multinomial.py)
 * sampling key k:  GREEDY x;  GUMBEL_ST (x + g) * inv_temp;  MULTINOMIAL, MULTINOMIAL_ST, TEACHER x * inv_temp + g
 * (g = 0 without noise), g = -log(-log(u + 1e-20) + 1e-20) in f32 (gumbel.py:6-11).  Ties: lowest column.
 * Uniform u of (row r, column c): U[r*ldu + c], or - U == NULL and philox != 0 - element elem0 + r*ldu + c of the
 * Philox stream of cic_uniform_f32(seed, offset 0) (counter = element / 4, lane = element % 4). */
typedef struct {
    int mode;                  /* CIC_SAMPLE_*; CIC_SAMPLE_NONE: planes 0-1 only */
    float inv_temp;
    int noise;                 /* != 0: the keys carry Gumbel noise */
    const float* U; int ldu;
    int philox; uint64_t seed; uint64_t elem0;
    const int32_t* cons_seq; int cons_ld; int cons_col;   /* constrained column of row r: cons_seq[r*cons_ld + cons_col]; NULL: none */
    float* part; int part_rows;
} cic_logit_epi_rows;
typedef struct cic_logit_epilogue {
    cic_logit_epi_rows blk[2];   /* rows [0, rows_blk) and [rows_blk, M) of the product (blk[0] alone without row blocks) */
} cic_logit_epilogue;
enum { CIC_PART_PLANES = 6, CIC_PART_MAX_ENTRIES = 16384 };   /* rows * nparts never exceeds 16384 per row block */
int cic_gemm_f32(const cic_gemm_args* a, cic_stream_t s);
/* x[i] = p0 + p1 + p2 with p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1) (round to nearest even; the residuals
 * are exact): parts[0..n), [n..2n), [2n..3n) - the split every bf16-part kernel applies to its operands on the fly. */
int cic_split_bf16x3(const float* x, int64_t n, uint16_t* parts, cic_stream_t s);
/* packed[i] = bf16(x[i]), round to nearest even: the one-part operand image of CIC_PRECISION_BF16 (n % 4 == 0) */
int cic_round_bf16(const float* x, int64_t n, uint16_t* packed, cic_stream_t s);
/* number of parts per row the fused epilogue of cic_gemm_f32 writes for these arguments; 0: not fused for them */
int cic_gemm_logit_parts(const cic_gemm_args* a);
/* the same partials from logits already in memory (any shape): rows of `logits` [M, ld], nparts parts of contiguous
 * columns */
int cic_logit_partials(const float* logits, int M, int N, int ld, const cic_logit_epi_rows* e, int nparts, cic_stream_t s);
/* 1 if cic_gemm_f32 implements the column split for these arguments, 0 if not */
int cic_gemm_split_ok(const cic_gemm_args* a);
/* measurement helper (stateless): average duration (us) of `iters` back-to-back launches of the product, HIP events on s */
int cic_gemm_f32_timed(const cic_gemm_args* a, int iters, double* avg_us, cic_stream_t s);
/* out[n] (+)= sum_m X[m*ldx + n]   — bias gradients. */
int cic_colsum_f32(const float* X, int M, int N, int ldx, float* out, int accumulate,
                   cic_stream_t s);

/* ---- speaker (att2in2) forward kernels ----------------------------------------------- */
/* Attention.forward, models/AttModel.py:465-489.  att_h = h2att(h) (bias included) comes
 * from cic_gemm_f32.  masks: f32[B,K] or NULL.  Writes att_res[B,H], alpha[B,K] and, if dot
 * is not NULL, the pre-softmax scores dot[B,K] (saved for the backward pass).
 * Requires K <= 64, A and H multiples of 4 and <= 1024. */
int cic_attn_fwd(const float* att_h, const float* p_att, const float* att, const float* w_alpha,
                 const float* b_alpha, const float* masks, float* att_res, float* alpha, float* dot,
                 int B, int K, int A, int H, cic_stream_t s);
/* Average duration of one cic_attn_fwd launch, HIP events on the stream, measured from C++ (helper of
 * bench.py: roofline.achieved = algorithmic bytes / this duration).  With a `pollute` buffer the launches
 * are interleaved with a kernel that streams it, and that kernel's own time is subtracted: the attention
 * kernel then runs in the cache state it sees inside a decode step. */
int cic_attn_fwd_timed(const float* att_h, const float* p_att, const float* att, const float* w_alpha,
                       const float* b_alpha, float* att_res, float* alpha, int B, int K, int A, int H,
                       int iters, const float* pollute, int64_t pollute_floats, double* avg_us, cic_stream_t s);
/* Att2in2Core.forward pointwise part, models/AttModel.py:515-529.  pre[B,5H] = i2h(x)+h2h(h)
 * with a2c(att_res) added to columns [3H,5H).  keep: u8[B,H] dropout keep mask or NULL. */
int cic_cell_fwd(const float* pre, const float* c_prev, const uint8_t* keep, float p_drop,
                 float* h_new, float* c_new, float* out, int B, int H, cic_stream_t s);
/* x = dropout(relu(E[it])), models/AttModel.py:74-76,399.  it: int32[B]. */
int cic_embed_fwd(const float* E, const int32_t* it, const uint8_t* keep, float p_drop, float* x,
                  int B, int Ed, cic_stream_t s);
/* y = x * keep / (1-p)  (n multiple of 4; keep NULL = copy), models/AttModel.py:82-85. */
int cic_apply_keep(const float* x, const uint8_t* keep, float p_drop, float* y, int64_t n,
                   cic_stream_t s);

/* Sampler modes of the fused log-softmax row kernel. */
enum {
    CIC_SAMPLE_NONE = 0,           /* log-softmax only (teacher forcing, AttModel.py:140) */
    CIC_SAMPLE_GREEDY = 1,         /* torch.max, ties -> lowest index (AttModel.py:328-329) */
    CIC_SAMPLE_MULTINOMIAL = 2,    /* it ~ softmax(logp/temp) (AttModel.py:332-343); Gumbel-max draw or pick[] */
    CIC_SAMPLE_GUMBEL_ST = 3,      /* models/gumbel.py:17-30 */
    CIC_SAMPLE_MULTINOMIAL_ST = 4, /* models/multinomial.py:4-27 */
    CIC_SAMPLE_GUMBEL_PS = 6,      /* partial sampling, models/gumbel_softmax.py:17-42 */
    CIC_SAMPLE_MULTINOMIAL_PS = 7, /* partial sampling, models/multinomial_soft.py:5-35 (exp(logp/tau), unnormalised) */
    CIC_SAMPLE_TEACHER = 5         /* teacher forcing (AttModel.forward :116-141): slp = logp[pick] (the target);
                                      next input = pick, or with scheduled sampling (:118-129) a draw from
                                      softmax(logp) for the rows whose ss_u < ss_prob */
};
typedef struct {
    float* logits;        /* [B, ld] in: logits, out: log-probs (in place) */
    int B, V1, ld;        /* V1 = vocab_size + 1 */
    int mode;
    float temp;           /* temperature / gumbel_temp / multinomial_temp */
    const float* U;       /* [B, ldu] uniforms for the Gumbel noise, or NULL */
    int ldu;
    const int64_t* pick;  /* [B] externally chosen tokens (multinomial modes) / targets (teacher), or NULL */
    const float* ss_u;    /* [B] scheduled-sampling uniforms (teacher mode) or NULL */
    float ss_prob;        /* scheduled-sampling probability */
    const int64_t* ss_pick; /* [B] externally drawn scheduled-sampling tokens, or NULL (Gumbel-max draw from U) */
    float* soft;          /* PS modes: out [B, ld_soft] the soft / straight-through row (first V1 columns), else NULL */
    int ld_soft;
    const float* ps_u;    /* PS modes: [B] row uniforms; rows with ps_u < ps_prob get the hard (straight-through) row */
    float ps_prob;        /* prob_gumbel_softmax / prob_multinomial_soft */
    int decoding_constraint; /* != 0: suppress the previously appended token seq[b, step-2] (step >= 2) */
    int step;             /* reference loop iteration t >= 1 whose input token is chosen */
    int32_t* unfinished;  /* [B] in/out */
    int32_t* it_next;     /* [B] out: un-masked token fed to the next core step */
    int32_t* seq;         /* [B, seq_ld] out: column step-1 = it * unfinished */
    float* slp;           /* [B, seq_ld] out: column step-1 = logp[it] */
    float* stv;           /* [B, seq_ld] out (ST modes, may be NULL): straight-through value */
    int seq_ld;
    int32_t* any_unfinished; /* [seq_length+1] zero-initialised flags, entry step is OR-ed */
    /* optional fused embedding of the chosen (un-masked) token = the next core step's input (AttModel.py:399):
     * emb_x[b,:] = dropout(relu(emb_w[it])) with keep mask emb_keep[b,:] (NULL: no dropout); emb_plain: no ReLU.
     * emb_dim % 4 == 0 and emb_dim <= 4096.  emb_x == NULL: not fused. */
    const float* emb_w;
    float* emb_x;
    const uint8_t* emb_keep;
    float emb_scale;
    int emb_dim;
    int emb_plain;
} cic_sampler_args;
/* logit bias/GEMM output -> F.log_softmax + sampling + EOS bookkeeping,
 * models/AttModel.py:328-365,401-434,438-444. */
int cic_logsoftmax_sample(const cic_sampler_args* a, cic_stream_t s);
/* L = first step t>=1 with no unfinished row, minus 1 (the reference's break, AttModel.py:407-408);
 * seq_length if none. */
int cic_finalize_len(const int32_t* any_unfinished, int T, int32_t* L, cic_stream_t s);

/* ---- speaker sequence engines (host loops over the kernels above, one stream, no sync) ---- */
typedef struct {
    int B, K, D, H, E, A, V; /* V = vocab_size: logits have V+1 columns, embedding V+2 rows.
                                K = 0: no region features (FCModel decodes, see fc_mode) */
    int T;                   /* seq_length */
    float p_drop;            /* drop_prob_lm */
    int compute_dtype;       /* CIC_DTYPE_F32 (0): the reference's arithmetic.  CIC_DTYPE_BF16: the reduced-precision variant
                                (BASELINE configs[1] "bf16") - the batched products run on bf16 operands with f32
                                accumulation (CIC_PRECISION_BF16), the embedded regions `att` and their projection `p_att`
                                are rounded to bf16 and the per-timestep attention streams them as bf16 (75,848 instead of
                                151,696 bytes per image and step); recurrent state, per-timestep products, softmax,
                                log-softmax, losses and the optimiser stay f32 */
} cic_speaker_dims;
enum { CIC_DTYPE_F32 = 0, CIC_DTYPE_BF16 = 1 };

/* Parameter (or gradient) pointers, named as the reference's state dict
 * (models/AttModel.py:74-88,462-463,503-505). */
typedef struct {
    float* embed_w;                       /* embed.0.weight                 [V+2, E] */
    float *att_embed_w, *att_embed_b;     /* att_embed.0                    [H, D], [H] */
    float *logit_w, *logit_b;             /* logit                          [V+1, H], [V+1] */
    float *ctx2att_w, *ctx2att_b;         /* ctx2att                        [A, H], [A] */
    float *a2c_w, *a2c_b;                 /* core.a2c                       [2H, H], [2H] */
    float *i2h_w, *i2h_b;                 /* core.i2h                       [5H, E], [5H] */
    float *h2h_w, *h2h_b;                 /* core.h2h                       [5H, H], [5H] */
    float *h2att_w, *h2att_b;             /* core.attention.h2att           [A, H], [A] */
    float *alpha_w, *alpha_b;             /* core.attention.alpha_net       [1, A], [1] */
} cic_speaker_params;

/* att_pre = relu(att_embed(att_raw)) [B*K, H] — AttModel.py:82-85,315 before the dropout.
 * It is shared by the decodes of one training step (their dropout masks differ). */
int cic_speaker_att_embed_fwd(const cic_speaker_dims* d, const cic_speaker_params* p,
                              const float* att_raw, float* att_pre, cic_stream_t s);

/* use_bn = 1: BatchNorm1d(att_feat_size) in front of att_embed's Linear (AttModel.py:82-85), applied by pack_wrapper (:44-51) to the
 * VALID region rows of a batch with ragged region counts (masks[r] > 0 marks them; rows = B*K).  The normalisation is folded into
 * the Linear, whose product then reads the raw features (csrc/batchnorm.hip):
 *   cic_bn_stats          per feature: mean and biased variance over the valid rows, and their count N           (training mode)
 *   cic_bn_running_update running_mean / running_var <- (1 - momentum) old + momentum batch (unbiased variance), one forward
 *   cic_bn_fold_fwd       W' = W diag(a), bias' = bias + W b with a = gamma / sqrt(var + eps), b = beta - mean a  -> cic_speaker_att_embed_fwd
 *   cic_bn_fold_bwd       from the Linear's RAW gradients (dW_raw = d_pre^T x, db_raw: what cic_speaker_decode_bwd accumulates into
 *                         grads->att_embed_w / _b when handed zeroed scratch): dW += gamma G + beta (x) db, dbias += db,
 *                         dgamma += colsum(W . G), dbeta += W^T db, G = (dW_raw - mean (x) db) / sqrt(var + eps)                    */
int cic_bn_stats(const float* x, const float* masks, int rows, int D, float* mean, float* var, float* count, cic_stream_t s);
int cic_bn_running_update(const float* mean, const float* var, const float* count, float momentum, int D, float* running_mean,
                          float* running_var, cic_stream_t s);
int cic_bn_fold_fwd(const float* W, const float* bias, const float* gamma, const float* beta, const float* mean, const float* var,
                    float eps, int H, int D, float* W_folded, float* bias_folded, cic_stream_t s);
int cic_bn_fold_bwd(const float* dW_raw, const float* db_raw, const float* W, const float* gamma, const float* beta,
                    const float* mean, const float* var, float eps, int H, int D, float* dW, float* dbias, float* dgamma,
                    float* dbeta, cic_stream_t s);

typedef struct {
    int mode;                 /* CIC_SAMPLE_* (not NONE) */
    float temp;               /* temperature / gumbel_temp / multinomial_temp */
    int decoding_constraint;
    const float* att_pre;     /* [B,K,H] from cic_speaker_att_embed_fwd */
    const float* att_masks;   /* [B,K] or NULL */
    const uint8_t* att_keep;  /* [B,K,H]   dropout keep masks; NULL = no dropout at that site */
    const uint8_t* x_keep;    /* [T+1,B,E] row t: token-embedding dropout of core step t */
    const uint8_t* out_keep;  /* [T+1,B,H] row t: core-output dropout of step t */
    const float* U;           /* [T+1,B,V+1] row t (t>=1): Gumbel uniforms used to pick the input of step t */
    const int64_t* pick;      /* [T+1,B] row t: externally chosen tokens (multinomial modes) or NULL.
                                 Teacher forcing (AttModel.forward, :103-148) = CIC_SAMPLE_TEACHER with
                                 pick[t] = labels[:, t]: slp then holds log p(target) of every step. */
    /* partial-sampling modes (CIC_SAMPLE_*_PS): the recurrent input of step t>=1 is relu_dropout(soft @ embed)
     * (AttModel.py:395-397) and the caption handed to the listener is the soft row */
    const float* ps_u;        /* [T+1,B] row uniforms (row t: the draw made when choosing the input of step t) */
    float ps_prob;
    float* soft_raw;          /* scratch/out [T,B,V+1]: un-masked soft rows (saved for the backward pass) */
    float* xpre;              /* scratch/out [T,B,E]: soft @ embed before ReLU/dropout (saved for the backward pass) */
    float* soft_out;          /* out [T,B,V+1]: soft rows with finished rows replaced by the EOS one-hot (:428-432) */
    const float* ss_u;        /* [T+1,B] scheduled-sampling uniforms, row t decides the input of step t (teacher mode) */
    float ss_prob;            /* model.ss_prob (train.py:80-85); 0 = plain teacher forcing */
    const int64_t* ss_pick;   /* [T+1,B] externally drawn scheduled-sampling tokens or NULL */
    /* FCModel (models/FCModel.py, the fc-feature speaker of the reference's CPU configuration): fc_mode != 0 with
     * dims.K == 0.  The decode is preceded by the image step (h0, c0) = LSTMCore(x0, 0) (:97-99,274-276), token
     * embeddings are plain rows (no ReLU / dropout, :66,119,302), there is no attention, and the recurrent state is
     * the DROPPED-OUT h (:38-42).  out_keep then has T+2 rows: row 0 the image step, row t+1 core step t.
     * Parameters: embed_w, i2h, h2h, logit; the attention / att_embed / ctx2att / a2c pointers are unused. */
    int fc_mode;
    const float* x0;          /* [B,E] img_embed(fc_feats) (a cic_gemm_f32 call of the caller) */
    const int64_t* first_token; /* [B] input token of step 0; NULL = <bos> = vocab_size+1 (:324-326).
                                 AttModel.forward starts from labels[:, 0] = 0 instead (:131).  When set
                                 (teacher forcing) all T steps count: L is written as T. */
    int32_t* seq;             /* out [B,T]  it * unfinished          (AttModel.py:409-415) */
    float* slp;               /* out [B,T]  sampled log-probs        (AttModel.py:413,423) */
    float* stv;               /* out [B,T]  straight-through values, or NULL */
    int32_t* L;               /* out [1]    number of columns the reference would return */
    /* U == NULL and u_philox != 0: the [T+1,B,V+1] uniforms are not materialised; element i of that slab is element i
     * of the stream cic_uniform_f32(out, n, u_seed, u_offset) would write (counter u_offset + i/4, lane i%4), drawn
     * inside the kernels that consume it (forward and backward).  Not for the partial-sampling modes. */
    int u_philox;
    uint64_t u_seed, u_offset;
    cic_timer* timer;         /* optional in-situ timing of this decode's kernels (forward and backward), or NULL */
    int device_shared;        /* != 0: other processes run kernels on this device at the same time.  The teacher-forced
                                 recurrence is then launched step by step (its one-launch form needs all its workgroups
                                 resident together) */
    uint32_t* status;         /* the caller's sticky status word on the device ("status word" above) or NULL: the one-launch
                                 loops of this decode - forward AND backward - OR their bit into it when a hand-off times out */
} cic_decode_io;

/* Bytes of workspace a decode needs; the same workspace must be handed, untouched, to
 * cic_speaker_decode_bwd.  Zero-fill a workspace ONCE when it is allocated (it may then be reused call after call): a
 * sampling decode whose captions have all ended (AttModel.py:401-408) returns from the heavy kernels of its remaining
 * steps at once and leaves their slabs as they were; the backward pass multiplies those by zero gradients, so they must
 * hold finite values.  Everything a caller can observe - tokens, log-probs, L, gradients - is that of the full loop. */
size_t cic_speaker_decode_ws_bytes(const cic_speaker_dims* d);
/* AttModel.sample (beam_size 1), models/AttModel.py:291-452: T core steps + samplers, no host
 * sync (the reference's early break is replaced by the device-side length L). */
int cic_speaker_decode_fwd(const cic_speaker_dims* d, const cic_speaker_params* p, const cic_decode_io* io,
                           void* ws, size_t ws_bytes, cic_stream_t s);

/* Two decodes of the same images and parameters (same dims, e.g. the sampled and the greedy decode of a
 * joint step, AlternatingJointModel.py:346,391-403) advanced in lock step: every per-timestep kernel runs once
 * over 2B rows.  Bit-identical to two cic_speaker_decode_fwd calls (which is also the fallback for shapes the
 * paired kernels do not cover); each decode keeps its own workspace for its backward pass. */
int cic_speaker_decode_fwd_pair(const cic_speaker_dims* d, const cic_speaker_params* p,
                                const cic_decode_io* io_a, void* ws_a, size_t ws_a_bytes,
                                const cic_decode_io* io_b, void* ws_b, size_t ws_b_bytes, cic_stream_t s);
/* 1 if cic_speaker_decode_fwd_pair runs these two decodes through shared launches, 0 if it runs them one after the other
 * (B not a multiple of 32 or above 128, partial-sampling modes, the fc speaker, widths outside the paired kernels) */
int cic_speaker_decode_pair_fused(const cic_speaker_dims* d, const cic_decode_io* io_a, const cic_decode_io* io_b);

typedef struct {
    const float* d_onehot;   /* [T,B,V+1] gradient w.r.t. the ST one-hot rows / the soft rows io->soft_out (from
                                cic_listener_bwd) or NULL */
    const float* dslp;       /* [B,T] gradient w.r.t. the sampled log-probs (io->slp) or NULL */
    const cic_speaker_params* grads; /* accumulated into (+=); grads->embed_w may be NULL (a frozen embedding table) */
    const float* att_raw;    /* [B,K,D] the raw region features (for the att_embed weight gradient); NULL in fc_mode */
    float* d_x0;             /* fc_mode: out [B,E] gradient w.r.t. io->x0 (the caller back-propagates img_embed) */
    /* 0: the whole backward pass.  Data-parallel callers split it in two calls on the same arguments so that the
     * logit layer's gradient (19.4 MB of the speaker's 57.8 MB, final before the time loop starts) can travel
     * under the BPTT loop:  CIC_BWD_LOGIT = d logits, d out and grads->logit_w / logit_b only;  CIC_BWD_REST =
     * everything after that (reads the d out the first call left in ws_bwd); CIC_BWD_REST itself in two calls:
     * CIC_BWD_LOOP = the BPTT loop only (one launch at the flagship widths: it fills every CU, so a collective does not run
     * beside it - the host lets pending exchanges land before it and starts the logit bucket after it), CIC_BWD_TAIL = the
     * batched weight gradients after the loop.  Partial-sampling decodes make their d logits inside the time loop: phase 0
     * only; LOOP / TAIL not for the fc speaker. */
    int phase;
    int device_shared;        /* != 0: other processes run kernels on this device at the same time.  The BPTT loop is then
                                 launched step by step (its one-launch form needs all its workgroups resident together) */
    const float* dslp_scale;  /* [1] on the device or NULL: dslp is multiplied by it (the upstream gradient of the scalar loss:
                                 loss.backward() hands it over as a device scalar; no elementwise launch for dslp * go).  Not with
                                 partial-sampling decodes */
} cic_decode_bwd_io;
enum { CIC_BWD_ALL = 0, CIC_BWD_LOGIT = 1, CIC_BWD_REST = 2, CIC_BWD_LOOP = 3, CIC_BWD_TAIL = 4 };
size_t cic_speaker_decode_bwd_ws_bytes(const cic_speaker_dims* d);
/* autograd of cic_speaker_decode_fwd: straight-through sampler, logit layer, BPTT through
 * Att2in2Core/Attention, embeddings, ctx2att, att_embed.  io must be the struct of the
 * forward call (same noise pointers), ws_fwd its untouched workspace.
 * Partial-sampling decodes (CIC_SAMPLE_*_PS): d_onehot is the gradient w.r.t. io->soft_out; the soft row
 * also feeds the next step's input (AttModel.py:395-397), so the sampler and logit-layer backward run
 * inside the time loop instead of batched over time. */
int cic_speaker_decode_bwd(const cic_speaker_dims* d, const cic_speaker_params* p, const cic_decode_io* io,
                           const cic_decode_bwd_io* bio, void* ws_fwd, size_t ws_fwd_bytes, void* ws_bwd,
                           size_t ws_bwd_bytes, cic_stream_t s);

/* ---- beam-search decode (evaluation): AttModel.sample_beam, models/AttModel.py:150-289 ------------------
 * All B x beam rows advance together; beam merge, state re-ordering and done-beam bookkeeping run on the device
 * (the reference decodes image by image and merges on the host).  Evaluation mode: no dropout.  Returns, per
 * image, the recorded beam the reference returns (see csrc/beam.hip for the reference's scoring quirk). */
typedef struct {
    int beam;                 /* beam_size, 1..16 and <= V+1 */
    int decoding_constraint;
    const float* att_pre;     /* [B,K,H] from cic_speaker_att_embed_fwd */
    const float* att_masks;   /* [B,K] or NULL */
    int32_t* seq;             /* out [B,T] tokens of the returned beam, 0 after its end */
    float* logps;             /* out [B,T] their log-probs, 0 after the end */
    float* score;             /* out [B]  done_beams[k][0]['p'] */
} cic_beam_io;
size_t cic_speaker_beam_ws_bytes(const cic_speaker_dims* d, int beam);
int cic_speaker_beam_search(const cic_speaker_dims* d, const cic_speaker_params* p, const cic_beam_io* io, void* ws,
                            size_t ws_bytes, cic_stream_t s);

/* ---- listener (VSE-fc) engines: models/VSEFCModel.py:12-241 ------------------------------ */
typedef struct {
    int B, F, E, J, V;   /* batch, fc_feat_size, input_encoding_size, vse_embed_size, vocab_size */
    int T;               /* seq_length of generated captions (Lp = T+1 for them) */
    int Lp;              /* token positions per caption: T+1 (generated) or labels.shape[1] */
    float margin;        /* vse_margin */
    int max_violation;   /* vse_max_violation */
    int no_imgnorm;      /* vse_no_imgnorm */
    int use_abs;         /* vse_use_abs */
    int pool;            /* vse_pool_type: 0 'last' (the scripts' setting), 1 'mean', 2 'max' (VSEFCModel.py:118-129) */
    int compute_dtype;   /* (r4) CIC_DTYPE_F32 (0) / CIC_DTYPE_BF16: as cic_speaker_dims.compute_dtype - the reduced-precision variant
                            runs the GRU pass and its BPTT loop on bf16 MFMA fragments and the batched products of the text encoder on
                            one bf16 part, f32 accumulation; image encoder, similarities, contrastive loss and all stored values f32 */
} cic_listener_dims;

typedef struct {
    float *img_fc_w, *img_fc_b;   /* img_enc.fc                 [J, F], [J] */
    float* embed_w;               /* txt_enc.embed.weight       [V+2, E] */
    float *w_ih, *w_hh;           /* txt_enc.rnn.weight_{ih,hh}_l0  [3J, E], [3J, J] */
    float *b_ih, *b_hh;           /* txt_enc.rnn.bias_{ih,hh}_l0    [3J] */
} cic_listener_params;

typedef struct {
    const float* fc_feats;    /* [B, F] */
    /* caption source A: ground-truth indices (VSEFCModel.py:106) */
    const int64_t* labels;    /* [B, Lp] or NULL */
    const float* masks;       /* [B, Lp] */
    /* caption source B: a decode's output (AlternatingJointModel.py:353-371) */
    const int32_t* seq;       /* [B, T] */
    const float* stv;         /* [B, T] straight-through values or NULL (plain indices) */
    const int32_t* L;         /* [1] */
    const float* soft;        /* [T,B,V+1] soft caption rows (partial sampling) or NULL: positions 1..T are then embedded
                                 by the dense product soft @ embed (VSEFCModel.py:102-104) instead of the gather */
    int only_one_retrieval;   /* 0 off, 1 'image', 2 'caption' (VSEFCModel.py:202-207) */
    float* loss_rows;         /* out [B]: per-row loss (whole_batch=True) */
    float* loss_sum;          /* out [1]: scalar loss (whole_batch=False) */
    float* img_emb_out;       /* out [B, J] or NULL */
    float* cap_emb_out;       /* out [B, J] or NULL */
    int device_shared;        /* != 0: other processes run kernels on this device at the same time.  The GRU pass is then
                                 launched step by step; 0 lets it run as ONE launch whose workgroups hand the hidden state
                                 to each other inside the launch, which needs all of them resident together (a second
                                 process's kernels could hold the CUs some of them wait for: the pass would then give up
                                 after its time budget and mark its outputs NaN rather than hang) */
    uint32_t* status;         /* the caller's sticky status word on the device ("status word" above) or NULL: the GRU pass and
                                 its BPTT loop (cic_listener_bwd reads this struct too) OR their bit into it on a time-out */
} cic_listener_io;

typedef struct {
    const float* g_rows;          /* [B] upstream gradient per row, or NULL */
    const float* g_scalar;        /* [1] upstream gradient of the scalar loss (used if g_rows is NULL) */
    const cic_listener_params* grads; /* accumulated into (+=); NULL = no parameter gradients */
    float* d_onehot;              /* out [T, B, V+1] gradient w.r.t. the one-hot rows of the generated
                                     tokens (time-major), or NULL */
    float g_scale;                /* factor on g_rows / g_scalar (a loss weight); 0 counts as 1 */
} cic_listener_bwd_io;

size_t cic_listener_ws_bytes(const cic_listener_dims* d);
/* VSEFCModel.forward (:230-241): EncoderImage, EncoderText (GRU, last-valid state), ContrastiveLoss. */
int cic_listener_fwd(const cic_listener_dims* d, const cic_listener_params* p, const cic_listener_io* io,
                     void* ws, size_t ws_bytes, cic_stream_t s);
/* autograd of the above; ws must be the workspace of the matching forward call. */
int cic_listener_bwd(const cic_listener_dims* d, const cic_listener_params* p, const cic_listener_io* io,
                     const cic_listener_bwd_io* bio, void* ws, size_t ws_bytes, cic_stream_t s);

/* ---- retrieval-rank evaluation of the listener: eval_utils.i2t / t2i (eval_utils.py:545-720), cosine measure ----
 * ims [n_images, J]: one embedding per image; caps [n_images*cpi, J]: its cpi captions in a row (5 ground-truth
 * captions, or 1 generated caption).  One similarity GEMM + rank kernels; a rank is the position in
 * np.argsort(d)[::-1], i.e. the number of candidates scoring higher (ties: the larger index first).
 *   ranks_i2t/top1_i2t [n_images]      best rank among the image's own captions, arg-max caption   (may be NULL)
 *   ranks_t2i/top1_t2i [n_images*cpi]  rank of the caption's own image, arg-max image              (may be NULL) */
size_t cic_retrieval_ws_bytes(int n_images, int cpi);
int cic_retrieval_ranks(const float* ims, const float* caps, int n_images, int cpi, int J, int32_t* ranks_i2t,
                        int32_t* top1_i2t, int32_t* ranks_t2i, int32_t* top1_t2i, void* ws, size_t ws_bytes,
                        cic_stream_t s);

/* ---- loss assembly and optimiser ------------------------------------------------------- */
/* loss = sum_{b,t<L} slp[b,t] * (coef_sign*coef[b]) * m[b,t] / sum m with m = gen_masks[:, 1:]
 * (m[b,0] = 1, m[b,t] = seq[b,t-1] > 0): the self-critical CIDEr term (coef = reward, sign -1;
 * AlternatingJointModel.py:421-428) and the REINFORCE term (coef = retrieval_loss - baseline, sign +1;
 * :292-297,:321-325).  dslp (+)= weight * d loss / d slp.  loss_out / dslp may be NULL. */
int cic_seq_loss(const float* slp, const int32_t* seq, const int32_t* L, const float* coef, float coef_sign,
                 float weight, int B, int T, float* loss_out, float* dslp, int accumulate, cic_stream_t s);
/* cic_seq_loss that also writes the STEP's loss when its term is the last one of the sum (AlternatingJointModel.py:470-503):
 * total[0] = sum_{i < count} term_weight[i] * term[i][0] + self_weight * (this term), added in that order - cic_loss_combine
 * over (term..., this term) without a launch of its own.  term / term_weight are host arrays of `count` < CIC_LOSS_MAX_TERMS
 * entries (device scalars / floats); total == NULL: cic_seq_loss. */
int cic_seq_loss_total(const float* slp, const int32_t* seq, const int32_t* L, const float* coef, float coef_sign,
                       float weight, int B, int T, float* loss_out, float* dslp, int accumulate,
                       const float* const* term, const float* term_weight, int count, float self_weight, float* total,
                       cic_stream_t s);
/* LanguageModelCriterion, misc/utils.py:49-58: loss = -sum slp*mask / sum mask (slp = log p(target));
 * dslp = weight * d loss / d slp. */
int cic_masked_nll(const float* slp, const float* mask, int mask_ld, float weight, int B, int T,
                   float* loss_out, float* dslp, cic_stream_t s);
/* The step's loss as the reference assembles it (AlternatingJointModel.py:470-503: loss = sum_i weight_i * term_i):
 * total[0] = sum over the `count` <= CIC_LOSS_MAX_TERMS device scalars term[i][0] of weight[i] * term[i][0], added in
 * index order.  `term` and `weight` are host arrays. */
#define CIC_LOSS_MAX_TERMS 8
int cic_loss_combine(const float* const* term, const float* weight, int count, float* total, cic_stream_t s);
/* clip_gradient (elementwise clamp to +-grad_clip, misc/utils.py:65-69) followed by one
 * torch.optim.Adam step (optimizer.py:25-27,233-242) over a flat buffer of n floats.  The gradient is
 * first multiplied by grad_scale (1/world_size after a sum all-reduce).  step >= 1 is Adam's t. */
int cic_clamp_adam(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                   double beta2, double eps, double weight_decay, double grad_clip, int step,
                   double grad_scale, cic_stream_t s);
/* The clamp propagates NaN as torch's clamp_ does (misc/utils.py:65-69: a NaN gradient stays NaN, and Adam then writes NaN
 * into the parameter - the reference's behaviour; fminf / fmaxf alone would turn it into -grad_clip).
 * The same update that also CLEARS the gradient buffer while each element is in registers (zero_grad != 0): the next
 * step's zeroing_optimizer (optimizer.py:224-230) then has nothing left to do.  zero_grad == 0: cic_clamp_adam. */
int cic_clamp_adam_zero(float* p, float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                        double eps, double weight_decay, double grad_clip, int step, double grad_scale, int zero_grad,
                        cic_stream_t s);
/* cic_clamp_adam_zero behind the status word ("status word" above): when *status != 0 at launch time the kernel touches
 * nothing - p, m, v AND g stay as they are (the poisoned gradient remains as evidence) - and sets CIC_STATUS_UPDATE_SKIPPED.
 * status == NULL: cic_clamp_adam_zero. */
int cic_clamp_adam_guarded(float* p, float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                           double eps, double weight_decay, double grad_clip, int step, double grad_scale, int zero_grad,
                           uint32_t* status, cic_stream_t s);

/* ---- self-critical CIDEr-D reward: misc/rewards.py:26-72, ciderD_scorer.py:13-215 ---------- */
typedef struct {
    int B, T;                 /* hypotheses per decode and their columns (T <= 16) */
    int n_images, spi;        /* len(data['gts']) and seq_per_img; B == n_images * spi */
    int R, Tr;                /* reference sentences in total and their columns (Tr <= 16) */
    const int32_t* gen;       /* [B,T] sampled captions (it * unfinished) */
    const int32_t* L_gen;     /* [1] columns the reference would have returned for them */
    const int32_t* greedy;    /* [B,T] greedy captions */
    const int32_t* L_greedy;  /* [1] */
    const int32_t* refs;      /* [R,Tr] ground-truth captions, image after image */
    const int32_t* ref_off;   /* [n_images+1] first reference row of each image */
    double* scores;           /* out [2B] CIDEr-D of sampled then greedy captions (x10, as the reference) */
    float* reward;            /* out [B] scores[:B] - scores[B:] (rewards.py:66) or NULL */
    double* stats;            /* out [2] mean sampled score, mean greedy score, or NULL */
    /* optional dumps of the exact integer tables (tests): sentence-major, 64 slots each */
    uint64_t* dbg_keys; int32_t* dbg_cnt; int32_t* dbg_df; int32_t* dbg_nuniq;
    /* REQUIRED: the vocabulary size V of the captions (token ids 0 .. V+1).  n-gram keys pack 15 bits per token, so
     * the call is refused unless 1 <= V <= 32766; a token outside [0, V+1] met on the device turns every score into
     * NaN (the reference's string n-grams have no such limit: refusing beats aliasing two words silently). */
    int vocab_size;
    int max_refs_per_image;   /* the largest reference count of an image if the caller knows it (it packed ref_off), else 0.  Up to 16
                                 the document frequencies are counted inside the n-gram launch; 0 or more: one more launch */
} cic_ciderd_args;
size_t cic_ciderd_ws_bytes(int B, int R);
int cic_ciderd_reward(const cic_ciderd_args* a, void* ws, size_t ws_bytes, cic_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* CIC_H */
