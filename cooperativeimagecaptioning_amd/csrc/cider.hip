// Self-critical CIDEr-D reward on the GPU (integer tokens in, fp64 scores out, no gradient).
// Reference: misc/rewards.py:26-72 + cider/pyciderevalcap/ciderD/ciderD_scorer.py:13-215
// (df mode "corpus", n = 4, sigma = 6), which runs as Python dicts of string n-grams on the host.
//
// MI355X design: a sentence has <= 16 tokens, hence <= 58 n-grams (16+15+14+13): exactly one
// 64-lane wavefront per sentence, one n-gram per lane.
//   1. ngram   : lane -> 64-bit key (order | 4 x 15-bit tokens: the caller declares its vocabulary size, which must
//                fit 15 bits, and a token outside [0, vocab_size + 1] turns every score into NaN instead of aliasing
//                silently); wave-wide bitonic sort with
//                shuffles; run-length -> unique keys + integer term frequencies   (precook :13-30)
//   2. df      : reference n-grams go into an open-addressing hash table (atomicCAS on the key,
//                atomicAdd on the count), once per image (binary search in the image's earlier
//                references removes duplicates)                         (compute_doc_freq :106-118)
//   3. vec     : tf-idf weights and per-order norms in fp64                     (counts2vec :121-146)
//   4. score   : one wave per hypothesis; each lane binary-searches its key in every reference
//                of the image; clipped cosine, Gaussian length penalty          (sim :148-175, :184-201)
// Integer quantities (n-gram counts, document frequencies, lengths) are exact.
#include "cic_common.h"
#include "engine_util.h"

namespace {

constexpr uint64_t EMPTY = 0xFFFFFFFFFFFFFFFFull;
constexpr int NMAX = 4;
constexpr int MAXTOK = 16;

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, m, 64);
    hi = __shfl_xor(hi, m, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl(lo, src, 64);
    hi = __shfl(hi, src, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint64_t u = (uint64_t)__double_as_longlong(v);
        u = shfl_xor_u64(u, o);
        v += __longlong_as_double((long long)u);
    }
    return v;
}
__device__ __forceinline__ uint64_t hash64(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}

// sentence s: [0,B) sampled rows, [B,2B) greedy rows, [2B,2B+R) reference rows
struct Sents {
    const int32_t *gen, *greedy, *refs;
    const int32_t *L_gen, *L_greedy;
    int B, T, R, Tr;
};
__device__ __forceinline__ const int32_t* sent_row(const Sents& s, int i, int& cols) {
    if (i < s.B) { cols = min(*s.L_gen, s.T); return s.gen + (size_t)i * s.T; }
    if (i < 2 * s.B) { cols = min(*s.L_greedy, s.T); return s.greedy + (size_t)(i - s.B) * s.T; }
    cols = s.Tr;
    return s.refs + (size_t)(i - 2 * s.B) * s.Tr;
}

// 1. one wave per sentence sid: its sorted unique n-gram keys and term frequencies -> keys / cnt / nuniq / blen (and the unique
//    keys into `lds_row` when given); returns the number of unique keys
__device__ __forceinline__ int sentence_ngrams(const Sents& s, int sid, int max_token, uint64_t* __restrict__ keys,
                                               int32_t* __restrict__ cnt, int32_t* __restrict__ nuniq,
                                               int32_t* __restrict__ blen, int32_t* __restrict__ bad, uint64_t* lds_row) {
    const int lane = threadIdx.x & 63;
    int cols;
    const int32_t* row = sent_row(s, sid, cols);
    // array_to_str (rewards.py:26-32): tokens up to AND INCLUDING the first 0
    int len = cols;
    for (int j = 0; j < cols; ++j)
        if (row[j] == 0) { len = j + 1; break; }
    // a token id that does not fit the declared vocabulary would alias another n-gram key: poison the result instead
    if (lane < len && (row[lane] < 0 || row[lane] > max_token)) atomicOr(bad, 1);
    // lane -> (order n, position p): unigrams 0..15, bigrams 16..30, trigrams 31..44, 4-grams 45..57
    int n, p;
    if (lane < 16) { n = 1; p = lane; }
    else if (lane < 31) { n = 2; p = lane - 16; }
    else if (lane < 45) { n = 3; p = lane - 31; }
    else if (lane < 58) { n = 4; p = lane - 45; }
    else { n = 5; p = 0; }
    uint64_t key = EMPTY;
    if (n <= NMAX && p + n <= len) {
        key = (uint64_t)n << 60;
        for (int j = 0; j < n; ++j) key |= (uint64_t)(row[p + j] & 0x7fff) << (45 - 15 * j);
    }
    // bitonic sort across the 64 lanes
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const uint64_t other = shfl_xor_u64(key, j);
            const bool up = (lane & k) == 0, lower = (lane & j) == 0;
            const uint64_t mn = key < other ? key : other, mxv = key < other ? other : key;
            key = (lower == up) ? mn : mxv;
        }
    const uint64_t prev = shfl_u64(key, lane == 0 ? 0 : lane - 1);
    const bool valid = key != EMPTY;
    const bool head = valid && (lane == 0 || key != prev);
    const unsigned long long heads = __ballot(head);
    const unsigned long long valids = __ballot(valid);
    const int nvalid = __popcll(valids);
    if (head) {
        const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1)) << (lane + 1);
        const int next = above ? __ffsll((long long)above) - 1 : nvalid;
        const int rank = __popcll(heads & ((1ull << lane) - 1ull));
        keys[(size_t)sid * 64 + rank] = key;
        cnt[(size_t)sid * 64 + rank] = next - lane;          // term frequency (exact integer)
        if (lds_row) lds_row[rank] = key;
    }
    const int nu = __popcll(heads);
    if (lane == 0) {
        nuniq[sid] = nu;
        blen[sid] = len >= 2 ? len - 1 : 0;                   // "length" = number of bigrams (:143-144)
    }
    return nu;
}

__device__ __forceinline__ int bsearch_lds(const uint64_t* a, int n, uint64_t k) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const uint64_t v = a[mid];
        if (v == k) return mid;
        if (v < k) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

// (r4) ONE launch for all tables of the step (was: ngram_kernel, then df_kernel).
//   workgroups [0, ceil(2B / 4)): one wave per hypothesis sentence (sampled rows, then greedy rows);
//   workgroups beyond: one per IMAGE - its reference sentences' n-grams (wave w takes references w, w + 4, ...; their unique keys
//   also stay in LDS), a barrier, then the document-frequency step of those references: a key counts once per image (binary
//   search in the image's EARLIER references, now in LDS), open-addressing insert (atomicCAS on the key, atomicAdd on the count).
// An image with more than DF_MAX_REFS references keeps its keys in memory only: df_kernel then runs as a launch of its own.
constexpr int DF_MAX_REFS = 16;
__global__ __launch_bounds__(256) void tables_kernel(Sents s, int hyp_wgs, int max_token, uint64_t* __restrict__ keys,
                                                     int32_t* __restrict__ cnt, int32_t* __restrict__ nuniq,
                                                     int32_t* __restrict__ blen, int32_t* __restrict__ bad,
                                                     const int32_t* __restrict__ ref_off, int weight,
                                                     uint64_t* __restrict__ ht_keys, int32_t* __restrict__ ht_df, uint32_t ht_mask) {
    __shared__ uint64_t keys_s[DF_MAX_REFS][64];
    __shared__ int nu_s[DF_MAX_REFS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if ((int)blockIdx.x < hyp_wgs) {
        const int sid = blockIdx.x * 4 + w;
        if (sid < 2 * s.B) sentence_ngrams(s, sid, max_token, keys, cnt, nuniq, blen, bad, nullptr);
        return;
    }
    const int img = blockIdx.x - hyp_wgs;
    const int r0 = ref_off[img], nr = ref_off[img + 1] - r0;
    const bool in_lds = nr <= DF_MAX_REFS;
    for (int q = w; q < nr; q += 4) {
        const int nu = sentence_ngrams(s, 2 * s.B + r0 + q, max_token, keys, cnt, nuniq, blen, bad, in_lds ? keys_s[q] : nullptr);
        if (in_lds && lane == 0) nu_s[q] = nu;
    }
    if (!in_lds) return;                                    // (workgroup-uniform) df_kernel follows
    __syncthreads();
    for (int q = w; q < nr; q += 4) {
        if (lane >= nu_s[q]) continue;
        const uint64_t key = keys_s[q][lane];
        bool seen = false;
        for (int e = 0; e < q && !seen; ++e) seen = bsearch_lds(keys_s[e], nu_s[e], key) >= 0;
        if (seen) continue;
        uint32_t slot = (uint32_t)hash64(key) & ht_mask;
        for (;;) {
            const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(ht_keys + slot),
                                                     (unsigned long long)EMPTY, (unsigned long long)key);
            if (old == EMPTY || old == key) { atomicAdd(ht_df + slot, weight); break; }
            slot = (slot + 1) & ht_mask;
        }
    }
}

__device__ __forceinline__ int bsearch_key(const uint64_t* __restrict__ a, int n, uint64_t k) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const uint64_t v = a[mid];
        if (v == k) return mid;
        if (v < k) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

// NQ independent binary searches for the same key advanced in lockstep: the loads of one round do not depend on each other,
// so a round is one memory round trip for all NQ arrays (a loop of bsearch_key calls is NQ x log2(n) dependent trips).
// pos[q] = index of k in a[q][0..n[q]) or -1; arrays with n[q] <= 0 are skipped.
template <int NQ>
__device__ __forceinline__ void bsearch_keys(const uint64_t* const (&a)[NQ], const int (&n)[NQ], uint64_t k, int (&pos)[NQ]) {
    int lo[NQ], hi[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { lo[q] = 0; hi[q] = n[q] - 1; pos[q] = -1; }
#pragma unroll 1
    for (int round = 0; round < 7; ++round) {                  // rows hold at most 64 keys: 7 probes decide
        uint64_t v[NQ];
        int mid[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            mid[q] = (lo[q] + hi[q]) >> 1;
            v[q] = (lo[q] <= hi[q]) ? a[q][mid[q]] : 0ull;
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (lo[q] <= hi[q]) {
                if (v[q] == k) { pos[q] = mid[q]; lo[q] = 1; hi[q] = 0; }
                else if (v[q] < k) lo[q] = mid[q] + 1;
                else hi[q] = mid[q] - 1;
            }
    }
}

// 2. document frequency: one wave per reference sentence
__global__ __launch_bounds__(256) void df_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ nuniq,
                                                 const int32_t* __restrict__ ref_img, const int32_t* __restrict__ ref_off,
                                                 int B2, int R, int weight, uint64_t* __restrict__ ht_keys,
                                                 int32_t* __restrict__ ht_df, uint32_t ht_mask) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int sid = B2 + r;
    if (lane >= nuniq[sid]) return;
    const uint64_t key = keys[(size_t)sid * 64 + lane];
    const int img = ref_img[r];
    if (ref_off[img + 1] - ref_off[img] <= DF_MAX_REFS) return;   // counted by tables_kernel
    for (int q0 = ref_off[img]; q0 < r; q0 += 4) {             // already counted by an earlier reference (four probed at a time)
        const uint64_t* rows[4];
        int nr[4], pos[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + j < r ? q0 + j : q0;
            rows[j] = keys + (size_t)(B2 + q) * 64;
            nr[j] = q0 + j < r ? nuniq[B2 + q] : 0;
        }
        bsearch_keys<4>(rows, nr, key, pos);
        if ((pos[0] & pos[1] & pos[2] & pos[3]) >= 0) return;   // some pos >= 0 (all four are -1 otherwise)
    }
    uint32_t slot = (uint32_t)hash64(key) & ht_mask;
    for (;;) {
        const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(ht_keys + slot),
                                                 (unsigned long long)EMPTY, (unsigned long long)key);
        if (old == EMPTY || old == key) {
            atomicAdd(ht_df + slot, weight);
            return;
        }
        slot = (slot + 1) & ht_mask;
    }
}

__device__ __forceinline__ int ht_lookup(const uint64_t* __restrict__ ht_keys, const int32_t* __restrict__ ht_df,
                                         uint32_t ht_mask, uint64_t key) {
    uint32_t slot = (uint32_t)hash64(key) & ht_mask;
    for (;;) {
        const uint64_t k = ht_keys[slot];
        if (k == key) return ht_df[slot];
        if (k == EMPTY) return 0;
        slot = (slot + 1) & ht_mask;
    }
}

// 4. (r4: with the tf-idf vectors and the reward in the same launch - was vec_kernel, score_kernel, reward_kernel)
//    One workgroup of SCORE_W waves per hypothesis h.  Every wave makes the hypothesis' tf-idf vector and norms for itself (one
//    hash lookup per lane: cheaper than a hand-over), wave w makes those of reference r0 + w, r0 + w + SCORE_W, ... in LDS and
//    scores it (a binary search, an f64 exp, four f64 wave sums and divisions), wave 0 adds the per-reference terms in reference
//    order.  The f64 operations and their order are those of vec_kernel + score_kernel, so the scores are bit-identical.
//    The workgroup whose score lands last (an arrival counter; scores stored write-through, read back with atomic loads) turns
//    the 2B scores into the reward and the two means exactly as reward_kernel's 256 threads did.
constexpr int SCORE_W = 5;                                     // COCO: five references per image
struct TfIdf { double v; int n; };
__device__ __forceinline__ TfIdf tfidf_of(uint64_t key, int count, bool on, double ref_len, const uint64_t* __restrict__ ht_keys,
                                          const int32_t* __restrict__ ht_df, uint32_t ht_mask, int* df_dbg) {
    TfIdf t = {0.0, 0};
    if (on) {
        const int df = ht_lookup(ht_keys, ht_df, ht_mask, key);
        t.n = (int)(key >> 60);
        const double d = log(df > 1 ? (double)df : 1.0);         // np.log(max(1.0, df))   (:132)
        t.v = (double)count * (ref_len - d);                     // tf * idf              (:136)
        if (df_dbg) *df_dbg = df;
    }
    return t;
}
__global__ __launch_bounds__(64 * SCORE_W) void score_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ cnt,
                                                    const int32_t* __restrict__ nuniq, const int32_t* __restrict__ blen,
                                                    double ref_len, const uint64_t* __restrict__ ht_keys,
                                                    const int32_t* __restrict__ ht_df, uint32_t ht_mask,
                                                    const int32_t* __restrict__ ref_off, int B, int spi, double sigma,
                                                    const int32_t* __restrict__ bad, double* __restrict__ scores,
                                                    int32_t* __restrict__ df_out, unsigned* __restrict__ arrived,
                                                    float* __restrict__ reward, double* __restrict__ stats) {
    __shared__ double part[SCORE_W][NMAX];
    __shared__ double vref_s[SCORE_W][64];
    __shared__ double red_s[2][4];
    __shared__ unsigned ticket_s;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int h = blockIdx.x;
    const int img = (h % B) / spi;                               // gts[i % batch_size // seq_per_img] (rewards.py:55)
    const bool on = lane < nuniq[h];
    const uint64_t key = on ? keys[(size_t)h * 64 + lane] : EMPTY;
    const TfIdf th = tfidf_of(key, on ? cnt[(size_t)h * 64 + lane] : 0, on, ref_len, ht_keys, ht_df, ht_mask,
                              (df_out && w == 0 && on) ? df_out + (size_t)h * 64 + lane : nullptr);
    const double vh = th.v;
    const int n = th.n;
    double nh[NMAX];
#pragma unroll
    for (int k = 1; k <= NMAX; ++k) nh[k - 1] = sqrt(wave_sum_f64((on && n == k) ? vh * vh : 0.0));
    double tot[NMAX] = {0.0, 0.0, 0.0, 0.0};
    const int r0 = ref_off[img], r1 = ref_off[img + 1];
    for (int rb = r0; rb < r1; rb += SCORE_W) {
        const int r = rb + w;
        if (r < r1) {                                            // wave-uniform
            const int rs = 2 * B + r;
            const int nur = nuniq[rs];
            const bool ron = lane < nur;
            const TfIdf tr = tfidf_of(ron ? keys[(size_t)rs * 64 + lane] : EMPTY, ron ? cnt[(size_t)rs * 64 + lane] : 0, ron,
                                      ref_len, ht_keys, ht_df, ht_mask, (df_out && ron) ? df_out + (size_t)rs * 64 + lane : nullptr);
            vref_s[w][lane] = tr.v;                              // this wave's own row: no barrier needed (wave-synchronous LDS)
            double nr[NMAX];
#pragma unroll
            for (int k = 1; k <= NMAX; ++k) nr[k - 1] = sqrt(wave_sum_f64((ron && tr.n == k) ? tr.v * tr.v : 0.0));
            double c = 0.0;
            if (on) {
                const int pos = bsearch_key(keys + (size_t)rs * 64, nur, key);
                if (pos >= 0) {
                    const double vr = vref_s[w][pos];
                    c = (vh < vr ? vh : vr) * vr;                // min(hyp, ref) * ref  (:165)
                }
            }
            const double delta = (double)(blen[h] - blen[rs]);
            const double pen = exp(-(delta * delta) / (2.0 * sigma * sigma));
#pragma unroll
            for (int k = 1; k <= NMAX; ++k) {
                double val = wave_sum_f64(n == k ? c : 0.0);
                if (nh[k - 1] != 0.0 && nr[k - 1] != 0.0) val /= (nh[k - 1] * nr[k - 1]);    // (:167-168)
                if (lane == 0) part[w][k - 1] = val * pen;
            }
        }
        __syncthreads();
        if (w == 0) {
            for (int j = 0; j < SCORE_W && rb + j < r1; ++j)
#pragma unroll
                for (int k = 0; k < NMAX; ++k) tot[k] += part[j][k];   // (:172), in reference order
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double avg = (tot[0] + tot[1] + tot[2] + tot[3]) / (double)NMAX;   // np.mean over n  (:194)
        avg /= (double)(r1 - r0);
        const double sc = *bad ? __longlong_as_double(0x7ff8000000000000ll) : avg * 10.0;   // out-of-vocabulary token: NaN
        __hip_atomic_store(scores + h, sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned t = 0;
        if (arrived) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ticket_s = t;
    }
    if (!arrived || !reward) return;
    __syncthreads();
    if (ticket_s != gridDim.x - 1u) return;                      // (workgroup-uniform)
    // the reward of every row and the two means: reward_kernel's 256 threads, sums and order
    if (threadIdx.x < 256) {
        double a = 0.0, g = 0.0;
        for (int b = threadIdx.x; b < B; b += 256) {
            const double ss = __hip_atomic_load(scores + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double sg = __hip_atomic_load(scores + B + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            reward[b] = (float)(ss - sg);                           // rewards.py:66
            a += ss;
            g += sg;
        }
        a = wave_sum_f64(a);
        g = wave_sum_f64(g);
        if ((threadIdx.x & 63) == 0) { red_s[0][threadIdx.x >> 6] = a; red_s[1][threadIdx.x >> 6] = g; }
    }
    __syncthreads();
    if (threadIdx.x == 0 && stats) {
        stats[0] = (red_s[0][0] + red_s[0][1] + red_s[0][2] + red_s[0][3]) / B;   // mean CIDEr-D of the sampled captions
        stats[1] = (red_s[1][0] + red_s[1][1] + red_s[1][2] + red_s[1][3]) / B;   // cider_greedy
    }
}

// (scores without a reward: callers that only want the scores)
__global__ void reward_kernel(const double* __restrict__ scores, int B, float* __restrict__ reward,
                              double* __restrict__ stats) {
    __shared__ double sh[2][4];
    double a = 0.0, g = 0.0;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const double d = scores[b] - scores[B + b];               // rewards.py:66
        reward[b] = (float)d;
        a += scores[b];
        g += scores[B + b];
    }
    a = wave_sum_f64(a);
    g = wave_sum_f64(g);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = g; }
    __syncthreads();
    if (threadIdx.x == 0 && stats) {
        stats[0] = (sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]) / B;   // mean CIDEr-D of the sampled captions
        stats[1] = (sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]) / B;   // cider_greedy
    }
}

// one launch readies the step's tables: empty hash table (keys all ones, document frequencies 0), the bad-token flag,
// and the image of every reference
__global__ __launch_bounds__(256) void cider_init_kernel(const int32_t* __restrict__ ref_off, int n_images,
                                                         int32_t* __restrict__ ref_img, uint64_t* __restrict__ ht_keys,
                                                         int32_t* __restrict__ ht_df, uint32_t ht_size, int32_t* __restrict__ bad,
                                                         unsigned* __restrict__ arrived) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ht_size) { ht_keys[i] = ~0ull; ht_df[i] = 0; }
    if (i == 0) { *bad = 0; *arrived = 0u; }
    if (i < (uint32_t)n_images)
        for (int r = ref_off[i]; r < ref_off[i + 1]; ++r) ref_img[r] = i;
}

struct CidWs {
    uint64_t *keys, *ht_keys;
    int32_t *cnt, *nuniq, *blen, *ht_df, *ref_img, *df, *bad;
    unsigned* arrived;        // score_kernel: workgroups whose score has landed (the last one makes the reward)
    double *vec, *norm;
    uint32_t ht_size;
    size_t bytes;
};
CidWs cid_carve(int B, int R, void* base) {
    CidWs w;
    Carver c(base);
    const size_t S = (size_t)2 * B + R;
    uint32_t ht = 1024;
    while (ht < 4u * (uint32_t)R * 64u) ht <<= 1;
    w.ht_size = ht;
    w.keys = c.u64(S * 64);
    w.ht_keys = c.u64(ht);
    w.vec = c.f64(S * 64);
    w.norm = c.f64(S * NMAX);
    w.cnt = c.i32(S * 64);
    w.df = c.i32(S * 64);
    w.nuniq = c.i32(S);
    w.blen = c.i32(S);
    w.ht_df = c.i32(ht);
    w.ref_img = c.i32(R);
    w.bad = c.i32(1);
    w.arrived = reinterpret_cast<unsigned*>(c.i32(1));
    w.bytes = c.used();
    return w;
}

}  // namespace

extern "C" size_t cic_ciderd_ws_bytes(int B, int R) { return cid_carve(B, R, nullptr).bytes; }

extern "C" int cic_ciderd_reward(const cic_ciderd_args* a, void* ws, size_t ws_bytes, cic_stream_t s) {
    CIC_REQUIRE(a && ws && a->gen && a->greedy && a->L_gen && a->L_greedy && a->refs && a->ref_off && a->scores);
    CIC_REQUIRE(a->B > 0 && a->R > 0 && a->n_images > 0 && a->spi > 0 && a->B == a->n_images * a->spi);
    CIC_REQUIRE(a->T > 0 && a->T <= MAXTOK && a->Tr > 0 && a->Tr <= MAXTOK);
    // n-gram keys hold 15 bits per token: ids 0 .. vocab_size + 1 (<eos>, the words, <bos>) must fit
    if (a->vocab_size <= 0 || a->vocab_size + 1 > 0x7fff) {
        cic_set_error("cic_ciderd_reward: vocab_size %d is outside 1..%d (n-gram keys pack 15 bits per token)",
                      a->vocab_size, 0x7fff - 1);
        return 1;
    }
    CidWs w = cid_carve(a->B, a->R, ws);
    CIC_REQUIRE(ws_bytes >= w.bytes);
    hipStream_t st = cic_s(s);
    const int B = a->B, R = a->R, S = 2 * B + R;
    Sents sn = {a->gen, a->greedy, a->refs, a->L_gen, a->L_greedy, B, a->T, R, a->Tr};
    // three launches (round 3: six): tables cleared; n-grams of every sentence + document frequencies; vectors, scores, reward
    hipLaunchKernelGGL(cider_init_kernel, dim3(cic_cdiv((int)(w.ht_size > (uint32_t)a->n_images ? w.ht_size : (uint32_t)a->n_images), 256)),
                       dim3(256), 0, st, a->ref_off, a->n_images, w.ref_img, w.ht_keys, w.ht_df, w.ht_size, w.bad, w.arrived);
    // every image's reference set is seen by 2*spi hypothesis entries (sampled + greedy halves, rewards.py:53-56)
    const int hyp_wgs = cic_cdiv(2 * B, 4);
    hipLaunchKernelGGL(tables_kernel, dim3(hyp_wgs + a->n_images), dim3(256), 0, st, sn, hyp_wgs, a->vocab_size + 1, w.keys, w.cnt,
                       w.nuniq, w.blen, w.bad, a->ref_off, 2 * a->spi, w.ht_keys, w.ht_df, w.ht_size - 1);
    // images with more references than tables_kernel keeps in LDS (the host knows the offsets only on the device: the launch
    // returns at once for every other image) - only when R allows such an image at all
    if (R > DF_MAX_REFS && (a->max_refs_per_image <= 0 || a->max_refs_per_image > DF_MAX_REFS))
        hipLaunchKernelGGL(df_kernel, dim3(cic_cdiv(R, 4)), dim3(256), 0, st, w.keys, w.nuniq, w.ref_img, a->ref_off, 2 * B,
                           R, 2 * a->spi, w.ht_keys, w.ht_df, w.ht_size - 1);
    const double ref_len = log((double)(2 * B));                  // np.log(float(len(self.crefs)))  (:178-179)
    hipLaunchKernelGGL(score_kernel, dim3(2 * B), dim3(64 * SCORE_W), 0, st, w.keys, w.cnt, w.nuniq, w.blen, ref_len, w.ht_keys,
                       w.ht_df, w.ht_size - 1, a->ref_off, B, a->spi, 6.0, w.bad, a->scores, a->dbg_keys ? w.df : nullptr,
                       w.arrived, a->reward, a->stats);
    CIC_LAUNCH_CHECK();
    if (a->dbg_keys) {   // exact-integer tables for the parity tests
        CIC_HIP(hipMemcpyAsync(a->dbg_keys, w.keys, sizeof(uint64_t) * S * 64, hipMemcpyDeviceToDevice, st));
        CIC_HIP(hipMemcpyAsync(a->dbg_cnt, w.cnt, sizeof(int32_t) * S * 64, hipMemcpyDeviceToDevice, st));
        CIC_HIP(hipMemcpyAsync(a->dbg_df, w.df, sizeof(int32_t) * S * 64, hipMemcpyDeviceToDevice, st));
        CIC_HIP(hipMemcpyAsync(a->dbg_nuniq, w.nuniq, sizeof(int32_t) * S, hipMemcpyDeviceToDevice, st));
    }
    return 0;
}
