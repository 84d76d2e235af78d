// Internal helpers shared by the HIP translation units of libcic_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "cic.h"

void cic_set_error(const char* fmt, ...);

#define CIC_HIP(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            cic_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return 2;                                                                      \
        }                                                                                  \
    } while (0)

#define CIC_REQUIRE(cond, ...)                                                             \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            cic_set_error("%s:%d requirement failed: %s", __FILE__, __LINE__, #cond);      \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

#define CIC_LAUNCH_CHECK() CIC_HIP(hipGetLastError())

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) holds PER DEVICE: a process that drives several GPUs has to set it on
// each of them.  first() is true once per device (a bit per device ordinal, set atomically; idempotent work behind it).
#include <atomic>
struct DeviceOnce {
    std::atomic<uint64_t> done{0};
    bool first() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return true;
        const uint64_t bit = 1ull << (dev & 63);
        return (done.fetch_or(bit, std::memory_order_relaxed) & bit) == 0;
    }
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// in-situ timing (core.hip): launches bracketed by HIP events of a caller-owned cic_timer; no-ops for a NULL timer
void* cic_timer_begin(cic_timer* t, int id, hipStream_t st);
void cic_timer_end(void* h, hipStream_t st);
#define CIC_TIMED(timer, id, st, stmt)                      \
    do {                                                    \
        void* ph_ = cic_timer_begin((timer), (id), (st));   \
        stmt;                                               \
        cic_timer_end(ph_, (st));                           \
    } while (0)

// Dispatch switches and in-kernel stamp buffers exist only in the development build of the library
// (-DCIC_DEVTOOLS -> libcic_hip_dev.so, setters declared in include/cic_dev.h; tools/ use it for A/B timing).  In the
// product build (libcic_hip.so) they are compile-time constants: the library holds no mutable global state.
#ifdef CIC_DEVTOOLS
#define CIC_SWITCH(name, value) int name = value
#define CIC_STAMP_BUF(sym) sym
#else
#define CIC_SWITCH(name, value) static constexpr int name = value
#define CIC_STAMP_BUF(sym) nullptr
#endif

uint64_t cic_hash_bytes(const void* p, size_t n, uint64_t h);

// ---- hand-off time-outs of the one-launch recurrences (cic.h: "status word") -----------------------------------------------
// Every spin of those kernels is bounded by `ticks` of the 100 MHz s_memrealtime counter (1 s).  A workgroup that gives up
// raises the loop's error word (cleared with the loop's counters), ORs the loop's bit into the caller's STICKY status word and
// poisons what it produces with NaN; once it has given up it does not wait again (a launch then ends within ~one bound, not
// hand-offs x steps of them).  Development build only: the bound can be lowered and ONE workgroup of a chosen loop told never
// to count itself in (cic_dev.h: cic_debug_spin_ticks, cic_debug_handoff_fault) - how tests force the failure.
#ifdef CIC_DEVTOOLS
extern unsigned long long g_spin_ticks;
extern int g_fault_loop, g_fault_wg;
#else
static constexpr unsigned long long g_spin_ticks = 1000ull * 100000ull;
static constexpr int g_fault_loop = 0, g_fault_wg = -1;
#endif
struct HandoffGuard {
    unsigned* err;                // the loop's error word (in its workspace)
    unsigned* status;             // the caller's sticky word or null
    unsigned bit;                 // CIC_STATUS_*
    unsigned long long ticks;     // spin bound
    int fault_wg;                 // blockIdx.x of the workgroup that withholds its counter adds, or -1
};
static inline HandoffGuard handoff_guard(unsigned* err, uint32_t* status, unsigned bit) {
    HandoffGuard g;
    g.err = err; g.status = status; g.bit = bit; g.ticks = g_spin_ticks;
    g.fault_wg = (g_fault_loop & (int)bit) ? g_fault_wg : -1;
    return g;
}
#ifdef __HIPCC__
// ONE lane: poll `*c` until it reaches `target` or the bound passes.  Returns 1 when it was reached.
__device__ __forceinline__ int handoff_poll(unsigned* c, unsigned target, const HandoffGuard& g) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_amdgcn_s_memrealtime() - t0 > g.ticks) {
            __hip_atomic_store(g.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (g.status) __hip_atomic_fetch_or(g.status, g.bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return 0;
        }
    }
    return 1;
}
// ONE lane counts its workgroup in (unless it is the development build's faulty workgroup)
__device__ __forceinline__ void handoff_arrive(unsigned* c, const HandoffGuard& g) {
    if ((int)blockIdx.x != g.fault_wg) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

static inline hipStream_t cic_s(cic_stream_t s) { return (hipStream_t)s; }

// A pointer per decode of a PAIR of decodes that advance in lock step through the same launches
// (rows [0,B) belong to decode a, rows [B,2B) to decode b; b is unused for a single decode).
template <typename T>
struct Dual {
    T* a;
    T* b;
    __host__ __device__ T* sel(bool second) const { return second ? b : a; }
};
template <typename T>
static inline Dual<T> dual1(T* p) { return Dual<T>{p, nullptr}; }
static inline int cic_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- wave / block reductions (wave = 64 lanes) ---------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- DPP cross-lane moves: VALU-rate lane exchange inside a 16-lane row (no LDS crossbar) -----------
// __shfl_xor lowers to ds_bpermute (an LDS-pipe instruction, ~60 cycles of latency per DEPENDENT step);
// a 64-lane reduction is 6 such steps.  The first four steps (xor 1, 2, then mirror within 8 and 16 lanes)
// exist as DPP modifiers on gfx950, leaving two cross-row steps.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_QUAD_XOR1 = 0xB1;       // quad_perm [1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;       // quad_perm [2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;   // lane i <-> 7-i within 8
constexpr int DPP_ROW_MIRROR = 0x140;        // lane i <-> 15-i within 16
constexpr int DPP_ROW_ROR8 = 0x128;          // lane i <- lane (i+8)%16: xor 8 within a row

// sum over the 8 lanes that share lane>>3 (all 8 end with the total)
__device__ __forceinline__ float sum8_dpp(float v) {
    v += dpp_f32<DPP_QUAD_XOR1>(v);
    v += dpp_f32<DPP_QUAD_XOR2>(v);
    v += dpp_f32<DPP_ROW_HALF_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float readlane_f32(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// whole-wave reductions without a single LDS-pipe instruction: 4 DPP steps inside each 16-lane row, then
// the four row totals through v_readlane (scalar broadcast).  (v_permlane32_swap via the compiler builtin
// returned wrong lanes on ROCm 7.2 / gfx950 in a direct test, so it is not used.)
__device__ __forceinline__ float wave_sum_fast(float v) {
    v = sum8_dpp(v);
    v += dpp_f32<DPP_ROW_MIRROR>(v);
    return (readlane_f32(v, 0) + readlane_f32(v, 16)) + (readlane_f32(v, 32) + readlane_f32(v, 48));
}
__device__ __forceinline__ float wave_max_fast(float v) {
    v = fmaxf(v, dpp_f32<DPP_QUAD_XOR1>(v));
    v = fmaxf(v, dpp_f32<DPP_QUAD_XOR2>(v));
    v = fmaxf(v, dpp_f32<DPP_ROW_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f32<DPP_ROW_MIRROR>(v));
    return fmaxf(fmaxf(readlane_f32(v, 0), readlane_f32(v, 16)), fmaxf(readlane_f32(v, 32), readlane_f32(v, 48)));
}
// sum over the 8 lanes that share lane&7 (xor 8, 16, 32)
__device__ __forceinline__ float sum_over_rg(float v) {
    v += dpp_f32<DPP_ROW_ROR8>(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// ---- fast, accurate-enough transcendental helpers (f32, abs err ~1e-7) ----------------
__device__ __forceinline__ float fast_tanh(float x) {
    // tanh(x) = sign(x) * (1 - e) / (1 + e),  e = exp(-2|x|)
    float ax = fabsf(x);
    float e = __expf(-2.0f * ax);
    float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);   // v_rcp_f32: 1 ulp, no div fix-up sequence
    return copysignf(t, x);
}
__device__ __forceinline__ float fast_sigmoid(float x) {
    // stable on both sides: 1/(1+e^-x) for x>=0, e^x/(1+e^x) for x<0
    float e = __expf(-fabsf(x));
    float r = __builtin_amdgcn_rcpf(1.0f + e);
    return x >= 0.0f ? r : e * r;
}

// ---- Philox4x32-10 ---------------------------------------------------------------------
struct Philox4 {
    uint32_t v[4];
};
// one Philox4x32 call as a state machine, so that its ten rounds can be spread over the slots of a software pipeline
struct PhiloxState {
    uint32_t c0, c1, c2, c3, k0, k1;
    __host__ __device__ __forceinline__ void init(uint64_t counter, uint64_t seed) {
        c0 = (uint32_t)counter; c1 = (uint32_t)(counter >> 32); c2 = 0u; c3 = 0u;
        k0 = (uint32_t)seed; k1 = (uint32_t)(seed >> 32);
    }
    __host__ __device__ __forceinline__ void round() {
        const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
};
__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint64_t counter, uint64_t seed) {
    PhiloxState s;
    s.init(counter, seed);
#pragma unroll
    for (int r = 0; r < 10; ++r) s.round();
    Philox4 o;
    o.v[0] = s.c0; o.v[1] = s.c1; o.v[2] = s.c2; o.v[3] = s.c3;
    return o;
}
__host__ __device__ __forceinline__ float u32_to_unit(uint32_t r) {
    return (float)(r >> 8) * (1.0f / 16777216.0f);   // [0,1), 24 bits like torch.rand
}

// ---- Gumbel noise and vocabulary row partials (cic.h: "Row partials of the vocabulary") ------------------------------
__device__ __forceinline__ float gumbel_from_u(float u) {
    // -log(-log(U + eps) + eps), eps = 1e-20, in f32 as models/gumbel.py:6-11.  __logf = v_log_f32 * ln 2 (1 ulp, two
    // instructions; the arguments are never denormal: u + eps >= 1e-20, -log(.) + eps in [1e-20, 46.1]): the library
    // logf's range handling would cost ~10x the issue slots in the logit walker's epilogue, for the same last-bit noise
    return -__logf(-__logf(u + 1e-20f) + 1e-20f);
}
// uniform `lane` (0..3) of Philox call `q` of the stream cic_uniform_f32(seed, offset 0) writes
__device__ __forceinline__ f32x4 philox_uniform4(uint64_t seed, uint64_t q) {
    const Philox4 r = philox4x32_10(q, seed);
    return f32x4{u32_to_unit(r.v[0]), u32_to_unit(r.v[1]), u32_to_unit(r.v[2]), u32_to_unit(r.v[3])};
}

// Running reduction of one row over a subset of its columns.  "Empty" is -FLT_MAX, not -inf: every update and merge
// below is then straight-line arithmetic (exp(-huge) = 0; no (-inf) - (-inf)), and a constrained column (x = -inf)
// simply adds exp(-inf) = 0.
constexpr float RP_EMPTY = -3.402823466e38f;
struct RowPart {
    float m1, s1;        // max x, sum exp(x - m1)
    float kbest, xbest;  // best key, logit there
    int kidx;            // its column (lowest among equal keys)
    float s2;            // see cic.h
    __device__ __forceinline__ void init() { m1 = RP_EMPTY; s1 = 0.f; kbest = RP_EMPTY; xbest = RP_EMPTY; kidx = 0x7fffffff; s2 = 0.f; }
};
// online softmax step: (m, s) absorbs one value; one exp per value, no branch
__device__ __forceinline__ void osm_add(float& m, float& s, float x) {
    const float e = __expf(-fabsf(x - m));
    s = x > m ? s * e + 1.0f : s + e;
    m = fmaxf(m, x);
}
// the same step for a sum whose reference maximum `m` is maintained by the caller (read only here)
__device__ __forceinline__ void osm_add_ref(float m, float& s, float x) {
    const float e = __expf(-fabsf(x - m));
    s = x > m ? s * e + 1.0f : s + e;
}
// merge of two (m, s) pairs
__device__ __forceinline__ void osm_merge(float& m, float& s, float m2, float s2) {
    const float M = fmaxf(m, m2);
    s = s * __expf(m - M) + s2 * __expf(m2 - M);
    m = M;
}
// MODE: CIC_SAMPLE_* (compile time); x: logit (already -inf at the constrained column); g: Gumbel noise or 0; col: its
// column.  Within one caller the columns arrive in increasing order, so a strictly larger key is the only way to
// replace the best one (ties keep the lowest column).
template <int MODE>
__device__ __forceinline__ void rowpart_add_m(RowPart& p, float inv_t, float x, float g, int col) {
    const float m_old = p.m1;
    osm_add(p.m1, p.s1, x);
    if (MODE == CIC_SAMPLE_NONE) return;
    if (MODE == CIC_SAMPLE_MULTINOMIAL_ST) {
        // s2 = sum exp((x - m1) * inv_t), re-based whenever the running maximum moves.  The DIFFERENCE is scaled, never
        // the operands: an empty side (m = -FLT_MAX) times inv_t > 1 would overflow to -inf and (-inf) - (-inf) = NaN
        p.s2 = p.s2 * __expf((m_old - p.m1) * inv_t) + __expf((x - p.m1) * inv_t);
    }
    const bool gum = MODE == CIC_SAMPLE_GUMBEL_ST;
    const float k = MODE == CIC_SAMPLE_GREEDY ? x : (gum ? (x + g) * inv_t : x * inv_t + g);
    if (gum) osm_add_ref(p.kbest, p.s2, k);      // s2 = sum exp(k - kbest): the online form with the best key as the maximum
    const bool better = k > p.kbest;
    p.kbest = better ? k : p.kbest;
    p.xbest = better ? x : p.xbest;
    p.kidx = better ? col : p.kidx;
}
// the runtime-mode form (a wave-uniform switch over the straight-line bodies)
__device__ __forceinline__ void rowpart_add(RowPart& p, int mode, float inv_t, float x, float g, int col) {
    switch (mode) {
        case CIC_SAMPLE_NONE: rowpart_add_m<CIC_SAMPLE_NONE>(p, inv_t, x, g, col); break;
        case CIC_SAMPLE_GREEDY: rowpart_add_m<CIC_SAMPLE_GREEDY>(p, inv_t, x, g, col); break;
        case CIC_SAMPLE_GUMBEL_ST: rowpart_add_m<CIC_SAMPLE_GUMBEL_ST>(p, inv_t, x, g, col); break;
        case CIC_SAMPLE_MULTINOMIAL_ST: rowpart_add_m<CIC_SAMPLE_MULTINOMIAL_ST>(p, inv_t, x, g, col); break;
        default: rowpart_add_m<CIC_SAMPLE_MULTINOMIAL>(p, inv_t, x, g, col); break;   // MULTINOMIAL, TEACHER
    }
}
__device__ __forceinline__ void rowpart_merge(RowPart& p, int mode, float inv_t, const RowPart& q) {
    if (mode == CIC_SAMPLE_MULTINOMIAL_ST) {
        const float M = fmaxf(p.m1, q.m1);      // (m - M) is 0 or about -FLT_MAX for an empty side: finite, exp -> 1 or 0
        p.s2 = p.s2 * __expf((p.m1 - M) * inv_t) + q.s2 * __expf((q.m1 - M) * inv_t);
    }
    if (mode == CIC_SAMPLE_GUMBEL_ST) {
        float kb = p.kbest, s = p.s2;
        osm_merge(kb, s, q.kbest, q.s2);
        p.s2 = s;
    }
    osm_merge(p.m1, p.s1, q.m1, q.s1);
    const bool better = q.kbest > p.kbest || (q.kbest == p.kbest && q.kidx < p.kidx);
    p.kbest = better ? q.kbest : p.kbest;
    p.xbest = better ? q.xbest : p.xbest;
    p.kidx = better ? q.kidx : p.kidx;
}

// What the sampler does with a row once its partials are merged (models/AttModel.py:328-365,401-434; gumbel.py:13-30;
// multinomial.py:4-27): token choice, gathered log-prob, straight-through value (sample_finish_kernel).  `tok` is the token
// the NEXT core step embeds (un-masked, :399); `it`/`slp`/`v` are what the bookkeeping stores.
struct RowChoice { int it, tok; float slp, v, lse; };
__device__ __forceinline__ RowChoice row_choice(const cic_sampler_args& a, const RowPart& rp, int b) {
    RowChoice c;
    const float inv_t = 1.0f / a.temp;
    c.lse = rp.m1 + logf(rp.s1);
    const bool gumbel_mode = a.mode == CIC_SAMPLE_GUMBEL_ST;
    const bool ss_on = a.mode == CIC_SAMPLE_TEACHER && a.ss_u && a.ss_prob > 0.f;
    // a row of NaN logits (state poisoned by a failed hand-off, NaN weights) has no arg max - no key compares greater - and
    // leaves the empty partial's column: the token then is 0 (<eos>), never an index outside the embedding table.  Its
    // log-prob is NaN, which is what the loss shows.
    const int kidx = (unsigned)rp.kidx < (unsigned)a.V1 ? rp.kidx : 0;
    int it = kidx;
    int it_feed = -1;                                   // teacher mode: token fed to the next step
    if (a.mode == CIC_SAMPLE_TEACHER) {
        const int target = (int)a.pick[b];
        const int drawn = a.ss_pick ? (int)a.ss_pick[b] : kidx;
        it_feed = (ss_on && a.ss_u[b] < a.ss_prob) ? drawn : target;   // AttModel.py:119-128
        it = target;                                     // the loss gathers log p(target)
    } else if (a.pick && a.mode != CIC_SAMPLE_GREEDY && !gumbel_mode) {
        it = (int)a.pick[b];
    }
    float x_it = rp.xbest;
    if (it != kidx || kidx != rp.kidx) {
        const int cons = (a.decoding_constraint && a.step >= 2) ? a.seq[(size_t)b * a.seq_ld + (a.step - 2)] : -1;
        x_it = it == cons ? -INFINITY : a.logits[(size_t)b * a.ld + it];
    }
    c.slp = x_it - c.lse;
    float v = 1.0f;
    if (gumbel_mode) {
        const float y = 1.0f / rp.s2;                   // softmax(k)[arg max k] = exp(0) / sum exp(k - kbest), gumbel.py:13-15
        v = (1.0f - y) + y;                              // (y_hard - y).detach() + y at the arg-max entry, gumbel.py:28
    } else if (a.mode == CIC_SAMPLE_MULTINOMIAL_ST) {
        const float y = __expf((x_it - rp.m1) * inv_t) / rp.s2;         // softmax(logp / tau)[it], multinomial.py:10-15
        v = (1.0f - y) + y;
    }
    c.v = v;
    c.it = it;
    c.tok = it_feed >= 0 ? it_feed : it;
    return c;
}
// EOS bookkeeping of one row (AttModel.py:401-434), by ONE lane
__device__ __forceinline__ void row_bookkeeping(const cic_sampler_args& a, const RowChoice& c, int b) {
    const int t = a.step;
    int unf = (c.it > 0) ? 1 : 0;
    if (t > 1) unf = unf & a.unfinished[b];
    a.unfinished[b] = unf;
    a.it_next[b] = c.tok;                                     // un-masked: embed(it) precedes the masking (:399)
    a.seq[(size_t)b * a.seq_ld + (t - 1)] = unf ? c.it : 0;   // it * unfinished (:409)
    a.slp[(size_t)b * a.seq_ld + (t - 1)] = c.slp;
    if (a.stv) a.stv[(size_t)b * a.seq_ld + (t - 1)] = unf ? c.v : 1.0f;   // finished rows -> exact EOS one-hot (:419-420)
    if (unf) atomicOr(a.any_unfinished + t, 1);
}
// xt = dropout(relu(embed(tok))) (AttModel.py:74-76,399) for four consecutive columns; kp: their four keep bytes
__device__ __forceinline__ f32x4 embed_transform(const cic_sampler_args& a, f32x4 ev, uint32_t kp) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float kf = (float)((kp >> (8 * e)) & 0xffu);
        const float r = a.emb_plain ? ev[e] : fmaxf(ev[e], 0.f);
        ev[e] = a.emb_keep ? r * (kf * a.emb_scale) : r;
    }
    return ev;
}
