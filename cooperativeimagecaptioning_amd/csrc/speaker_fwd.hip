// Speaker (att2in2) forward kernels: per-timestep additive attention over the K embedded
// regions, the maxout-LSTM cell pointwise part, token embedding, and the fused
// log-softmax + sampler + EOS bookkeeping row kernel.
//
// All kernels are HBM/L2-streaming (no contraction): coalesced 16-B loads, wave-64 shuffles
// for the per-region reductions, LDS only for the cross-wave hand-off.  The GEMMs around them
// (h2att, i2h/h2h, a2c, logit) are cic_gemm_f32.
#include "cic_common.h"
#include <mutex>
#include "engine_util.h"

namespace {

// ---------------------------------------------------------------------------------------
// K3 attention step  (reference: Attention.forward, models/AttModel.py:465-489)
//   dot[k]  = b_a + sum_a w_a * tanh(p_att[b,k,a] + att_h[b,a])
//   alpha   = softmax_k(dot)  (optionally * mask, renormalised)
//   att_res = sum_k alpha[k] * att[b,k,:]
// One workgroup (4 waves) per image.  Wave w owns regions k = w, w+4, ...; lane l owns the
// float4 columns l, l+64, ... of a region row, so p_att and att rows are read with fully
// coalesced 1-KiB wave loads.  HOLD=true keeps the att tile of the wave's regions in
// registers across the softmax barrier (one pass over HBM/L2 for both operands).
// ---------------------------------------------------------------------------------------
template <int NI, int KPW, int NW, bool HOLD>
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(const float* __restrict__ att_h,   // [B,A]
                                                       const float* __restrict__ p_att,   // [B,K,A]
                                                       const float* __restrict__ att,     // [B,K,H]
                                                       const float* __restrict__ w_alpha, // [A]
                                                       const float* __restrict__ b_alpha, // [1]
                                                       const float* __restrict__ masks,   // [B,K] or null
                                                       float* __restrict__ att_res,       // [B,H]
                                                       float* __restrict__ alpha_out,     // [B,K]
                                                       float* __restrict__ dot_out,       // [B,K] or null
                                                       int K, int A, int H, int att_div) {
    __shared__ float sdot[64];
    __shared__ __attribute__((aligned(16))) float sacc[NW * NI * 256];
    __shared__ __attribute__((aligned(16))) float sacc2[4 * NI * 256];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int bi = b / att_div;                 // image whose regions row b attends to (beam search: att_div rows per image)
    const int A4 = A >> 2, H4 = H >> 2;
    const f32x4* pa4 = reinterpret_cast<const f32x4*>(p_att + (size_t)bi * K * A);
    const f32x4* at4 = reinterpret_cast<const f32x4*>(att + (size_t)bi * K * H);
    const f32x4* ah4 = reinterpret_cast<const f32x4*>(att_h + (size_t)b * A);
    const f32x4* wa4 = reinterpret_cast<const f32x4*>(w_alpha);

    f32x4 ah[NI], wa[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        ah[i] = c < A4 ? ah4[c] : z;
        wa[i] = c < A4 ? wa4[c] : z;
    }
    const float ba = b_alpha[0];

    f32x4 av[HOLD ? KPW : 1][NI];
#pragma unroll
    for (int j = 0; j < KPW; ++j) {
        const int k = w + NW * j;
        if (k < K) {   // wave-uniform
            f32x4 p[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int c = lane + 64 * i;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                p[i] = c < A4 ? pa4[(size_t)k * A4 + c] : z;
                if (HOLD) av[j][i] = c < H4 ? at4[(size_t)k * H4 + c] : z;
            }
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) part += wa[i][e] * fast_tanh(p[i][e] + ah[i][e]);
            part = wave_sum(part);
            if (lane == 0) sdot[k] = part + ba;
        }
    }
    __syncthreads();
    // softmax over the K regions — every thread redoes the tiny reduction from LDS
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, sdot[k]);
    float sum = 0.f;
    for (int k = 0; k < K; ++k) sum += __expf(sdot[k] - mx);
    const float inv = 1.0f / sum;
    float minv = 1.0f;
    if (masks) {   // weight = weight * mask; weight /= weight.sum()   (AttModel.py:481-483)
        float ms = 0.f;
        for (int k = 0; k < K; ++k) ms += __expf(sdot[k] - mx) * inv * masks[(size_t)bi * K + k];
        minv = 1.0f / ms;
    }
    f32x4 acc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KPW; ++j) {
        const int k = w + NW * j;
        if (k < K) {
            float a = __expf(sdot[k] - mx) * inv;
            if (masks) a = a * masks[(size_t)bi * K + k] * minv;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int c = lane + 64 * i;
                f32x4 v;
                if (HOLD) {
                    v = av[j][i];
                } else {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    v = c < H4 ? at4[(size_t)k * H4 + c] : z;
                }
                acc[i] += a * v;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
        *reinterpret_cast<f32x4*>(&sacc[(w * NI * 64 + i * 64 + lane) * 4]) = acc[i];
    __syncthreads();
    // two-level cross-wave sum: 4 partial groups of NW/4 waves, then the 4 partials
    for (int t = tid; t < 4 * H4; t += NW * 64) {
        const int part = t / H4, c = t % H4;
        const int i = c >> 6, l = c & 63;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NW / 4; ++q)
            s += *reinterpret_cast<f32x4*>(&sacc[(((part * (NW / 4) + q) * NI + i) * 64 + l) * 4]);
        *reinterpret_cast<f32x4*>(&sacc2[((part * NI + i) * 64 + l) * 4]) = s;
    }
    __syncthreads();
    for (int c = tid; c < H4; c += NW * 64) {
        const int i = c >> 6, l = c & 63;
        f32x4 s = *reinterpret_cast<f32x4*>(&sacc2[((0 * NI + i) * 64 + l) * 4]);
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) s += *reinterpret_cast<f32x4*>(&sacc2[((ww * NI + i) * 64 + l) * 4]);
        reinterpret_cast<f32x4*>(att_res + (size_t)b * H)[c] = s;
    }
    if (tid < K) {
        float a = __expf(sdot[tid] - mx) * inv;
        if (masks) a = a * masks[(size_t)bi * K + tid] * minv;
        alpha_out[(size_t)b * K + tid] = a;
        if (dot_out) dot_out[(size_t)b * K + tid] = sdot[tid];
    }
}


// ---------------------------------------------------------------------------------------------
// K3 attention step, column-owner layout (A == H, H % 32 == 0, the shapes the model uses).
// One workgroup per image, NW = H/32 waves (16 for H = 512).  Wave w owns the 32 columns
// [32w, 32w+32) of every region row; lane = (rg = lane>>3 region sub-index, c = lane&7 float4 column),
// so ONE wave load instruction reads 8 regions x 128 B (whole cache lines) and all loads of the
// wave (<= 2*JMAX float4 per lane) are issued before the first use.
//   phase 1: per-lane 4-column partial of w.tanh(p+ah), 3 xor-shuffles over c -> partial dot of
//            (wave, region) -> LDS [NW][K];  ONE workgroup barrier
//   phase 2: every wave redundantly sums the NW partials of region k in lane k and does the
//            36-way softmax with wave shuffles (no second barrier, no LDS broadcast)
//   phase 3: alpha-weighted sum of the att values still held in registers, 3 xor-shuffles over
//            rg -> lanes 0..7 store the wave's 32 output columns (one 128-B line)
// No cross-wave vector reduction at all.
// ---------------------------------------------------------------------------------------------
// TWIN: two workgroups per image (grid = 2B) when the batch alone cannot fill the chip: both compute
// all K scores (each reads all of p_att), each produces one half of the output columns (reads half of
// att).  Per-CU bytes drop from 147 KB to 110 KB and all 256 CUs pull from the Infinity Cache.
#ifdef CIC_DEVTOOLS
__device__ unsigned long long* g_attn_stamps = nullptr;   // diagnostics (cic_debug_set_attn_stamps, development build)
#endif

// four consecutive features of a region row: f32 storage, or bf16 storage widened to f32 (half the bytes per image)
template <typename ST>
__device__ __forceinline__ f32x4 ld_feat4(const ST* __restrict__ base, size_t idx4);
template <>
__device__ __forceinline__ f32x4 ld_feat4<float>(const float* __restrict__ base, size_t idx4) {
    return reinterpret_cast<const f32x4*>(base)[idx4];
}
template <>
__device__ __forceinline__ f32x4 ld_feat4<uint16_t>(const uint16_t* __restrict__ base, size_t idx4) {
    const uint2 v = reinterpret_cast<const uint2*>(base)[idx4];
    return f32x4{__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                 __uint_as_float(v.y & 0xffff0000u)};
}

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// a wave-uniform buffer descriptor over [ptr, ptr + bytes): raw buffer loads / stores with aux = 16 (sc1) are the write-through
// stores and cache-bypassing loads of the in-launch hand-offs
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc_f32(const float* ptr, size_t bytes) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(ptr);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo), 0, (int)bytes, 0x00020000);
}

// The attention of ONE row (image x decode) by one workgroup of H / (32 NCG) waves: the body of attn_fwd_cols_kernel, and the
// first phase of attn_a2c_cell_kernel (WT: att_res leaves through write-through stores - it is handed over inside the launch).
// `bx` = the row's index in [0, nb * B0); `wg` = the workgroup's index (stamps).
template <int JMAX, int NCG, typename ST, bool WT>   // region groups of 8: K <= 8*JMAX;  NCG 32-column groups per wave
__device__ __forceinline__ void attn_cols_row(int bx, int wg, float* sp, Dual<const float> att_h_d, Dual<const ST> p_att_d,
                                              Dual<const ST> att_d, const float* __restrict__ w_alpha,
                                              const float* __restrict__ b_alpha, const float* __restrict__ masks,
                                              Dual<float> att_res_d, Dual<float> alpha_d, Dual<float> dot_d, int B0, int K, int H,
                                              int att_div, float poison) {
    unsigned long long* stamps = CIC_STAMP_BUF(g_attn_stamps);
    unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (stamps) s0 = __builtin_amdgcn_s_memrealtime();
    // rows [0,B0) are images of decode a, rows [B0, 2*B0) the same images in decode b (its own activations)
    const bool second = bx >= B0;
    const int b = second ? bx - B0 : bx;
    const float* __restrict__ att_h = att_h_d.sel(second);
    const ST* __restrict__ p_att = p_att_d.sel(second);
    const ST* __restrict__ att = att_d.sel(second);
    float* __restrict__ att_res = att_res_d.sel(second);
    float* __restrict__ alpha_out = alpha_d.sel(second);
    float* __restrict__ dot_out = dot_d.sel(second);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int NW = blockDim.x >> 6;                     // H / (32 * NCG) waves
    const int c = lane & 7, rg = lane >> 3;
    const int H4 = H >> 2;
    const int bi = b / att_div;                 // image whose regions row b attends to (beam search: att_div rows per image)
    const ST* pa4 = p_att + (size_t)bi * K * H;
    const ST* at4 = att + (size_t)bi * K * H;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    int col4[NCG];
    f32x4 ah[NCG], wa[NCG];
#pragma unroll
    for (int i = 0; i < NCG; ++i) {
        col4[i] = 8 * (w + NW * i) + c;                 // this lane's float4 column in column group i
        ah[i] = reinterpret_cast<const f32x4*>(att_h + (size_t)b * H)[col4[i]];
        wa[i] = reinterpret_cast<const f32x4*>(w_alpha)[col4[i]];
    }
    f32x4 pv[NCG][JMAX], av[NCG][JMAX];
#pragma unroll
    for (int i = 0; i < NCG; ++i)
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            const int k = 8 * j + rg;
            pv[i][j] = k < K ? ld_feat4<ST>(pa4, (size_t)k * H4 + col4[i]) : z4;
            av[i][j] = k < K ? ld_feat4<ST>(at4, (size_t)k * H4 + col4[i]) : z4;
        }
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < NCG; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) part += wa[i][e] * fast_tanh(pv[i][j][e] + ah[i][e]);
        part = sum8_dpp(part);                           // over the 8 float4 columns of a group (DPP, no LDS)
        const int k = 8 * j + rg;
        if (c == 0 && k < K) sp[w * 64 + k] = part;
        if (stamps && j == 0) s1 = __builtin_amdgcn_s_memrealtime();
    }
    if (stamps) s2 = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    if (stamps) s3 = __builtin_amdgcn_s_memrealtime();
    // lane k: full dot of region k, then softmax across lanes (every wave does the same arithmetic)
    float dot = -INFINITY;
    if (lane < K) {
        float s = 0.f;
        for (int q = 0; q < NW; ++q) s += sp[q * 64 + lane];
        dot = s + b_alpha[0];
    }
    const float mx = wave_max_fast(dot);
    float ex = lane < K ? __expf(dot - mx) : 0.f;
    const float sum = wave_sum_fast(ex);
    float al = ex * (1.0f / sum);
    if (masks) {   // weight = weight * mask; weight /= weight.sum()   (AttModel.py:481-483)
        al = lane < K ? al * masks[(size_t)bi * K + lane] : 0.f;
        const float ms = wave_sum_fast(al);
        al = al * (1.0f / ms);
    }
    if (w == 0 && lane < K) {
        alpha_out[(size_t)b * K + lane] = al;
        if (dot_out) dot_out[(size_t)b * K + lane] = dot;
    }
    float aj[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) aj[j] = __shfl(al, 8 * j + rg, 64);      // lanes >= K hold 0
    __amdgpu_buffer_rsrc_t r_res;
    if (WT) r_res = uniform_rsrc_f32(att_res + (size_t)b * H, (size_t)H * sizeof(float));
#pragma unroll
    for (int i = 0; i < NCG; ++i) {
        f32x4 acc = z4;
#pragma unroll
        for (int j = 0; j < JMAX; ++j) acc += aj[j] * av[i][j];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = sum_over_rg(acc[e]);
        if (WT && poison != 0.f) acc = f32x4{poison, poison, poison, poison};
        if (rg == 0) {
            if (WT) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc), r_res, col4[i] * 16, 0, 16);
            else reinterpret_cast<f32x4*>(att_res + (size_t)b * H)[col4[i]] = acc;
        }
    }
    if (stamps && lane == 0) {
        unsigned long long* o = stamps + ((size_t)wg * 16 + w) * 5;
        o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3; o[4] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int JMAX, int NCG, typename ST = float>
__global__ __launch_bounds__(1024) void attn_fwd_cols_kernel(Dual<const float> att_h_d, Dual<const ST> p_att_d,
                                                             Dual<const ST> att_d, const float* __restrict__ w_alpha,
                                                             const float* __restrict__ b_alpha, const float* __restrict__ masks,
                                                             Dual<float> att_res_d, Dual<float> alpha_d, Dual<float> dot_d, int B0,
                                                             int K, int H, int att_div, Dual<const int32_t> live) {
    __shared__ float sp[16 * 64];
    if (live.a && *live.a == 0 && (!live.b || *live.b == 0)) return;   // every caption has ended (AttModel.py:401-408)
    attn_cols_row<JMAX, NCG, ST, false>((int)blockIdx.x, (int)blockIdx.x, sp, att_h_d, p_att_d, att_d, w_alpha, b_alpha, masks, att_res_d,
                                        alpha_d, dot_d, B0, K, H, att_div, 0.f);
}



// ---------------------------------------------------------------------------------------
// K4 cell pointwise  (Att2in2Core.forward, models/AttModel.py:515-529)
//   pre[b, 0:5H] = i2h(x)+h2h(h) with a2c(att_res) already added to [3H:5H]
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cell_fwd_kernel(Dual<const float> pre_d, Dual<const float> c_prev_d,
                                                       Dual<const uint8_t> keep_d, float scale, Dual<float> h_new_d,
                                                       Dual<float> c_new_d, Dual<float> out_d, int B, int nb, int H,
                                                       int state_dropped) {
    // state_dropped: LSTMCore (models/FCModel.py:38-42) feeds the DROPPED-OUT h back as the recurrent state
    const int H4 = H >> 2;
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nb * B * H4) return;
    const bool second = idx >= B * H4;            // rows [B, 2B): the second decode of a pair
    if (second) idx -= B * H4;
    const float* __restrict__ pre = pre_d.sel(second);
    const float* __restrict__ c_prev = c_prev_d.sel(second);
    const uint8_t* __restrict__ keep = keep_d.sel(second);
    float* __restrict__ h_new = h_new_d.sel(second);
    float* __restrict__ c_new = c_new_d.sel(second);
    float* __restrict__ out = out_d.sel(second);
    const int b = idx / H4, j = idx % H4;
    const f32x4* p = reinterpret_cast<const f32x4*>(pre + (size_t)b * 5 * H);
    const f32x4 pi = p[j], pf = p[H4 + j], po = p[2 * H4 + j], pa = p[3 * H4 + j], pb = p[4 * H4 + j];
    const f32x4 cp = reinterpret_cast<const f32x4*>(c_prev)[idx];
    f32x4 hn, cn, o;
    uint32_t kp = 0x01010101u;
    if (keep) kp = *reinterpret_cast<const uint32_t*>(keep + (size_t)idx * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float ig = fast_sigmoid(pi[e]), fg = fast_sigmoid(pf[e]), og = fast_sigmoid(po[e]);
        const float g = fmaxf(pa[e], pb[e]);
        const float c2 = fg * cp[e] + ig * g;
        const float h2 = og * fast_tanh(c2);
        cn[e] = c2;
        hn[e] = h2;
        const float kf = (float)((kp >> (8 * e)) & 0xffu);
        o[e] = keep ? h2 * (kf * scale) : h2;
    }
    reinterpret_cast<f32x4*>(h_new)[idx] = state_dropped ? o : hn;
    reinterpret_cast<f32x4*>(c_new)[idx] = cn;
    reinterpret_cast<f32x4*>(out)[idx] = o;
}


// ---------------------------------------------------------------------------------------------
// Fused a2c product + maxout-LSTM cell (flagship width H = 512): in_transform += a2c(att_res) and the whole pointwise
// cell of Att2in2Core.forward (models/AttModel.py:515-529) in ONE launch.  A workgroup owns a 32-row strip and one
// 16-wide tile j of hidden units: it computes the two a2c tiles the units need (weight rows j and H+j, i.e. the two
// maxout halves) on v_mfma_f32_16x16x4_f32 with K = H split over its 8 waves, sums the partial tiles through LDS and
// applies the cell to the outputs it holds, reading the i/f/o pre-activations the i2h/h2h product left in `pre`.
// The summed candidates are written back into pre[:, 3H:5H] (the backward pass reads them).  Rows [B0, 2*B0) of a
// decode pair go through the .b pointers.
// ---------------------------------------------------------------------------------------------
typedef float f32x4acc __attribute__((ext_vector_type(4)));
// One (32-row strip, 16-unit tile) of the fused a2c product + cell: the body of a2c_cell_fused_kernel and the second phase of
// attn_a2c_cell_kernel.  `gstrip` counts the strips of decode a, then those of decode b; only `worker` waves (the first KS of the
// workgroup) compute, every wave meets the barriers.  wait_att_res(): called by every thread right before att_res is read - the
// fused kernel waits there for the strip's attention rows (and returns NaN when it gave up), SC1 then reads them past the caches.
template <int GPS, int KS, bool SC1, typename WaitF>   // H = 16*GPS*KS
__device__ __forceinline__ void a2c_cell_tile(int gstrip, int jt, bool worker, float* red, Dual<const float> att_res_d,
                                              const float* __restrict__ Wa, const float* __restrict__ ba, Dual<float> pre_d,
                                              Dual<const float> c_prev_d, Dual<const uint8_t> keep_d, float scale,
                                              Dual<float> h_new_d, Dual<float> c_new_d, Dual<float> out_d, int B0, int H,
                                              WaitF wait_att_res) {
    static_assert(KS == 8, "8 accumulator registers (2 row tiles x 4) dealt one per wave");
    const int tid = threadIdx.x, lane = tid & 63, ks = worker ? tid >> 6 : 0;
    const int li = lane & 15, lq = lane >> 4;
    const int strips_per = (B0 + 31) / 32;
    const bool second = gstrip >= strips_per;
    const int strip = second ? gstrip - strips_per : gstrip;
    const float* __restrict__ att_res = att_res_d.sel(second);
    float* __restrict__ pre = pre_d.sel(second);
    const float* __restrict__ c_prev = c_prev_d.sel(second);
    const uint8_t* __restrict__ keep = keep_d.sel(second);
    float* __restrict__ h_new = h_new_d.sel(second);
    float* __restrict__ c_new = c_new_d.sel(second);
    float* __restrict__ out = out_d.sel(second);
    const int m0 = strip * 32;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const int col = jt * 16 + li;
    const int orow = m0 + 16 * (ks >> 2) + 4 * lq + (ks & 3);     // the output this wave finishes after the cross-wave sums
    const int orc = orow < B0 ? orow : B0 - 1;
    // epilogue operands first (used last)
    float pi = 0.f, pf = 0.f, po = 0.f, pa = 0.f, pb = 0.f, cp = 0.f, kf = 1.0f;
    // both maxout halves' weight tiles are requested up front (the second one used to be requested behind the first one's MFMAs:
    // a dependent round trip), both products run back to back, ONE barrier covers both cross-wave sums; per element the
    // arithmetic and its order are unchanged
    f32x4 bf[2][GPS];
    if (worker) {
        const float* prow = pre + (size_t)orc * 5 * H;
        pi = prow[col]; pf = prow[H + col]; po = prow[2 * H + col]; pa = prow[3 * H + col]; pb = prow[4 * H + col];
        cp = c_prev[(size_t)orc * H + col];
        kf = keep ? (float)keep[(size_t)orc * H + col] * scale : 1.0f;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const float* wrow = Wa + ((size_t)g * H + col) * H;
#pragma unroll
            for (int i = 0; i < GPS; ++i) bf[g][i] = *reinterpret_cast<const f32x4*>(wrow + 16 * (ks * GPS + i) + 4 * lq);
        }
    }
    const float poison = wait_att_res();
    if (worker) {
        f32x4 af[2][GPS];
        __amdgpu_buffer_rsrc_t r_res;
        if (SC1) r_res = uniform_rsrc_f32(att_res, (size_t)B0 * H * sizeof(float));
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int m = m0 + 16 * rt + li;
            const int mc = m < B0 ? m : B0 - 1;
#pragma unroll
            for (int i = 0; i < GPS; ++i) {
                const size_t off = (size_t)mc * H + 16 * (ks * GPS + i) + 4 * lq;
                const f32x4 a = SC1 ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_res, (int)(off * 4), 0, 16))
                                    : *reinterpret_cast<const f32x4*>(att_res + off);
                af[rt][i] = m < B0 ? a : z4;
            }
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f32x4acc acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < GPS; ++i)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][i][s], bf[g][i][s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][i][s], bf[g][i][s], acc1, 0, 0, 0);
                }
            float* rb = red + g * (KS * 8 * 64);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                rb[(ks * 8 + v) * 64 + lane] = acc0[v];
                rb[(ks * 8 + 4 + v) * 64 + lane] = acc1[v];
            }
        }
    }
    __syncthreads();
    if (!worker) return;
    float av[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const float* rb = red + g * (KS * 8 * 64);
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < KS; ++w) v += rb[(w * 8 + ks) * 64 + lane];
        av[g] = v + ba[g * H + col];
    }
    if (orow < B0) {
        const float a = pa + av[0], b = pb + av[1];               // in_transform halves (:521-522), kept for the backward pass
        float* pw = pre + (size_t)orow * 5 * H;
        pw[3 * H + col] = a;
        pw[4 * H + col] = b;
        const float ig = fast_sigmoid(pi), fg = fast_sigmoid(pf), og = fast_sigmoid(po);
        float c2 = fg * cp + ig * fmaxf(a, b);                    // :523-526
        float h2 = og * fast_tanh(c2);                            // :527
        if (SC1 && poison != 0.f) { c2 = poison; h2 = poison; }
        c_new[(size_t)orow * H + col] = c2;
        h_new[(size_t)orow * H + col] = h2;
        out[(size_t)orow * H + col] = keep ? h2 * kf : h2;        // :529
    }
}

template <int GPS, int KS>   // H = 16*GPS*KS
__global__ __launch_bounds__(KS * 64) void a2c_cell_fused_kernel(Dual<const float> att_res_d, const float* __restrict__ Wa,
                                                             const float* __restrict__ ba, Dual<float> pre_d,
                                                             Dual<const float> c_prev_d, Dual<const uint8_t> keep_d, float scale,
                                                             Dual<float> h_new_d, Dual<float> c_new_d, Dual<float> out_d,
                                                             int B0, int nb, int H, Dual<const int32_t> live) {
    __shared__ float red[2 * KS * 8 * 64];
    if (live.a && *live.a == 0 && (!live.b || *live.b == 0)) return;   // every caption has ended (AttModel.py:401-408)
    const int tiles_j = H / 16;
    a2c_cell_tile<GPS, KS, false>((int)blockIdx.x / tiles_j, (int)blockIdx.x % tiles_j, true, red, att_res_d, Wa, ba, pre_d, c_prev_d,
                                  keep_d, scale, h_new_d, c_new_d, out_d, B0, H, [] { return 0.f; });
}

// ---------------------------------------------------------------------------------------------
// (r4) Attention + att2ctx product + cell of a decode step in ONE launch (flagship width H = 512, all workgroups resident): workgroup
// i first runs the attention of row i exactly as attn_fwd_cols_kernel does, hands att_res over inside the launch (write-through
// stores, drain, barrier, ONE counter add per workgroup on its 32-row strip's counter), then - as tile i of the
// (strip, 16-unit tile) grid of a2c_cell_fused_kernel - waits for its strip's rows (one poller, bounded: cic_common.h) and runs that
// kernel's body on sc1 loads.  Same arithmetic in the same order as the two launches: bit-identical activations and tokens.  What
// the fusion saves is one dependent launch boundary per step (~1 us) and the refetch of the cell's operands behind the attention
// (they are requested before the wait).  With 8 strips in a 256-workgroup grid a strip's rows and tiles share one XCD.
// A workgroup that gives up raises the error / status words and writes NaN into h, c and out of its tile.
// ---------------------------------------------------------------------------------------------
template <typename ST>
struct AttnCellArgs {
    Dual<const float> att_h;
    Dual<const ST> p_att, att;
    const float *w_alpha, *b_alpha, *masks;
    Dual<float> att_res, alpha, dot;
    const float *Wa, *ba;
    Dual<float> pre;
    Dual<const float> c_prev;
    Dual<const uint8_t> keep;
    float scale;
    Dual<float> h_new, c_new, out;
    Dual<const int32_t> live;
    unsigned* cnt;                // [nb * strips] arrival counters of THIS step (zeroed once per decode)
    HandoffGuard hg;
    int B0, nb, K;
};
template <int JMAX, typename ST>
__global__ __launch_bounds__(1024) void attn_a2c_cell_kernel(AttnCellArgs<ST> a) {
    constexpr int H = 512, TJ = H / 16;
    __shared__ float sp[16 * 64];
    __shared__ float red[2 * 8 * 8 * 64];
    __shared__ int ok_s;
    if (a.live.a && *a.live.a == 0 && (!a.live.b || *a.live.b == 0)) return;   // every caption has ended (grid-uniform)
    const int tid = threadIdx.x;
    const int rows = a.nb * a.B0, strips_per = (a.B0 + 31) / 32, nstrips = a.nb * strips_per;
    // workgroup -> (attention row, cell tile).  8 strips of 32 rows in a 256-workgroup grid: strip x = the workgroups of XCD x
    int row = blockIdx.x, gstrip = blockIdx.x / TJ, jt = blockIdx.x % TJ;
    if (gridDim.x == 256 && nstrips == 8 && (a.B0 & 31) == 0) {
        gstrip = blockIdx.x & 7; jt = blockIdx.x >> 3;
        row = 32 * gstrip + jt;
    }
    if (tid == 0) ok_s = 1;
    if (row < rows) {
        attn_cols_row<JMAX, 1, ST, true>(row, (int)blockIdx.x, sp, a.att_h, a.p_att, a.att, a.w_alpha, a.b_alpha, a.masks, a.att_res,
                                         a.alpha, a.dot, a.B0, a.K, H, 1, 0.f);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // EVERY storing wave drains before the signal
        __syncthreads();
        const bool second = row >= a.B0;
        const int rs = (second ? strips_per : 0) + (second ? row - a.B0 : row) / 32;
        if (tid == 0) handoff_arrive(a.cnt + rs, a.hg);
    }
    if (gstrip >= nstrips) return;
    const int s_local = gstrip % strips_per;
    const unsigned want = (unsigned)min(32, a.B0 - 32 * s_local);
    a2c_cell_tile<4, 8, true>(gstrip, jt, tid < 512, red, Dual<const float>{a.att_res.a, a.att_res.b}, a.Wa, a.ba, a.pre, a.c_prev, a.keep,
                              a.scale, a.h_new, a.c_new, a.out, a.B0, H, [&] {
                                  if (tid == 0) ok_s = handoff_poll(a.cnt + gstrip, want, a.hg);
                                  __syncthreads();
                                  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // compiler only: no load of handed-off bytes above the poll
                                  return ok_s ? 0.f : __builtin_nanf("");
                              });
}

// ---------------------------------------------------------------------------------------------
// The teacher-forced recurrence (AttModel.forward with ss_prob = 0, models/AttModel.py:103-148) in ONE launch.
// With the fed tokens known in advance nothing in the recurrence waits for a logit: x_t i2h^T of ALL steps is one batched
// product before the loop, and what remains per step - h h2h^T and h h2att^T, the attention, att_res a2c^T and the cell - is
// the forward twin of spk_bptt_seq_kernel: 16-row strips x 16-unit tiles, one workgroup per CU, all resident,
//   * the weight rows of the workgroup's units stay in it as MFMA B fragments (v_mfma_f32_16x16x4_f32), K = H split over
//     8 waves: the five h2h tiles and the h2att tile in registers (96 VGPRs per lane), the two a2c tiles in LDS (64 KB);
//   * the cell state c never leaves the lane that owns (row, unit); h_t, the attention query and the attention result travel
//     through the activation slabs the backward pass reads anyway, with three in-strip hand-offs per step (write-through
//     stores, drain, barrier, counter add + one poller, barrier, sc1 loads; 32 workgroups of a strip on ONE XCD at B = 128);
//   * the attention of the strip's 16 images runs on its even workgroups (8 waves x 64 columns, two passes of 36 registers:
//     p_att for the scores - requested before the first hand-off is waited for -, then att for the weighted sum).
// Per step: 3 launches of 20 + 9 + 9 us become ~13 us.  Every spin is bounded (1 s): a workgroup that gives up raises *err and
// poisons what it produces with NaN.
// ---------------------------------------------------------------------------------------------
struct TeacherSeqArgs {
    const float *h2h_w, *h2att_w, *h2att_b, *a2c_w, *a2c_b, *alpha_w, *alpha_b;
    const float *p_att, *att, *masks;                         // [B,K,H] x2, [B,K] or null
    const uint8_t* out_keep;                                  // [T,B,H] or null
    float *pre_all, *h_all, *c_all, *att_h_all, *att_res_all, *alpha_all, *dot_all, *out_all;   // the decode's activation slabs
    unsigned* cnt;                                            // [strips][T][3] counters (zeroed by the launcher)
    HandoffGuard hg;                                          // error word behind them, status word, spin bound (cic_common.h)
    float scale;
    int B, K, T;
    int row0, row_end;                                        // rows [row0, row_end) of the batch: one launch per row block
};
constexpr size_t TEACHER_LDS_BYTES = sizeof(float) * ((size_t)8 * 8 * 64 * 4 + 8 * 6 * 4 * 64 + 8 * 64);
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x8_t __attribute__((ext_vector_type(8)));
// eight consecutive f32 -> one bf16 MFMA fragment (round to nearest even; a plain cast keeps a NaN a NaN)
__device__ __forceinline__ bf16x8_t to_bf16x8(const f32x4 lo, const f32x4 hi) {
    const f32x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_convertvector(v, bf16x8_t);
}
// BF (r4, compute_dtype bf16 = BASELINE configs[1]'s "bf16"): the stationary weight tiles are held as bf16 MFMA fragments
// (v_mfma_f32_16x16x32_bf16: 8 k per lane and register quad - ALL eight tiles of a workgroup fit in 64 VGPRs, none in LDS), the
// rows handed over between workgroups (h, att_res) are rounded to bf16 as they are loaded, accumulation stays f32: a step's
// products are 16 bf16 MFMAs of 16 cycles instead of 128 f32 MFMAs of 32.  Cell, softmax and the slabs the backward pass reads
// stay f32.  Same tiling, hand-offs and output order as the f32 form.
template <int KS, bool BF>
__global__ __launch_bounds__(KS * 64) void spk_teacher_seq_kernel(TeacherSeqArgs a) {
    static_assert(KS == 8, "K = 512 split over 8 waves: 4 k groups of 16 each");
    constexpr int H = 512, H5 = 5 * H, TJ = H / 16, GP = 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x4* wl = reinterpret_cast<f32x4*>(lds);            // [KS][2 a2c tiles][GP][64]
    float* red = lds + (size_t)KS * 2 * GP * 64 * 4;      // [KS][6 tiles][4][64]
    float* sp = red + KS * 6 * 4 * 64;                    // [KS][64] per-wave partial scores
    __shared__ int ok_s;
    const int tid = threadIdx.x, lane = tid & 63, ks = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const bool owner = ks < 4;                            // waves 0-3 finish the 16 x 16 outputs: register ks of every tile
    const int orow = m0 + 4 * lq + (ks & 3);
    const int orc = orow < RE ? orow : RE - 1;
    const int mc = min(m0 + li, RE - 1);                  // A rows; rows past the block repeat its last row: their sums are never stored
    unsigned* cnt = a.cnt + (size_t)(a.row0 / 16 + strip) * T * 3;
    // ---- the weight tiles, once: B fragments (k = 16 (4 ks + i) + 4 lq + s, n = unit li of the tile) -----------------------------
    f32x4 wh[BF ? 1 : 6][GP];
    bf16x8_t whb[BF ? 8 : 1][2];                          // BF: tiles 0-4 h2h, 5 h2att, 6-7 a2c; k-steps of 32 inside the wave's 64 k
#pragma unroll
    for (int tau = 0; tau < 6; ++tau) {
        const float* wrow = tau < 5 ? a.h2h_w + ((size_t)tau * H + col) * H : a.h2att_w + (size_t)col * H;
        if (BF) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float* q = wrow + 64 * ks + 32 * j + 8 * lq;
                whb[BF ? tau : 0][j] = to_bf16x8(*reinterpret_cast<const f32x4*>(q), *reinterpret_cast<const f32x4*>(q + 4));
            }
        } else {
#pragma unroll
            for (int i = 0; i < GP; ++i) wh[BF ? 0 : tau][i] = *reinterpret_cast<const f32x4*>(wrow + 16 * (GP * ks + i) + 4 * lq);
        }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const float* wrow = a.a2c_w + ((size_t)g * H + col) * H;
        if (BF) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float* q = wrow + 64 * ks + 32 * j + 8 * lq;
                whb[BF ? 6 + g : 0][j] = to_bf16x8(*reinterpret_cast<const f32x4*>(q), *reinterpret_cast<const f32x4*>(q + 4));
            }
        } else {
#pragma unroll
            for (int i = 0; i < GP; ++i)
                wl[((ks * 2 + g) * GP + i) * 64 + lane] = *reinterpret_cast<const f32x4*>(wrow + 16 * (GP * ks + i) + 4 * lq);
        }
    }
    const float b_att = a.h2att_b[col], b_a0 = a.a2c_b[col], b_a1 = a.a2c_b[H + col];
    // attention: this wave's 64 columns (float4 column col4), lane -> (ac = column quad, rg = region group)
    const int ac = lane & 15, rg = lane >> 4;
    const int col4 = 16 * ks + ac;
    const f32x4 wa = reinterpret_cast<const f32x4*>(a.alpha_w)[col4];
    const float b_alpha = a.alpha_b[0];
    const int img = m0 + (jt >> 1);
    const bool att_wg = (jt & 1) == 0;
    const int imc = img < RE ? img : RE - 1;
    const bool do_att = att_wg && img < RE;
    auto uniform_rsrc = [](const float* ptr, size_t bytes) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(ptr);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo), 0, (int)bytes, 0x00020000);
    };
    const auto r_h = uniform_rsrc(a.h_all, (size_t)(T + 1) * B * H * sizeof(float));
    const auto r_ah = uniform_rsrc(a.att_h_all, (size_t)T * B * H * sizeof(float));
    const auto r_res = uniform_rsrc(a.att_res_all, (size_t)T * B * H * sizeof(float));
    const auto r_patt = uniform_rsrc(a.p_att + (size_t)imc * K * H, (size_t)K * H * sizeof(float));
    const auto r_att = uniform_rsrc(a.att + (size_t)imc * K * H, (size_t)K * H * sizeof(float));
    if (tid == 0) ok_s = 1;                               // sticky: a workgroup that has given up once does not wait again
    __syncthreads();
    float c_state = 0.f;                                  // c_{t-1} of this lane's (row, unit) (owners); init_hidden: zeros
    float poison = 0.f;
    auto publish = [&](unsigned* c, bool add) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // EVERY storing wave drains before the signal
        __syncthreads();
        if (tid == 0 && add) handoff_arrive(c, a.hg);
    };
    auto wait_for = [&](unsigned* c, unsigned target) {
        if (tid == 0 && ok_s) ok_s = handoff_poll(c, target, a.hg);
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    // compiler only: no load of handed-off bytes above the poll
        if (!ok_s) poison = __builtin_nanf("");
    };
    for (int t = 0; t < T; ++t) {
        // per-lane index values made opaque once per step (see spk_bptt_seq_kernel: keeps address arithmetic out of registers)
        int lq_t = lq, mc_t = mc, orc_t = orc, orow_t = orow, col_t = col, col4_t = col4, rg_t = rg, lane_t = lane;
        asm volatile("" : "+v"(lq_t), "+v"(mc_t), "+v"(orc_t), "+v"(orow_t), "+v"(col_t), "+v"(col4_t), "+v"(rg_t), "+v"(lane_t));
        const size_t rowH = (size_t)t * B * H;
        const int so_h = __builtin_amdgcn_readfirstlane((int)(rowH * sizeof(float)));      // slab t of the [B,H] slabs
        float* pre = a.pre_all + (size_t)t * B * H5 + (size_t)orc_t * H5 + col_t;
        // this step's x_t i2h^T + bias (written before the launch) and dropout mask: requested first, used last
        float px[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        float kf = 1.0f;
        if (owner) {
#pragma unroll
            for (int g = 0; g < 5; ++g) px[g] = pre[g * H];
            if (a.out_keep) kf = (float)a.out_keep[rowH + (size_t)orc_t * H + col_t] * a.scale;
        }
        // ---- 1. h_{t-1} h2h^T (five gate tiles) and the attention query h_{t-1} h2att^T + b ---------------------------------------
        float hs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (t > 0) {
            wait_for(cnt + (t - 1) * 3 + 2, TJ);            // h_{t-1} of the strip (slab t of h_all)
            f32x4 af[GP];
            // (BF: the same four 16-byte loads per lane, as two runs of eight consecutive k: 64 ks + 32 j + 8 lq + 0..7)
#pragma unroll
            for (int i = 0; i < GP; ++i)
                af[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    r_h, (int)(((size_t)mc_t * H + (BF ? 64 * ks + 32 * (i >> 1) + 8 * lq_t + 4 * (i & 1) : 16 * (GP * ks + i) + 4 * lq_t)) * 4),
                    so_h, 16));
            f32x4acc acc[6];
#pragma unroll
            for (int tau = 0; tau < 6; ++tau) acc[tau] = f32x4acc{0.f, 0.f, 0.f, 0.f};
            if (BF) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bf16x8_t ab = to_bf16x8(af[2 * j], af[2 * j + 1]);
#pragma unroll
                    for (int tau = 0; tau < 6; ++tau)
                        acc[tau] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, whb[BF ? tau : 0][j], acc[tau], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < GP; ++i)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int tau = 0; tau < 6; ++tau)
                            acc[tau] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], wh[BF ? 0 : tau][i][s], acc[tau], 0, 0, 0);
            }
#pragma unroll
            for (int tau = 0; tau < 6; ++tau)
#pragma unroll
                for (int v = 0; v < 4; ++v) red[((ks * 6 + tau) * 4 + v) * 64 + lane_t] = acc[tau][v];
            __syncthreads();
            if (owner) {
#pragma unroll
                for (int tau = 0; tau < 6; ++tau) {
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < KS; ++w) v += red[((w * 6 + tau) * 4 + ks) * 64 + lane_t];
                    hs[tau] = v;
                }
            }
        }
        if (owner && orow_t < RE) {
            float q = hs[5] + b_att;
            if (poison != 0.f) q = poison;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, q), r_ah, (int)(((size_t)orow_t * H + col_t) * 4), so_h, 16);
        }
        publish(cnt + t * 3 + 0, true);
        // ---- 2. attention of image img on the even workgroups (attn_fwd_cols_kernel, 8 waves x 64 columns) ---------------------------
        if (att_wg) {
            constexpr int JMAX = 9;                          // regions 4 j + rg, K <= 36
            const int vo = (rg_t * H + 4 * col4_t) * 4;
            f32x4 rv[JMAX];
#pragma unroll
            for (int j = 0; j < JMAX; ++j)
                rv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_patt, vo, 4 * j * H * 4, 0));
            wait_for(cnt + t * 3 + 0, TJ);
            const f32x4 ah = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_ah, (int)(((size_t)imc * H + 4 * col4_t) * 4), so_h, 16));
#pragma unroll
            for (int j = 0; j < JMAX; ++j) {
                float part = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) part += wa[e] * fast_tanh(rv[j][e] + ah[e]);
                part = sum8_dpp(part);
                part += dpp_f32<DPP_ROW_MIRROR>(part);
                const int k = 4 * j + rg_t;
                if (ac == 0 && k < K) sp[ks * 64 + k] = part;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < JMAX; ++j)
                rv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_att, vo, 4 * j * H * 4, 0));
            __syncthreads();                                 // (workgroup-uniform branch)
            float dot = -INFINITY;
            if (lane < K) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < KS; ++q) s += sp[q * 64 + lane];
                dot = s + b_alpha;
            }
            const float mx = wave_max_fast(dot);
            const float ex = lane < K ? __expf(dot - mx) : 0.f;
            const float sum = wave_sum_fast(ex);
            float al = ex * (1.0f / sum);
            if (a.masks) {   // weight = weight * mask; weight /= weight.sum()   (AttModel.py:481-483)
                al = lane < K ? al * a.masks[(size_t)imc * K + lane] : 0.f;
                const float ms = wave_sum_fast(al);
                al = al * (1.0f / ms);
            }
            if (poison != 0.f) al = poison;
            if (ks == 0 && lane < K && do_att) {
                a.alpha_all[((size_t)t * B + img) * K + lane] = al;
                a.dot_all[((size_t)t * B + img) * K + lane] = dot;
            }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < JMAX; ++j) {
                const float aj = __shfl(al, 4 * j + rg_t, 64);      // lanes >= K hold 0
                acc += aj * rv[j];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[e];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                acc[e] = v;
            }
            if (rg_t == 0 && do_att)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc), r_res, (int)(((size_t)img * H + 4 * col4_t) * 4), so_h, 16);
            publish(cnt + t * 3 + 1, true);
        }
        // ---- 3. in_transform += att_res a2c^T + b, the cell, h_t --------------------------------------------------------------------------
        wait_for(cnt + t * 3 + 1, TJ / 2);
        float av0 = 0.f, av1 = 0.f;
        {
            f32x4 af[GP];
#pragma unroll
            for (int i = 0; i < GP; ++i)
                af[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    r_res, (int)(((size_t)mc_t * H + (BF ? 64 * ks + 32 * (i >> 1) + 8 * lq_t + 4 * (i & 1) : 16 * (GP * ks + i) + 4 * lq_t)) * 4),
                    so_h, 16));
            f32x4acc a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
            if (BF) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bf16x8_t ab = to_bf16x8(af[2 * j], af[2 * j + 1]);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, whb[BF ? 6 : 0][j], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, whb[BF ? 7 : 0][j], a1, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < GP; ++i) {
                    const f32x4 b0 = wl[((ks * 2 + 0) * GP + i) * 64 + lane_t], b1 = wl[((ks * 2 + 1) * GP + i) * 64 + lane_t];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], b0[s], a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], b1[s], a1, 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                red[((ks * 6 + 0) * 4 + v) * 64 + lane_t] = a0[v];
                red[((ks * 6 + 1) * 4 + v) * 64 + lane_t] = a1[v];
            }
            __syncthreads();
            if (owner) {
#pragma unroll
                for (int w = 0; w < KS; ++w) {
                    av0 += red[((w * 6 + 0) * 4 + ks) * 64 + lane_t];
                    av1 += red[((w * 6 + 1) * 4 + ks) * 64 + lane_t];
                }
            }
        }
        if (owner) {
            const float pi = px[0] + hs[0], pf = px[1] + hs[1], po = px[2] + hs[2];
            const float pa = (px[3] + hs[3]) + (av0 + b_a0), pb = (px[4] + hs[4]) + (av1 + b_a1);     // in_transform halves (:521-522)
            const float ig = fast_sigmoid(pi), fg = fast_sigmoid(pf), og = fast_sigmoid(po);
            float c2 = fg * c_state + ig * fmaxf(pa, pb);          // :523-526
            float h2 = og * fast_tanh(c2);                         // :527
            if (poison != 0.f) { c2 = poison; h2 = poison; }
            c_state = c2;
            if (orow_t < RE) {
                pre[0] = pi; pre[H] = pf; pre[2 * H] = po; pre[3 * H] = pa; pre[4 * H] = pb;      // kept for the backward pass
                const size_t e = (size_t)orow_t * H + col_t;
                a.c_all[rowH + (size_t)B * H + e] = c2;
                a.out_all[rowH + e] = a.out_keep ? h2 * kf : h2;  // :529
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, h2), r_h, (int)(e * 4),
                                                      so_h + (int)((size_t)B * H * sizeof(float)), 16);
            }
        }
        if (t + 1 < T) publish(cnt + t * 3 + 2, true);
    }
}

// ---------------------------------------------------------------------------------------
// K7 token embedding: x = dropout(relu(E[it]))   (models/AttModel.py:74-76,399)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_fwd_kernel(const float* __restrict__ E, Dual<const int32_t> it_d,
                                                        Dual<const uint8_t> keep_d, float scale, Dual<float> x_d, int B,
                                                        int nb, int Ed, int plain) {
    // plain: a bare embedding row (FCModel, models/FCModel.py:66,119): no ReLU, no dropout
    const int E4 = Ed >> 2;
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nb * B * E4) return;
    const bool second = idx >= B * E4;
    if (second) idx -= B * E4;
    const int32_t* __restrict__ it = it_d.sel(second);
    const uint8_t* __restrict__ keep = keep_d.sel(second);
    float* __restrict__ x = x_d.sel(second);
    const int b = idx / E4, j = idx % E4;
    f32x4 v = reinterpret_cast<const f32x4*>(E + (size_t)it[b] * Ed)[j];
    uint32_t kp = 0x01010101u;
    if (keep) kp = *reinterpret_cast<const uint32_t*>(keep + (size_t)idx * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float kf = (float)((kp >> (8 * e)) & 0xffu);
        const float r = plain ? v[e] : fmaxf(v[e], 0.f);
        v[e] = keep ? r * (kf * scale) : r;
    }
    reinterpret_cast<f32x4*>(x)[idx] = v;
}

// y = x * keep * scale (dropout of the embedded regions, models/AttModel.py:82-85)
__global__ __launch_bounds__(256) void apply_keep_kernel(const float* __restrict__ x, const uint8_t* __restrict__ keep,
                                                         float scale, float* __restrict__ y, int64_t n4) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n4) return;
    f32x4 v = reinterpret_cast<const f32x4*>(x)[idx];
    if (keep) {
        const uint32_t kp = *reinterpret_cast<const uint32_t*>(keep + idx * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * ((float)((kp >> (8 * e)) & 0xffu) * scale);
    }
    reinterpret_cast<f32x4*>(y)[idx] = v;
}

// ... for the two decodes of a pair in ONE launch (blockIdx.y picks the decode: same x, its own mask and output)
__global__ __launch_bounds__(256) void apply_keep2_kernel(const float* __restrict__ x, Dual<const uint8_t> keep, float scale,
                                                          Dual<float> y, int64_t n4) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n4) return;
    const uint8_t* kq = keep.sel(blockIdx.y != 0);
    f32x4 v = reinterpret_cast<const f32x4*>(x)[idx];
    if (kq) {
        const uint32_t kp = *reinterpret_cast<const uint32_t*>(kq + idx * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * ((float)((kp >> (8 * e)) & 0xffu) * scale);
    }
    reinterpret_cast<f32x4*>(y.sel(blockIdx.y != 0))[idx] = v;
}

// Ragged region counts (att_masks != None): pack_wrapper (models/AttModel.py:30-51) embeds only the first
// len_b = sum_k att_masks[b,k] region rows of image b and pads the rest back with ZEROS, so rows k >= len_b of the
// embedded features are 0 (not relu(bias)):  y[b,k,:] = k < len_b ? x * keep * scale : 0
__global__ __launch_bounds__(256) void att_keep_rows_kernel(const float* __restrict__ x, const uint8_t* __restrict__ keep,
                                                            float scale, const float* __restrict__ masks,
                                                            float* __restrict__ y, int B, int K, int H4) {
    const int row = blockIdx.x;                       // one (image, region) row per workgroup
    const int b = row / K, k = row % K;
    float len = 0.f;
    for (int j = 0; j < K; ++j) len += (float)(int)masks[(size_t)b * K + j];    // att_masks.data.long().sum(1)
    const bool valid = (float)k < len;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (int c = threadIdx.x; c < H4; c += 256) {
        const size_t idx = (size_t)row * H4 + c;
        f32x4 v = reinterpret_cast<const f32x4*>(x)[idx];
        if (keep) {
            const uint32_t kp = *reinterpret_cast<const uint32_t*>(keep + idx * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * ((float)((kp >> (8 * e)) & 0xffu) * scale);
        }
        reinterpret_cast<f32x4*>(y)[idx] = valid ? v : z4;
    }
}

// compute_dtype bf16: x <- bf16(x) widened back to f32 (in place) and the packed bf16 copy the attention kernel streams
__global__ __launch_bounds__(256) void round_pack_bf16_kernel(float* __restrict__ x, uint16_t* __restrict__ packed, int64_t n4) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n4) return;
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[idx];
    const bf16x4_t h = __builtin_convertvector(v, bf16x4_t);          // round to nearest even
    reinterpret_cast<bf16x4_t*>(packed)[idx] = h;
    reinterpret_cast<f32x4*>(x)[idx] = __builtin_convertvector(h, f32x4);
}

// x = dropout(relu(xpre))  — self.relu_dropout on soft_vec @ embed (models/AttModel.py:77-78,396-397)
__global__ __launch_bounds__(256) void relu_keep_fwd_kernel(const float* __restrict__ xpre, const uint8_t* __restrict__ keep,
                                                            float scale, float* __restrict__ x, int64_t n4) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n4) return;
    f32x4 v = reinterpret_cast<const f32x4*>(xpre)[idx];
    uint32_t kp = 0x01010101u;
    if (keep) kp = *reinterpret_cast<const uint32_t*>(keep + idx * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float r = fmaxf(v[e], 0.f);
        v[e] = keep ? r * ((float)((kp >> (8 * e)) & 0xffu) * scale) : r;
    }
    reinterpret_cast<f32x4*>(x)[idx] = v;
}

// soft_out[t,b,:] = unfinished ? soft_raw[t,b,:] : one-hot(EOS = 0)    (models/AttModel.py:428-432)
__global__ __launch_bounds__(256) void soft_mask_kernel(const float* __restrict__ soft_raw, const int32_t* __restrict__ seq,
                                                        const int32_t* __restrict__ Lp, float* __restrict__ soft_out, int T,
                                                        int B, int V1) {
    const int row = blockIdx.x, t = row / B, b = row % B;
    const bool unf = t < *Lp && seq[(size_t)b * T + t] > 0;
    const float* src = soft_raw + (size_t)row * V1;
    float* dst = soft_out + (size_t)row * V1;
    for (int c = threadIdx.x; c < V1; c += 256) dst[c] = unf ? src[c] : (c == 0 ? 1.0f : 0.0f);
}

// ---------------------------------------------------------------------------------------
// K5b/K6 row kernel: log-softmax over the vocabulary + sampler + EOS bookkeeping
//   (models/AttModel.py:328-365,401-434,438-444; models/gumbel.py:6-30; models/multinomial.py:4-27)
// One workgroup per batch row; the row (<= RV*1024 floats) lives in registers between passes.
// ---------------------------------------------------------------------------------------
struct ArgMax {
    float v;
    int i;
};
__device__ __forceinline__ ArgMax amax_better(ArgMax a, ArgMax b) {
    // larger value wins; ties -> lowest index (torch.max on CPU)
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ ArgMax wave_argmax(ArgMax a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMax b;
        b.v = __shfl_xor(a.v, o, 64);
        b.i = __shfl_xor(a.i, o, 64);
        a = amax_better(a, b);
    }
    return a;
}
constexpr int SNW = 16;   // waves per row workgroup of the sampler kernel
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum_fast(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int w = 0; w < SNW; ++w) r += sh[w];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max_fast(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sh[0];
#pragma unroll
    for (int w = 1; w < SNW; ++w) r = fmaxf(r, sh[w]);
    return r;
}
// three block sums for the price of one barrier round (sh3: 3 * SNW floats)
__device__ __forceinline__ void block_sum3(float& a, float& b, float& c, float* sh3) {
    a = wave_sum_fast(a);
    b = wave_sum_fast(b);
    c = wave_sum_fast(c);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        sh3[w] = a; sh3[SNW + w] = b; sh3[2 * SNW + w] = c;
    }
    __syncthreads();
    float ra = 0.f, rb = 0.f, rc = 0.f;
#pragma unroll
    for (int w = 0; w < SNW; ++w) { ra += sh3[w]; rb += sh3[SNW + w]; rc += sh3[2 * SNW + w]; }
    a = ra; b = rb; c = rc;
}
__device__ __forceinline__ ArgMax block_argmax(ArgMax a, float* shv, int* shi) {
    a = wave_argmax(a);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        shv[threadIdx.x >> 6] = a.v;
        shi[threadIdx.x >> 6] = a.i;
    }
    __syncthreads();
    ArgMax r = {shv[0], shi[0]};
#pragma unroll
    for (int w = 1; w < SNW; ++w) r = amax_better(r, ArgMax{shv[w], shi[w]});
    return r;
}


template <int RV>   // RV float4 per thread: rows up to RV*4096 floats
__global__ __launch_bounds__(SNW * 64) void logsoftmax_sample_kernel(cic_sampler_args a0, cic_sampler_args a1) {
    constexpr int NT = SNW * 64;
    unsigned long long* stamps = CIC_STAMP_BUF(g_attn_stamps);          // diagnostics: [row][wave][8] phase stamps (100 MHz)
    if (stamps) stamps += ((size_t)blockIdx.x * SNW + (threadIdx.x >> 6)) * 8;
    int sidx = 0;
    auto stamp = [&]() { if (stamps && (threadIdx.x & 63) == 0 && sidx < 8) stamps[sidx] = __builtin_amdgcn_s_memrealtime(); ++sidx; };
    stamp();
    __shared__ float shf[SNW];
    __shared__ int shi[SNW];
    __shared__ float sh3[3 * SNW];
    // rows [0, a0.B) run the sampler of decode a, rows beyond it that of decode b (own mode, noise and outputs):
    // a uniform, field-wise select of the kernel arguments
    const bool second = (int)blockIdx.x >= a0.B;
    const cic_sampler_args a = second ? a1 : a0;
    const int b = second ? blockIdx.x - a0.B : blockIdx.x, tid = threadIdx.x;
    const int V1 = a.V1;
    float* row = a.logits + (size_t)b * a.ld;
    const bool vec = ((a.ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.logits) & 15) == 0);
    const int nq = (V1 + 3) >> 2;

    float x[RV][4];
    const int cons = (a.decoding_constraint && a.step >= 2) ? a.seq[(size_t)b * a.seq_ld + (a.step - 2)] : -1;
    // the noise row is needed only after the log-softmax: its loads go out together with the logits'
    const float* urow = a.U ? a.U + (size_t)b * a.ldu : nullptr;
    const bool ss_on = a.mode == CIC_SAMPLE_TEACHER && a.ss_u && a.ss_prob > 0.f;
    const bool gumbel_mode = a.mode == CIC_SAMPLE_GUMBEL_ST || a.mode == CIC_SAMPLE_GUMBEL_PS;
    const bool use_noise = a.mode == CIC_SAMPLE_TEACHER
                               ? (ss_on && !a.ss_pick)
                               : ((a.mode != CIC_SAMPLE_GREEDY && a.mode != CIC_SAMPLE_NONE) && !(a.pick && !gumbel_mode));
    float un[RV][4];
    float mx = -INFINITY;
    // Fast path (rows of whole, 16-byte aligned float4: the vocabulary rows of the engines): EVERY load of the row and
    // of its noise goes out before the first use, unconditionally (threads beyond the row re-read its last float4 and
    // mask the values afterwards).  Element loads under `col < V1` conditions compile to one load + s_waitcnt vmcnt(0)
    // per element, i.e. 15 serial memory round trips at RV = 3 (measured: 17.4 us per launch with them).
    const bool fast = vec && (V1 & 3) == 0 &&
                      (!use_noise || (((a.ldu & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.U) & 15) == 0)));
    if (fast) {
        f32x4 xv[RV], uv[RV];
#pragma unroll
        for (int r = 0; r < RV; ++r) {
            const int q = tid + NT * r;
            xv[r] = reinterpret_cast<const f32x4*>(row)[q < nq ? q : nq - 1];
        }
        if (use_noise) {
#pragma unroll
            for (int r = 0; r < RV; ++r) {
                const int q = tid + NT * r;
                uv[r] = reinterpret_cast<const f32x4*>(urow)[q < nq ? q : nq - 1];
            }
        } else {
#pragma unroll
            for (int r = 0; r < RV; ++r) uv[r] = f32x4{0.5f, 0.5f, 0.5f, 0.5f};
        }
#pragma unroll
        for (int r = 0; r < RV; ++r) {
            const int q = tid + NT * r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float xe = q < nq ? xv[r][e] : -INFINITY;
                if (4 * q + e == cons) xe = -INFINITY;        // decoding_constraint, AttModel.py:438-442
                x[r][e] = xe;
                un[r][e] = uv[r][e];                          // beyond the row: never used (z = -inf there)
                mx = fmaxf(mx, xe);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < RV; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = 4 * (tid + NT * r) + e;
                un[r][e] = (use_noise && (tid + NT * r) < nq && col < V1) ? urow[col] : 0.5f;
            }
#pragma unroll
        for (int r = 0; r < RV; ++r) {
            const int q = tid + NT * r;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[r][e] = -INFINITY;
            if (q < nq) {
                if (vec && 4 * q + 3 < V1) {
                    const f32x4 v = reinterpret_cast<const f32x4*>(row)[q];
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[r][e] = v[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (4 * q + e < V1) x[r][e] = row[4 * q + e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (4 * q + e == cons) x[r][e] = -INFINITY;   // decoding_constraint, AttModel.py:438-442
                    mx = fmaxf(mx, x[r][e]);
                }
            }
        }
    }
    if (stamps) { asm volatile("" :: "v"(mx)); stamp(); }
    mx = block_max(mx, shf);
    stamp();
    float se = 0.f;
#pragma unroll
    for (int r = 0; r < RV; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) se += __expf(x[r][e] - mx);   // exp(-inf) = 0 for padding
    se = block_sum(se, shf);
    stamp();
    const float lse = mx + logf(se);
    // log-probs back to memory (saved for the backward pass) and kept in registers
#pragma unroll
    for (int r = 0; r < RV; ++r) {
        const int q = tid + NT * r;
#pragma unroll
        for (int e = 0; e < 4; ++e) x[r][e] -= lse;
        if (fast) {
            if (q < nq) reinterpret_cast<f32x4*>(row)[q] = f32x4{x[r][0], x[r][1], x[r][2], x[r][3]};
        } else if (q < nq) {
            if (vec && 4 * q + 3 < V1) {
                f32x4 v = {x[r][0], x[r][1], x[r][2], x[r][3]};
                reinterpret_cast<f32x4*>(row)[q] = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * q + e < V1) row[4 * q + e] = x[r][e];
            }
        }
    }
    if (a.mode == CIC_SAMPLE_NONE) return;

    // ---- choose the token -------------------------------------------------------------
    const float inv_t = 1.0f / a.temp;
    ArgMax best = {-INFINITY, 0x7fffffff};
    float z[RV][4];
#pragma unroll
    for (int r = 0; r < RV; ++r) {
        const int q = tid + NT * r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int col = 4 * q + e;
            float zz = -INFINITY;
            if (q < nq && col < V1) {
                if (a.mode == CIC_SAMPLE_GREEDY) {
                    zz = x[r][e];
                } else if (gumbel_mode) {
                    zz = (x[r][e] + gumbel_from_u(un[r][e])) * inv_t;            // gumbel.py:13-15
                } else {   // multinomial flavours: Gumbel-max draw from softmax(logp / temp)
                    zz = x[r][e] * inv_t;
                    if (use_noise) zz += gumbel_from_u(un[r][e]);
                }
            }
            z[r][e] = zz;
            best = amax_better(best, ArgMax{zz, col});
        }
    }
    if (stamps) { asm volatile("" :: "v"(best.v)); stamp(); }
    best = block_argmax(best, shf, shi);
    stamp();
    int it = best.i;
    int it_feed = -1;                                   // teacher mode: token fed to the next step
    if (a.mode == CIC_SAMPLE_TEACHER) {
        const int target = (int)a.pick[b];
        const int drawn = a.ss_pick ? (int)a.ss_pick[b] : best.i;
        it_feed = (ss_on && a.ss_u[b] < a.ss_prob) ? drawn : target;   // AttModel.py:119-128
        it = target;                                     // the loss gathers log p(target)
    } else if (a.pick && a.mode != CIC_SAMPLE_GREEDY && !gumbel_mode) {
        it = (int)a.pick[b];
    }

    // sampled log-prob (gather) and the straight-through value v = (1 - y_it) + y_it
    float slp_part = 0.f;
#pragma unroll
    for (int r = 0; r < RV; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * (tid + NT * r) + e == it) slp_part = x[r][e];
    float slp;
    float v = 1.0f;
    float zm = 0.f, s2 = 1.f;
    if (gumbel_mode) {
        // y = softmax(z), z = (logp+g)/tau, max(z) = the arg-max value: slp, sum exp(z - zm) and z_it in ONE round
        zm = best.v;
        float zi = 0.f;
        s2 = 0.f;
#pragma unroll
        for (int r = 0; r < RV; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s2 += __expf(z[r][e] - zm);
                if (4 * (tid + NT * r) + e == it) zi = z[r][e];
            }
        block_sum3(slp_part, s2, zi, sh3);
        stamp();
        slp = slp_part;
        const float y = __expf(zi - zm) / s2;
        v = (1.0f - y) + y;    // (y_hard - y).detach() + y at the arg-max entry, gumbel.py:28
    } else if (a.mode == CIC_SAMPLE_MULTINOMIAL_ST) {
        // y = softmax(logp/tau)  (multinomial.py:10-15)
        slp = block_sum(slp_part, shf);
        float zi = -INFINITY;
        s2 = 0.f;
        float m2 = -INFINITY;
#pragma unroll
        for (int r = 0; r < RV; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) m2 = fmaxf(m2, x[r][e] * inv_t);
        zm = block_max(m2, shf);
#pragma unroll
        for (int r = 0; r < RV; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = x[r][e] * inv_t;
                s2 += __expf(t - zm);
                if (4 * (tid + NT * r) + e == it) zi = t;
            }
        s2 = block_sum(s2, shf);
        zi = block_max(zi, shf);
        const float y = __expf(zi - zm) / s2;
        v = (1.0f - y) + y;
    } else {
        slp = block_sum(slp_part, shf);
    }
    if (a.soft) {
        // partial sampling (gumbel_softmax.py:30-41, multinomial_soft.py:23-33): rows drawn with u < prob get the
        // straight-through row (exactly 0 off the token), the others the distribution itself
        const bool hard = a.ps_u && a.ps_prob > 0.f && a.ps_u[b] < a.ps_prob;
        float* srow = a.soft + (size_t)b * a.ld_soft;
        const float inv_s = 1.0f / s2;
        float v_it = 0.f;
#pragma unroll
        for (int r = 0; r < RV; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = 4 * (tid + NT * r) + e;
                if ((tid + NT * r) < nq && col < V1) {
                    const float yj = gumbel_mode ? __expf(z[r][e] - zm) * inv_s : __expf(x[r][e] * inv_t);
                    const float o = hard ? (col == it ? (1.0f - yj) + yj : 0.f) : yj;
                    srow[col] = o;
                    if (col == it) v_it = o;
                }
            }
        v = block_sum(v_it, shf);
    }

    // ---- next step's input: xt = dropout(relu(embed(it))) (AttModel.py:74-76,399), un-masked `it`, fused here so that
    // the decode loop needs no embedding launch after its first step
    if (a.emb_x) {
        const int tok = it_feed >= 0 ? it_feed : it;
        const int E4 = a.emb_dim >> 2;
        if (tid < E4) {
            f32x4 ev = reinterpret_cast<const f32x4*>(a.emb_w + (size_t)tok * a.emb_dim)[tid];
            uint32_t kp = 0x01010101u;
            if (a.emb_keep) kp = *reinterpret_cast<const uint32_t*>(a.emb_keep + (size_t)b * a.emb_dim + 4 * tid);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float kf = (float)((kp >> (8 * e)) & 0xffu);
                const float r = a.emb_plain ? ev[e] : fmaxf(ev[e], 0.f);
                ev[e] = a.emb_keep ? r * (kf * a.emb_scale) : r;
            }
            reinterpret_cast<f32x4*>(a.emb_x + (size_t)b * a.emb_dim)[tid] = ev;
        }
    }

    // ---- EOS bookkeeping (AttModel.py:401-434) -----------------------------------------
    if (tid == 0) {
        const int t = a.step;   // loop iteration of the reference (1-based for sampled tokens)
        int unf = (it > 0) ? 1 : 0;
        if (t > 1) unf = unf & a.unfinished[b];
        a.unfinished[b] = unf;
        a.it_next[b] = it_feed >= 0 ? it_feed : it;         // un-masked: embed(it) precedes the masking (:399)
        a.seq[(size_t)b * a.seq_ld + (t - 1)] = unf ? it : 0;   // it * unfinished (:409)
        a.slp[(size_t)b * a.seq_ld + (t - 1)] = slp;
        if (a.stv) a.stv[(size_t)b * a.seq_ld + (t - 1)] = unf ? v : 1.0f;   // finished rows -> exact EOS one-hot (:419-420)
        if (unf) atomicOr(a.any_unfinished + t, 1);
    }
    stamp();
}

// ---------------------------------------------------------------------------------------
// Row partials of the vocabulary from logits in memory (cic.h): the stand-in of the logit walker's fused epilogue for
// shapes that walker does not take.  One wave per (row, part); part p = columns [p*chunk, (p+1)*chunk).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ RowPart wave_merge_rowpart(RowPart rp, int mode, float inv_t) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        RowPart q;
        q.m1 = __shfl_xor(rp.m1, o, 64); q.s1 = __shfl_xor(rp.s1, o, 64);
        q.kbest = __shfl_xor(rp.kbest, o, 64); q.xbest = __shfl_xor(rp.xbest, o, 64);
        q.kidx = __shfl_xor(rp.kidx, o, 64); q.s2 = __shfl_xor(rp.s2, o, 64);
        rowpart_merge(rp, mode, inv_t, q);
    }
    return rp;
}

__global__ __launch_bounds__(256) void logit_partials_kernel(const float* __restrict__ logits, int M, int N, int ld,
                                                             cic_logit_epi_rows e, int np, int chunk) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int row = wid / np, p = wid % np;
    if (row >= M) return;
    const int cons = e.cons_seq ? e.cons_seq[(size_t)row * e.cons_ld + e.cons_col] : -1;
    RowPart rp;
    rp.init();
    const int c1 = min(N, (p + 1) * chunk);
    for (int c = p * chunk + lane; c < c1; c += 64) {
        float x = logits[(size_t)row * ld + c];
        if (c == cons) x = -INFINITY;                                   // decoding constraint, AttModel.py:438-442
        float g = 0.f;
        if (e.noise) {
            float u;
            if (e.U) u = e.U[(size_t)row * e.ldu + c];
            else {
                const uint64_t el = e.elem0 + (uint64_t)row * (uint64_t)e.ldu + (uint64_t)c;
                u = philox_uniform4(e.seed, el >> 2)[el & 3];
            }
            g = gumbel_from_u(u);
        }
        rowpart_add(rp, e.mode, e.inv_temp, x, g, c);
    }
    rp = wave_merge_rowpart(rp, e.mode, e.inv_temp);
    if (lane == 0) {
        const size_t plane = (size_t)e.part_rows * np;
        float* pp = e.part + (size_t)row * np + p;
        pp[0] = rp.m1; pp[plane] = rp.s1; pp[2 * plane] = rp.kbest; pp[3 * plane] = rp.xbest;
        pp[4 * plane] = __int_as_float(rp.kidx); pp[5 * plane] = rp.s2;
    }
}

// ---------------------------------------------------------------------------------------
// The sampler on row partials: log-sum-exp, token choice, gathered log-prob, straight-through value, EOS bookkeeping
// and the next step's embedding (models/AttModel.py:328-365,401-434,438-444; gumbel.py:13-30; multinomial.py:4-27) from
// the `np` partials of a row - no pass over the vocabulary.  One wave per batch row.  a.logits holds the RAW logits
// (read only for a token that is not the row's best key: an injected pick, a teacher target); the row's lse goes to
// `lse` so that the backward pass can normalise them.
// ---------------------------------------------------------------------------------------
struct FinishArgs {
    cic_sampler_args s;
    const float* part;
    int part_rows;
    float* lse;            // [B]
};
// The decode's length with the LAST sampler launch (r4; was a launch of its own): every workgroup counts itself in after its
// rows' bookkeeping (any_unfinished[step] is OR-ed with device-scope atomics); the one whose add comes last reads the flags of
// all steps (atomic = sc1 loads) and writes L as finalize_len_kernel does.  The counter is any_unfinished[0] of the first
// decode: cleared by decode_init_kernel, read by nobody else.
struct FinishLen {
    int32_t* counter;              // null: not the last step
    Dual<const int32_t> any_unf;   // [T+1] per decode (b null for a single decode)
    Dual<int32_t> L;
    int T;
};
__device__ __forceinline__ void finish_len_tail(const FinishLen& fl) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's bookkeeping atomics have left
    __syncthreads();
    __shared__ unsigned ticket_s;
    if (threadIdx.x == 0)
        ticket_s = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(fl.counter), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket_s != gridDim.x - 1u || threadIdx.x >= 2) return;
    // everyone has arrived: the counter goes back to 0 (the workspace then reads the same whether a decode ran alone or paired)
    if (threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<unsigned*>(fl.counter), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool second = threadIdx.x == 1;
    const int32_t* any_unf = fl.any_unf.sel(second);
    if (!any_unf) return;
    int l = fl.T;
    for (int t = 1; t <= fl.T; ++t)
        if (__hip_atomic_load(any_unf + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) { l = t - 1; break; }
    *fl.L.sel(second) = l;
}
__device__ __forceinline__ void sample_finish_rows(const FinishArgs& a0, const FinishArgs& a1, int np) {
    const int lane = threadIdx.x & 63;
    const int rg = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool second = rg >= a0.s.B;
    const FinishArgs fa = second ? a1 : a0;
    const cic_sampler_args& a = fa.s;
    const int b = second ? rg - a0.s.B : rg;
    if (b >= a.B) return;
    const float inv_t = 1.0f / a.temp;
    const size_t plane = (size_t)fa.part_rows * np;
    RowPart rp;
    rp.init();
    for (int p = lane; p < np; p += 64) {
        const float* pp = fa.part + (size_t)b * np + p;
        RowPart q;
        q.m1 = pp[0]; q.s1 = pp[plane]; q.kbest = pp[2 * plane]; q.xbest = pp[3 * plane];
        q.kidx = __float_as_int(pp[4 * plane]); q.s2 = pp[5 * plane];
        rowpart_merge(rp, a.mode, inv_t, q);
    }
    rp = wave_merge_rowpart(rp, a.mode, inv_t);
    if (a.mode == CIC_SAMPLE_NONE) {
        if (fa.lse && lane == 0) fa.lse[b] = rp.m1 + logf(rp.s1);
        return;
    }
    const RowChoice c = row_choice(a, rp, b);
    if (fa.lse && lane == 0) fa.lse[b] = c.lse;
    // next step's input: xt = dropout(relu(embed(it))) (AttModel.py:74-76,399), un-masked `it`
    if (a.emb_x) {
        const int E4 = a.emb_dim >> 2;
        for (int j = lane; j < E4; j += 64) {
            const f32x4 ev = reinterpret_cast<const f32x4*>(a.emb_w + (size_t)c.tok * a.emb_dim)[j];
            uint32_t kp = 0x01010101u;
            if (a.emb_keep) kp = *reinterpret_cast<const uint32_t*>(a.emb_keep + (size_t)b * a.emb_dim + 4 * j);
            reinterpret_cast<f32x4*>(a.emb_x + (size_t)b * a.emb_dim)[j] = embed_transform(a, ev, kp);
        }
    }
    if (lane == 0) row_bookkeeping(a, c, b);      // EOS bookkeeping (AttModel.py:401-434)
}
__global__ __launch_bounds__(256) void sample_finish_kernel(FinishArgs a0, FinishArgs a1, int np, FinishLen fl) {
    sample_finish_rows(a0, a1, np);
    if (fl.counter) finish_len_tail(fl);      // (grid-uniform)
}

// ---- teacher forcing without scheduled sampling: the fed tokens are the targets, known before the loop ------------------
// (AttModel.forward, models/AttModel.py:103-148).  The token columns, the unfinished chain and the caption matrix are
// written up front; the logit product, the log-sum-exp and the gathered log-probabilities run ONCE over all T*B rows
// after the recurrence (cic_speaker_decode_fwd).
__global__ __launch_bounds__(256) void teacher_tokens_kernel(const int64_t* __restrict__ pick, int32_t* __restrict__ it_all,
                                                             int32_t* __restrict__ unfinished, int32_t* __restrict__ any_unf,
                                                             int32_t* __restrict__ seq, int T, int B,
                                                             unsigned* __restrict__ zsync, int nzsync) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    // hand-off counters of spk_teacher_seq_kernel (this launch precedes it on the stream), or null
    for (int i = b; i < nzsync; i += gridDim.x * blockDim.x) zsync[i] = 0u;
    if (b >= B) return;
    int unf = 1;
    for (int t = 1; t <= T; ++t) {
        const int it = (int)pick[(size_t)t * B + b];
        unf = (t > 1 ? unf : 1) & (it > 0 ? 1 : 0);             // sample_finish_kernel's bookkeeping, step by step
        it_all[(size_t)t * B + b] = it;
        seq[(size_t)b * T + (t - 1)] = unf ? it : 0;
        if (unf) atomicOr(any_unf + t, 1);
    }
    unfinished[b] = unf;
}

// one wave per row r = t*B + b of the raw logits: lse from the row's partials, slp[b, t] = logit[target] - lse
__global__ __launch_bounds__(256) void teacher_finish_all_kernel(const float* __restrict__ part, int np, int part_rows,
                                                                 const float* __restrict__ logits, int ld,
                                                                 const int64_t* __restrict__ pick, float* __restrict__ lse_all,
                                                                 float* __restrict__ slp, int T, int B) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= T * B) return;
    const int t = r / B, b = r % B;
    const size_t plane = (size_t)part_rows * np;
    RowPart rp;
    rp.init();
    for (int p = lane; p < np; p += 64) {
        const float* pp = part + (size_t)r * np + p;
        RowPart q;
        q.init();
        q.m1 = pp[0]; q.s1 = pp[plane];
        rowpart_merge(rp, CIC_SAMPLE_NONE, 1.0f, q);
    }
    rp = wave_merge_rowpart(rp, CIC_SAMPLE_NONE, 1.0f);
    if (lane == 0) {
        const float lse = rp.m1 + logf(rp.s1);
        lse_all[r] = lse;
        const int it = (int)pick[(size_t)(t + 1) * B + b];
        slp[(size_t)b * T + t] = logits[(size_t)r * ld + it] - lse;
    }
}

// L = number of appended columns: the reference breaks at the first t >= 1 whose unfinished
// sum is 0 (AttModel.py:407-408); otherwise seq_length.
// (block q = decode q of a pair)
__global__ void finalize_len_kernel(Dual<const int> any_unf_d, int T, Dual<int> L_d) {
    if (threadIdx.x == 0) {
        const int* __restrict__ any_unf = any_unf_d.sel(blockIdx.x == 1);
        int l = T;
        for (int t = 1; t <= T; ++t)
            if (any_unf[t] == 0) {
                l = t - 1;
                break;
            }
        *L_d.sel(blockIdx.x == 1) = l;
    }
}

}  // namespace

// ---- launchers ---------------------------------------------------------------------------
#ifdef CIC_DEVTOOLS
extern "C" int cic_debug_set_attn_stamps(unsigned long long* buf) {
    CIC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamps), &buf, sizeof(buf)));
    return 0;
}
#endif

extern "C" int cic_attn_fwd(const float* att_h, const float* p_att, const float* att, const float* w_alpha,
                            const float* b_alpha, const float* masks, float* att_res, float* alpha,
                            float* dot, int B, int K, int A, int H, cic_stream_t s) {
    return cic_attn_fwd2(dual1(att_h), dual1(p_att), dual1(att), w_alpha, b_alpha, masks, dual1(att_res), dual1(alpha),
                         dual1(dot), B, 1, K, A, H, cic_s(s), 1);
}

bool cic_attn_pair_ok(int K, int A, int H) { return A == H && (H & 63) == 0 && H <= 512 && K <= 64; }

int cic_attn_fwd2(Dual<const float> att_h_d, Dual<const float> p_att_d, Dual<const float> att_d, const float* w_alpha,
                  const float* b_alpha, const float* masks, Dual<float> att_res_d, Dual<float> alpha_d, Dual<float> dot_d,
                  int B, int nb, int K, int A, int H, hipStream_t st, int att_div, Dual<const uint16_t> p_att_bf,
                  Dual<const uint16_t> att_bf, Dual<const int32_t> live) {
    CIC_REQUIRE(att_div >= 1);
    const bool bf = p_att_bf.a != nullptr;
    CIC_REQUIRE(!bf || (att_bf.a && cic_attn_pair_ok(K, A, H) && (nb == 1 || (p_att_bf.b && att_bf.b))));
    const float *att_h = att_h_d.a, *p_att = p_att_d.a, *att = att_d.a;
    float *att_res = att_res_d.a, *alpha = alpha_d.a, *dot = dot_d.a;
    CIC_REQUIRE(att_h && p_att && att && w_alpha && b_alpha && att_res && alpha);
    CIC_REQUIRE(B > 0 && K > 0 && K <= 64 && (A & 3) == 0 && (H & 3) == 0);
    CIC_REQUIRE(nb == 1 || (nb == 2 && cic_attn_pair_ok(K, A, H) && att_h_d.b && p_att_d.b && att_d.b && att_res_d.b && alpha_d.b));
    const int mx = A > H ? A : H;
    CIC_REQUIRE(mx <= 1024);
    dim3 grid(nb * B);
    if (cic_attn_pair_ok(K, A, H)) {   // column-owner kernel (one barrier, no vector reduce)
        // one 32-column group per wave (16 waves at H = 512).  Two groups per wave (8 waves) measured slower:
        // 6.9 us vs 6.4 us in-kernel span at B = 128.
        const int ncg = 1;
        dim3 blk((H / 32 / ncg) * 64);
        if (bf) {      // bf16 storage of the region features (compute_dtype bf16): half the bytes per image
#define GOB(J) hipLaunchKernelGGL((attn_fwd_cols_kernel<J, 1, uint16_t>), grid, blk, 0, st, att_h_d, p_att_bf, att_bf, w_alpha, \
                                  b_alpha, masks, att_res_d, alpha_d, dot_d, B, K, H, att_div, live)
            if (K <= 8) GOB(1); else if (K <= 16) GOB(2); else if (K <= 24) GOB(3); else if (K <= 32) GOB(4);
            else if (K <= 40) GOB(5); else if (K <= 48) GOB(6); else GOB(8);
#undef GOB
            CIC_LAUNCH_CHECK();
            return 0;
        }
#define GOC(J)                                                                                                       \
    do {                                                                                                             \
        if (ncg == 2) hipLaunchKernelGGL((attn_fwd_cols_kernel<J, 2>), grid, blk, 0, st, att_h_d, p_att_d, att_d, w_alpha, \
                                         b_alpha, masks, att_res_d, alpha_d, dot_d, B, K, H, att_div, live);               \
        else hipLaunchKernelGGL((attn_fwd_cols_kernel<J, 1>), grid, blk, 0, st, att_h_d, p_att_d, att_d, w_alpha,     \
                                b_alpha, masks, att_res_d, alpha_d, dot_d, B, K, H, att_div, live);                        \
    } while (0)
        if (K <= 8) GOC(1); else if (K <= 16) GOC(2); else if (K <= 24) GOC(3); else if (K <= 32) GOC(4);
        else if (K <= 40) GOC(5); else if (K <= 48) GOC(6); else GOC(8);
#undef GOC
        CIC_LAUNCH_CHECK();
        return 0;
    }
    // 16 waves per image: each wave owns <= 4 regions, whose p_att and att rows are all in flight at once
#define GO(NI, KPW, HOLD)                                                                                        \
    hipLaunchKernelGGL((attn_fwd_kernel<NI, KPW, 16, HOLD>), grid, dim3(1024), 0, st, att_h, p_att, att, w_alpha, \
                       b_alpha, masks, att_res, alpha, dot, K, A, H, att_div)
    if (mx <= 256) {
        if (K <= 48) GO(1, 3, true); else GO(1, 4, true);
    } else if (mx <= 512) {
        if (K <= 48) GO(2, 3, true); else GO(2, 4, true);
    } else {
        GO(4, 4, false);
    }
#undef GO
    CIC_LAUNCH_CHECK();
    return 0;
}

// Streams `n4` float4 through the chip (sum into one word): stands in for the ~45 MB of weights, logits and
// noise that the other kernels of a decode step move between two attention launches, so that the timed
// attention launches below see the cache state they see inside the real step (att / p_att evicted from L2).
__global__ __launch_bounds__(256) void pollute_kernel(const f32x4* __restrict__ src, int64_t n4, float* __restrict__ sink) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = src[i];
        acc += v[0] + v[1] + v[2] + v[3];
    }
    if (acc == 123456.789f) *sink = acc;   // keeps the loads alive; practically never taken
}

// Average duration of one attention launch in the cache state of the real decode step, measured from C++ with
// HIP events on `s`:  ( time of `iters` x [polluter, attention]  -  time of `iters` x [polluter] ) / iters.
// pollute == NULL times plain back-to-back launches (att / p_att then stay L2-resident: an upper bound).
extern "C" int cic_attn_fwd_timed(const float* att_h, const float* p_att, const float* att, const float* w_alpha,
                                  const float* b_alpha, float* att_res, float* alpha, int B, int K, int A, int H,
                                  int iters, const float* pollute, int64_t pollute_floats, double* avg_us,
                                  cic_stream_t s) {
    CIC_REQUIRE(iters > 0 && avg_us);
    hipStream_t st = cic_s(s);
    hipEvent_t e0, e1, e2;
    CIC_HIP(hipEventCreate(&e0));
    CIC_HIP(hipEventCreate(&e1));
    CIC_HIP(hipEventCreate(&e2));
    int rc = 0;
    const int64_t n4 = pollute ? pollute_floats / 4 : 0;
    auto poll = [&]() {
        if (n4 > 0) hipLaunchKernelGGL(pollute_kernel, dim3(2048), dim3(256), 0, st, reinterpret_cast<const f32x4*>(pollute), n4, att_res);
    };
    for (int i = 0; i < 5 && !rc; ++i) {
        poll();
        rc = cic_attn_fwd(att_h, p_att, att, w_alpha, b_alpha, nullptr, att_res, alpha, nullptr, B, K, A, H, s);
    }
    CIC_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < iters && !rc; ++i) {
        poll();
        rc = cic_attn_fwd(att_h, p_att, att, w_alpha, b_alpha, nullptr, att_res, alpha, nullptr, B, K, A, H, s);
    }
    CIC_HIP(hipEventRecord(e1, st));
    for (int i = 0; i < iters && n4 > 0; ++i) poll();
    CIC_HIP(hipEventRecord(e2, st));
    CIC_HIP(hipEventSynchronize(e2));
    float ms_pair = 0.f, ms_poll = 0.f;
    CIC_HIP(hipEventElapsedTime(&ms_pair, e0, e1));
    CIC_HIP(hipEventElapsedTime(&ms_poll, e1, e2));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipEventDestroy(e2);
    *avg_us = ((double)ms_pair - (n4 > 0 ? (double)ms_poll : 0.0)) * 1e3 / iters;
    return rc;
}

extern "C" int cic_cell_fwd(const float* pre, const float* c_prev, const uint8_t* keep, float p_drop,
                            float* h_new, float* c_new, float* out, int B, int H, cic_stream_t s) {
    return cic_cell_fwd2(dual1(pre), dual1(c_prev), dual1(keep), p_drop, dual1(h_new), dual1(c_new), dual1(out), B, 1, H,
                         cic_s(s));
}
int cic_cell_fwd2(Dual<const float> pre, Dual<const float> c_prev, Dual<const uint8_t> keep, float p_drop, Dual<float> h_new,
                  Dual<float> c_new, Dual<float> out, int B, int nb, int H, hipStream_t st, int state_dropped) {
    CIC_REQUIRE(pre.a && c_prev.a && h_new.a && c_new.a && out.a && B > 0 && H > 0 && (H & 3) == 0);
    CIC_REQUIRE(nb == 1 || (nb == 2 && pre.b && c_prev.b && h_new.b && c_new.b && out.b));
    const int n = nb * B * (H / 4);
    hipLaunchKernelGGL(cell_fwd_kernel, dim3(cic_cdiv(n, 256)), dim3(256), 0, st, pre, c_prev, keep,
                       1.0f / (1.0f - p_drop), h_new, c_new, out, B, nb, H, state_dropped);
    CIC_LAUNCH_CHECK();
    return 0;
}

CIC_SWITCH(g_a2c_cell_fused, 1);
#ifdef CIC_DEVTOOLS
extern "C" int cic_debug_a2c_cell_fused(int on) {
    g_a2c_cell_fused = on;
    return 0;
}
#endif
bool cic_a2c_cell_fused_ok(int H) { return g_a2c_cell_fused && H == 512; }

int cic_a2c_cell_fused(Dual<const float> att_res, const float* Wa, const float* ba, Dual<float> pre, Dual<const float> c_prev,
                       Dual<const uint8_t> keep, float p_drop, Dual<float> h_new, Dual<float> c_new, Dual<float> out, int B,
                       int nb, int H, hipStream_t st, Dual<const int32_t> live) {
    CIC_REQUIRE(cic_a2c_cell_fused_ok(H) && att_res.a && Wa && ba && pre.a && c_prev.a && h_new.a && c_new.a && out.a && B > 0);
    CIC_REQUIRE(nb == 1 || (nb == 2 && att_res.b && pre.b && c_prev.b && h_new.b && c_new.b && out.b));
    const int grid = nb * cic_cdiv(B, 32) * (H / 16);
    hipLaunchKernelGGL((a2c_cell_fused_kernel<4, 8>), dim3(grid), dim3(512), 0, st, att_res, Wa, ba, pre, c_prev, keep,
                       1.0f / (1.0f - p_drop), h_new, c_new, out, B, nb, H, live);
    CIC_LAUNCH_CHECK();
    return 0;
}

// ---- attention + att2ctx + cell as one launch (attn_a2c_cell_kernel) ------------------------------------------------------------
CIC_SWITCH(g_attn_cell_fused, 1);
#ifdef CIC_DEVTOOLS
extern "C" int cic_debug_attn_cell_fused(int on) {
    g_attn_cell_fused = on;
    return 0;
}
#endif
template <int J, typename ST>
static int attn_cell_resident() {
    return cic_resident_cus(reinterpret_cast<const void*>(&attn_a2c_cell_kernel<J, ST>), 1024, 0);
}
static int attn_cell_grid(int B, int nb) {
    const int rows = nb * B, tiles = nb * cic_cdiv(B, 32) * 32;
    return rows > tiles ? rows : tiles;
}
// the fused launch needs every workgroup resident at once (one 16-wave workgroup per CU) and this process alone on the GPU
bool cic_attn_cell_fused_ok(int B, int nb, int K, int A, int H, bool bf, int device_shared) {
    if (!g_attn_cell_fused || device_shared || H != 512 || A != 512 || K < 1 || K > 40 || !cic_a2c_cell_fused_ok(H)) return false;
    const int cus = bf ? attn_cell_resident<5, uint16_t>() : attn_cell_resident<5, float>();
    return attn_cell_grid(B, nb) <= cus;
}
int cic_attn_a2c_cell(const AttnCellLaunch& L, hipStream_t st) {
    const bool bf = L.p_att_bf.a != nullptr;
    CIC_REQUIRE(L.att_h.a && L.w_alpha && L.b_alpha && L.att_res.a && L.alpha.a && L.Wa && L.ba && L.pre.a && L.c_prev.a && L.h_new.a &&
                L.c_new.a && L.out.a && L.cnt && L.err && L.B > 0 && (L.nb == 1 || L.nb == 2));
    CIC_REQUIRE(bf ? (L.att_bf.a != nullptr) : (L.p_att.a && L.att.a));
    CIC_REQUIRE(L.nb == 1 || (L.att_h.b && L.att_res.b && L.alpha.b && L.pre.b && L.c_prev.b && L.h_new.b && L.c_new.b && L.out.b));
    CIC_REQUIRE(cic_attn_cell_fused_ok(L.B, L.nb, L.K, 512, 512, bf, 0));
    const int grid = attn_cell_grid(L.B, L.nb);
    const HandoffGuard hg = handoff_guard(L.err, L.status, CIC_STATUS_DECODE_STEP);
#define FILL(a)                                                                                                                  \
    a.att_h = L.att_h; a.w_alpha = L.w_alpha; a.b_alpha = L.b_alpha; a.masks = L.masks; a.att_res = L.att_res; a.alpha = L.alpha;     \
    a.dot = L.dot; a.Wa = L.Wa; a.ba = L.ba; a.pre = L.pre; a.c_prev = L.c_prev; a.keep = L.keep;                                      \
    a.scale = 1.0f / (1.0f - L.p_drop); a.h_new = L.h_new; a.c_new = L.c_new; a.out = L.out; a.live = L.live; a.cnt = L.cnt;           \
    a.hg = hg; a.B0 = L.B; a.nb = L.nb; a.K = L.K
#define GOJ(ST, J) hipLaunchKernelGGL((attn_a2c_cell_kernel<J, ST>), dim3(grid), dim3(1024), 0, st, a)
#define GOK(ST)                                                                                                                  \
    do {                                                                                                                         \
        if (L.K <= 8) GOJ(ST, 1); else if (L.K <= 16) GOJ(ST, 2); else if (L.K <= 24) GOJ(ST, 3); else if (L.K <= 32) GOJ(ST, 4);   \
        else GOJ(ST, 5);                                                                                                         \
    } while (0)
    if (bf) {
        AttnCellArgs<uint16_t> a;
        FILL(a); a.p_att = L.p_att_bf; a.att = L.att_bf;
        GOK(uint16_t);
    } else {
        AttnCellArgs<float> a;
        FILL(a); a.p_att = L.p_att; a.att = L.att;
        GOK(float);
    }
#undef GOK
#undef GOJ
#undef FILL
    CIC_LAUNCH_CHECK();
    return 0;
}

extern "C" int cic_embed_fwd(const float* E, const int32_t* it, const uint8_t* keep, float p_drop, float* x,
                             int B, int Ed, cic_stream_t s) {
    return cic_embed_fwd2(E, dual1(it), dual1(keep), p_drop, dual1(x), B, 1, Ed, cic_s(s));
}
int cic_embed_fwd2(const float* E, Dual<const int32_t> it, Dual<const uint8_t> keep, float p_drop, Dual<float> x, int B,
                   int nb, int Ed, hipStream_t st, int plain) {
    CIC_REQUIRE(E && it.a && x.a && B > 0 && Ed > 0 && (Ed & 3) == 0);
    CIC_REQUIRE(nb == 1 || (nb == 2 && it.b && x.b));
    const int n = nb * B * (Ed / 4);
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(cic_cdiv(n, 256)), dim3(256), 0, st, E, it, keep, 1.0f / (1.0f - p_drop), x,
                       B, nb, Ed, plain);
    CIC_LAUNCH_CHECK();
    return 0;
}

extern "C" int cic_apply_keep(const float* x, const uint8_t* keep, float p_drop, float* y, int64_t n,
                              cic_stream_t s) {
    CIC_REQUIRE(x && y && n > 0 && (n & 3) == 0);
    hipLaunchKernelGGL(apply_keep_kernel, dim3(cic_cdiv(n / 4, 256)), dim3(256), 0, cic_s(s), x, keep,
                       1.0f / (1.0f - p_drop), y, n / 4);
    CIC_LAUNCH_CHECK();
    return 0;
}

int cic_apply_keep2(const float* x, Dual<const uint8_t> keep, float p_drop, Dual<float> y, int64_t n, hipStream_t st) {
    CIC_REQUIRE(x && y.a && y.b && n > 0 && (n & 3) == 0);
    hipLaunchKernelGGL(apply_keep2_kernel, dim3(cic_cdiv(n / 4, 256), 2), dim3(256), 0, st, x, keep, 1.0f / (1.0f - p_drop), y, n / 4);
    CIC_LAUNCH_CHECK();
    return 0;
}

int cic_att_keep_rows(const float* x, const uint8_t* keep, float p_drop, const float* masks, float* y, int B, int K, int H,
                      hipStream_t st) {
    CIC_REQUIRE(x && y && masks && B > 0 && K > 0 && H > 0 && (H & 3) == 0);
    hipLaunchKernelGGL(att_keep_rows_kernel, dim3(B * K), dim3(256), 0, st, x, keep, 1.0f / (1.0f - p_drop), masks, y, B, K,
                       H / 4);
    CIC_LAUNCH_CHECK();
    return 0;
}

int cic_round_pack_bf16(float* x, uint16_t* packed, int64_t n, hipStream_t st) {
    CIC_REQUIRE(x && packed && n > 0 && (n & 3) == 0);
    hipLaunchKernelGGL(round_pack_bf16_kernel, dim3(cic_cdiv(n / 4, 256)), dim3(256), 0, st, x, packed, n / 4);
    CIC_LAUNCH_CHECK();
    return 0;
}

int cic_relu_keep_fwd(const float* xpre, const uint8_t* keep, float p_drop, float* x, int64_t n, hipStream_t st) {
    CIC_REQUIRE(xpre && x && n > 0 && (n & 3) == 0);
    hipLaunchKernelGGL(relu_keep_fwd_kernel, dim3(cic_cdiv(n / 4, 256)), dim3(256), 0, st, xpre, keep, 1.0f / (1.0f - p_drop),
                       x, n / 4);
    CIC_LAUNCH_CHECK();
    return 0;
}
int cic_soft_mask(const float* soft_raw, const int32_t* seq, const int32_t* L, float* soft_out, int T, int B, int V1,
                  hipStream_t st) {
    hipLaunchKernelGGL(soft_mask_kernel, dim3(T * B), dim3(256), 0, st, soft_raw, seq, L, soft_out, T, B, V1);
    CIC_LAUNCH_CHECK();
    return 0;
}

static int check_sampler_args(const cic_sampler_args* a, bool noise_in_partials = false);

extern "C" int cic_logsoftmax_sample(const cic_sampler_args* a, cic_stream_t s) {
    return cic_logsoftmax_sample2(a, nullptr, cic_s(s));
}

// one launch for the samplers of one decode (b == NULL) or of a pair of decodes (rows of b follow those of a)
int cic_logsoftmax_sample2(const cic_sampler_args* a, const cic_sampler_args* b, hipStream_t st) {
    if (int rc = check_sampler_args(a)) return rc;
    if (b) {
        if (int rc = check_sampler_args(b)) return rc;
        CIC_REQUIRE(b->V1 == a->V1);
    }
    dim3 grid(a->B + (b ? b->B : 0)), blk(1024);
    const cic_sampler_args& b_ = b ? *b : *a;
    if (a->V1 <= 4096) hipLaunchKernelGGL((logsoftmax_sample_kernel<1>), grid, blk, 0, st, *a, b_);
    else if (a->V1 <= 12288) hipLaunchKernelGGL((logsoftmax_sample_kernel<3>), grid, blk, 0, st, *a, b_);
    else if (a->V1 <= 32768) hipLaunchKernelGGL((logsoftmax_sample_kernel<8>), grid, blk, 0, st, *a, b_);
    else {
        cic_set_error("cic_logsoftmax_sample: vocabulary %d too large (max 32768)", a->V1);
        return 1;
    }
    CIC_LAUNCH_CHECK();
    return 0;
}

extern "C" int cic_logit_partials(const float* logits, int M, int N, int ld, const cic_logit_epi_rows* e, int nparts,
                                  cic_stream_t s) {
    CIC_REQUIRE(logits && e && e->part && M > 0 && N > 0 && ld >= N && nparts > 0 && (int64_t)M * nparts <= CIC_PART_MAX_ENTRIES);
    CIC_REQUIRE(e->part_rows >= M && (!e->noise || e->U || e->philox));
    const int chunk = cic_cdiv(N, nparts);
    hipLaunchKernelGGL(logit_partials_kernel, dim3(cic_cdiv((int64_t)M * nparts, 4)), dim3(256), 0, cic_s(s), logits, M, N, ld, *e,
                       nparts, chunk);
    CIC_LAUNCH_CHECK();
    return 0;
}

int cic_teacher_tokens(const int64_t* pick, int32_t* it_all, int32_t* unfinished, int32_t* any_unf, int32_t* seq, int T, int B,
                       hipStream_t st, unsigned* zsync, int nzsync) {
    CIC_REQUIRE(pick && it_all && unfinished && any_unf && seq && T > 0 && B > 0);
    hipLaunchKernelGGL(teacher_tokens_kernel, dim3(cic_cdiv(B, 256)), dim3(256), 0, st, pick, it_all, unfinished, any_unf, seq, T, B,
                       zsync, zsync ? nzsync : 0);
    CIC_LAUNCH_CHECK();
    return 0;
}
int cic_teacher_finish_all(const float* part, int np, int part_rows, const float* logits, int ld, const int64_t* pick,
                           float* lse_all, float* slp, int T, int B, hipStream_t st) {
    CIC_REQUIRE(part && np > 0 && part_rows >= T * B && logits && pick && lse_all && slp && T > 0 && B > 0);
    hipLaunchKernelGGL(teacher_finish_all_kernel, dim3(cic_cdiv(T * B, 4)), dim3(256), 0, st, part, np, part_rows, logits, ld,
                       pick, lse_all, slp, T, B);
    CIC_LAUNCH_CHECK();
    return 0;
}

// the sampler of one decode (b == NULL) or of a pair on the row partials of this step's logits
int cic_sample_finish2(const cic_sampler_args* a, const float* part_a, int part_rows_a, float* lse_a,
                       const cic_sampler_args* b, const float* part_b, int part_rows_b, float* lse_b, int np, hipStream_t st,
                       int32_t* L_a, int32_t* L_b, int T) {
    // L_a (and L_b for a pair): this is the decode's LAST sampler launch; it also writes the decodes' lengths
    if (int rc = check_sampler_args(a, true)) return rc;
    if (b) { if (int rc = check_sampler_args(b, true)) return rc; }
    CIC_REQUIRE(part_a && np > 0 && (!b || part_b));
    auto ok = [](const cic_sampler_args* x) {
        return x->mode != CIC_SAMPLE_GUMBEL_PS && x->mode != CIC_SAMPLE_MULTINOMIAL_PS;
    };
    CIC_REQUIRE(ok(a) && (!b || ok(b)));
    FinishArgs fa{*a, part_a, part_rows_a, lse_a};
    FinishArgs fb = b ? FinishArgs{*b, part_b, part_rows_b, lse_b} : fa;
    const int rows = a->B + (b ? b->B : 0);
    FinishLen fl = {};
    if (L_a) {
        CIC_REQUIRE(T > 0 && a->step == T && (!b || L_b));
        fl.counter = a->any_unfinished;                      // entry 0: no step's flag
        fl.any_unf = Dual<const int32_t>{a->any_unfinished, b ? b->any_unfinished : nullptr};
        fl.L = Dual<int32_t>{L_a, b ? L_b : nullptr};
        fl.T = T;
    }
    hipLaunchKernelGGL(sample_finish_kernel, dim3(cic_cdiv(rows, 4)), dim3(256), 0, st, fa, fb, np, fl);
    CIC_LAUNCH_CHECK();
    return 0;
}

static int check_sampler_args(const cic_sampler_args* a, bool noise_in_partials) {
    CIC_REQUIRE(a && a->logits && a->B > 0 && a->V1 > 0 && a->ld >= a->V1);
    CIC_REQUIRE(!a->emb_x || (a->emb_w && a->mode != CIC_SAMPLE_NONE && a->emb_dim > 0 && (a->emb_dim & 3) == 0 && a->emb_dim <= 4096));
    if (a->mode != CIC_SAMPLE_NONE) {
        CIC_REQUIRE(a->unfinished && a->it_next && a->seq && a->slp && a->any_unfinished && a->step >= 1);
        CIC_REQUIRE(a->temp > 0.f);
        const bool needs_u = a->mode == CIC_SAMPLE_GUMBEL_ST || a->mode == CIC_SAMPLE_GUMBEL_PS ||
                             ((a->mode == CIC_SAMPLE_MULTINOMIAL || a->mode == CIC_SAMPLE_MULTINOMIAL_ST ||
                               a->mode == CIC_SAMPLE_MULTINOMIAL_PS) && !a->pick) ||
                             (a->mode == CIC_SAMPLE_TEACHER && a->ss_u && a->ss_prob > 0.f && !a->ss_pick);
        CIC_REQUIRE(a->mode != CIC_SAMPLE_TEACHER || a->pick);
        const bool ps = a->mode == CIC_SAMPLE_GUMBEL_PS || a->mode == CIC_SAMPLE_MULTINOMIAL_PS;
        CIC_REQUIRE(!ps || (a->soft && a->ld_soft >= a->V1));
        CIC_REQUIRE(!needs_u || noise_in_partials || (a->U && a->ldu >= a->V1));
    }
    return 0;
}

extern "C" int cic_finalize_len(const int* any_unfinished, int T, int* L, cic_stream_t s) {
    return cic_finalize_len2(dual1(any_unfinished), T, dual1(L), 1, cic_s(s));
}
int cic_finalize_len2(Dual<const int> any_unfinished, int T, Dual<int> L, int nb, hipStream_t st) {
    CIC_REQUIRE(any_unfinished.a && L.a && T > 0 && (nb == 1 || (nb == 2 && any_unfinished.b && L.b)));
    hipLaunchKernelGGL(finalize_len_kernel, dim3(nb), dim3(64), 0, st, any_unfinished, T, L);
    CIC_LAUNCH_CHECK();
    return 0;
}


// the teacher-forced recurrence as one launch (see spk_teacher_seq_kernel); pre_all must hold x_t i2h^T + bias of every step
bool cic_teacher_seq_ok(int B, int K, int H, int A, int E) {
    if (!(H == 512 && A == 512 && K >= 1 && K <= 36)) return false;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    return cus >= H / 16;          // at least one strip of 16 rows resident: larger batches are walked in row blocks
}
// CUs of the current device if each of them admits one workgroup of `kernel` (threads, dynamic LDS bytes) by the occupancy
// query, else 0: the grid bound of the one-launch recurrences (residency was taken from the CU count alone).  The
// query knows this process's kernel only; a second process on the GPU has to be declared (device_shared).
int cic_resident_cus(const void* kernel, int threads, size_t lds_bytes) {
    // the answer is a constant of (kernel, device): asked once, then served from a small table (a step asks four times)
    struct Entry { const void* k; int dev; int cus; };
    static Entry table[32];
    static std::atomic<int> filled{0};
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    const int n = filled.load(std::memory_order_acquire);
    for (int i = 0; i < n; ++i)
        if (table[i].k == kernel && table[i].dev == dev) return table[i].cus;
    int cus = 0, blocks = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kernel, threads, lds_bytes) != hipSuccess) { (void)hipGetLastError(); return 0; }
    const int ans = blocks >= 1 ? cus : 0;
    std::lock_guard<std::mutex> lock(mu);
    const int m = filled.load(std::memory_order_relaxed);
    if (m < 32) { table[m] = Entry{kernel, dev, ans}; filled.store(m + 1, std::memory_order_release); }
    return ans;
}
int cic_teacher_seq(const TeacherSeqLaunch& L, hipStream_t st) {
    static DeviceOnce attr_set;
    if (attr_set.first()) {
        CIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&spk_teacher_seq_kernel<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)TEACHER_LDS_BYTES));
        CIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&spk_teacher_seq_kernel<8, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)TEACHER_LDS_BYTES));
    }
    const int strips = cic_cdiv(L.B, 16);
    // (the hand-off counters were cleared by cic_teacher_tokens, which precedes this launch on the stream)
    TeacherSeqArgs a = {};
    a.h2h_w = L.h2h_w; a.h2att_w = L.h2att_w; a.h2att_b = L.h2att_b; a.a2c_w = L.a2c_w; a.a2c_b = L.a2c_b;
    a.alpha_w = L.alpha_w; a.alpha_b = L.alpha_b; a.p_att = L.p_att; a.att = L.att; a.masks = L.masks; a.out_keep = L.out_keep;
    a.pre_all = L.pre_all; a.h_all = L.h_all; a.c_all = L.c_all; a.att_h_all = L.att_h_all; a.att_res_all = L.att_res_all;
    a.alpha_all = L.alpha_all; a.dot_all = L.dot_all; a.out_all = L.out_all;
    a.cnt = L.sync;
    a.hg = handoff_guard(L.sync + (size_t)strips * L.T * 3, L.status, CIC_STATUS_TEACHER);
    a.scale = L.scale; a.B = L.B; a.K = L.K; a.T = L.T;
    // every workgroup of a launch must be resident at once: one per CU, and only where the occupancy query admits one
    const int cus = L.bf16 ? cic_resident_cus(reinterpret_cast<const void*>(&spk_teacher_seq_kernel<8, true>), 512, TEACHER_LDS_BYTES)
                           : cic_resident_cus(reinterpret_cast<const void*>(&spk_teacher_seq_kernel<8, false>), 512, TEACHER_LDS_BYTES);
    const int seq_rows = (cus / 32) * 16;                       // rows one launch can walk with every workgroup resident
    if (seq_rows < 16) { cic_set_error("teacher_seq: fewer than 32 CUs admit the kernel"); return 1; }
    for (int row0 = 0; row0 < L.B; row0 += seq_rows) {          // (B = 128: one launch; B = 256: two row blocks)
        a.row0 = row0;
        a.row_end = row0 + seq_rows < L.B ? row0 + seq_rows : L.B;
        if (L.bf16)
            hipLaunchKernelGGL((spk_teacher_seq_kernel<8, true>), dim3(cic_cdiv(a.row_end - row0, 16) * 32), dim3(512), TEACHER_LDS_BYTES, st, a);
        else
            hipLaunchKernelGGL((spk_teacher_seq_kernel<8, false>), dim3(cic_cdiv(a.row_end - row0, 16) * 32), dim3(512), TEACHER_LDS_BYTES, st, a);
        CIC_LAUNCH_CHECK();
    }
    return 0;
}
