// Listener (VSE++ "fc" model): image FC encoder, GRU text encoder over (straight-through)
// token sequences, max-violation hinge contrastive loss — forward and backward.
// Reference: models/VSEFCModel.py:12-241.
//
// Design notes (MI355X-first):
//  * the reference multiplies a dense one-hot [B,L,V+2] by the embedding table
//    (VSEFCModel.py:102-104); (y_hard - y).detach() + y is EXACTLY zero off the sampled
//    token, so the forward is a row gather scaled by the straight-through value stv;
//    only the backward needs the dense product (d one_hot = dX E^T), done by cic_gemm_f32.
//  * packed sequences are replaced by a length mask: rows stop updating h at t >= len, which
//    yields the same h[len-1] that pack_padded_sequence + gather(len-1) picks (:108-129).
//  * the input projection of all time steps is ONE GEMM; only h W_hh^T is recurrent.
#include "cic_common.h"
#include "engine_util.h"

namespace {

// 2 = the whole GRU pass (and its BPTT loop) as one launch each with W_hh stationary in the workgroups (gru_seq_kernel,
// gru_seq_bwd_kernel; need one resident workgroup per 16-row strip x 32-unit tile, else 1); 1 = one fused launch per step;
// 0 = a GEMM + a cell launch per step.
// Development build: cic_debug_gru_fused(n) selects (A/B measurement, bit-exactness tests of 2 against 1).
CIC_SWITCH(g_gru_fused, 2);
#ifdef CIC_DEVTOOLS
__device__ unsigned long long* g_gru_stamps = nullptr;   // development build: [workgroup][step][8] s_memrealtime stamps of gru_seq_bwd_kernel
#endif

// ---- token preparation -----------------------------------------------------------------
// generated captions: tokens = [<bos>, seq[:, 0:L]], lens from masks [1,1,(seq>0)[:, :L-1]]
// (models/AlternatingJointModel.py:353-370)
// (r4) ... and the caption's embedded input rows x_emb[t,b,:] = val[b,t] * E[idx[b,t],:] in the same launch (embed_st_fwd_kernel's
// arithmetic; the tokens are in the wave's lanes already: one launch and one dependent round trip less per step)
__global__ __launch_bounds__(256) void prep_generated_kernel(const int32_t* __restrict__ seq, const float* __restrict__ stv,
                                      const int32_t* __restrict__ Lp, int B, int T, int bos, int dense,
                                      int32_t* __restrict__ idx, float* __restrict__ val,
                                      int32_t* __restrict__ len, unsigned* __restrict__ sync, int nsync,
                                      const float* __restrict__ E, float* __restrict__ x_emb, int Ed) {
    // dense != 0 (soft caption rows): positions 1..T are embedded by a dense product added afterwards, so
    // their gather contributes nothing (val = 0)
    // one workgroup (4 waves) per caption, lane j = token j (T <= 63; longer captions: lanes stride): every wave reads the row (one
    // round trip), wave 0 keeps the books, the T + 1 embedded rows are dealt over the four waves (was: one wave per caption walking
    // its 17 rows one dependent gather after the other, 20 us)
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = gtid; i < nsync; i += gridDim.x * blockDim.x) sync[i] = 0u;    // hand-off counters of gru_seq_kernel
    const int b = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    if (b >= B) return;
    const int L = *Lp;
    int cnt = 0;
    int tok_l = 0;                 // this lane's position j = lane (captions of up to 63 tokens: the fused embedding's case)
    float val_l = 1.0f;
    for (int j = lane; j < T; j += 64) {
        const int tok = seq[(size_t)b * T + j];
        const int ti = (j < L && !dense) ? tok : 0;
        const float tv = dense ? 0.0f : ((j < L && stv) ? stv[(size_t)b * T + j] : 1.0f);
        if (wv == 0) {
            idx[(size_t)b * (T + 1) + 1 + j] = ti;
            val[(size_t)b * (T + 1) + 1 + j] = tv;
        }
        if (j == lane) { tok_l = ti; val_l = tv; }
        if (j < L - 1 && tok > 0) ++cnt;
    }
    if (x_emb && T < 64) {
        // position p of the caption: p = 0 is <bos> (value 1), p >= 1 the token lane p-1 holds; time-major rows [p, b, :]
        const int E4 = Ed >> 2;
        for (int p = wv; p <= T; p += nwv) {
            const int tok = p == 0 ? bos : __shfl(tok_l, p - 1, 64);
            const float v = p == 0 ? 1.0f : __shfl(val_l, p - 1, 64);
            const f32x4* er = reinterpret_cast<const f32x4*>(E + (size_t)tok * Ed);
            f32x4* xr = reinterpret_cast<f32x4*>(x_emb + ((size_t)p * B + b) * Ed);
            for (int c = lane; c < E4; c += 64) xr[c] = er[c] * v;
        }
    }
    if (wv != 0) return;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) {
        idx[(size_t)b * (T + 1)] = bos;
        val[(size_t)b * (T + 1)] = 1.0f;
        int n = 2 + cnt;
        if (n > L + 1) n = L + 1;
        len[b] = n;
    }
}
// ground-truth labels: idx = labels, lens = sum(masks > 0)   (VSEFCModel.py:83-85)
__global__ void prep_labels_kernel(const int64_t* __restrict__ labels, const float* __restrict__ masks, int B, int Lp,
                                   int32_t* __restrict__ idx, float* __restrict__ val, int32_t* __restrict__ len,
                                   unsigned* __restrict__ sync, int nsync) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = b; i < nsync; i += gridDim.x * blockDim.x) sync[i] = 0u;    // hand-off counters of gru_seq_kernel
    if (b >= B) return;
    int n = 0;
    for (int j = 0; j < Lp; ++j) {
        idx[(size_t)b * Lp + j] = (int32_t)labels[(size_t)b * Lp + j];
        val[(size_t)b * Lp + j] = 1.0f;
        if (masks[(size_t)b * Lp + j] > 0.f) ++n;
    }
    len[b] = n;
}

// x_emb[t,b,:] = val[b,t] * E[idx[b,t],:]     (time-major so each GRU step is one contiguous slab)
__global__ __launch_bounds__(256) void embed_st_fwd_kernel(const float* __restrict__ E, const int32_t* __restrict__ idx,
                                                           const float* __restrict__ val, float* __restrict__ x,
                                                           int B, int Lp, int Ed) {
    const int E4 = Ed >> 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)Lp * B * E4) return;
    const int j = (int)(i % E4);
    const int b = (int)((i / E4) % B), t = (int)(i / ((int64_t)E4 * B));
    const int tok = idx[(size_t)b * Lp + t];
    const float v = val[(size_t)b * Lp + t];
    f32x4 e = reinterpret_cast<const f32x4*>(E + (size_t)tok * Ed)[j];
    reinterpret_cast<f32x4*>(x)[i] = e * v;
}
// dE[idx[b,t],:] += val[b,t] * dx[t,b,:]   for t < len[b]  (dx is already 0 elsewhere)
__global__ __launch_bounds__(256) void embed_st_bwd_kernel(const float* __restrict__ dx, const int32_t* __restrict__ idx,
                                                           const float* __restrict__ val, const int32_t* __restrict__ len,
                                                           float* __restrict__ dE, int B, int Lp, int Ed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)Lp * B * Ed) return;
    const int j = (int)(i % Ed);
    const int b = (int)((i / Ed) % B), t = (int)(i / ((int64_t)Ed * B));
    if (t >= len[b]) return;
    const int tok = idx[(size_t)b * Lp + t];
    atomicAdd(dE + (size_t)tok * Ed + j, val[(size_t)b * Lp + t] * dx[i]);
}

// ---- GRU cell (torch.nn.GRU gate order r, z, n) ------------------------------------------
__global__ __launch_bounds__(256) void gru_cell_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                           const float* __restrict__ h, const int32_t* __restrict__ len,
                                                           int t, float* __restrict__ h_new, int B, int J) {
    const int J4 = J >> 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * J4) return;
    const int b = idx / J4, j = idx % J4;
    const f32x4 hp = reinterpret_cast<const f32x4*>(h)[idx];
    f32x4 hn = hp;
    if (t < len[b]) {
        const f32x4* a = reinterpret_cast<const f32x4*>(gi + (size_t)b * 3 * J);
        const f32x4* c = reinterpret_cast<const f32x4*>(gh + (size_t)b * 3 * J);
        const f32x4 ir = a[j], iz = a[J4 + j], in = a[2 * J4 + j];
        const f32x4 hr = c[j], hz = c[J4 + j], hnn = c[2 * J4 + j];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float r = fast_sigmoid(ir[e] + hr[e]);
            const float z = fast_sigmoid(iz[e] + hz[e]);
            const float n = fast_tanh(in[e] + r * hnn[e]);
            hn[e] = (1.0f - z) * n + z * hp[e];
        }
    }
    reinterpret_cast<f32x4*>(h_new)[idx] = hn;
}

// ---- fused GRU step (flagship width J = 1024) ---------------------------------------------------------------------
// gh = h W_hh^T + b_hh and the gate arithmetic of one time step in ONE launch.  A workgroup owns a 32-row strip of
// the batch and one 16-wide tile j of hidden units: it computes the three gate tiles (r, z, n: weight rows j, J+j,
// 2J+j) on v_mfma_f32_16x16x4_f32 (two 16-row MFMA tiles per gate share every B fragment), the K range split over
// its KS = 8 waves (the A fragments of the strip's h rows stay in registers, the B fragments of the next gate are in
// flight under the current gate's MFMA chain), sums the KS partial tiles through LDS and applies the cell to the
// outputs it holds: the pre-activations never make a round trip through memory before the cell and a step is one
// launch instead of a GEMM + a cell kernel.  16-wide unit tiles give 4 x 64 = 256 workgroups at B = 128: one per CU.
// gh is still written (the backward pass reads it).  Step 0 (h = 0) skips the products: gh = b_hh.
//   16x16x4 operand layout: lane l -> (i = l & 15, q = l >> 4); A: row i, k = 4q+s for MFMA s of a 16-k group
//   (one dwordx4 per group); B: column i likewise; D: 4 registers v: row 4q+v, column i.
typedef float f32x4acc __attribute__((ext_vector_type(4)));
template <int GPS, int KS>   // GPS groups of 16 k per wave: K slice = 16*GPS, J = 16*GPS*KS
__global__ __launch_bounds__(KS * 64) void gru_step_fused_kernel(const float* __restrict__ h, const float* __restrict__ W,
                                                             const float* __restrict__ b_hh, const float* __restrict__ gi,
                                                             const int32_t* __restrict__ len, int t, float* __restrict__ gh_out,
                                                             float* __restrict__ h_new, int B, int J, int h_is_zero) {
    static_assert(KS == 8, "8 accumulator registers (2 row tiles x 4) dealt one per wave");
    __shared__ float red[2 * KS * 8 * 64];
    const int tid = threadIdx.x, lane = tid & 63, ks = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int tiles_j = J / 16;
    const int strip = blockIdx.x / tiles_j, jt = blockIdx.x % tiles_j;   // blockIdx % 8 == jt % 8: a weight tile's readers share an XCD
    const int m0 = strip * 32;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // the output this wave finishes after the cross-wave sums: register ks of the 8 -> row tile ks>>2, register ks&3
    const int col = jt * 16 + li;
    const int orow = m0 + 16 * (ks >> 2) + 4 * lq + (ks & 3);
    const int orc = orow < B ? orow : B - 1;
    const float gir = gi[(size_t)orc * 3 * J + col], giz = gi[(size_t)orc * 3 * J + J + col], gin = gi[(size_t)orc * 3 * J + 2 * J + col];
    const float hp = h[(size_t)orc * J + col];
    const int ln = len[orc];
    float ghv[3];
    if (!h_is_zero) {
        f32x4 af[2][GPS];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int m = m0 + 16 * rt + li;
            const int mc = m < B ? m : B - 1;
#pragma unroll
            for (int i = 0; i < GPS; ++i) {
                const int k = 16 * (ks * GPS + i) + 4 * lq;
                const f32x4 a = *reinterpret_cast<const f32x4*>(h + (size_t)mc * J + k);
                af[rt][i] = m < B ? a : z4;
            }
        }
        f32x4 bf[GPS];
        auto load_b = [&](int g) {
            const float* wrow = W + ((size_t)g * J + col) * J;
#pragma unroll
            for (int i = 0; i < GPS; ++i) bf[i] = *reinterpret_cast<const f32x4*>(wrow + 16 * (ks * GPS + i) + 4 * lq);
        };
        load_b(0);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            f32x4acc acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < GPS; ++i)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][i][s], bf[i][s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][i][s], bf[i][s], acc1, 0, 0, 0);
                }
            if (g < 2) load_b(g + 1);                      // the next gate's weight tile, in flight under the sums below
            float* rb = red + (g & 1) * (KS * 8 * 64);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                rb[(ks * 8 + v) * 64 + lane] = acc0[v];
                rb[(ks * 8 + 4 + v) * 64 + lane] = acc1[v];
            }
            __syncthreads();
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < KS; ++w) v += rb[(w * 8 + ks) * 64 + lane];
            ghv[g] = v + b_hh[g * J + col];
        }
    } else {
#pragma unroll
        for (int g = 0; g < 3; ++g) ghv[g] = b_hh[g * J + col];
    }
    // the cell (torch.nn.GRU gate order r, z, n) on the output this lane holds; rows past their caption's length keep h
    if (orow < B) {
        const size_t o = (size_t)orow * 3 * J + col;
        gh_out[o] = ghv[0];
        gh_out[o + J] = ghv[1];
        gh_out[o + 2 * J] = ghv[2];
        float hn = hp;
        if (t < ln) {
            const float rr = fast_sigmoid(gir + ghv[0]);
            const float zz = fast_sigmoid(giz + ghv[1]);
            const float nn = fast_tanh(gin + rr * ghv[2]);
            hn = (1.0f - zz) * nn + zz * hp;
        }
        h_new[(size_t)orow * J + col] = hn;
    }
}

// ---- the whole GRU pass in ONE launch: recurrent weights stationary in the workgroups --------------------------------
// gru_step_fused_kernel re-streams its weight tile (3 gates x 16 units x J floats) in every one of the Lp steps although it
// never changes.  This kernel is that step kernel with the time loop inside and the weight tile held by the workgroup: same
// per-element arithmetic (K split over 8 waves, same MFMA order, same cross-wave sums, same gate arithmetic) - so h and gh
// come out bit for bit as from the per-step launches -, but W_hh is read once, there are no launch boundaries, and a step
// moves only the strip's h rows (one workgroup per CU, all of them resident).
//
// What a step t >= 1 needs from OTHER workgroups is h_t of its own strip, written by the workgroups of that strip in
// step t-1.  Hand-off per strip and step (MI355X_MICROARCH.md, inter-workgroup visibility, first row of the sc1
// table; hipMalloc'ed memory, one workgroup per CU):
//   producer: every lane stores its h element write-through (buffer_store ... sc1), every wave drains (s_waitcnt
//             vmcnt(0)), workgroup barrier, ONE lane adds 1 to the strip's counter of step t+1 (agent-scope atomic);
//   consumer: ONE lane polls that counter (relaxed agent-scope load = global_load sc1, s_sleep between polls) until it
//             reads tiles_j, workgroup barrier, then EVERY load of the handed-off rows is a buffer_load ... sc1.
// No fence on either side; the counters are zeroed by the token-preparation kernel that precedes this launch on the
// stream.  Nothing here depends on dispatch order or placement; it needs every workgroup RESIDENT (grid <= CUs, checked
// by the launcher, which otherwise uses the per-step kernel).  Every spin is bounded by a wall-clock budget (1 s): a workgroup
// that gives up raises *err and poisons its outputs with NaN (the loss then says so) instead of hanging the device.
// (bound, error word, sticky status word and the development build's fault injection: HandoffGuard, cic_common.h)
// Tiling (r3): a workgroup owns a 16-row strip x a 32-unit tile (its backward twin's shape): 8 strips x 32 tiles at B = 128, so
// that the 32 workgroups of a strip share ONE XCD (blockIdx % 8: the hand-off stays inside an L2's reach; speed only) and a
// step moves 64 KB of h rows per workgroup instead of 128.  The weight tile (3 gates x 32 units x J floats = 393 KB) is split:
// GRUF_WR of a wave's 48 fragments in registers, the rest in LDS in fragment order.  Per element the arithmetic is unchanged
// (same K slice per wave, same MFMA order, same cross-wave order), so the pass is still bit-identical to the per-step kernel.
// BF (r4, cic_listener_dims.compute_dtype = bf16): the weight tile is held as bf16 MFMA fragments (v_mfma_f32_16x16x32_bf16: 24
// fragments of 8 k per lane = 96 VGPRs, none in LDS), the strip's h rows are rounded to bf16 as they are loaded, accumulation, gate
// arithmetic and everything stored stay f32: a step's products are 24 MFMAs of 16 cycles per wave instead of 192 of 32.  Same tiling,
// hand-off and output order.
typedef __bf16 lbf16x8_t __attribute__((ext_vector_type(8)));
typedef float lf32x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ lbf16x8_t l_to_bf16x8(const f32x4 lo, const f32x4 hi) {
    const lf32x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_convertvector(v, lbf16x8_t);       // round to nearest even; a NaN stays a NaN
}
constexpr int GRUF_WR = 33, GRUF_WL = 48 - GRUF_WR;
constexpr size_t GRUF_LDS_BYTES = sizeof(float) * ((size_t)8 * GRUF_WL * 64 * 4 + 2 * 8 * 8 * 64);
template <int GPS, int KS, bool BF = false>
__global__ __launch_bounds__(KS * 64) void gru_seq_kernel(float* __restrict__ h_all, const float* __restrict__ W,
                                                          const float* __restrict__ b_hh, const float* __restrict__ gi_all,
                                                          const int32_t* __restrict__ len, float* __restrict__ gh_all,
                                                          unsigned* __restrict__ cnt_base, HandoffGuard hg, int B, int J, int Lp,
                                                          int row0, int row_end) {
    // rows [row0, row_end) of the batch: a batch of more than CUs / 32 strips is walked in row blocks, one launch each (rows are
    // independent of each other); B stays the row count of the slabs
    static_assert(KS == 8 && GPS == 8, "J = 1024: 8 k groups of 16 per wave; 8 accumulator registers (2 unit tiles x 4) dealt one per wave");
    constexpr int WR = GRUF_WR, WL = GRUF_WL;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x4* wl = reinterpret_cast<f32x4*>(lds);            // [KS][WL][64] B fragments
    float* red = lds + (size_t)KS * WL * 64 * 4;          // [2][KS][8][64]
    __shared__ int ok_s;
    const int tid = threadIdx.x, lane = tid & 63, ks = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int tiles_j = J / 32, strips = gridDim.x / tiles_j;
    int strip = blockIdx.x / tiles_j, jt = blockIdx.x % tiles_j;
    if (strips <= 8 && (8 % strips) == 0 && (tiles_j % (8 / strips)) == 0) {
        // speed only: blocks b and b + 8 share an XCD - the workgroups of a strip (they exchange h among themselves) on as few
        // XCDs as possible
        const int xs = 8 / strips, xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        strip = xcd / xs;
        jt = (xcd % xs) * (tiles_j / xs) + local;
    }
    const int m0 = row0 + strip * 16;
    const int col = jt * 32 + 16 * (ks >> 2) + li;        // the output this wave finishes: register ks & 3 of unit tile ks >> 2
    const int orow = m0 + 4 * lq + (ks & 3);
    const int orc = orow < row_end ? orow : row_end - 1;
    const int ln = len[orc];
    unsigned* cnt = cnt_base + (size_t)(row0 / 16 + strip) * (Lp + 1);  // cnt[t]: workgroups of this strip that have published h_t
    const size_t slab = (size_t)B * J;
    // the weight tile, once: B fragments f = (gate g, unit tile ct, k group i) of this wave's K slice
    f32x4 wf[BF ? 1 : WR];
    lbf16x8_t wb[BF ? 24 : 1];                             // BF: fragment (gate g, unit tile ct, k-step j of 32): k = 128 ks + 32 j + 8 lq + 0..7
    if (BF) {
#pragma unroll
        for (int f = 0; f < 24; ++f) {
            const int g = f / 8, ct = (f / 4) & 1, j = f & 3;
            const float* q = W + ((size_t)g * J + jt * 32 + 16 * ct + li) * J + 128 * ks + 32 * j + 8 * lq;
            wb[BF ? f : 0] = l_to_bf16x8(*reinterpret_cast<const f32x4*>(q), *reinterpret_cast<const f32x4*>(q + 4));
        }
    } else {
#pragma unroll
        for (int f = 0; f < 48; ++f) {
            const int g = f / 16, ct = (f / 8) & 1, i = f & 7;
            const float* wrow = W + ((size_t)g * J + jt * 32 + 16 * ct + li) * J;
            const f32x4 v = *reinterpret_cast<const f32x4*>(wrow + 16 * (ks * GPS + i) + 4 * lq);
            if (f < WR) wf[(!BF && f < WR) ? f : 0] = v;
            else wl[(ks * WL + (f - WR)) * 64 + lane] = v;
        }
    }
    const float bh0 = b_hh[col], bh1 = b_hh[J + col], bh2 = b_hh[2 * J + col];
    float poison = 0.f;
    if (orow < row_end) h_all[(size_t)orow * J + col] = 0.f;    // h_0 (read by the backward pass; this kernel never reads it)
    if (tid == 0) ok_s = 1;                               // sticky: a workgroup that has given up once does not wait again
    __syncthreads();
    for (int t = 0; t < Lp; ++t) {
        const float* gi = gi_all + (size_t)t * B * 3 * J;
        float* gh_out = gh_all + (size_t)t * B * 3 * J;
        float* h_new = h_all + (size_t)(t + 1) * slab;
        // this step's input projections: written before the launch, independent of the hand-off
        const float gir = gi[(size_t)orc * 3 * J + col], giz = gi[(size_t)orc * 3 * J + J + col],
                    gin = gi[(size_t)orc * 3 * J + 2 * J + col];
        float ghv[3] = {bh0, bh1, bh2};
        float hp = 0.f;
        if (t > 0) {
            if (tid == 0 && ok_s) ok_s = handoff_poll(cnt + t, (unsigned)tiles_j, hg);
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    // compiler only: no load of h_t above the poll
            if (!ok_s) poison = __builtin_nanf("");
            // h_t of the strip: every load of the handed-off bytes is sc1 (aux 16)
            const auto hsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(h_all + (size_t)t * slab), 0,
                                                                (int)(slab * sizeof(float)), 0x00020000);
            f32x4 af[GPS];
            const int mc = min(m0 + li, row_end - 1);       // rows past the block repeat its last row: their sums are never stored
#pragma unroll
            for (int i = 0; i < GPS; ++i) {
                // (BF: the same eight 16-byte loads per lane, as four runs of eight consecutive k: 128 ks + 32 j + 8 lq + 0..7)
                const int k = BF ? 128 * ks + 32 * (i >> 1) + 8 * lq + 4 * (i & 1) : 16 * (ks * GPS + i) + 4 * lq;
                af[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hsrc, (int)(((size_t)mc * J + k) * 4), 0, 16));
            }
            hp = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hsrc, (int)(((size_t)orc * J + col) * 4), 0, 16));
            __builtin_amdgcn_sched_barrier(0);      // all of the wave's h rows are requested before the first MFMA waits
            lbf16x8_t ab[BF ? 4 : 1];
            if (BF) {
#pragma unroll
                for (int j = 0; j < 4; ++j) ab[BF ? j : 0] = l_to_bf16x8(af[2 * j], af[2 * j + 1]);
            }
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                f32x4acc acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                if (BF) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[BF ? j : 0], wb[BF ? g * 8 + j : 0], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[BF ? j : 0], wb[BF ? g * 8 + 4 + j : 0], acc1, 0, 0, 0);
                    }
                } else {
#pragma unroll
                for (int i = 0; i < GPS; ++i) {
                    const int f0 = g * 16 + i, f1 = g * 16 + 8 + i;
                    const f32x4 b0 = (!BF && f0 < WR) ? wf[(!BF && f0 < WR) ? f0 : 0] : wl[(ks * WL + (f0 < WR ? 0 : f0 - WR)) * 64 + lane];
                    const f32x4 b1 = (!BF && f1 < WR) ? wf[(!BF && f1 < WR) ? f1 : 0] : wl[(ks * WL + (f1 < WR ? 0 : f1 - WR)) * 64 + lane];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], b0[s], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], b1[s], acc1, 0, 0, 0);
                    }
                }
                }
                float* rb = red + (g & 1) * (KS * 8 * 64);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    rb[(ks * 8 + v) * 64 + lane] = acc0[v];
                    rb[(ks * 8 + 4 + v) * 64 + lane] = acc1[v];
                }
                __syncthreads();
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < KS; ++w) v += rb[(w * 8 + ks) * 64 + lane];
                ghv[g] = v + (g == 0 ? bh0 : (g == 1 ? bh1 : bh2));
            }
        }
        if (orow < row_end) {
            const size_t o = (size_t)orow * 3 * J + col;
            gh_out[o] = ghv[0];
            gh_out[o + J] = ghv[1];
            gh_out[o + 2 * J] = ghv[2];
            float hn = hp;
            if (t < ln) {
                const float rr = fast_sigmoid(gir + ghv[0]);
                const float zz = fast_sigmoid(giz + ghv[1]);
                const float nn = fast_tanh(gin + rr * ghv[2]);
                hn = (1.0f - zz) * nn + zz * hp;
            }
            if (poison != 0.f) hn = poison;      // NaN != 0: a workgroup that gave up on a hand-off marks what it produced
            // publish: write-through store of the one element this lane owns
            const auto hdst = __builtin_amdgcn_make_buffer_rsrc(h_new, 0, (int)(slab * sizeof(float)), 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hn), hdst, (int)(((size_t)orow * J + col) * 4), 0, 16);
        }
        if (t + 1 < Lp) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // EVERY storing wave drains before the signal
            __syncthreads();
            if (tid == 0) handoff_arrive(cnt + t + 1, hg);
        }
    }
}

// ---- the GRU's BPTT loop in ONE launch -------------------------------------------------------------------------------
// The backward twin of gru_seq_kernel.  Per step the plain form launches gru_cell_bwd_kernel (d h_{t+1} -> d gates) and a
// product dh_t += dgh_t W_hh that re-streams the 12.6 MB of W_hh from 32 fat workgroups (5 + 17 us, Lp times).  Here 256
// workgroups - a 16-row strip of the batch x a 32-unit tile each, one per CU - stay resident for the whole loop:
//   * a workgroup keeps the COLUMNS jt*32 .. +32 of W_hh (all 3J rows: the contraction index of dX = dY W) as MFMA B fragments,
//     K = 3J split over its 8 waves: 393 KB, two thirds of them in registers (128 VGPRs per lane) and one third in LDS
//     (128 KB, stored in fragment order: one conflict-free ds_read_b128 per fragment), read from memory once;
//   * the gradient w.r.t. the hidden state never goes to memory: after the cross-wave sum every lane holds dh of ONE
//     (row, unit), which is exactly what the gate derivative of the previous step needs at that (row, unit) - the cell
//     backward runs on it in place and yields the three dgi / three dgh values of that unit;
//   * what a step needs from the other 31 workgroups of its strip are the strip's dgh_t rows (16 x 3J floats: the A operand
//     of the product): published and consumed with the hand-off of gru_seq_kernel (write-through stores, drain, barrier, one
//     counter add; one poller, barrier, sc1 loads) on counters of their own.
// Why 16 x 32 and not the forward kernel's 32 x 16: bytes handed over inside a launch are served at the fabric's rate
// (~7.4 TB/s chip-wide measured, whichever XCD the reader sits on and whether it loads sc1 or acquires and loads plainly:
// tools/gru_stamps.py), and every workgroup of a strip reads ALL of the strip's dgh rows, so a step moves
// rows-per-strip x 12 KB x 256 workgroups: 100 MB (13.6 us) with 32-row strips, 50 MB with 16 - about the 5.9 us of the step's
// f32 MFMAs.  With 8 strips x 32 tiles the workgroups of a strip also share ONE XCD (blockIdx % 8; speed only).
// Steps at or beyond the longest caption of a strip carry no gate gradient: their slabs are zeroed and neither the hand-off
// nor the product runs (the same decision in every workgroup of the strip: it depends on the strip's lengths only).
constexpr int GRUB_GPS = 24, GRUB_GR = 16, GRUB_GL = GRUB_GPS - GRUB_GR;       // k groups of 16 per wave: in registers / in LDS
constexpr size_t GRUB_LDS_BYTES = sizeof(float) * ((size_t)8 * GRUB_GL * 2 * 64 * 4 + 8 * 8 * 64);
// BF (r4): as gru_seq_kernel's - the wave's 384 k of the weight columns as 12 x 2 bf16 fragments (96 VGPRs, none in LDS), the strip's
// dgh rows rounded to bf16 as they are loaded (chunks of four k-steps, two in flight), f32 accumulation and gate arithmetic.
template <int KS, bool BF = false>   // 3J = 16 * GRUB_GPS * KS
__global__ __launch_bounds__(KS * 64) void gru_seq_bwd_kernel(const float* __restrict__ h_all, const float* __restrict__ W,
                                                              const float* __restrict__ gi_all, const float* __restrict__ gh_all,
                                                              const int32_t* __restrict__ len, const float* __restrict__ dh_init,
                                                              float* __restrict__ dgi_all, float* __restrict__ dgh_all,
                                                              unsigned* __restrict__ cnt_base, HandoffGuard hg,
                                                              int B, int J, int Lp, int pool, const float* __restrict__ d_pool,
                                                              const int32_t* __restrict__ pool_arg,
                                                              // [Lp,B,zero_n] cleared on the way (zero_n <= J: the K-sliced product
                                                              // dx_emb = dgi W_ih after the loop adds into it), or null
                                                              float* __restrict__ zero_lbe, int zero_n, int row0, int row_end) {
    // rows [row0, row_end) of the batch (see gru_seq_kernel)
    static_assert(KS == 8, "8 accumulator registers (2 column tiles x 4) dealt one per wave");
    constexpr int GPS = GRUB_GPS, GR = GRUB_GR, GL = GRUB_GL;
    constexpr int GPC = 4, NCH = GPS / GPC, NBUF = 3;     // the A rows arrive in chunks of GPC groups, NBUF chunks in flight
    static_assert(GPS % GPC == 0 && GR % GPC == 0 && NBUF <= NCH, "whole chunks");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x4* wl = reinterpret_cast<f32x4*>(lds);            // [KS][GL][2 column tiles][64 lanes] B fragments
    float* red = lds + (size_t)KS * GL * 2 * 64 * 4;      // [KS][8][64]
    __shared__ int ok_s, smax_s;
    const int tid = threadIdx.x, lane = tid & 63, ks = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int tiles_j = J / 32, strips = gridDim.x / tiles_j;
    int strip = blockIdx.x / tiles_j, jt = blockIdx.x % tiles_j;
    if (strips <= 8 && (8 % strips) == 0 && (tiles_j % (8 / strips)) == 0) {
        // speed only: blocks b and b + 8 share an XCD - the workgroups of a strip (they read the same dgh rows) on as few
        // XCDs as possible
        const int xs = 8 / strips, xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
        strip = xcd / xs;
        jt = (xcd % xs) * (tiles_j / xs) + local;
    }
    const int m0 = row0 + strip * 16;
    const int col = jt * 32 + 16 * (ks >> 2) + li;          // the output this wave finishes: register ks & 3 of column tile ks >> 2
    const int orow = m0 + 4 * lq + (ks & 3);
    const int orc = orow < row_end ? orow : row_end - 1;
    const int ln = len[orc];
    const int J3 = 3 * J;
    unsigned* cnt = cnt_base + (size_t)(row0 / 16 + strip) * (Lp + 1);    // cnt[t]: workgroups of this strip that have published dgh_t
    if (tid == 0) { smax_s = 0; ok_s = 1; }               // ok_s is sticky: a workgroup that has given up once does not wait again
    __syncthreads();
    if (tid < 16) atomicMax(&smax_s, len[min(m0 + tid, row_end - 1)]);
    // the weight tile, once: B fragments (k = 16*group + 4*lq + s, n = column li of tile ct) of this wave's K slice
    f32x4 wf[BF ? 1 : GR][2];
    constexpr int JSB = GPS / 2;                          // BF: k-steps of 32 per wave (12): k = 384 ks + 32 j + 8 lq + 0..7
    lbf16x8_t wb[BF ? JSB : 1][2];
    if (BF) {
#pragma unroll
        for (int j = 0; j < JSB; ++j) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const float* wk = W + (size_t)(16 * GPS * ks + 32 * j + 8 * lq) * J + jt * 32 + 16 * ct + li;
                f32x4 lo, hi;
#pragma unroll
                for (int e = 0; e < 4; ++e) { lo[e] = wk[(size_t)e * J]; hi[e] = wk[(size_t)(4 + e) * J]; }
                wb[BF ? j : 0][ct] = l_to_bf16x8(lo, hi);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < GPS; ++i) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const float* wk = W + (size_t)(16 * (ks * GPS + i) + 4 * lq) * J + jt * 32 + 16 * ct + li;
                f32x4 v;
#pragma unroll
                for (int s = 0; s < 4; ++s) v[s] = wk[(size_t)s * J];
                if (i < GR) wf[(!BF && i < GR) ? i : 0][ct] = v;
                else wl[((ks * GL + (i - GR)) * 2 + ct) * 64 + lane] = v;
            }
        }
    }
    __syncthreads();
    const int smax = smax_s;
    const size_t gslab = (size_t)B * J3;
    float d = dh_init[(size_t)orc * J + col];              // d loss / d h_{t+1} at this lane's (row, unit)
    float poison = 0.f;
    unsigned long long* stamps = CIC_STAMP_BUF(g_gru_stamps);
#define GRU_STAMP(i) if (stamps && tid == 0) stamps[((size_t)blockIdx.x * Lp + t) * 8 + (i)] = __builtin_amdgcn_s_memrealtime()
    for (int t = Lp - 1; t >= 0; --t) {
        GRU_STAMP(0);
        const size_t o = (size_t)orc * J3 + col;
        float* dgi = dgi_all + (size_t)t * gslab;
        float* dgh = dgh_all + (size_t)t * gslab;
        const auto hdst = __builtin_amdgcn_make_buffer_rsrc(dgh, 0, (int)(gslab * sizeof(float)), 0x00020000);
        if (zero_lbe && orow < row_end && col < zero_n) zero_lbe[((size_t)t * B + orow) * zero_n + col] = 0.f;
        if (t >= smax) {
            // no caption of the strip reaches step t: zero gate gradients, d passes through (no pooled term either)
            if (orow < row_end) {
                dgi[o] = 0.f; dgi[o + J] = 0.f; dgi[o + 2 * J] = 0.f;
                dgh[o] = 0.f; dgh[o + J] = 0.f; dgh[o + 2 * J] = 0.f;
            }
            continue;
        }
        // ---- gate derivative of step t at (orow, col): gru_cell_bwd_kernel on one element -------------------------------
        {
            const float* gi = gi_all + (size_t)t * gslab;
            const float* gh = gh_all + (size_t)t * gslab;
            const float ir = gi[o], iz = gi[o + J], in = gi[o + 2 * J];
            const float hr = gh[o], hz = gh[o + J], hnn = gh[o + 2 * J];
            const float hp = h_all[(size_t)t * B * J + (size_t)orc * J + col];
            if (pool && t < ln) {
                const float dp = d_pool[(size_t)orc * J + col];
                if (pool == 1) d += dp * (1.0f / (float)ln);
                else if (pool_arg[(size_t)orc * J + col] == t) d += dp;
            }
            float gir = 0.f, giz = 0.f, gin = 0.f, ghn = 0.f, dprev = d;
            if (t < ln) {
                const float r = fast_sigmoid(ir + hr);
                const float z = fast_sigmoid(iz + hz);
                const float n = fast_tanh(in + r * hnn);
                const float dn = d * (1.0f - z);
                const float dz = d * (hp - n);
                const float dan = dn * (1.0f - n * n);
                const float dr = dan * hnn;
                gir = dr * r * (1.0f - r);
                giz = dz * z * (1.0f - z);
                gin = dan;
                ghn = dan * r;
                dprev = d * z;
            }
            if (poison != 0.f) { gir = poison; giz = poison; gin = poison; ghn = poison; }
            if (orow < row_end) {
                dgi[o] = gir; dgi[o + J] = giz; dgi[o + 2 * J] = gin;
                // dgh_t is handed to the strip's other workgroups: write-through stores
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gir), hdst, (int)(o * 4), 0, 16);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, giz), hdst, (int)((o + J) * 4), 0, 16);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ghn), hdst, (int)((o + 2 * J) * 4), 0, 16);
            }
            d = dprev;
        }
        if (t == 0) break;                                  // h_0 is the constant zero state: nothing flows further
        // ---- publish dgh_t, wait for the strip, dh_t = d h direct + dgh_t W_hh --------------------------------------------
        GRU_STAMP(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // EVERY storing wave drains before the signal
        __syncthreads();
        GRU_STAMP(2);
        if (tid == 0) {
            handoff_arrive(cnt + t, hg);
            if (ok_s) ok_s = handoff_poll(cnt + t, (unsigned)tiles_j, hg);
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    // compiler only: no load of dgh_t above the poll
        if (!ok_s) poison = __builtin_nanf("");
        GRU_STAMP(3);
        const int mc = min(m0 + li, row_end - 1);           // rows past the block repeat its last row: their sums are never stored
        f32x4acc acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (BF) {
            // chunks of four k-steps (eight 16-byte loads per lane), two chunks in flight
            constexpr int CS = 4, NCB = JSB / CS;
            f32x4 ab[2][2 * CS];
            auto load_chunk_b = [&](int c) {                // every load of the handed-off bytes is sc1 (aux 16)
#pragma unroll
                for (int i = 0; i < 2 * CS; ++i) {
                    const int k = 16 * GPS * ks + 32 * (c * CS + (i >> 1)) + 8 * lq + 4 * (i & 1);
                    ab[c & 1][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hdst, (int)(((size_t)mc * J3 + k) * 4), 0, 16));
                }
            };
            load_chunk_b(0);
            load_chunk_b(1);
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
#pragma unroll
                for (int jj = 0; jj < CS; ++jj) {
                    const lbf16x8_t a8 = l_to_bf16x8(ab[c & 1][2 * jj], ab[c & 1][2 * jj + 1]);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, wb[BF ? c * CS + jj : 0][0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, wb[BF ? c * CS + jj : 0][1], acc1, 0, 0, 0);
                }
                if (c + 2 < NCB) load_chunk_b(c + 2);       // into the registers this chunk's MFMAs have just read
            }
        }
        f32x4 af[NBUF][GPC];
        auto load_chunk = [&](int c) {                      // every load of the handed-off bytes is sc1 (aux 16)
#pragma unroll
            for (int i = 0; i < GPC; ++i) {
                const int k = 16 * (ks * GPS + c * GPC + i) + 4 * lq;
                af[c % NBUF][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hdst, (int)(((size_t)mc * J3 + k) * 4), 0, 16));
            }
        };
#pragma unroll
        for (int c = 0; c < NBUF && !BF; ++c) load_chunk(c);
#pragma unroll
        for (int c = 0; c < NCH && !BF; ++c) {
#pragma unroll
            for (int i = 0; i < GPC; ++i) {
                const int gidx = c * GPC + i;
                f32x4 b0, b1;
                if (!BF && gidx < GR) { b0 = wf[(!BF && gidx < GR) ? gidx : 0][0]; b1 = wf[(!BF && gidx < GR) ? gidx : 0][1]; }
                else {
                    b0 = wl[((ks * GL + (gidx - GR)) * 2 + 0) * 64 + lane];
                    b1 = wl[((ks * GL + (gidx - GR)) * 2 + 1) * 64 + lane];
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c % NBUF][i][s], b0[s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c % NBUF][i][s], b1[s], acc1, 0, 0, 0);
                }
            }
            if (c + NBUF < NCH) load_chunk(c + NBUF);       // into the registers this chunk's MFMAs have just read
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            red[(ks * 8 + v) * 64 + lane] = acc0[v];
            red[(ks * 8 + 4 + v) * 64 + lane] = acc1[v];
        }
        GRU_STAMP(4);
        __syncthreads();
        GRU_STAMP(5);
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < KS; ++w) v += red[(w * 8 + ks) * 64 + lane];
        d += v;                                             // red is rewritten only behind the next step's barriers
    }
#undef GRU_STAMP
}

// dh (in/out): gradient w.r.t. h_{t+1} in, w.r.t. the direct h_t path out (the W_hh path is added by a GEMM)
__global__ __launch_bounds__(256) void gru_cell_bwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                           const float* __restrict__ h, const int32_t* __restrict__ len,
                                                           int t, const float* __restrict__ dh_in,
                                                           float* __restrict__ dh_out, float* __restrict__ dgi,
                                                           float* __restrict__ dgh, int B, int J, int pool,
                                                           const float* __restrict__ d_pool,
                                                           const int32_t* __restrict__ pool_arg) {
    // pool != 0: the caption embedding pools the per-step outputs, so step t < len receives d_pool / len (mean)
    // or d_pool where it was the maximum (max) on top of the carried gradient
    const int J4 = J >> 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * J4) return;
    const int b = idx / J4, j = idx % J4;
    f32x4 d = reinterpret_cast<const f32x4*>(dh_in)[idx];
    if (pool && t < len[b]) {
        const f32x4 dp = reinterpret_cast<const f32x4*>(d_pool)[idx];
        if (pool == 1) {
            d += dp * (1.0f / (float)len[b]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (pool_arg[(size_t)idx * 4 + e] == t) d[e] += dp[e];
        }
    }
    f32x4* oi = reinterpret_cast<f32x4*>(dgi + (size_t)b * 3 * J);
    f32x4* oh = reinterpret_cast<f32x4*>(dgh + (size_t)b * 3 * J);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if (t >= len[b]) {
        oi[j] = zero; oi[J4 + j] = zero; oi[2 * J4 + j] = zero;
        oh[j] = zero; oh[J4 + j] = zero; oh[2 * J4 + j] = zero;
        reinterpret_cast<f32x4*>(dh_out)[idx] = d;
        return;
    }
    const f32x4 hp = reinterpret_cast<const f32x4*>(h)[idx];
    const f32x4* a = reinterpret_cast<const f32x4*>(gi + (size_t)b * 3 * J);
    const f32x4* c = reinterpret_cast<const f32x4*>(gh + (size_t)b * 3 * J);
    const f32x4 ir = a[j], iz = a[J4 + j], in = a[2 * J4 + j];
    const f32x4 hr = c[j], hz = c[J4 + j], hnn = c[2 * J4 + j];
    f32x4 gir, giz, gin, ghn, dprev;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float r = fast_sigmoid(ir[e] + hr[e]);
        const float z = fast_sigmoid(iz[e] + hz[e]);
        const float n = fast_tanh(in[e] + r * hnn[e]);
        const float dn = d[e] * (1.0f - z);
        const float dz = d[e] * (hp[e] - n);
        const float dan = dn * (1.0f - n * n);
        const float dr = dan * hnn[e];
        gir[e] = dr * r * (1.0f - r);
        giz[e] = dz * z * (1.0f - z);
        gin[e] = dan;
        ghn[e] = dan * r;
        dprev[e] = d[e] * z;
    }
    oi[j] = gir; oi[J4 + j] = giz; oi[2 * J4 + j] = gin;
    oh[j] = gir; oh[J4 + j] = giz; oh[2 * J4 + j] = ghn;
    reinterpret_cast<f32x4*>(dh_out)[idx] = dprev;
}

// ---- pooling over the valid GRU outputs (vse_pool_type 'mean' / 'max', VSEFCModel.py:118-127) ----------------
// h_all: [Lp+1,B,J], slab t+1 = output of step t; pad_packed_sequence leaves zeros past a caption's length, which
// the masks exclude (mean: sum_{t<len} h_t / len;  max: the masked maximum, ties -> lowest t).  Masks are the 0/1
// prefix masks the reference builds (len = their sum).
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ h_all, const int32_t* __restrict__ len,
                                                       int mode, float* __restrict__ pooled, int32_t* __restrict__ arg,
                                                       int B, int J, int Lp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * J) return;
    const int b = i / J;
    const int n = len[b];
    if (mode == 1) {
        float s = 0.f;
        for (int t = 0; t < n; ++t) s += h_all[(size_t)(t + 1) * B * J + i];
        pooled[i] = s / (float)n;
    } else {
        float m = -1e10f;                       // an all-masked row pools to the reference's fill value
        int a = -1;
        for (int t = 0; t < n; ++t) {
            const float v = h_all[(size_t)(t + 1) * B * J + i];
            if (v > m) { m = v; a = t; }
        }
        pooled[i] = m;
        arg[i] = a;
    }
}

// ---- l2norm (VSEFCModel.py:12-17): y = x / (||x|| + 1e-7) ---------------------------------
// two row sets in ONE launch (the image embeddings and the caption embeddings of a listener pass): workgroup b < B is row b of
// the first set, workgroup B + b row b of the second; per row y = x / (||x|| + 1e-7) (|y| with use_abs), nrm = ||x||
__global__ __launch_bounds__(256) void l2norm_fwd2_kernel(const float* __restrict__ x0, float* __restrict__ y0, float* __restrict__ n0,
                                                          int norm0, const float* __restrict__ x1, float* __restrict__ y1,
                                                          float* __restrict__ n1, int norm1, int B, int J, int use_abs) {
    __shared__ float sh[4];
    const bool second = (int)blockIdx.x >= B;
    const int b = second ? blockIdx.x - B : blockIdx.x;
    const float* x = second ? x1 : x0;
    float* y = second ? y1 : y0;
    float* nrm = second ? n1 : n0;
    const int do_norm = second ? norm1 : norm0;
    float s = 0.f;
    for (int j = threadIdx.x; j < J; j += 256) {
        const float v = x[(size_t)b * J + j];
        s += v * v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    const float n = sqrtf(sh[0] + sh[1] + sh[2] + sh[3]);
    const float inv = do_norm ? 1.0f / (n + 1e-7f) : 1.0f;
    for (int j = threadIdx.x; j < J; j += 256) {
        float v = x[(size_t)b * J + j] * inv;
        if (use_abs) v = fabsf(v);
        y[(size_t)b * J + j] = v;
    }
    if (threadIdx.x == 0) nrm[b] = n;
}
// dx = (dy' - yn * (yn . dy') * (n + eps) / n) / (n + eps),  yn = x/(n+eps), dy' = dy*sign(yn) if abs
// the backward twin of l2norm_fwd2_kernel: two row sets in one launch (gridDim.x = B: the second set only, rows b of
// (x1, n1, dy1, dx1); gridDim.x = 2B: workgroups [0, B) the first set, [B, 2B) the second)
__global__ __launch_bounds__(256) void l2norm_bwd2_kernel(const float* __restrict__ x0, const float* __restrict__ n0,
                                                          const float* __restrict__ dy0, float* __restrict__ dx0, int norm0,
                                                          const float* __restrict__ x1, const float* __restrict__ n1,
                                                          const float* __restrict__ dy1, float* __restrict__ dx1, int norm1,
                                                          int B, int J, int use_abs) {
    __shared__ float sh[4];
    const bool first = (int)gridDim.x == 2 * B && (int)blockIdx.x < B;
    const int b = ((int)gridDim.x == 2 * B && !first) ? blockIdx.x - B : blockIdx.x;
    const float* x = first ? x0 : x1;
    const float* nrm = first ? n0 : n1;
    const float* dy = first ? dy0 : dy1;
    float* dx = first ? dx0 : dx1;
    const int do_norm = first ? norm0 : norm1;
    const float n = nrm[b];
    const float inv = do_norm ? 1.0f / (n + 1e-7f) : 1.0f;
    float s = 0.f;
    for (int j = threadIdx.x; j < J; j += 256) {
        const float yn = x[(size_t)b * J + j] * inv;
        float g = dy[(size_t)b * J + j];
        if (use_abs) g = yn > 0.f ? g : (yn < 0.f ? -g : 0.f);
        s += yn * g;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    const float dot = sh[0] + sh[1] + sh[2] + sh[3];
    const float k = (n > 0.f) ? dot * (n + 1e-7f) / n : 0.f;
    for (int j = threadIdx.x; j < J; j += 256) {
        const float yn = x[(size_t)b * J + j] * inv;
        float g = dy[(size_t)b * J + j];
        if (use_abs) g = yn > 0.f ? g : (yn < 0.f ? -g : 0.f);
        dx[(size_t)b * J + j] = do_norm ? (g - yn * k) * inv : g;
    }
}

// ---- contrastive loss (VSEFCModel.py:167-207) ------------------------------------------------
// scores S[B,B] = im s^T from cic_gemm_f32.  One workgroup does the whole [B,B] reduction
// (B <= 1024): thread i owns row i (caption retrieval) and column i (image retrieval).
// out_rows[i] = sel_s*cost_s[i] + sel_im*cost_im[i];  out_sum = sum_i out_rows[i].
// one wave per row i: the row of cost_s and the column of cost_im reduced across the lanes (ties -> lowest index, as
// the strict > of a left-to-right scan)
__global__ __launch_bounds__(64) void contrastive_fwd_kernel(const float* __restrict__ S, int B, float margin,
                                                             int max_violation, int sel_s, int sel_im,
                                                             float* __restrict__ out_rows, int32_t* __restrict__ arg_s,
                                                             int32_t* __restrict__ arg_im, unsigned* __restrict__ arrived,
                                                             float* __restrict__ out_sum) {
    // arrived / out_sum: (r4) the scalar loss in the same launch.  Every workgroup (one wave) stores its row's loss write-through,
    // drains, and counts itself in; the one whose add comes last sums the B rows in row order with sc1 loads - exactly
    // contrastive_sum_kernel's chunks of 64 added in order, so the sum is bit-identical to the two-launch form.  `arrived` is
    // zeroed with the listener's hand-off counters by the token-preparation kernel.
    const int i = blockIdx.x, lane = threadIdx.x;
    const float d = S[(size_t)i * B + i];
    float cs = 0.f, ci = 0.f;
    int as = 0x7fffffff, ai = 0x7fffffff;
    for (int j = lane; j < B; j += 64) {
        if (j == i) continue;
        const float v = fmaxf(margin + S[(size_t)i * B + j] - d, 0.f);      // cost_s[i,j]   (:176)
        const float u = fmaxf(margin + S[(size_t)j * B + i] - d, 0.f);      // cost_im[j,i]  (:179)
        if (max_violation) {
            if (v > cs) { cs = v; as = j; }
            if (u > ci) { ci = u; ai = j; }
        } else {
            cs += v;
            ci += u;
        }
    }
    if (max_violation) {
        // masked_fill(diag, 0) then max: the maximum starts from 0 (no index) and a strict > keeps the first maximum
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(cs, o, 64);
            const int oi = __shfl_xor(as, o, 64);
            if (ov > cs || (ov == cs && oi < as)) { cs = ov; as = oi; }
            const float pv = __shfl_xor(ci, o, 64);
            const int pi = __shfl_xor(ai, o, 64);
            if (pv > ci || (pv == ci && pi < ai)) { ci = pv; ai = pi; }
        }
        if (cs <= 0.f) as = -1;
        if (ci <= 0.f) ai = -1;
    } else {
        cs = wave_sum(cs) / (float)B;
        ci = wave_sum(ci) / (float)B;
        as = ai = -1;
    }
    unsigned ticket = 0;
    if (lane == 0) {
        __hip_atomic_store(out_rows + i, (sel_s ? cs : 0.f) + (sel_im ? ci : 0.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arg_s) arg_s[i] = as;
        if (arg_im) arg_im[i] = ai;
        if (arrived) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ticket = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!arrived) return;
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    if (ticket != (unsigned)B - 1u) return;              // (wave-uniform)
    float s = 0.f;
    for (int j0 = 0; j0 < B; j0 += 64) {
        const int j = j0 + lane;
        s += wave_sum(j < B ? __hip_atomic_load(out_rows + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f);
    }
    if (lane == 0) *out_sum = s;
}
// dS from per-row upstream gradients g_rows[i] (scalar loss: all equal), one thread per entry of dS: what the reference's
// autograd scatters (row i: +g at its hardest negative(s), -g on the diagonal) is gathered per (i, j) - no zero-fill of dS
// before, no atomics, a fixed summation order.
__global__ __launch_bounds__(256) void contrastive_bwd_kernel(const float* __restrict__ S, int B, float margin,
                                                              int max_violation, int sel_s, int sel_im,
                                                              const float* __restrict__ g_rows, const float* __restrict__ g_scalar,
                                                              float g_scale, const int32_t* __restrict__ arg_s,
                                                              const int32_t* __restrict__ arg_im, float* __restrict__ dS,
                                                              unsigned* __restrict__ sync_bwd, int nsync_bwd) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    // hand-off counters of gru_seq_bwd_kernel (the first kernel of the backward pass clears them: a second backward pass
    // over the same forward state starts from zero as well)
    for (int k = idx; k < nsync_bwd; k += gridDim.x * blockDim.x) sync_bwd[k] = 0u;
    if (idx >= B * B) return;
    const int i = idx / B, j = idx % B;
    auto G = [&](int r) { return (g_rows ? g_rows[r] : *g_scalar) * g_scale; };
    float v = 0.f;
    if (max_violation) {
        // row r: dS[r][arg_s[r]] += g_r, dS[arg_im[r]][r] += g_r, dS[r][r] -= g_r for each of the two that exists
        if (sel_s && arg_s[i] == j) v += G(i);
        if (sel_im && arg_im[j] == i) v += G(j);
        if (i == j) {
            if (sel_s && arg_s[i] >= 0) v -= G(i);
            if (sel_im && arg_im[i] >= 0) v -= G(i);
        }
    } else {
        // row r, every c != r: margin + S[r][c] - S[r][r] > 0: dS[r][c] += g_r / B, dS[r][r] -= g_r / B;
        //                      margin + S[c][r] - S[r][r] > 0: dS[c][r] += g_r / B, dS[r][r] -= g_r / B
        const float inv_b = 1.0f / (float)B;
        if (i != j) {
            const float sij = S[(size_t)i * B + j];
            if (sel_s && margin + sij - S[(size_t)i * B + i] > 0.f) v += G(i) * inv_b;
            if (sel_im && margin + sij - S[(size_t)j * B + j] > 0.f) v += G(j) * inv_b;
        } else {
            const float d = S[(size_t)i * B + i], gb = G(i) * inv_b;
            for (int c = 0; c < B; ++c) {
                if (c == i) continue;
                if (sel_s && margin + S[(size_t)i * B + c] - d > 0.f) v -= gb;
                if (sel_im && margin + S[(size_t)c * B + i] - d > 0.f) v -= gb;
            }
        }
    }
    dS[idx] = v;
}

struct LstWs {
    unsigned* sync;           // gru_seq_kernel / gru_seq_bwd_kernel: 2 x [strips][Lp+1] hand-off counters + 1 error word (zeroed by the prep kernels)
    int nsync;
    int32_t *idx, *len, *arg_s, *arg_im;
    float *val, *x_emb, *gi_all, *gh_all, *h_all, *img_lin, *img_emb, *cap_emb, *nrm_img, *nrm_cap, *S;
    // backward scratch
    float *dS, *d_img, *d_cap, *d_lin, *dh, *dh2, *dgi_all, *dgh_all, *dx_emb;
    float *pooled, *d_pool;   // [B,J] vse_pool_type mean / max
    int32_t* pool_arg;        // [B,J] time step of the maximum
    size_t bytes;
};
LstWs lst_carve(const cic_listener_dims& d, void* base) {
    LstWs w;
    Carver c(base);
    const size_t B = d.B, J = d.J, E = d.E, Lp = d.Lp;
    // forward counters, backward counters (16-row strips each), error word, arrival counter of the contrastive loss's row sum
    w.nsync = (int)(2 * ((B + 15) / 16) * (Lp + 1) + 2);
    w.sync = reinterpret_cast<unsigned*>(c.i32((size_t)(w.nsync + 3) / 4 * 4));
    w.idx = c.i32(B * Lp);
    w.len = c.i32(B);
    w.arg_s = c.i32(B);
    w.arg_im = c.i32(B);
    w.val = c.f32(B * Lp);
    w.x_emb = c.f32(Lp * B * E);
    w.gi_all = c.f32(Lp * B * 3 * J);
    w.gh_all = c.f32(Lp * B * 3 * J);
    w.h_all = c.f32((Lp + 1) * B * J);
    w.img_lin = c.f32(B * J);
    w.img_emb = c.f32(B * J);
    w.cap_emb = c.f32(B * J);
    w.nrm_img = c.f32(B);
    w.nrm_cap = c.f32(B);
    w.S = c.f32(B * B);
    w.dS = c.f32(B * B);
    w.d_img = c.f32(B * J);
    w.d_cap = c.f32(B * J);
    w.d_lin = c.f32(B * J);
    w.dh = c.f32(B * J);
    w.dh2 = c.f32(B * J);
    w.dgi_all = c.f32(Lp * B * 3 * J);
    w.dgh_all = c.f32(Lp * B * 3 * J);
    w.dx_emb = c.f32(Lp * B * E);
    w.pooled = c.f32(B * J);
    w.d_pool = c.f32(B * J);
    w.pool_arg = c.i32(B * J);
    w.bytes = c.used();
    return w;
}

int check_ldims(const cic_listener_dims& d) {
    CIC_REQUIRE(d.B > 0 && d.B <= 1024 && d.Lp > 0 && d.Lp <= 128);
    CIC_REQUIRE((d.J & 3) == 0 && (d.E & 3) == 0 && d.F > 0 && d.V > 0);
    CIC_REQUIRE(d.pool >= 0 && d.pool <= 2);
    return 0;
}

}  // namespace

#ifdef CIC_DEVTOOLS
extern "C" int cic_debug_gru_fused(int on) {
    g_gru_fused = on;
    return 0;
}
extern "C" int cic_debug_set_gru_stamps(unsigned long long* buf) {
    CIC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_gru_stamps), &buf, sizeof(buf)));
    return 0;
}
#endif

extern "C" size_t cic_listener_ws_bytes(const cic_listener_dims* d) {
    if (!d) return 0;
    return lst_carve(*d, nullptr).bytes;
}

static int listener_fwd_impl(const cic_listener_dims* dp, const cic_listener_params* p, const cic_listener_io* io,
                             void* ws, size_t ws_bytes, cic_stream_t s);
static int listener_bwd_impl(const cic_listener_dims* dp, const cic_listener_params* p, const cic_listener_io* io,
                             const cic_listener_bwd_io* bio, void* ws, size_t ws_bytes, cic_stream_t s);

extern "C" int cic_listener_fwd(const cic_listener_dims* dp, const cic_listener_params* p, const cic_listener_io* io,
                                void* ws, size_t ws_bytes, cic_stream_t s) {
    CIC_REQUIRE(dp && p && io && ws);
    return listener_fwd_impl(dp, p, io, ws, ws_bytes, s);
}

extern "C" int cic_listener_bwd(const cic_listener_dims* dp, const cic_listener_params* p, const cic_listener_io* io,
                                const cic_listener_bwd_io* bio, void* ws, size_t ws_bytes, cic_stream_t s) {
    CIC_REQUIRE(dp && p && io && bio && ws);
    return listener_bwd_impl(dp, p, io, bio, ws, ws_bytes, s);
}

static int listener_fwd_impl(const cic_listener_dims* dp, const cic_listener_params* p, const cic_listener_io* io,
                             void* ws, size_t ws_bytes, cic_stream_t s) {
    CIC_REQUIRE(dp && p && io && ws);
    const cic_listener_dims& d = *dp;
    if (int rc = check_ldims(d)) return rc;
    LstWs w = lst_carve(d, ws);
    CIC_REQUIRE(ws_bytes >= w.bytes);
    CIC_REQUIRE(io->fc_feats && io->loss_rows && io->loss_sum);
    hipStream_t st = cic_s(s);
    // compute_dtype bf16: the text encoder's batched products on one bf16 part, its GRU pass on bf16 fragments; the image encoder, the
    // similarity matrix and the loss stay f32
    const bool bf = d.compute_dtype == CIC_DTYPE_BF16;
    const GemmCtx stb(st, bf ? CIC_PRECISION_BF16 : CIC_PRECISION_F32);
    const int B = d.B, J = d.J, E = d.E, Lp = d.Lp;
    int rc;
#define RUN(x) if ((rc = (x)) != 0) return rc
    // tokens
    bool embedded = false;         // generated captions: the preparation kernel embeds the rows itself
    if (io->labels) {
        CIC_REQUIRE(io->masks);
        hipLaunchKernelGGL(prep_labels_kernel, dim3(cic_cdiv(B, 256)), dim3(256), 0, st, io->labels, io->masks, B, Lp,
                           w.idx, w.val, w.len, w.sync, w.nsync);
    } else {
        CIC_REQUIRE(io->seq && io->L && Lp == d.T + 1);
        embedded = d.T < 64 && (E & 3) == 0;
        hipLaunchKernelGGL(prep_generated_kernel, dim3(B), dim3(256), 0, st, io->seq, io->stv, io->L, B,
                           d.T, d.V + 1, io->soft ? 1 : 0, w.idx, w.val, w.len, w.sync, w.nsync,
                           embedded ? p->embed_w : nullptr, embedded ? w.x_emb : nullptr, E);
    }
    CIC_LAUNCH_CHECK();
    // image encoder: l2norm(fc W^T + b)                                  (VSEFCModel.py:40-54); its l2norm shares a launch with
    // the caption's, after the GRU pass
    RUN(gemm_nt(io->fc_feats, d.F, p->img_fc_w, d.F, w.img_lin, J, B, J, d.F, p->img_fc_b, false, false, st));
    // text encoder                                                       (VSEFCModel.py:95-140)
    if (!embedded) {
        const int64_t n = (int64_t)Lp * B * (E / 4);
        hipLaunchKernelGGL(embed_st_fwd_kernel, dim3(cic_cdiv(n, 256)), dim3(256), 0, st, p->embed_w, w.idx, w.val,
                           w.x_emb, B, Lp, E);
        CIC_LAUNCH_CHECK();
    }
    if (io->soft && !io->labels) {
        // soft caption rows: x_emb[1..T] += soft @ embed[0:V+1]          (VSEFCModel.py:102-104)
        RUN(gemm_nn_fwd(io->soft, d.V + 1, p->embed_w, E, w.x_emb + (size_t)B * E, E, (Lp - 1) * B, E, d.V + 1, true, stb));
    }
    RUN(gemm_nt(w.x_emb, E, p->w_ih, E, w.gi_all, 3 * J, Lp * B, 3 * J, E, p->b_ih, false, false, stb));
    const bool fused_step = g_gru_fused && J == 1024;       // the flagship width: one launch per step (see the kernel)
    bool seq_kernel = false;
    int seq_rows = 0;
    if (g_gru_fused >= 2 && J == 1024 && !io->device_shared) {
        // the one-launch form needs every workgroup resident at once: one per CU (512 threads holding the weight tile in
        // ~200 VGPRs each fill a CU's register file)
        static DeviceOnce attr_set;
        if (attr_set.first()) {
            CIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_seq_kernel<8, 8, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)GRUF_LDS_BYTES));
            CIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_seq_kernel<8, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)GRUF_LDS_BYTES));
        }
        const int cus = bf ? cic_resident_cus(reinterpret_cast<const void*>(&gru_seq_kernel<8, 8, true>), 512, GRUF_LDS_BYTES)
                           : cic_resident_cus(reinterpret_cast<const void*>(&gru_seq_kernel<8, 8, false>), 512, GRUF_LDS_BYTES);
        seq_rows = (cus / (J / 32)) * 16;            // rows one launch can walk with every workgroup resident
        seq_kernel = seq_rows >= 16;
    }
    if (seq_kernel) {
        for (int row0 = 0; row0 < B; row0 += seq_rows) {      // (B = 128: one launch; B = 256: two row blocks)
            const int row_end = row0 + seq_rows < B ? row0 + seq_rows : B;
            const HandoffGuard hg = handoff_guard(w.sync + 2 * (size_t)cic_cdiv(B, 16) * (Lp + 1), io->status, CIC_STATUS_GRU_FWD);
            const dim3 grid(cic_cdiv(row_end - row0, 16) * (J / 32));
            if (bf)
                hipLaunchKernelGGL((gru_seq_kernel<8, 8, true>), grid, dim3(512), GRUF_LDS_BYTES, st, w.h_all, p->w_hh, p->b_hh, w.gi_all, w.len,
                                   w.gh_all, w.sync, hg, B, J, Lp, row0, row_end);
            else
                hipLaunchKernelGGL((gru_seq_kernel<8, 8, false>), grid, dim3(512), GRUF_LDS_BYTES, st, w.h_all, p->w_hh, p->b_hh, w.gi_all, w.len,
                                   w.gh_all, w.sync, hg, B, J, Lp, row0, row_end);
            CIC_LAUNCH_CHECK();
        }
    } else {
        CIC_HIP(hipMemsetAsync(w.h_all, 0, sizeof(float) * B * J, st));
    }
    for (int t = 0; t < Lp && !seq_kernel; ++t) {
        float* h = w.h_all + (size_t)t * B * J;
        float* gh = w.gh_all + (size_t)t * B * 3 * J;
        if (fused_step) {
            hipLaunchKernelGGL((gru_step_fused_kernel<8, 8>), dim3(cic_cdiv(B, 32) * (J / 16)), dim3(512), 0, st, h, p->w_hh,
                               p->b_hh, w.gi_all + (size_t)t * B * 3 * J, w.len, t, gh, h + (size_t)B * J, B, J, t == 0 ? 1 : 0);
            CIC_LAUNCH_CHECK();
            continue;
        }
        RUN(gemm_nt(h, J, p->w_hh, J, gh, 3 * J, B, 3 * J, J, p->b_hh, false, false, st));
        hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3(cic_cdiv(B * (J / 4), 256)), dim3(256), 0, st,
                           w.gi_all + (size_t)t * B * 3 * J, gh, h, w.len, t, h + (size_t)B * J, B, J);
        CIC_LAUNCH_CHECK();
    }
    const float* cap_raw = w.h_all + (size_t)Lp * B * J;      // 'last': the state after the last valid step (:128-129)
    if (d.pool) {
        hipLaunchKernelGGL(pool_fwd_kernel, dim3(cic_cdiv(B * J, 256)), dim3(256), 0, st, w.h_all, w.len, d.pool, w.pooled,
                           w.pool_arg, B, J, Lp);
        CIC_LAUNCH_CHECK();
        cap_raw = w.pooled;
    }
    hipLaunchKernelGGL(l2norm_fwd2_kernel, dim3(2 * B), dim3(256), 0, st, w.img_lin, w.img_emb, w.nrm_img, !d.no_imgnorm,
                       cap_raw, w.cap_emb, w.nrm_cap, 1, B, J, d.use_abs);
    CIC_LAUNCH_CHECK();
    // contrastive loss                                                   (VSEFCModel.py:167-207)
    RUN(gemm_nt(w.img_emb, J, w.cap_emb, J, w.S, B, B, B, J, nullptr, false, false, st));
    const int sel_s = io->only_one_retrieval != 1, sel_im = io->only_one_retrieval != 2;
    hipLaunchKernelGGL(contrastive_fwd_kernel, dim3(B), dim3(64), 0, st, w.S, B, d.margin, d.max_violation, sel_s, sel_im,
                       io->loss_rows, w.arg_s, w.arg_im, w.sync + 2 * (size_t)cic_cdiv(B, 16) * (Lp + 1) + 1, io->loss_sum);
    CIC_LAUNCH_CHECK();
    if (io->img_emb_out) CIC_HIP(hipMemcpyAsync(io->img_emb_out, w.img_emb, sizeof(float) * B * J, hipMemcpyDeviceToDevice, st));
    if (io->cap_emb_out) CIC_HIP(hipMemcpyAsync(io->cap_emb_out, w.cap_emb, sizeof(float) * B * J, hipMemcpyDeviceToDevice, st));
#undef RUN
    return 0;
}

static int listener_bwd_impl(const cic_listener_dims* dp, const cic_listener_params* p, const cic_listener_io* io,
                             const cic_listener_bwd_io* bio, void* ws, size_t ws_bytes, cic_stream_t s) {
    CIC_REQUIRE(dp && p && io && bio && ws);
    const cic_listener_dims& d = *dp;
    if (int rc = check_ldims(d)) return rc;
    LstWs w = lst_carve(d, ws);
    CIC_REQUIRE(ws_bytes >= w.bytes);
    CIC_REQUIRE(bio->g_rows || bio->g_scalar);
    hipStream_t st = cic_s(s);
    const bool bf = d.compute_dtype == CIC_DTYPE_BF16;      // (see listener_fwd_impl)
    const GemmCtx stb(st, bf ? CIC_PRECISION_BF16 : CIC_PRECISION_F32);
    const int B = d.B, J = d.J, E = d.E, Lp = d.Lp;
    const cic_listener_params* g = bio->grads;   // may be NULL: no parameter gradients wanted
    int rc;
#define RUN(x) if ((rc = (x)) != 0) return rc
    const int sel_s = io->only_one_retrieval != 1, sel_im = io->only_one_retrieval != 2;
    hipLaunchKernelGGL(contrastive_bwd_kernel, dim3(cic_cdiv(B * B, 256)), dim3(256), 0, st, w.S, B, d.margin,
                       d.max_violation, sel_s, sel_im, bio->g_rows, bio->g_scalar, bio->g_scale != 0.f ? bio->g_scale : 1.0f,
                       w.arg_s, w.arg_im, w.dS, w.sync + (size_t)cic_cdiv(B, 16) * (Lp + 1), cic_cdiv(B, 16) * (Lp + 1));
    CIC_LAUNCH_CHECK();
    // S = im cap^T  ->  d_im = dS cap,  d_cap = dS^T im
    RUN(gemm_nn(w.dS, B, w.cap_emb, J, w.d_img, J, B, J, B, false, st));
    RUN(gemm_tn(w.dS, B, w.img_emb, J, w.d_cap, J, B, J, B, false, st));
    // back through both l2norms in ONE launch (image rows only when parameter gradients are wanted), then the image branch's
    // weight gradient; text branch: GRU BPTT
    hipLaunchKernelGGL(l2norm_bwd2_kernel, dim3(g ? 2 * B : B), dim3(256), 0, st, w.img_lin, w.nrm_img, w.d_img, w.d_lin,
                       !d.no_imgnorm, d.pool ? w.pooled : w.h_all + (size_t)Lp * B * J, w.nrm_cap, w.d_cap,
                       d.pool ? w.d_pool : w.dh, 1, B, J, d.use_abs);
    CIC_LAUNCH_CHECK();
    if (g) RUN(gemm_tn(w.d_lin, J, io->fc_feats, d.F, g->img_fc_w, d.F, J, d.F, B, true, st, g->img_fc_b));
    if (d.pool) CIC_HIP(hipMemsetAsync(w.dh, 0, sizeof(float) * B * J, st));   // nothing reaches the final state directly
    float* dh = w.dh;
    float* dh2 = w.dh2;
    bool seq_kernel = false;
    int seq_rows = 0;
    if (g_gru_fused >= 2 && J == 1024 && !io->device_shared) {
        // as in the forward pass: every workgroup resident at once, one per CU
        static DeviceOnce attr_set;
        if (attr_set.first()) {
            CIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_seq_bwd_kernel<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)GRUB_LDS_BYTES));
            CIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_seq_bwd_kernel<8, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)GRUB_LDS_BYTES));
        }
        const int cus = bf ? cic_resident_cus(reinterpret_cast<const void*>(&gru_seq_bwd_kernel<8, true>), 512, GRUB_LDS_BYTES)
                           : cic_resident_cus(reinterpret_cast<const void*>(&gru_seq_bwd_kernel<8, false>), 512, GRUB_LDS_BYTES);
        seq_rows = (cus / (J / 32)) * 16;
        seq_kernel = seq_rows >= 16;
    }
    if (seq_kernel) {
        unsigned* cnt_b = w.sync + (size_t)cic_cdiv(B, 16) * (Lp + 1);
        for (int row0 = 0; row0 < B; row0 += seq_rows) {
            const int row_end = row0 + seq_rows < B ? row0 + seq_rows : B;
            const HandoffGuard hg = handoff_guard(cnt_b + (size_t)cic_cdiv(B, 16) * (Lp + 1), io->status, CIC_STATUS_GRU_BWD);
            const dim3 grid(cic_cdiv(row_end - row0, 16) * (J / 32));
            if (bf)
                hipLaunchKernelGGL((gru_seq_bwd_kernel<8, true>), grid, dim3(512), GRUB_LDS_BYTES, st, w.h_all, p->w_hh, w.gi_all, w.gh_all, w.len,
                                   dh, w.dgi_all, w.dgh_all, cnt_b, hg, B, J, Lp, d.pool, w.d_pool, w.pool_arg, E <= J ? w.dx_emb : nullptr, E,
                                   row0, row_end);
            else
                hipLaunchKernelGGL((gru_seq_bwd_kernel<8, false>), grid, dim3(512), GRUB_LDS_BYTES, st, w.h_all, p->w_hh, w.gi_all, w.gh_all, w.len,
                                   dh, w.dgi_all, w.dgh_all, cnt_b, hg, B, J, Lp, d.pool, w.d_pool, w.pool_arg, E <= J ? w.dx_emb : nullptr, E,
                                   row0, row_end);
            CIC_LAUNCH_CHECK();
        }
    }
    for (int t = Lp - 1; t >= 0 && !seq_kernel; --t) {
        float* dgi = w.dgi_all + (size_t)t * B * 3 * J;
        float* dgh = w.dgh_all + (size_t)t * B * 3 * J;
        hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3(cic_cdiv(B * (J / 4), 256)), dim3(256), 0, st,
                           w.gi_all + (size_t)t * B * 3 * J, w.gh_all + (size_t)t * B * 3 * J,
                           w.h_all + (size_t)t * B * J, w.len, t, dh, dh2, dgi, dgh, B, J, d.pool, w.d_pool, w.pool_arg);
        CIC_LAUNCH_CHECK();
        if (t > 0) RUN(gemm_nn(dgh, 3 * J, p->w_hh, J, dh2, J, B, J, 3 * J, true, st));   // += dgh W_hh
        float* tmp = dh; dh = dh2; dh2 = tmp;
    }
    if (g) {
        RUN(gemm_tn(w.dgh_all, 3 * J, w.h_all, J, g->w_hh, J, 3 * J, J, Lp * B, true, stb, g->b_hh));
        RUN(gemm_tn(w.dgi_all, 3 * J, w.x_emb, E, g->w_ih, E, 3 * J, E, Lp * B, true, stb, g->b_ih));
    }
    if ((g && g->embed_w) || bio->d_onehot) {
        // dx_emb = dgi W_ih            [Lp*B, 3J] x [3J, E]
        RUN(gemm_nn(w.dgi_all, 3 * J, p->w_ih, E, w.dx_emb, E, Lp * B, E, 3 * J, false, stb, true, seq_kernel && E <= J));   // (cleared by the BPTT kernel)
        if (g && g->embed_w) {
            const int64_t n = (int64_t)Lp * B * E;
            hipLaunchKernelGGL(embed_st_bwd_kernel, dim3(cic_cdiv(n, 256)), dim3(256), 0, st, w.dx_emb, w.idx, w.val,
                               w.len, g->embed_w, B, Lp, E);
            CIC_LAUNCH_CHECK();
            if (io->soft && !io->labels)   // dense rows: d embed[0:V+1] += soft^T dx_emb[1..T]
                RUN(gemm_tn(io->soft, d.V + 1, w.dx_emb + (size_t)B * E, E, g->embed_w, E, d.V + 1, E, (Lp - 1) * B, true, stb));
        }
        if (bio->d_onehot) {
            // straight-through path back to the speaker: d one_hot[t,b,0:V+1] = dx_emb[t,b,:] E[0:V+1,:]^T for the
            // generated positions t = 1..T (position 0 is <bos>); time-major [T,B,V+1]   (VSEFCModel.py:104)
            RUN(gemm_nt(w.dx_emb + (size_t)B * E, E, p->embed_w, E, bio->d_onehot, d.V + 1, (Lp - 1) * B, d.V + 1, E,
                        nullptr, false, false, stb));
        }
    }
#undef RUN
    return 0;
}
