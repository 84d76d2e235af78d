// f32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, k-ordered fma chain).
//
// One kernel template, three tile shapes:
//   128x128 (4 waves, 2x2, each wave 64x64 = 2x2 MFMA tiles)   big M*N
//    64x64  (4 waves, 2x2, each wave 32x32)                    mid-size / too few big tiles
//   128x32  (4 waves, 4x1, each wave 32x32)                    skinny per-timestep GEMMs (M = batch)
// K is walked in tiles of 32 through LDS with a register prefetch of the next tile.
//
// LDS images (no bank conflicts, see MI355X_MICROARCH §LDS):
//   K-contiguous operand  -> [row][32+4]: filled by ds_write_b128, fragments read by ONE
//                            ds_read_b128 per 4 MFMAs (lane half h reads k = 8g+4h .. +3)
//   K-strided operand     -> [k][rows+4]: filled by ds_write_b128 along rows, fragments read by
//                            ds_read_b32 (32 consecutive rows per lane half)
// Both operands use the same k permutation inside each group of 8 (MFMA step j multiplies
// k = 8g+j (lanes 0-31) and k = 8g+4+j (lanes 32-63)), so the sum order is fixed and
// reproducible run to run.
#include <type_traits>
#include "cic_common.h"

namespace {

bool aligned16(const void* p);

// dispatch switches: constants in the product build; the development build sets them through cic_debug_gemm_tail_split(flags)
CIC_SWITCH(g_tail_split, 1);   // bit 0: K-sliced tail tiles and K split over workgroups (float atomics) on / off
CIC_SWITCH(g_force_tile, 0);   // bits 8..15: 1 = 128x128, 2 = 64x64 tiles forced
CIC_SWITCH(g_walk, 1);         // bit 16 set: strip walkers off
CIC_SWITCH(g_walk16, 1);       // bit 21 set: 16-wide strip walkers off
CIC_SWITCH(g_ldsb, 1);         // bit 22 set: LDS-staged column walker (K = 512 logit product) off
CIC_SWITCH(g_ldsb2, 2);        // K parts per row tile of the logit walker: 2 or 4 (bits 25..26 of the debug word: 1 -> 4-wave form, 2 -> 4 parts)
CIC_SWITCH(g_rega2, 1);        // bit 24 set: two-strip dX kernel (gemm_rega2_kernel) off
CIC_SWITCH(g_bfx, 1);          // bit 28 set: the LDS-tiled products stay on the f32-input MFMA (no bf16-part kernel)
CIC_SWITCH(g_logit_epi, 1);    // bit 27 set: no fused vocabulary epilogue in the logit walker (cic_gemm_logit_parts() = 0)
CIC_SWITCH(g_split_rows, 1);   // bit 23 set: 129..256-row products are not handed to the register-streaming kernels as two row blocks

constexpr int BK = 32;
constexpr int KCS = BK + 4;  // row stride of a K-contiguous LDS image (floats)

template <int ROWS, bool KC, bool VEC, int THREADS>
struct Tile {
    static constexpr int NV = ROWS * BK / 4 / THREADS;   // float4 per thread per tile
    static constexpr int LDS_FLOATS = KC ? ROWS * KCS : BK * (ROWS + 4);
    static_assert(NV >= 1 && NV * THREADS * 4 == ROWS * BK, "tile/threads mismatch");

    // global -> registers.  Out-of-range elements read as 0.
    __device__ static __forceinline__ void load(f32x4 (&r)[NV], const float* __restrict__ P, int ld,
                                                int row0, int nrows, int k0, int K, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * THREADS;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (KC) {
                const int rr = idx >> 3, q = idx & 7;
                const int row = row0 + rr, k = k0 + 4 * q;
                if (row < nrows) {
                    const float* p = P + (size_t)row * ld + k;
                    if (VEC) {
                        if (k < K) v = *reinterpret_cast<const f32x4*>(p);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (k + j < K) v[j] = p[j];
                    }
                }
            } else {
                constexpr int QR = ROWS / 4;
                const int kk = idx / QR, q = idx % QR;
                const int k = k0 + kk, row = row0 + 4 * q;
                if (k < K) {
                    const float* p = P + (size_t)k * ld + row;
                    if (VEC) {
                        if (row < nrows) v = *reinterpret_cast<const f32x4*>(p);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (row + j < nrows) v[j] = p[j];
                    }
                }
            }
            r[i] = v;
        }
    }

    // registers -> LDS
    __device__ static __forceinline__ void store(const f32x4 (&r)[NV], float* L, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * THREADS;
            if (KC) {
                const int rr = idx >> 3, q = idx & 7;
                *reinterpret_cast<f32x4*>(L + rr * KCS + 4 * q) = r[i];
            } else {
                constexpr int QR = ROWS / 4;
                const int kk = idx / QR, q = idx % QR;
                *reinterpret_cast<f32x4*>(L + kk * (ROWS + 4) + 4 * q) = r[i];
            }
        }
    }

    // LDS -> MFMA fragment for the group g of 8 k's: f[j] = op[row][8g + 4h + j]
    __device__ static __forceinline__ f32x4 frag(const float* L, int row, int g, int h) {
        if (KC) {
            return *reinterpret_cast<const f32x4*>(L + row * KCS + 8 * g + 4 * h);
        } else {
            f32x4 f;
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = L[(8 * g + 4 * h + j) * (ROWS + 4) + row];
            return f;
        }
    }
};

// Tail balancing (`full`, `ks`): with T output tiles on 256 CUs the last partial round of tiles would leave most
// of the chip idle (300 tiles = 1 full round + 44 tiles), so only the first `full` tiles (a multiple of the CU
// count) run their whole K range; each remaining tile is cut into `ks` K-slices handled by `ks` workgroups that
// add their partial tile into C with float atomics (C pre-zeroed by the launcher unless it accumulates).
// full == all tiles: plain data-parallel GEMM with a deterministic, fixed summation order.
template <int BM, int BN, int WM, int WN, bool KCA, bool KCB, bool VEC>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(cic_gemm_args g, int full, int ks) {
    constexpr int THREADS = WM * WN * 64;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    using TA = Tile<BM, KCA, VEC, THREADS>;
    using TB = Tile<BN, KCB, VEC, THREADS>;
    __shared__ __attribute__((aligned(16))) float lds[TA::LDS_FLOATS + TB::LDS_FLOATS];
    float* LA = lds;
    float* LB = lds + TA::LDS_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5, r = lane & 31;

    // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2), so give each XCD a
    // contiguous run of tiles; neighbouring tiles share operand panels.  Speed only.
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    int slice = 0, nslice = 1;
    if (bid >= full) {               // K-sliced tail tile
        const int rem = bid - full;
        slice = rem % ks;
        nslice = ks;
        bid = full + rem / ks;
    }
    const int tiles_n = (g.N + BN - 1) / BN;
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f32x4 ra[TA::NV], rb[TB::NV];
    const bool do_colsum = g.colsum_A != nullptr && n0 == 0;
    float csum = 0.f;
#pragma unroll 1
    for (int pair = 0; pair < 2; ++pair) {
        const float* A = pair ? g.A2 : g.A;
        const float* B = pair ? g.B2 : g.B;
        const int lda = pair ? g.lda2 : g.lda, ldb = pair ? g.ldb2 : g.ldb;
        const int K = pair ? g.K2 : g.K;
        if (K <= 0) continue;
        const int nk_all = (K + BK - 1) / BK;
        const int per = (nk_all + nslice - 1) / nslice;     // K-tiles per slice (a tail slice may be empty)
        const int kt0 = slice * per;
        const int nk = min(nk_all, kt0 + per);
        if (kt0 >= nk) continue;
        TA::load(ra, A, lda, m0, g.M, kt0 * BK, K, tid);
        TB::load(rb, B, ldb, n0, g.N, kt0 * BK, K, tid);
#pragma unroll 1
        for (int kt = kt0; kt < nk; ++kt) {
            __syncthreads();   // everyone is done reading the previous tile
            TA::store(ra, LA, tid);
            TB::store(rb, LB, tid);
            __syncthreads();
            if (!KCA && do_colsum) {
                // bias-gradient by-product (dW = dY^T X: the column sums of dY are db): this K-tile of op(A) is in LDS
                // as [k][rows]; the workgroups of the first tile column add it up (out-of-range k / rows hold zeros).
                // All threads share the work (THREADS / BM k-ranges per column): a single wave walking all 32 k's is a
                // ~1000-cycle dependent chain per K-tile that every other wave then waits for at the barrier.
                constexpr int PARTS = THREADS / BM, KPP = BK / PARTS;
                const int cc = tid % BM, part = tid / BM;
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int kk = 0; kk < KPP; kk += 2) {
                    s0 += LA[(part * KPP + kk) * (BM + 4) + cc];
                    s1 += LA[(part * KPP + kk + 1) * (BM + 4) + cc];
                }
                csum += s0 + s1;
            }
            if (kt + 1 < nk) {   // next tile in flight under the MFMAs
                TA::load(ra, A, lda, m0, g.M, (kt + 1) * BK, K, tid);
                TB::load(rb, B, ldb, n0, g.N, (kt + 1) * BK, K, tid);
            }
#pragma unroll
            for (int grp = 0; grp < BK / 8; ++grp) {
                f32x4 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = TA::frag(LA, wm * (BM / WM) + i * 32 + r, grp, h);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = TB::frag(LB, wn * (BN / WN) + j * 32 + r, grp, h);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
            }
        }
    }

    if (!KCA && do_colsum) {           // block-uniform
        constexpr int PARTS = THREADS / BM;
        __syncthreads();               // the last K-tile's fragment reads are done: LDS is free
        lds[tid] = csum;
        __syncthreads();
        if (tid < BM && m0 + tid < g.M) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < PARTS; ++q) t += lds[q * BM + tid];
            atomicAdd(g.colsum_A + m0 + tid, t);
            if (g.colsum_A2) atomicAdd(g.colsum_A2 + m0 + tid, t);
        }
    }
    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / WN) + j * 32 + r;
            const bool nok = n < g.N;
            const int nc = nok ? n : g.N - 1;
            const float bv = (g.bias && slice == 0) ? g.bias[nc] : 0.f;
            if (nslice > 1) {                    // block-uniform
                // (values finished on the straight-line path, see below: an add of the loaded bias inside the
                // bounds-checked blocks made every atomic wait for the completion of the previous one)
                float va[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[i][j][e] + bv;
                    asm volatile("" : "+v"(v));
                    va[e] = v;
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    // no-return global_atomic_add_f32, executed at the memory side
                    if (m < g.M && nok) atomicAdd(g.C + (size_t)m * g.ldc + n, va[e]);
                }
                continue;
            }
            // C += ...: the 16 old values of this 32x32 block are fetched in ONE batch of unconditional loads (rows and
            // columns beyond the matrix re-read its last ones); a load under the bounds check of each element compiled
            // to load + full wait + store per element, 64 serial memory round trips per thread in a 128x128 tile
            float cold[16];
            if (g.accumulate) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    cold[e] = g.C[(size_t)(m < g.M ? m : g.M - 1) * g.ldc + nc];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) cold[e] = 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
            // the sums are finished on the straight-line path (the empty asm keeps them there): computed inside the
            // bounds-checked store blocks, every block would wait for ALL earlier memory operations, stores included
            float vv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[i][j][e] + bv + cold[e];
                if (g.relu) v = fmaxf(v, 0.f);
                asm volatile("" : "+v"(v));
                vv[e] = v;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < g.M && nok) g.C[(size_t)m * g.ldc + n] = vv[e];
            }
        }
}


// ---------------------------------------------------------------------------------------------
// The same LDS-tiled product on the bf16 matrix cores, at f32 accuracy: every f32 operand element is cut into NP bf16
// parts while its tile is staged into LDS (x = x1 + x2 + x3, each part the bf16 rounding of what the previous ones left:
// 3 x 8 mantissa bits = the 24 of an f32), and a product is the sum of the part products that matter:
//     a b  ~=  a1 b1 + (a1 b2 + a2 b1) + (a2 b2 + a1 b3 + a3 b1)          (the dropped terms are <= 2^-24 |a b|)
// Part products are exact in f32 (8 x 8 mantissa bits) and are summed in f32 accumulators, small terms first.
// v_mfma_f32_32x32x16_bf16 does 16 k per 32 cycles where the f32-input v_mfma_f32_32x32x2_f32 does 2 k per 64: six part
// products per k-step are 2.67x the f32 MFMA rate for the same f32 inputs and f32 result (error vs an f64 product
// ~1e-7 relative, the level of the f32 MFMA's own fma chain; tests/test_gpu_kernels.py).  NP = 1 is the plain bf16
// product (operands rounded to bf16 once): the reduced-precision variant of BASELINE configs[1].
//   LDS images, per part:  K-contiguous operand  [rows][BK] bf16, 80-byte rows  -> one ds_read_b128 per fragment;
//                          K-strided operand     [BK][rows] bf16 (+64 B / row)  -> two ds_read_b64_tr_b16 per fragment
//                          (the hardware transpose read hands every lane 4 consecutive k of its row; no transposing store).
// Same workgroup / tail / epilogue scheme as gemm_kernel.
// ---------------------------------------------------------------------------------------------
#ifdef CIC_DEVTOOLS
__device__ unsigned long long* g_bfx_stamps = nullptr;   // development build: phase times of gemm_bfx_kernel's K loop
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int NP>
__device__ __forceinline__ void split_bf16(const f32x4 v, bf16x4 (&out)[NP]) {
    f32x4 r = v;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const bf16x4 h = __builtin_convertvector(r, bf16x4);        // round to nearest even (v_cvt_pk_bf16_f32)
        out[p] = h;
        if (p + 1 < NP) r = r - __builtin_convertvector(h, f32x4);  // exact: the part is a prefix of r's mantissa
    }
}

template <int ROWS, bool KC, int THREADS, int NP>
struct TileBf {
    static constexpr int NV = ROWS * BK / 4 / THREADS;                 // float4 per thread per tile (as Tile<>)
    static constexpr int KSTRIDE = BK + 8;                             // bf16 per row of a K-contiguous image: 80 bytes
    static constexpr int RSTRIDE = ROWS + 32;                          // bf16 per k-row of a K-strided image: +64 bytes
    static constexpr int PART = KC ? ROWS * KSTRIDE : BK * RSTRIDE;    // bf16 per part
    static constexpr int LDS_BF16 = NP * PART;
    static_assert(NV >= 1 && NV * THREADS * 4 == ROWS * BK, "tile/threads mismatch");

    // registers (the f32 tile as Tile<ROWS, KC, true, THREADS>::load left it) -> bf16 parts in LDS
    __device__ static __forceinline__ void store(const f32x4 (&r)[NV], __bf16* L, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * THREADS;
            bf16x4 parts[NP];
            split_bf16<NP>(r[i], parts);
            int off;
            if (KC) { const int rr = idx >> 3, q = idx & 7; off = rr * KSTRIDE + 4 * q; }          // 4 k of one row
            else { constexpr int QR = ROWS / 4; const int kk = idx / QR, q = idx % QR; off = kk * RSTRIDE + 4 * q; }   // 4 rows of one k
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<bf16x4*>(L + p * PART + off) = parts[p];
        }
    }

    // MFMA fragment of part p for the 32 rows starting at row0, k-step s (16 k): lane (r = lane & 31, h = lane >> 5)
    // gets op[row0 + r][16 s + 8 h + 0..7]
    __device__ static __forceinline__ bf16x8 frag(const __bf16* L, int p, int row0, int s, int lane) {
        if (KC) {
            const int r = lane & 31, h = lane >> 5;
            return *reinterpret_cast<const bf16x8*>(L + p * PART + (row0 + r) * KSTRIDE + 16 * s + 8 * h);
        } else {
            // 16-lane group g: rows row0 + 16 (g & 1) + 0..15, k 16 s + 8 (g >> 1) + 0..7 as two 4-k blocks; lane 4q + pp of
            // the group addresses k-row q, rows 4 pp .. 4 pp + 3 of the block
            const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
            const __bf16* base = L + p * PART + (16 * s + 8 * (g >> 1) + q) * RSTRIDE + row0 + 16 * (g & 1) + 4 * pp;
            typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + 4 * RSTRIDE));
            union { struct { s16x4 a, b; } s; bf16x8 v; } u;
            u.s.a = lo; u.s.b = hi;
            return u.v;
        }
    }
};

template <int BM, int BN, int WM, int WN, bool KCA, bool KCB, int NP>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bfx_kernel(cic_gemm_args g, int full, int ks) {
    constexpr int THREADS = WM * WN * 64;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    using LA_ = Tile<BM, KCA, true, THREADS>;            // global -> registers (f32)
    using LB_ = Tile<BN, KCB, true, THREADS>;
    using TA = TileBf<BM, KCA, THREADS, NP>;
    using TB = TileBf<BN, KCB, THREADS, NP>;
    __shared__ __attribute__((aligned(16))) __bf16 lds[TA::LDS_BF16 + TB::LDS_BF16];
    __bf16* LA = lds;
    __bf16* LB = lds + TA::LDS_BF16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5, r = lane & 31;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);      // XCD-aware tile order (as gemm_kernel)
    int slice = 0, nslice = 1;
    if (bid >= full) {               // K-sliced tail tile
        const int rem = bid - full;
        slice = rem % ks;
        nslice = ks;
        bid = full + rem / ks;
    }
    const int tiles_n = (g.N + BN - 1) / BN;
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f32x4 ra[LA_::NV], rb[LB_::NV];
    const bool do_colsum = !KCA && g.colsum_A != nullptr && n0 == 0;
    f32x4 cs4 = {0.f, 0.f, 0.f, 0.f};                   // column sums of op(A) = bias gradient: rows 4q .. 4q+3 of this thread
    const int K = g.K;                                  // (no second operand pair on this path)
    const int nk_all = (K + BK - 1) / BK;
    const int per = (nk_all + nslice - 1) / nslice;
    const int kt0 = slice * per;
    const int nk = min(nk_all, kt0 + per);
    // development build: cumulative phase times of the K loop per wave (cic_debug_set_bfx_stamps): [workgroup][wave][8] =
    // waiting at the first barrier, store phase (incl. the wait for the tile's global loads), second barrier, load issue, MFMAs,
    // K tiles, start, end  (100 MHz ticks)
    unsigned long long* stamps = CIC_STAMP_BUF(g_bfx_stamps);
    unsigned long long tacc[5] = {0, 0, 0, 0, 0}, tprev = 0, tstart = 0;
    if (stamps) tstart = tprev = __builtin_amdgcn_s_memrealtime();
#define BFX_T(i) if (stamps) { const unsigned long long tn = __builtin_amdgcn_s_memrealtime(); tacc[i] += tn - tprev; tprev = tn; }
    auto load_tiles = [&](int kt) {
        LA_::load(ra, g.A, g.lda, m0, g.M, kt * BK, K, tid);
        LB_::load(rb, g.B, g.ldb, n0, g.N, kt * BK, K, tid);
    };
    if (kt0 < nk) {
        load_tiles(kt0);
#pragma unroll 1
        for (int kt = kt0; kt < nk; ++kt) {
            __syncthreads();   // everyone is done reading the previous tile
            BFX_T(0)
            TA::store(ra, LA, tid);
            TB::store(rb, LB, tid);
            if (do_colsum) {
#pragma unroll
                for (int i = 0; i < LA_::NV; ++i) cs4 += ra[i];
            }
            BFX_T(1)
            __syncthreads();
            BFX_T(2)
            if (kt + 1 < nk) load_tiles(kt + 1);   // next tile in flight under the MFMAs
            BFX_T(3)
#pragma unroll
            for (int s = 0; s < BK / 16; ++s) {
                bf16x8 af[TM][NP], bf[TN][NP];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int p = 0; p < NP; ++p) af[i][p] = TA::frag(LA, p, wm * (BM / WM) + i * 32, s, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int p = 0; p < NP; ++p) bf[j][p] = TB::frag(LB, p, wn * (BN / WN) + j * 32, s, lane);
                // part products, smallest first: (pa, pb) with pa + pb = 2, then 1, then 0
#pragma unroll
                for (int order = NP - 1; order >= 0; --order)
#pragma unroll
                    for (int pa = 0; pa <= order; ++pa) {
                        const int pb = order - pa;
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][pa], bf[j][pb], acc[i][j], 0, 0, 0);
                    }
            }
            if (stamps) { asm volatile("" :: "v"(acc[0][0][0])); BFX_T(4) }
        }
    }
#undef BFX_T
    if (stamps && lane == 0) {
        unsigned long long* o = stamps + ((size_t)blockIdx.x * (THREADS / 64) + wave) * 8;
#pragma unroll
        for (int i = 0; i < 5; ++i) o[i] = tacc[i];
        o[5] = (unsigned long long)(nk - kt0); o[6] = tstart; o[7] = __builtin_amdgcn_s_memrealtime();
    }
    if (do_colsum) {                    // block-uniform
        constexpr int QR = BM / 4, GROUPS = THREADS / QR;
        __syncthreads();                // the last K-tile's fragment reads are done: LDS is free
        float* lf = reinterpret_cast<float*>(lds);
        static_assert(sizeof(float) * GROUPS * BM <= sizeof(__bf16) * (TA::LDS_BF16 + TB::LDS_BF16), "colsum scratch");
        *reinterpret_cast<f32x4*>(lf + (tid / QR) * BM + 4 * (tid % QR)) = cs4;
        __syncthreads();
        if (tid < BM && m0 + tid < g.M) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < GROUPS; ++q) t += lf[q * BM + tid];
            atomicAdd(g.colsum_A + m0 + tid, t);
            if (g.colsum_A2) atomicAdd(g.colsum_A2 + m0 + tid, t);
        }
    }
    // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)   (as gemm_kernel)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / WN) + j * 32 + r;
            const bool nok = n < g.N;
            const int nc = nok ? n : g.N - 1;
            const float bv = (g.bias && slice == 0) ? g.bias[nc] : 0.f;
            if (nslice > 1) {                    // block-uniform
                float va[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[i][j][e] + bv;
                    asm volatile("" : "+v"(v));
                    va[e] = v;
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (m < g.M && nok) atomicAdd(g.C + (size_t)m * g.ldc + n, va[e]);
                }
                continue;
            }
            float cold[16];
            if (g.accumulate) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    cold[e] = g.C[(size_t)(m < g.M ? m : g.M - 1) * g.ldc + nc];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) cold[e] = 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
            float vv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[i][j][e] + bv + cold[e];
                if (g.relu) v = fmaxf(v, 0.f);
                asm volatile("" : "+v"(v));
                vv[e] = v;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < g.M && nok) g.C[(size_t)m * g.ldc + n] = vv[e];
            }
        }
}


// ---------------------------------------------------------------------------------------------
// Skinny GEMM for the per-timestep products (M = batch <= 128): one 32-column output strip per
// workgroup, the K range split over KS wave groups INSIDE the workgroup (each group owns its own
// LDS tiles and keeps its own global loads in flight), partial accumulators combined through LDS.
// A per-step GEMM is latency-bound (a single wave's MFMA chain over K and one L2 round trip per
// K-tile), so the lever is more independent chains and more loads in flight per CU, not tile reuse.
//   waves = WM * KS;  wave -> (ks = wave / WM, wm = wave % WM);  BM = 32*WM, BN = 32, BK = 32
// K-tiles are dealt round-robin to the groups (tile j -> group j % KS), so the summation order is
// fixed: deterministic run to run.
// ---------------------------------------------------------------------------------------------
template <int WM, int KS, bool KCA, bool KCB, bool VEC>
__global__ __launch_bounds__(WM* KS * 64) void gemm_skinny_kernel(cic_gemm_args g) {
    constexpr int BM = 32 * WM, BN = 32;
    constexpr int GT = WM * 64;   // threads per K-group
    using TA = Tile<BM, KCA, VEC, GT>;
    using TB = Tile<BN, KCB, VEC, GT>;
    constexpr int GROUP_FLOATS = TA::LDS_FLOATS + TB::LDS_FLOATS;
    constexpr int RED_FLOATS = KS * WM * 16 * 64;
    constexpr int LDS_FLOATS = GROUP_FLOATS * KS > RED_FLOATS ? GROUP_FLOATS * KS : RED_FLOATS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ks = wave / WM, wm = wave % WM;
    const int gtid = tid - ks * GT;
    const int h = lane >> 5, r = lane & 31;
    float* LA = lds + ks * GROUP_FLOATS;
    float* LB = LA + TA::LDS_FLOATS;

    const int tiles_n = (g.N + BN - 1) / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;

    const int nk1 = (g.K + BK - 1) / BK, nk2 = g.K2 > 0 ? (g.K2 + BK - 1) / BK : 0;
    const int nk = nk1 + nk2;
    const int iters = (nk + KS - 1) / KS;

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    f32x4 ra[TA::NV], rb[TB::NV];

    auto load_tile = [&](int j) {
        if (j < nk1) {
            TA::load(ra, g.A, g.lda, m0, g.M, j * BK, g.K, gtid);
            TB::load(rb, g.B, g.ldb, n0, g.N, j * BK, g.K, gtid);
        } else if (j < nk) {
            TA::load(ra, g.A2, g.lda2, m0, g.M, (j - nk1) * BK, g.K2, gtid);
            TB::load(rb, g.B2, g.ldb2, n0, g.N, (j - nk1) * BK, g.K2, gtid);
        } else {
#pragma unroll
            for (int i = 0; i < TA::NV; ++i) ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < TB::NV; ++i) rb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    load_tile(ks);
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        TA::store(ra, LA, gtid);
        TB::store(rb, LB, gtid);
        __syncthreads();
        if (it + 1 < iters) load_tile((it + 1) * KS + ks);
        if (it * KS + ks < nk) {
#pragma unroll
            for (int grp = 0; grp < BK / 8; ++grp) {
                const f32x4 af = TA::frag(LA, wm * 32 + r, grp, h);
                const f32x4 bf = TB::frag(LB, r, grp, h);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc, 0, 0, 0);
            }
        }
    }
    // combine the KS partial tiles through LDS: red[ks][wm][reg][lane]
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) lds[((ks * WM + wm) * 16 + e) * 64 + lane] = acc[e];
    __syncthreads();
    // wave (ks, wm) finishes registers e = ks, ks+KS, ... of row tile wm
    const int n = n0 + r;
    if (n < g.N) {
        const float bv = g.bias ? g.bias[n] : 0.f;
        for (int e = ks; e < 16; e += KS) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < KS; ++q) v += lds[((q * WM + wm) * 16 + e) * 64 + lane];
            const int m = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (m < g.M) {
                float* c = g.C + (size_t)m * g.ldc + n;
                v += bv;
                if (g.accumulate) v += *c;
                if (g.relu) v = fmaxf(v, 0.f);
                *c = v;
            }
        }
    }
}

template <int WM, int KS>
int launch_skinny(const cic_gemm_args& g, bool vec, hipStream_t st) {
    constexpr int BM = 32 * WM;
    const int grid = cic_cdiv(g.M, BM) * cic_cdiv(g.N, 32);
    dim3 blk(WM * KS * 64);
    const int code = (g.a_kc ? 4 : 0) | (g.b_kc ? 2 : 0) | (vec ? 1 : 0);
#define CIC_SK_GO(KA, KB, V)                                                                                  \
    do {                                                                                                      \
        using TA = Tile<BM, KA, V, WM * 64>;                                                                  \
        using TB = Tile<32, KB, V, WM * 64>;                                                                  \
        constexpr int GF = (TA::LDS_FLOATS + TB::LDS_FLOATS) * KS, RF = KS * WM * 16 * 64;                    \
        constexpr size_t shm = sizeof(float) * (GF > RF ? GF : RF);                                           \
        static DeviceOnce attr_set;                                                                         \
        if (attr_set.first()) {                                                                                      \
            CIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_kernel<WM, KS, KA, KB, V>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));               \
        }                                                                                                     \
        hipLaunchKernelGGL((gemm_skinny_kernel<WM, KS, KA, KB, V>), dim3(grid), blk, shm, st, g);             \
    } while (0)
    switch (code) {
        case 7: CIC_SK_GO(true, true, true); break;
        case 6: CIC_SK_GO(true, true, false); break;
        case 5: CIC_SK_GO(true, false, true); break;
        case 4: CIC_SK_GO(true, false, false); break;
        case 3: CIC_SK_GO(false, true, true); break;
        case 2: CIC_SK_GO(false, true, false); break;
        case 1: CIC_SK_GO(false, false, true); break;
        default: CIC_SK_GO(false, false, false); break;
    }
#undef CIC_SK_GO
    CIC_LAUNCH_CHECK();
    return 0;
}


// ---------------------------------------------------------------------------------------------
// Register-streaming GEMM for the per-timestep products (M = batch, K-contiguous activations):
// no LDS staging at all.  A wave (wm, ks) loads its MFMA A fragments for 32 rows x one K slice
// straight from global memory into registers, the matching B fragments of ONE 32-column strip the
// same way, runs the MFMA chain, and the KS partial 32x32 tiles of a strip are summed through LDS
// (wave ks finishes accumulator register ks of the tile: 16 registers <-> 16 waves).
// Why: at M = 128 a product is a few hundred thousand MFMAs; what matters is that every SIMD of the
// chip gets an equal, short chain (K slice of 32..256) and that each wave issues ALL its loads up
// front (one memory round trip per wave), not tile reuse through LDS.
//   fragments: lane (r = lane&31, h = lane>>5) holds op[row r][8g + 4h .. +3] for group g of 8 k's,
//   i.e. one float4 per group when the operand is K-contiguous (weights W[N,K], activations x[M,K]);
//   a K-strided B (dX = dY W) is read as 4 row-coalesced dwords per group.
// ---------------------------------------------------------------------------------------------
#ifdef CIC_DEVTOOLS
__device__ unsigned long long* g_stamp_buf = nullptr;
#endif
__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};   // A operand of k's beyond K   // diagnostics: per-workgroup phase stamps (cic_debug_set_stamps)
// cic_gemm_args.live: the decode loop this launch belongs to has nothing left to do at its step (see cic.h)
__device__ __forceinline__ bool gemm_skipped(const cic_gemm_args& g) {
    if (!g.live) return false;
    const int need = g.live_min > 0 ? g.live_min : 1;
    return *g.live < need && (!g.live_b || *g.live_b < need);
}

template <int KS, bool KCB>   // waves = KS (one 32-row strip per workgroup)
__global__ __launch_bounds__(KS * 64) void gemm_rega_kernel(cic_gemm_args g, int gps) {
    unsigned long long* stamps = CIC_STAMP_BUF(g_stamp_buf);
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0;
    if (stamps) st0 = __builtin_amdgcn_s_memrealtime();
    // gps = groups (of 8 k) per K slice, a multiple of CH; processed in chunks of CH groups with a
    // register double buffer (next chunk's loads in flight under the current chunk's MFMAs)
    constexpr int CH = 4;
    __shared__ float red[KS * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, ks = tid >> 6;
    const int h = lane >> 5, r = lane & 31;
    const int tiles_n = (g.N + 31) / 32;
    int m0 = (blockIdx.x / tiles_n) * 32;
    const int n0 = (blockIdx.x % tiles_n) * 32;
    // row blocks: strips at or beyond rows_blk read/write the second decode's buffers (strips never straddle)
    const bool blk2 = g.rows_blk > 0 && m0 >= g.rows_blk;
    const float* __restrict__ gA = blk2 ? g.A_b : g.A;
    const float* __restrict__ gA2 = blk2 ? g.A2_b : g.A2;
    float* __restrict__ gC = blk2 ? g.C_b : g.C;
    const int Mloc = g.rows_blk > 0 ? (blk2 ? g.M - g.rows_blk : min(g.M, g.rows_blk)) : g.M;
    if (blk2) m0 -= g.rows_blk;
    const int m = m0 + r, n = n0 + r;
    const int K1 = g.K, Kt = g.K + g.K2;
    const bool mok = m < Mloc, nok = n < g.N;

    // Loads are UNCONDITIONAL and nothing selects on a loaded value: rows / columns beyond M / N re-read the last
    // valid one (their outputs are never stored), and a k beyond K reads A from a block of zeros (B from a clamped,
    // valid k), so the padding contributes 0.  A select or a branch on the loaded data makes hipcc wait vmcnt(0)
    // right behind each load, which serialises every round trip of the prefetch (measured: 2x on these launches).
    const int mc = mok ? m : Mloc - 1, nc = nok ? n : g.N - 1;
    const float* zeros = g_zero16;
    auto load_chunk = [&](f32x4 (&af)[CH], f32x4 (&bf)[CH], int c) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            // first of this lane's 4 k's (K1 % 8 == 0); blockIdx.y = K part of a product split over workgroups
            const int k = 8 * ((blockIdx.y * KS + ks) * gps + c * CH + i) + 4 * h;
            const bool kok = k < Kt;
            const bool second = kok && k >= K1;
            const float* A = second ? gA2 : gA;
            const float* B = second ? g.B2 : g.B;
            const int lda = second ? g.lda2 : g.lda, ldb = second ? g.ldb2 : g.ldb;
            const int kp = second ? g.K2 : K1;
            int kk = second ? k - K1 : k;
            kk = kk < kp - 4 ? kk : kp - 4;
            const float* pa = A + (size_t)mc * lda + kk;
            pa = kok ? pa : zeros;
            af[i] = *reinterpret_cast<const f32x4*>(pa);
            if (KCB) {
                bf[i] = *reinterpret_cast<const f32x4*>(B + (size_t)nc * ldb + kk);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[i][j] = B[(size_t)(kk + j) * ldb + nc];
            }
        }
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    auto mma_chunk = [&](const f32x4 (&af)[CH], const f32x4 (&bf)[CH]) {
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[i][s], acc, 0, 0, 0);
    };
    const int nch = gps / CH;
    f32x4 a0[CH], b0[CH], a1[CH], b1[CH];
    load_chunk(a0, b0, 0);
#pragma unroll 1
    for (int c = 0; c < nch; c += 2) {
        if (c + 1 < nch) load_chunk(a1, b1, c + 1);
        mma_chunk(a0, b0);
        if (stamps && c == 0) { asm volatile("" :: "v"(acc[0])); st1 = __builtin_amdgcn_s_memrealtime(); }
        if (c + 2 < nch) load_chunk(a0, b0, c + 2);
        if (c + 1 < nch) mma_chunk(a1, b1);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(ks * 16 + e) * 64 + lane] = acc[e];
    if (stamps) st2 = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    if (stamps) st3 = __builtin_amdgcn_s_memrealtime();
    // wave ks sums accumulator register e = ks, ks + KS, ... over the KS partial tiles (fixed order)
    for (int e = ks; e < 16; e += KS) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < KS; ++q) v += red[(q * 16 + e) * 64 + lane];
        const int mm = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (mm < Mloc && nok) {
            float* c = gC + (size_t)mm * g.ldc + n;
            if (gridDim.y > 1) {
                // K split over workgroups (gradient products only): partial tiles meet in C through float atomics;
                // C holds the accumulation target, or zeros written by the launcher / the producing kernel
                if (g.bias && blockIdx.y == 0) v += g.bias[n];
                atomicAdd(c, v);
            } else {
                if (g.bias) v += g.bias[n];
                if (g.accumulate) v += *c;
                if (g.relu) v = fmaxf(v, 0.f);
                *c = v;
            }
        }
    }
    if (stamps && lane == 0) {   // [block][wave][5]: start, first chunk done, MFMAs done, barrier passed, end (100 MHz ticks)
        unsigned long long* o = stamps + ((size_t)blockIdx.x * KS + ks) * 6;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = __builtin_amdgcn_s_memrealtime();
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[5] = xcc;
    }
}


// Two-strip variant for the BPTT dX products (dX = dY W, W stored [K, N]: the K-strided B operand is the expensive
// one, four row-coalesced dword loads per group): a workgroup owns 64 rows x 32 columns, every B fragment feeds the
// MFMA chains of BOTH 32-row strips.  These launches are bound by the bytes a CU has to pull in (~15 B/clk per CU with
// every CU loading): per unit of output a workgroup reads (64 + 32) k-columns instead of 2 x (32 + 32), 25 % fewer
// bytes and 40 % fewer load instructions, at twice the K split over workgroups (float atomics, gradient products
// only).  K slices of CH groups, register double buffer as in gemm_rega_kernel; the 16 x 32 partial accumulator
// registers of a tile meet in 128 KB of LDS and every wave finishes two of them.
template <int KS, int CH>
__global__ __launch_bounds__(KS * 64) void gemm_rega2_kernel(cic_gemm_args g, int gps) {
    static_assert(KS == 16, "32 accumulator registers dealt two per wave");
    extern __shared__ __attribute__((aligned(16))) float red2[];   // KS * 32 * 64 floats
    if (gemm_skipped(g)) return;                            // BPTT step at or beyond the decode's length: C stays cleared
    const int tid = threadIdx.x, lane = tid & 63, ks = tid >> 6;
    const int h = lane >> 5, r = lane & 31;
    const int tiles_n = (g.N + 31) / 32;
    int m0 = (blockIdx.x / tiles_n) * 64;
    const int n0 = (blockIdx.x % tiles_n) * 32;
    const bool blk2 = g.rows_blk > 0 && m0 >= g.rows_blk;   // row blocks (rows_blk % 64 == 0): see gemm_rega_kernel
    const float* __restrict__ gA = blk2 ? g.A_b : g.A;
    const float* __restrict__ gA2 = blk2 ? g.A2_b : g.A2;
    float* __restrict__ gC = blk2 ? g.C_b : g.C;
    const int Mloc = g.rows_blk > 0 ? (blk2 ? g.M - g.rows_blk : min(g.M, g.rows_blk)) : g.M;
    if (blk2) m0 -= g.rows_blk;
    const int n = n0 + r;
    const int K1 = g.K, Kt = g.K + g.K2;
    const bool nok = n < g.N;
    const int nc = nok ? n : g.N - 1;
    int mc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) mc[t] = (m0 + 32 * t + r) < Mloc ? (m0 + 32 * t + r) : Mloc - 1;
    const float* zeros = g_zero16;
    // unconditional loads, no select on loaded values (see gemm_rega_kernel)
    auto load_chunk = [&](f32x4 (&af)[2][CH], f32x4 (&bf)[CH], int c) {
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int k = 8 * ((blockIdx.y * KS + ks) * gps + c * CH + i) + 4 * h;
            const bool kok = k < Kt;
            const bool second = kok && k >= K1;
            const float* A = second ? gA2 : gA;
            const float* B = second ? g.B2 : g.B;
            const int lda = second ? g.lda2 : g.lda, ldb = second ? g.ldb2 : g.ldb;
            const int kp = second ? g.K2 : K1;
            int kk = second ? k - K1 : k;
            kk = kk < kp - 4 ? kk : kp - 4;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float* pa = A + (size_t)mc[t] * lda + kk;
                pa = kok ? pa : zeros;
                af[t][i] = *reinterpret_cast<const f32x4*>(pa);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[i][j] = B[(size_t)(kk + j) * ldb + nc];
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    auto mma_chunk = [&](const f32x4 (&af)[2][CH], const f32x4 (&bf)[CH]) {
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][i][s], bf[i][s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][i][s], bf[i][s], acc[1], 0, 0, 0);
            }
    };
    const int nch = gps / CH;
    f32x4 a0[2][CH], b0[CH], a1[2][CH], b1[CH];
    load_chunk(a0, b0, 0);
#pragma unroll 1
    for (int c = 0; c < nch; c += 2) {
        if (c + 1 < nch) load_chunk(a1, b1, c + 1);
        mma_chunk(a0, b0);
        if (c + 2 < nch) load_chunk(a0, b0, c + 2);
        if (c + 1 < nch) mma_chunk(a1, b1);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) red2[(ks * 32 + 16 * t + e) * 64 + lane] = acc[t][e];
    __syncthreads();
    // wave ks sums registers ks and ks + 16 of the 32 (strip 0 / strip 1, accumulator register ks) over the KS partial
    // tiles in a fixed order; the K parts of different workgroups then meet in C through float atomics
    float bias_v = 0.f;
    if (g.bias && blockIdx.y == 0) bias_v = g.bias[nc];
    float v2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < KS; ++q) v += red2[(q * 32 + 16 * t + ks) * 64 + lane];
        v += bias_v;
        asm volatile("" : "+v"(v));          // finished outside the bounds-checked block (see the gemm_kernel epilogue)
        v2[t] = v;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int mm = m0 + 32 * t + (ks & 3) + 8 * (ks >> 2) + 4 * h;
        if (mm < Mloc && nok) atomicAdd(gC + (size_t)mm * g.ldc + n, v2[t]);
    }
}


// Persistent-strip variant for many column strips (the logit product): a workgroup owns ONE 32-row
// strip of A — each wave keeps the MFMA A fragments of its K slice in registers for the whole
// launch — and walks column tiles t = first, first+step, ...  The B fragments of the next tile are
// in flight while the current tile's MFMA chain and cross-wave sum run, and the partial-tile buffer
// in LDS is double buffered (ONE barrier per tile).  Measured for [128 x 9488 x 512]: 23 us vs 33 us
// for the one-tile-per-workgroup kernel above (MFMA time 8 us; the 16-way cross-wave sum through LDS
// and the lock-step phases account for the rest).
template <int GPS, int KS, bool KCB>   // K slice = 8*GPS per wave
__global__ __launch_bounds__(KS * 64) void gemm_rega_loop_kernel(cic_gemm_args g, int strips) {
    static_assert(KS == 16, "one accumulator register per wave in the cross-wave sum");
    __shared__ float red[2 * KS * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, ks = tid >> 6;
    const int h = lane >> 5, r = lane & 31;
    const int tiles_n = (g.N + 31) / 32;
    // blockIdx -> (strip, walker): workgroups b and b+8 share an XCD (and its L2).  All `strips` workgroups that walk
    // the SAME column tiles must sit on one XCD, so that a weight tile leaves the Infinity Cache / HBM once per
    // launch and its other readers hit the L2: walker = blockIdx % step with step a multiple of 8.
    const int step = gridDim.x / strips;
    const int first = blockIdx.x % step, strip = blockIdx.x / step;
    int m0 = strip * 32;
    const bool blk2 = g.rows_blk > 0 && m0 >= g.rows_blk;   // row blocks: see gemm_rega_kernel
    const float* __restrict__ gA = blk2 ? g.A_b : g.A;
    const float* __restrict__ gA2 = blk2 ? g.A2_b : g.A2;
    float* __restrict__ gC = blk2 ? g.C_b : g.C;
    const int Mloc = g.rows_blk > 0 ? (blk2 ? g.M - g.rows_blk : min(g.M, g.rows_blk)) : g.M;
    if (blk2) m0 -= g.rows_blk;
    const int m = m0 + r;
    const int K1 = g.K, Kt = g.K + g.K2;
    const bool mok = m < Mloc;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const int mc = mok ? m : Mloc - 1;
    f32x4 af[GPS];
#pragma unroll
    for (int i = 0; i < GPS; ++i) {
        const int k = 8 * (ks * GPS + i) + 4 * h;
        const bool kok = k < Kt;
        const bool second = kok && k >= K1;
        const int kp = second ? g.K2 : K1;
        int kk = second ? k - K1 : k;
        kk = kk < kp - 4 ? kk : kp - 4;
        const f32x4 a = *reinterpret_cast<const f32x4*>((second ? gA2 : gA) + (size_t)mc * (second ? g.lda2 : g.lda) + kk);
        af[i] = (kok && mok) ? a : z4;
    }
    auto load_b = [&](f32x4 (&bf)[GPS], int t) {
        const int tt = t < tiles_n ? t : tiles_n - 1;
        const int n = tt * 32 + r;
        const bool nok = t < tiles_n && n < g.N;
        const int nc = n < g.N ? n : g.N - 1;
#pragma unroll
        for (int i = 0; i < GPS; ++i) {
            const int k = 8 * (ks * GPS + i) + 4 * h;
            const bool kok = k < Kt;
            const bool second = kok && k >= K1;
            const float* B = second ? g.B2 : g.B;
            const int ldb = second ? g.ldb2 : g.ldb;
            const int kp = second ? g.K2 : K1;
            int kk = second ? k - K1 : k;
            kk = kk < kp - 4 ? kk : kp - 4;
            f32x4 b;
            if (KCB) {
                b = *reinterpret_cast<const f32x4*>(B + (size_t)nc * ldb + kk);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = B[(size_t)(kk + j) * ldb + nc];
            }
            bf[i] = (kok && nok) ? b : z4;
        }
    };
    f32x4 bcur[GPS], bnext[GPS];
    load_b(bcur, first);
    int buf = 0;
    const int e_own = ks;                              // wave ks finishes accumulator register ks
    const int mm = m0 + (e_own & 3) + 8 * (e_own >> 2) + 4 * h;
    const int mcl = mm < Mloc ? mm : Mloc - 1;
#pragma unroll 1
    for (int t = first; t < tiles_n; t += step) {
        load_b(bnext, t + step);                       // next tile's fragments in flight
        const int n = t * 32 + r;
        const int ncl = n < g.N ? n : g.N - 1;
        float bias_v = 0.f, cold = 0.f;                // epilogue operands fetched before the barrier
        if (g.bias) bias_v = g.bias[ncl];
        if (g.accumulate) cold = gC[(size_t)mcl * g.ldc + ncl];
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int i = 0; i < GPS; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bcur[i][s], acc, 0, 0, 0);
        float* rb = red + buf * (KS * 16 * 64);
#pragma unroll
        for (int e = 0; e < 16; ++e) rb[(ks * 16 + e) * 64 + lane] = acc[e];
        __syncthreads();
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < KS; ++q) v += rb[(q * 16 + e_own) * 64 + lane];
        v += bias_v + cold;
        if (g.relu) v = fmaxf(v, 0.f);
        if (mm < Mloc && n < g.N) gC[(size_t)mm * g.ldc + n] = v;
        buf ^= 1;
#pragma unroll
        for (int i = 0; i < GPS; ++i) bcur[i] = bnext[i];
    }
}


// Strip walker: the persistent-strip idea above with FEWER, LONGER K slices.  A workgroup of KS waves owns one 32-row
// strip of A (each wave keeps the MFMA A fragments of its K slice of 8*GPS in registers for the whole launch) and
// walks column tiles first, first+step, ...; the next tile's B fragments are in flight under the current tile's
// MFMA chain (GPS*4 MFMAs per wave and tile, 64 at GPS = 16 against 16 in the 16-wave kernels), and only KS
// partial tiles meet in LDS.  With KS = 4 two or three workgroups share a CU, so one workgroup's cross-wave sum
// and store run under the other's MFMAs.  The summation order (k inside a slice, then slices 0..KS-1) depends on
// (K, KS, GPS) only, never on M or on the launch geometry.
template <int GPS, int KS, bool KCB>
__global__ __launch_bounds__(KS * 64) void gemm_walk_kernel(cic_gemm_args g, int strips) {
    static_assert(16 % KS == 0, "accumulator registers are dealt evenly to the waves");
    constexpr int EPW = 16 / KS;                       // accumulator registers each wave finishes
    __shared__ float red[2 * KS * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, ks = tid >> 6;
    const int h = lane >> 5, r = lane & 31;
    const int tiles_n = (g.N + 31) / 32;
    // walker = blockIdx % step (step a multiple of 8): the workgroups that read the same weight tiles share an XCD
    const int step = gridDim.x / strips;
    const int first = blockIdx.x % step, strip = blockIdx.x / step;
    int m0 = strip * 32;
    const bool blk2 = g.rows_blk > 0 && m0 >= g.rows_blk;   // row blocks: see gemm_rega_kernel
    const float* __restrict__ gA = blk2 ? g.A_b : g.A;
    const float* __restrict__ gA2 = blk2 ? g.A2_b : g.A2;
    float* __restrict__ gC = blk2 ? g.C_b : g.C;
    const int Mloc = g.rows_blk > 0 ? (blk2 ? g.M - g.rows_blk : min(g.M, g.rows_blk)) : g.M;
    if (blk2) m0 -= g.rows_blk;
    const int m = m0 + r;
    const int K1 = g.K, Kt = g.K + g.K2;
    const bool mok = m < Mloc;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const int mc = mok ? m : Mloc - 1;
    f32x4 af[GPS];
#pragma unroll
    for (int i = 0; i < GPS; ++i) {
        const int k = 8 * (ks * GPS + i) + 4 * h;
        const bool kok = k < Kt;
        const bool second = kok && k >= K1;
        const int kp = second ? g.K2 : K1;
        int kk = second ? k - K1 : k;
        kk = kk < kp - 4 ? kk : kp - 4;
        const f32x4 a = *reinterpret_cast<const f32x4*>((second ? gA2 : gA) + (size_t)mc * (second ? g.lda2 : g.lda) + kk);
        af[i] = (kok && mok) ? a : z4;
    }
    // B fragments live in ONE register set that is refilled chunk by chunk: as soon as the MFMAs of chunk c (CH groups
    // of 8 k) of the current tile are issued, the loads of chunk c of the NEXT tile go out into the same registers, a
    // full tile period (~GPS*4 MFMAs) before they are needed.
    constexpr int CH = 4;
    static_assert(GPS % CH == 0, "whole chunks");
    f32x4 bf[GPS];
    auto load_chunk = [&](int c, int t) {
        const int tt = t < tiles_n ? t : tiles_n - 1;
        const int n = tt * 32 + r;
        const bool nok = t < tiles_n && n < g.N;
        const int nc = n < g.N ? n : g.N - 1;
#pragma unroll
        for (int ii = 0; ii < CH; ++ii) {
            const int i = c * CH + ii;
            const int k = 8 * (ks * GPS + i) + 4 * h;
            const bool kok = k < Kt;
            const bool second = kok && k >= K1;
            const float* B = second ? g.B2 : g.B;
            const int ldb = second ? g.ldb2 : g.ldb;
            const int kp = second ? g.K2 : K1;
            int kk = second ? k - K1 : k;
            kk = kk < kp - 4 ? kk : kp - 4;
            f32x4 b;
            if (KCB) {
                b = *reinterpret_cast<const f32x4*>(B + (size_t)nc * ldb + kk);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = B[(size_t)(kk + j) * ldb + nc];
            }
            bf[i] = (kok && nok) ? b : z4;
        }
    };
#pragma unroll
    for (int c = 0; c < GPS / CH; ++c) load_chunk(c, first);
    int buf = 0;
#pragma unroll 1
    for (int t = first; t < tiles_n; t += step) {
        const int n = t * 32 + r;
        const int ncl = n < g.N ? n : g.N - 1;
        float bias_v = 0.f, cold[EPW];                 // epilogue operands fetched before the barrier
        if (g.bias) bias_v = g.bias[ncl];
#pragma unroll
        for (int q = 0; q < EPW; ++q) {
            const int e = ks + KS * q;
            const int mm = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            const int mcl = mm < Mloc ? mm : Mloc - 1;
            cold[q] = g.accumulate ? gC[(size_t)mcl * g.ldc + ncl] : 0.f;
        }
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int c = 0; c < GPS / CH; ++c) {
#pragma unroll
            for (int ii = 0; ii < CH; ++ii)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c * CH + ii][s], bf[c * CH + ii][s], acc, 0, 0, 0);
            load_chunk(c, t + step);                   // refill: the same chunk of the next tile
        }
        float* rb = red + buf * (KS * 16 * 64);
#pragma unroll
        for (int e = 0; e < 16; ++e) rb[(ks * 16 + e) * 64 + lane] = acc[e];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < EPW; ++q) {
            const int e = ks + KS * q;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < KS; ++w) v += rb[(w * 16 + e) * 64 + lane];
            v += bias_v + cold[q];
            if (g.relu) v = fmaxf(v, 0.f);
            const int mm = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (mm < Mloc && n < g.N) gC[(size_t)mm * g.ldc + n] = v;
        }
        buf ^= 1;
    }
}



// Strip walker on v_mfma_f32_16x16x4_f32: 16-wide column tiles.  Same structure as gemm_walk_kernel (a workgroup of
// KS = 8 waves owns a 32-row strip, keeps the MFMA A fragments of its two 16-row tiles for its K slice in registers
// and walks column tiles; the B fragments of the next tile are loaded while the current tile's partial sums meet in
// LDS), but a tile is 32 x 16: twice as many work units as 32-wide tiles (9488 / 16 = 593 tiles x 8 strips spread
// evenly over the workgroups), every B fragment feeds two MFMA chains, a fragment load touches 16 cache lines of 64 B
// instead of 32 of 32 B, and only 8 accumulator registers per wave cross LDS.
//   operand layout: lane l -> (i = l & 15, q = l >> 4); A: row i, k = 16g + 4q + s for MFMA s of group g (one dwordx4
//   per group); B: column i likewise; D: 4 registers v: row 4q + v, column i.
typedef float f32x4acc __attribute__((ext_vector_type(4)));
template <int GPS, int KS, bool DEEP>   // K slice per wave = 16*GPS; DEEP: B fragments two tiles ahead (short K slices)
__global__ __launch_bounds__(KS * 64) void gemm_walk16_kernel(cic_gemm_args g, int strips) {
    static_assert(KS == 8, "8 accumulator registers (2 row tiles x 4) dealt one per wave");
    __shared__ float red[2 * KS * 8 * 64];
    const int tid = threadIdx.x, lane = tid & 63, ks = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform
    const int li = lane & 15, lq = lane >> 4;
    const int tiles_n = (g.N + 15) / 16;
    const int step = gridDim.x / strips;                     // walkers per strip, a multiple of 8 (XCD sharing)
    const int first = blockIdx.x % step, strip = blockIdx.x / step;
    int m0 = strip * 32;
    const bool blk2 = g.rows_blk > 0 && m0 >= g.rows_blk;   // row blocks: see gemm_rega_kernel
    const float* __restrict__ gA = blk2 ? g.A_b : g.A;
    const float* __restrict__ gA2 = blk2 ? g.A2_b : g.A2;
    float* __restrict__ gC = blk2 ? g.C_b : g.C;
    const int Mloc = g.rows_blk > 0 ? (blk2 ? g.M - g.rows_blk : min(g.M, g.rows_blk)) : g.M;
    if (blk2) m0 -= g.rows_blk;
    const int K1 = g.K, Kt = g.K + g.K2;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 af[2][GPS];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int m = m0 + 16 * rt + li;
        const bool mok = m < Mloc;
        const int mc = mok ? m : Mloc - 1;
#pragma unroll
        for (int i = 0; i < GPS; ++i) {
            const int k = 16 * (ks * GPS + i) + 4 * lq;
            const bool kok = k < Kt;
            const bool second = kok && k >= K1;
            const int kp = second ? g.K2 : K1;
            int kk = second ? k - K1 : k;
            kk = kk < kp - 4 ? kk : kp - 4;
            const f32x4 a = *reinterpret_cast<const f32x4*>((second ? gA2 : gA) + (size_t)mc * (second ? g.lda2 : g.lda) + kk);
            af[rt][i] = (kok && mok) ? a : z4;
        }
    }
    f32x4 bf[GPS], bq[DEEP ? GPS : 1];
    // column split (see cic.h): tiles at or beyond n_split take the second operand pair only, from B2_tail
    const int split_t = g.n_split > 0 ? g.n_split / 16 : tiles_n;
    auto load_b = [&](f32x4 (&dst)[GPS], int t) {
        const int tt = t < tiles_n ? t : tiles_n - 1;
        const int n = tt * 16 + li;
        const bool nok = t < tiles_n && n < g.N;
        const int nc = n < g.N ? n : g.N - 1;
        const bool tail = tt >= split_t;
#pragma unroll
        for (int i = 0; i < GPS; ++i) {
            const int k0 = 16 * (ks * GPS + i);             // wave-uniform: the operand choice stays in scalar registers
            const int k = k0 + 4 * lq;
            const bool kok = k < Kt;
            const bool second = k0 >= K1;                   // K1 % 16 == 0 (rega_ok: K % 8, walk16: Kt == 1024, K1 = 512)
            const float* B = second ? (tail ? g.B2_tail : g.B2) : g.B;
            const int ldb = second ? (tail ? g.ldb2_tail : g.ldb2) : g.ldb;
            const int kp = second ? g.K2 : K1;
            int kk = second ? k - K1 : k;
            kk = kk < kp - 4 ? kk : kp - 4;
            const int row = tail ? nc - g.n_split : nc;
            const bool use = kok && nok && (second || !tail);
            // unused fragments (padding, and the first pair's K slices of tail tiles) read a block of zeros
            const float* pb = use ? B + (size_t)row * ldb + kk : g_zero16;
            dst[i] = *reinterpret_cast<const f32x4*>(pb);
        }
    };
    load_b(bf, first);
    if (DEEP) load_b(reinterpret_cast<f32x4 (&)[GPS]>(bq), first + step);
    // the output this wave finishes: accumulator register ks of the 8 -> row tile ks >> 2, register ks & 3
    const int mm = m0 + 16 * (ks >> 2) + 4 * lq + (ks & 3);
    const int mcl = mm < Mloc ? mm : Mloc - 1;
    int buf = 0;
#pragma unroll 1
    for (int t = first; t < tiles_n; t += step) {
        const int n = t * 16 + li;
        const int ncl = n < g.N ? n : g.N - 1;
        float bias_v = 0.f, cold = 0.f;                 // epilogue operands fetched before the barrier
        const bool tail = t >= split_t;
        if (tail) { if (g.bias_tail) bias_v = g.bias_tail[ncl - g.n_split]; }
        else if (g.bias) bias_v = g.bias[ncl];
        if (g.accumulate && !tail) cold = gC[(size_t)mcl * g.ldc + ncl];
        f32x4acc acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < GPS; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][i][s], bf[i][s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][i][s], bf[i][s], acc1, 0, 0, 0);
            }
        if (DEEP) {                                     // tile t+1 is already here; tile t+2 goes out now
#pragma unroll
            for (int i = 0; i < GPS; ++i) bf[i] = bq[i];
            load_b(reinterpret_cast<f32x4 (&)[GPS]>(bq), t + 2 * step);
        } else {
            load_b(bf, t + step);                       // next tile's fragments in flight under the cross-wave sum
        }
        float* rb = red + buf * (KS * 8 * 64);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            rb[(ks * 8 + v) * 64 + lane] = acc0[v];
            rb[(ks * 8 + 4 + v) * 64 + lane] = acc1[v];
        }
        __syncthreads();
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < KS; ++w) v += rb[(w * 8 + ks) * 64 + lane];
        v += bias_v + cold;
        if (g.relu) v = fmaxf(v, 0.f);
        if (mm < Mloc && n < g.N) {
            if (tail) (blk2 ? g.C_tail_b : g.C_tail)[(size_t)mm * g.ldc_tail + (n - g.n_split)] = v;
            else gC[(size_t)mm * g.ldc + n] = v;
        }
        buf ^= 1;
    }
}

// The 16-wide strip walker on bf16 parts (f32 results, see split_bf16 / gemm_bfx_kernel): v_mfma_f32_16x16x32_bf16 runs
// 16x the K per instruction of the f32 MFMA, six part products replace one f32 product (2.67x the MFMA rate), and the
// split of a B fragment (13 VALU instructions per two floats) frees its raw registers at once, so the next tile's
// loads are in flight under this tile's MFMAs AND the cross-wave sum.  Same work split and output order as
// gemm_walk16_kernel<8, 8, false>; K slice per wave = 128 = 4 k-steps of 32 (Kt == 1024, K1 % 32 == 0: no K padding).
//   operand layout: lane l -> (i = l & 15, q = l >> 4); A: row i, k = 32 j + 8 q + 0..7 for k-step j (two dwordx4);
//   B: column i likewise; D: 4 registers v: row 4q + v, column i.
// NP = 1 (r4, CIC_PRECISION_BF16): one bf16 part per operand, one MFMA per k-step.  B_parts / B2_parts / B2_tail_parts (bf16
// images of the three weight matrices, cic_round_bf16) replace the f32 weights where given: half the streamed bytes.
template <int KS, int NP>
__global__ __launch_bounds__(KS * 64) void gemm_walk16bf_kernel(cic_gemm_args g, int strips) {
    static_assert(KS == 8, "8 accumulator registers (2 row tiles x 4) dealt one per wave");
    constexpr int JS = 4;                                   // k-steps per wave
    __shared__ float red[2 * KS * 8 * 64];
    const int tid = threadIdx.x, lane = tid & 63, ks = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform
    const int li = lane & 15, lq = lane >> 4;
    const int tiles_n = (g.N + 15) / 16;
    const int step = gridDim.x / strips;                     // walkers per strip, a multiple of 8 (XCD sharing)
    const int first = blockIdx.x % step, strip = blockIdx.x / step;
    int m0 = strip * 32;
    const bool blk2 = g.rows_blk > 0 && m0 >= g.rows_blk;   // row blocks: see gemm_rega_kernel
    const float* __restrict__ gA = blk2 ? g.A_b : g.A;
    const float* __restrict__ gA2 = blk2 ? g.A2_b : g.A2;
    float* __restrict__ gC = blk2 ? g.C_b : g.C;
    const int Mloc = g.rows_blk > 0 ? (blk2 ? g.M - g.rows_blk : min(g.M, g.rows_blk)) : g.M;
    if (blk2) m0 -= g.rows_blk;
    const int K1 = g.K;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    if (gemm_skipped(g)) return;                            // every caption has ended (grid-uniform)
    auto parts8 = [](const f32x4 lo, const f32x4 hi, bf16x8 (&dst)[NP][JS], int j) {
        bf16x4 pl[NP], ph[NP];
        split_bf16<NP>(lo, pl);
        split_bf16<NP>(hi, ph);
#pragma unroll
        for (int p = 0; p < NP; ++p) dst[p][j] = __builtin_shufflevector(pl[p], ph[p], 0, 1, 2, 3, 4, 5, 6, 7);
    };
    bf16x8 ap[2][NP][JS];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int m = m0 + 16 * rt + li;
        const bool mok = m < Mloc;
        const int mc = mok ? m : Mloc - 1;
#pragma unroll
        for (int j = 0; j < JS; ++j) {
            const int k0 = 32 * (ks * JS + j);             // wave-uniform
            const bool second = k0 >= K1;
            const float* src = (second ? gA2 : gA) + (size_t)mc * (second ? g.lda2 : g.lda) + (second ? k0 - K1 : k0) + 8 * lq;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
            parts8(mok ? lo : z4, mok ? hi : z4, ap[rt], j);
        }
    }
    f32x4 raw[2 * JS];
    // the weights as bf16 images (NP == 1 only): one 16-byte load per k-step instead of two, no conversion
    const bool img = NP == 1 && g.B_parts && g.B2_parts && (g.n_split == 0 || g.B2_tail_parts);
    bf16x8 rawh[NP == 1 ? JS : 1];
    // column split (see cic.h): tiles at or beyond n_split take the second operand pair only, from B2_tail
    const int split_t = g.n_split > 0 ? g.n_split / 16 : tiles_n;
    auto load_raw = [&](int t) {
        if (NP == 1 && img) {
            const int tt = t < tiles_n ? t : tiles_n - 1;
            const int n = tt * 16 + li;
            const bool nok = t < tiles_n && n < g.N;
            const int nc = n < g.N ? n : g.N - 1;
            const bool tail = tt >= split_t;
#pragma unroll
            for (int j = 0; j < JS; ++j) {
                const int k0 = 32 * (ks * JS + j);
                const bool second = k0 >= K1;
                const uint16_t* B = second ? (tail ? g.B2_tail_parts : g.B2_parts) : g.B_parts;
                const int ldb = second ? (tail ? g.ldb2_tail : g.ldb2) : g.ldb;
                const int row = tail ? nc - g.n_split : nc;
                const bool use = nok && (second || !tail);
                const uint16_t* pb = use ? B + (size_t)row * ldb + (second ? k0 - K1 : k0) + 8 * lq : reinterpret_cast<const uint16_t*>(g_zero16);
                rawh[NP == 1 ? j : 0] = *reinterpret_cast<const bf16x8*>(pb);
            }
            return;
        }
        const int tt = t < tiles_n ? t : tiles_n - 1;
        const int n = tt * 16 + li;
        const bool nok = t < tiles_n && n < g.N;
        const int nc = n < g.N ? n : g.N - 1;
        const bool tail = tt >= split_t;
#pragma unroll
        for (int j = 0; j < JS; ++j) {
            const int k0 = 32 * (ks * JS + j);             // wave-uniform: the operand choice stays in scalar registers
            const bool second = k0 >= K1;
            const float* B = second ? (tail ? g.B2_tail : g.B2) : g.B;
            const int ldb = second ? (tail ? g.ldb2_tail : g.ldb2) : g.ldb;
            const int row = tail ? nc - g.n_split : nc;
            const bool use = nok && (second || !tail);
            // unused fragments (padding, and the first pair's K slices of tail tiles) read a block of zeros
            const float* pb = use ? B + (size_t)row * ldb + (second ? k0 - K1 : k0) + 8 * lq : g_zero16;
            raw[2 * j] = *reinterpret_cast<const f32x4*>(pb);
            raw[2 * j + 1] = *reinterpret_cast<const f32x4*>(pb + 4);
        }
    };
    load_raw(first);
    // the output this wave finishes: accumulator register ks of the 8 -> row tile ks >> 2, register ks & 3
    const int mm = m0 + 16 * (ks >> 2) + 4 * lq + (ks & 3);
    const int mcl = mm < Mloc ? mm : Mloc - 1;
    int buf = 0;
#pragma unroll 1
    for (int t = first; t < tiles_n; t += step) {
        const int n = t * 16 + li;
        const int ncl = n < g.N ? n : g.N - 1;
        float bias_v = 0.f, cold = 0.f;                 // epilogue operands fetched before the barrier
        const bool tail = t >= split_t;
        if (tail) { if (g.bias_tail) bias_v = g.bias_tail[ncl - g.n_split]; }
        else if (g.bias) bias_v = g.bias[ncl];
        if (g.accumulate && !tail) cold = gC[(size_t)mcl * g.ldc + ncl];
        bf16x8 bp[NP][JS];
        if (NP == 1 && img) {
#pragma unroll
            for (int j = 0; j < JS; ++j) bp[0][j] = rawh[NP == 1 ? j : 0];
        } else {
#pragma unroll
            for (int j = 0; j < JS; ++j) parts8(raw[2 * j], raw[2 * j + 1], bp, j);
        }
        load_raw(t + step);                             // next tile in flight under the MFMAs and the cross-wave sum
        f32x4acc acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        // part products with pa + pb <= 2, smallest first (parts 0 / 1 / 2 carry bits 1-8 / 9-16 / 17-24); NP = 1: the one product
        constexpr int NC = NP == 3 ? 6 : 1;
        constexpr int PA[6] = {NP == 3 ? 0 : 0, NP == 3 ? 2 : 0, NP == 3 ? 1 : 0, 0, NP == 3 ? 1 : 0, 0};
        constexpr int PB[6] = {NP == 3 ? 2 : 0, 0, NP == 3 ? 1 : 0, NP == 3 ? 1 : 0, 0, 0};
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int j = 0; j < JS; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0][PA[c]][j], bp[PB[c]][j], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[1][PA[c]][j], bp[PB[c]][j], acc1, 0, 0, 0);
            }
        float* rb = red + buf * (KS * 8 * 64);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            rb[(ks * 8 + v) * 64 + lane] = acc0[v];
            rb[(ks * 8 + 4 + v) * 64 + lane] = acc1[v];
        }
        __syncthreads();
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < KS; ++w) v += rb[(w * 8 + ks) * 64 + lane];
        v += bias_v + cold;
        if (g.relu) v = fmaxf(v, 0.f);
        if (mm < Mloc && n < g.N) {
            if (tail) (blk2 ? g.C_tail_b : g.C_tail)[(size_t)mm * g.ldc_tail + (n - g.n_split)] = v;
            else gC[(size_t)mm * g.ldc + n] = v;
        }
        buf ^= 1;
    }
}

// Column walker with the weight tile in LDS (the logit product, K = 16*NG = 512): NO K split, so no cross-wave sum.
// A workgroup of 4 waves (one per SIMD) owns 64 rows: wave w keeps the MFMA A fragments of its 16 rows for the WHOLE
// K in registers (K/4 VGPRs per lane) and the workgroup walks 16-column weight tiles.  A tile [16 x K] is staged in LDS
// once per workgroup with fully coalesced 2 KB row reads (double buffered: the next tile's global loads are issued at
// the top of a tile, its LDS writes are interleaved with the last MFMAs of the tile, ONE barrier per tile), each wave
// reads its B fragments back with one ds_read_b128 per four MFMAs (row stride K + 4 floats), a chunk of 8 reads ahead
// of the MFMAs that use them, and runs four independent accumulator chains (MFMA s of every group; summed
// (0+1)+(2+3) at the end, a fixed order that depends on K only).  The previous tile's stores ride in the first MFMA
// chunk of the next tile.  Scheduling barriers pin this order: left alone, hipcc sinks every LDS read to just before
// its first use and hoists the waits for the staged loads above the MFMA chain.
// Work units = row groups x column tiles (4 x 593 for the paired logit product) dealt round-robin to one workgroup
// per CU: 9.3 tiles per workgroup, 4096 MFMA cycles each.
// The A fragments are read ONCE per workgroup (128 KB per CU: ~5 us at the rate a CU fills when all of them load).
//   operand layout as in gemm_walk16_kernel: lane l -> (i = l & 15, q = l >> 4); A: row i, k = 16g + 4q + s;
//   B: column i likewise; D: 4 registers v: row 4q + v, column i.
template <int NG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_ldsb_walk_kernel(
    cic_gemm_args g, int row_groups, int walkers) {
    constexpr int K = 16 * NG, LDB = K + 4, TILE = 16 * LDB;
    constexpr int CH = 8, NC = NG / CH;
    constexpr int F4 = 16 * (K / 4) / 256;                 // float4 per thread and staged tile
    static_assert(F4 == CH, "one staged float4 rides behind each MFMA group of the last chunk");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform
    const int li = lane & 15, lq = lane >> 4;
    unsigned long long* stamps = CIC_STAMP_BUF(g_stamp_buf);             // diagnostics: [workgroup][wave][64] phase stamps (100 MHz)
    if (stamps) stamps += ((size_t)blockIdx.x * 4 + w) * 64;
    int sidx = 0;
    auto stamp = [&]() { if (stamps && lane == 0 && sidx < 60) stamps[sidx] = __builtin_amdgcn_s_memrealtime(); ++sidx; };
    stamp();
    // workgroups b, b+8, ... share an XCD: give the row groups that read the same weight tile at the same time
    // consecutive ids on ONE XCD, so that a tile leaves HBM / the Infinity Cache once per launch
    const int per_xcd = gridDim.x / 8;
    const int wg = (gridDim.x % 8 == 0) ? (blockIdx.x % 8) * per_xcd + blockIdx.x / 8 : blockIdx.x;
    const int rg = wg % row_groups, first = wg / row_groups;
    if (first >= walkers) return;                          // whole workgroup: no barrier is skipped by a part of it
    const int tiles_n = (g.N + 15) / 16;
    // row blocks (see gemm_rega_kernel): each block is cut into its own groups of 64 rows, groups never straddle
    const int rg_a = g.rows_blk > 0 ? (g.rows_blk + 63) / 64 : row_groups;
    const bool blk2 = rg >= rg_a;
    const int m0 = (blk2 ? rg - rg_a : rg) * 64;
    const float* __restrict__ gA = blk2 ? g.A_b : g.A;
    float* __restrict__ gC = blk2 ? g.C_b : g.C;
    const int Mloc = g.rows_blk > 0 ? (blk2 ? g.M - g.rows_blk : min(g.M, g.rows_blk)) : g.M;
    const int m = m0 + 16 * w + li;
    const int mc = m < Mloc ? m : Mloc - 1;                // rows beyond M re-read row M-1: never stored

    // staging: the tile is 16 rows of K floats = 16 * K/4 float4, dealt to the 256 threads row-major.  Rows beyond N
    // re-read row N-1 (their columns are never stored; no select on the loaded value: it would make the wave wait for
    // these loads before the MFMA chain instead of after it).
    f32x4 stg[F4];
    auto load_tile = [&](int t) {
        const int tt = t < tiles_n ? t : tiles_n - 1;
#pragma unroll
        for (int e = 0; e < F4; ++e) {
            const int j = tid + 256 * e;
            const int r = j / (K / 4), c4 = j % (K / 4);
            const int n = tt * 16 + r;
            const int nc = n < g.N ? n : g.N - 1;
            stg[e] = *reinterpret_cast<const f32x4*>(g.B + (size_t)nc * g.ldb + 4 * c4);
        }
    };
    auto store_piece = [&](float* dst, int e) {
        const int j = tid + 256 * e;
        const int r = j / (K / 4), c4 = j % (K / 4);
        *reinterpret_cast<f32x4*>(dst + r * LDB + 4 * c4) = stg[e];
    };
    load_tile(first);                                      // the first weight tile goes out before the A fragments
    f32x4 af[NG];
    // the first half of the A fragments goes out with the first tile; the second half after the first barrier, so
    // that the first tile's MFMA chain starts after 24 of the 40 prologue loads and the rest arrives under it
    // (hipcc waits for ALL outstanding loads before the first LDS write)
    const float* arow = gA + (size_t)mc * g.lda + 4 * lq;
#pragma unroll
    for (int i = 0; i < NG / 2; ++i) af[i] = *reinterpret_cast<const f32x4*>(arow + 16 * i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < F4; ++e) store_piece(lds, e);
    __syncthreads();
#pragma unroll
    for (int i = NG / 2; i < NG; ++i) af[i] = *reinterpret_cast<const f32x4*>(arow + 16 * i);
    __builtin_amdgcn_sched_barrier(0);
    stamp();
    if (stamps && lane == 0) { stamps[60] = __builtin_amdgcn_s_memrealtime(); stamps[61] = __builtin_amdgcn_s_memtime(); }
    int buf = 0;
    f32x4acc prev[4];                                      // the previous tile's accumulators, stored under this tile's MFMAs
    int prev_n = -1;
    float prev_bias = 0.f;
    auto store_row = [&](int v) {                          // output register v of the previous tile: row 4q + v, column i
        const int mm = m0 + 16 * w + 4 * lq + v;
        const float x = (prev[0][v] + prev[1][v]) + (prev[2][v] + prev[3][v]) + prev_bias;
        if (prev_n >= 0 && mm < Mloc && prev_n < g.N) gC[(size_t)mm * g.ldc + prev_n] = x;
    };
#pragma unroll 1
    for (int t = first; t < tiles_n; t += walkers) {
        load_tile(t + walkers);                            // next tile's rows in flight under this tile's MFMAs
        const int n = t * 16 + li;
        const int ncl = n < g.N ? n : g.N - 1;
        float bias_v = 0.f;
        if (g.bias) bias_v = g.bias[ncl];
        const float* bt = lds + buf * TILE + li * LDB + 4 * lq;
        float* nxt = lds + (buf ^ 1) * TILE;               // every wave left this buffer before the last barrier
        f32x4acc acc[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[s] = f32x4acc{0.f, 0.f, 0.f, 0.f};
        f32x4 bq[2][CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) bq[0][i] = *reinterpret_cast<const f32x4*>(bt + 16 * i);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (c + 1 < NC) {
#pragma unroll
                for (int i = 0; i < CH; ++i) bq[(c + 1) & 1][i] = *reinterpret_cast<const f32x4*>(bt + 16 * ((c + 1) * CH + i));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < CH; ++i) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c * CH + i][s], bq[c & 1][i][s], acc[s], 0, 0, 0);
                if (c == 0 && i < 4) { store_row(i); __builtin_amdgcn_sched_barrier(0); }
                if (c == NC - 1) { store_piece(nxt, i); __builtin_amdgcn_sched_barrier(0); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (stamps) { asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0])); stamp(); }
#pragma unroll
        for (int s = 0; s < 4; ++s) prev[s] = acc[s];
        prev_n = n;
        prev_bias = bias_v;
        if (stamps) stamp();
        __syncthreads();
        if (stamps) stamp();
        buf ^= 1;
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) store_row(v);
    if (stamps && lane == 0) { stamps[62] = __builtin_amdgcn_s_memrealtime(); stamps[63] = __builtin_amdgcn_s_memtime(); }
}

// Eight-wave form of the walker above: the same 64 rows x 16-column tiles, but every 16-row tile is shared by TWO waves
// that each take one half of K (A fragments: 64 VGPRs per lane), so a SIMD holds two waves and one wave's LDS reads,
// staging writes, stores and waits run under the other's MFMAs (with one wave per SIMD they all sit inside the chain:
// 2.9 us per tile for 1.9 us of MFMAs).  The two K halves of a tile meet through 4 KB of LDS: the upper-half wave
// parks its four sums there before the tile's barrier, the lower-half wave adds them (lower + upper: a fixed order)
// while it stores the tile under the next tile's MFMAs.
//
// EPI: the fused vocabulary epilogue (cic.h, "Row partials of the vocabulary").  The lower-half wave of a row tile is the
// one that finishes a logit (lower + upper + bias) while the NEXT tile's MFMAs run; with EPI it also feeds the value
// to the running partial of its row - 4 rows x 1 column per lane and tile, reduced over the 16 lanes of a row group
// once, after the walk.  Part index = the walker's index: a workgroup's columns are the tiles first, first + walkers, ...
// Noise: injected uniforms (tests) or ONE Philox call per lane and tile - lane c of a quad draws the four uniforms
// of row 4*lq + c at the quad's four columns, and a 4x4 transpose inside the quad (DPP) hands every lane its own
// column of rows 4*lq + 0..3.
template <int NG, int KH, int EPI>   // KH K parts per row tile: 4 * KH waves, KH per SIMD; EPI: the fused vocabulary epilogue
__global__ __launch_bounds__(256 * KH) __attribute__((amdgpu_waves_per_eu(KH, KH))) void gemm_ldsb2_walk_kernel(
    cic_gemm_args g, int row_groups, int walkers, cic_logit_epilogue epi) {
    constexpr int K = 16 * NG, LDB = K + 4, TILE = 16 * LDB, NH = NG / KH, NT = 256 * KH;
    constexpr int CH = 8, NC = NH / CH;
    constexpr int F4 = 16 * (K / 4) / NT;                  // float4 per thread and staged tile
    static_assert(F4 <= CH && NC >= 1, "the staged float4 ride behind the MFMA groups of the last chunk");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* pairbuf = lds + 2 * TILE;                       // [2 parities][4 row tiles][KH - 1 upper parts][4 registers][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform
    const int rt = w & 3, kh = w >> 2;                     // row tile, K half
    const int li = lane & 15, lq = lane >> 4;
    const int per_xcd = gridDim.x / 8;
    const int wg = (gridDim.x % 8 == 0) ? (blockIdx.x % 8) * per_xcd + blockIdx.x / 8 : blockIdx.x;
    const int rg = wg % row_groups, first = wg / row_groups;
    if (first >= walkers) return;                          // whole workgroup: no barrier is skipped by a part of it
    const int tiles_n = (g.N + 15) / 16;
    const int rg_a = g.rows_blk > 0 ? (g.rows_blk + 63) / 64 : row_groups;
    const bool blk2 = rg >= rg_a;
    const int m0 = (blk2 ? rg - rg_a : rg) * 64;
    const float* __restrict__ gA = blk2 ? g.A_b : g.A;
    float* __restrict__ gC = blk2 ? g.C_b : g.C;
    const int Mloc = g.rows_blk > 0 ? (blk2 ? g.M - g.rows_blk : min(g.M, g.rows_blk)) : g.M;
    const int m = m0 + 16 * rt + li;
    const int mc = m < Mloc ? m : Mloc - 1;                // rows beyond M re-read row M-1: never stored
    f32x4 stg[F4];
    auto load_tile = [&](int t) {
        const int tt = t < tiles_n ? t : tiles_n - 1;
#pragma unroll
        for (int e = 0; e < F4; ++e) {
            const int j = tid + NT * e;
            const int r = j / (K / 4), c4 = j % (K / 4);
            const int n = tt * 16 + r;
            const int nc = n < g.N ? n : g.N - 1;
            stg[e] = *reinterpret_cast<const f32x4*>(g.B + (size_t)nc * g.ldb + 4 * c4);
        }
    };
    auto store_piece = [&](float* dst, int e) {
        const int j = tid + NT * e;
        const int r = j / (K / 4), c4 = j % (K / 4);
        *reinterpret_cast<f32x4*>(dst + r * LDB + 4 * c4) = stg[e];
    };
    load_tile(first);
    f32x4 af[NH];
    const float* arow = gA + (size_t)mc * g.lda + 16 * NH * kh + 4 * lq;
#pragma unroll
    for (int i = 0; i < NH; ++i) af[i] = *reinterpret_cast<const f32x4*>(arow + 16 * i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < F4; ++e) store_piece(lds, e);
    __syncthreads();
    int buf = 0;
    float prev[4];                                         // the previous tile's sums of this wave's K half
    int prev_n = -1;
    float prev_bias = 0.f;
    // ---- fused epilogue (EPI).  Two software pipelines ride in the 16 slots between the MFMA groups of a tile:
    //   upper-half wave (kh == 1): the Gumbel noise of THIS tile's 4 rows x 1 column per lane - one Philox call per lane
    //     (lane c of a quad draws the four uniforms of row 4*lq + c at the quad's four columns, one round per slot), a 4x4
    //     transpose inside the quad (DPP), -log(-log u) - parked in LDS for the lower-half wave (tests inject U instead);
    //   lower-half wave (kh == 0): finishes the PREVIOUS tile's logits (lower + upper + bias, stored raw) and feeds them,
    //     with that noise, to the running partial of their row: one (row, column) per slot group, straight-line code.
    const cic_logit_epi_rows er = epi.blk[blk2 ? 1 : 0];
    float* noisebuf = pairbuf + 2 * 4 * (KH - 1) * 4 * 64;  // [2 parities][4 row tiles][4 registers][64 lanes]
    RowPart rp[4];
    int cons[4];
    float gprev[4] = {0.f, 0.f, 0.f, 0.f};
    PhiloxState phs;
    float uq[4] = {0.5f, 0.5f, 0.5f, 0.5f}, u4[4] = {0.5f, 0.5f, 0.5f, 0.5f};
#pragma unroll
    for (int v = 0; v < 4; ++v) { rp[v].init(); cons[v] = -1; }
    phs.init(0, 0);
    if (EPI && kh == 0 && er.cons_seq) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int mm = m0 + 16 * rt + 4 * lq + v;
            cons[v] = er.cons_seq[(size_t)(mm < Mloc ? mm : Mloc - 1) * er.cons_ld + er.cons_col];
        }
    }
    // ROLE: 0 = no epilogue (every wave; the lower halves store their rows); with EPI: 1 = lower-half wave (MODE = its rows'
    // sampling mode), 2 / 3 = upper-half wave making the noise from the Philox stream / from injected uniforms, 4 = an upper
    // wave with nothing to add.  One straight-line loop body per role: no wave-uniform branch inside the MFMA stream.
    auto walk = [&](auto role_, auto mode_) {
        constexpr int ROLE = decltype(role_)::value, MODE = decltype(mode_)::value;
        auto noise_slot = [&](int sl, int n) {             // slot sl of 16, tile column n of this lane
            if (ROLE == 3) {                               // injected uniforms (tests)
                if (sl == 0) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int mm = m0 + 16 * rt + 4 * lq + v;
                        uq[v] = er.U[(size_t)(mm < Mloc ? mm : Mloc - 1) * er.ldu + (n < g.N ? n : g.N - 1)];
                    }
                }
            } else if (sl == 0) {
                const int c = li & 3;
                const int qrow = m0 + 16 * rt + 4 * lq + c;
                phs.init((er.elem0 + (uint64_t)qrow * (uint64_t)er.ldu + (uint64_t)(n - c)) >> 2, er.seed);   // the quad's 4 columns
            } else if (sl <= 5) {
                phs.round();
                phs.round();
            } else if (sl <= 9) {
                // 4x4 transpose inside the quad, one destination register per slot: lane c takes element c of quad lane V
                const int c = li & 3;
                if (sl == 6) { u4[0] = u32_to_unit(phs.c0); u4[1] = u32_to_unit(phs.c1); u4[2] = u32_to_unit(phs.c2); u4[3] = u32_to_unit(phs.c3); }
#define QB(x, V) dpp_f32<(V) * 0x55>(x)                        /* quad_perm [V,V,V,V]: lane V of the quad to all four */
#define PICK(V) { const float b0 = QB(u4[0], V), b1 = QB(u4[1], V), b2 = QB(u4[2], V), b3 = QB(u4[3], V);          \
                  uq[V] = c == 0 ? b0 : (c == 1 ? b1 : (c == 2 ? b2 : b3)); }
                if (sl == 6) PICK(0)
                if (sl == 7) PICK(1)
                if (sl == 8) PICK(2)
                if (sl == 9) PICK(3)
#undef PICK
#undef QB
            }
            if (sl >= 10 && sl <= 13) noisebuf[((buf * 4 + rt) * 4 + (sl - 10)) * 64 + lane] = gumbel_from_u(uq[sl - 10]);
        };
        auto finish_row = [&](int v) {                     // lower-half waves: output register v of the previous tile
            const int mm = m0 + 16 * rt + 4 * lq + v;
            if ((ROLE == 1 || kh == 0) && prev_n >= 0) {
                float x = prev[v];
#pragma unroll
                for (int u = 0; u < KH - 1; ++u) x += pairbuf[((((buf ^ 1) * 4 + rt) * (KH - 1) + u) * 4 + v) * 64 + lane];
                x += prev_bias;
                if (mm < Mloc && prev_n < g.N) {
                    gC[(size_t)mm * g.ldc + prev_n] = x;
                    if (ROLE == 1) {
                        const float xe = prev_n == cons[v] ? -INFINITY : x;      // decoding constraint, AttModel.py:438-442
                        rowpart_add_m<MODE>(rp[v], er.inv_temp, xe, gprev[v], prev_n);
                    }
                }
            }
        };
        auto load_noise = [&]() {
            if (er.noise && prev_n >= 0) {
#pragma unroll
                for (int v = 0; v < 4; ++v) gprev[v] = noisebuf[(((buf ^ 1) * 4 + rt) * 4 + v) * 64 + lane];
            }
        };
#pragma unroll 1
        for (int t = first; t < tiles_n; t += walkers) {
            load_tile(t + walkers);                        // next tile's rows in flight under this tile's MFMAs
            const int n = t * 16 + li;
            const int ncl = n < g.N ? n : g.N - 1;
            float bias_v = 0.f;
            if (g.bias) bias_v = g.bias[ncl];
            const float* bt = lds + buf * TILE + li * LDB + 16 * NH * kh + 4 * lq;
            float* nxt = lds + (buf ^ 1) * TILE;           // every wave left this buffer before the last barrier
            f32x4acc acc[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[s] = f32x4acc{0.f, 0.f, 0.f, 0.f};
            f32x4 bq[2][CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) bq[0][i] = *reinterpret_cast<const f32x4*>(bt + 16 * i);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (c + 1 < NC) {
#pragma unroll
                    for (int i = 0; i < CH; ++i) bq[(c + 1) & 1][i] = *reinterpret_cast<const f32x4*>(bt + 16 * ((c + 1) * CH + i));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < CH; ++i) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c * CH + i][s], bq[c & 1][i][s], acc[s], 0, 0, 0);
                    if (ROLE == 0) {
                        if (c == 0 && i < 4) { finish_row(i); __builtin_amdgcn_sched_barrier(0); }
                    } else if (ROLE != 4) {
                        // 16 slots spread over the NC * CH MFMA groups of the tile
                        constexpr int GN = NC * CH;
#pragma unroll
                        for (int sl = (c * CH + i) * 16 / GN; sl < (c * CH + i + 1) * 16 / GN; ++sl) {
                            if (ROLE == 1) {
                                if (sl == 0) load_noise();
                                if ((sl & 3) == 1) finish_row(sl >> 2);
                            } else {
                                noise_slot(sl, n);
                            }
                        }
                        // the lower wave's slots are pinned between the MFMA groups; the upper wave's noise pipeline is left to
                        // the compiler's scheduler (measured: pinned 40.7 us, every 4th group 40.7, free 39.0 per launch)
                        if (ROLE == 1) __builtin_amdgcn_sched_barrier(0);
                    }
                    if (c == NC - 1 && i < F4) { store_piece(nxt, i); __builtin_amdgcn_sched_barrier(0); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) prev[v] = (acc[0][v] + acc[1][v]) + (acc[2][v] + acc[3][v]);
            if (kh >= 1) {
#pragma unroll
                for (int v = 0; v < 4; ++v) pairbuf[(((buf * 4 + rt) * (KH - 1) + (kh - 1)) * 4 + v) * 64 + lane] = prev[v];
            }
            prev_n = n;
            prev_bias = bias_v;
            __syncthreads();
            buf ^= 1;
        }
        if (ROLE == 1) load_noise();
        if (ROLE == 0 || ROLE == 1) {
#pragma unroll
            for (int v = 0; v < 4; ++v) finish_row(v);
        }
    };
    using std::integral_constant;
    if (!EPI) {
        walk(integral_constant<int, 0>{}, integral_constant<int, 0>{});
    } else if (kh == 0) {
        switch (er.mode) {
            case CIC_SAMPLE_NONE: walk(integral_constant<int, 1>{}, integral_constant<int, CIC_SAMPLE_NONE>{}); break;
            case CIC_SAMPLE_GREEDY: walk(integral_constant<int, 1>{}, integral_constant<int, CIC_SAMPLE_GREEDY>{}); break;
            case CIC_SAMPLE_GUMBEL_ST: walk(integral_constant<int, 1>{}, integral_constant<int, CIC_SAMPLE_GUMBEL_ST>{}); break;
            case CIC_SAMPLE_MULTINOMIAL_ST: walk(integral_constant<int, 1>{}, integral_constant<int, CIC_SAMPLE_MULTINOMIAL_ST>{}); break;
            default: walk(integral_constant<int, 1>{}, integral_constant<int, CIC_SAMPLE_MULTINOMIAL>{}); break;   // + TEACHER
        }
    } else if (kh == 1 && er.noise) {
        if (er.U) walk(integral_constant<int, 3>{}, integral_constant<int, 0>{});
        else walk(integral_constant<int, 2>{}, integral_constant<int, 0>{});
    } else {
        walk(integral_constant<int, 4>{}, integral_constant<int, 0>{});
    }
    if (EPI && kh == 0) {
        // the 16 lanes of a row group (same lq: one DPP row) hold the 16 columns of every tile: all-reduce them with DPP
        // lane exchanges (xor 1, xor 2, mirror within 8, mirror within 16 - no LDS crossbar), lane li == 0 writes
#define RP_STEP(CTRL)                                                                                              \
    {                                                                                                              \
        RowPart q;                                                                                                 \
        q.m1 = dpp_f32<CTRL>(rp[v].m1); q.s1 = dpp_f32<CTRL>(rp[v].s1);                                              \
        q.kbest = dpp_f32<CTRL>(rp[v].kbest); q.xbest = dpp_f32<CTRL>(rp[v].xbest);                                  \
        q.kidx = __float_as_int(dpp_f32<CTRL>(__int_as_float(rp[v].kidx))); q.s2 = dpp_f32<CTRL>(rp[v].s2);         \
        rowpart_merge(rp[v], er.mode, er.inv_temp, q);                                                             \
    }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            RP_STEP(DPP_QUAD_XOR1) RP_STEP(DPP_QUAD_XOR2) RP_STEP(DPP_ROW_HALF_MIRROR) RP_STEP(DPP_ROW_MIRROR)
            const int mm = m0 + 16 * rt + 4 * lq + v;
            if (li == 0 && mm < Mloc && er.part) {
                const size_t plane = (size_t)er.part_rows * walkers;
                float* pp = er.part + (size_t)mm * walkers + first;
                pp[0] = rp[v].m1; pp[plane] = rp[v].s1; pp[2 * plane] = rp[v].kbest; pp[3 * plane] = rp[v].xbest;
                pp[4 * plane] = __int_as_float(rp[v].kidx); pp[5 * plane] = rp[v].s2;
            }
        }
#undef RP_STEP
    }
}

// The eight-wave logit walker on bf16 parts (f32 results: split_bf16, six part products per k-step on
// v_mfma_f32_16x16x32_bf16, 2.67x the MFMA rate of the f32-input instruction).  Same work split, roles, epilogue and
// output order as gemm_ldsb2_walk_kernel<NG, 2, EPI>; what changes is the operand path: the staged weight tile is split
// on its way into LDS (three bf16 images of [16 columns][K + 8], rows 1040 bytes apart: conflict-free ds_read_b128), a
// wave reads the three parts of a k-step (32 k) one step ahead of the MFMAs that use them, and its A fragments are the
// three parts of its K half (96 VGPRs).  Terms of like magnitude share an accumulator chain (parts 0x2, 2x0, 1x1 | 0x1,
// 1x0 | 0x0), the chains are added smallest first.
// NP = 3: f32 results from three bf16 parts per operand; NP = 1 (r4, CIC_PRECISION_BF16): ONE part - operands rounded to bf16
// once, one MFMA per k-step, a third of the staged bytes - the logit product of the reduced-precision variant.
template <int NG, int EPI, int PRE, int NP>   // PRE: the weights arrive already cut (NP bf16 images [NP][N][K], cic_split_bf16x3 / cic_round_bf16)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_ldsb2bf_walk_kernel(
    cic_gemm_args g, int row_groups, int walkers, cic_logit_epilogue epi, const __bf16* __restrict__ wparts) {
    constexpr int KH = 2, K = 16 * NG, NT = 512;
    constexpr int LDBH = K + 8, PART = 16 * LDBH, TILEH = NP * PART;   // bf16 per staged row / part / tile
    constexpr int JS = K / KH / 32;                        // k-steps of 32 per wave
    constexpr int F4 = 16 * (K / 4) / NT;                  // float4 per thread and staged tile
    constexpr int NPC = PRE ? NP * 16 * (K / 8) / NT : F4;  // staged pieces per thread and tile (PRE: 16-byte pieces of the images)
    static_assert(NPC <= JS && (TILEH % 8) == 0, "the staged pieces ride behind the last k-steps of a tile");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __bf16* ldsh = reinterpret_cast<__bf16*>(lds);
    float* pairbuf = lds + TILEH;                          // (2 tiles of TILEH bf16 = TILEH floats) [2 parities][4 row tiles][1 upper part][4 registers][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform
    const int rt = w & 3, kh = w >> 2;                     // row tile, K half
    const int li = lane & 15, lq = lane >> 4;
    // development build: [workgroup][wave][64] phase stamps (100 MHz): start, A fragments ready, first tile staged; per tile: k loop
    // done, barrier passed; end (tools/ldsb_bf_stamps.py)
    unsigned long long* stamps = CIC_STAMP_BUF(g_stamp_buf);
    if (stamps) stamps += ((size_t)blockIdx.x * 8 + w) * 64;
    int sidx = 0;
    auto stamp = [&]() { if (stamps && lane == 0 && sidx < 62) stamps[sidx] = __builtin_amdgcn_s_memrealtime(); ++sidx; };
    stamp();
    const int per_xcd = gridDim.x / 8;
    const int wg = (gridDim.x % 8 == 0) ? (blockIdx.x % 8) * per_xcd + blockIdx.x / 8 : blockIdx.x;
    const int rg = wg % row_groups, first = wg / row_groups;
    if (first >= walkers) return;                          // whole workgroup: no barrier is skipped by a part of it
    if (gemm_skipped(g)) return;                            // every caption has ended (grid-uniform)
    const int tiles_n = (g.N + 15) / 16;
    const int rg_a = g.rows_blk > 0 ? (g.rows_blk + 63) / 64 : row_groups;
    const bool blk2 = rg >= rg_a;
    const int m0 = (blk2 ? rg - rg_a : rg) * 64;
    const float* __restrict__ gA = blk2 ? g.A_b : g.A;
    float* __restrict__ gC = blk2 ? g.C_b : g.C;
    const int Mloc = g.rows_blk > 0 ? (blk2 ? g.M - g.rows_blk : min(g.M, g.rows_blk)) : g.M;
    const int m = m0 + 16 * rt + li;
    const int mc = m < Mloc ? m : Mloc - 1;                // rows beyond M re-read row M-1: never stored
    f32x4 stg[F4];
    bf16x8 stp[PRE ? NPC : 1];
    auto load_tile = [&](int t) {
        const int tt = t < tiles_n ? t : tiles_n - 1;
        if (PRE) {
#pragma unroll
            for (int e = 0; e < NPC; ++e) {
                const int j = tid + NT * e;
                const int p = j / (16 * (K / 8)), rem = j % (16 * (K / 8));
                const int r = rem / (K / 8), c8 = rem % (K / 8);
                const int n = tt * 16 + r;
                const int nc = n < g.N ? n : g.N - 1;
                stp[e] = *reinterpret_cast<const bf16x8*>(wparts + ((size_t)p * g.N + nc) * K + 8 * c8);
            }
            return;
        }
#pragma unroll
        for (int e = 0; e < F4; ++e) {
            const int j = tid + NT * e;
            const int r = j / (K / 4), c4 = j % (K / 4);
            const int n = tt * 16 + r;
            const int nc = n < g.N ? n : g.N - 1;
            stg[e] = *reinterpret_cast<const f32x4*>(g.B + (size_t)nc * g.ldb + 4 * c4);
        }
    };
    auto store_piece = [&](__bf16* dst, int e) {          // four k of one column -> the three bf16 images
        if (PRE) {                                         // a 16-byte piece of one image, as it is
            const int j = tid + NT * e;
            const int p = j / (16 * (K / 8)), rem = j % (16 * (K / 8));
            const int r = rem / (K / 8), c8 = rem % (K / 8);
            *reinterpret_cast<bf16x8*>(dst + p * PART + r * LDBH + 8 * c8) = stp[e];
            return;
        }
        const int j = tid + NT * e;
        const int r = j / (K / 4), c4 = j % (K / 4);
        bf16x4 parts[NP];
        split_bf16<NP>(stg[e], parts);
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<bf16x4*>(dst + p * PART + r * LDBH + 4 * c4) = parts[p];
    };
    load_tile(first);
    bf16x8 ap[NP][JS];
    {
        const float* arow = gA + (size_t)mc * g.lda + (K / KH) * kh + 8 * lq;
#pragma unroll
        for (int j = 0; j < JS; ++j) {
            bf16x4 pl[NP], ph[NP];
            split_bf16<NP>(*reinterpret_cast<const f32x4*>(arow + 32 * j), pl);
            split_bf16<NP>(*reinterpret_cast<const f32x4*>(arow + 32 * j + 4), ph);
#pragma unroll
            for (int p = 0; p < NP; ++p) ap[p][j] = __builtin_shufflevector(pl[p], ph[p], 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (stamps) { asm volatile("" :: "v"(ap[0][0]), "v"(ap[NP - 1][JS - 1])); stamp(); }
#pragma unroll
    for (int e = 0; e < NPC; ++e) store_piece(ldsh, e);
    __syncthreads();
    stamp();
    int buf = 0;
    float prev[4];                                         // the previous tile's sums of this wave's K half
    int prev_n = -1;
    float prev_bias = 0.f;
    // ---- fused epilogue (EPI).  Two software pipelines ride in the 16 slots between the MFMA groups of a tile:
    //   upper-half wave (kh == 1): the Gumbel noise of THIS tile's 4 rows x 1 column per lane - one Philox call per lane
    //     (lane c of a quad draws the four uniforms of row 4*lq + c at the quad's four columns, one round per slot), a 4x4
    //     transpose inside the quad (DPP), -log(-log u) - parked in LDS for the lower-half wave (tests inject U instead);
    //   lower-half wave (kh == 0): finishes the PREVIOUS tile's logits (lower + upper + bias, stored raw) and feeds them,
    //     with that noise, to the running partial of their row: one (row, column) per slot group, straight-line code.
    const cic_logit_epi_rows er = epi.blk[blk2 ? 1 : 0];
    float* noisebuf = pairbuf + 2 * 4 * (KH - 1) * 4 * 64;  // [2 parities][4 row tiles][4 registers][64 lanes]
    RowPart rp[4];
    int cons[4];
    float gprev[4] = {0.f, 0.f, 0.f, 0.f};
    PhiloxState phs;
    float uq[4] = {0.5f, 0.5f, 0.5f, 0.5f}, u4[4] = {0.5f, 0.5f, 0.5f, 0.5f};
#pragma unroll
    for (int v = 0; v < 4; ++v) { rp[v].init(); cons[v] = -1; }
    phs.init(0, 0);
    if (EPI && kh == 0 && er.cons_seq) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int mm = m0 + 16 * rt + 4 * lq + v;
            cons[v] = er.cons_seq[(size_t)(mm < Mloc ? mm : Mloc - 1) * er.cons_ld + er.cons_col];
        }
    }
    // ROLE: 0 = no epilogue (every wave; the lower halves store their rows); with EPI: 1 = lower-half wave (MODE = its rows'
    // sampling mode), 2 / 3 = upper-half wave making the noise from the Philox stream / from injected uniforms, 4 = an upper
    // wave with nothing to add.  One straight-line loop body per role: no wave-uniform branch inside the MFMA stream.
    auto walk = [&](auto role_, auto mode_) {
        constexpr int ROLE = decltype(role_)::value, MODE = decltype(mode_)::value;
        auto noise_slot = [&](int sl, int n) {             // slot sl of 16, tile column n of this lane
            if (ROLE == 3) {                               // injected uniforms (tests)
                if (sl == 0) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int mm = m0 + 16 * rt + 4 * lq + v;
                        uq[v] = er.U[(size_t)(mm < Mloc ? mm : Mloc - 1) * er.ldu + (n < g.N ? n : g.N - 1)];
                    }
                }
            } else if (sl == 0) {
                const int c = li & 3;
                const int qrow = m0 + 16 * rt + 4 * lq + c;
                phs.init((er.elem0 + (uint64_t)qrow * (uint64_t)er.ldu + (uint64_t)(n - c)) >> 2, er.seed);   // the quad's 4 columns
            } else if (sl <= 5) {
                phs.round();
                phs.round();
            } else if (sl <= 9) {
                // 4x4 transpose inside the quad, one destination register per slot: lane c takes element c of quad lane V
                const int c = li & 3;
                if (sl == 6) { u4[0] = u32_to_unit(phs.c0); u4[1] = u32_to_unit(phs.c1); u4[2] = u32_to_unit(phs.c2); u4[3] = u32_to_unit(phs.c3); }
#define QB(x, V) dpp_f32<(V) * 0x55>(x)                        /* quad_perm [V,V,V,V]: lane V of the quad to all four */
#define PICK(V) { const float b0 = QB(u4[0], V), b1 = QB(u4[1], V), b2 = QB(u4[2], V), b3 = QB(u4[3], V);          \
                  uq[V] = c == 0 ? b0 : (c == 1 ? b1 : (c == 2 ? b2 : b3)); }
                if (sl == 6) PICK(0)
                if (sl == 7) PICK(1)
                if (sl == 8) PICK(2)
                if (sl == 9) PICK(3)
#undef PICK
#undef QB
            }
            if (sl >= 10 && sl <= 13) noisebuf[((buf * 4 + rt) * 4 + (sl - 10)) * 64 + lane] = gumbel_from_u(uq[sl - 10]);
        };
        auto finish_row = [&](int v) {                     // lower-half waves: output register v of the previous tile
            const int mm = m0 + 16 * rt + 4 * lq + v;
            if ((ROLE == 1 || kh == 0) && prev_n >= 0) {
                float x = prev[v];
#pragma unroll
                for (int u = 0; u < KH - 1; ++u) x += pairbuf[((((buf ^ 1) * 4 + rt) * (KH - 1) + u) * 4 + v) * 64 + lane];
                x += prev_bias;
                if (mm < Mloc && prev_n < g.N) {
                    gC[(size_t)mm * g.ldc + prev_n] = x;
                    if (ROLE == 1) {
                        const float xe = prev_n == cons[v] ? -INFINITY : x;      // decoding constraint, AttModel.py:438-442
                        rowpart_add_m<MODE>(rp[v], er.inv_temp, xe, gprev[v], prev_n);
                    }
                }
            }
        };
        auto load_noise = [&]() {
            if (er.noise && prev_n >= 0) {
#pragma unroll
                for (int v = 0; v < 4; ++v) gprev[v] = noisebuf[(((buf ^ 1) * 4 + rt) * 4 + v) * 64 + lane];
            }
        };
#pragma unroll 1
        for (int t = first; t < tiles_n; t += walkers) {
            load_tile(t + walkers);                        // next tile's rows in flight under this tile's MFMAs
            const int n = t * 16 + li;
            const int ncl = n < g.N ? n : g.N - 1;
            float bias_v = 0.f;
            if (g.bias) bias_v = g.bias[ncl];
            const __bf16* bt = ldsh + buf * TILEH + li * LDBH + (K / KH) * kh + 8 * lq;
            __bf16* nxt = ldsh + (buf ^ 1) * TILEH;        // every wave left this buffer before the last barrier
            f32x4acc aS = {0.f, 0.f, 0.f, 0.f}, aM = {0.f, 0.f, 0.f, 0.f}, aB = {0.f, 0.f, 0.f, 0.f};
            bf16x8 bq[2][NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) bq[0][p] = *reinterpret_cast<const bf16x8*>(bt + p * PART);
#pragma unroll
            for (int j = 0; j < JS; ++j) {
                if (j + 1 < JS) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) bq[(j + 1) & 1][p] = *reinterpret_cast<const bf16x8*>(bt + p * PART + 32 * (j + 1));
                }
                __builtin_amdgcn_sched_barrier(0);
                if (NP == 3) {
                    const bf16x8 b0 = bq[j & 1][0], b1 = bq[j & 1][NP > 1 ? 1 : 0], b2 = bq[j & 1][NP > 2 ? 2 : 0];
                    aS = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0][j], b2, aS, 0, 0, 0);
                    aM = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0][j], b1, aM, 0, 0, 0);
                    aB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0][j], b0, aB, 0, 0, 0);
                    aS = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[NP > 2 ? 2 : 0][j], b0, aS, 0, 0, 0);
                    aM = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[NP > 1 ? 1 : 0][j], b0, aM, 0, 0, 0);
                    aS = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[NP > 1 ? 1 : 0][j], b1, aS, 0, 0, 0);
                } else {
                    aB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[0][j], bq[j & 1][0], aB, 0, 0, 0);
                }
                if (ROLE == 0) {
                    if (j < 4) { finish_row(j); __builtin_amdgcn_sched_barrier(0); }
                } else if (ROLE != 4) {
                    // 16 slots spread over the JS k-steps of the tile
#pragma unroll
                    for (int sl = j * 16 / JS; sl < (j + 1) * 16 / JS; ++sl) {
                        if (ROLE == 1) {
                            if (sl == 0) load_noise();
                            if ((sl & 3) == 1) finish_row(sl >> 2);
                        } else {
                            noise_slot(sl, n);
                        }
                    }
                    if (ROLE == 1) __builtin_amdgcn_sched_barrier(0);   // the lower wave's slots stay between the k-steps
                }
                if (j >= JS - NPC) { store_piece(nxt, j - (JS - NPC)); __builtin_amdgcn_sched_barrier(0); }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) prev[v] = (aS[v] + aM[v]) + aB[v];
            if (kh >= 1) {
#pragma unroll
                for (int v = 0; v < 4; ++v) pairbuf[(((buf * 4 + rt) * (KH - 1) + (kh - 1)) * 4 + v) * 64 + lane] = prev[v];
            }
            prev_n = n;
            prev_bias = bias_v;
            if (stamps) { asm volatile("" :: "v"(prev[0]), "v"(prev[3])); stamp(); }
            __syncthreads();
            stamp();
            buf ^= 1;
        }
        if (ROLE == 1) load_noise();
        if (ROLE == 0 || ROLE == 1) {
#pragma unroll
            for (int v = 0; v < 4; ++v) finish_row(v);
        }
    };
    using std::integral_constant;
    if (!EPI) {
        walk(integral_constant<int, 0>{}, integral_constant<int, 0>{});
    } else if (kh == 0) {
        switch (er.mode) {
            case CIC_SAMPLE_NONE: walk(integral_constant<int, 1>{}, integral_constant<int, CIC_SAMPLE_NONE>{}); break;
            case CIC_SAMPLE_GREEDY: walk(integral_constant<int, 1>{}, integral_constant<int, CIC_SAMPLE_GREEDY>{}); break;
            case CIC_SAMPLE_GUMBEL_ST: walk(integral_constant<int, 1>{}, integral_constant<int, CIC_SAMPLE_GUMBEL_ST>{}); break;
            case CIC_SAMPLE_MULTINOMIAL_ST: walk(integral_constant<int, 1>{}, integral_constant<int, CIC_SAMPLE_MULTINOMIAL_ST>{}); break;
            default: walk(integral_constant<int, 1>{}, integral_constant<int, CIC_SAMPLE_MULTINOMIAL>{}); break;   // + TEACHER
        }
    } else if (kh == 1 && er.noise) {
        if (er.U) walk(integral_constant<int, 3>{}, integral_constant<int, 0>{});
        else walk(integral_constant<int, 2>{}, integral_constant<int, 0>{});
    } else {
        walk(integral_constant<int, 4>{}, integral_constant<int, 0>{});
    }
    if (EPI && kh == 0) {
        // the 16 lanes of a row group (same lq: one DPP row) hold the 16 columns of every tile: all-reduce them with DPP
        // lane exchanges (xor 1, xor 2, mirror within 8, mirror within 16 - no LDS crossbar), lane li == 0 writes
#define RP_STEP(CTRL)                                                                                              \
    {                                                                                                              \
        RowPart q;                                                                                                 \
        q.m1 = dpp_f32<CTRL>(rp[v].m1); q.s1 = dpp_f32<CTRL>(rp[v].s1);                                              \
        q.kbest = dpp_f32<CTRL>(rp[v].kbest); q.xbest = dpp_f32<CTRL>(rp[v].xbest);                                  \
        q.kidx = __float_as_int(dpp_f32<CTRL>(__int_as_float(rp[v].kidx))); q.s2 = dpp_f32<CTRL>(rp[v].s2);         \
        rowpart_merge(rp[v], er.mode, er.inv_temp, q);                                                             \
    }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            RP_STEP(DPP_QUAD_XOR1) RP_STEP(DPP_QUAD_XOR2) RP_STEP(DPP_ROW_HALF_MIRROR) RP_STEP(DPP_ROW_MIRROR)
            const int mm = m0 + 16 * rt + 4 * lq + v;
            if (li == 0 && mm < Mloc && er.part) {
                const size_t plane = (size_t)er.part_rows * walkers;
                float* pp = er.part + (size_t)mm * walkers + first;
                pp[0] = rp[v].m1; pp[plane] = rp[v].s1; pp[2 * plane] = rp[v].kbest; pp[3 * plane] = rp[v].xbest;
                pp[4 * plane] = __int_as_float(rp[v].kidx); pp[5 * plane] = rp[v].s2;
            }
        }
#undef RP_STEP
    }
    if (stamps && lane == 0) stamps[63] = __builtin_amdgcn_s_memrealtime();
}

template <int NP>
__global__ __launch_bounds__(256) void split_bf16_kernel(const f32x4* __restrict__ x, int64_t n4, bf16x4* __restrict__ parts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    bf16x4 p[NP];
    split_bf16<NP>(x[i], p);
#pragma unroll
    for (int q = 0; q < NP; ++q) parts[(int64_t)q * n4 + i] = p[q];
}

bool ldsb_walk_ok(const cic_gemm_args& g) {
    return g.a_kc && g.b_kc && g.K2 == 0 && g.K == 512 && g.N >= 2048 && !g.accumulate && !g.relu &&
           (g.lda & 3) == 0 && (g.ldb & 3) == 0 && aligned16(g.A) && aligned16(g.B) && (g.rows_blk == 0 || aligned16(g.A_b));
}

int launch_ldsb_walk(const cic_gemm_args& g, hipStream_t st) {
    constexpr int NG = 32;
    constexpr size_t lds_bytes = 2 * 16 * (16 * NG + 4) * sizeof(float);
    static DeviceOnce attr_set;
    if (attr_set.first()) {
        CIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ldsb_walk_kernel<NG>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    }
    const int Mtot = g.M;
    // row groups of 64; with row blocks each block is rounded up on its own (groups never straddle the blocks)
    int row_groups;
    if (g.rows_blk > 0) row_groups = cic_cdiv(g.rows_blk, 64) + cic_cdiv(Mtot - g.rows_blk, 64);
    else row_groups = cic_cdiv(Mtot, 64);
    const int tiles_n = cic_cdiv(g.N, 16);
    int walkers = 256 / row_groups;                        // one workgroup per CU
    if (walkers < 1) walkers = 1;
    if (walkers > tiles_n) walkers = tiles_n;
    int grid = row_groups * walkers;
    if (g_ldsb2 == 2 || g_ldsb2 == 4) {
        // staged tiles + the K-half hand-off + the noise hand-off of the fused epilogue
        const size_t lds2_bytes = lds_bytes + 2 * 4 * (g_ldsb2 - 1) * 4 * 64 * sizeof(float) + 2 * 4 * 4 * 64 * sizeof(float);
        static DeviceOnce attr2_set;
        if (attr2_set.first()) {
            CIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ldsb2_walk_kernel<NG, 2, 0>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_bytes + 2 * 4 * 1 * 4 * 64 * 4 + 8192)));
            CIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ldsb2_walk_kernel<NG, 4, 0>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_bytes + 2 * 4 * 3 * 4 * 64 * 4 + 8192)));
            CIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ldsb2_walk_kernel<NG, 2, 1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_bytes + 2 * 4 * 1 * 4 * 64 * 4 + 8192)));
            CIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ldsb2_walk_kernel<NG, 4, 1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_bytes + 2 * 4 * 3 * 4 * 64 * 4 + 8192)));
        }
        cic_logit_epilogue epi = {};
        if (g.epi) epi = *g.epi;
        if (g_ldsb2 == 2 && g_bfx && g.precision != CIC_PRECISION_F32_MFMA) {
            // bf16-part form: two tiles of NP bf16 images + the two hand-off buffers (NP = 3: f32 results; NP = 1: the
            // reduced-precision variant, CIC_PRECISION_BF16 - B_parts then holds ONE image, cic_round_bf16)
            const __bf16* wparts = reinterpret_cast<const __bf16*>(g.B_parts);
#define WALK_GO(NPv)                                                                                                                 \
    do {                                                                                                                             \
        constexpr size_t bf_bytes = 2 * NPv * 16 * (16 * NG + 8) * 2 + 2 * 4 * 1 * 4 * 64 * sizeof(float) + 2 * 4 * 4 * 64 * sizeof(float); \
        static DeviceOnce attr3_set;                                                                                                 \
        if (attr3_set.first()) {                                                                                                     \
            CIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ldsb2bf_walk_kernel<NG, 0, 0, NPv>),                     \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)bf_bytes));                                 \
            CIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ldsb2bf_walk_kernel<NG, 1, 0, NPv>),                     \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)bf_bytes));                                 \
            CIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ldsb2bf_walk_kernel<NG, 0, 1, NPv>),                     \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)bf_bytes));                                 \
            CIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ldsb2bf_walk_kernel<NG, 1, 1, NPv>),                     \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)bf_bytes));                                 \
        }                                                                                                                            \
        if (wparts && g.epi)                                                                                                         \
            hipLaunchKernelGGL((gemm_ldsb2bf_walk_kernel<NG, 1, 1, NPv>), dim3(grid), dim3(512), bf_bytes, st, g, row_groups, walkers, epi, wparts); \
        else if (wparts)                                                                                                             \
            hipLaunchKernelGGL((gemm_ldsb2bf_walk_kernel<NG, 0, 1, NPv>), dim3(grid), dim3(512), bf_bytes, st, g, row_groups, walkers, epi, wparts); \
        else if (g.epi)                                                                                                              \
            hipLaunchKernelGGL((gemm_ldsb2bf_walk_kernel<NG, 1, 0, NPv>), dim3(grid), dim3(512), bf_bytes, st, g, row_groups, walkers, epi, wparts); \
        else                                                                                                                         \
            hipLaunchKernelGGL((gemm_ldsb2bf_walk_kernel<NG, 0, 0, NPv>), dim3(grid), dim3(512), bf_bytes, st, g, row_groups, walkers, epi, wparts); \
    } while (0)
            if (g.precision == CIC_PRECISION_BF16) WALK_GO(1); else WALK_GO(3);
#undef WALK_GO
            CIC_LAUNCH_CHECK();
            return 0;
        }
        if (g.epi) {
            if (g_ldsb2 == 2)
                hipLaunchKernelGGL((gemm_ldsb2_walk_kernel<NG, 2, 1>), dim3(grid), dim3(512), lds2_bytes, st, g, row_groups, walkers, epi);
            else
                hipLaunchKernelGGL((gemm_ldsb2_walk_kernel<NG, 4, 1>), dim3(grid), dim3(1024), lds2_bytes, st, g, row_groups, walkers, epi);
        } else if (g_ldsb2 == 2)
            hipLaunchKernelGGL((gemm_ldsb2_walk_kernel<NG, 2, 0>), dim3(grid), dim3(512), lds2_bytes, st, g, row_groups, walkers, epi);
        else
            hipLaunchKernelGGL((gemm_ldsb2_walk_kernel<NG, 4, 0>), dim3(grid), dim3(1024), lds2_bytes, st, g, row_groups, walkers, epi);
        CIC_LAUNCH_CHECK();
        return 0;
    }
    CIC_REQUIRE(!g.epi);
    hipLaunchKernelGGL((gemm_ldsb_walk_kernel<NG>), dim3(grid), dim3(256), lds_bytes, st, g, row_groups, walkers);
    CIC_LAUNCH_CHECK();
    return 0;
}

bool rega_ok(const cic_gemm_args& g) {
    if (!g.a_kc || g.M > (g.rows_blk > 0 ? 256 : 128)) return false;
    if (g.rows_blk > 0 && (!aligned16(g.A_b) || (g.K2 > 0 && !aligned16(g.A2_b)))) return false;
    if ((g.K & 7) || (g.K2 & 7)) return false;
    if (!aligned16(g.A) || (g.lda & 3) || !aligned16(g.B) || (g.ldb & 3)) return false;
    if (g.K2 > 0 && (!aligned16(g.A2) || (g.lda2 & 3) || !aligned16(g.B2) || (g.ldb2 & 3))) return false;
    return true;
}

int launch_rega(const cic_gemm_args& g, hipStream_t st) {
    const int Kt = g.K + g.K2;
    const int grid = cic_cdiv(g.M, 32) * cic_cdiv(g.N, 32);
    const int groups = cic_cdiv(Kt, 8);
    const int gps = cic_cdiv(cic_cdiv(groups, 16), 4) * 4;   // groups per K slice, multiple of the chunk size
    const int strips = cic_cdiv(g.M, 32), tiles_n = cic_cdiv(g.N, 32);
    // Many column tiles (the logit, i2h/h2h and GRU products): strip walkers.  The choice depends on N and K only
    // (never on M) so that a pair of decodes and a single decode sum every output in the same order.
    // (measured, tools/step_gemms.py: at K = 1024 [M x 2560]: 30 vs 36 us, [128 x 3072]: 23 vs 26 us; at K = 512 the
    // 16-wave persistent strips below are faster: 36 vs 42 us for the logit product.  With no B reloads and no
    // cross-wave sum the walkers run their MFMA chains at ~90 % of the f32 rate: what holds these launches at 2x the
    // MFMA floor is operand delivery, 32 cache lines per fragment load, and ~5 us of launch + first-load latency.)
    // 16-wide walkers (see gemm_walk16_kernel) for the K = 1024 products with many column tiles (measured,
    // tools/step_gemms.py: [256 x 2560 x 1024] 22.6 us vs 29.3 us with 32-wide tiles, [128 x 3072 x 1024] 16.6 vs 22.7;
    // at K = 512 the 16-wave persistent strips below stay faster: 35 vs 42-48 us for the logit product)
    if (g_walk16 && g.b_kc && g.N >= 2048 && Kt == 1024 && (g.K % 16) == 0) {
        int nb = 256 / strips;                         // 8-wave workgroups, one per CU
        const int t16 = cic_cdiv(g.N, 16);
        if (nb > t16) nb = t16;
        if (nb >= 8) nb &= ~7;
        if (nb < 1) nb = 1;
        // f32 results either way: three bf16 parts per operand (2.67x the MFMA rate) unless the caller asks for the
        // f32-input instruction
        if (g_bfx && g.precision == CIC_PRECISION_BF16 && (g.K % 32) == 0)
            hipLaunchKernelGGL((gemm_walk16bf_kernel<8, 1>), dim3(strips * nb), dim3(512), 0, st, g, strips);
        else if (g_bfx && g.precision != CIC_PRECISION_F32_MFMA && (g.K % 32) == 0)
            hipLaunchKernelGGL((gemm_walk16bf_kernel<8, 3>), dim3(strips * nb), dim3(512), 0, st, g, strips);
        else
            hipLaunchKernelGGL((gemm_walk16_kernel<8, 8, false>), dim3(strips * nb), dim3(512), 0, st, g, strips);
        CIC_LAUNCH_CHECK();
        return 0;
    }
    if (g_walk && g.b_kc && tiles_n >= 64 && Kt == 1024) {
        int nb = 256 / strips;                         // 8-wave workgroups, one per CU
        if (nb > tiles_n) nb = tiles_n;
        if (nb >= 8) nb &= ~7;
        if (nb < 1) nb = 1;
        hipLaunchKernelGGL((gemm_walk_kernel<16, 8, true>), dim3(strips * nb), dim3(512), 0, st, g, strips);
        CIC_LAUNCH_CHECK();
        return 0;
    }
    if (gps <= 4 && strips * tiles_n > 512) {
        // persistent strips: one workgroup per CU, each walking several column tiles
        int nb = 256 / strips;
        if (nb < 1) nb = 1;
        if (nb > tiles_n) nb = tiles_n;
        if (nb >= 8) nb &= ~7;            // walkers per strip: a multiple of the XCD count (see the kernel)
        dim3 lgrid(strips * nb), blk(1024);
        if (g.b_kc) hipLaunchKernelGGL((gemm_rega_loop_kernel<4, 16, true>), lgrid, blk, 0, st, g, strips);
        else hipLaunchKernelGGL((gemm_rega_loop_kernel<4, 16, false>), lgrid, blk, 0, st, g, strips);
        CIC_LAUNCH_CHECK();
        return 0;
    }
    // Few output tiles and a long K (dX = dY W of the BPTT loops: 64..128 tiles, K = 1024..3072): split K over
    // `ky` workgroups per tile so that the launch covers the chip; the partial tiles are summed with float atomics,
    // which the caller allows for gradient products (sum_order_free) whose C accumulates or is already zero
    // (c_is_zero: the caller or the producing kernel cleared it).
    int ky = 1, gps_y = gps;
    // dX products of the BPTT loops (K-strided weights, gradient sums): 64-row tiles, the B fragments shared by two strips
    if (g_rega2 && g_tail_split && !g.b_kc && g.sum_order_free && !g.relu && (g.accumulate || g.c_is_zero) && groups >= 128 &&
        (g.rows_blk % 64) == 0 && g.M >= 64) {
        const int strips64 = g.rows_blk > 0 ? g.rows_blk / 64 + cic_cdiv(g.M - g.rows_blk, 64) : cic_cdiv(g.M, 64);
        const int grid2 = strips64 * tiles_n;
        int ky2 = 256 / grid2;
        if (ky2 > groups / 32) ky2 = groups / 32;      // at least 2 groups (16 k) per wave
        if (ky2 > 16) ky2 = 16;
        if (grid2 <= 128 && ky2 >= 2) {
            const int per = cic_cdiv(groups, 16 * ky2);          // groups per wave
            const int ch = (per % 3 == 0 && per % 2 != 0) ? 3 : 2;
            const int gps2 = cic_cdiv(per, ch) * ch;
            constexpr size_t lds_bytes = 16 * 32 * 64 * sizeof(float);
            static DeviceOnce attr_set;
            if (attr_set.first()) {
                CIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rega2_kernel<16, 2>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                CIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rega2_kernel<16, 3>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            }
            if (ch == 3) hipLaunchKernelGGL((gemm_rega2_kernel<16, 3>), dim3(grid2, ky2), dim3(1024), lds_bytes, st, g, gps2);
            else hipLaunchKernelGGL((gemm_rega2_kernel<16, 2>), dim3(grid2, ky2), dim3(1024), lds_bytes, st, g, gps2);
            CIC_LAUNCH_CHECK();
            return 0;
        }
    }
    if (g_tail_split && g.sum_order_free && !g.relu && (g.accumulate || g.c_is_zero) && grid <= 128 && groups >= 128) {
        ky = 256 / grid;
        if (ky > groups / 64) ky = groups / 64;        // at least 4 groups (32 k) per wave
        if (ky > 8) ky = 8;
        if (ky >= 2) gps_y = cic_cdiv(cic_cdiv(groups, 16 * ky), 4) * 4; else ky = 1;
    }
    if (g.b_kc) hipLaunchKernelGGL((gemm_rega_kernel<16, true>), dim3(grid, ky), dim3(1024), 0, st, g, gps_y);
    else hipLaunchKernelGGL((gemm_rega_kernel<16, false>), dim3(grid, ky), dim3(1024), 0, st, g, gps_y);
    CIC_LAUNCH_CHECK();
    return 0;
}


template <int BM, int BN, int WM, int WN>
int launch_shape(const cic_gemm_args& g, bool vec, bool want_tail, hipStream_t st) {
    const int tiles = cic_cdiv(g.M, BM) * cic_cdiv(g.N, BN);
    constexpr int CUS = 256;
    int full = tiles, ks = 1;
    const int nk = cic_cdiv(g.K, BK);
    const int tail = tiles % CUS;
    // a tail of fewer than ~3/4 of the CUs is K-sliced (needs a plain sum epilogue: no second operand pair, no ReLU,
    // and a caller that accepts an unordered sum)
    if (want_tail && g_tail_split && g.sum_order_free && tail > 0 && tail <= (CUS * 3) / 4 && g.K2 == 0 && !g.relu && nk >= 8) {
        ks = CUS / tail;
        if (tiles < CUS) ks = (2 * CUS) / tail;     // nothing but a tail: two co-resident workgroups per CU
        if (ks > nk / 4) ks = nk / 4;               // at least 4 K-tiles (128 k) per slice
        if (ks > 32) ks = 32;
        if (ks >= 2) full = tiles - tail; else ks = 1;
    }
    const int grid = full + (tiles - full) * ks;
    if (ks > 1 && !g.accumulate && !g.c_is_zero) {
        // the sliced tiles sum into C: zero the rows x columns they cover (whole C when everything is tail), unless the caller
        // (or the kernel that produced the operands) has cleared C already
        if (full == 0) {
            CIC_HIP(hipMemset2DAsync(g.C, sizeof(float) * g.ldc, 0, sizeof(float) * g.N, g.M, st));
        } else {
            const int tiles_n = cic_cdiv(g.N, BN);
            const int first_row = (full / tiles_n) * BM;      // tail tiles start somewhere in this tile row
            CIC_HIP(hipMemset2DAsync(g.C + (size_t)first_row * g.ldc, sizeof(float) * g.ldc, 0, sizeof(float) * g.N,
                                     g.M - first_row, st));
        }
    }
    dim3 blk(WM * WN * 64);
    // bf16-part products (f32 accuracy at 2.67x the f32 MFMA rate; precision 2: plain bf16) for aligned single-pair
    // products; CIC_PRECISION_F32_MFMA keeps the exact-f32 MFMA kernel
    const int np = (!vec || g.K2 > 0 || !g_bfx || g.precision == CIC_PRECISION_F32_MFMA) ? 0
                   : (g.precision == CIC_PRECISION_BF16 ? 1 : 3);
    if (np) {
#define CIC_BFX_GO(KA, KB)                                                                                                   \
    do {                                                                                                                     \
        if (np == 3) hipLaunchKernelGGL((gemm_bfx_kernel<BM, BN, WM, WN, KA, KB, 3>), dim3(grid), blk, 0, st, g, full, ks);   \
        else hipLaunchKernelGGL((gemm_bfx_kernel<BM, BN, WM, WN, KA, KB, 1>), dim3(grid), blk, 0, st, g, full, ks);           \
    } while (0)
        switch ((g.a_kc ? 2 : 0) | (g.b_kc ? 1 : 0)) {
            case 3: CIC_BFX_GO(true, true); break;
            case 2: CIC_BFX_GO(true, false); break;
            case 1: CIC_BFX_GO(false, true); break;
            default: CIC_BFX_GO(false, false); break;
        }
#undef CIC_BFX_GO
        CIC_LAUNCH_CHECK();
        return 0;
    }
#define CIC_GEMM_GO(KA, KB, V) \
    hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, KA, KB, V>), dim3(grid), blk, 0, st, g, full, ks)
    const int code = (g.a_kc ? 4 : 0) | (g.b_kc ? 2 : 0) | (vec ? 1 : 0);
    switch (code) {
        case 7: CIC_GEMM_GO(true, true, true); break;
        case 6: CIC_GEMM_GO(true, true, false); break;
        case 5: CIC_GEMM_GO(true, false, true); break;
        case 4: CIC_GEMM_GO(true, false, false); break;
        case 3: CIC_GEMM_GO(false, true, true); break;
        case 2: CIC_GEMM_GO(false, true, false); break;
        case 1: CIC_GEMM_GO(false, false, true); break;
        default: CIC_GEMM_GO(false, false, false); break;
    }
#undef CIC_GEMM_GO
    CIC_LAUNCH_CHECK();
    return 0;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// the float4 path needs every 4-group of an operand row to be wholly in or out of range
bool operand_vec_ok(const float* P, int ld, int kc, int rows, int K) {
    if (!P) return true;
    if (!aligned16(P) || (ld & 3)) return false;
    return kc ? (K & 3) == 0 : (rows & 3) == 0;
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int M, int N, int ldx,
                                                     float* __restrict__ out, int accumulate) {
    // block = 64 columns x 4 row-groups; grid.y splits M, partial sums combined with atomics
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const int rows_per = (M + gridDim.y - 1) / gridDim.y;
    const int mb = blockIdx.y * rows_per, me = min(M, mb + rows_per);
    float s = 0.f;
    if (n < N)
        for (int m = mb + rg; m < me; m += 4) s += X[(size_t)m * ldx + n];
    part[rg][c] = s;
    __syncthreads();
    if (rg == 0 && n < N) {
        s = part[0][c] + part[1][c] + part[2][c] + part[3][c];
        if (gridDim.y == 1) {
            out[n] = accumulate ? out[n] + s : s;
        } else {
            atomicAdd(out + n, s);
        }
    }
}

}  // namespace

#ifdef CIC_DEVTOOLS
extern "C" int cic_debug_gemm_tail_split(int on) {
    g_tail_split = on & 0xff;
    g_force_tile = (on >> 8) & 0xff;
    g_walk = ((on >> 16) & 1) ? 0 : 1;
    g_walk16 = ((on >> 21) & 1) ? 0 : 1;
    g_ldsb = ((on >> 22) & 1) ? 0 : 1;
    g_split_rows = ((on >> 23) & 1) ? 0 : 1;
    g_rega2 = ((on >> 24) & 1) ? 0 : 1;
    g_ldsb2 = ((on >> 25) & 3) == 1 ? 0 : (((on >> 25) & 3) == 2 ? 4 : 2);
    g_logit_epi = ((on >> 27) & 1) ? 0 : 1;
    g_bfx = ((on >> 28) & 1) ? 0 : 1;
    return 0;
}

extern "C" int cic_debug_set_bfx_stamps(unsigned long long* buf) {
    CIC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_bfx_stamps), &buf, sizeof(buf)));
    return 0;
}
extern "C" int cic_debug_set_stamps(unsigned long long* buf) {
    CIC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf)));
    return 0;
}
#endif

static bool gemm_split_supported(const cic_gemm_args& g) {
    // the 16-wide strip walker (launch_rega): K-contiguous operands, x and h halves of 512, whole 16-column tiles
    return g.n_split > 0 && g.a_kc && g.b_kc && g.K == 512 && g.K2 == 512 && (g.n_split % 16) == 0 && g.n_split >= 2048 &&
           g.N > g.n_split && g.B2_tail && g.C_tail && (g.ldb2_tail & 3) == 0 && aligned16(g.B2_tail) && !g.accumulate &&
           !g.relu && (g.rows_blk == 0 || g.C_tail_b) && g_walk16 && rega_ok(g);   // K = K2 = 512: the walker's shape
}
extern "C" int cic_gemm_split_ok(const cic_gemm_args* a) { return a && gemm_split_supported(*a) ? 1 : 0; }

// rows 129..256 of K-contiguous activations are handed to the register-streaming kernels as two blocks of 128
static bool split_rows_case(const cic_gemm_args& g) {
    return g.rows_blk == 0 && g_split_rows && g.a_kc && g.M > 128 && g.M <= 256 && !g.colsum_A;
}
static int ldsb_walk_parts(const cic_gemm_args& g) {     // the geometry of launch_ldsb_walk
    const int row_groups = g.rows_blk > 0 ? cic_cdiv(g.rows_blk, 64) + cic_cdiv(g.M - g.rows_blk, 64) : cic_cdiv(g.M, 64);
    const int tiles_n = cic_cdiv(g.N, 16);
    int walkers = 256 / row_groups;
    if (walkers < 1) walkers = 1;
    if (walkers > tiles_n) walkers = tiles_n;
    return walkers;
}
extern "C" int cic_gemm_logit_parts(const cic_gemm_args* a) {
    if (!a || a->n_split > 0 || a->colsum_A || !g_logit_epi || !(g_ldsb && (g_ldsb2 == 2 || g_ldsb2 == 4))) return 0;
    cic_gemm_args g = *a;
    if (split_rows_case(g)) {
        g.rows_blk = 128; g.A_b = g.A + (size_t)128 * g.lda; g.C_b = g.C + (size_t)128 * g.ldc;
        if (!rega_ok(g)) return 0;
    } else if (g.rows_blk > 0 && !rega_ok(g)) {
        return 0;
    }
    if (!ldsb_walk_ok(g) || g.M > 256) return 0;
    return ldsb_walk_parts(g);
}

extern "C" int cic_gemm_f32(const cic_gemm_args* a, cic_stream_t s) {
    CIC_REQUIRE(a != nullptr);
    const cic_gemm_args& g = *a;
    CIC_REQUIRE(!g.epi || cic_gemm_logit_parts(a) > 0);
    if (g.n_split > 0) {
        CIC_REQUIRE(gemm_split_supported(g));
        return launch_rega(g, cic_s(s));
    }
    CIC_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0 && g.K2 >= 0);
    CIC_REQUIRE(g.A && g.B && g.C);
    CIC_REQUIRE(g.K2 == 0 || (g.A2 && g.B2));
    bool vec = operand_vec_ok(g.A, g.lda, g.a_kc, g.M, g.K) && operand_vec_ok(g.B, g.ldb, g.b_kc, g.N, g.K);
    if (g.K2 > 0)
        vec = vec && operand_vec_ok(g.A2, g.lda2, g.a_kc, g.M, g.K2) && operand_vec_ok(g.B2, g.ldb2, g.b_kc, g.N, g.K2);
    const int64_t big_tiles = (int64_t)cic_cdiv(g.M, 128) * cic_cdiv(g.N, 128);
    if (g.colsum_A) {
        CIC_REQUIRE(!g.a_kc && g.K2 == 0);      // the column sums of a [K, M] operand (dW = dY^T X)
        if (g.M <= 128) {                       // the skinny kernels do not carry the by-product: a launch of its own
            cic_gemm_args g2 = g;
            g2.colsum_A = nullptr; g2.colsum_A2 = nullptr;
            if (int rc = cic_colsum_f32(g.A, g.K, g.M, g.lda, g.colsum_A, 1, s)) return rc;
            if (g.colsum_A2) { if (int rc = cic_colsum_f32(g.A, g.K, g.M, g.lda, g.colsum_A2, 1, s)) return rc; }
            return cic_gemm_f32(&g2, s);
        }
    }
    if (g.rows_blk == 0 && g_split_rows && g.a_kc && g.M > 128 && g.M <= 256 && !g.colsum_A) {
        // 129..256 rows of K-contiguous activations (a per-timestep product of a B = 256 decode): the register-streaming
        // kernels address two row blocks, so hand them the matrix as two halves (same arithmetic per row)
        cic_gemm_args h2 = g;
        h2.rows_blk = 128;
        h2.A_b = g.A + (size_t)128 * g.lda;
        h2.C_b = g.C + (size_t)128 * g.ldc;
        if (g.K2 > 0) h2.A2_b = g.A2 + (size_t)128 * g.lda2;
        cic_logit_epilogue e2;
        if (g.epi) {                             // the second row block continues the first one's rows
            e2 = *g.epi;
            cic_logit_epi_rows& b = e2.blk[1];
            b = e2.blk[0];
            if (b.U) b.U += (size_t)128 * b.ldu;
            b.elem0 += (uint64_t)128 * (uint64_t)b.ldu;
            if (b.cons_seq) b.cons_seq += (size_t)128 * b.cons_ld;
            if (b.part) b.part += (size_t)128 * ldsb_walk_parts(h2);
            h2.epi = &e2;
        }
        if (rega_ok(h2)) {
            if (g_ldsb && ldsb_walk_ok(h2)) return launch_ldsb_walk(h2, cic_s(s));
            return launch_rega(h2, cic_s(s));
        }
    }
    if (g.rows_blk > 0) {
        CIC_REQUIRE(g.a_kc && (g.rows_blk & 31) == 0 && g.M > g.rows_blk && g.M <= 2 * g.rows_blk && g.A_b && g.C_b);
        CIC_REQUIRE(g.K2 == 0 || g.A2_b);
        CIC_REQUIRE(rega_ok(g));     // only the register-streaming kernels address row blocks
    }
    // the logit product of up to 256 rows (K = 512, many column tiles): weight tiles through LDS, no K split
    // (tools/step_gemms.py: M = 256 as a pair 33.3 us vs 35.0 us, M = 128 20.0 vs 20.5; from M = 384 on the
    // LDS-tiled kernels below win: 43.5 vs 47 us)
    if (g_ldsb && ldsb_walk_ok(g) && g.M <= 256) return launch_ldsb_walk(g, cic_s(s));
    if (rega_ok(g)) return launch_rega(g, cic_s(s));
    if (g.M <= 128) {
        // per-timestep products: in-workgroup split-K (see gemm_skinny_kernel)
        const int ktot = g.K + g.K2;
        if (g.M > 64 && (cic_cdiv(g.N, 32) >= 128 || ktot < 2048)) return launch_skinny<4, 4>(g, vec, cic_s(s));
        return launch_skinny<2, 8>(g, vec, cic_s(s));
    }
    if (g_force_tile == 1) return launch_shape<128, 128, 2, 2>(g, vec, true, cic_s(s));
    if (g_force_tile == 2) return launch_shape<64, 64, 2, 2>(g, vec, true, cic_s(s));
    if (g_force_tile == 3) return launch_shape<128, 128, 2, 4>(g, vec, true, cic_s(s));   // 8 waves, 64x32 per wave
    if (g_force_tile == 4) return launch_shape<128, 128, 4, 2>(g, vec, true, cic_s(s));   // 8 waves, 32x64 per wave
    // Tile choice (measured on the shapes of the B = 128 joint step with the bf16-part kernel, tools/gemm_sweep.py ->
    // profiles/r02_gemm_sweep.log): a 128x128 workgroup of 8 waves (4x2: 32x64 per wave, two waves per SIMD) is the
    // efficient tile; what decides is how evenly the tiles cover 256 CUs.
    //   >= 400 big tiles: 128x128, several rounds of full tiles;
    //   a free summation order (gradient products), >= 48 big tiles and K >= 1024: 128x128 tiles, the tail round (or every
    //     tile when there are fewer than 256) K-sliced so that ~512 workgroups cover the chip - d_out 135 us (64x64
    //     tiles: 266), dW logit 136 (177), dx 56 (76), listener dW hh 95 (120);
    //   otherwise 64x64 tiles, K-sliced only when there are fewer than one per CU.
    const int64_t small_tiles = (int64_t)cic_cdiv(g.M, 64) * cic_cdiv(g.N, 64);
    const bool free_sum = g.sum_order_free && g_tail_split && g.K2 == 0 && !g.relu;
    if (big_tiles >= 400) return launch_shape<128, 128, 4, 2>(g, vec, true, cic_s(s));
    if (free_sum && big_tiles >= 48 && g.K >= 1024) return launch_shape<128, 128, 4, 2>(g, vec, true, cic_s(s));
    return launch_shape<64, 64, 2, 2>(g, vec, small_tiles < 256, cic_s(s));
}

// Average duration of one cic_gemm_f32 launch: `iters` back-to-back launches between two HIP events on `s`.
extern "C" int cic_split_bf16x3(const float* x, int64_t n, uint16_t* parts, cic_stream_t s) {
    CIC_REQUIRE(x && parts && n > 0 && (n & 3) == 0 && aligned16(x) && ((uintptr_t)parts & 7) == 0);
    hipLaunchKernelGGL(split_bf16_kernel<3>, dim3(cic_cdiv(n / 4, 256)), dim3(256), 0, cic_s(s), reinterpret_cast<const f32x4*>(x),
                       n / 4, reinterpret_cast<bf16x4*>(parts));
    CIC_LAUNCH_CHECK();
    return 0;
}
// the one-part form: packed[i] = bf16(x[i]) (round to nearest even) - the operand image of the reduced-precision variant
extern "C" int cic_round_bf16(const float* x, int64_t n, uint16_t* packed, cic_stream_t s) {
    CIC_REQUIRE(x && packed && n > 0 && (n & 3) == 0 && aligned16(x) && ((uintptr_t)packed & 7) == 0);
    hipLaunchKernelGGL(split_bf16_kernel<1>, dim3(cic_cdiv(n / 4, 256)), dim3(256), 0, cic_s(s), reinterpret_cast<const f32x4*>(x),
                       n / 4, reinterpret_cast<bf16x4*>(packed));
    CIC_LAUNCH_CHECK();
    return 0;
}

extern "C" int cic_gemm_f32_timed(const cic_gemm_args* a, int iters, double* avg_us, cic_stream_t s) {
    CIC_REQUIRE(a && iters > 0 && avg_us);
    hipEvent_t e0, e1;
    CIC_HIP(hipEventCreate(&e0));
    CIC_HIP(hipEventCreate(&e1));
    int rc = 0;
    for (int i = 0; i < 3 && !rc; ++i) rc = cic_gemm_f32(a, s);
    CIC_HIP(hipEventRecord(e0, cic_s(s)));
    for (int i = 0; i < iters && !rc; ++i) rc = cic_gemm_f32(a, s);
    CIC_HIP(hipEventRecord(e1, cic_s(s)));
    CIC_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    CIC_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_us = (double)ms * 1e3 / iters;
    return rc;
}

extern "C" int cic_colsum_f32(const float* X, int M, int N, int ldx, float* out, int accumulate,
                              cic_stream_t s) {
    CIC_REQUIRE(X && out && M > 0 && N > 0 && ldx >= N);
    int ysplit = 1;
    if (M >= 2048) ysplit = 8;
    if (ysplit > 1 && !accumulate) CIC_HIP(hipMemsetAsync(out, 0, sizeof(float) * N, cic_s(s)));
    hipLaunchKernelGGL(colsum_kernel, dim3(cic_cdiv(N, 64), ysplit), dim3(256), 0, cic_s(s), X, M, N, ldx, out,
                       accumulate);
    CIC_LAUNCH_CHECK();
    return 0;
}
