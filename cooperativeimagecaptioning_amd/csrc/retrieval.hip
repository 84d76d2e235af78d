// Retrieval-rank evaluation of the listener: eval_utils.i2t / t2i (eval_utils.py:545-720), cosine measure.
// The reference scores one query at a time on the host (np.dot + np.argsort per image / caption: O(N^2 log N) over
// 5000 x 25000 similarities).  Here the whole similarity matrix is ONE f32 MFMA product S = ims cap^T and a rank is
// what argsort()[::-1] positions mean: the number of candidates scoring higher than the correct one (equal scores:
// the reversed ascending order puts the larger index first), so no sort is needed at all.
#include "cic_common.h"
#include "engine_util.h"

namespace {

// i2t: one workgroup per image row i of S [N, C] (C = cpi * N captions).  rank = min over the image's own captions
// c in [cpi*i, cpi*i + cpi) of #{j : S[i,j] > S[i,c]  or  (S[i,j] == S[i,c] and j > c)};  top1 = arg max (ties: larger j).
__global__ __launch_bounds__(256) void rank_i2t_kernel(const float* __restrict__ S, int N, int C, int cpi,
                                                       int32_t* __restrict__ ranks, int32_t* __restrict__ top1) {
    __shared__ int cnt[8];
    __shared__ float shv[4];
    __shared__ int shi[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    const float* row = S + (size_t)i * C;
    float tv[8];
    for (int q = 0; q < cpi; ++q) tv[q] = row[cpi * i + q];
    if (tid < 8) cnt[tid] = 0;
    __syncthreads();
    int loc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float bv = -INFINITY;
    int bi = -1;
    for (int j = tid; j < C; j += 256) {
        const float v = row[j];
        if (v > bv || (v == bv && j > bi)) { bv = v; bi = j; }
        for (int q = 0; q < cpi; ++q) {
            const int c = cpi * i + q;
            loc[q] += (v > tv[q] || (v == tv[q] && j > c)) ? 1 : 0;
        }
    }
    for (int q = 0; q < cpi; ++q) {
        int v = loc[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((tid & 63) == 0) atomicAdd(&cnt[q], v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi > bi)) { bv = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { shv[tid >> 6] = bv; shi[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        int r = cnt[0];
        for (int q = 1; q < cpi; ++q) r = min(r, cnt[q]);
        ranks[i] = r;
        for (int w = 1; w < 4; ++w)
            if (shv[w] > bv || (shv[w] == bv && shi[w] > bi)) { bv = shv[w]; bi = shi[w]; }
        top1[i] = bi;
    }
}

// t2i: one thread per caption column c of S [N, C]; its image is c / cpi.  Adjacent threads read adjacent columns,
// so every row step is one coalesced read.
__global__ __launch_bounds__(256) void rank_t2i_kernel(const float* __restrict__ S, int N, int C, int cpi,
                                                       int32_t* __restrict__ ranks, int32_t* __restrict__ top1) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int own = c / cpi;
    const float tv = S[(size_t)own * C + c];
    int cnt = 0, bi = -1;
    float bv = -INFINITY;
    for (int m = 0; m < N; ++m) {
        const float v = S[(size_t)m * C + c];
        cnt += (v > tv || (v == tv && m > own)) ? 1 : 0;
        if (v > bv || (v == bv && m > bi)) { bv = v; bi = m; }
    }
    ranks[c] = cnt;
    top1[c] = bi;
}

}  // namespace

extern "C" size_t cic_retrieval_ws_bytes(int n_images, int cpi) {
    if (n_images <= 0 || cpi <= 0) return 0;
    return sizeof(float) * (size_t)n_images * n_images * cpi + 256;
}

extern "C" int cic_retrieval_ranks(const float* ims, const float* caps, int n_images, int cpi, int J, int32_t* ranks_i2t,
                                   int32_t* top1_i2t, int32_t* ranks_t2i, int32_t* top1_t2i, void* ws, size_t ws_bytes,
                                   cic_stream_t s) {
    CIC_REQUIRE(ims && caps && ws && n_images > 0 && cpi >= 1 && cpi <= 8 && J > 0);
    CIC_REQUIRE(ws_bytes >= cic_retrieval_ws_bytes(n_images, cpi));
    const int N = n_images, C = n_images * cpi;
    float* S = static_cast<float*>(ws);
    hipStream_t st = cic_s(s);
    // S[m, c] = <ims[m], caps[c]>                                                 (np.dot, eval_utils.py:573,647)
    if (int rc = gemm_nt(ims, J, caps, J, S, C, N, C, J, nullptr, false, false, st)) return rc;
    if (ranks_i2t) {
        CIC_REQUIRE(top1_i2t);
        hipLaunchKernelGGL(rank_i2t_kernel, dim3(N), dim3(256), 0, st, S, N, C, cpi, ranks_i2t, top1_i2t);
        CIC_LAUNCH_CHECK();
    }
    if (ranks_t2i) {
        CIC_REQUIRE(top1_t2i);
        hipLaunchKernelGGL(rank_t2i_kernel, dim3(cic_cdiv(C, 256)), dim3(256), 0, st, S, N, C, cpi, ranks_t2i, top1_t2i);
        CIC_LAUNCH_CHECK();
    }
    return 0;
}
