// use_bn = 1: nn.BatchNorm1d(att_feat_size) in front of the region embedding (models/AttModel.py:82-85), applied by
// pack_wrapper (:44-51) to the PACKED valid region rows [N, D] of a batch with ragged region counts.
//
// MI355X form: the normalisation is a per-feature affine map y = a x + b (a = gamma / sqrt(var + eps), b = beta - mean a) in
// front of a Linear, so it is FOLDED into that Linear - W' = W diag(a), bias' = bias + W b - and the [B*K, D] x [D, H] product
// reads the raw features as before: no normalised copy of the 37 MB feature tensor is ever written.  The backward pass needs
// nothing but the Linear's raw weight gradient dWraw = d_pre^T x and db = colsum(d_pre), which the engine computes anyway:
//     G     = rstd (dWraw - mean (x) db)                (= d_pre^T x_hat)
//     dW   += gamma G + beta (x) db                     (= d_pre^T y)
//     dgamma_j += sum_h W_hj G_hj,   dbeta_j += sum_h W_hj db_h
// (the statistics depend on the input features only, which take no gradient).
#include "cic_common.h"

namespace {

// per feature j: mean and BIASED variance over the valid rows (two passes: sum, then squared deviations).
// One workgroup = 64 features x 4 row groups; a wave reads 64 consecutive floats of a row (256 B).
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, const float* __restrict__ masks, int rows, int D,
                                                       float* __restrict__ mean, float* __restrict__ var, float* __restrict__ count) {
    __shared__ float sh[4][64];
    __shared__ float cnt_s[4];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + c;
    float s = 0.f, n = 0.f;
    for (int r = rg; r < rows; r += 4) {
        const bool on = masks[r] > 0.f;
        if (on) { n += 1.f; if (j < D) s += x[(size_t)r * D + j]; }
    }
    sh[rg][c] = s;
    if (c == 0) cnt_s[rg] = n;
    __syncthreads();
    const float N = cnt_s[0] + cnt_s[1] + cnt_s[2] + cnt_s[3];
    const float mu = (sh[0][c] + sh[1][c] + sh[2][c] + sh[3][c]) / N;
    __syncthreads();
    float q = 0.f;
    for (int r = rg; r < rows; r += 4)
        if (masks[r] > 0.f && j < D) { const float dlt = x[(size_t)r * D + j] - mu; q += dlt * dlt; }
    sh[rg][c] = q;
    __syncthreads();
    if (rg == 0 && j < D) {
        mean[j] = mu;
        var[j] = (sh[0][c] + sh[1][c] + sh[2][c] + sh[3][c]) / N;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *count = N;
}

// running statistics of one training-mode forward: r <- (1 - m) r + m batch (variance: the UNBIASED one, N / (N - 1))
__global__ __launch_bounds__(256) void bn_running_kernel(const float* __restrict__ mean, const float* __restrict__ var,
                                                         const float* __restrict__ count, float momentum, int D,
                                                         float* __restrict__ rmean, float* __restrict__ rvar) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= D) return;
    const float N = *count;
    const float unb = N > 1.f ? var[j] * (N / (N - 1.f)) : var[j];
    rmean[j] = (1.f - momentum) * rmean[j] + momentum * mean[j];
    rvar[j] = (1.f - momentum) * rvar[j] + momentum * unb;
}

// one workgroup per output unit h: W'[h][j] = W[h][j] a_j, bias'[h] = bias[h] + sum_j W[h][j] b_j
__global__ __launch_bounds__(256) void bn_fold_fwd_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                                          int D, float* __restrict__ Wf, float* __restrict__ biasf) {
    __shared__ float sh[4];
    const int h = blockIdx.x;
    float dot = 0.f;
    for (int j = threadIdx.x; j < D; j += 256) {
        const float a = gamma[j] / sqrtf(var[j] + eps);
        const float b = beta[j] - mean[j] * a;
        const float w = W[(size_t)h * D + j];
        Wf[(size_t)h * D + j] = w * a;
        dot += w * b;
    }
    dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) biasf[h] = bias[h] + ((sh[0] + sh[1]) + (sh[2] + sh[3]));
}

// one thread per feature j walks the H rows (coalesced across j)
__global__ __launch_bounds__(256) void bn_fold_bwd_kernel(const float* __restrict__ dWraw, const float* __restrict__ dbraw,
                                                          const float* __restrict__ W, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ mean,
                                                          const float* __restrict__ var, float eps, int H, int D,
                                                          float* __restrict__ dW, float* __restrict__ dbias,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < D) {
        const float rstd = 1.0f / sqrtf(var[j] + eps), mu = mean[j], ga = gamma[j], be = beta[j];
        float dg = 0.f, db = 0.f;
        for (int h = 0; h < H; ++h) {
            const float dbh = dbraw[h];
            const float G = rstd * (dWraw[(size_t)h * D + j] - mu * dbh);
            const float w = W[(size_t)h * D + j];
            dW[(size_t)h * D + j] += ga * G + be * dbh;
            dg += w * G;
            db += w * dbh;
        }
        if (dgamma) dgamma[j] += dg;
        if (dbeta) dbeta[j] += db;
    }
    if (blockIdx.x == 0)
        for (int h = threadIdx.x; h < H; h += blockDim.x) dbias[h] += dbraw[h];
}

}  // namespace

extern "C" int cic_bn_stats(const float* x, const float* masks, int rows, int D, float* mean, float* var, float* count,
                            cic_stream_t s) {
    CIC_REQUIRE(x && masks && mean && var && count && rows > 0 && D > 0);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(cic_cdiv(D, 64)), dim3(256), 0, cic_s(s), x, masks, rows, D, mean, var, count);
    CIC_LAUNCH_CHECK();
    return 0;
}

extern "C" int cic_bn_running_update(const float* mean, const float* var, const float* count, float momentum, int D,
                                     float* running_mean, float* running_var, cic_stream_t s) {
    CIC_REQUIRE(mean && var && count && running_mean && running_var && D > 0 && momentum >= 0.f && momentum <= 1.f);
    hipLaunchKernelGGL(bn_running_kernel, dim3(cic_cdiv(D, 256)), dim3(256), 0, cic_s(s), mean, var, count, momentum, D,
                       running_mean, running_var);
    CIC_LAUNCH_CHECK();
    return 0;
}

extern "C" int cic_bn_fold_fwd(const float* W, const float* bias, const float* gamma, const float* beta, const float* mean,
                               const float* var, float eps, int H, int D, float* W_folded, float* bias_folded, cic_stream_t s) {
    CIC_REQUIRE(W && bias && gamma && beta && mean && var && W_folded && bias_folded && H > 0 && D > 0 && eps > 0.f);
    hipLaunchKernelGGL(bn_fold_fwd_kernel, dim3(H), dim3(256), 0, cic_s(s), W, bias, gamma, beta, mean, var, eps, D, W_folded,
                       bias_folded);
    CIC_LAUNCH_CHECK();
    return 0;
}

extern "C" int cic_bn_fold_bwd(const float* dW_raw, const float* db_raw, const float* W, const float* gamma, const float* beta,
                               const float* mean, const float* var, float eps, int H, int D, float* dW, float* dbias,
                               float* dgamma, float* dbeta, cic_stream_t s) {
    CIC_REQUIRE(dW_raw && db_raw && W && gamma && beta && mean && var && dW && dbias && H > 0 && D > 0 && eps > 0.f);
    hipLaunchKernelGGL(bn_fold_bwd_kernel, dim3(cic_cdiv(D, 256)), dim3(256), 0, cic_s(s), dW_raw, db_raw, W, gamma, beta, mean,
                       var, eps, H, D, dW, dbias, dgamma, dbeta);
    CIC_LAUNCH_CHECK();
    return 0;
}
