// trlda_amd/csrc/elbo_kernels.h -- the variational lower bound after an E-step
// (reference code/trlda/src/lda.cpp:297-360), the step after the hot path.
//
// Two kernels over data the E-step left on the device (lambda, psiSum, gamma, sstats):
//   elbo_dense_kernel   sum_kw (eta + f sstats - lambda)(psi(lambda) - psiSum_k)   (:317)
//                       and sum_kw lgamma(lambda)                                     (:357)
//   elbo_docs_kernel    per document the terms of :325-351
// Per-block / per-document partial results are added up on the host in index order, so the
// bound is reproducible run to run.
//
// The reference recomputes phi as softmax_k(psi(lambda_kw) - psiSum_k + psi(gamma_k)) and forms
//   tmp_j = sum_k E[log theta_k] phi_kj - sum_k phi_kj log phi_kj            (:343-344)
// With log phi_kj = E[log beta_kw] + psi(gamma_k) - LSE_j this is
//   tmp_j = LSE_j - psi(sum gamma) - sum_k phi_kj E[log beta_kw],
// which needs one pass over k (a running maximum, sum and weighted sum per lane, combined
// across the wave).  lda.cpp:334 indexes psiLambda by row where the column of the word is
// meant; this is the intended formula, see DESIGN.md.
#pragma once

#include "estep_kernels.h"

namespace trlda {

template <int T>
__global__ __launch_bounds__(T) void elbo_dense_kernel(int K, size_t total, double eta, double factor,
                                                       const double *__restrict__ lambda,
                                                       const double *__restrict__ psi_sum,
                                                       const double *__restrict__ sstats,
                                                       double *__restrict__ partial /* grid x 2 */)
{
    __shared__ double red[2][T / kWave];
    const size_t stride = (size_t)gridDim.x * T;
    size_t i = (size_t)blockIdx.x * T + threadIdx.x;
    int k = (int)(i % (size_t)K);
    const int kstep = (int)(stride % (size_t)K);
    double a = 0.0, b = 0.0;
    for (; i < total; i += stride) {
        const double l = lambda[i];
        a += (eta + factor * sstats[i] - l) * (digamma(l) - psi_sum[k]);
        b += lgamma(l);
        k += kstep;
        if (k >= K)
            k -= K;
    }
    a = wave_sum_dpp(a);
    b = wave_sum_dpp(b);
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    if (lane == 0) {
        red[0][wid] = a;
        red[1][wid] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sa = 0.0, sb = 0.0;
        for (int w = 0; w < T / kWave; ++w) {
            sa += red[0][w];
            sb += red[1][w];
        }
        partial[2 * blockIdx.x] = sa;
        partial[2 * blockIdx.x + 1] = sb;
    }
}

__device__ __forceinline__ double wave_max_all(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v = fmax(v, __shfl_xor(v, off, kWave));
    return v;
}

// out[2 d] = sum_j cnt_j tmp_j (:347), out[2 d + 1] = the document's part of E[log p(theta)] -
// E[log q(theta)] (:349-351)
template <int T>
__global__ __launch_bounds__(T) void elbo_docs_kernel(int K, const int32_t *__restrict__ indptr,
                                                      const int32_t *__restrict__ ids,
                                                      const int32_t *__restrict__ cnts,
                                                      const double *__restrict__ lambda,
                                                      const double *__restrict__ psi_sum,
                                                      const double *__restrict__ alpha,
                                                      const double *__restrict__ gamma,
                                                      double *__restrict__ out)
{
    constexpr int W = T / kWave;
    extern __shared__ double psig[];                 // K: psi(gamma_k), then W x 4 scratch
    double *red = psig + K;
    const int d = blockIdx.x;
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    const double *g = gamma + (size_t)d * K;

    double gs = 0.0, lg = 0.0, t1 = 0.0, t2 = 0.0;   // sum gamma, sum lgamma, sum (a-g) psi, sum (a-g)
    for (int k = threadIdx.x; k < K; k += T) {
        const double gk = g[k], pg = digamma(gk), ag = alpha[k] - gk;
        psig[k] = pg;
        gs += gk;
        lg += lgamma(gk);
        t1 += ag * pg;
        t2 += ag;
    }
    gs = wave_sum_dpp(gs);
    lg = wave_sum_dpp(lg);
    t1 = wave_sum_dpp(t1);
    t2 = wave_sum_dpp(t2);
    if (lane == 0) {
        red[wid * 4] = gs;
        red[wid * 4 + 1] = lg;
        red[wid * 4 + 2] = t1;
        red[wid * 4 + 3] = t2;
    }
    __syncthreads();
    gs = lg = t1 = t2 = 0.0;
    for (int w = 0; w < W; ++w) {
        gs += red[w * 4];
        lg += red[w * 4 + 1];
        t1 += red[w * 4 + 2];
        t2 += red[w * 4 + 3];
    }
    const double psi_gs = digamma(gs);
    __syncthreads();

    double pz = 0.0;
    for (int p = indptr[d] + wid; p < indptr[d + 1]; p += W) {
        const double *lrow = lambda + (size_t)ids[p] * K;
        double m = -__builtin_huge_val(), s = 0.0, t = 0.0;
        for (int k = lane; k < K; k += kWave) {
            const double a = digamma(lrow[k]) - psi_sum[k];  // E[log beta_kw]
            const double b = a + psig[k];
            if (b > m) {
                const double sc = exp(m - b);                // 0 on the first element
                s = fma(s, sc, 1.0);
                t = fma(t, sc, a);
                m = b;
            } else {
                const double e = exp(b - m);
                s += e;
                t = fma(e, a, t);
            }
        }
        const double M = wave_max_all(m);
        const double sc = (m > -__builtin_huge_val()) ? exp(m - M) : 0.0;
        const double S = wave_sum_dpp(s * sc), Tt = wave_sum_dpp(t * sc);
        pz += (double)cnts[p] * ((M + log(S)) - psi_gs - Tt / S);
    }
    if (lane == 0)
        red[wid] = pz;
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0.0;
        for (int w = 0; w < W; ++w)
            sum += red[w];
        out[2 * d] = sum;
        out[2 * d + 1] = (t1 - t2 * psi_gs) - lgamma(gs) + lg;
    }
}

}  // namespace trlda
