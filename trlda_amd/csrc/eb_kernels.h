// eb_kernels.h -- the reductions behind the empirical-Bayes steps for alpha and eta and the
// adaptive learning rate of OnlineLDA::updateParameters (reference src/onlinelda.cpp:116-175;
// the same sums feed BatchLDA's and CumulativeLDA's line searches, src/batchlda.cpp:66-205,
// src/cumulativelda.cpp:76-150).  The sums run where gamma, lambda and the statistics live;
// only K-sized results and a few scalars go back to the host, which keeps the Newton steps.
//
//   eb_gamma_kernel     out[c][k] = sum_{d in chunk c} psi(gamma_dk) - psi(sum_k gamma_dk)   :123-128
//   eb_lambda_kernel    out[b]    = sum over the block's elements of psi(lambda)             :153
//   adaptive_kernel     g = (1 - 1/tau) g + 1/tau (lambdaHat - lambda'); block sums of
//                       |lambdaHat - lambda'|^2 and |g|^2                                     :168-172
// Block results are added in block order (rowsum_combine_wave_kernel, or on the host for the
// scalars): reproducible run to run.
#pragma once

#include "estep_kernels.h"

namespace trlda {

constexpr int kEbDocsPerBlock = 8;

// One workgroup per chunk of kEbDocsPerBlock documents; thread t owns topics t, t + T, ..
// (their running sums in LDS).  Per document: sum_k gamma_dk by a block reduction, psi of it
// once, then psi(gamma_dk) - psi(sum) per topic.
template <int T>
__global__ __launch_bounds__(T) void eb_gamma_kernel(int K, int B, const double *__restrict__ gamma,
                                                     double *__restrict__ out /* chunks x K */)
{
    extern __shared__ double acc[];                  // K running sums | T/64 wave sums | 1
    double *wsum = acc + K;
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    for (int k = threadIdx.x; k < K; k += T)
        acc[k] = 0.0;
    const int d0 = blockIdx.x * kEbDocsPerBlock, d1 = min(B, d0 + kEbDocsPerBlock);
    for (int d = d0; d < d1; ++d) {
        const double *g = gamma + (size_t)d * K;
        double part = 0.0;
        for (int k = threadIdx.x; k < K; k += T)
            part += g[k];
        part = wave_sum_dpp(part);
        __syncthreads();                             // wsum of the previous document is consumed
        if (lane == 0)
            wsum[wid] = part;
        __syncthreads();
        double total = 0.0;
#pragma unroll
        for (int q = 0; q < T / kWave; ++q)
            total += wsum[q];
        const double psi_total = digamma(total);     // the same value in every thread
        for (int k = threadIdx.x; k < K; k += T)
            acc[k] += digamma(g[k]) - psi_total;
    }
    for (int k = threadIdx.x; k < K; k += T)
        out[(size_t)blockIdx.x * K + k] = acc[k];
}

// sum of psi(lambda) over the flat array: grid-stride, one result per block
template <int T>
__global__ __launch_bounds__(T) void eb_lambda_kernel(size_t total, const double *__restrict__ lambda,
                                                      double *__restrict__ out /* grid */)
{
    __shared__ double red[T / kWave];
    const size_t stride = (size_t)gridDim.x * T;
    double a[2] = {0.0, 0.0};
    size_t i = (size_t)blockIdx.x * T + threadIdx.x;
    for (; i + stride < total; i += 2 * stride) {
        const double l0 = lambda[i], l1 = lambda[i + stride];
        a[0] += digamma(l0);
        a[1] += digamma(l1);
    }
    if (i < total)
        a[0] += digamma(lambda[i]);
    const double s = wave_sum_dpp(a[0] + a[1]);
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    if (lane == 0)
        red[wid] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < T / kWave; ++w)
            t += red[w];
        out[blockIdx.x] = t;
    }
}

// onlinelda.cpp:168-172.  out[2 b] = sum upd^2, out[2 b + 1] = sum g^2 over the block's elements.
template <int T>
__global__ __launch_bounds__(T) void adaptive_kernel(size_t total, double eta, double scale, double tau,
                                                     const double *__restrict__ sstats,
                                                     const double *__restrict__ lambda_prime,
                                                     double *__restrict__ gradient,
                                                     double *__restrict__ out /* grid x 2 */)
{
    __shared__ double red[2][T / kWave];
    const size_t stride = (size_t)gridDim.x * T;
    const double keep = 1. - 1. / tau, add = 1. / tau;
    double u2 = 0.0, g2 = 0.0;
    for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < total; i += stride) {
        const double upd = (eta + scale * sstats[i]) - lambda_prime[i];
        const double g = keep * gradient[i] + add * upd;
        gradient[i] = g;
        u2 += upd * upd;
        g2 += g * g;
    }
    u2 = wave_sum_dpp(u2);
    g2 = wave_sum_dpp(g2);
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    if (lane == 0) {
        red[0][wid] = u2;
        red[1][wid] = g2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < T / kWave; ++w) {
            a += red[0][w];
            b += red[1][w];
        }
        out[2 * blockIdx.x] = a;
        out[2 * blockIdx.x + 1] = b;
    }
}

}  // namespace trlda
