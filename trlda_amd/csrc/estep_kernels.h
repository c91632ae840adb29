// estep_kernels.h -- HIP kernels for gfx950 (MI355X) implementing
// LDA::updateVariablesVI (reference src/lda.cpp:160-220) and the lambda M-step
// (src/onlinelda.cpp:79-110).  fp64 throughout; matrices column-major (a word's
// K values are contiguous), documents CSR int32.
//
// Launch sequence of one E-step (host side: trlda_hip.hip):
//   1. rowsum_psi_kernel      psiSum_k = psi(sum_w lambda_kw)            lda.cpp:172
//   2. exp_elog_beta_kernel   eeb = exp(psi(lambda) - psiSum)            lda.cpp:173
//   3. estep_docs_kernel      per-document gamma fixed point             lda.cpp:174-204
//                             (+ atomics into sstats in ATOMIC mode,     lda.cpp:207-213)
//   4. sstats_words_kernel    ordered per-word sums * eeb                lda.cpp:207-217
//      or finish_kernel       sstats *= eeb (ATOMIC mode)                lda.cpp:217
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "psi.h"

namespace trlda {

constexpr int kWave = 64;

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off, kWave);
    return v;  // valid in lane 0
}

// ---------------------------------------------------------------------------
// 1. psiSum_k = psi(sum_w lambda[k, w]).
//
// Grid of G blocks; block b owns words [b*wpb, (b+1)*wpb).  Thread t covers topic
// k = t % K of word slot t / K, so a block pass reads floor(T/K)*K consecutive
// doubles (fully coalesced) and each thread keeps one running sum.  Block partials
// go to partial[b][k]; the last block to arrive (agent-scope counter) adds the
// partials in block order -- a fixed order, so psiSum is bitwise reproducible --
// and applies psi.  For K > T the topics are tiled in chunks of T.
// ---------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(T) void rowsum_psi_kernel(
    int K, int V, int wpb, const double *__restrict__ lambda,
    double *__restrict__ partial /* G x K */, double *__restrict__ psi_sum /* K */,
    unsigned int *__restrict__ counter)
{
    __shared__ double red[T];
    __shared__ bool is_last;
    const int tid = threadIdx.x;
    const int w0 = blockIdx.x * wpb;
    const int w1 = min(V, w0 + wpb);

    for (int kbase = 0; kbase < K; kbase += T) {
        const int kc = min(T, K - kbase);      // topics in this chunk
        const int slots = T / kc;              // word slots per pass
        const int slot = tid / kc;
        const int k = kbase + tid % kc;
        double acc = 0.0;
        if (slot < slots)
            for (int w = w0 + slot; w < w1; w += slots)
                acc += lambda[(size_t)w * K + k];
        red[tid] = acc;
        __syncthreads();
        if (tid < kc) {
            double s = red[tid];
            for (int sl = 1; sl < slots; ++sl)
                s += red[sl * kc + tid];
            partial[(size_t)blockIdx.x * K + kbase + tid] = s;
        }
        __syncthreads();
    }

    // publish partials, find out whether this block is the last one
    __threadfence();
    if (tid == 0) {
        unsigned int prev = atomicAdd(counter, 1u);
        is_last = (prev == gridDim.x - 1);
    }
    __syncthreads();
    if (!is_last)
        return;
    __threadfence();
    for (int k = tid; k < K; k += T) {
        double s = 0.0;
        for (unsigned int b = 0; b < gridDim.x; ++b)
            s += __builtin_nontemporal_load(&partial[(size_t)b * K + k]);
        psi_sum[k] = digamma(s);
    }
    if (tid == 0)
        *counter = 0;  // ready for the next launch on this stream
}

// ---------------------------------------------------------------------------
// 2. eeb[i] = exp(psi(lambda[i]) - psiSum[i % K]) over the flat K*V array.
// Grid-stride; the topic index advances incrementally (no per-element modulo).
// ---------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(T) void exp_elog_beta_kernel(
    int K, size_t total, const double *__restrict__ lambda,
    const double *__restrict__ psi_sum, double *__restrict__ eeb)
{
    const size_t stride = (size_t)gridDim.x * T;
    size_t i = (size_t)blockIdx.x * T + threadIdx.x;
    int k = (int)(i % (size_t)K);
    const int kstep = (int)(stride % (size_t)K);
    for (; i < total; i += stride) {
        eeb[i] = exp(digamma(lambda[i]) - psi_sum[k]);
        k += kstep;
        if (k >= K)
            k -= K;
    }
}

// ---------------------------------------------------------------------------
// 3. Per-document fixed point.  One workgroup of T threads per document.
//
// LDS (doubles):  beta[n][Kp] (the document's slice of eeb, row = word, Kp odd so
// that both access directions are bank-conflict free) | g[K] gamma | e[K]
// exp(psi(gamma)) | tw[n] cnt_j/phinorm_j | part[max(T,K)] | wsum[T/64].
//
// Two matrix-vector products per iteration over the same K x n slice:
//   B: acc_k  = sum_j tw_j beta[j][k]        lanes run over k  (lda.cpp:189-193)
//   E: phin_j = sum_k e_k  beta[j][k]        lanes run over j  (lda.cpp:199)
// Both read LDS conflict-free; when K (or n) is at most T/2 the other index is split
// over thread groups whose partial sums are combined in a fixed order.
//
// Documents whose slice does not fit the LDS budget (n > n_cap) take the streaming
// path: beta is re-read from eeb (L2 / Infinity Cache) each iteration with lanes
// over k, and tw lives in the global tw_csr scratch.
// ---------------------------------------------------------------------------
struct DocKernelArgs {
    int K, Kp, n_cap, B;
    const int32_t *indptr, *ids, *cnts;
    const int32_t *order;     // optional processing order (long documents first)
    const double *eeb;        // K x V
    const double *alpha;      // K
    double *gamma;            // K x B in/out
    double *epg;              // K x B out: exp(psi(gamma)) of the returned gamma
    double *tw_csr;           // nnz: cnt/phinorm in CSR order (scratch + output)
    const int32_t *wrank;     // nnz: CSR position -> word-major rank (segmented mode)
    double *tw_word;          // nnz: cnt/phinorm in word-major order (segmented mode)
    double *sstats_acc;       // K x V atomic target (atomic mode) or nullptr
    int max_iter;
    double threshold;
    int32_t *iters_out;       // B or nullptr
};

template <int T>
__global__ __launch_bounds__(T) void estep_docs_kernel(DocKernelArgs a)
{
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid = tid / kWave;
    constexpr int W = T / kWave;

    const int d = a.order ? a.order[blockIdx.x] : blockIdx.x;
    const int K = a.K, Kp = a.Kp;
    const int p0 = a.indptr[d];
    const int n = a.indptr[d + 1] - p0;
    const bool staged = n <= a.n_cap;
    const int32_t *ids = a.ids + p0;
    const int32_t *cnts = a.cnts + p0;

    // LDS carve-up (beta first: its size depends on the launch's n_cap)
    double *beta = lds;
    double *g = beta + (size_t)a.n_cap * Kp;
    double *e = g + K;
    double *tw_l = e + K;
    double *part = tw_l + a.n_cap;
    double *wsum = part + (K > T ? K : T);
    double *tw = staged ? tw_l : a.tw_csr + p0;

    double *gamma_d = a.gamma + (size_t)d * K;

    for (int k = tid; k < K; k += T) {               // lda.cpp:174
        double gk = gamma_d[k];
        g[k] = gk;
        e[k] = exp_digamma(gk);
    }
    if (staged) {                                    // lda.cpp:179-181
        for (int j = wid; j < n; j += W) {
            const double *src = a.eeb + (size_t)ids[j] * K;
            double *dst = beta + (size_t)j * Kp;
            for (int k = lane; k < K; k += kWave)
                dst[k] = src[k];
        }
    }
    __syncthreads();

    // geometry of the two products
    const int kslots = min((K + kWave - 1) / kWave * kWave, T);
    const int jparts = T / kslots;                   // >= 1
    const int ks = tid % kslots, jp = tid / kslots;
    const int jslots = min((n + kWave - 1) / kWave * kWave, T);
    const int kparts = n > 0 ? T / jslots : 1;
    const int js = n > 0 ? tid % jslots : 0, kp = n > 0 ? tid / jslots : 0;
    const int kchunk = (K + kparts - 1) / kparts;

    // phinorm / tw from the current e (lda.cpp:183 and :199)
    auto phase_E = [&]() {
        if (staged) {
            for (int j0 = 0; j0 < n; j0 += jslots) {
                const int j = j0 + js;
                double s = 0.0;
                if (j < n && kp < kparts) {
                    const int klo = kp * kchunk, khi = min(K, klo + kchunk);
                    const double *row = beta + (size_t)j * Kp;
                    for (int k = klo; k < khi; ++k)
                        s += e[k] * row[k];
                }
                if (kparts > 1) {
                    part[tid] = s;
                    __syncthreads();
                    if (kp == 0 && j < n)
                        for (int q = 1; q < kparts; ++q)
                            s += part[q * jslots + js];
                }
                if (kp == 0 && j < n)
                    tw[j] = (double)cnts[j] / (s + 1e-100);
                if (kparts > 1 && j0 + jslots < n)
                    __syncthreads();
            }
        } else {
            for (int j = wid; j < n; j += W) {
                const double *col = a.eeb + (size_t)ids[j] * K;
                double s = 0.0;
                for (int k = lane; k < K; k += kWave)
                    s += e[k] * col[k];
                s = wave_sum(s);
                if (lane == 0)
                    tw[j] = (double)cnts[j] / (s + 1e-100);
            }
        }
    };

    phase_E();
    __syncthreads();

    int it = 0;
    while (it < a.max_iter) {                        // lda.cpp:185-204
        // B: acc_k = sum_j tw_j * beta[j][k]
        for (int k = ks; k < K; k += kslots) {
            double acc = 0.0;
            if (jp < jparts) {
                if (staged) {
                    for (int j = jp; j < n; j += jparts)
                        acc += tw[j] * beta[(size_t)j * Kp + k];
                } else {
                    for (int j = jp; j < n; j += jparts)
                        acc += tw[j] * a.eeb[(size_t)ids[j] * K + k];
                }
                part[jp * K + k] = acc;
            }
        }
        __syncthreads();

        // gamma_k = alpha_k + e_k * acc_k ; e_k = exp(psi(gamma_k))   lda.cpp:194-197
        double diff = 0.0;
        for (int k = tid; k < K; k += T) {
            double acc = part[k];
            for (int q = 1; q < jparts; ++q)
                acc += part[q * K + k];
            double gnew = acc * e[k] + a.alpha[k];
            diff += fabs(g[k] - gnew);
            g[k] = gnew;
            e[k] = exp_digamma(gnew);
        }
        diff = wave_sum(diff);
        if (lane == 0)
            wsum[wid] = diff;
        __syncthreads();

        phase_E();
        ++it;

        double change = 0.0;                         // lda.cpp:202-203
#pragma unroll
        for (int q = 0; q < W; ++q)
            change += wsum[q];
        __syncthreads();
        if (change / (double)K < a.threshold)
            break;
    }

    // results
    for (int k = tid; k < K; k += T) {
        gamma_d[k] = g[k];
        a.epg[(size_t)d * K + k] = e[k];
    }
    if (tid == 0 && a.iters_out)
        a.iters_out[d] = it;

    if (a.sstats_acc) {                              // lda.cpp:207-213, atomic form
        for (int j = wid; j < n; j += W) {
            const double c = tw[j];
            double *col = a.sstats_acc + (size_t)ids[j] * K;
            for (int k = lane; k < K; k += kWave)
                unsafeAtomicAdd(&col[k], c * e[k]);
        }
    } else {
        for (int j = tid; j < n; j += T)
            a.tw_word[a.wrank[p0 + j]] = tw[j];
    }
}

// ---------------------------------------------------------------------------
// 4a. Segmented sufficient statistics: one wavefront per word, lanes over k.
//   sstats[k, w] = eeb[k, w] * sum_{q in word w} tw_word[q] * epg[k, wdoc[q]]
// The entries of a word are stored in document order, so the additions happen in
// the order of the reference's serial loop (lda.cpp:207-213) and the result is
// bitwise reproducible.  Words without entries get 0 (lda.cpp:169).
// Optionally fuses the M-step blend (onlinelda.cpp:99-100):
//   lambda_out = (1-rho) lambda' + rho (eta + scale * sstats)
// ---------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(T) void sstats_words_kernel(
    int K, int V, const int32_t *__restrict__ wptr, const int32_t *__restrict__ wdoc,
    const double *__restrict__ tw_word, const double *__restrict__ epg,
    const double *__restrict__ eeb, double *__restrict__ sstats)
{
    const int lane = threadIdx.x & (kWave - 1);
    // wave-uniform word index in an SGPR: wptr / wdoc / tw_word become scalar loads
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * (T / kWave) + threadIdx.x / kWave);
    if (w >= V)
        return;
    const int q0 = wptr[w], q1 = wptr[w + 1];
    for (int kb = 0; kb < K; kb += kWave) {
        const int k = kb + lane;
        const bool on = k < K;
        double acc = 0.0;
        int q = q0;
        for (; q + 4 <= q1; q += 4) {               // four gathers in flight
            const int d0 = wdoc[q], d1 = wdoc[q + 1], d2 = wdoc[q + 2], d3 = wdoc[q + 3];
            const double t0 = tw_word[q], t1 = tw_word[q + 1], t2 = tw_word[q + 2],
                         t3 = tw_word[q + 3];
            const double e0 = on ? epg[(size_t)d0 * K + k] : 0.0;
            const double e1 = on ? epg[(size_t)d1 * K + k] : 0.0;
            const double e2 = on ? epg[(size_t)d2 * K + k] : 0.0;
            const double e3 = on ? epg[(size_t)d3 * K + k] : 0.0;
            acc += t0 * e0;
            acc += t1 * e1;
            acc += t2 * e2;
            acc += t3 * e3;
        }
        for (; q < q1; ++q) {
            const double ev = on ? epg[(size_t)wdoc[q] * K + k] : 0.0;
            acc += tw_word[q] * ev;
        }
        if (on) {
            const size_t i = (size_t)w * K + k;
            sstats[i] = acc * eeb[i];
        }
    }
}

// 4b. Atomic mode finish: sstats *= eeb (lda.cpp:217).
template <int T>
__global__ __launch_bounds__(T) void finish_kernel(size_t total, const double *__restrict__ eeb,
                                                   double *__restrict__ sstats)
{
    const size_t stride = (size_t)gridDim.x * T;
    for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < total; i += stride)
        sstats[i] *= eeb[i];
}

// ---------------------------------------------------------------------------
// M-step kernels.
// ---------------------------------------------------------------------------

// lambda = (1-rho) lambda' + rho (eta + scale * sstats)     onlinelda.cpp:99-100
template <int T>
__global__ __launch_bounds__(T) void blend_kernel(size_t total, double rho, double eta,
                                                  double scale,
                                                  const double *__restrict__ lambda_prime,
                                                  const double *__restrict__ sstats,
                                                  double *__restrict__ lambda)
{
    const size_t stride = (size_t)gridDim.x * T;
    for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < total; i += stride) {
        const double hat = eta + scale * sstats[i];
        lambda[i] = (1. - rho) * lambda_prime[i] + rho * hat;
    }
}

// wordcounts[w] += cnt (integers in fp64: exact, order-free)   onlinelda.cpp:79-82
template <int T>
__global__ __launch_bounds__(T) void wordcount_kernel(int64_t nnz, const int32_t *__restrict__ ids,
                                                      const int32_t *__restrict__ cnts,
                                                      double *__restrict__ wordcounts)
{
    const size_t stride = (size_t)gridDim.x * T;
    for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < (size_t)nnz; i += stride)
        unsafeAtomicAdd(&wordcounts[ids[i]], (double)cnts[i]);
}

// lambda[:, w] = (1-rho) lambda'[:, w] + rho (eta + coef * wordcounts[w])  onlinelda.cpp:85-86
template <int T>
__global__ __launch_bounds__(T) void tr_init_kernel(int K, size_t total, double rho, double eta,
                                                    double coef,
                                                    const double *__restrict__ wordcounts,
                                                    const double *__restrict__ lambda_prime,
                                                    double *__restrict__ lambda)
{
    const size_t stride = (size_t)gridDim.x * T;
    for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < total; i += stride) {
        const size_t w = i / (size_t)K;
        const double add = rho * (eta + coef * wordcounts[w]);
        lambda[i] = (1. - rho) * lambda_prime[i] + add;
    }
}

}  // namespace trlda
