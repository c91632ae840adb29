// estep_kernels.h -- HIP kernels for gfx950 (MI355X) implementing
// LDA::updateVariablesVI (reference src/lda.cpp:160-220) and the lambda M-step
// (src/onlinelda.cpp:79-110).  fp64 throughout; matrices column-major (a word's
// K values are contiguous), documents CSR int32.
//
// Launch sequence of one E-step (host side: trlda_hip.hip, estep_device):
//   1. rowsum_partial_kernel  block partials of sum_w lambda_kw           lda.cpp:172
//      (+ rowsum_combine_kernel on large tables)
//   2. exp_elog_beta_kernel   psiSum, eeb = exp(psi(lambda) - psiSum)     lda.cpp:172-173
//      -- or, on small tables, 1 + 2 as preamble_fused_kernel (section 2b)
//   3. the per-document gamma fixed point                                 lda.cpp:174-204
//        estep_docs_reg_kernel   K <= 128, at most 192 words (section 3c)
//        estep_docs_wide_kernel  K <= 512, any length (estep_wide.h)
//        estep_docs_kernel       any K (section 3)
//      (+ atomics into sstats in ATOMIC mode,                             lda.cpp:207-213)
//   4. sstats_words_kernel    ordered per-word sums * eeb                 lda.cpp:207-217
//      or finish_kernel       sstats *= eeb (ATOMIC mode)                 lda.cpp:217
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "index_params.h"
#include "psi.h"

namespace trlda {

constexpr int kWave = 64;

// "Everything this thread has stored so far has been acknowledged by the memory side" -- the
// release half of every counter / flag hand-off between workgroups of one launch in this library
// (finish_partial_groups, segment_finish, the prefetched preamble's row sums, estep_merged.h).
// The data that crosses workgroups goes out with agent-scope (sc1, write-through) stores; what
// remains is that those stores have LEFT before the workgroup's counter increment is issued.
// A workgroup-scope release fence does not do that: outside threadgroup-split mode the compiler
// emits no vector-memory wait for it (`global_store ... sc1 ; s_barrier ;
// global_atomic_add` with only lgkmcnt(0) in between), so the wait is spelled out -- on gfx9
// stores count in vmcnt like loads.  An agent-scope fence would add the write-back of the XCD's
// whole L2 (300 us per launch when it was tried) and is not needed for
// write-through stores.  The fence that follows keeps the compiler from moving stores across.
__device__ __forceinline__ void stores_acknowledged()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}

// The acquire half on a waiter's side, after it has seen its flag: compiler ordering only (no
// buffer_inv -- estep_merged.h, merged_wait_docs, says why no cache line can be stale), so that
// no load of the data is hoisted above the flag's.
__device__ __forceinline__ void flag_seen()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    asm volatile("" ::: "memory");
}

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
// go to partial[b][k]; rowsum_finish_kernel adds them in block order -- a fixed
// order, so psiSum is bitwise reproducible -- and applies psi.  For K > T the topics
// are tiled in chunks of T.
// ---------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(T) void rowsum_partial_kernel(
    int K, int V, int wpb, const double *__restrict__ lambda, double *__restrict__ partial /* G x K */)
{
    __shared__ double red[T];
    const int tid = threadIdx.x;
    const int w0 = blockIdx.x * wpb;
    const int w1 = min(V, w0 + wpb);

    for (int kbase = 0; kbase < K; kbase += T) {
        const int kc = min(T, K - kbase);      // topics in this chunk
        const int slots = T / kc;              // word slots per pass
        const int slot = tid / kc;
        const int k = kbase + tid % kc;
        // sixteen loads in flight (clamped word index, masked value: no branch between
        // loads), four accumulation chains combined pairwise -- a fixed order
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        if (slot < slots) {
            for (int w = w0 + slot; w < w1; w += 16 * slots) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    v[u] = lambda[(size_t)min(w + u * slots, w1 - 1) * K + k];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    acc[u & 3] += (w + u * slots < w1) ? v[u] : 0.0;
            }
        }
        red[tid] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        __syncthreads();
        if (tid < kc) {
            double s = red[tid];
            for (int sl = 1; sl < slots; ++sl)
                s += red[sl * kc + tid];
            partial[(size_t)blockIdx.x * K + kbase + tid] = s;
        }
        __syncthreads();
    }
}

// 1b. Large K x V only: the row-sum grid is wider than 64 blocks there (it has to pull
// hundreds of MB through HBM) and the eeb blocks should not each re-add that many partials.
// combined[k] = sum_b partial[b][k]: eight threads per topic add G/8 block partials each
// (all loads in flight), one thread combines the eight in order -- a fixed order.
template <int T>
__global__ __launch_bounds__(T) void rowsum_combine_kernel(int K, int G,
                                                           const double *__restrict__ partial,
                                                           double *__restrict__ combined)
{
    __shared__ double scratch[T];
    constexpr int TPB = T / 8;                       // topics per block
    const int kl = threadIdx.x % TPB, part = threadIdx.x / TPB;
    const int k = blockIdx.x * TPB + kl;
    const int per = (G + 7) / 8;
    const int b0 = part * per, b1 = min(G, b0 + per);
    double acc[2] = {0.0, 0.0};
    if (k < K) {
        for (int b = b0; b < b1; b += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = partial[(size_t)min(b + u, G - 1) * K + k];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                acc[u & 1] += (b + u < b1) ? v[u] : 0.0;
        }
    }
    scratch[part * TPB + kl] = acc[0] + acc[1];
    __syncthreads();
    if (part == 0 && k < K) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            v[u] = scratch[u * TPB + kl];
        combined[k] = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
}

// ---------------------------------------------------------------------------
// 2. eeb[i] = exp(psi(lambda[i]) - psiSum[i % K]) over the flat K*V array.
//
// Prologue: every block forms psiSum_k = psi(sum_b partial[b][k]) itself (G loads in flight
// per topic, four chains, pairwise combine -- the same fixed order in every block) and
// keeps it in LDS.  That costs each block about a microsecond of latency but saves a
// launch plus a single-block reduction kernel; the grid is one fat block per CU so the
// redundant psi evaluations stay below 4 % of the element work.
// Main loop: grid-stride; the topic index advances incrementally (no per-element modulo).
//
// Only the columns of words that occur in the batch are ever read (by the document kernels
// and, times the word's statistics, by the statistics kernel; untouched words get
// sstats = 0 without reading eeb), so by default only those columns are computed: the
// `active` list comes with the batch.  gamma and sstats are unchanged by this; eeb is not an
// output.  trlda_model_set_dense_preamble(model, 1) fills all V columns as the reference does.
// ---------------------------------------------------------------------------
constexpr int kRowsumBlocks = 64;

template <int T>
__global__ __launch_bounds__(T) void exp_elog_beta_kernel(
    int K, size_t total, int G, const double *__restrict__ lambda,
    const double *__restrict__ partial, double *__restrict__ psi_sum_out /* 2K: psi, then sums */,
    double *__restrict__ eeb, const int32_t *__restrict__ active /* word ids or nullptr */)
{
    extern __shared__ double psi_sum[];             // K, then 8 x K scratch
    double *scratch = psi_sum + K;
    // eight threads per topic, each adds G/8 block partials (all loads in flight), then
    // one thread per topic combines the eight in order and applies psi
    const int per = (G + 7) / 8;
    for (int t = threadIdx.x; t < 8 * K; t += T) {
        const int k = t % K, part = t / K;
        const int b0 = part * per, b1 = min(G, b0 + per);
        double acc[2] = {0.0, 0.0};
        for (int b = b0; b < b1; b += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = partial[(size_t)min(b + u, G - 1) * K + k];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                acc[u & 1] += (b + u < b1) ? v[u] : 0.0;
        }
        scratch[part * K + k] = acc[0] + acc[1];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += T) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            v[u] = scratch[u * K + k];
        const double rs = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        const double ps = digamma(rs);
        psi_sum[k] = ps;
        if (blockIdx.x == 0) {
            psi_sum_out[k] = ps;                     // kept for later kernels (elbo_kernels.h)
            psi_sum_out[K + k] = rs;
        }
    }
    __syncthreads();

    // `total` = K * (number of columns to fill).  With `active` the flat index i walks the
    // batch's active words only: column a = i / K is word active[a].
    const size_t stride = (size_t)gridDim.x * T;
    size_t i = (size_t)blockIdx.x * T + threadIdx.x;
    int k = (int)(i % (size_t)K);
    size_t a = i / (size_t)K;
    const int kstep = (int)(stride % (size_t)K);
    const size_t astep = stride / (size_t)K;
    // Four elements per pass, their loads (word id -> lambda: two dependent latencies) in flight
    // together and issued unconditionally (a load behind a lane-dependent branch is waited for
    // on the spot; elements past the end repeat the pass's first one and are not stored): with
    // one element per pass a thread's chain of ~100 passes was latency-bound (190 us for 28 M
    // elements at K = 500, B = 4096).
    constexpr int U = 4;                             // (eight at 512 threads: slower, 3 waves per SIMD)
    for (; i < total; i += U * stride) {
        size_t au[U], idx[U];
        int ku[U], wu[U];
        double lam[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool on = i + u * stride < total;
            ku[u] = on ? k : ku[0];
            au[u] = on ? a : au[0];
            k += kstep;
            a += astep;
            if (k >= K) {
                k -= K;
                a += 1;
            }
        }
        if (active) {                                // launch-uniform
#pragma unroll
            for (int u = 0; u < U; ++u)
                wu[u] = active[au[u]];
#pragma unroll
            for (int u = 0; u < U; ++u)
                idx[u] = (size_t)wu[u] * K + ku[u];
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u)
                idx[u] = au[u] * (size_t)K + ku[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            lam[u] = lambda[idx[u]];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double v = exp_digamma_minus(lam[u], psi_sum[ku[u]]);
            if (i + u * stride < total)
                eeb[idx[u]] = v;
        }
    }
}

// ---------------------------------------------------------------------------
// 2b. Small tables (the register-resident document kernel takes the whole batch): ONE launch
// instead of kernels 1 and 2.  exp(psi(lambda) - psiSum) = exp(psi(lambda)) * exp(-psiSum):
// the first factor does not need the row sums, so G workgroups add up the row sums while the
// others fill u = exp(psi(lambda)) for the batch's active words.  The K factors
// c_k = exp(-psi(sum_b partial[b][k])) are formed by every document workgroup between two
// barriers it has anyway (topic_scale_load / _finish) and folded into exp(psi(gamma)):
//   phinorm_j = sum_k (c_k e_k) u_jk,   gamma_k = alpha_k + (c_k e_k) sum_j tw_j u_jk,
// so the document kernel works on u with e~ = c e in place of e, leaves e~ in epg, and the
// statistics kernel's product with "eeb" (= u here) is again sstats (lda.cpp:207-217).
// Both kernels this replaces spend most of their few microseconds on launch and on waiting
// for the row sums.  Against the one-exponential form the results move by a few ulp.
// ---------------------------------------------------------------------------
// (vb, nb: this workgroup's index and the number of workgroups doing this job -- the kernel's
// own grid, or the extra workgroups of a document-kernel launch that prepare the NEXT batch's
// preamble, estep_docs_reg_kernel; red: T doubles of LDS)
// A deferred launch's helper publishes its NEXT item (estep_merged.h, deferred_helper) from inside
// the current one, at a point where the item's loads have been consumed and none of its stores has
// been issued: the value comes from an atomic issued before the item, the counter that orders
// vector-memory operations is one per wave and in order, so reading it THERE waits for nothing that
// is not waited for anyway -- at the item's end it would wait for the stores' acknowledgements.
struct NoPublish {
    __device__ __forceinline__ void operator()() const {}
};
struct PublishNext {
    unsigned int *slot;           // LDS
    unsigned int fetched, base;
    __device__ __forceinline__ void operator()() const
    {
        if (threadIdx.x == 0)
            *slot = fetched - base;
    }
};

// FAT: the fill workgroups take 16 elements per thread, all their loads in flight before the first
// exp(psi) (a deferred launch's helpers: one 512-thread workgroup per CU, so a workgroup's time is
// its chain of memory latencies -- two passes of two dependent loads each cost four of them)
template <int T, bool FAT = false, class Pub = NoPublish>
__device__ __forceinline__ void preamble_fused_body(
    int vb, int nb, double *red, int K, int V, int G, int wpb, size_t total,
    const double *__restrict__ lambda, double *__restrict__ partial /* G x K */,
    double *__restrict__ u, const int32_t *__restrict__ active /* word ids or nullptr */, int GC,
    const double *__restrict__ carry_rows /* carry_n x K */, int carry_n,
    const double *__restrict__ carry_base /* K or nullptr */, double *__restrict__ carry_out /* GC x K */,
    double *__restrict__ c_out = nullptr /* 3 K: psi(row sums), row sums, exp(-psi) */,
    unsigned int *c_counter = nullptr, double *c_scratch = nullptr /* 8 K doubles of LDS */,
    const Pub &pub = Pub(), int tid_bias = 0 /* see deferred_helper: an opaque zero */)
{
    const int tid = (int)threadIdx.x + tid_bias;
    // Row sums carried over from the kernel that wrote lambda, still in carry_n block partials
    // (sstats_update_kernel, or the streaming pass of the initial step): the first GC workgroups
    // add up a contiguous range of the rows each -- four threads per topic, eight loads in
    // flight, parts combined in order; workgroup 0 adds the share of the words outside the
    // batch -- and the document workgroups finish the sum over the GC rows (topic_scale_*).
    if (vb < GC) {                      // block-uniform
        const int per = (carry_n + GC - 1) / GC;
        const int r0 = min(carry_n, vb * per), r1 = min(carry_n, r0 + per);
        const int k = tid % 128, part = tid / 128;   // K <= 128, T = 512: four parts
        double acc[2] = {0.0, 0.0};
        if (k < K) {
            for (int r = r0 + part; r < r1; r += 4 * 8) {
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    v[q] = carry_rows[(size_t)min(r + 4 * q, carry_n - 1) * K + k];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    acc[q & 1] += (r + 4 * q < r1) ? v[q] : 0.0;
            }
        }
        red[tid] = acc[0] + acc[1];
        pub();
        __syncthreads();
        if (tid < K) {
            double sum = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
            if (vb == 0 && carry_base)
                sum += carry_base[tid];
            carry_out[(size_t)vb * K + tid] = sum;
        }
        return;
    }
    const int lead = GC + G;
    // row sums: the first G workgroups, words [b * wpb, (b + 1) * wpb) each (K <= T here),
    // as rowsum_partial_kernel
    if (vb < lead) {                    // block-uniform
        const int w0 = (vb - GC) * wpb;
        const int w1 = min(V, w0 + wpb);
        const int slots = T / K;
        const int slot = tid / K;
        const int k = tid % K;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        if (slot < slots) {
            for (int w = w0 + slot; w < w1; w += 16 * slots) {
                double v[16];
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    v[q] = lambda[(size_t)min(w + q * slots, w1 - 1) * K + k];
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    acc[q & 3] += (w + q * slots < w1) ? v[q] : 0.0;
            }
        }
        red[tid] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        pub();
        __syncthreads();
        if (tid < K) {
            double sum = red[tid];
            for (int sl = 1; sl < slots; ++sl)
                sum += red[sl * K + tid];
            if (c_out)                               // (read by another workgroup of this launch)
                __hip_atomic_store(partial + (size_t)(vb - GC) * K + tid, sum, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            else
                partial[(size_t)(vb - GC) * K + tid] = sum;
        }
        // The topic factors c_k = exp(-psi(row sum_k)), finished HERE when nobody is waiting for
        // this launch's end (the preamble of the NEXT batch, riding on a document launch): the
        // last row-sum workgroup to finish adds up the G rows -- in topic_scale_*'s order, so
        // that the factors are bitwise those every document workgroup would form -- and the
        // documents of the next launch load K numbers instead of summing 64 rows and evaluating
        // psi and exp between two of their barriers (~1500 cycles of every document's latency).
        // Visibility of the rows as in finish_partial_groups: sc1 stores / loads, a workgroup
        // release before the counter, no agent-scope fence.
        if (c_out) {                                 // launch-uniform
            __shared__ int last_rows;
            stores_acknowledged();
            __syncthreads();
            if (tid == 0) {
                const unsigned int seen = __hip_atomic_fetch_add(c_counter, 1u, __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_AGENT);
                const int last = seen + 1u == (unsigned int)G;
                if (last)
                    __hip_atomic_store(c_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last_rows = last;
            }
            __syncthreads();
            if (last_rows) {
                const int per = (G + 7) / 8;         // topic_scale_load / _partials, sc1 loads
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int t = tid + h * T;
                    if (t < 8 * K) {
                        const int k = t % K, b0 = (t / K) * per, b1 = min(G, b0 + per);
                        double v[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            v[q] = __hip_atomic_load(partial + (size_t)min(b0 + q, G - 1) * K + k,
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        double acc[2] = {0.0, 0.0};
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            acc[q & 1] += (b0 + q < b1) ? v[q] : 0.0;
                        c_scratch[t] = acc[0] + acc[1];
                    }
                }
                __syncthreads();
                if (tid < K) {                       // topic_scale_combine
                    double w[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        w[q] = c_scratch[q * K + tid];
                    const double rs = ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));
                    const double ps = digamma(rs);
                    c_out[tid] = ps;
                    c_out[K + tid] = rs;
                    c_out[2 * K + tid] = exp(-ps);
                }
            }
        }
        return;
    }
    // u = exp(psi(lambda)) on the active words, the other workgroups: grid-stride over the
    // flat index i = a * K + k of (active word a, topic k), two elements per pass so that
    // their loads overlap
    // (total < 2^22 here: 32-bit index arithmetic, a 64-bit division costs as much as psi)
    const unsigned stride = (nb - lead) * T, tot = (unsigned)total, Ku = (unsigned)K;
    if constexpr (FAT) {
        constexpr int U = 16;
        for (unsigned i0 = (vb - lead) * T + tid; i0 < tot; i0 += U * stride) {
            size_t idx[U];
            double l[U];
            unsigned a[U];
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const unsigned i = min(i0 + q * stride, tot - 1u);
                a[q] = i / Ku;
                idx[q] = i - a[q] * Ku;
            }
            if (active) {                            // launch-uniform
                int w[U];
#pragma unroll
                for (int q = 0; q < U; ++q)
                    w[q] = active[a[q]];
#pragma unroll
                for (int q = 0; q < U; ++q)
                    idx[q] += (size_t)w[q] * K;
            } else {
#pragma unroll
                for (int q = 0; q < U; ++q)
                    idx[q] += (size_t)a[q] * K;
            }
#pragma unroll
            for (int q = 0; q < U; ++q)
                l[q] = lambda[idx[q]];
#pragma unroll
            for (int q = 0; q < U; ++q)
                l[q] = exp_digamma(l[q]);
            if (i0 == (vb - lead) * T + tid)         // (first pass: every load consumed, no store issued)
                pub();
#pragma unroll
            for (int q = 0; q < U; ++q)
                if (i0 + q * stride < tot)
                    u[idx[q]] = l[q];
        }
        if ((vb - lead) * T + tid >= tot)            // (a thread without a pass: thread 0 has one whenever
            pub();                                   //  the workgroup has any element; else publish here)
        return;
    }
    for (unsigned i = (vb - lead) * T + tid; i < tot; i += 2 * stride) {
        const unsigned i2 = i + stride;
        const bool two = i2 < tot;
        const unsigned j2 = two ? i2 : i;
        const unsigned a1 = i / Ku, a2 = j2 / Ku;
        const size_t idx = active ? (size_t)active[a1] * K + (i - a1 * Ku) : i;
        const size_t idx2 = active ? (size_t)active[a2] * K + (j2 - a2 * Ku) : j2;
        const double l1 = lambda[idx], l2 = lambda[idx2];
        const double u1 = exp_digamma(l1), u2 = exp_digamma(l2);
        u[idx] = u1;
        if (two)
            u[idx2] = u2;
    }
    pub();
}

template <int T>
__global__ __launch_bounds__(T) void preamble_fused_kernel(
    int K, int V, int G, int wpb, size_t total, const double *__restrict__ lambda,
    double *__restrict__ partial, double *__restrict__ u, const int32_t *__restrict__ active, int GC,
    const double *__restrict__ carry_rows, int carry_n, const double *__restrict__ carry_base,
    double *__restrict__ carry_out)
{
    __shared__ double red[T];
    preamble_fused_body<T>((int)blockIdx.x, (int)gridDim.x, red, K, V, G, wpb, total, lambda, partial, u,
                           active, GC, carry_rows, carry_n, carry_base, carry_out);
}

// c[k] = exp(-psi(sum_b partial[b][k])), k < K <= 128, G <= 64 block partials, formed by a
// workgroup of T = 512 threads in two steps so that the loads can be issued before the
// workgroup's other (dependent) global loads and consumed after them:
//   topic_scale_load    task t = (part, k), part < 8: thread t and t + T each fetch the up to
//                       eight partials of their task (clamped, masked later)
//   topic_scale_partials  adds them in a fixed order into scratch (8 K doubles of LDS)
//   topic_scale_combine   (after a barrier) thread k combines the eight parts of topic k in
//                       order, psi, exp.  Workgroup 0 also leaves psi(sum), the sum and c in
//                       out[0..3K).
// The order of additions is the same in every workgroup, so every document sees the same c.
template <int T>
__device__ __forceinline__ void topic_scale_load(int K, int G, const double *__restrict__ partial,
                                                 double (&v)[2][8])
{
    const int per = (G + 7) / 8;                     // <= 8
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int t = min((int)threadIdx.x + h * T, 8 * K - 1);
        const int k = t % K, b0 = (t / K) * per;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            v[h][q] = partial[(size_t)min(b0 + q, G - 1) * K + k];
    }
}
template <int T>
__device__ __forceinline__ void topic_scale_partials(int K, int G, const double (&v)[2][8],
                                                     double *scratch)
{
    const int per = (G + 7) / 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int t = threadIdx.x + h * T;
        if (t < 8 * K) {
            const int b0 = (t / K) * per, b1 = min(G, b0 + per);
            double acc[2] = {0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 8; ++q)
                acc[q & 1] += (b0 + q < b1) ? v[h][q] : 0.0;
            scratch[t] = acc[0] + acc[1];            // scratch[part * K + k]
        }
    }
}
// after a barrier: thread k < K combines the eight parts of topic k -> c_k
__device__ __forceinline__ double topic_scale_combine(int K, int k, const double *scratch, double *out)
{
    double w[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
        w[q] = scratch[q * K + k];
    const double rs = ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));
    const double ps = digamma(rs);
    const double ck = exp(-ps);
    if (blockIdx.x == 0 && out) {
        out[k] = ps;
        out[K + k] = rs;
        out[2 * K + k] = ck;
    }
    return ck;
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
// Diagnostic builds only (-DTRLDA_STAMPS, tools/stamps.sh): thread 0 of every workgroup adds
// the s_memtime cycles of each segment into stamps[block][8].  Never compiled into the
// shipped library; the values go to a buffer nothing else reads.
#ifdef TRLDA_STAMPS
// segment sums stay in registers; one global write per segment at the very end
#ifndef TRLDA_STAMP_THREAD
#define TRLDA_STAMP_THREAD 0      // whose clock is reported (-DTRLDA_STAMP_THREAD=448: wave 7)
#endif
#define TRLDA_STAMP_DECL unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define TRLDA_STAMP(i)                                                        \
    do {                                                                      \
        unsigned long long now__ = __builtin_amdgcn_s_memtime();             \
        stamp_acc[i] += now__ - stamp_last;                                   \
        stamp_last = now__;                                                   \
    } while (0)
#define TRLDA_STAMP_FLUSH                                                     \
    do {                                                                      \
        if (threadIdx.x == TRLDA_STAMP_THREAD)                                \
            for (int q__ = 0; q__ < 8; ++q__)                                 \
                a.stamps[doc_block(a) * 8 + q__] += stamp_acc[q__];           \
    } while (0)
#else
#define TRLDA_STAMP_DECL do { } while (0)
#define TRLDA_STAMP(i) do { } while (0)
#define TRLDA_STAMP_FLUSH do { } while (0)
#endif

struct DocKernelArgs {
    int K, Kp, n_cap, B;
    unsigned long long *stamps;   // diagnostic builds only, else nullptr
    const int32_t *indptr, *ids, *cnts;
    const int32_t *order;     // optional processing order (long documents first)
    const double *eeb;        // K x V
    const double *alpha;      // K
    const double *gamma_in;   // K x B initial gamma (may alias gamma)
    double *gamma;            // K x B out
    double *epg;              // K x B out: exp(psi(gamma)) of the returned gamma
    double *tw_csr;           // nnz: cnt/phinorm in CSR order (streaming-path scratch)
    const double *scale_in;   // 3 K finished topic factors (psi, row sums, c) or NULL: form them
                              // from `partial` (fused preamble)
    const int32_t *wrank;     // nnz: CSR position -> word-major rank (segmented mode); NULL: the
                              // weights stay in CSR order (data-parallel factor exchange)
    double *tw_word;          // nnz: cnt/phinorm in word-major order (segmented mode)
    double *sstats_acc;       // K x V atomic target (atomic mode) or nullptr
    int max_iter;
    double threshold;
    int32_t *iters_out;       // B or nullptr
    // fused preamble (preamble_fused_kernel): eeb holds exp(psi(lambda)) and the document
    // kernel scales by c_k = exp(-psiSum_k) itself; nullptr = eeb is already exp E[log beta]
    const double *partial;    // G x K block partials of the row sums of lambda
    int G;
    double *scale_out;        // 3K: psi(sum), sum, c (written by workgroup 0)
    // register kernel: per workgroup (document, length, CSR offset, 0) and kRegMaxN padded ids
    const int32_t *pad_meta, *pad_ids;
    // split documents (estep_docs_reg_body<0, true>): meta_i4 = 2 int4 per workgroup, the second
    // one (segment, segments, first exchange row of the document, document length); xbuf holds
    // (max_iter + 1) x segments x K doubles per split document, NaN before the launch
    int meta_i4;
    double *xbuf;
    int *xerr;                // set to 1 when an exchange gave up waiting (never in a sane run)
    // merged launch (estep_merged.h): the statistics are workgroups of THIS launch -- the outputs
    // they read (epg, tw_word) go out with agent-scope stores and every document workgroup counts
    // itself done; the topic factors `scale_in` are being finished by workgroups of this launch
    // and are there when *scale_wait has reached scale_target
    unsigned int *done_counter;           // or nullptr
    unsigned int done_target;             // the workgroup that brings the counter here ...
    unsigned int *go_flags;               // ... writes `epoch` into the n_go flags of the statistics
    int n_go;                             //     workgroups (kMergedFlagStride apart)
    unsigned int epoch;
    const unsigned int *scale_wait;       // this workgroup's flag, or nullptr: scale_in (if any) is
                                          // complete; there when the flag has reached `epoch`
    int block0;                           // workgroups of the grid in front of the documents (a merged
                                          // launch's topic-factor workgroups come FIRST, so that nothing
                                          // a document waits for can be queued behind it); else 0
    int docs_per_wg;                      // 1; 8: a WAVE per document (estep_docs_small_body, K <= 32)
    // ... from document workgroup small_block0 on (a tiered launch: the workgroups before it take one
    // document of more than 128 words each, the batch's first small_first in its sorted order)
    int small_block0, small_first;
};

// the document workgroup's index among the launch's document workgroups
__device__ __forceinline__ int doc_block(const DocKernelArgs &a) { return (int)blockIdx.x - a.block0; }

// an output of the document kernels that the statistics stage reads
__device__ __forceinline__ void merged_store(double *p, double v, bool coherent)
{
    if (coherent)                                    // launch-uniform
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        *p = v;
}

// The finished topic factors of a merged launch: written by workgroups of the same launch
// (merged_combine) that come BEFORE the documents in the grid, i.e. are dispatched before them
// and wait for nothing.  Thread k < K; document workgroup 0 also leaves the three values in
// a.scale_out.
__device__ __forceinline__ double scale_wait_load(const DocKernelArgs &a, int K, int k)
{
    int spins = 0;
    while ((int)(__hip_atomic_load(a.scale_wait + (size_t)doc_block(a) * 16, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT) - a.epoch) < 0) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 22)) {                   // never in a sane run: fail the call, do not hang
            if (a.xerr)
                __hip_atomic_store(a.xerr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
    flag_seen();
    double *in = const_cast<double *>(a.scale_in);
    const double ck = __hip_atomic_load(in + 2 * K + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (doc_block(a) == 0 && a.scale_out) {
        a.scale_out[k] = __hip_atomic_load(in + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.scale_out[K + k] = __hip_atomic_load(in + K + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.scale_out[2 * K + k] = ck;
    }
    return ck;
}

// sum_{i<count} a[i] * b[i * stride] with NA independent accumulators.  A dependent fp64
// fma chain on gfx950 advances one link per ~37 cycles while a wave can issue one every
// ~9 (tools/probes): NA >= 4 chains keep the pipe busy.  Element i goes to chain i % NA and
// the chains are combined pairwise -- a fixed order, so results are reproducible.
template <int NA>
__device__ __forceinline__ double lds_dot(const double *a, const double *b, int stride, int count)
{
    double acc[NA];
#pragma unroll
    for (int u = 0; u < NA; ++u)
        acc[u] = 0.0;
    int i = 0;
    for (; i + NA <= count; i += NA) {
        double x[NA], y[NA];
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            x[u] = a[i + u];
            y[u] = b[(i + u) * stride];
        }
#pragma unroll
        for (int u = 0; u < NA; ++u)
            acc[u] = fma(x[u], y[u], acc[u]);
    }
#pragma unroll
    for (int u = 0; u < NA - 1; ++u)
        if (i + u < count)
            acc[u] = fma(a[i + u], b[(i + u) * stride], acc[u]);
#pragma unroll
    for (int w = NA / 2; w > 0; w >>= 1)
#pragma unroll
        for (int u = 0; u < w; ++u)
            acc[u] += acc[u + w];
    return acc[0];
}

// The fixed point of one document whose beta slice is staged in LDS.
// Work split (W = T/64 wavefronts):
//   product B: wave -> (k block of 64 topics, contiguous j part); partials part[jp][k]
//   product E: wave -> (j block of 64 words,  contiguous k part); partials part[kp][j]
// A wave's lanes read 64 consecutive topics of one word (B) or one topic of 64
// consecutive words at stride Kp (odd) (E): both conflict-free.  The other operand
// (tw_j resp. e_k) is the same LDS address in every lane (broadcast).
template <int T>
__device__ __forceinline__ int doc_fixed_point_lds(
    const DocKernelArgs &a, int n, const int32_t *__restrict__ ids,
    const int32_t *__restrict__ cnts, double *__restrict__ beta, double *__restrict__ g,
    double *__restrict__ e, double *__restrict__ tw, double *__restrict__ cntd,
    double *__restrict__ part, double *__restrict__ wsum)
{
    constexpr int W = T / kWave;
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid = tid / kWave;
    const int K = a.K, Kp = a.Kp;

    // stage the slice: one wave per word, lanes over topics          lda.cpp:179-181
    for (int j = wid; j < n; j += W) {
        const double *src = a.eeb + (size_t)ids[j] * K;
        double *dst = beta + j * Kp;
        for (int k = lane; k < K; k += kWave)
            dst[k] = src[k];
    }
    for (int j = tid; j < n; j += T)
        cntd[j] = (double)cnts[j];

    // wave roles
    const int KB = (K + kWave - 1) / kWave;          // topic blocks
    const int JP = KB <= W ? W / KB : 1;             // j parts (product B)
    const int JC = (n + JP - 1) / JP;
    const int JB = (n + kWave - 1) / kWave;          // word blocks
    const int KPn = (JB > 0 && JB <= W) ? W / JB : 1;  // k parts (product E)
    const int KC = (K + KPn - 1) / KPn;

    auto product_E = [&]() {                         // lda.cpp:183 / :199
        if (JB <= W) {
            if (wid < JB * KPn) {
                const int jb = wid % JB, kp = wid / JB;
                const int j = jb * kWave + lane;
                const int k0 = kp * KC, k1 = min(K, k0 + KC);
                if (j < n)
                    part[kp * n + j] = lds_dot<8>(e + k0, beta + j * Kp + k0, 1, k1 - k0);
            }
            __syncthreads();
            for (int j = tid; j < n; j += T) {
                double s = part[j];
                for (int q = 1; q < KPn; ++q)
                    s += part[q * n + j];
                tw[j] = cntd[j] / (s + 1e-100);
            }
        } else {
            // more word blocks than waves: every wave owns whole words, no partials
            for (int jb = wid; jb < JB; jb += W) {
                const int j = jb * kWave + lane;
                if (j < n)
                    tw[j] = cntd[j] / (lds_dot<8>(e, beta + j * Kp, 1, K) + 1e-100);
            }
        }
        __syncthreads();
    };

    __syncthreads();
    product_E();

    int it = 0;
    while (it < a.max_iter) {                        // lda.cpp:185-204
        // product B: acc_k = sum_j tw_j beta[j][k]                   lda.cpp:189-193
        if (KB <= W) {
            if (wid < KB * JP) {
                const int kb = wid % KB, jp = wid / KB;
                const int k = kb * kWave + lane;
                const int j0 = min(n, jp * JC), j1 = min(n, j0 + JC);
                if (k < K)
                    part[jp * K + k] = lds_dot<8>(tw + j0, beta + j0 * Kp + k, Kp, j1 - j0);
            }
        } else {
            for (int kb = wid; kb < KB; kb += W) {
                const int k = kb * kWave + lane;
                if (k < K)
                    part[k] = lds_dot<8>(tw, beta + k, Kp, n);
            }
        }
        __syncthreads();

        // gamma_k = alpha_k + e_k * acc_k ; e_k = exp(psi(gamma_k))   lda.cpp:194-197
        double diff = 0.0;
        for (int k = tid; k < K; k += T) {
            double acc = part[k];
            for (int q = 1; q < JP; ++q)
                acc += part[q * K + k];
            const double gnew = acc * e[k] + a.alpha[k];
            diff += fabs(g[k] - gnew);
            g[k] = gnew;
            e[k] = exp_digamma(gnew);
        }
        diff = wave_sum(diff);
        if (lane == 0)
            wsum[wid] = diff;
        __syncthreads();

        double change = 0.0;                         // lda.cpp:202-203
#pragma unroll
        for (int q = 0; q < W; ++q)
            change += wsum[q];

        product_E();                                 // ends with a barrier
        ++it;
        if (change / (double)K < a.threshold)
            break;
    }
    return it;
}

// Streaming variant for documents whose slice exceeds the LDS budget: beta is re-read
// from eeb (L2 / Infinity Cache) in both products, lanes over topics so that every
// access is a coalesced K-vector; tw lives in global scratch.
template <int T>
__device__ __forceinline__ int doc_fixed_point_stream(
    const DocKernelArgs &a, int n, const int32_t *__restrict__ ids,
    const int32_t *__restrict__ cnts, double *__restrict__ g, double *__restrict__ e,
    double *__restrict__ tw, double *__restrict__ part, double *__restrict__ wsum)
{
    constexpr int W = T / kWave;
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid = tid / kWave;
    const int K = a.K;
    const int kslots = min((K + kWave - 1) / kWave * kWave, T);
    const int jparts = T / kslots;
    const int ks = tid % kslots, jp = tid / kslots;

    auto product_E = [&]() {
        for (int j = wid; j < n; j += W) {
            const double *col = a.eeb + (size_t)ids[j] * K;
            double s = 0.0;
            for (int k = lane; k < K; k += kWave)
                s = fma(e[k], col[k], s);
            s = wave_sum(s);
            if (lane == 0)
                tw[j] = (double)cnts[j] / (s + 1e-100);
        }
        __syncthreads();
    };

    __syncthreads();
    product_E();

    int it = 0;
    while (it < a.max_iter) {
        for (int k = ks; k < K; k += kslots) {
            if (jp < jparts) {
                double acc = 0.0;
                for (int j = jp; j < n; j += jparts)
                    acc = fma(tw[j], a.eeb[(size_t)ids[j] * K + k], acc);
                part[jp * K + k] = acc;
            }
        }
        __syncthreads();
        double diff = 0.0;
        for (int k = tid; k < K; k += T) {
            double acc = part[k];
            for (int q = 1; q < jparts; ++q)
                acc += part[q * K + k];
            const double gnew = acc * e[k] + a.alpha[k];
            diff += fabs(g[k] - gnew);
            g[k] = gnew;
            e[k] = exp_digamma(gnew);
        }
        diff = wave_sum(diff);
        if (lane == 0)
            wsum[wid] = diff;
        __syncthreads();
        double change = 0.0;
#pragma unroll
        for (int q = 0; q < W; ++q)
            change += wsum[q];
        product_E();
        ++it;
        if (change / (double)K < a.threshold)
            break;
    }
    return it;
}

template <int T>
__global__ __launch_bounds__(T) void estep_docs_kernel(DocKernelArgs a)
{
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid = tid / kWave;
    constexpr int W = T / kWave;

    const int d = a.order ? a.order[doc_block(a)] : doc_block(a);
    const int K = a.K;
    const int p0 = a.indptr[d];
    const int n = a.indptr[d + 1] - p0;
    const bool staged = n <= a.n_cap;
    const int32_t *ids = a.ids + p0;
    const int32_t *cnts = a.cnts + p0;

    // LDS carve-up (beta first: its size depends on the launch's n_cap)
    double *beta = lds;
    double *g = beta + a.n_cap * a.Kp;
    double *e = g + K;
    double *tw_l = e + K;
    double *cntd = tw_l + a.n_cap;
    double *part = cntd + a.n_cap;
    double *wsum = part + (K > T ? K : T);

    [[maybe_unused]] unsigned long long stamp_last = 0;
#ifdef TRLDA_STAMPS
    stamp_last = __builtin_amdgcn_s_memtime();
#endif
    double *gamma_d = a.gamma + (size_t)d * K;
    const double *gamma0_d = a.gamma_in + (size_t)d * K;
    for (int k = tid; k < K; k += T) {               // lda.cpp:174
        const double gk = gamma0_d[k];
        g[k] = gk;
        e[k] = exp_digamma(gk);
    }

    int it;
    if (staged)
        it = doc_fixed_point_lds<T>(a, n, ids, cnts, beta, g, e, tw_l, cntd, part, wsum);
    else
        it = doc_fixed_point_stream<T>(a, n, ids, cnts, g, e, a.tw_csr + p0, part, wsum);

    // results
    for (int k = tid; k < K; k += T) {
        gamma_d[k] = g[k];
        a.epg[(size_t)d * K + k] = e[k];
    }
    if (tid == 0 && a.iters_out)
        a.iters_out[d] = it;

    if (a.sstats_acc) {                              // lda.cpp:207-213, atomic form
        for (int j = wid; j < n; j += W) {
            const double c = staged ? tw_l[j] : a.tw_csr[p0 + j];
            double *col = a.sstats_acc + (size_t)ids[j] * K;
            for (int k = lane; k < K; k += kWave)
                unsafeAtomicAdd(&col[k], c * e[k]);
        }
    } else {
        for (int j = tid; j < n; j += T)
            a.tw_word[a.wrank ? a.wrank[p0 + j] : p0 + j] = staged ? tw_l[j] : a.tw_csr[p0 + j];
    }
}

// ---------------------------------------------------------------------------
// 3c. Register-resident kernel: K <= 128 topics, documents of at most 128 words (192 with
// in the 144-word variant).
//
// tools/probes on gfx950: with the slice in LDS the two products are bound by LDS bandwidth
// -- every iteration re-reads the 8*K*n-byte slice twice, and each broadcast operand costs a
// full 512-byte LDS access per wave.  Here the slice lives in VGPRs, in BOTH orientations
// (2 x 64 doubles per thread, 512 threads), read from eeb once; per iteration LDS only
// carries the two K- resp. n-vectors and the partial sums.
//
//   lane l owns topics l and l+64 and words l and l+64
//   wave w (0..7), product B: words [16 w, 16 w + 16) of the document
//                  product E: topics [w*KC, (w+1)*KC),               KC <= 16
//                  psi      : waves 0 and 1, topics 0..63 and 64..127
// ---------------------------------------------------------------------------
constexpr int kRegThreads = 512;
constexpr int kRegMaxK = 128;
constexpr int kRegStride = 129;    // LDS row stride of the transposition / tail buffer (odd)
constexpr int kRegPart = 192;      // row length of the partial-sum arrays
// (kRegMaxN = 144 words of the longest register variant, 8 waves x 18; documents of more than kSplitMinN
// = 192 words are split into ceil(n / kSplitSegN) segments of at most 128 words, up to kSplitMaxSeg = 16
// of them: index_params.h)

// g[2] | alpha | e[2] (+16 zero pad each) | tw (+16 zero pad) | cnt | part[8][192] | misc[8] |
// buffer [128][129]: transposition scratch while the registers are filled, then the rows
// of words 128.. of a long document.  g and e are double-buffered (iteration parity) so
// that the wave forming mean|gamma - last| can read the old values while the two waves that
// own the topics write the new ones -- no barrier in between.
constexpr int kRegSmallDoubles = 2 * 128 + 2 * 128 + 2 * 144 + 208 + 192 + 8 * kRegPart + 8;
constexpr size_t kRegLdsBytes = (size_t)(kRegSmallDoubles + 128 * kRegStride) * sizeof(double);

// Sum over the 64 lanes of a wave in 6 DPP steps (row shifts, then row broadcasts); the total
// ends up in lane 63 and is returned to every lane through two v_readlane.  __shfl_down on a
// double goes through ds_bpermute, i.e. an LDS round trip per step.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    // lanes outside ROW_MASK (or shifted in from outside the row) receive +0.0
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v)
{
    v = dpp_add<0x111, 0xf>(v);      // row_shr:1
    v = dpp_add<0x112, 0xf>(v);      // row_shr:2
    v = dpp_add<0x114, 0xf>(v);      // row_shr:4
    v = dpp_add<0x118, 0xf>(v);      // row_shr:8  -> lane 15 of every row holds the row sum
    v = dpp_add<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3 -> lane 63 = total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------
// Transposing butterfly steps.  fold<D>(a, b): lanes whose bit D of the lane number is clear
// return a(l) + a(l ^ D), the others b(l ^ D) + b(l).  fold<D>(v, v) is the plain all-reduce
// step.  D = 32, 16: gfx950's v_permlane{32,16}_swap; D = 8, 4: DPP row rotation / shifts with
// bank masks; D = 2, 1: quad permutes after a select.
// ---------------------------------------------------------------------------
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ double dpp_merge(double old, double src)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xf,
                                               BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xf,
                                               BANK_MASK, false);
    return __hiloint2double(hi, lo);
}

template <int D>
__device__ __forceinline__ double fold(double a, double b)
{
    if constexpr (D == 32) {
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a),
                                                         (unsigned)__double2loint(b), false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a),
                                                         (unsigned)__double2hiint(b), false, false);
        return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    } else if constexpr (D == 16) {
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a),
                                                         (unsigned)__double2loint(b), false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a),
                                                         (unsigned)__double2hiint(b), false, false);
        return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    } else if constexpr (D == 8) {
        const double a2 = dpp_merge<0x128, 0xc>(a, b);   // row_ror:8 into lanes 8..15: b(l-8)
        const double b2 = dpp_merge<0x128, 0x3>(b, a);   // row_ror:8 into lanes 0..7 : a(l+8)
        return a2 + b2;
    } else if constexpr (D == 4) {
        const double a2 = dpp_merge<0x114, 0xa>(a, b);   // row_shr:4 into quads 1, 3: b(l-4)
        const double b2 = dpp_merge<0x104, 0x5>(b, a);   // row_shl:4 into quads 0, 2: a(l+4)
        return a2 + b2;
    } else {
        static_assert(D == 2 || D == 1, "fold distance");
        const bool up = (threadIdx.x & D) != 0;
        const double keep = up ? b : a, send = up ? a : b;
        constexpr int CTRL = D == 2 ? 0x4e : 0xb1;       // quad_perm [2,3,0,1] / [1,0,3,2]
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(send), CTRL, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(send), CTRL, 0xf, 0xf, true);
        return keep + __hiloint2double(hi, lo);
    }
}

// acc += vals(lane N of this lane's ROW of 16) * b -- v_fmac_f64 with the DPP control row_newbcast:N
// (gfx90a+: the one DPP form 64-bit operands have, made for exactly this).  A wave's sixteen
// weights sit in one register (lane l holds weight l & 15, the same in all four rows) and are handed
// out inside the vector ALU; as eight broadcast ds_read_b128 per wave and product they were 64 kB of
// LDS traffic per stage and workgroup -- the LDS pipe, 512 of a stage's ~800 cycles, was what bounded
// the two products.  `vals` comes straight from an LDS load and is never written by
// the vector ALU, so the VALU-write -> DPP-read hazard does not arise; the s_nop in front of the
// FIRST use of a register covers a copy the register allocator might put there.
template <int N, bool FIRST = false>
__device__ __forceinline__ void fmac_row_bcast(double &acc, double vals, double b)
{
    static_assert(N >= 0 && N < 16, "a lane of the row");
    if constexpr (FIRST)
        asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
            : "+v"(acc)
            : "v"(vals), "v"(b), "n"(N));
    else
        asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(vals), "v"(b), "n"(N));
}
// The weights tw_j of product B formed by the wave that consumes them, from product E's partials (no
// LDS round trip, one barrier fewer per iteration): measured and NOT adopted -- in the plain register
// kernel a document goes from 67.5 k to 65.8 k cycles, inside the deferred and merged launches (other
// register allocation, all eight waves on the LDS pipe at once) from 27.2 to 27.5 us; same box:
// two lanes 26.8 = 26.8 us per step, one lane 31.55 against 31.26, update_parameters(TR 10) 0.465
// against 0.458 ms, B = 1600 2.09 against 2.11 (profiles/r05_wave_weights_ab.txt).  -DTRLDA_WAVE_WEIGHTS
#ifdef TRLDA_WAVE_WEIGHTS
constexpr bool kWaveWeights = true;                  // (estep_docs_reg_body, wave_weights)
#else
constexpr bool kWaveWeights = false;                 // a thread per word, through LDS and a barrier
#endif
#ifndef TRLDA_NO_DPP_BCAST
constexpr bool kDppBcast = true;
#else
constexpr bool kDppBcast = false;                    // (A/B: the broadcast reads of rounds 1-4)
#endif

// eight partial sums at a compile-time stride, all reads in flight, pairwise combine
template <int STRIDE>
__device__ __forceinline__ double sum8_strided(const double *p)
{
    const double v0 = p[0], v1 = p[STRIDE], v2 = p[2 * STRIDE], v3 = p[3 * STRIDE],
                 v4 = p[4 * STRIDE], v5 = p[5 * STRIDE], v6 = p[6 * STRIDE], v7 = p[7 * STRIDE];
    return ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7));
}

// MODE 0: documents of at most 128 words.  MODE 1: at most 144 words, all in registers (18
// words per wave, a third register block for the words 128..143 in the second orientation).
// Longer documents: estep_wide.h (single orientation), or split over several workgroups.
// The preamble of the NEXT batch as extra workgroups of this launch: a 200-document batch leaves
// 56 of the 256 CUs idle for the ~31 us the document workgroups run, and consecutive E-steps on an
// unchanged lambda are independent of each other -- so the workgroups past the documents fill
// exp(psi(lambda)) and the row-sum partials for the batch that the host announced as the next
// one, into the model's alternate buffers; that batch's E-step then starts with its document
// kernel.  Everything stays on one stream: no cross-stream hand-off (which was measured and lost).
struct PreArgs {
    int n_docs;               // workgroups [0, n_docs) are documents
    int nb;                   // workgroups [n_docs, n_docs + nb) do the preamble; 0 = none
    int K, V, G, wpb;
    size_t total;
    const double *lambda;
    double *partial, *u;
    const int32_t *active;
    double *c_out;            // 3 K: the finished topic factors of that batch's preamble
    unsigned int *c_counter;
};

// the preamble workgroups of a document-kernel launch (block >= pre.n_docs, counted from the
// first document workgroup)
template <bool FAT = false, class Pub = NoPublish>
__device__ __forceinline__ void docs_launch_preamble(const PreArgs &pre, double *lds, int block,
                                                     const Pub &pub = Pub(), int tid_bias = 0)
{
    preamble_fused_body<kRegThreads, FAT, Pub>(block - pre.n_docs, pre.nb, lds, pre.K, pre.V, pre.G,
                                     pre.wpb, pre.total, pre.lambda, pre.partial, pre.u, pre.active,
                                     0, nullptr, 0, nullptr, nullptr, pre.c_out, pre.c_counter,
                                     lds + kRegThreads, pub, tid_bias);
}

// One document = one workgroup: the body of estep_docs_reg_kernel<MODE>, also instantiated by
// the tiered kernel of estep_wide.h, where every workgroup takes the variant its own document
// needs.  The document is the one of a.pad_meta[blockIdx.x].
// SPLIT: this workgroup holds ONE SEGMENT (at most 128 words) of a document that several
// workgroups share.  Everything a word needs of the other words goes through gamma: every
// iteration each segment publishes the K sums acc_k over its own words, all segments add up the
// published rows in segment order (bitwise the same gamma and exp(psi(gamma)) everywhere, so the
// same decision to stop), and go on with their own words.  The rows travel through global memory
// with agent-scope (sc1) accesses -- workgroups of a launch sit on different XCDs -- into a
// buffer with one row per (document, iteration, segment) that starts out as NaN: a row is there
// when it is not NaN any more (tools/probes/xcu_probe: 0.6-0.9 us per exchange, against 2-3 with
// a flag per row).  Segments of a document are consecutive workgroups at the front of the grid
// (long documents first), so they are resident together.
template <int MODE, bool SPLIT = false>
__device__ __forceinline__ void estep_docs_reg_body(const DocKernelArgs &a, double *lds)
{
    static_assert(MODE == 0 || MODE == 1, "register variants: 128 or 144 words");
    static_assert(!SPLIT || MODE == 0, "segments hold at most 128 words");
    constexpr bool MID = MODE == 1;
    constexpr int JC = MID ? 18 : 16;                // words per wave: wave w owns [JC w, JC w + JC)
    constexpr int NREG = 8 * JC;                     // words held in registers
    constexpr int T = kRegThreads, W = T / kWave;    // 8 waves
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(tid / kWave);

    const int K = a.K;
    // fused preamble: the block partials of the row sums are fetched first -- they depend on
    // nothing -- and consumed once the row loads below are in flight
    double pv[2][8];
    if (a.partial && !a.scale_in)                    // launch-uniform
        topic_scale_load<T>(K, a.G, a.partial, pv);

    // One load gives (document, length, CSR offset); the word ids sit at an address that
    // depends on the workgroup index only, so they are fetched at the same time: two dependent
    // memory latencies (descriptor | ids -> rows) instead of four (order -> indptr -> ids -> rows)
    const int bid = doc_block(a);
    const int4 meta = reinterpret_cast<const int4 *>(a.pad_meta)[(size_t)bid * a.meta_i4];
    [[maybe_unused]] int4 seg = make_int4(0, 1, 0, 0);
    if constexpr (SPLIT)
        seg = reinterpret_cast<const int4 *>(a.pad_meta)[(size_t)bid * a.meta_i4 + 1];
    const int32_t *__restrict__ pids = a.pad_ids + (size_t)bid * kRegMaxN;
    const int myid = pids[wid * JC + min(lane, JC - 1)];   // word wid * JC + i of the document
    const int d = meta.x, n = meta.y, p0 = meta.z;
    const int32_t *__restrict__ cnts = a.cnts + p0;
    // gamma0 / alpha of topic tid: requested BEFORE the 32 row loads below -- loads return in
    // order, so a request behind them could not be consumed (exp(psi(gamma0)), ~700 cycles on the
    // two waves that own the topics) until every row had landed
    double gk0 = 1.5, ak0 = 0.0, ck0 = 1.0;
    if (tid < K) {
        gk0 = a.gamma_in[(size_t)d * K + tid];
        ak0 = a.alpha[tid];
        if (a.scale_in && !a.scale_wait) {           // finished by the launch that prepared them
            ck0 = a.scale_in[2 * K + tid];
            if (bid == 0 && a.scale_out) {
                a.scale_out[tid] = a.scale_in[tid];
                a.scale_out[K + tid] = a.scale_in[K + tid];
                a.scale_out[2 * K + tid] = ck0;
            }
        }
    }

    double *gbuf = lds;               // 2 x 128
    double *alpha_l = gbuf + 256;     // 128
    double *c_l = alpha_l + 128;      // 128: exp(-psiSum_k) (fused preamble) or 1
    double *ebuf = c_l + 128;         // 2 x 144
    double *tw = ebuf + 288;          // 208
    double *cntd = tw + 208;          // 192
    double *part = cntd + 192;        // 8 x 192
    double *misc = part + 8 * kRegPart;    // 8
    double *tbuf = misc + 8;          // 128 x 129

    [[maybe_unused]] unsigned long long stamp_last = 0;
    TRLDA_STAMP_DECL;
#ifdef TRLDA_STAMPS
    stamp_last = __builtin_amdgcn_s_memtime();
#endif

    const int nm = min(n, NREG);                     // words held in registers
    const int KC = (((K + W - 1) / W) + 1) & ~1;     // topics per wave (even), <= 16
    const int j0 = wid * JC, k0 = wid * KC;
    const bool k_lo = lane < K, k_hi = lane + 64 < K;

    // ---- the slice: orientation B straight from eeb (coalesced K-vectors)  lda.cpp:179-181
    // Word ids of this wave's 16 rows: one vector load (lane i -> row i), handed to every lane
    // by v_readlane -- one memory latency for the ids, then all 32 row loads are in flight
    // while the gamma / alpha / count initialisation below runs.  Every load is unconditional
    // (clamped indices); out-of-range elements are zeroed after the loads have landed.
    double bB0[JC], bB1[JC];          // beta[j0+i][lane], beta[j0+i][lane+64]
    {
        const int kl = min(lane, K - 1), kh = min(lane + 64, K - 1);
#pragma unroll
        for (int i = 0; i < JC; ++i) {
            const double *rowp = a.eeb + (size_t)__builtin_amdgcn_readlane(myid, i) * K;
            bB0[i] = rowp[kl];
            bB1[i] = rowp[kh];
        }
    }

    // ---- MODE 1: the words 128..143 in the second orientation.  Only 16 words: four lane groups
    // share them, each with four of the wave's 16 topics -- lane l holds word 128 + (l & 15), topics
    // k0 + 4 (l >> 4) + {0..3}: 4 doubles per lane instead of 16 that three quarters of the lanes
    // would not use.  They are read from eeb directly, 32 contiguous bytes per lane of rows that
    // wave 7 requests anyway (round 5; through the transposition buffer they cost a second pass
    // with two more barriers: staging 12.6 k cycles against 10.7 k, profiles/r05_stamps_modes.txt).
    [[maybe_unused]] double bE2[4] = {0.0, 0.0, 0.0, 0.0};
    [[maybe_unused]] const int kq = 4 * (lane >> 4); // this lane group's topic offset
    if constexpr (MID) {
        const int KCm = (((K + W - 1) / W) + 1) & ~1;
        const int k2 = wid * KCm + kq;               // first of this lane's four topics
        const int id2 = pids[128 + (lane & 15)];
        const double *row2 = a.eeb + (size_t)id2 * K;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            bE2[i] = row2[min(k2 + i, K - 1)];
    }

    double *gamma_d = a.gamma + (size_t)d * K;
    double e0 = 0.0;                                 // thread k < K: exp(psi(gamma0_k))
    if (tid < 144) {                                 // lda.cpp:174
        if (tid < K) {
            gbuf[tid] = gk0;
            alpha_l[tid] = ak0;
            e0 = exp_digamma(gk0);
        }
        ebuf[144 + tid] = 0.0;                       // zero beyond K, in both buffers
        if (!a.partial || tid >= K)
            ebuf[tid] = e0;
    }
    if (a.partial && !a.scale_in)                    // launch-uniform; `part` is idle until
        topic_scale_partials<T>(K, a.G, pv, part);   // the first product
    for (int j = tid; j < 208; j += T) {
        tw[j] = 0.0;
        if (j < 192)
            cntd[j] = j < n ? (double)cnts[j] : 0.0;
        if (j < 8)
            misc[j] = 0.0;                           // [2], [3]: an exchange gave up (SPLIT)
    }
#pragma unroll
    for (int i = 0; i < JC; ++i) {
        const bool row = j0 + i < nm;
        bB0[i] = (row && k_lo) ? bB0[i] : 0.0;
        bB1[i] = (row && k_hi) ? bB1[i] : 0.0;
    }

    // ---- orientation E through an LDS transposition (stride 129: conflict-free both ways)
#pragma unroll
    for (int i = 0; i < JC; ++i) {
        if (j0 + i < 128) {                          // wave-uniform; rows >= n hold zeros
            tbuf[(j0 + i) * kRegStride + lane] = bB0[i];
            tbuf[(j0 + i) * kRegStride + 64 + lane] = bB1[i];
        }
    }
    __syncthreads();
    // fused preamble: thread k forms c_k = exp(-psiSum_k) from the partial sums parked in
    // `part` and scales its exp(psi(gamma0_k)) -- while the other waves read the transposition
    if (tid < K) {
        double ck = 1.0;
        if (a.partial) {                             // launch-uniform
            // (merged launch: being finished by workgroups of this launch while the slice was staged)
            ck = a.scale_in ? (a.scale_wait ? scale_wait_load(a, K, tid) : ck0)
                            : topic_scale_combine(K, tid, part, a.scale_out);
            ebuf[tid] = e0 * ck;
        }
        c_l[tid] = ck;
    }
    double bE0[16], bE1[16];          // beta[lane][k0+i], beta[lane+64][k0+i]
    {
        const double *t0 = tbuf + lane * kRegStride + k0;
        const double *t1 = tbuf + (lane + 64) * kRegStride + k0;
        constexpr bool r_lo = true, r_hi = true;     // all 128 rows were written
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool colv = i < KC && k0 + i < K;  // wave-uniform
            const double v0 = t0[colv ? i : 0], v1 = t1[colv ? i : 0];
            bE0[i] = (colv && r_lo) ? v0 : 0.0;
            bE1[i] = (colv && r_hi) ? v1 : 0.0;
        }
    }
    __syncthreads();
    if constexpr (MID) {                             // (out-of-range words and topics: zero)
        const bool row2_on = 128 + (lane & 15) < nm;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool colv = kq + i < KC && k0 + kq + i < K;
            bE2[i] = (colv && row2_on) ? bE2[i] : 0.0;
        }
    }
    TRLDA_STAMP(1);
    // (kWaveWeights: the partial sums of product B live in the transposition buffer, which is free
    // from here on -- the waves of product B read product E's partials while others already write
    // their own)
    double *partB = kWaveWeights ? tbuf : part;      // 8 x 192

    // phinorm_j = sum_k e_k beta[j][k]: its eight partials per word     lda.cpp:183 / :199
    auto product_E = [&](const double *e) {
        // k0 is even and e is 16-byte aligned: one ds_read_b128 broadcasts two weights.  All
        // eight reads are issued before the first fma and nothing is branched over: weights
        // past K are zero in LDS and the matching registers are zero.  Eight chains.
        const double2 *ep = reinterpret_cast<const double2 *>(e + k0);
        [[maybe_unused]] double2 ew[8];
        [[maybe_unused]] double evals = 0.0, evals2 = 0.0;
        if constexpr (kDppBcast) {
            evals = e[k0 + (lane & 15)];             // (16-lane rows: every row holds the wave's 16 weights)
            if constexpr (MID)
                evals2 = e[k0 + kq + (lane & 3)];    // row q: its own four
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                ew[i] = ep[i];
        }
        [[maybe_unused]] double s2 = 0.0;
        if constexpr (MID && kDppBcast) {            // words 128 + (lane & 15): the row's four topics
            double s2b = 0.0;
            fmac_row_bcast<0, true>(s2, evals2, bE2[0]);
            fmac_row_bcast<1>(s2b, evals2, bE2[1]);
            fmac_row_bcast<2>(s2, evals2, bE2[2]);
            fmac_row_bcast<3>(s2b, evals2, bE2[3]);
            s2 += s2b;
            s2 = fold<32>(s2, s2);                   // + the other lane groups
            s2 = fold<16>(s2, s2);
        } else if constexpr (MID) {                  // words 128 + (lane & 15)
            // the lane group's four weights (not a broadcast: they differ between groups); the
            // dependent chain of the two folds is started first and runs under the products
#ifndef TRLDA_EXPT_M1_NOS2                           // (timing experiments: results are wrong)
            const double2 *eq = reinterpret_cast<const double2 *>(e + k0 + kq);
            const double2 ea = eq[0], eb = eq[1];
            s2 = fma(ea.x, bE2[0], ea.y * bE2[1]) + fma(eb.x, bE2[2], eb.y * bE2[3]);
#ifndef TRLDA_EXPT_M1_NOFOLD
            s2 = fold<32>(s2, s2);                   // + the other lane groups
            s2 = fold<16>(s2, s2);
#endif
#endif
        }
        double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};
        if constexpr (kDppBcast) {
#define TRLDA_E_STEP(t, F)                                       \
    fmac_row_bcast<t, F>(s0[(t) & 3], evals, bE0[t]);            \
    fmac_row_bcast<t>(s1[(t) & 3], evals, bE1[t]);
            TRLDA_E_STEP(0, true) TRLDA_E_STEP(1, false) TRLDA_E_STEP(2, false) TRLDA_E_STEP(3, false)
            TRLDA_E_STEP(4, false) TRLDA_E_STEP(5, false) TRLDA_E_STEP(6, false) TRLDA_E_STEP(7, false)
            TRLDA_E_STEP(8, false) TRLDA_E_STEP(9, false) TRLDA_E_STEP(10, false) TRLDA_E_STEP(11, false)
            TRLDA_E_STEP(12, false) TRLDA_E_STEP(13, false) TRLDA_E_STEP(14, false) TRLDA_E_STEP(15, false)
#undef TRLDA_E_STEP
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                s0[(2 * i) & 3] = fma(ew[i].x, bE0[2 * i], s0[(2 * i) & 3]);
                s1[(2 * i) & 3] = fma(ew[i].x, bE1[2 * i], s1[(2 * i) & 3]);
                s0[(2 * i + 1) & 3] = fma(ew[i].y, bE0[2 * i + 1], s0[(2 * i + 1) & 3]);
                s1[(2 * i + 1) & 3] = fma(ew[i].y, bE1[2 * i + 1], s1[(2 * i + 1) & 3]);
            }
        }
        part[wid * kRegPart + lane] = (s0[0] + s0[1]) + (s0[2] + s0[3]);
        part[wid * kRegPart + 64 + lane] = (s1[0] + s1[1]) + (s1[2] + s1[3]);
        if constexpr (MID) {
            if (lane < 16)
                part[wid * kRegPart + 128 + lane] = s2;
        }
        __syncthreads();
        if constexpr (!kWaveWeights) {
            if (tid < 128 || tid < n)
                tw[tid] = cntd[tid] * rcp_pos<true>(sum8_strided<kRegPart>(part + tid) + 1e-100);
            __syncthreads();
        }
    };
    // tw_j = cnt_j / phinorm_j for the words of THIS wave's product B (lane l: word j0 + (l & 15), or
    // j0 + min(l, 17) in the 144-word variant): read back from LDS, where one thread per word has left
    // them behind a barrier -- or (kWaveWeights, not adopted) formed by the consuming wave from the
    // eight partials with the same additions, reciprocal and product.  0 beyond n (cnt is 0).
    auto wave_weights = [&]() {
        const int jl = j0 + (MID ? min(lane, JC - 1) : (lane & 15));
        if constexpr (!kWaveWeights)
            return tw[jl];
        return cntd[jl] * rcp_pos<true>(sum8_strided<kRegPart>(part + jl) + 1e-100);
    };

    product_E(ebuf);
    TRLDA_STAMP(2);

    const int k_psi = (wid & 1) * 64 + lane;         // topic of this lane in the psi stage
    const bool psi_on = k_psi < K;
    // e is kept as c_k exp(psi(gamma_k)) throughout (c = 1 without the fused preamble)
    const double c_psi = c_l[psi_on ? k_psi : 0];

    const double thresholdK = a.threshold * (double)K;
    int it = 0;
    while (it < a.max_iter) {                        // lda.cpp:185-204
        // acc_k = sum_j tw_j beta[j][k]                               lda.cpp:189-193
        {
            double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0};
            // (the 144-word variant keeps its broadcast reads here: with the DPP form the tiered kernels,
            // at the register limit, spilled and this stage went from 840 to 939 cycles)
            const double tvals_w = wave_weights();
#ifdef TRLDA_M1_DPP_B
            constexpr bool dpp_b = kDppBcast;        // (A/B: also in the 144-word variant)
#else
            constexpr bool dpp_b = kDppBcast && !MID;
#endif
            if constexpr (dpp_b) {
                const double tvals = MID ? tw[j0 + min(lane & 15, JC - 1)] : tvals_w;
                [[maybe_unused]] double tvals2 = 0.0;
                if constexpr (MID)
                    tvals2 = tw[j0 + 16 + (lane & 1)];           // words 16, 17 of the wave
#define TRLDA_B_STEP(w, F)                                       \
    fmac_row_bcast<w, F>(a0[(w) & 3], tvals, bB0[w]);            \
    fmac_row_bcast<w>(a1[(w) & 3], tvals, bB1[w]);
                TRLDA_B_STEP(0, true) TRLDA_B_STEP(1, false) TRLDA_B_STEP(2, false) TRLDA_B_STEP(3, false)
                TRLDA_B_STEP(4, false) TRLDA_B_STEP(5, false) TRLDA_B_STEP(6, false) TRLDA_B_STEP(7, false)
                TRLDA_B_STEP(8, false) TRLDA_B_STEP(9, false) TRLDA_B_STEP(10, false) TRLDA_B_STEP(11, false)
                TRLDA_B_STEP(12, false) TRLDA_B_STEP(13, false) TRLDA_B_STEP(14, false) TRLDA_B_STEP(15, false)
#undef TRLDA_B_STEP
                if constexpr (MID) {
                    fmac_row_bcast<0, true>(a0[0], tvals2, bB0[16]);
                    fmac_row_bcast<0>(a1[0], tvals2, bB1[16]);
                    fmac_row_bcast<1>(a0[1], tvals2, bB0[17]);
                    fmac_row_bcast<1>(a1[1], tvals2, bB1[17]);
                }
            } else {
                if constexpr (kWaveWeights) {
                    // (the wave's own words through LDS: one wave's LDS traffic is in order, no barrier)
                    if (lane < JC)
                        tw[j0 + lane] = tvals_w;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                const double2 *tp = reinterpret_cast<const double2 *>(tw + j0);
                double2 tv[JC / 2];
#pragma unroll
                for (int i = 0; i < JC / 2; ++i)
                    tv[i] = tp[i];
#pragma unroll
                for (int i = 0; i < JC / 2; ++i) {
                    a0[(2 * i) & 3] = fma(tv[i].x, bB0[2 * i], a0[(2 * i) & 3]);
                    a1[(2 * i) & 3] = fma(tv[i].x, bB1[2 * i], a1[(2 * i) & 3]);
                    a0[(2 * i + 1) & 3] = fma(tv[i].y, bB0[2 * i + 1], a0[(2 * i + 1) & 3]);
                    a1[(2 * i + 1) & 3] = fma(tv[i].y, bB1[2 * i + 1], a1[(2 * i + 1) & 3]);
                }
            }
            partB[wid * kRegPart + lane] = (a0[0] + a0[1]) + (a0[2] + a0[3]);
            partB[wid * kRegPart + 64 + lane] = (a1[0] + a1[1]) + (a1[2] + a1[3]);
        }
        __syncthreads();
        TRLDA_STAMP(3);

        // gamma_k = alpha_k + e_k * acc_k ; e_k = exp(psi(gamma_k))     lda.cpp:194-197
        // Two waves (topics 0..63, 64..127).  tools/probes/probe6: the whole log-free
        // exp(psi) chain is ~670 cycles for one thread, no more than its slowest piece plus
        // the exchange when it is spread over four waves -- so it is not split.
        const double *g_old = gbuf + (it & 1) * 128, *e_old = ebuf + (it & 1) * 144;
        double *g_new = gbuf + ((it + 1) & 1) * 128, *e_new = ebuf + ((it + 1) & 1) * 144;
        if (wid < 2) {
            const int kk = psi_on ? k_psi : 0;
            // (keeping the lane's own e_k / alpha_k in registers across iterations instead of these two
            // LDS reads was measured: no gain, and the tiered kernels, at the register limit, spilled)
            const double ek = e_old[kk], ak = alpha_l[kk];
            double acc = sum8_strided<kRegPart>(partB + kk);
            if constexpr (SPLIT) {
                // this segment's row out, every segment's row in (segment order)
                double *rows = a.xbuf + ((size_t)seg.z * (size_t)(a.max_iter + 1) +
                                         (size_t)it * (size_t)seg.y) * (size_t)K;
                // (a row that has not arrived is the bit pattern the buffer was filled with, all
                // ones; a sum that IS NaN -- NaN / inf in lambda or in a caller's gamma0 -- is
                // published in the canonical form, so that it arrives like any other number and
                // the document's results are NaN at once, as in the reference, lda.cpp:176-204)
                if (psi_on)
                    __hip_atomic_store(rows + (size_t)seg.x * K + k_psi,
                                       acc == acc ? acc : __longlong_as_double(0x7FF8000000000000ll),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // eight rows at a time: all their loads in flight, then again for those that
                // were not there yet
                double total = 0.0;
                for (int c0 = 0; c0 < seg.y; c0 += 8) {          // wave-uniform bounds
                    double v[8];
                    const int nc = min(8, seg.y - c0);
                    unsigned need = (1u << nc) - 1u;             // rows not seen yet
                    for (int spins = 0;; ++spins) {
#pragma unroll
                        for (int c = 0; c < 8; ++c)
                            if (need >> c & 1u)
                                v[c] = __hip_atomic_load(rows + (size_t)(c0 + c) * K + kk, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
                        unsigned still = 0u;
#pragma unroll
                        for (int c = 0; c < 8; ++c)
                            if ((need >> c & 1u) && __double_as_longlong(v[c]) == -1ll)
                                still |= 1u << c;
                        need = still;
                        if (!__any(need != 0u))                  // wave-uniform: every lane has every row
                            break;
                        // a peer never came (or another workgroup of the launch has given up
                        // already: looked at now and then): give up, loudly -- the whole workgroup
                        // leaves the iteration loop after this stage's barrier, so the launch ends
                        // after ONE wait, whatever max_iter is
                        const bool lost = spins > (1 << 21) ||
                                          ((spins & 1023) == 1023 &&
                                           __hip_atomic_load(a.xerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0);
                        if (lost) {
                            if (lane == 0) {
                                __hip_atomic_store(a.xerr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                misc[2 + wid] = 1.0;
                            }
#pragma unroll
                            for (int c = 0; c < 8; ++c)
                                if (need >> c & 1u)
                                    v[c] = 0.0;
                            break;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 8; ++c)                  // segment order, on every segment
                        if (c < nc)
                            total += v[c];
                }
                acc = total;
            }
            const double gnew = acc * ek + ak;
            const double enew = exp_digamma(gnew) * c_psi;
            if (psi_on) {
                g_new[k_psi] = gnew;
                e_new[k_psi] = enew;
            }
            if constexpr (SPLIT) {                               // sum |gamma - last| here too
                const double half = wave_sum_dpp(psi_on ? fabs(g_old[kk] - gnew) : 0.0);
                if (lane == 0)
                    misc[wid] = half;
            }
        } else if (!SPLIT && wid >= W - 2) {
            // mean |gamma - last| (lda.cpp:202): two otherwise idle waves (on SIMDs the exp(psi)
            // waves do not use) recompute gamma for one topic per lane from the same partial
            // sums (bitwise the same value as the owners'), reduce with DPP and leave the two
            // halves of the sum -- all underneath the exp(psi) chain.  (One wave doing both
            // halves was the longest wave of the stage once exp(psi) got shorter.)
            const int kc = (wid - (W - 2)) * 64 + lane;
            const bool on = kc < K;
            const int ka = on ? kc : 0;
            // (every read requested before the first is consumed -- nothing is scheduled across the
            // sched_barrier: in the tiered kernels, close to the register limit, the compiler took them
            // one at a time, six LDS round trips in a row, and these waves needed 927 cycles where the
            // psi waves need 730 and the stage waits for both: every document of a launch with one
            // 129-word document paid 220 cycles per iteration, 38 % of the headline's launches;
            // profiles/r05_stamps_waves.txt.  The same barrier inside sum8_strided, for the psi waves
            // too: 26.2 against 26.0 us per step -- here only)
            const double *pp = partB + ka;
            const double p0v = pp[0], p1v = pp[kRegPart], p2v = pp[2 * kRegPart], p3v = pp[3 * kRegPart],
                         p4v = pp[4 * kRegPart], p5v = pp[5 * kRegPart], p6v = pp[6 * kRegPart],
                         p7v = pp[7 * kRegPart];
            const double eo = e_old[ka], al = alpha_l[ka], go = g_old[ka];
            __builtin_amdgcn_sched_barrier(0);
            const double ga = (((p0v + p1v) + (p2v + p3v)) + ((p4v + p5v) + (p6v + p7v))) * eo + al;
            const double v = on ? fabs(go - ga) : 0.0;
            const double half = wave_sum_dpp(v);
            if (lane == 0)
                misc[wid - (W - 2)] = half;
        }
        TRLDA_STAMP(7);
        __syncthreads();
        TRLDA_STAMP(4);
        // sum_k |gamma_k - last_k| (read now, used at the loop's end).  Every wave of this kernel
        // is bound by its instruction count, so the mean's division is not taken here: the
        // sum is compared with threshold * K instead
        const double change_sum = misc[0] + misc[1];
        if constexpr (SPLIT) {
            if (misc[2] + misc[3] != 0.0)            // an exchange gave up: results are void
                break;
        }

        product_E(e_new);                            // ends with a barrier
        TRLDA_STAMP(5);
        ++it;
        if (change_sum < thresholdK)                 // lda.cpp:202-203: mean < threshold
            break;
    }
    const double *g = gbuf + (it & 1) * 128, *e = ebuf + (it & 1) * 144;
    // the weights that go with the last phinorm (every product E ends with a barrier)
    if constexpr (kWaveWeights) {
        if (tid < 128 || tid < n)                    // 0 beyond n (cnt is 0 there)
            tw[tid] = cntd[tid] * rcp_pos<true>(sum8_strided<kRegPart>(part + tid) + 1e-100);
        __syncthreads();
    }

    // results (a split document's gamma by its first segment; every segment holds the same)
    if (!SPLIT || seg.x == 0) {
        for (int k = tid; k < K; k += T) {
            gamma_d[k] = g[k];
            merged_store(a.epg + (size_t)d * K + k, e[k], a.done_counter != nullptr);
        }
        if (tid == 0 && a.iters_out)
            a.iters_out[d] = it;
    }
    if (a.sstats_acc) {                              // lda.cpp:207-213, atomic form
        for (int j = wid; j < n; j += W) {
            const double c = tw[j];
            double *col = a.sstats_acc + (size_t)pids[j] * K;
            for (int k = lane; k < K; k += kWave)
                unsafeAtomicAdd(&col[k], c * e[k]);
        }
    } else {
        if (tid < n)
            merged_store(a.tw_word + (a.wrank ? a.wrank[p0 + tid] : p0 + tid), tw[tid],
                         a.done_counter != nullptr);
    }
    TRLDA_STAMP(6);
    TRLDA_STAMP_FLUSH;
}

// ---------------------------------------------------------------------------
// K <= 32 (round 6): a WAVE per document, eight documents per workgroup.
//
// The register body above spends a document's 27 us on 20 x (four barriers, three LDS round trips,
// exp(psi) on two of eight waves): none of it shrinks with K -- at K = 10 or 20 (the reference's own
// tests, onlinelda_test.py:39-68; BASELINE config 1) 54 or 44 of a product's 64 topic lanes idle and
// the document takes as long as at K = 128.  With few topics a document fits ONE wave, and a wave
// needs neither LDS nor barriers inside the loop:
//   * lane l holds the exp(psi(lambda)) rows of words l and l + 64 (K doubles each, in registers);
//   * phinorm_j = sum_k e_k u_jk: 2 K fmas per lane, e_k handed out by v_readlane (a scalar operand);
//   * acc_k = sum_j tw_j u_jk: per lane tw_l u_lk + tw_(l+64) u_(l+64)k, then ONE transposing
//     butterfly over the 32 topic vectors (fold<32> .. fold<1>: 16 + 8 + 4 + 2 + 1 + 1 steps) -- lane l
//     ends with the total of topic l >> 1;
//   * gamma, exp(psi(gamma)) and the change sum live in those lanes (each topic twice).
// The fixed point, its break test (sum |gamma - last| < threshold K, strict) and what is left behind
// -- gamma, c_k exp(psi(gamma)) rows, cnt / phinorm in word-major order, iteration counts -- are the
// register body's (lda.cpp:174-204); the sums are taken in another order (butterfly instead of eight
// partials), so the results agree to rounding, the iteration counts in every test.  The topic factors
// of the fused preamble are formed once per workgroup by the same code as above.  Every document of
// the body's workgroups has at most 128 words (the host checks); a.docs_per_wg = 8.  In a tiered launch the
// longer documents -- the first of the batch's sorted order -- keep a workgroup each in front of them.
__device__ __forceinline__ void estep_docs_small_body(const DocKernelArgs &a, double *lds)
{
    constexpr int T = kRegThreads, KM = 32;
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int K = a.K;
    double pv[2][8];
    if (a.partial && !a.scale_in)                    // launch-uniform
        topic_scale_load<T>(K, a.G, a.partial, pv);

    const int bid = doc_block(a);
    const int di = a.small_first + (bid - a.small_block0) * 8 + wid;   // this wave's document, in the batch's sorted order
    const bool doc_on = di < a.B;                    // wave-uniform
    const int dc = min(di, a.B - 1);
    int4 meta;
    int id0, id1;
    if (a.meta_i4 == 2) {
        // (launch-uniform: the workgroups in front hold segments of split documents, pad_meta / pad_ids are
        // THEIR layout -- the waves find their documents through the batch's order and its CSR arrays)
        const int dd = a.order[dc];
        const int q0 = a.indptr[dd], nn = a.indptr[dd + 1] - q0;
        meta = make_int4(dd, nn, q0, 0);
        id0 = a.ids[q0 + min(lane, max(nn - 1, 0))];
        id1 = a.ids[q0 + min(lane + 64, max(nn - 1, 0))];
        if (nn == 0)
            id0 = id1 = 0;
    } else {
        meta = reinterpret_cast<const int4 *>(a.pad_meta)[dc];
        const int32_t *__restrict__ pids = a.pad_ids + (size_t)dc * kRegMaxN;
        id0 = pids[lane];
        id1 = pids[64 + lane];
    }
    const int d = meta.x, n = doc_on ? meta.y : 0, p0 = meta.z;
    const int kt = lane >> 1;                        // this lane's topic (lanes 2 k and 2 k + 1: topic k)
    const bool k_on = kt < K;
    const int kc = k_on ? kt : 0;
    double g = a.gamma_in[(size_t)d * K + kc];
    const double ak = a.alpha[kc];
    const double cnt0 = lane < n ? (double)a.cnts[p0 + lane] : 0.0;
    const double cnt1 = lane + 64 < n ? (double)a.cnts[p0 + 64 + lane] : 0.0;
    double ck0 = 1.0;
    if (tid < K && a.scale_in && !a.scale_wait) {    // finished by the launch that prepared them
        ck0 = a.scale_in[2 * K + tid];
        if (bid == 0 && a.scale_out) {               // (a tiered launch: document workgroup 0 is a register body's)
            a.scale_out[tid] = a.scale_in[tid];
            a.scale_out[K + tid] = a.scale_in[K + tid];
            a.scale_out[2 * K + tid] = ck0;
        }
    }
    // the rows: K contiguous doubles per word (K is even: 16-byte loads), zero past the document's end
    double u0[KM], u1[KM];
    {
        const double *r0 = a.eeb + (size_t)id0 * K, *r1 = a.eeb + (size_t)id1 * K;
        const bool on0 = lane < n, on1 = lane + 64 < n;
        const bool k_even = (K & 1) == 0;            // (rows of an odd K are 8-byte aligned only: a double at a time)
#pragma unroll
        for (int k = 0; k < KM; k += 2) {
            double2 v0 = make_double2(0.0, 0.0), v1 = make_double2(0.0, 0.0);
            if (k < K && k_even) {                   // launch-uniform
                v0 = *reinterpret_cast<const double2 *>(r0 + k);
                v1 = *reinterpret_cast<const double2 *>(r1 + k);
            } else if (k < K) {
                v0.x = r0[k];
                v1.x = r1[k];
                if (k + 1 < K) {
                    v0.y = r0[k + 1];
                    v1.y = r1[k + 1];
                }
            }
            u0[k] = on0 ? v0.x : 0.0; u0[k + 1] = on0 ? v0.y : 0.0;
            u1[k] = on1 ? v1.x : 0.0; u1[k + 1] = on1 ? v1.y : 0.0;
        }
    }
    // topic factors: once per workgroup, the register body's code (bitwise its c)
    double *part = lds, *c_l = lds + 8 * 128;
    if (a.partial && !a.scale_in)
        topic_scale_partials<T>(K, a.G, pv, part);
    __syncthreads();
    if (tid < K) {
        double ck = 1.0;
        if (a.partial)
            ck = a.scale_in ? (a.scale_wait ? scale_wait_load(a, K, tid) : ck0)
                            : topic_scale_combine(K, tid, part, a.scale_out);
        c_l[tid] = ck;
    }
    __syncthreads();
    if (!doc_on)                                     // (a wave without a document: it meets the others at the
        return;                                      //  barriers behind the body)
    const double ck = c_l[kc];
    double e = k_on ? exp_digamma(g) * ck : 0.0;     // e is kept as c_k exp(psi(gamma_k)) throughout
    const int elo0 = 0;
    (void)elo0;
    double tw0 = 0.0, tw1 = 0.0;
    auto weights = [&]() {                           // tw_j = cnt_j / (sum_k e_k u_jk + 1e-100)   lda.cpp:183 / :199
        const int lo = __double2loint(e), hi = __double2hiint(e);
        double s0[2] = {0.0, 0.0}, s1[2] = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            if (k < K) {                             // launch-uniform
                const double ek = __hiloint2double(__builtin_amdgcn_readlane(hi, 2 * k),
                                                   __builtin_amdgcn_readlane(lo, 2 * k));
                s0[k & 1] = fma(ek, u0[k], s0[k & 1]);
                s1[k & 1] = fma(ek, u1[k], s1[k & 1]);
            }
        }
        tw0 = cnt0 * rcp_pos<true>((s0[0] + s0[1]) + 1e-100);
        tw1 = cnt1 * rcp_pos<true>((s1[0] + s1[1]) + 1e-100);
    };
    weights();
    const double thresholdK = a.threshold * (double)K;
    int it = 0;
    while (it < a.max_iter) {                        // lda.cpp:185-204 (wave-uniform: no barrier inside)
        // acc_k = sum_j tw_j u_jk: this lane's two words, then the transposing butterfly
        double q16[16], q8[8], q4[4], q2[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double pa = i < K ? fma(tw1, u1[i], tw0 * u0[i]) : 0.0;
            const double pb = i + 16 < K ? fma(tw1, u1[i + 16], tw0 * u0[i + 16]) : 0.0;
            q16[i] = fold<32>(pa, pb);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
            q8[i] = fold<16>(q16[i], q16[i + 8]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            q4[i] = fold<8>(q8[i], q8[i + 4]);
        q2[0] = fold<4>(q4[0], q4[2]);
        q2[1] = fold<4>(q4[1], q4[3]);
        const double q1 = fold<2>(q2[0], q2[1]);
        const double acc = fold<1>(q1, q1);          // lane l: topic 16 b5 + 8 b4 + 4 b3 + 2 b2 + b1 = l >> 1
        const double gnew = acc * e + ak;            // lda.cpp:193
        const double enew = k_on ? exp_digamma(gnew) * ck : 0.0;
        const double change_sum = wave_sum_dpp((k_on && !(lane & 1)) ? fabs(g - gnew) : 0.0);
        g = gnew;
        e = enew;
        weights();
        ++it;
        if (change_sum < thresholdK)                 // lda.cpp:202-203: mean < threshold
            break;
    }
    if (k_on && !(lane & 1)) {
        a.gamma[(size_t)d * K + kt] = g;
        merged_store(a.epg + (size_t)d * K + kt, e, a.done_counter != nullptr);
    }
    if (lane == 0 && a.iters_out)
        a.iters_out[d] = it;
    if (lane < n)
        merged_store(a.tw_word + (a.wrank ? a.wrank[p0 + lane] : p0 + lane), tw0, a.done_counter != nullptr);
    if (lane + 64 < n)
        merged_store(a.tw_word + (a.wrank ? a.wrank[p0 + 64 + lane] : p0 + 64 + lane), tw1,
                     a.done_counter != nullptr);
}

template <int MODE>
__global__ __launch_bounds__(kRegThreads) void estep_docs_reg_kernel(DocKernelArgs a, PreArgs pre)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if ((int)blockIdx.x >= pre.n_docs) {             // block-uniform
        docs_launch_preamble(pre, lds, (int)blockIdx.x);
        return;
    }
    if (a.docs_per_wg == 8)                          // launch-uniform
        estep_docs_small_body(a, lds);
    else
        estep_docs_reg_body<MODE>(a, lds);
}

// ---------------------------------------------------------------------------
// 4a. Segmented sufficient statistics: one wavefront per word, lanes over k.
//   sstats[k, w] = eeb[k, w] * sum_{q in word w} tw_word[q] * epg[k, wdoc[q]]
// The entries of a word are stored in document order, so the additions happen in
// the order of the reference's serial loop (lda.cpp:207-213) and the result is
// bitwise reproducible.  Words without entries get 0 (lda.cpp:169).
// sstats_words_kernel writes the statistics only; sstats_update_kernel (4c, below) also applies
// the M-step and collects the row sums of the lambda it writes.
// ---------------------------------------------------------------------------
// The weights cnt / phinorm in word-major order: a plain array, or -- data-parallel factor
// exchange, dp_kernels.h -- the gathered buffer of all ranks' weights (CSR order per rank)
// seen through an index that is static per (mini-batch, shard cuts).
struct TwView {
    const double *v;
    const int32_t *idx;       // or nullptr
    __device__ __forceinline__ double operator[](int q) const { return idx ? v[idx[q]] : v[q]; }
};

// acc[h] += sum_{q in [q0, q1)} tw_word[q] * epg[kbase + 64 h + lane, wdoc[q]], h < NH topic
// halves; q0, q1 are wave-uniform.  Per pass of 16 entries: lane u fetches (document,
// weight) of entry u with one coalesced load each -- one memory latency for the whole list
// instead of a dependent scalar load per entry -- v_readlane hands entry u to every lane,
// the row gathers of a group of four are in flight together (groups past the end are
// skipped: most words have a handful of entries), products are added in list order.
#ifndef TRLDA_STAT_ROWS
#define TRLDA_STAT_ROWS 4
#endif
#ifndef TRLDA_STAT_ROWS2
#define TRLDA_STAT_ROWS2 4
#endif
constexpr int kStatRows = TRLDA_STAT_ROWS;      // rows of exp(psi(gamma)) in flight per wave (8-byte gathers)
constexpr int kStatRows2 = TRLDA_STAT_ROWS2;    // ... (16-byte gathers)

template <int NH>
__device__ __forceinline__ void word_segment_sum(int q0, int q1, int K, int kbase,
                                                 const int32_t *__restrict__ wdoc,
                                                 TwView tw_word,
                                                 const double *__restrict__ epg, double *acc)
{
    const int lane = threadIdx.x & (kWave - 1);
    int kk[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h)
        kk[h] = min(kbase + 64 * h + lane, K - 1);  // lanes past K gather topic K-1, unused
    // (the next pass's entries are requested before this pass's rows: their latency runs under
    // the gathers instead of in front of them)
    int dl = lane < q1 - q0 ? wdoc[q0 + lane] : 0;   // entries past the end: row 0, weight 0
    double tl = lane < q1 - q0 ? tw_word[q0 + lane] : 0.0;
    for (int q = q0; q < q1; q += 16) {
        const int cnt = min(16, q1 - q);
        const int dcur = dl;
        const int tlo = __double2loint(tl), thi = __double2hiint(tl);
        if (q + 16 < q1) {                           // wave-uniform
            const bool more = lane < q1 - q - 16;
            dl = more ? wdoc[q + 16 + lane] : 0;
            tl = more ? tw_word[q + 16 + lane] : 0.0;
        }
#pragma unroll
        for (int grp = 0; grp < 16 / kStatRows; ++grp) {
            if (kStatRows * grp < cnt) {             // wave-uniform
                double ev[kStatRows][NH];
#pragma unroll
                for (int u = 0; u < kStatRows; ++u) {
                    const size_t row = (size_t)__builtin_amdgcn_readlane(dcur, kStatRows * grp + u) * K;
#pragma unroll
                    for (int h = 0; h < NH; ++h)
                        ev[u][h] = epg[row + kk[h]];
                }
#pragma unroll
                for (int u = 0; u < kStatRows; ++u) {
                    const double tu =
                        __hiloint2double(__builtin_amdgcn_readlane(thi, kStatRows * grp + u),
                                         __builtin_amdgcn_readlane(tlo, kStatRows * grp + u));
#pragma unroll
                    for (int h = 0; h < NH; ++h)
                        acc[h] = fma(tu, ev[u][h], acc[h]);   // +0 * finite past the end
                }
            }
        }
    }
}

// Entries above which a word's list is split over the waves of a block instead of being walked by
// one wave: at least kLongWord, and per batch the smallest power-of-two multiple of it that leaves
// at most kLongWordsTarget such words (trlda_batch::long_len, chosen when the batch is indexed).
// A block per word pays off for the few hundred longest lists; at 12 500 documents most active
// words have more than 16 entries, and walking ~20 000 of them with 256 blocks, a few dependent
// latencies per word, was the whole 430 us of the kernel.
// (kLongWord = 16, kLongWordsTarget = 512: index_params.h)

// VERY long lists (round 4).  A word present in all 12 500 documents of a BatchLDA shard kept ONE
// workgroup busy for ~100 us -- 98 passes of 16 entries per wave, each a dependent gather -- while
// the mean wavefront of the launch lived a tenth of that (profiles/r03_stats_kernel_notes.txt); at
// 1600 documents the longest list alone set the kernel's 24 us.  A list of more than seg_len
// entries is therefore cut into SEGMENTS of at most seg_len: a segment is a task for a whole
// workgroup (a pass or two per wave), its K sums go to a row of `seg_partial`, and the workgroup
// that finishes a word's LAST segment (a counter per word; rows stored and loaded with agent-scope
// accesses, as in finish_partial_groups) adds the rows up in segment order and applies the
// epilogue.  The sums and their order are fixed by the list's length alone: bitwise reproducible,
// the same on every rank.
// The segment length is chosen per batch (trlda_batch::seg_len): the power of two between
// kSegMin and kSegMax that makes at most about kSegTasks tasks of the entries in the lists of more than
// kSegMin -- 256 (a pass per wave and task) where there are few such entries and the launch is a
// handful of latencies long (1600 documents at K = 100; one rank's range of it: 7.7 us), 1024 where
// there are a million of them and a task's fixed costs (~3 us of barriers and round trips) would
// otherwise dominate (256 everywhere: K = 100 / 6400 documents 60 -> 70 us; 1024 everywhere: one
// rank of eight at 8 x 200 documents 7.7 -> 15 us).
// (kSegMin = 256, kSegMax = 1024, kSegTasks = 512; kOneWaveMax = 256 -- long_len never exceeds it: a wave
// walks 16 entries per pass: index_params.h)
struct VeryLongArgs {
    int G_seg;                    // workgroups walking the segment tasks (0: none)
    int seg_len;                  // lists of more than this many entries are cut into segments
    int n_tasks, n_words;         // of this launch (a rank's slice, data-parallel)
    const int4 *task;             // (word index j, segment, first entry, entries), by (j, segment)
    const int4 *word;             // j -> (word id, first task = first row of seg_partial, segments, 0)
    int j0, t0;                   // first word index / task of the launch: rows and tasks are relative to them
    double *seg_partial;          // n_tasks x K
    unsigned int *seg_counter;    // one per word (indexed by j), zero between launches
    int row_base;                 // o.partial row of word j: row_base + j - j0
    int n_rows;                   // rows of o.partial in all (for the groups)
};

// At large batches (long_len above its floor: most entries sit in the longest lists) the blocks
// that walk those lists are the last to finish: they take the FIRST physical workgroups of the
// launch (dispatched first) and the one-wave-per-word blocks follow -- K = 200 / 12 500 documents:
// 400 -> 364 us, K = 500 / 4096: 400 -> 362, K = 100 / 6400: 70 -> 64.  Small batches keep the
// plain order (the 200-document headline: 5.9 us either way, 6.4 when rotated).  `bid` is the
// logical block (row of o.partial; long blocks at G_short ..): the arithmetic and its order do not
// depend on the physical placement.
__device__ __forceinline__ int long_first_block(int G_short, int long_len)
{
    if (long_len <= kLongWord)
        return (int)blockIdx.x;
    const int nb = (int)gridDim.x;
    const int b = (int)blockIdx.x + G_short;
    return b >= nb ? b - nb : b;
}

template <int T>
__global__ __launch_bounds__(T) void sstats_words_kernel(
    int K, int V, int G_short, int long_len, const int32_t *__restrict__ wptr,
    const int32_t *__restrict__ wdoc, const int32_t *__restrict__ long_words, TwView tw_word,
    const double *__restrict__ epg, const double *__restrict__ eeb, double *__restrict__ sstats)
{
    constexpr int W = T / kWave;
    extern __shared__ double wpart[];                // W x K partial sums (long words)
    const int lane = threadIdx.x & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int bid = long_first_block(G_short, long_len);

    if (bid < G_short) {
        // ---- one wavefront per word; lists longer than long_len are left to the blocks
        // below.  Words are dealt round-robin over blocks (w = wave * G + block).
        const int w = wid * G_short + bid;
        if (w >= V)
            return;
        const int q0 = __builtin_amdgcn_readfirstlane(wptr[w]);
        const int len = __builtin_amdgcn_readfirstlane(wptr[w + 1]) - q0;
        if (len > long_len)
            return;
        for (int kb = 0; kb < K; kb += 2 * kWave) {
            double acc[2] = {0.0, 0.0};
            if (len > 0)
                word_segment_sum<2>(q0, q0 + len, K, kb, wdoc, tw_word, epg, acc);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = kb + 64 * h + lane;
                if (k < K) {
                    const size_t i = (size_t)w * K + k;
                    // lda.cpp:169: words the batch does not touch are 0 (eeb is not read)
                    sstats[i] = len > 0 ? acc[h] * eeb[i] : 0.0;
                }
            }
        }
        return;
    }

    // ---- one block per long list (they start together with the short-word blocks): the
    // entries are split into W contiguous chunks, chunk sums are combined in chunk order
    // (fixed by the list length -> bitwise reproducible)
    const int w = long_words[bid - G_short];
    const int base = __builtin_amdgcn_readfirstlane(wptr[w]);
    const int L = __builtin_amdgcn_readfirstlane(wptr[w + 1]) - base;
    const int chunk = (L + W - 1) / W;
    const int c0 = __builtin_amdgcn_readfirstlane(min(L, wid * chunk));
    const int c1 = __builtin_amdgcn_readfirstlane(min(L, c0 + chunk));
    for (int kb = 0; kb < K; kb += 2 * kWave) {
        double acc[2] = {0.0, 0.0};
        word_segment_sum<2>(base + c0, base + c1, K, kb, wdoc, tw_word, epg, acc);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = kb + 64 * h + lane;
            if (k < K)
                wpart[wid * K + k] = acc[h];
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += T) {
        double p[W];
#pragma unroll
        for (int c = 0; c < W; ++c)
            p[c] = wpart[c * K + k];
        double acc = p[0];
#pragma unroll
        for (int c = 1; c < W; ++c)
            acc += p[c];
        const size_t idx = (size_t)w * K + k;
        sstats[idx] = acc * eeb[idx];
    }
}

// ---------------------------------------------------------------------------
// 4c. Statistics, M-step and the next E-step's row sums in ONE pass over the words (K <= 512).
//
//   s[k, w]      = eeb[k, w] * sum_{q in word w} tw_word[q] * epg[k, wdoc[q]]    lda.cpp:207-217
//   lambda[k, w] = omr * lambda'[k, w] + rho * (eta + scale * s[k, w])
//       OnlineLDA  onlinelda.cpp:99-100, :108-109   omr = 1 - rho, scale = D / B
//       BatchLDA   batchlda.cpp:60                  no lambda' term, rho = 1, scale = 1
//       Cumulative cumulativelda.cpp:70             omr = 1, rho = 1, eta = 0, scale = 1
//   partial[block][k] = sum over the block's words of lambda[k, w]                lda.cpp:172
//
// The same per-word ordered sums as sstats_words_kernel (bitwise the same s), but the words
// come from a list -- all V, or only the batch's active words: inside a trust-region loop the
// inactive ones do not change (stream_kernels.h) -- and a wave walks its share of the list
// (positions wave * G + block, then every W * G further on) keeping the row sums of what it
// writes in registers.  One kernel boundary later rowsum_combine_wave_kernel adds the block
// partials (in block order) to the row sums of the inactive words: the next E-step starts
// without reading lambda at all.  lambda' may be lambda itself (in-place update): an element
// is read and written by the same thread.
// ---------------------------------------------------------------------------
struct UpdateOut {
    double *sstats;               // K x V, or nullptr
    double *lambda;               // K x V, or nullptr: no M-step
    const double *lambda_prime;   // K x V, or nullptr: lambda = rho * (eta + scale * s)
    double omr, rho, eta, scale;
    double *partial;              // gridDim.x x K, or nullptr
    // The next E-step's preamble, while lambda is in registers: u = exp(psi(lambda)) of every
    // element written (may be the buffer `eeb` is read from: an element is read and replaced by
    // the same thread), and the block partials of the row sums combined in groups of
    // `group_size` consecutive blocks by the LAST block of each group to finish (a counter per
    // group, reset by that block) -- few enough rows for the document kernel to add up itself
    // (estep_kernels.h, topic_scale_*), in a fixed order whatever block comes last.
    double *u_out;                // K x V, or nullptr
    double *group_rows;           // ceil(gridDim.x / group_size) x K, or nullptr
    const double *group_base;     // K, added into group 0 (the words outside the batch), or nullptr
    unsigned int *group_counter;  // one per group, zero between launches
    int group_size;
    // word-sharded M-step (data-parallel, dp_kernels.h): without a list the kernel walks the words
    // w0 .. w0 + N - 1 -- this rank's range of the vocabulary
    int w0;
};

// Row k of a block's partial sums, for finish_partial_groups: stored and loaded with agent-scope
// atomics (on gfx950: sc1 accesses, served by the memory side, coherent across the XCDs' L2s).
// The ordering that makes the last block see every row is then a matter of the store having
// been acknowledged before the block's counter increment is issued -- a workgroup-scope release
// (s_waitcnt vmcnt(0)) -- instead of an agent-scope fence, whose L2 write-back of everything the
// kernel has written so far (all of lambda) cost 300 us per launch when it was tried.
__device__ __forceinline__ void store_partial(const UpdateOut &o, size_t idx, double v)
{
    if (o.group_rows)
        __hip_atomic_store(o.partial + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        o.partial[idx] = v;
}

// the end of a block of the statistics kernels, after its row of o.partial has been written
template <int T>
__device__ __forceinline__ void finish_partial_groups(const UpdateOut &o, int K, int bid, int rows)
{
    if (!o.group_rows)                               // launch-uniform
        return;
#ifdef TRLDA_EXPT_NOGROUP                            // timing experiment: results are wrong
    return;
#endif
    __shared__ int last_of_group;
    const int g = bid / o.group_size;
    const int r0 = g * o.group_size, r1 = min(rows, r0 + o.group_size);
    stores_acknowledged();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int seen = __hip_atomic_fetch_add(&o.group_counter[g], 1u, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
        const int last = seen + 1u == (unsigned int)(r1 - r0);
        if (last)                                    // for the next launch (stream order)
            __hip_atomic_store(&o.group_counter[g], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_of_group = last;
    }
    __syncthreads();
    if (!last_of_group)
        return;
    for (int k = threadIdx.x; k < K; k += T) {
        double sum = (g == 0 && o.group_base) ? o.group_base[k] : 0.0;
        for (int r = r0; r < r1; ++r)
            sum += __hip_atomic_load(o.partial + (size_t)r * K + k, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
        o.group_rows[(size_t)g * K + k] = sum;
    }
}

template <bool EMIT>
__device__ __forceinline__ double update_one(const UpdateOut &o, size_t i, double s)
{
    if (o.sstats)
        o.sstats[i] = s;
    double lam = 0.0;
    if (o.lambda) {
        const double hat = o.eta + o.scale * s;
        lam = o.lambda_prime ? o.omr * o.lambda_prime[i] + o.rho * hat : o.rho * hat;
        o.lambda[i] = lam;
        if (EMIT && o.u_out)                         // (lam > 0: the host only asks for u_out then)
            o.u_out[i] = exp_digamma_positive<true>(lam);
    }
    return lam;
}

// The end of a segment task of a very long list (VeryLongArgs): the W per-wave sums of the segment
// are in wpart[W][K]; they become one row of seg_partial, and the workgroup that brings the word's
// counter to its number of segments adds all rows up (segment order, four or eight loads in flight) and
// applies the epilogue; its lambdas are a row of o.partial of their own (so that the row sums do
// not depend on which workgroup came last).
template <int T, int W, bool EMIT>
__device__ __forceinline__ void segment_finish(const UpdateOut &o, const VeryLongArgs &vl, int K, int4 tk,
                                               const double *wpart, const double *eeb)
{
    __shared__ int last_seg;
    const int tid = threadIdx.x;
    const int4 wd = vl.word[tk.x];                   // (word, first task, segments, 0)
    const size_t row = (size_t)(wd.y + tk.y - vl.t0);
    __syncthreads();                                 // wpart complete
    for (int k = tid; k < K; k += T) {
        double sum = wpart[k];
#pragma unroll
        for (int c = 1; c < W; ++c)
            sum += wpart[c * K + k];
        __hip_atomic_store(vl.seg_partial + row * K + k, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    stores_acknowledged();
    __syncthreads();
    if (tid == 0) {
        const unsigned int seen = __hip_atomic_fetch_add(vl.seg_counter + tk.x, 1u, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
        const int last = seen + 1u == (unsigned int)wd.z;
        if (last)                                    // for the next launch (stream order)
            __hip_atomic_store(vl.seg_counter + tk.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_seg = last;
    }
    __syncthreads();
    if (last_seg) {                                  // block-uniform
        const double *rows = vl.seg_partial + (size_t)(wd.y - vl.t0) * K;
        const int prow = vl.row_base + tk.x - vl.j0;
        for (int k = tid; k < K; k += T) {
            const size_t i = (size_t)wd.x * K + k;
            const double ek = eeb[i];
            double sum = 0.0;
            constexpr int NV = EMIT ? 4 : 8;         // (the emitting instantiation lives on 64 VGPRs)
            for (int s0 = 0; s0 < wd.z; s0 += NV) {
                double v[NV];
#pragma unroll
                for (int q = 0; q < NV; ++q)
                    v[q] = __hip_atomic_load(rows + (size_t)min(s0 + q, wd.z - 1) * K + k, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int q = 0; q < NV; ++q)
                    sum += (s0 + q < wd.z) ? v[q] : 0.0;
            }
            const double lam = update_one<EMIT>(o, i, sum * ek);
            if (o.partial)
                store_partial(o, (size_t)prow * K + k, lam);
        }
        if (o.partial)
            finish_partial_groups<T>(o, K, prow, vl.n_rows);
    }
    __syncthreads();                                 // wpart is free again
}

template <int T, int NKB, bool EMIT>                 // EMIT: also UpdateOut::u_out (K <= 128 only)
__global__ __launch_bounds__(T, EMIT ? (T <= 512 ? 6 : 8) : 1) void sstats_update_kernel(
    int K, int N, int G_short, int n_long, int long_len, const int32_t *__restrict__ list,
    const int32_t *__restrict__ wptr, const int32_t *__restrict__ wdoc,
    const int32_t *__restrict__ long_words, TwView tw_word,
    const double *__restrict__ epg, const double *eeb /* may be o.u_out */, UpdateOut o, VeryLongArgs vl)
{
    constexpr int W = T / kWave;
    extern __shared__ double wpart[];                // W x K
    const int lane = threadIdx.x & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);

    const int bid = long_first_block(G_short, long_len);
    if (bid < G_short) {
        double rs[NKB][2];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
            rs[kb][0] = rs[kb][1] = 0.0;
        for (int p = wid * G_short + bid; p < N; p += W * G_short) {
            const int w = __builtin_amdgcn_readfirstlane(list ? list[p] : p + o.w0);
            const int q0 = __builtin_amdgcn_readfirstlane(wptr[w]);
            const int len = __builtin_amdgcn_readfirstlane(wptr[w + 1]) - q0;
            if (len > long_len)
                continue;                            // left to the blocks below
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                if (kb * 2 * kWave < K) {            // wave-uniform
                    double acc[2] = {0.0, 0.0};
                    if (len > 0)
                        word_segment_sum<2>(q0, q0 + len, K, kb * 2 * kWave, wdoc, tw_word, epg, acc);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int k = kb * 2 * kWave + 64 * h + lane;
                        if (k < K) {
                            const size_t i = (size_t)w * K + k;
                            // lda.cpp:169: words the batch does not touch are 0 (eeb is not read)
                            const double s = len > 0 ? acc[h] * eeb[i] : 0.0;
                            rs[kb][h] += update_one<EMIT>(o, i, s);
                        }
                    }
                }
            }
        }
        if (o.partial) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int k = kb * 2 * kWave + 64 * h + lane;
                    if (k < K)
                        wpart[wid * K + k] = rs[kb][h];
                }
            __syncthreads();
            for (int k = threadIdx.x; k < K; k += T) {
                double sum = wpart[k];
                for (int c = 1; c < W; ++c)
                    sum += wpart[c * K + k];
                store_partial(o, (size_t)bid * K + k, sum);
            }
            finish_partial_groups<T>(o, K, bid, vl.n_rows);
        }
        return;
    }

    // ---- segments of the very long lists (VeryLongArgs): the last vl.G_seg blocks
    const int G_long = (int)gridDim.x - G_short - vl.G_seg;
    if (bid >= G_short + G_long) {
        for (int t = bid - G_short - G_long; t < vl.n_tasks; t += vl.G_seg) {      // block-uniform
            const int4 tk = vl.task[vl.t0 + t];      // (word index, segment, first entry, entries)
            const int chunk = (tk.w + W - 1) / W;
            const int c0 = __builtin_amdgcn_readfirstlane(min(tk.w, wid * chunk));
            const int c1 = __builtin_amdgcn_readfirstlane(min(tk.w, c0 + chunk));
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                if (kb * 2 * kWave < K) {
                    double acc[2] = {0.0, 0.0};
                    word_segment_sum<2>(tk.z + c0, tk.z + c1, K, kb * 2 * kWave, wdoc, tw_word, epg, acc);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int k = kb * 2 * kWave + 64 * h + lane;
                        if (k < K)
                            wpart[wid * K + k] = acc[h];
                    }
                }
            }
            segment_finish<T, W, EMIT>(o, vl, K, tk, wpart, eeb);
        }
        return;
    }

    // ---- long lists: the remaining blocks walk long_words, one word per pass, the entries
    // split into W contiguous chunks whose sums are combined in chunk order
    double rsl[(512 + T - 1) / T];                   // row sums of thread k = tid (+ T ..)
#pragma unroll
    for (int c = 0; c < (512 + T - 1) / T; ++c)
        rsl[c] = 0.0;
    for (int lw = bid - G_short; lw < n_long; lw += G_long) {
        const int w = long_words[lw];
        const int base = __builtin_amdgcn_readfirstlane(wptr[w]);
        const int L = __builtin_amdgcn_readfirstlane(wptr[w + 1]) - base;
        if (vl.G_seg > 0 && L > vl.seg_len)
            continue;                                // cut into segments (above)
        const int chunk = (L + W - 1) / W;
        const int c0 = __builtin_amdgcn_readfirstlane(min(L, wid * chunk));
        const int c1 = __builtin_amdgcn_readfirstlane(min(L, c0 + chunk));
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb * 2 * kWave < K) {
                double acc[2] = {0.0, 0.0};
                word_segment_sum<2>(base + c0, base + c1, K, kb * 2 * kWave, wdoc, tw_word, epg, acc);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int k = kb * 2 * kWave + 64 * h + lane;
                    if (k < K)
                        wpart[wid * K + k] = acc[h];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < (512 + T - 1) / T; ++c) {
            const int k = threadIdx.x + c * T;
            if (k < K) {
                // (chunk order; EMIT: read and added one after the other -- sixteen values held at
                // once are 32 of the 64 VGPRs this instantiation may use)
                double acc;
                if constexpr (EMIT) {
                    acc = wpart[k];
#pragma unroll 4
                    for (int q = 1; q < W; ++q)
                        acc += wpart[q * K + k];
                } else {
                    double pv[W];
#pragma unroll
                    for (int q = 0; q < W; ++q)
                        pv[q] = wpart[q * K + k];
                    acc = pv[0];
#pragma unroll
                    for (int q = 1; q < W; ++q)
                        acc += pv[q];
                }
                const size_t i = (size_t)w * K + k;
                rsl[c] += update_one<EMIT>(o, i, acc * eeb[i]);
            }
        }
        __syncthreads();
    }
    if (o.partial) {
#pragma unroll
        for (int c = 0; c < (512 + T - 1) / T; ++c) {
            const int k = threadIdx.x + c * T;
            if (k < K)
                store_partial(o, (size_t)bid * K + k, rsl[c]);
        }
        finish_partial_groups<T>(o, K, bid, vl.n_rows);
    }
}

// ---------------------------------------------------------------------------
// 4d. The same kernel for even K: every lane owns two ADJACENT topics, so that a row of
// exp(psi(gamma)) is gathered with one 16-byte load per lane (8-byte gathers reach about 0.6 of
// the 16-byte rate on gfx950) and a pass over a word's entry list serves 256 topics instead of
// 128 -- half as many walks of the list at K = 500.  Same sums in the same order per topic:
// bitwise the same statistics as 4c.
// ---------------------------------------------------------------------------
template <int NH>
__device__ __forceinline__ void word_segment_sum2(int q0, int q1, int K, int kbase,
                                                  const int32_t *__restrict__ wdoc,
                                                  TwView tw_word,
                                                  const double *__restrict__ epg, double2 *acc)
{
    const int lane = threadIdx.x & (kWave - 1);
    int kk[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h)
        kk[h] = min(kbase + 128 * h + 2 * lane, K - 2);   // lanes past K gather the last pair, unused
    int dl = lane < q1 - q0 ? wdoc[q0 + lane] : 0;
    double tl = lane < q1 - q0 ? tw_word[q0 + lane] : 0.0;
    for (int q = q0; q < q1; q += 16) {
        const int cnt = min(16, q1 - q);
        const int dcur = dl;
        const int tlo = __double2loint(tl), thi = __double2hiint(tl);
        if (q + 16 < q1) {                           // wave-uniform
            const bool more = lane < q1 - q - 16;
            dl = more ? wdoc[q + 16 + lane] : 0;
            tl = more ? tw_word[q + 16 + lane] : 0.0;
        }
#pragma unroll
        for (int grp = 0; grp < 16 / kStatRows2; ++grp) {
            if (kStatRows2 * grp < cnt) {            // wave-uniform
                double2 ev[kStatRows2][NH];
#pragma unroll
                for (int u = 0; u < kStatRows2; ++u) {
                    const size_t row = (size_t)__builtin_amdgcn_readlane(dcur, kStatRows2 * grp + u) * K;
#pragma unroll
                    for (int h = 0; h < NH; ++h)
                        ev[u][h] = *reinterpret_cast<const double2 *>(epg + row + kk[h]);
                }
#pragma unroll
                for (int u = 0; u < kStatRows2; ++u) {
                    const double tu =
                        __hiloint2double(__builtin_amdgcn_readlane(thi, kStatRows2 * grp + u),
                                         __builtin_amdgcn_readlane(tlo, kStatRows2 * grp + u));
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        acc[h].x = fma(tu, ev[u][h].x, acc[h].x);
                        acc[h].y = fma(tu, ev[u][h].y, acc[h].y);
                    }
                }
            }
        }
    }
}

// the pair (i, i + 1) of one word: statistics, M-step, returns the two lambdas written
// (lp: lambda'[i], fetched by the caller ahead of the word's gathers -- loads return in order,
// and a request made here, behind them, would be one more memory latency per word)
template <bool EMIT>
__device__ __forceinline__ double2 update_pair(const UpdateOut &o, size_t i, double2 s, double2 lp)
{
    if (o.sstats)
        *reinterpret_cast<double2 *>(o.sstats + i) = s;
    double2 lam = make_double2(0.0, 0.0);
    if (o.lambda) {
        const double hx = o.eta + o.scale * s.x, hy = o.eta + o.scale * s.y;
        if (o.lambda_prime) {
            lam.x = o.omr * lp.x + o.rho * hx;
            lam.y = o.omr * lp.y + o.rho * hy;
        } else {
            lam.x = o.rho * hx;
            lam.y = o.rho * hy;
        }
        *reinterpret_cast<double2 *>(o.lambda + i) = lam;
        if (EMIT && o.u_out) {
            // (lam > 0: the host only asks for u_out then.)  One after the other, not interleaved:
            // the two chains side by side need ~100 VGPRs, and above 64 a CU holds ONE 1024-thread
            // workgroup of this kernel instead of two -- the launch then runs in two rounds (15.4 us
            // against 8.2 us without this stream in round 3)
            const double ux = exp_digamma_positive<true>(lam.x);
            double ly = lam.y;
            asm volatile("" : "+v"(ly) : "v"(ux));
            *reinterpret_cast<double2 *>(o.u_out + i) = make_double2(ux, exp_digamma_positive<true>(ly));
        }
    }
    return lam;
}

template <int T, int NKB, int NH, bool EMIT>         // NKB = ceil(K / (128 NH)), K even
__global__ __launch_bounds__(T, (EMIT && NH <= 1 && NKB == 1) ? (T <= 512 ? 6 : 8)
                                   : (EMIT && T <= 512) ? 4 : 1) void sstats_update2_kernel(
    int K, int N, int G_short, int n_long, int long_len, const int32_t *__restrict__ list,
    const int32_t *__restrict__ wptr, const int32_t *__restrict__ wdoc,
    const int32_t *__restrict__ long_words, TwView tw_word,
    const double *__restrict__ epg, const double *eeb /* may be o.u_out */, UpdateOut o, VeryLongArgs vl)
{
    constexpr int W = T / kWave;
    constexpr int BW = 128 * NH;                     // topics per pass over a word's list
    extern __shared__ __attribute__((aligned(16))) double wpart2[];   // W x K
    const int lane = threadIdx.x & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);

    const int bid = long_first_block(G_short, long_len);
    if (bid < G_short) {
        double2 rs[NKB][NH];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int h = 0; h < NH; ++h)
                rs[kb][h] = make_double2(0.0, 0.0);
        for (int p = wid * G_short + bid; p < N; p += W * G_short) {
            const int w = __builtin_amdgcn_readfirstlane(list ? list[p] : p + o.w0);
            const int q0 = __builtin_amdgcn_readfirstlane(wptr[w]);
            const int len = __builtin_amdgcn_readfirstlane(wptr[w + 1]) - q0;
            if (len > long_len)
                continue;                            // left to the blocks below
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                if (kb * BW < K) {            // wave-uniform
                    double2 acc[NH], e2[NH], lp[NH];
                    // exp E[log beta] and lambda' of this word's elements: requested before the
                    // gathers (clamped, unconditional), consumed after them
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        acc[h] = make_double2(0.0, 0.0);
                        const size_t ic = (size_t)w * K + min(kb * BW + 128 * h + 2 * lane, K - 2);
                        // (a word without entries has zero statistics: its exp E[log beta] is
                        // neither defined -- only the batch's words are filled -- nor read)
                        e2[h] = len > 0 ? *reinterpret_cast<const double2 *>(eeb + ic)
                                        : make_double2(0.0, 0.0);
                        lp[h] = o.lambda_prime ? *reinterpret_cast<const double2 *>(o.lambda_prime + ic)
                                               : make_double2(0.0, 0.0);
                    }
                    if (len > 0)
                        word_segment_sum2<NH>(q0, q0 + len, K, kb * BW, wdoc, tw_word, epg, acc);
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        const int k = kb * BW + 128 * h + 2 * lane;
                        if (k < K) {
                            const size_t i = (size_t)w * K + k;
                            double2 s = make_double2(0.0, 0.0);   // lda.cpp:169
                            if (len > 0)
                                s = make_double2(acc[h].x * e2[h].x, acc[h].y * e2[h].y);
                            const double2 lam = update_pair<EMIT>(o, i, s, lp[h]);
                            rs[kb][h].x += lam.x;
                            rs[kb][h].y += lam.y;
                        }
                    }
                }
            }
        }
        if (o.partial) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const int k = kb * BW + 128 * h + 2 * lane;
                    if (k < K)
                        *reinterpret_cast<double2 *>(wpart2 + wid * K + k) = rs[kb][h];
                }
            __syncthreads();
            for (int k = threadIdx.x; k < K; k += T) {
                double sum = wpart2[k];
                for (int c = 1; c < W; ++c)
                    sum += wpart2[c * K + k];
                store_partial(o, (size_t)bid * K + k, sum);
            }
            finish_partial_groups<T>(o, K, bid, vl.n_rows);
        }
        return;
    }

    // ---- segments of the very long lists, as in 4c
    const int G_long = (int)gridDim.x - G_short - vl.G_seg;
    if (bid >= G_short + G_long) {
        for (int t = bid - G_short - G_long; t < vl.n_tasks; t += vl.G_seg) {      // block-uniform
            const int4 tk = vl.task[vl.t0 + t];
            const int chunk = (tk.w + W - 1) / W;
            const int c0 = __builtin_amdgcn_readfirstlane(min(tk.w, wid * chunk));
            const int c1 = __builtin_amdgcn_readfirstlane(min(tk.w, c0 + chunk));
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                if (kb * BW < K) {
                    double2 acc[NH];
#pragma unroll
                    for (int h = 0; h < NH; ++h)
                        acc[h] = make_double2(0.0, 0.0);
                    word_segment_sum2<NH>(tk.z + c0, tk.z + c1, K, kb * BW, wdoc, tw_word, epg, acc);
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        const int k = kb * BW + 128 * h + 2 * lane;
                        if (k < K)
                            *reinterpret_cast<double2 *>(wpart2 + wid * K + k) = acc[h];
                    }
                }
            }
            segment_finish<T, W, EMIT>(o, vl, K, tk, wpart2, eeb);
        }
        return;
    }

    // ---- long lists, as in 4c
    double rsl[(512 + T - 1) / T];
#pragma unroll
    for (int c = 0; c < (512 + T - 1) / T; ++c)
        rsl[c] = 0.0;
    for (int lw = bid - G_short; lw < n_long; lw += G_long) {
        const int w = long_words[lw];
        const int base = __builtin_amdgcn_readfirstlane(wptr[w]);
        const int L = __builtin_amdgcn_readfirstlane(wptr[w + 1]) - base;
        if (vl.G_seg > 0 && L > vl.seg_len)
            continue;                                // cut into segments (above)
        const int chunk = (L + W - 1) / W;
        const int c0 = __builtin_amdgcn_readfirstlane(min(L, wid * chunk));
        const int c1 = __builtin_amdgcn_readfirstlane(min(L, c0 + chunk));
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb * BW < K) {
                double2 acc[NH];
#pragma unroll
                    for (int h = 0; h < NH; ++h)
                        acc[h] = make_double2(0.0, 0.0);
                word_segment_sum2<NH>(base + c0, base + c1, K, kb * BW, wdoc, tw_word, epg, acc);
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const int k = kb * BW + 128 * h + 2 * lane;
                    if (k < K)
                        *reinterpret_cast<double2 *>(wpart2 + wid * K + k) = acc[h];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < (512 + T - 1) / T; ++c) {
            const int k = threadIdx.x + c * T;
            if (k < K) {
                // (chunk order; EMIT: read and added one after the other -- sixteen values held at
                // once are 32 of the 64 VGPRs this instantiation may use)
                double acc;
                if constexpr (EMIT) {
                    acc = wpart2[k];
#pragma unroll 4
                    for (int q = 1; q < W; ++q)
                        acc += wpart2[q * K + k];
                } else {
                    double pv[W];
#pragma unroll
                    for (int q = 0; q < W; ++q)
                        pv[q] = wpart2[q * K + k];
                    acc = pv[0];
#pragma unroll
                    for (int q = 1; q < W; ++q)
                        acc += pv[q];
                }
                const size_t i = (size_t)w * K + k;
                rsl[c] += update_one<EMIT>(o, i, acc * eeb[i]);
            }
        }
        __syncthreads();
    }
    if (o.partial) {
#pragma unroll
        for (int c = 0; c < (512 + T - 1) / T; ++c) {
            const int k = threadIdx.x + c * T;
            if (k < K)
                store_partial(o, (size_t)bid * K + k, rsl[c]);
        }
        finish_partial_groups<T>(o, K, bid, vl.n_rows);
    }
}

// 4b. Atomic mode finish: sstats *= eeb (lda.cpp:217): FinishOp in stream_kernels.h.

// psi[i] = psi(x[i]); epsi[i] = exp(psi(x[i])) as the document kernels compute it (no
// logarithm); epsi_lean[i] = the call-free form for positive arguments (psi.h,
// exp_digamma_positive: what the M-step kernels emit); eminus[i] =
// exp(psi(x[i]) - c): test hook for the device special functions
// (tests/test_gpu_parity.py::test_device_digamma_table).
__global__ void digamma_table_kernel(int n, double c, const double *__restrict__ x,
                                     double *__restrict__ psi, double *__restrict__ epsi,
                                     double *__restrict__ epsi_lean, double *__restrict__ eminus)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const double v = x[i];
    psi[i] = digamma(v);
    epsi[i] = exp_digamma(v);
    epsi_lean[i] = v > 0.0 ? exp_digamma_positive(v) : exp_digamma(v);   // the M-step's form
    eminus[i] = exp_digamma_minus(v, c);
}

// ---------------------------------------------------------------------------
// M-step kernels.
// ---------------------------------------------------------------------------

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
