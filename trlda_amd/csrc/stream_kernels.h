// stream_kernels.h -- the dense K x V passes of the path on big tables, written to run at
// HBM speed on gfx950 (MI355X): row sums of lambda (reference src/lda.cpp:172), the
// trust-region initial step and the decay of the words a mini-batch does not touch
// (src/onlinelda.cpp:79-86, :99-100), and the plain element-wise M-step passes
// (src/onlinelda.cpp:99-100, src/batchlda.cpp:60, src/cumulativelda.cpp:70).
//
// Column-slot streaming.  Matrices are column-major (a word's K values contiguous).  A pass
// that also needs the row sums sum_w x[k, w] wants every thread to meet ONE topic only, so that
// the running sum lives in a register.  With VEC = 2 doubles per access when K is even (one
// 16-byte load per lane), P = K / VEC accesses cover a column; a workgroup of T threads holds
// cpb = floor(T / P) column slots, thread (slot, kp) walks the columns
//     col = (block * cpb + slot) + r * m,   m = gridDim.x * cpb,   r = 0, 1, ..
// at the fixed offset kp * VEC: consecutive threads of a workgroup read consecutive 16-byte
// pieces of cpb adjacent columns (fully coalesced), U columns are in flight per thread
// (unconditional loads of clamped columns, masked afterwards), and the block's slots are
// added up through LDS in slot order into partial[block][k] -- a fixed order, so the row sums
// are reproducible run to run.  512 workgroups of 1024 threads (two per CU) keep 16 B x 8
// x 2048 = 256 KiB in flight per CU.
#pragma once
#include <algorithm>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "psi.h"

namespace trlda {

constexpr int kStreamThreads = 1024;
constexpr int kStreamMaxBlocks = 512;
constexpr int kStreamUnroll = 8;

struct StreamGeom {
    int vec;      // doubles per access: 2 when K is even, else 1
    int P;        // accesses per column
    int cpb;      // column slots per workgroup
    int G;        // workgroups
};

// K <= kStreamThreads * vec (callers keep K <= 512)
inline StreamGeom stream_geometry(int K, int V)
{
    StreamGeom g;
    g.vec = (K % 2 == 0) ? 2 : 1;
    g.P = K / g.vec;
    g.cpb = kStreamThreads / g.P;
    const long long cols_per_round = (long long)g.cpb * kStreamUnroll;
    long long G = (V + cols_per_round - 1) / cols_per_round;   // >= one unrolled pass each
    // (a small table -- K = 100, V = 7000: 44 blocks by that rule -- still wants every CU: one
    // workgroup per CU as long as each gets a column slot's worth; round 4: the update call's
    // initial step 14.8 us for 11 MB on 44 CUs)
    if (G < 256)
        G = std::max<long long>(G, std::min<long long>(256, (V + g.cpb - 1) / g.cpb));
    if (G > kStreamMaxBlocks)
        G = kStreamMaxBlocks;
    if (G < 1)
        G = 1;
    g.G = (int)G;
    return g;
}

template <int VEC>
struct vec_t;
template <>
struct vec_t<1> {
    using type = double;
};
template <>
struct vec_t<2> {
    using type = double2;
};

template <int VEC>
__device__ __forceinline__ typename vec_t<VEC>::type vload(const double *p)
{
    return *reinterpret_cast<const typename vec_t<VEC>::type *>(p);
}
// the same load with the non-temporal hint: for bytes that are read once and must not push
// re-used lines (exp E[log beta], exp(psi(gamma))) out of L2 / the Infinity Cache
template <int VEC>
__device__ __forceinline__ typename vec_t<VEC>::type vload_nt(const double *p)
{
    if constexpr (VEC == 2) {
        double2 r;
        r.x = __builtin_nontemporal_load(p);
        r.y = __builtin_nontemporal_load(p + 1);
        return r;
    } else {
        return __builtin_nontemporal_load(p);
    }
}
template <int VEC>
__device__ __forceinline__ void vstore(double *p, typename vec_t<VEC>::type v)
{
    *reinterpret_cast<typename vec_t<VEC>::type *>(p) = v;
}
__device__ __forceinline__ double vget(double v, int) { return v; }
__device__ __forceinline__ double vget(double2 v, int i) { return i ? v.y : v.x; }
__device__ __forceinline__ void vset(double &v, int, double x) { v = x; }
__device__ __forceinline__ void vset(double2 &v, int i, double x)
{
    if (i)
        v.y = x;
    else
        v.x = x;
}

// block partial of per-thread row sums: acc[VEC] of thread (slot, kp) -> partial[block][k],
// slots added in slot order.  scratch: cpb * K doubles of LDS (<= T * VEC).
template <int T, int VEC>
__device__ __forceinline__ void stream_block_partial(int K, int P, int cpb, int slot, int kp,
                                                     const double (&acc)[VEC], double *scratch,
                                                     double *__restrict__ partial_row)
{
    if (slot < cpb) {
#pragma unroll
        for (int v = 0; v < VEC; ++v)
            scratch[slot * K + kp * VEC + v] = acc[v];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += T) {
        double s = scratch[k];
        for (int sl = 1; sl < cpb; ++sl)
            s += scratch[sl * K + k];
        partial_row[k] = s;
    }
}

// ---------------------------------------------------------------------------
// Row sums of lambda (lda.cpp:172): partial[block][k] = sum over the block's columns.
// ---------------------------------------------------------------------------
template <int T, int VEC, int U = kStreamUnroll, bool NT = true>
__global__ __launch_bounds__(T) void rowsum_stream_kernel(int K, int V, int P, int cpb,
                                                          const double *__restrict__ lambda,
                                                          double *__restrict__ partial)
{
    extern __shared__ double scratch[];
    using V_t = typename vec_t<VEC>::type;
    const int slot = threadIdx.x / P, kp = threadIdx.x - slot * P;
    const int m = gridDim.x * cpb;
    double acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v)
        acc[v] = 0.0;
    if (slot < cpb) {
        const double *base = lambda + (size_t)kp * VEC;
        for (int col = blockIdx.x * cpb + slot; col < V; col += U * m) {
            V_t x[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                x[u] = NT ? vload_nt<VEC>(base + (size_t)min(col + u * m, V - 1) * K)
                          : vload<VEC>(base + (size_t)min(col + u * m, V - 1) * K);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool on = col + u * m < V;
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    acc[v] += on ? vget(x[u], v) : 0.0;
            }
        }
    }
    stream_block_partial<T, VEC>(K, P, cpb, slot, kp, acc, scratch, partial + (size_t)blockIdx.x * K);
}

// out[k] = psi(rs[k]), out[K + k] = rs[k], out[2 K + k] = exp(-psi(rs[k])): what topic_scale_combine
// leaves behind for the K <= 128 kernels (estep_kernels.h), here from finished row sums -- for the
// single-orientation document kernel when the M-step has left exp(psi(lambda)) behind (round 4)
template <int T>
__global__ __launch_bounds__(T) void topic_factors_kernel(int K, const double *__restrict__ rs,
                                                          double *__restrict__ out)
{
    const int k = blockIdx.x * T + threadIdx.x;
    if (k >= K)
        return;
    const double s = rs[k];
    const double ps = digamma(s);
    out[k] = ps;
    out[K + k] = s;
    out[2 * K + k] = exp(-ps);
}

// combined[k] = (base ? base[k] : 0) + sum_{g < G} partial[g][k]: one wavefront per topic, lane
// l adds the rows l, l + 64, .. (eight loads in flight), the 64 lane sums are combined by a
// fixed butterfly (xor shuffles) -- every lane ends with the same value.  G <= a few thousand.
template <int T>
__global__ __launch_bounds__(T) void rowsum_combine_wave_kernel(int K, int G,
                                                                const double *__restrict__ partial,
                                                                const double *__restrict__ base,
                                                                double *__restrict__ combined)
{
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * (T / 64) + threadIdx.x / 64;
    if (k >= K)
        return;
    double acc[2] = {0.0, 0.0};
    for (int g = lane; g < G; g += 64 * 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            v[u] = partial[(size_t)min(g + 64 * u, G - 1) * K + k];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            acc[u & 1] += (g + 64 * u < G) ? v[u] : 0.0;
    }
    double s = acc[0] + acc[1];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        s += __shfl_xor(s, off, 64);
    if (lane == 0)
        combined[k] = (base ? base[k] : 0.0) + s;
}

// ---------------------------------------------------------------------------
// The words a mini-batch does NOT touch, and the hand-over of those it does.
//
// Inside one update (src/onlinelda.cpp:68-110) lambda' and rho are fixed and the sufficient
// statistics of a word outside the batch are zero (src/lda.cpp:169), so for such a word every
// M-step of the trust-region loop writes the same value
//     lambda[:, w] = (1 - rho) lambda'[:, w] + rho eta
// -- which is also what the initial step of :85-86 gives it (its word count is zero).  It is
// written ONCE, in place, by this pass; the loop then only ever touches the batch's active
// words.  For an active word the pass saves lambda' (the loop's M-steps need it) and either
// applies the initial step (ACT_TRINIT, :85-86) or leaves lambda alone (ACT_KEEP: the E-step
// that follows still has to see the old value; src/onlinelda.cpp:103-109, src/batchlda.cpp:60).
// The row sums of what lambda holds afterwards come out in two parts: `part_static` over the
// inactive words (valid for the whole update) and, with ACT_TRINIT, `part_active` over the
// active ones (replaced by every M-step, see sstats_update_kernel).
//   a, b        inactive word: lambda = a * lambda + b   (a = 1 - rho, b = rho * eta;
//               BatchLDA: a = 0 -> lambda = b = eta without reading it)
//   SAVE_ALL    also copy the inactive columns into lambda' (adaptive learning rate: it needs
//               the whole lambda', src/onlinelda.cpp:167-175)
// ---------------------------------------------------------------------------
constexpr int ACT_KEEP = 0, ACT_TRINIT = 1;

//   src           what is read (lambda itself for the in-place forms; a separate lambda' for the
//                 multi-GPU composition's trlda_model_tr_init_wc)
//   active_flag   V bytes, or nullptr: every word counts as active
//   lambda_prime  where the active columns of src are saved, or nullptr: no copy
// ACT_KEEP without a copy does not read the active columns at all: the flags of the U columns of
// a pass are fetched first (V bytes: cache hits), the column loads are predicated on them.
template <int T, int VEC, int ACT, bool SAVE_ALL>
__global__ __launch_bounds__(T) void inactive_update_stream_kernel(
    int K, int V, int P, int cpb, double a, double b, double rho, double eta, double coef,
    const uint8_t *__restrict__ active_flag, const double *__restrict__ wordcounts,
    const int32_t *__restrict__ wordcounts_i32 /* or nullptr: the doubles */,
    const double *src, double *lambda, double *lambda_prime, double *__restrict__ part_static,
    double *__restrict__ part_active)
{
    extern __shared__ double scratch[];
    constexpr int U = kStreamUnroll;
    using V_t = typename vec_t<VEC>::type;
    const int slot = threadIdx.x / P, kp = threadIdx.x - slot * P;
    const int m = gridDim.x * cpb;
    double acc_s[VEC], acc_a[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v)
        acc_s[v] = acc_a[v] = 0.0;
    const bool read_active = ACT == ACT_TRINIT || lambda_prime != nullptr;   // launch-uniform
    const bool read_inactive = a != 0.0 || SAVE_ALL;
    if (slot < cpb) {
        const size_t off = (size_t)kp * VEC;
        for (int col = blockIdx.x * cpb + slot; col < V; col += U * m) {
            V_t x[U];
            bool fl[U];
            double wc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = min(col + u * m, V - 1);
                fl[u] = active_flag ? active_flag[c] != 0 : true;
                wc[u] = ACT != ACT_TRINIT ? 0.0 : wordcounts_i32 ? (double)wordcounts_i32[c] : wordcounts[c];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = min(col + u * m, V - 1);
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    vset(x[u], v, 0.0);
                if (fl[u] ? read_active : read_inactive)
                    x[u] = vload_nt<VEC>(src + (size_t)c * K + off);   // read once
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = col + u * m;
                if (c < V) {
                    double *lp = lambda + (size_t)c * K + off;
                    if (fl[u]) {
                        if (lambda_prime)                // lambdaPrime = mLambda   onlinelda.cpp:68
                            vstore<VEC>(lambda_prime + (size_t)c * K + off, x[u]);
                        if (ACT == ACT_TRINIT) {         // onlinelda.cpp:85-86
                            const double add = rho * (eta + coef * wc[u]);
                            V_t y;
#pragma unroll
                            for (int v = 0; v < VEC; ++v) {
                                const double yv = (1. - rho) * vget(x[u], v) + add;
                                vset(y, v, yv);
                                acc_a[v] += yv;
                            }
                            vstore<VEC>(lp, y);
                        }
                    } else {
                        if (SAVE_ALL)
                            vstore<VEC>(lambda_prime + (size_t)c * K + off, x[u]);
                        V_t y;
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                            const double yv = a != 0.0 ? a * vget(x[u], v) + b : b;
                            vset(y, v, yv);
                            acc_s[v] += yv;
                        }
                        vstore<VEC>(lp, y);
                    }
                }
            }
        }
    }
    stream_block_partial<T, VEC>(K, P, cpb, slot, kp, acc_s, scratch,
                                 part_static + (size_t)blockIdx.x * K);
    __syncthreads();
    if (ACT == ACT_TRINIT)
        stream_block_partial<T, VEC>(K, P, cpb, slot, kp, acc_a, scratch,
                                     part_active + (size_t)blockIdx.x * K);
}

// ---------------------------------------------------------------------------
// Element-wise passes without a topic index: flat, 16 bytes per lane where the length allows
// (every buffer comes from hipMalloc: 256-byte aligned), U accesses in flight.
//   f(i) -> value stored at out[i]
// ---------------------------------------------------------------------------
template <int T, class F>
__global__ __launch_bounds__(T) void elementwise_stream_kernel(size_t total, F f)
{
    constexpr int U = 4;
    const size_t pairs = total / 2;
    const size_t stride = (size_t)gridDim.x * T;
    for (size_t j = (size_t)blockIdx.x * T + threadIdx.x; j < pairs; j += U * stride) {
        typename F::in_t x[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            x[u] = f.load(min(j + u * stride, pairs - 1));
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (j + u * stride < pairs)
                f.store(j + u * stride, x[u]);
    }
    if ((total & 1) && blockIdx.x == 0 && threadIdx.x == 0)
        f.tail(total - 1);
}

// lambda = (1-rho) lambda' + rho (eta + scale * sstats)     onlinelda.cpp:99-100
struct BlendOp {
    struct in_t {
        double2 lp, s;
    };
    double rho, eta, scale;
    const double *lambda_prime, *sstats;
    double *lambda;
    __device__ __forceinline__ double one(double lp, double s) const
    {
        const double hat = eta + scale * s;
        return (1. - rho) * lp + rho * hat;
    }
    __device__ __forceinline__ in_t load(size_t j) const
    {
        in_t r;
        r.lp = reinterpret_cast<const double2 *>(lambda_prime)[j];
        r.s = reinterpret_cast<const double2 *>(sstats)[j];
        return r;
    }
    __device__ __forceinline__ void store(size_t j, const in_t &x) const
    {
        reinterpret_cast<double2 *>(lambda)[j] = make_double2(one(x.lp.x, x.s.x), one(x.lp.y, x.s.y));
    }
    __device__ __forceinline__ void tail(size_t i) const { lambda[i] = one(lambda_prime[i], sstats[i]); }
};

// lambda = eta + sstats                                         batchlda.cpp:60
struct SetOp {
    struct in_t {
        double2 s;
    };
    double eta;
    const double *sstats;
    double *lambda;
    __device__ __forceinline__ in_t load(size_t j) const
    {
        in_t r;
        r.s = reinterpret_cast<const double2 *>(sstats)[j];
        return r;
    }
    __device__ __forceinline__ void store(size_t j, const in_t &x) const
    {
        reinterpret_cast<double2 *>(lambda)[j] = make_double2(eta + x.s.x, eta + x.s.y);
    }
    __device__ __forceinline__ void tail(size_t i) const { lambda[i] = eta + sstats[i]; }
};

// lambda = lambda' + sstats                                     cumulativelda.cpp:70
struct AccumulateOp {
    struct in_t {
        double2 lp, s;
    };
    const double *lambda_prime, *sstats;
    double *lambda;
    __device__ __forceinline__ in_t load(size_t j) const
    {
        in_t r;
        r.lp = reinterpret_cast<const double2 *>(lambda_prime)[j];
        r.s = reinterpret_cast<const double2 *>(sstats)[j];
        return r;
    }
    __device__ __forceinline__ void store(size_t j, const in_t &x) const
    {
        reinterpret_cast<double2 *>(lambda)[j] = make_double2(x.lp.x + x.s.x, x.lp.y + x.s.y);
    }
    __device__ __forceinline__ void tail(size_t i) const { lambda[i] = lambda_prime[i] + sstats[i]; }
};

// sstats *= eeb (atomic statistics mode)                        lda.cpp:217
struct FinishOp {
    struct in_t {
        double2 s, e;
    };
    const double *eeb;
    double *sstats;
    __device__ __forceinline__ in_t load(size_t j) const
    {
        in_t r;
        r.s = reinterpret_cast<const double2 *>(sstats)[j];
        r.e = reinterpret_cast<const double2 *>(eeb)[j];
        return r;
    }
    __device__ __forceinline__ void store(size_t j, const in_t &x) const
    {
        reinterpret_cast<double2 *>(sstats)[j] = make_double2(x.s.x * x.e.x, x.s.y * x.e.y);
    }
    __device__ __forceinline__ void tail(size_t i) const { sstats[i] *= eeb[i]; }
};

}  // namespace trlda
