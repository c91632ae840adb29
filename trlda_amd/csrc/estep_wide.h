// trlda_amd/csrc/estep_wide.h -- the document kernel for 128 < K <= 512 topics and for
// documents too long for the dual-orientation register kernel (estep_kernels.h, section 3c).
//
// Same fixed point as everywhere (reference code/trlda/src/lda.cpp:174-213), laid out for a
// slice beta_d (K x n_d, lda.cpp:179-181) that is too large to be kept twice:
//
//   * one workgroup of 8 wavefronts per document; lane l owns the topics l, l+64, ..,
//     l+64(KS-1) ("slots"); wave w owns the words w, w+8, w+16, ..  Up to 8*JW words live in
//     registers (JW*KS doubles per lane), the next ones in LDS rows, anything beyond that is
//     streamed from L2 every iteration -- no limit on the document length.
//   * phinorm_j = sum_k e_k beta_jk (lda.cpp:183/:199) is a sum ACROSS lanes.  A wave reduces 16
//     words at once with a transposing butterfly (v_permlane32_swap / v_permlane16_swap, then
//     DPP row rotations): 57 instructions for 16 sums instead of 16 x 18, and the 16 results
//     end up one per quad of lanes, where cnt_j / phinorm_j is formed.
//   * acc_k = sum_j (cnt_j / phinorm_j) beta_jk (lda.cpp:189-193) is lane-local: the weight
//     of word j is handed to every lane with v_readlane and multiplies the registers.  Only
//     the 8 per-wave partial sums meet in LDS: two barriers per iteration.
//   * gamma, alpha and exp(psi(gamma)) of topic k live in the registers of thread k.
//
// Additions happen in a fixed order (no atomics): results are reproducible run to run.
#pragma once

#include "estep_kernels.h"

namespace trlda {

constexpr int kWideThreads = 512;
constexpr int kWideWaves = kWideThreads / kWave;     // 8
constexpr int kWideMaxK = 512;

template <int KS>
struct wide_cfg {
    // words per wave held in registers
    static constexpr int JW = KS <= 2 ? 32 : KS == 3 ? 24 : KS == 4 ? 20 : KS == 5 ? 16
                              : KS == 6 ? 12 : KS == 7 ? 11 : 10;
    static constexpr int NG = (JW + 15) / 16;        // groups of 16 words (one fold each)
    static constexpr int NSET = KS >= 8 ? 1 : KS >= 4 ? 2 : KS >= 2 ? 4 : 8;   // >= 8 fma chains
    static constexpr int TCH = KS <= 4 ? 4 : 2;      // words of a tail chunk
    static constexpr int KP = 64 * KS;               // padded topic count
};

// LDS carve (doubles): part[8][KP] | ebuf[2][KP] | misc[2][8] | cnt_tail[tail] | rows[tail][KP]
__host__ __device__ constexpr size_t wide_lds_doubles(int KS, int tail_words)
{
    return (size_t)(kWideWaves + 2) * 64 * KS + 16 + (size_t)tail_words * (64 * KS + 1);
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

// sum over the quad (lanes l ^ 1, l ^ 2): the value every lane holds afterwards
__device__ __forceinline__ double quad_sum(double v)
{
    const int lo2 = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x4e, 0xf, 0xf, true);
    const int hi2 = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x4e, 0xf, 0xf, true);
    v += __hiloint2double(hi2, lo2);
    const int lo1 = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xb1, 0xf, 0xf, true);
    const int hi1 = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xb1, 0xf, 0xf, true);
    return v + __hiloint2double(hi1, lo1);
}

// Sums over the 64 lanes of 16 values at once.  Lane l receives the total of value
//   fold16_index(l) = bit5(l) + 2 bit4(l) + 4 bit3(l) + 8 bit2(l)
// (the same in all four lanes of a quad).  NV: values at or beyond it are known to be zero
// and their steps are skipped at compile time.
template <int NV>
__device__ __forceinline__ double fold16(const double (&v)[16])
{
    double a[8], b[4], c[2];
#pragma unroll
    for (int p = 0; p < 8; ++p)
        a[p] = (2 * p < NV) ? fold<32>(v[2 * p], v[2 * p + 1]) : 0.0;
#pragma unroll
    for (int p = 0; p < 4; ++p)
        b[p] = (4 * p < NV) ? fold<16>(a[2 * p], a[2 * p + 1]) : 0.0;
#pragma unroll
    for (int p = 0; p < 2; ++p)
        c[p] = (8 * p < NV) ? fold<8>(b[2 * p], b[2 * p + 1]) : 0.0;
    return quad_sum(fold<4>(c[0], c[1]));
}
__device__ __forceinline__ int fold16_index(int lane)
{
    return ((lane >> 5) & 1) | ((lane >> 3) & 2) | ((lane >> 1) & 4) | ((lane << 1) & 8);
}
__host__ __device__ constexpr int fold16_lane(int idx)
{
    return ((idx & 1) << 5) | ((idx & 2) << 3) | ((idx & 4) << 1) | ((idx & 8) >> 1);
}

// The same for a chunk of 4 (or 2) values: index = bit5 + 2 bit4 (or bit5), all-reduce below.
template <int N>
__device__ __forceinline__ double fold_chunk(const double (&v)[N])
{
    static_assert(N == 4 || N == 2, "chunk size");
    double r;
    if constexpr (N == 4) {
        const double a0 = fold<32>(v[0], v[1]), a1 = fold<32>(v[2], v[3]);
        r = fold<16>(a0, a1);
    } else {
        const double a0 = fold<32>(v[0], v[1]);
        r = fold<16>(a0, a0);
    }
    r = fold<8>(r, r);
    r = fold<4>(r, r);
    return quad_sum(r);
}
template <int N>
__device__ __forceinline__ int fold_chunk_index(int lane)
{
    return N == 4 ? (((lane >> 5) & 1) | ((lane >> 3) & 2)) : ((lane >> 5) & 1);
}
template <int N>
__host__ __device__ constexpr int fold_chunk_lane(int idx)
{
    return N == 4 ? (((idx & 1) << 5) | ((idx & 2) << 3)) : ((idx & 1) << 5);
}

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// diagnostic: out[l] = fold16<16> of in[l][0..15] (tests/test_gpu_parity.py)
__global__ void debug_fold16_kernel(const double *in, double *out, double *out4, double *out2)
{
    const int lane = threadIdx.x;
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
        v[i] = in[lane * 16 + i];
    out[lane] = fold16<16>(v);
    const double v4[4] = {v[0], v[1], v[2], v[3]};
    out4[lane] = fold_chunk<4>(v4);
    const double v2[2] = {v[0], v[1]};
    out2[lane] = fold_chunk<2>(v2);
}

// ---------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(kWideThreads) void estep_docs_wide_kernel(DocKernelArgs a, int tail_cap)
{
    using cfg = wide_cfg<KS>;
    constexpr int JW = cfg::JW, NG = cfg::NG, NSET = cfg::NSET, TCH = cfg::TCH, KP = cfg::KP;
    constexpr int W = kWideWaves;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(tid / kWave);

    const int d = a.order[blockIdx.x];
    const int K = a.K;
    const int p0 = a.indptr[d];
    const int n = a.indptr[d + 1] - p0;
    const int32_t *__restrict__ ids = a.ids + p0;
    const int32_t *__restrict__ cnts = a.cnts + p0;

    double *part = lds;                              // 8 x KP
    double *ebuf = part + W * KP;                    // 2 x KP
    double *misc = ebuf + 2 * KP;                    // 2 x 8
    double *cnt_tail = misc + 16;                    // tail_cap
    double *rows = cnt_tail + tail_cap;              // tail_cap x KP

    const int n_reg = min(n, W * JW);                // words in registers: j = i * 8 + wid
    const int n_lds = min(n - n_reg, tail_cap);      // words in LDS rows
    // register words of a wave are used in chunks of four: bound (wave-uniform, block-uniform)
    const int JE = min(JW, (((n_reg + W - 1) / W) + 3) & ~3);

    bool kv[KS];                                     // topic l + 64 s exists
#pragma unroll
    for (int s = 0; s < KS; ++s)
        kv[s] = lane + 64 * s < K;

    // ---- the slice (lda.cpp:179-181): JW x KS coalesced loads per lane, all independent.
    // Word ids: one vector load (lane i -> word i of this wave), handed out with v_readlane.
    double beta[JW][KS];
    {
        const int jl = (lane < JW ? lane : 0) * W + wid;
        const int myid = n > 0 ? ids[min(jl, n - 1)] : 0;
#pragma unroll
        for (int i = 0; i < JW; ++i) {
            const double *rowp = a.eeb + (size_t)__builtin_amdgcn_readlane(myid, i) * K;
#pragma unroll
            for (int s = 0; s < KS; ++s)
                beta[i][s] = rowp[min(lane + 64 * s, K - 1)];
        }
    }
    // counts of the words whose phinorm this lane will hold after the fold
    double cntv[NG];
    int jv[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int i = 16 * g + fold16_index(lane);
        jv[g] = i * W + wid;
        const bool ok = i < JW && jv[g] < n_reg;
        cntv[g] = 0.0;
        if (ok)
            cntv[g] = (double)cnts[jv[g]];
        jv[g] = ok ? jv[g] : -1;
    }
    // gamma / alpha / exp(psi(gamma)) of topic tid                       lda.cpp:174
    const bool k_on = tid < K;
    double gk = 1.0, ak = 0.0, ek = 0.0;
    if (tid < KP) {
        if (k_on) {
            gk = a.gamma_in[(size_t)d * K + tid];
            ak = a.alpha[tid];
            ek = exp_digamma(gk);
        }
        ebuf[tid] = ek;                              // zero beyond K
    }
    // tail rows in LDS, zero beyond K
    for (int t = wid; t < n_lds; t += W) {
        const double *rowp = a.eeb + (size_t)ids[n_reg + t] * K;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const double v = rowp[min(lane + 64 * s, K - 1)];
            rows[(size_t)t * KP + lane + 64 * s] = kv[s] ? v : 0.0;
        }
        if (lane == 0)
            cnt_tail[t] = (double)cnts[n_reg + t];
    }
#pragma unroll
    for (int i = 0; i < JW; ++i) {
        const bool row = i * W + wid < n_reg;
#pragma unroll
        for (int s = 0; s < KS; ++s)
            beta[i][s] = (row && kv[s]) ? beta[i][s] : 0.0;
    }
    __syncthreads();

    double twv[NG];                                  // cnt / phinorm of word jv[g]
    double acc[NSET][KS];
    double e[KS];
    int it = 0;
    int cur = 0;                                     // ebuf / misc buffer holding the current e
    double mean_change = 0.0;
    for (;;) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
            e[s] = ebuf[cur * KP + lane + 64 * s];
        if (it > 0) {                                // mean |gamma - last|        lda.cpp:202
            double sum = 0.0;
#pragma unroll
            for (int w = 0; w < (KS < W ? KS : W); ++w)
                sum += misc[cur * 8 + w];
            mean_change = sum / (double)K;
        }
#pragma unroll
        for (int u = 0; u < NSET; ++u)
#pragma unroll
            for (int s = 0; s < KS; ++s)
                acc[u][s] = 0.0;

        // ---- phinorm and cnt / phinorm of the register words           lda.cpp:183 / :199
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            double sv[16];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool live = 16 * g + 4 * c < JE;           // wave-uniform
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    sv[4 * c + u] = 0.0;
                if (16 * g + 4 * c < JW && live) {
#pragma unroll
                    for (int s = 0; s < KS; ++s)
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int i = 16 * g + 4 * c + u;
                            if (i < JW)
                                sv[4 * c + u] = fma(e[s], beta[i < JW ? i : 0][s], sv[4 * c + u]);
                        }
                }
            }
            const double tot = (g == NG - 1 && (JW & 15) != 0) ? fold16<(JW & 15)>(sv)
                                                               : fold16<16>(sv);
            twv[g] = cntv[g] * rcp_pos<true>(tot + 1e-100);
        }

        // ---- words in LDS rows and streamed words: phinorm, weight and the update of acc
        // from one read of the row
        auto tail_chunk = [&](int t0, bool from_lds) {
            // words j_t = n_reg + t0 + u, u < TCH (those at or beyond n contribute zero)
            double r[TCH][KS];
            const int my_u = fold_chunk_index<TCH>(lane);
            double my_cnt;
            if (from_lds) {
#pragma unroll
                for (int u = 0; u < TCH; ++u) {
                    const int t = min(t0 + u, n_lds - 1);
#pragma unroll
                    for (int s = 0; s < KS; ++s)
                        r[u][s] = rows[(size_t)t * KP + lane + 64 * s];
                }
                my_cnt = t0 + my_u < n_lds ? cnt_tail[t0 + my_u] : 0.0;
            } else {
#pragma unroll
                for (int u = 0; u < TCH; ++u) {
                    const int j = min(n_reg + t0 + u, n - 1);
                    const double *rowp = a.eeb + (size_t)ids[j] * K;
#pragma unroll
                    for (int s = 0; s < KS; ++s)
                        r[u][s] = rowp[min(lane + 64 * s, K - 1)];
                }
                const int j = n_reg + t0 + my_u;
                my_cnt = j < n ? (double)cnts[j] : 0.0;
#pragma unroll
                for (int u = 0; u < TCH; ++u)
#pragma unroll
                    for (int s = 0; s < KS; ++s)
                        r[u][s] = kv[s] ? r[u][s] : 0.0;
            }
            double sv[TCH];
#pragma unroll
            for (int u = 0; u < TCH; ++u) {
                sv[u] = 0.0;
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    sv[u] = fma(e[s], r[u][s], sv[u]);
            }
            const double tw = my_cnt * rcp_pos<true>(fold_chunk<TCH>(sv) + 1e-100);
            if ((lane & (TCH == 4 ? 15 : 31)) == 0) {     // one lane per word of the chunk
                const int j = n_reg + t0 + my_u;
                if (from_lds ? t0 + my_u < n_lds : j < n)
                    a.tw_csr[p0 + j] = tw;
            }
#pragma unroll
            for (int u = 0; u < TCH; ++u) {
                const double twu = readlane_f64(tw, fold_chunk_lane<TCH>(u));
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    acc[u % NSET][s] = fma(twu, r[u][s], acc[u % NSET][s]);
            }
        };
        for (int t0 = wid * TCH; t0 < n_lds; t0 += W * TCH)
            tail_chunk(t0, true);
        for (int t0 = n_lds + wid * TCH; n_reg + t0 < n; t0 += W * TCH)
            tail_chunk(t0, false);

        if (it >= a.max_iter || (it > 0 && mean_change < a.threshold))    // lda.cpp:185, :202-203
            break;

        // ---- acc_k = sum_j tw_j beta[j][k] over this wave's words        lda.cpp:189-193
#pragma unroll
        for (int c = 0; c < (JW + 3) / 4; ++c) {
            if (4 * c < JE) {                        // wave-uniform
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = 4 * c + u;
                    if (i < JW) {
                        const double twi = readlane_f64(twv[i / 16], fold16_lane(i & 15));
#pragma unroll
                        for (int s = 0; s < KS; ++s)
                            acc[i % NSET][s] = fma(twi, beta[i][s], acc[i % NSET][s]);
                    }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            double v = acc[0][s];
            if constexpr (NSET == 2)
                v = acc[0][s] + acc[1][s];
            if constexpr (NSET == 4)
                v = (acc[0][s] + acc[1][s]) + (acc[2][s] + acc[3][s]);
            if constexpr (NSET == 8)
                v = ((acc[0][s] + acc[1][s]) + (acc[2][s] + acc[3][s])) +
                    ((acc[4][s] + acc[5][s]) + (acc[6][s] + acc[7][s]));
            part[wid * KP + lane + 64 * s] = v;
        }
        __syncthreads();

        // ---- gamma_k = alpha_k + e_k acc_k ; e_k = exp(psi(gamma_k))     lda.cpp:194-197
        const int nxt = cur ^ 1;
        if (wid < KS) {                              // tid < KP
            const double accs = sum8_strided<KP>(part + tid);
            const double gnew = k_on ? fma(accs, ek, ak) : 1.0;
            const double diff = k_on ? fabs(gk - gnew) : 0.0;
            gk = gnew;
            const double enew = exp_digamma(gnew);
            ek = k_on ? enew : 0.0;
            ebuf[nxt * KP + tid] = ek;
            const double dsum = wave_sum_dpp(diff);
            if (lane == 0)
                misc[nxt * 8 + wid] = dsum;
        }
        __syncthreads();
        cur = nxt;
        ++it;
    }

    // ---- results
    if (k_on) {
        a.gamma[(size_t)d * K + tid] = gk;
        a.epg[(size_t)d * K + tid] = ek;
    }
    if (tid == 0 && a.iters_out)
        a.iters_out[d] = it;
    if (a.sstats_acc) {                              // lda.cpp:207-213, atomic form
        __syncthreads();                             // tw_csr of the tail words
#pragma unroll
        for (int i = 0; i < JW; ++i) {
            const double twi = readlane_f64(twv[i / 16], fold16_lane(i & 15));
            const int j = i * W + wid;
            if (j < n_reg) {
                double *col = a.sstats_acc + (size_t)ids[j] * K;
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    if (kv[s])
                        unsafeAtomicAdd(&col[lane + 64 * s], twi * e[s]);
            }
        }
        for (int j = n_reg + wid; j < n; j += W) {
            const double twj = a.tw_csr[p0 + j];
            double *col = a.sstats_acc + (size_t)ids[j] * K;
#pragma unroll
            for (int s = 0; s < KS; ++s)
                if (kv[s])
                    unsafeAtomicAdd(&col[lane + 64 * s], twj * e[s]);
        }
    } else {
        if ((lane & 3) == 0) {
#pragma unroll
            for (int g = 0; g < NG; ++g)
                if (jv[g] >= 0)
                    a.tw_word[a.wrank[p0 + jv[g]]] = twv[g];
        }
        __syncthreads();                             // tw_csr of the tail words
        for (int j = n_reg + tid; j < n; j += kWideThreads)
            a.tw_word[a.wrank[p0 + j]] = a.tw_csr[p0 + j];
    }
}

}  // namespace trlda
