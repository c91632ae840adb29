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
    // words per wave held in registers (JW * KS doubles per lane: about half of the register
    // file; the rest is needed by exp(psi) and the in-flight rows of the tail).  Chosen with
    // tools/jw_sweep.sh: more slots start to spill and lose.
#ifdef TRLDA_WIDE_JW
    static constexpr int JW = TRLDA_WIDE_JW;
#else
    static constexpr int JW = KS == 1 ? 32 : KS == 2 ? 24 : KS == 3 ? 16 : KS == 4 ? 12
                              : KS == 5 ? 12 : KS == 6 ? 10 : 8;
#endif
    // groups of 16 register slots, one transposing fold each (57 instructions for 16 sums:
    // 3.6 per word; groups of 8 cost 6.5 per word -- 87 against 57 instructions at JW = 12)
    static constexpr int NH = (JW + 15) / 16;
    static constexpr int NSET = KS >= 8 ? 1 : KS >= 4 ? 2 : KS >= 2 ? 4 : 8;   // >= 8 fma chains
    // words of a tail group (one read of their rows serves phinorm and the update of acc; ONE
    // transposing fold per group): TG * KS doubles in flight per lane
#ifdef TRLDA_WIDE_TG
    static constexpr int TCH = TRLDA_WIDE_TG;
#else
    // (16 per group at K <= 128 with 16 register words per wave was measured: a group costs
    // ~2000 cycles, LDS-bandwidth- and issue-bound like the register words, ~15 cycles per word
    // and workgroup either way -- but a document one word over the registers pays for a whole
    // group: 67 against 57 us at 145..192 words, 263 against 292 us at 600.  Small groups stay.)
    static constexpr int TCH = KS <= 4 ? 4 : 2;
#endif
    static constexpr int KP = 64 * KS;               // padded topic count
    // exp(psi)'s polynomial coefficients from scalar registers (psi.h, exp_nonpos<SC>): where the
    // vector registers run out -- at KS >= 7 the kernel spilled without it (1518 -> 1318 us per 4096
    // documents at K = 500); KS = 5, 6 gain 1.5 %, KS <= 4 lose 1 % (profiles/r04_scoef_*.txt)
#ifdef TRLDA_WIDE_SC
    static constexpr bool SC = KS >= TRLDA_WIDE_SC;
#else
    static constexpr bool SC = KS >= 5;
#endif
    // exp(psi)'s rational part by Horner's rule in x (psi.h, exp_psi_regular<.., XF>): the widest
    // variant spills four registers with the shorter form in u = x (x + 9)
    static constexpr bool XF = KS >= 8;
};

// LDS carve (doubles): part[8][KP] | ebuf[2][KP] | cbuf[KP] (topic factors, FACTORS at K > 128) |
// misc[2][8] | cnt_tail[lds_rows] |
// rows[lds_rows][KP]
__host__ __device__ constexpr size_t wide_lds_doubles(int KS, int lds_rows)
{
    return (size_t)(kWideWaves + 3) * 64 * KS + 16 + (size_t)lds_rows * (64 * KS + 1);
}

// (fold<D>, the transposing butterfly step, lives in estep_kernels.h: the register kernel uses
// it too.)

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

// Eight values: lane l receives the total of value bit5(l) + 2 bit4(l) + 4 bit3(l).
template <int NV>
__device__ __forceinline__ double fold8(const double (&v)[8])
{
    double a[4], b[2];
#pragma unroll
    for (int p = 0; p < 4; ++p)
        a[p] = (2 * p < NV) ? fold<32>(v[2 * p], v[2 * p + 1]) : 0.0;
#pragma unroll
    for (int p = 0; p < 2; ++p)
        b[p] = (4 * p < NV) ? fold<16>(a[2 * p], a[2 * p + 1]) : 0.0;
    const double c = fold<8>(b[0], b[1]);
    return quad_sum(fold<4>(c, c));
}
__device__ __forceinline__ int fold8_index(int lane)
{
    return ((lane >> 5) & 1) | ((lane >> 3) & 2) | ((lane >> 1) & 4);
}
__host__ __device__ constexpr int fold8_lane(int idx)
{
    return ((idx & 1) << 5) | ((idx & 2) << 3) | ((idx & 4) << 1);
}
// all 64 lanes receive the total
__device__ __forceinline__ double wave_sum_all(double v)
{
    v = fold<32>(v, v);
    v = fold<16>(v, v);
    v = fold<8>(v, v);
    return quad_sum(fold<4>(v, v));
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

// One fold for a group of N = 16, 8, 4 or 2 values; lane l receives the total of value
// fold_group_index<N>(l), found in every lane fold_group_lane<N>(i) + {0..3}.
template <int N>
__device__ __forceinline__ double fold_group(const double (&v)[N])
{
    if constexpr (N == 16)
        return fold16<16>(v);
    else if constexpr (N == 8)
        return fold8<8>(v);
    else
        return fold_chunk<N>(v);
}
template <int N>
__device__ __forceinline__ int fold_group_index(int lane)
{
    if constexpr (N == 16)
        return fold16_index(lane);
    else if constexpr (N == 8)
        return fold8_index(lane);
    else
        return fold_chunk_index<N>(lane);
}
template <int N>
__host__ __device__ constexpr int fold_group_lane(int idx)
{
    return N == 16 ? fold16_lane(idx) : N == 8 ? fold8_lane(idx) : fold_chunk_lane<N>(idx);
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
// The first 8 JW words of the document live in registers: word j belongs to wave j % 8, slot
// j / 8.  The words after them ("tail") are dealt to the waves in chunks of TCH consecutive
// words; their rows sit in LDS (the first lds_rows of them) or are streamed from eeb (L2) in
// every iteration -- one read of a row serves its phinorm and its update of acc.
// ---------------------------------------------------------------------------
// Document d (n words, CSR offset p0) on this workgroup.  FACTORS: the fused preamble's topic
// factors (K <= 128: a.partial / a.scale_in, estep_kernels.h 2b) are formed and folded into
// exp(psi(gamma)) as the register kernel does -- bitwise the same factors, so that documents of
// both kernels can share a launch and a statistics pass.
template <int KS, bool FACTORS>
__device__ __forceinline__ void estep_docs_wide_body(const DocKernelArgs &a, int lds_rows, double *lds,
                                                     int d, int p0, int n)
{
    using cfg = wide_cfg<KS>;
    constexpr int JW = cfg::JW, NH = cfg::NH, NSET = cfg::NSET, TCH = cfg::TCH, KP = cfg::KP;
    constexpr int W = kWideWaves;
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(tid / kWave);

    const int K = a.K;
    // fused preamble: the block partials of the row sums are fetched first (they depend on
    // nothing) and consumed once the row loads below are in flight
    [[maybe_unused]] double pv[2][8];
    if constexpr (FACTORS && KS <= 2) {              // (K > 128: the factors always come finished)
        if (a.partial && !a.scale_in)                // launch-uniform
            topic_scale_load<kWideThreads>(K, a.G, a.partial, pv);
    }
    const int32_t *__restrict__ ids = a.ids + p0;
    const int32_t *__restrict__ cnts = a.cnts + p0;

    double *part = lds;                              // 8 x KP
    double *ebuf = part + W * KP;                    // 2 x KP
    double *cbuf = ebuf + 2 * KP;                    // KP: c_k at K > 128 (two registers fewer in the loop:
                                                     // the K = 500 instantiation spilled four with them)
    double *misc = cbuf + KP;                        // 2 x 8
    double *cnt_tail = misc + 16;                    // lds_rows: counts of the words in LDS rows
    double *rows = cnt_tail + lds_rows;              // lds_rows x KP

    [[maybe_unused]] unsigned long long stamp_last = 0;
    TRLDA_STAMP_DECL;
#ifdef TRLDA_STAMPS
    stamp_last = __builtin_amdgcn_s_memtime();
#endif

    const int n_reg = min(n, W * JW);                // words in registers
    const int n_lds = min(n - n_reg, lds_rows);      // tail words with their row in LDS
    // register slots are used in chunks of four: bound (block-uniform)
    const int JE = min(JW, (((n_reg + W - 1) / W) + 3) & ~3);

    bool kv[KS];                                     // topic lane + 64 s exists
#pragma unroll
    for (int s = 0; s < KS; ++s)
        kv[s] = lane + 64 * s < K;

    // gamma0 / alpha (and the mirrored pair) are requested BEFORE the JW x KS row loads: loads
    // return in order, so behind the rows exp(psi(gamma0)) could not start until all had landed
    constexpr bool MIRROR = 2 * KS <= W;
    const int km = tid - KP;                         // the mirrored topic (below)
    const bool m_on = MIRROR && km >= 0 && km < K;
    const bool k_on = tid < K;
    double gk = 1.5, ak = 0.0;                       // (1.5: idle lanes must not take psi's integer branch)
    double gm = 0.0, am = 0.0;
    [[maybe_unused]] double ck = 1.0;                // topic factor exp(-psiSum_k) (FACTORS) or 1
    if (k_on) {
        gk = a.gamma_in[(size_t)d * K + tid];
        ak = a.alpha[tid];
        if constexpr (FACTORS) {
            if (a.scale_in && !a.scale_wait) {       // finished by the launch that prepared them
                ck = a.scale_in[2 * K + tid];
                if (doc_block(a) == 0 && a.scale_out) {
                    a.scale_out[tid] = a.scale_in[tid];
                    a.scale_out[K + tid] = a.scale_in[K + tid];
                    a.scale_out[2 * K + tid] = ck;
                }
            }
        }
    }
    if (m_on) {
        gm = a.gamma_in[(size_t)d * K + km];
        am = a.alpha[km];
    }

    // ---- the slice (lda.cpp:179-181): JW x KS coalesced loads per lane, all independent.
    // Word ids: one vector load (lane i -> slot i of this wave), handed out with v_readlane.
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
    // counts of the words whose phinorm this lane holds after the fold of group g
    double cntv[NH];
    int jv[NH];
#pragma unroll
    for (int g = 0; g < NH; ++g) {
        const int i = 16 * g + fold16_index(lane);
        jv[g] = i * W + wid;
        cntv[g] = 0.0;
        if (i < JW && jv[g] < n_reg)
            cntv[g] = (double)cnts[jv[g]];
        else
            jv[g] = -1;
    }
    // Up to 256 topics the waves KS .. 2 KS - 1 have nothing to do while waves 0 .. KS - 1
    // evaluate exp(psi): they mirror gamma of the topic tid - KP (recomputed from the same
    // partial sums: bitwise the owner's value) and form sum |gamma - last| (lda.cpp:202) there,
    // off the exp(psi) waves' instruction streams.
    // (MIRROR, gm / am: loaded above, ahead of the rows)
    // gamma / alpha / exp(psi(gamma)) of topic tid                       lda.cpp:174
    double ek = 0.0;
    if (tid < KP) {
        const double e0 = exp_digamma<cfg::SC, cfg::XF>(gk);
        ek = k_on ? e0 : 0.0;
        ebuf[tid] = ek;                              // zero beyond K
    }
    if constexpr (FACTORS && KS <= 2) {
        if (a.partial && !a.scale_in)                // `part` is idle until the first product
            topic_scale_partials<kWideThreads>(K, a.G, pv, part);
    }
    // rows of the first tail words, zero beyond K
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
    if constexpr (FACTORS) {
        if (a.partial) {                             // launch-uniform
            // e is kept as c_k exp(psi(gamma_k)) throughout: phinorm and gamma come out the same
            // with u = exp(psi(lambda)) in place of exp E[log beta] (estep_kernels.h 2b)
            if (k_on) {
                if constexpr (KS <= 2) {
                    if (!a.scale_in)
                        ck = topic_scale_combine(K, tid, part, a.scale_out);
                    else if (a.scale_wait)           // merged launch (estep_merged.h)
                        ck = scale_wait_load(a, K, tid);
                }
                ek *= ck;
                ebuf[tid] = ek;
                if constexpr (KS > 2)
                    cbuf[tid] = ck;
            }
            __syncthreads();                         // `part` is free again, e complete
        }
    }
    TRLDA_STAMP(0);

    double e[KS];
    double acc[NSET][KS];
    // Tail words n_reg + t0 + u, u < TCH: those at or beyond `end` (n_lds for LDS rows, the
    // document's tail length for streamed rows) contribute zero.
    auto tail_chunk = [&](int t0, bool from_lds) {
        double r[TCH][KS];
        const int my_u = fold_group_index<TCH>(lane);
        const int end = from_lds ? n_lds : n - n_reg;
        // (the count comes from LDS for the words whose rows are there: a global load per chunk
        // and iteration would sit in front of the weight)
        double my_cnt = 0.0;
        if (t0 + my_u < end)
            my_cnt = from_lds ? cnt_tail[t0 + my_u] : (double)cnts[n_reg + t0 + my_u];
        if (from_lds) {
#pragma unroll
            for (int u = 0; u < TCH; ++u) {
                const double *rowp = rows + (size_t)min(t0 + u, n_lds - 1) * KP + lane;
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    r[u][s] = rowp[64 * s];
            }
        } else {
#pragma unroll
            for (int u = 0; u < TCH; ++u) {
                const double *rowp = a.eeb + (size_t)ids[min(n_reg + t0 + u, n - 1)] * K;
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    r[u][s] = rowp[min(lane + 64 * s, K - 1)];
            }
#pragma unroll
            for (int u = 0; u < TCH; ++u)
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    r[u][s] = kv[s] ? r[u][s] : 0.0;
        }
        double sv[TCH];
#pragma unroll
        for (int u = 0; u < TCH; ++u) {
            double d0 = 0.0, d1 = 0.0;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s & 1)
                    d1 = fma(e[s], r[u][s], d1);
                else
                    d0 = fma(e[s], r[u][s], d0);
            }
            sv[u] = d0 + d1;
        }
        const double tw = my_cnt * rcp_pos<true>(fold_group<TCH>(sv) + 1e-100);
        // one lane per word (a word's result sits in 64 / TCH lanes: the first of them stores)
        if (lane == fold_group_lane<TCH>(my_u) && t0 + my_u < end)
            a.tw_csr[p0 + n_reg + t0 + my_u] = tw;
#pragma unroll
        for (int u = 0; u < TCH; ++u) {
            const double twu = readlane_f64(tw, fold_group_lane<TCH>(u));
#pragma unroll
            for (int s = 0; s < KS; ++s)
                acc[u % NSET][s] = fma(twu, r[u][s], acc[u % NSET][s]);
        }
    };

    double twv[NH];                                  // cnt / phinorm of word jv[g]
    int it = 0;
    int cur = 0;                                     // ebuf / misc buffer holding the current e
    // sum_k |gamma_k - last_k| against threshold * K: the mean's division would sit in every
    // wave's instruction stream (lda.cpp:202)
    double change_sum = 0.0;
    const double thresholdK = a.threshold * (double)K;
    for (;;) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
            e[s] = ebuf[cur * KP + lane + 64 * s];
        if (it > 0) {                                // mean |gamma - last|        lda.cpp:202
            double sum = 0.0;
#pragma unroll
            for (int w = 0; w < (KS < W ? KS : W); ++w)
                sum += misc[cur * 8 + w];
            change_sum = sum;
        }
#pragma unroll
        for (int u = 0; u < NSET; ++u)
#pragma unroll
            for (int s = 0; s < KS; ++s)
                acc[u][s] = 0.0;

        // ---- phinorm and cnt / phinorm of the register words           lda.cpp:183 / :199
#pragma unroll
        for (int g = 0; g < NH; ++g) {
            twv[g] = 0.0;
            if (16 * g < JE) {                       // wave-uniform: the group has words
                double sv[16];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        sv[4 * c + u] = 0.0;
                    const int i0 = 16 * g + 4 * c;
                    if (i0 < JW && i0 < JE) {
#pragma unroll
                        for (int s = 0; s < KS; ++s)
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (i0 + u < JW)
                                    sv[4 * c + u] = fma(e[s], beta[i0 + u < JW ? i0 + u : 0][s],
                                                        sv[4 * c + u]);
                    }
                }
                const double tot = (16 * g + 16 <= JW) ? fold16<16>(sv)
                                                       : fold16<((JW & 15) ? (JW & 15) : 16)>(sv);
                twv[g] = cntv[g] * rcp_pos<true>(tot + 1e-100);
            }
        }
        TRLDA_STAMP(1);
        // ---- tail words: phinorm, weight and update of acc from one read of the row
        // (dealt from wave 0 up: of the two waves that share a SIMD the older one runs at its
        // lone-wave speed and waits for the younger at the barrier -- the slack a document a few
        // words over the registers spends here; dealing from wave 7 down was measured: 63.8
        // against 60.6 us per step with one 193-word document in the batch)
        for (int t0 = wid * TCH; t0 < n_lds; t0 += W * TCH)
            tail_chunk(t0, true);
        for (int t0 = n_lds + wid * TCH; n_reg + t0 < n; t0 += W * TCH)
            tail_chunk(t0, false);
        TRLDA_STAMP(2);
        if (it >= a.max_iter || (it > 0 && change_sum < thresholdK))      // lda.cpp:185, :202-203
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
                            acc[i % NSET][s] = fma(twi, beta[i < JW ? i : 0][s], acc[i % NSET][s]);
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
        TRLDA_STAMP(3);
        __syncthreads();
        TRLDA_STAMP(4);

        // ---- gamma_k = alpha_k + e_k acc_k ; e_k = exp(psi(gamma_k))     lda.cpp:194-197
        const int nxt = cur ^ 1;
        if (wid < KS) {                              // tid < KP
            const double accs = sum8_strided<KP>(part + tid);
            const double gnew = k_on ? fma(accs, ek, ak) : 1.5;
            [[maybe_unused]] const double diff = k_on ? fabs(gk - gnew) : 0.0;
            gk = gnew;
            double enew = exp_digamma<cfg::SC, cfg::XF>(gnew);
            if constexpr (FACTORS)
                enew *= KS > 2 ? cbuf[tid] : ck;
            ek = k_on ? enew : 0.0;
            ebuf[nxt * KP + tid] = ek;
            if constexpr (!MIRROR) {
                const double dsum = wave_sum_dpp(diff);
                if (lane == 0)
                    misc[nxt * 8 + wid] = dsum;
            }
        } else if (MIRROR && wid < 2 * KS) {
            const int kc = m_on ? km : 0;
            const double gnew = fma(sum8_strided<KP>(part + kc), ebuf[cur * KP + kc], am);
            const double dsum = wave_sum_dpp(m_on ? fabs(gm - gnew) : 0.0);
            gm = gnew;
            if (lane == 0)
                misc[nxt * 8 + (wid - KS)] = dsum;
        }
        TRLDA_STAMP(5);
        __syncthreads();
        TRLDA_STAMP(6);
        cur = nxt;
        ++it;
    }

    // ---- results
    if (k_on) {
        a.gamma[(size_t)d * K + tid] = gk;
        merged_store(a.epg + (size_t)d * K + tid, ek, a.done_counter != nullptr);
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
        if ((lane & 3) == 0) {                       // (a fold's result sits in all four lanes of a quad)
#pragma unroll
            for (int g = 0; g < NH; ++g)
                if (jv[g] >= 0)
                    merged_store(a.tw_word + (a.wrank ? a.wrank[p0 + jv[g]] : p0 + jv[g]), twv[g],
                                 a.done_counter != nullptr);
        }
        __syncthreads();                             // tw_csr of the tail words
        for (int j = n_reg + tid; j < n; j += kWideThreads)
            merged_store(a.tw_word + (a.wrank ? a.wrank[p0 + j] : p0 + j), a.tw_csr[p0 + j],
                         a.done_counter != nullptr);
    }
    TRLDA_STAMP(7);
    TRLDA_STAMP_FLUSH;
}

// FACTORS: exp(psi(lambda)) in place of exp E[log beta] (left behind by the M-step of the previous
// trust-region iteration, sstats_update2_kernel<.., EMIT>), the topic factors from a.scale_in
// (stream_kernels.h, topic_factors_kernel)
template <int KS, bool FACTORS = false>
__global__ __launch_bounds__(kWideThreads) void estep_docs_wide_kernel(DocKernelArgs a, int lds_rows)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int d = a.order[doc_block(a)];
    const int p0 = a.indptr[d];
    estep_docs_wide_body<KS, FACTORS>(a, lds_rows, lds, d, p0, a.indptr[d + 1] - p0);
}

// ---------------------------------------------------------------------------
// K <= 128: ONE launch for a batch whose documents differ in length -- every workgroup takes the
// variant its own document needs (block-uniform branch; all variants want the same 512 threads
// and ~250 registers):
//   up to 128 words   both orientations in registers               (estep_docs_reg_body<0>)
//   up to 144         the same with 18 words per wave              (<1>)
//   up to 192         one orientation, all words in registers
//                     (estep_docs_wide_body, with the fused preamble's topic factors)
//   up to 1024        SPLIT over ceil(n / 128) workgroups that exchange K sums per iteration
//                     (estep_docs_reg_body<0, true>): a 600-word document takes ~50 us on five
//                     CUs instead of 270 on one
//   beyond            one orientation, words past the registers in LDS / streamed from L2
// (Rounds 1-2 had a third register variant for 145..192 words, the words past 128 as LDS rows read
// in both orientations: measured equal to the single-orientation body at 145..160 words and
// slower at 176..192 -- 60.5 against 57.2 us per step, profiles/r03_sweep_tier2_*.txt -- and
// removed.)
// Documents are ordered by decreasing length, so the long ones start first; the launch lasts as
// long as its slowest document.  Workgroups past pre.n_docs prepare the next batch's preamble as
// in estep_docs_reg_kernel.  (A launch per variant, one behind the other, made a batch with one
// 193-word document 2.5 times slower than without it: profiles/r03_length_sweep_before.txt.)
// ---------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(kRegThreads) void estep_docs_tiered_kernel(DocKernelArgs a, PreArgs pre,
                                                                         int lds_rows)
{
    static_assert(kRegThreads == kWideThreads, "one workgroup shape for every tier");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if ((int)blockIdx.x >= pre.n_docs) {             // block-uniform
        docs_launch_preamble(pre, lds, (int)blockIdx.x);
        return;
    }
    if (a.docs_per_wg == 8 && (int)blockIdx.x >= a.small_block0) {   // block-uniform: K <= 32, eight short documents
        estep_docs_small_body(a, lds);
        return;
    }
    // (document, length, CSR offset, 0) [, (segment, segments, exchange row, document length)]
    const int n = a.pad_meta[4 * (size_t)blockIdx.x * a.meta_i4 + 1];
    if (a.meta_i4 == 2 && a.pad_meta[4 * ((size_t)blockIdx.x * 2 + 1) + 1] > 1) {
        estep_docs_reg_body<0, true>(a, lds);        // one segment of a document split over CUs
    } else if (n <= 128) {
        estep_docs_reg_body<0>(a, lds);
    } else if (n <= 144) {
        estep_docs_reg_body<1>(a, lds);
    } else {
        const int4 meta = reinterpret_cast<const int4 *>(a.pad_meta)[(size_t)blockIdx.x * a.meta_i4];
        estep_docs_wide_body<KS, true>(a, lds_rows, lds, meta.x, meta.z, meta.y);
    }
}

}  // namespace trlda
