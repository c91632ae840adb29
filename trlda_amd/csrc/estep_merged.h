// estep_merged.h -- ONE launch per E-step on small tables: the statistics stage (src/lda.cpp:207-217),
// the M-step (src/onlinelda.cpp:99-100, src/batchlda.cpp:60) and the row sums the next E-step needs
// (src/lda.cpp:172) as WORKGROUPS OF A DOCUMENT LAUNCH.  Two forms:
//
// MERGED (update calls; round 4, the stage rewritten in round 5): the statistics of THIS launch's
// documents.  A kernel of this path costs ~4.5 us before it does anything (the durations of a
// stream's dependent kernels are back to back, profiles/r04_timeline_*.txt), and a trust-region
// iteration had two of them.  Here the stage is `n_short + n_long` extra workgroups of the document
// launch.  They
//   * start as soon as a CU has room and use the wait for everything that does not depend on the
//     documents: their words' descriptors, the documents of their entries (static per batch),
//     exp(psi(lambda)) and lambda' of their words, the zero columns of the words outside the batch;
//   * wait until every document workgroup has counted itself done (`docs_done`, a counter that only
//     grows).  A document workgroup stores exp(psi(gamma)) and the weights cnt / phinorm with
//     agent-scope (write-through) stores, waits for their acknowledgement (stores_acknowledged),
//     counts; the last one writes the launch's number into one FLAG PER WAITER (thousands of waves
//     polling one address cost 20 us per launch);
//   * then have the weights and ALL rows of their entries in flight together, one round whatever
//     the lists' lengths (merged_stats_slots: 16 slots per wave, lists in classes by length);
//   * leave their row of the new lambda's row sums in `o.partial`; the NEXT launch's `n_comb`
//     workgroups add the rows up and finish the topic factors c_k = exp(-psi(row sum)) while the
//     documents stage their slices.
// Order in the grid = order of dispatch: topic factors, documents, (next batch's preamble,)
// statistics -- nothing waits for a workgroup behind it, so the launch makes progress whatever part
// of it is resident.  The host takes the path for launches whose documents leave CUs free, because
// only then do the helpers run UNDER the documents: a matter of speed, not of safety.
//
// DEFERRED (a stream of E-steps on an unchanged lambda; round 5): the statistics of the PREVIOUS
// call's documents and the preamble of the NEXT call's batch, as items that helper workgroups take
// from a counter (deferred_helper) -- nobody waits for anybody, a step is one launch of the
// documents' length.
//
// Arithmetic, both forms: a word's entries are added in document order by ONE wave; a word of more
// than 16 entries is cut into the same sixteen chunks as the 1024-thread kernel cuts it and
// combined in the same order: bitwise the statistics of the stand-alone kernel
// (tests/test_gpu_merged.py, tests/test_gpu_deferred.py).
#pragma once
#include "estep_kernels.h"
#include "estep_wide.h"
#include "rng_kernels.h"
#include "stream_kernels.h"

namespace trlda {

constexpr int kMergedMaxDocWgs = 224;   // document workgroups of a merged launch: all resident at once
constexpr int kMergedComb = 4;          // workgroups that finish the topic factors
constexpr int kMergedNW = 4;            // words a wave of the statistics stage works on at a time
constexpr int kMergedFlagStride = 16;   // unsigned ints between two waiters' flags (64 bytes)
constexpr int kMergedMaxHelpers = 512;  // flags of the statistics workgroups

struct MergedArgs {
    int first;                    // the statistics workgroups start here, counted from the first document
                                  // workgroup (documents + next-batch preamble before)
    int n_comb, n_short, n_long;  // topic factors (the FIRST workgroups of the grid) | words of <= 16
                                  // entries | longer lists
    // topic factors of THIS E-step from the block rows the previous M-step left (n_comb > 0)
    const double *c_rows;         // c_n x K
    int c_n;
    const double *c_base;         // K or nullptr: the share of the words outside the batch
    double *c_out;                // 3 K: psi(row sum), row sum, exp(-psi)
    unsigned int *c_ready;        // += 1 per combine workgroup
    unsigned int c_target;
    // statistics
    int K, V, N_short, N_long;
    const int4 *desc;             // N_short + N_long x (word, first entry, entries, 0), by decreasing length
    const int32_t *wdoc;          // document of each word-major entry
    const double *tw_word;        // cnt / phinorm in word-major order (written by this launch's documents)
    const double *epg;            // exp(psi(gamma)) rows (ditto); row -1 is zero
    const double *eeb;            // exp(psi(lambda)) of the batch's words (may be o.u_out)
    UpdateOut o;                  // o.partial: n_short + n_long rows
    const uint8_t *active_flag;   // V bytes, or nullptr: no zero columns to write
    unsigned int *docs_done;      // += 1 per document workgroup
    unsigned int docs_target;
    int *xerr;                    // set when a wait gave up (never in a sane run)
    // Nobody polls the counters: thousands of waves asking for ONE address every few hundred
    // cycles kept its memory channel -- and the fabric towards it -- so busy that the documents'
    // own loads took twice as long (60 us per launch instead of 37, round 4).  Every waiter has a
    // flag of its own, 64 bytes apart; whoever brings a counter to its target writes the launch's
    // number (`epoch`, it only grows) into the flags of those who wait for it.
    unsigned int *go_flags;       // one per statistics workgroup (kMergedFlagStride apart)
    unsigned int *c_flags;        // one per document workgroup
    unsigned int epoch;
    int n_docs;                   // document workgroups of the launch (flags to set)
    // deferred launch: the short / long lists by length class, longest first (the descriptors are
    // sorted by decreasing length): entries 9..16 | 5..8 | 3..4 | 1..2 of a short list, chunks (a
    // sixteenth of a long list, rounded up) of 9..16 | 5..8 | 3..4 | 1..2 entries
    int cls_short[4], cls_long[4];
    int slots;                    // merged launch: the statistics stage in its 16-slot form (merged_stats_slots)
    unsigned int *work_counter;   // deferred launch: the helpers' item counter (only grows) ...
    unsigned int work_base;       // ... and its value at the start of the launch
    int first_static;             // H (every helper starts with item h: all of them are resident from the
                                  // start) or 0 (every first item comes from the counter)
    unsigned long long *tstamps;  // diagnostics (TRLDA_MERGED_STAMPS=1, tools/merged_stamps.py) or nullptr:
                                  // s_memrealtime (100 MHz, one clock for the chip) of [start, flag seen,
                                  // end] per statistics workgroup, then [start, end of the document,
                                  // counted] per document workgroup from 3 * 512 on
};

// AUXILIARY workgroups of a merged launch (round 6): work of the CALL that does not depend on this
// launch's documents, on the CUs the documents leave free, in front of the statistics workgroups in
// the grid (which only wait there): the NEXT fresh gamma0 (rng_kernels.h, aux_draw_workgroup).
// Nothing in the launch waits for them; what they write is read by later kernels of the stream.
// ... and, in an update call without trust-region loop (onlinelda.cpp:103-109), the decay of the words
// OUTSIDE the mini-batch, lambda[:, w] = (1 - rho) lambda[:, w] + rho eta: inactive_update_stream_kernel's
// job (stream_kernels.h, ACT_KEEP), whose 1024-thread workgroups a 512-thread one goes through slot
// group by slot group -- the same per-thread sums, added up in the same slot order: bitwise the same
// lambda and the same block rows of its row sums.  (The launch's own preamble has read the old lambda
// in an earlier kernel; the documents read exp(psi(lambda)), the M-step the ACTIVE columns.)
struct AuxInactiveArgs {
    int n_vb;                     // the stream kernel's grid: items (0: none in this launch)
    int K, V, P, cpb;             // stream_geometry(K, V): vec = 2 (K even)
    double a, b;                  // lambda = a * lambda + b
    const uint8_t *active_flag;   // V bytes
    double *lambda;
    double *part_static;          // n_vb x K: block rows of the inactive words' row sums
};

struct AuxArgs {
    int n;                        // workgroups [first - n, first) of the grid, counted from the first document
    int n_items;                  // draw.n + inact.n_vb; workgroup v starts with item v, the counter hands
    unsigned int *work_counter;   // out the items from n on (it only grows: work_base at the launch's start)
    unsigned int work_base;
    AuxDrawArgs draw;             // items [0, draw.n)
    AuxInactiveArgs inact;        // items [draw.n, n_items)
};

// The kernel's arguments (~700 bytes: three structures) are fetched where they are first used, a
// scalar load and a wait at a time -- in the helper's path ten of them one behind the other, each a
// miss in the scalar cache: ~3 us before a helper's first item (profiles/r05_deferred_notes.txt:
// documents start 0.2 us into the launch, helpers 3.2).  One word of every 64-byte line of the
// argument segment, all requested at once, brings the lines into the scalar cache.
template <bool AUX = false>
__device__ __forceinline__ void warm_kernel_arguments()
{
    // (the deferred kernels' explicit arguments: 696 bytes in the code object's metadata, the tiered
    // one with its extra int; eleven lines = 704 bytes, none of them past the segment's last line;
    // AUX: the merged kernels' AuxArgs behind them -- four more lines)
    static_assert(sizeof(DocKernelArgs) + sizeof(PreArgs) + sizeof(MergedArgs) <= 11 * 64 &&
                      sizeof(DocKernelArgs) + sizeof(PreArgs) + sizeof(MergedArgs) > 10 * 64,
                  "eleven lines cover the argument segment and none lies beyond it");
    static_assert(sizeof(AuxArgs) > 4 * 64 - 8, "the four extra lines of the merged kernels lie inside AuxArgs");
    typedef __attribute__((address_space(4))) const unsigned int *karg_ptr;
    karg_ptr kp = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned int r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10;
    // (one block: left to the compiler the twelve loads came in three batches with a wait each)
    asm volatile("s_load_dword %0, %11, 0x0\n\t"
                 "s_load_dword %1, %11, 0x40\n\t"
                 "s_load_dword %2, %11, 0x80\n\t"
                 "s_load_dword %3, %11, 0xc0\n\t"
                 "s_load_dword %4, %11, 0x100\n\t"
                 "s_load_dword %5, %11, 0x140\n\t"
                 "s_load_dword %6, %11, 0x180\n\t"
                 "s_load_dword %7, %11, 0x1c0\n\t"
                 "s_load_dword %8, %11, 0x200\n\t"
                 "s_load_dword %9, %11, 0x240\n\t"
                 "s_load_dword %10, %11, 0x280\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(r0), "=&s"(r1), "=&s"(r2), "=&s"(r3), "=&s"(r4), "=&s"(r5), "=&s"(r6), "=&s"(r7),
                   "=&s"(r8), "=&s"(r9), "=&s"(r10)
                 : "s"(kp)
                 : "memory");
    if constexpr (AUX) {
        unsigned int a0, a1, a2, a3;
        asm volatile("s_load_dword %0, %4, 0x2c0\n\t"
                     "s_load_dword %1, %4, 0x300\n\t"
                     "s_load_dword %2, %4, 0x340\n\t"
                     "s_load_dword %3, %4, 0x380\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&s"(a0), "=&s"(a1), "=&s"(a2), "=&s"(a3)
                     : "s"(kp)
                     : "memory");
    }
}

// ---- the document side --------------------------------------------------------------------
// (estep_kernels.h / estep_wide.h: every output the statistics read goes out with merged_store)
__device__ __forceinline__ void docs_done_signal(const DocKernelArgs &a)
{
    if (!a.done_counter)                             // launch-uniform
        return;
    __shared__ int last_doc;
    stores_acknowledged();                           // this thread's epg / tw_word stores have left
    __syncthreads();
    if (threadIdx.x == 0)
        last_doc = __hip_atomic_fetch_add(a.done_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u ==
                   a.done_target;
    __syncthreads();
    if (last_doc)                                    // every document has stored and counted: go
        for (int i = threadIdx.x; i < a.n_go; i += kRegThreads)
            __hip_atomic_store(a.go_flags + (size_t)i * kMergedFlagStride, a.epoch, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
}

// ---- topic factors ----------------------------------------------------------------------------
// c_k = exp(-psi(base_k + sum_r rows[r][k])): workgroup vb of n_comb takes a contiguous range of
// topics; sixteen threads per topic add every sixteenth row (eight loads in flight), the parts are
// combined in order.  The same sums on every device: replicas stay bitwise equal.
__device__ __forceinline__ void merged_combine(const MergedArgs &mg, int vb, double *lds)
{
    const int tid = threadIdx.x, K = mg.K;
    const int kper = (K + mg.n_comb - 1) / mg.n_comb;            // <= 32 (K <= 128, four workgroups)
    const int k0 = vb * kper, kn = max(0, min(kper, K - k0));
    const int kk = tid & 31, part = tid >> 5;                    // 32 topics x 16 parts
    double acc = 0.0;
    if (kk < kn) {
        const double *col = mg.c_rows + k0 + kk;
        for (int r = part; r < mg.c_n; r += 16 * 8) {
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                v[q] = col[(size_t)min(r + 16 * q, mg.c_n - 1) * K];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                acc += (r + 16 * q < mg.c_n) ? v[q] : 0.0;
        }
    }
    lds[part * 32 + kk] = acc;
    __syncthreads();
    if (tid < kn) {
        double rs = mg.c_base ? mg.c_base[k0 + tid] : 0.0;
#pragma unroll
        for (int p = 0; p < 16; ++p)
            rs += lds[p * 32 + tid];
        const double ps = digamma(rs);
        const int k = k0 + tid;
        __hip_atomic_store(mg.c_out + k, ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mg.c_out + K + k, rs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mg.c_out + 2 * K + k, exp(-ps), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __shared__ int last_comb;
    stores_acknowledged();                           // c_out has left before the count
    __syncthreads();
    if (tid == 0)
        last_comb = __hip_atomic_fetch_add(mg.c_ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u ==
                    mg.c_target;
    __syncthreads();
    if (last_comb)                                   // all of c_out is in memory: tell the documents
        for (int i = tid; i < mg.n_docs; i += kRegThreads)
            __hip_atomic_store(mg.c_flags + (size_t)i * kMergedFlagStride, mg.epoch, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
}

// ---- statistics ---------------------------------------------------------------------------------
// every document workgroup of this launch has stored its outputs
// Every wave watches its workgroup's flag (the last document to finish sets it) on its own: eight
// waves per address, at a rate that costs the memory system nothing.  `seen` is the flag as the wave
// read it FIRST THING -- a workgroup that is dispatched when the documents have already finished
// finds it set and goes straight on, its weights in flight together with its static loads.  (A flag
// that never comes -- it cannot: the documents are resident and wait for nothing that waits for
// them -- ends the wait after ~1 s and fails the next synchronising call.)
__device__ __forceinline__ unsigned int merged_flag_load(const MergedArgs &mg, int vb)
{
    return __hip_atomic_load(mg.go_flags + (size_t)vb * kMergedFlagStride, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void merged_wait_docs(const MergedArgs &mg, int vb, unsigned int seen)
{
    int spins = 0;
    while ((int)(seen - mg.epoch) < 0) {
        __builtin_amdgcn_s_sleep(8);
        seen = merged_flag_load(mg, vb);
        if (++spins > (1 << 22)) {
            if (mg.xerr)
                __hip_atomic_store(mg.xerr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
    }
    flag_seen();                                     // (no load of the data above the flag's)
    if (mg.tstamps && threadIdx.x == 0)
        mg.tstamps[3 * vb + 1] = __builtin_amdgcn_s_memrealtime();
    // No cache invalidation here.  An agent-scope acquire (buffer_inv sc1) per wave is two thousand
    // invalidations of the XCDs' L2s, one after the other.  Nor is one needed: the only data of
    // this launch that a waiter reads and another workgroup of the launch has written are epg and
    // tw_word; they went out write-through (merged_store) before the flag; no line of them can
    // sit in this CU's L1 or this XCD's L2 from BEFORE that -- the caches start a kernel empty of
    // them (the kernel boundary's acquire), a document workgroup only writes them (bytes it wrote
    // are valid, the others are fetched), and nothing of this stage reads them before this
    // point.  The weights are read with agent-scope loads anyway; tests/test_gpu_merged.py
    // alternates two batches through the same buffers 600 times.
}

// NS list segments of at most 16 entries each, side by side: acc[j] += sum_u tw[q0_j + u] *
// epg[doc_u, 2 lane .. 2 lane + 1] in entry order, four rows per segment in flight per round.
// docs[j]: lane u holds the document of entry u (or -1, the zero row); `maxlen`: the longest of
// the segments (wave-uniform): rounds past it are skipped.
template <int NS, int R = 4, int MAXLEN = 16>
__device__ __forceinline__ void merged_segments(const int (&docs)[NS], const double (&tw)[NS], int maxlen,
                                                const double *__restrict__ epg, int K, int kk,
                                                double2 (&acc)[NS])
{
    int tlo[NS], thi[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        tlo[j] = __double2loint(tw[j]);
        thi[j] = __double2hiint(tw[j]);
    }
#pragma unroll
    for (int r = 0; r < MAXLEN; r += R) {
        if (r < maxlen) {                            // wave-uniform
            double2 ev[NS][R];
#pragma unroll
            for (int j = 0; j < NS; ++j)
#pragma unroll
                for (int u = 0; u < R; ++u) {
                    const long long row = (long long)__builtin_amdgcn_readlane(docs[j], r + u) * K;
                    ev[j][u] = *reinterpret_cast<const double2 *>(epg + row + kk);
                }
#pragma unroll
            for (int j = 0; j < NS; ++j)
#pragma unroll
                for (int u = 0; u < R; ++u) {
                    const double tu = __hiloint2double(__builtin_amdgcn_readlane(thi[j], r + u),
                                                       __builtin_amdgcn_readlane(tlo[j], r + u));
                    acc[j].x = fma(tu, ev[j][u].x, acc[j].x);   // (+0 * 0 past a segment's end)
                    acc[j].y = fma(tu, ev[j][u].y, acc[j].y);
                }
        }
    }
}

// the pair (i, i + 1) of one word: statistics, M-step, exp(psi(lambda)) where asked for;
// returns the two lambdas written (update_pair with the emission as a run-time switch)
__device__ __forceinline__ double2 merged_update_pair(const UpdateOut &o, size_t i, double2 s, double2 lp)
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
        if (o.u_out)                                 // launch-uniform; lam > 0 (the host knows)
            *reinterpret_cast<double2 *>(o.u_out + i) =
                make_double2(exp_digamma_positive(lam.x), exp_digamma_positive(lam.y));
    }
    return lam;
}

// DEFER: the stage works on the PREVIOUS E-step's outputs (a fixed-lambda stream of E-steps,
// trlda_model_set_deferred_stats): nothing of this launch is waited for, no flag is looked at, the
// weights are ordinary loads, and there is no M-step (o.lambda is null by construction).
template <bool DEFER = false>
__device__ __forceinline__ void merged_stats(const MergedArgs &mg, int vb, double *lds)
{
    constexpr int W = kRegThreads / kWave;           // 8 waves
    constexpr int NW = kMergedNW;
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int K = mg.K;
    const UpdateOut &o = mg.o;
    const int kk = min(2 * lane, K - 2);             // this lane's pair of topics (K even)
    const bool k_on = 2 * lane < K;
    const int n_stat = mg.n_short + mg.n_long;
    [[maybe_unused]] unsigned int seen = 0u;
    if constexpr (!DEFER)
        seen = merged_flag_load(mg, vb);             // (requested before everything else)

    // columns of the words outside the batch: zero (lda.cpp:169), written while the documents run
    if (mg.active_flag && o.sstats && (DEFER || !o.lambda)) {   // launch-uniform
        for (int w = vb * W + wid; w < mg.V; w += n_stat * W)
            if (!mg.active_flag[w] && k_on)
                *reinterpret_cast<double2 *>(o.sstats + (size_t)w * K + 2 * lane) = make_double2(0.0, 0.0);
    }

    if (vb < mg.n_short) {
        // ---- a wave per word, NW words at a time: words gw, gw + NWAVES, .. of the short list
        const int NWAVES = mg.n_short * W;
        const int gw = vb * W + wid;
        double2 rs = make_double2(0.0, 0.0);
        bool waited = false;
        for (int t0 = gw; t0 < mg.N_short; t0 += NW * NWAVES) {     // wave-uniform
            int wv[NW], len[NW], docs[NW], q0[NW];
            double2 e2[NW], lp[NW], acc[NW];
            int maxlen = 0;
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const int t = t0 + j * NWAVES;
                const int4 d = mg.desc[min(t, mg.N_short - 1)];
                wv[j] = __builtin_amdgcn_readfirstlane(d.x);
                q0[j] = __builtin_amdgcn_readfirstlane(d.y);
                len[j] = t < mg.N_short ? __builtin_amdgcn_readfirstlane(d.z) : 0;
                maxlen = max(maxlen, len[j]);
            }
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                docs[j] = lane < len[j] ? mg.wdoc[q0[j] + lane] : -1;       // -1: the zero row
                const size_t ic = (size_t)wv[j] * K + kk;
                e2[j] = *reinterpret_cast<const double2 *>(mg.eeb + ic);
                lp[j] = (!DEFER && o.lambda_prime) ? *reinterpret_cast<const double2 *>(o.lambda_prime + ic)
                                                   : make_double2(0.0, 0.0);
                acc[j] = make_double2(0.0, 0.0);
            }
            if constexpr (!DEFER) {
                if (!waited) {
                    merged_wait_docs(mg, vb, seen);
                    waited = true;
                }
            }
            double tw[NW];
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                if constexpr (DEFER)
                    tw[j] = lane < len[j] ? mg.tw_word[q0[j] + lane] : 0.0;
                else
                    tw[j] = lane < len[j] ? __hip_atomic_load(const_cast<double *>(mg.tw_word) + q0[j] + lane,
                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                          : 0.0;
            }
            merged_segments<NW>(docs, tw, maxlen, mg.epg, K, kk, acc);
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                if (len[j] > 0 && k_on) {            // (len: wave-uniform)
                    const double2 s = make_double2(acc[j].x * e2[j].x, acc[j].y * e2[j].y);
                    if constexpr (DEFER) {
                        *reinterpret_cast<double2 *>(o.sstats + (size_t)wv[j] * K + 2 * lane) = s;
                    } else {
                        const double2 lam = merged_update_pair(o, (size_t)wv[j] * K + 2 * lane, s, lp[j]);
                        rs.x += lam.x;
                        rs.y += lam.y;
                    }
                }
            }
        }
        if (!DEFER && o.partial) {                   // launch-uniform
            if (k_on)
                *reinterpret_cast<double2 *>(lds + wid * K + 2 * lane) = rs;
            __syncthreads();
            for (int k = tid; k < K; k += kRegThreads) {
                double sum = lds[k];
#pragma unroll
                for (int c = 1; c < W; ++c)
                    sum += lds[c * K + k];
                o.partial[(size_t)vb * K + k] = sum;
            }
        }
        return;
    }

    // ---- a workgroup per long list: the sixteen contiguous chunks of the 1024-thread kernel
    // (chunk = ceil(L / 16) <= 16 entries here), two per wave side by side, combined in chunk order
    const int lb = vb - mg.n_short;
    double rsl = 0.0;                                // thread k: row sum of what it writes
    bool waited = false;
    for (int t = lb; t < mg.N_long; t += mg.n_long) {
        const int4 d = mg.desc[mg.N_short + t];
        const int w = __builtin_amdgcn_readfirstlane(d.x), base = __builtin_amdgcn_readfirstlane(d.y);
        const int L = __builtin_amdgcn_readfirstlane(d.z);
        const int chunk = (L + 15) / 16;             // <= 16 (the host takes this path for L <= 256)
        int docs[2], clen[2], c0[2];
        double2 acc[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int v = wid + 8 * h;               // chunk index
            c0[h] = min(L, v * chunk);
            clen[h] = min(L, c0[h] + chunk) - c0[h];
            docs[h] = lane < clen[h] ? mg.wdoc[base + c0[h] + lane] : -1;
            acc[h] = make_double2(0.0, 0.0);
        }
        // exp(psi(lambda)) and lambda' of the element this thread finishes: requested now, while
        // the documents still run (behind the barrier below they would be one more memory latency)
        const size_t i = (size_t)w * K + min(tid, K - 1);
        const double ek = mg.eeb[i];
        const double lpk = (!DEFER && o.lambda_prime) ? o.lambda_prime[i] : 0.0;
        if constexpr (!DEFER) {
            if (!waited) {
                merged_wait_docs(mg, vb, seen);
                waited = true;
            }
        }
        double tw[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if constexpr (DEFER)
                tw[h] = lane < clen[h] ? mg.tw_word[base + c0[h] + lane] : 0.0;
            else
                tw[h] = lane < clen[h] ? __hip_atomic_load(const_cast<double *>(mg.tw_word) + base + c0[h] + lane,
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                       : 0.0;
        }
        merged_segments<2>(docs, tw, max(clen[0], clen[1]), mg.epg, K, kk, acc);
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (k_on)
                *reinterpret_cast<double2 *>(lds + (wid + 8 * h) * K + 2 * lane) = acc[h];
        __syncthreads();
        if (tid < K) {
            double cv[16];                           // (requested together, added in chunk order)
#pragma unroll
            for (int c = 0; c < 16; ++c)
                cv[c] = lds[c * K + tid];
            __builtin_amdgcn_sched_barrier(0);
            double sum = cv[0];
#pragma unroll
            for (int c = 1; c < 16; ++c)
                sum += cv[c];
            const double s = sum * ek;
            if (o.sstats)
                o.sstats[i] = s;
            if (!DEFER && o.lambda) {
                const double hat = o.eta + o.scale * s;
                const double lam = o.lambda_prime ? o.omr * lpk + o.rho * hat : o.rho * hat;
                o.lambda[i] = lam;
                if (o.u_out)
                    o.u_out[i] = exp_digamma_positive(lam);
                rsl += lam;
            }
        }
        __syncthreads();
    }
    if (!DEFER && o.partial && tid < K)
        o.partial[(size_t)vb * K + tid] = rsl;
}

__device__ __forceinline__ void merged_stats_slots(const MergedArgs &mg, int vb, double *lds);   // (below)

// the statistics workgroups: past the documents and the next batch's preamble
__device__ __forceinline__ void merged_helper(const MergedArgs &mg, double *lds, int vb)
{
    if (mg.tstamps && threadIdx.x == 0)
        mg.tstamps[3 * vb] = __builtin_amdgcn_s_memrealtime();
    if (mg.slots)                                    // launch-uniform
        merged_stats_slots(mg, vb, lds);
    else
        merged_stats(mg, vb, lds);
    if (mg.tstamps && threadIdx.x == 0)
        mg.tstamps[3 * vb + 2] = __builtin_amdgcn_s_memrealtime();
}

// diagnostics: a document workgroup's [start, end of its document, counted]
__device__ __forceinline__ void merged_doc_stamp(const MergedArgs &mg, int which)
{
    if (mg.tstamps && threadIdx.x == 0)
        mg.tstamps[3 * (512 + (int)blockIdx.x - mg.n_comb) + which] = __builtin_amdgcn_s_memrealtime();
}

// Grid of a merged launch: [0, n_comb) topic factors | documents | the next batch's preamble
// (pre.nb) | auxiliary work of the call (aux.n) | statistics (n_short + n_long).  Workgroups are dispatched in this order, and nothing
// waits for anything behind it: the topic-factor workgroups wait for nothing, a document only for
// them, a statistics workgroup only for the documents.  So a merged launch ends whatever part of it
// is resident at a time -- on a device with fewer CUs than document workgroups, or beside another
// stream's kernels, it is slower, never stuck (round 4 had the topic factors BEHIND the
// documents and relied on every document being resident at once).
// one block of inactive_update_stream_kernel<1024, 2, ACT_KEEP, false> by 512 threads
__device__ __forceinline__ void aux_inactive_block(const AuxInactiveArgs &x, int vb, double *scratch)
{
    constexpr int U = kStreamUnroll, T = kRegThreads;
    const int K = x.K, P = x.P, cpb = x.cpb;
    const int spp = T / P;                           // slots per pass (>= 8: P <= 64)
    const int sl = threadIdx.x / P, kp = threadIdx.x - sl * P;
    const int m = x.n_vb * cpb;
    const size_t off = (size_t)kp * 2;
    for (int s0 = 0; s0 < cpb; s0 += spp) {          // block-uniform
        const int slot = s0 + sl;
        if (sl < spp && slot < cpb) {
            double acc0 = 0.0, acc1 = 0.0;
            for (int col = vb * cpb + slot; col < x.V; col += U * m) {
                double2 v[U];
                bool fl[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    fl[u] = x.active_flag[min(col + u * m, x.V - 1)] != 0;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    v[u] = make_double2(0.0, 0.0);
                    if (!fl[u] && col + u * m < x.V)
                        v[u] = vload_nt<2>(x.lambda + (size_t)(col + u * m) * K + off);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int c = col + u * m;
                    if (c < x.V && !fl[u]) {
                        const double y0 = fma(x.a, v[u].x, x.b), y1 = fma(x.a, v[u].y, x.b);   // (as the stream kernel compiles)
                        acc0 += y0;
                        acc1 += y1;
                        *reinterpret_cast<double2 *>(x.lambda + (size_t)c * K + off) = make_double2(y0, y1);
                    }
                }
            }
            scratch[slot * K + kp * 2] = acc0;
            scratch[slot * K + kp * 2 + 1] = acc1;
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += T) {       // (the slots in slot order: stream_block_partial)
        double sum = scratch[k];
        for (int q = 1; q < cpb; ++q)
            sum += scratch[q * K + k];
        x.part_static[(size_t)vb * K + k] = sum;
    }
}

// the auxiliary workgroups: [first - aux.n, first)
__device__ __forceinline__ void merged_aux(const AuxArgs &aux, double *lds, int v)
{
    __shared__ unsigned int aux_next;
    int item = v;
    while (item < aux.n_items) {                     // block-uniform
        unsigned int fetched = 0u;
        if (threadIdx.x == 0)                        // (the next item: requested now, needed at the end)
            fetched = __hip_atomic_fetch_add(aux.work_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (item < aux.draw.n)
            aux_draw_workgroup<kRegThreads>(aux.draw, item, lds);
        else
            aux_inactive_block(aux.inact, item - aux.draw.n, lds);
        if (threadIdx.x == 0)
            aux_next = fetched - aux.work_base + (unsigned int)aux.n;
        __syncthreads();
        item = (int)aux_next;
        __syncthreads();
    }
}

template <int MODE>
__global__ __launch_bounds__(kRegThreads) void estep_docs_reg_merged_kernel(DocKernelArgs a, PreArgs pre,
                                                                             MergedArgs mg, AuxArgs aux)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    warm_kernel_arguments<true>();
    if ((int)blockIdx.x < mg.n_comb) {               // block-uniform
        merged_combine(mg, (int)blockIdx.x, lds);
        return;
    }
    const int rel = (int)blockIdx.x - mg.n_comb;     // == doc_block(a): a.block0 = mg.n_comb
    if (rel >= mg.first) {
        merged_helper(mg, lds, rel - mg.first);
        return;
    }
    if (rel >= mg.first - aux.n) {
        merged_aux(aux, lds, rel - (mg.first - aux.n));
        return;
    }
    if (rel >= pre.n_docs) {
        docs_launch_preamble(pre, lds, rel);
        return;
    }
    merged_doc_stamp(mg, 0);
    if (a.docs_per_wg == 8)                          // launch-uniform: a wave per document (K <= 32)
        estep_docs_small_body(a, lds);
    else
        estep_docs_reg_body<MODE>(a, lds);
    merged_doc_stamp(mg, 1);
    docs_done_signal(a);
    merged_doc_stamp(mg, 2);
}

template <int KS>
__global__ __launch_bounds__(kRegThreads) void estep_docs_tiered_merged_kernel(DocKernelArgs a, PreArgs pre,
                                                                                int lds_rows, MergedArgs mg,
                                                                                AuxArgs aux)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    warm_kernel_arguments<true>();
    if ((int)blockIdx.x < mg.n_comb) {               // block-uniform
        merged_combine(mg, (int)blockIdx.x, lds);
        return;
    }
    const int rel = (int)blockIdx.x - mg.n_comb;
    if (rel >= mg.first) {
        merged_helper(mg, lds, rel - mg.first);
        return;
    }
    if (rel >= mg.first - aux.n) {
        merged_aux(aux, lds, rel - (mg.first - aux.n));
        return;
    }
    if (rel >= pre.n_docs) {
        docs_launch_preamble(pre, lds, rel);
        return;
    }
    merged_doc_stamp(mg, 0);
    const int n = a.pad_meta[4 * (size_t)rel * a.meta_i4 + 1];
    if (a.docs_per_wg == 8 && rel >= a.small_block0) {   // block-uniform: K <= 32, eight short documents
        estep_docs_small_body(a, lds);
    } else if (a.meta_i4 == 2 && a.pad_meta[4 * ((size_t)rel * 2 + 1) + 1] > 1) {
        estep_docs_reg_body<0, true>(a, lds);        // one segment of a document split over CUs
    } else if (n <= 128) {
        estep_docs_reg_body<0>(a, lds);
    } else if (n <= 144) {
        estep_docs_reg_body<1>(a, lds);
    } else {
        const int4 meta = reinterpret_cast<const int4 *>(a.pad_meta)[(size_t)rel * a.meta_i4];
        estep_docs_wide_body<KS, true>(a, lds_rows, lds, meta.x, meta.z, meta.y);
    }
    merged_doc_stamp(mg, 1);
    docs_done_signal(a);
    merged_doc_stamp(mg, 2);
}

// ---- the statistics stage of a deferred launch ---------------------------------------------------
// merged_stats<true> does the job (and did, in the first form of the deferred launch: 34.3 us per
// launch where the documents alone take 30.9 -- profiles/r05_deferred_first.txt).  What differs
// here is what bounds it: nobody waits for a flag, so there is no idle time that hides anything;
// the helpers (this stage + the next batch's preamble) share the ~56 CUs the documents leave free,
// ONE 512-thread workgroup per CU at a time (the launch's registers are the documents'), and a
// workgroup lasts as long as its chain of dependent memory latencies -- ~1 us each.  The stamps of
// that first form (profiles/r05_deferred_stamps_first.txt) show workgroups of 4 to 11 us, the
// longest the ones whose words have the longest lists: merged_segments walks a list four rows at
// a time, a list of 16 entries is four dependent rounds of gathers.  Here every workgroup is ONE
// round: the lists are in classes by length (they are sorted by it), and a wave takes as many
// words of a class as keep 16 rows (16-byte gathers per lane) in flight --
//     entries   9..16 | 5..8 | 3..4 | 1..2          chunk of a long list  9..16 | 5..8 | 3..4 | 1..2
//     words / group 1 |   2  |   4  |   8          lists / workgroup          1  |   1  |   2  |   4
// (a wave takes TWO groups per item: deferred_short_wave)
// -- so a workgroup is: descriptors -> (documents, weights, exp(psi(lambda))) -> rows -> store.
// The zero columns of the words outside the batch: a wave reads 64 flags with ONE load and writes
// the zeros of those that are clear.  The sums and their order are unchanged (a word's entries in
// document order; long lists in the sixteen chunks of the 1024-thread kernel, combined in chunk
// order): bitwise the statistics of sstats_update2_kernel.
constexpr int kDeferShortNW[4] = {2, 4, 8, 16};  // words per wave, by class (two groups of 16 slots)
constexpr int kDeferLongLW[4] = {1, 1, 2, 4};    // lists per workgroup, by class
constexpr int kDeferLog2R[4] = {4, 3, 2, 1};     // log2 of the longest list / chunk of a class

// workgroups the stage takes for a batch with these class counts (host and device agree through it)
__host__ __device__ inline int deferred_short_items(const int (&c)[4])
{
    int n = 0;
    for (int i = 0; i < 4; ++i)
        n += (c[i] + 8 * kDeferShortNW[i] - 1) / (8 * kDeferShortNW[i]);
    return n;
}
__host__ __device__ inline int deferred_long_items(const int (&c)[4])
{
    int n = 0;
    for (int i = 0; i < 4; ++i)
        n += (c[i] + kDeferLongLW[i] - 1) / kDeferLongLW[i];
    return n;
}

// ONE code path for every class: a wave has 16 SLOTS, slot s = (segment s >> lr, entry s & (R - 1)),
// R = 1 << lr the class's longest list (wave-uniform at run time).  Lane l < 16 looks after slot l:
// it fetches its segment's descriptor, then the document and the weight of its entry (all
// addresses per lane: no arrays indexed at run time); the 16 rows are gathered through v_readlane
// with constant lane numbers and added up slot by slot -- a segment's entries in order -- into one
// running pair that is finished and reset at every segment boundary (a wave-uniform test).
// (Eight instantiations of the array form, one per class, spilled 60 to 90 vector registers of
// the whole launch.)
//
// short lists: the wave's words [t, t + 2 (16 >> lr)) below t_end, as TWO groups of 16 slots.  A group
// is three dependent memory latencies (descriptors -> documents and weights -> rows); the second
// group's first two are requested with the first group's, so that a wave pays four for two groups
// where two items paid six (round 5: the helpers' CU-time is a sixth of a two-lane step).
template <class Pub>
__device__ __forceinline__ void deferred_short_wave(const MergedArgs &mg, int t, int t_end, int lr, int lane,
                                                    int kk, bool k_on, const Pub &pub)
{
    const int K = mg.K;
    const int R = 1 << lr, nw = 16 >> lr;
    const int sl = min(lane, 15), j = sl >> lr, u = sl & (R - 1);
    const bool second = t + nw < t_end;              // wave-uniform
    // descriptors of both groups, then their documents and weights
    int4 d[2];
    bool w_on[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int tg = t + g * nw;
        w_on[g] = tg + j < t_end;
        d[g] = mg.desc[min(tg + j, t_end - 1)];      // (word, first entry, entries, 0)
    }
    int doc[2], tlo[2], thi[2], wword[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const bool on = lane < 16 && w_on[g] && u < d[g].z;
        const int q = on ? d[g].y + u : 0;
        doc[g] = on ? mg.wdoc[q] : -1;               // -1: the zero row
        const double tw = on ? mg.tw_word[q] : 0.0;
        tlo[g] = __double2loint(tw);
        thi[g] = __double2hiint(tw);
        wword[g] = w_on[g] ? d[g].x : -1;
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        if (g == 1 && !second)
            break;
        // exp(psi(lambda)) of the segments' words: requested with the rows
        double2 e2[8];                               // (R >= 2: segments start at even slots)
#pragma unroll
        for (int s0 = 0; s0 < 16; s0 += 2) {
            if ((s0 & (R - 1)) == 0) {               // wave-uniform: a segment starts here
                const int w = max(__builtin_amdgcn_readlane(wword[g], s0), 0);
                e2[s0 >> 1] = *reinterpret_cast<const double2 *>(mg.eeb + (size_t)w * K + kk);
            }
        }
        double2 ev[16];
#pragma unroll
        for (int s0 = 0; s0 < 16; ++s0) {
            const long long row = (long long)__builtin_amdgcn_readlane(doc[g], s0) * K;
            ev[s0] = *reinterpret_cast<const double2 *>(mg.epg + row + kk);
        }
        if (g == 0)
            pub();                                   // (the first group's loads requested, no store issued yet)
        double2 acc = make_double2(0.0, 0.0), ec = make_double2(0.0, 0.0);
#pragma unroll
        for (int s0 = 0; s0 < 16; ++s0) {
            if ((s0 & 1) == 0 && (s0 & (R - 1)) == 0)    // (the segment's exp(psi(lambda)): constant index --
                ec = e2[s0 >> 1];                        // an index computed at run time would put e2 in scratch)
            const double tu = __hiloint2double(__builtin_amdgcn_readlane(thi[g], s0),
                                               __builtin_amdgcn_readlane(tlo[g], s0));
            acc.x = fma(tu, ev[s0].x, acc.x);            // (+0 * 0 past a list's end)
            acc.y = fma(tu, ev[s0].y, acc.y);
            if (((s0 + 1) & (R - 1)) == 0) {             // wave-uniform: the segment ends here
                const int w = __builtin_amdgcn_readlane(wword[g], s0);
                if (w >= 0 && k_on)
                    *reinterpret_cast<double2 *>(mg.o.sstats + (size_t)w * K + 2 * lane) =
                        make_double2(acc.x * ec.x, acc.y * ec.y);
                acc = make_double2(0.0, 0.0);
            }
        }
    }
}

// long lists: LW = 16 >> (lr + 1) lists [t, t + LW) (below t_end; indices into the long
// descriptors) per workgroup and pass; a list is cut into the sixteen chunks of the 1024-thread
// kernel (chunk = ceil(L / 16) <= R entries), wave `wid` walks chunks wid and wid + 8 of each list:
// segment g = (list g >> 1, half g & 1).  The longest class (chunks of 9..16 entries) takes the
// two halves in two passes.  Thread (i, k) = (tid / TPW, tid % TPW) finishes topic k of list i.
template <class Pub>
__device__ __forceinline__ void deferred_long_group(const MergedArgs &mg, int t, int t_end, int lr, double *lds,
                                                    int tid, int lane, int wid, int kk, bool k_on, const Pub &pub)
{
    const int K = mg.K;
    const int R = 1 << lr;
    const bool two_pass = lr == 4;                   // one list, its two chunks one after the other
    const int LW = two_pass ? 1 : 16 >> (lr + 1);
    const int TPW = kRegThreads / LW;                // >= 128 >= K
    const int fi = tid / TPW, fk = tid % TPW;
    // the epilogue's operands: requested first
    const int4 df = mg.desc[mg.N_short + min(t + fi, t_end - 1)];
    const bool fin_on = t + fi < t_end && fk < K;
    const size_t fidx = (size_t)df.x * K + min(fk, K - 1);
    const double ek = mg.eeb[fidx];
    for (int pass = 0; pass < (two_pass ? 2 : 1); ++pass) {          // block-uniform
        const int sl = min(lane, 15), g = two_pass ? pass : sl >> lr, u = sl & (R - 1);
        const int i = g >> 1, h = g & 1;
        const bool w_on = t + i < t_end;
        const int4 d = mg.desc[mg.N_short + min(t + i, t_end - 1)];
        const int L = w_on ? d.z : 0;
        const int chunk = (L + 15) / 16;             // <= R
        const int c0 = min(L, (wid + 8 * h) * chunk);
        const int cl = min(L, c0 + chunk) - c0;
        const bool on = lane < 16 && u < cl;
        const int q = on ? d.y + c0 + u : 0;
        const int doc = on ? mg.wdoc[q] : -1;
        const double tw = on ? mg.tw_word[q] : 0.0;
        const int tlo = __double2loint(tw), thi = __double2hiint(tw);
        double2 ev[16];
#pragma unroll
        for (int s0 = 0; s0 < 16; ++s0) {
            const long long row = (long long)__builtin_amdgcn_readlane(doc, s0) * K;
            ev[s0] = *reinterpret_cast<const double2 *>(mg.epg + row + kk);
        }
        double2 acc = make_double2(0.0, 0.0);
#pragma unroll
        for (int s0 = 0; s0 < 16; ++s0) {
            const double tu = __hiloint2double(__builtin_amdgcn_readlane(thi, s0),
                                               __builtin_amdgcn_readlane(tlo, s0));
            acc.x = fma(tu, ev[s0].x, acc.x);
            acc.y = fma(tu, ev[s0].y, acc.y);
            if (((s0 + 1) & (R - 1)) == 0) {         // wave-uniform: the segment ends here
                const int gs = two_pass ? pass : s0 >> lr;           // (list gs >> 1, half gs & 1)
                if (k_on)
                    *reinterpret_cast<double2 *>(lds + (size_t)((gs >> 1) * 16 + wid + 8 * (gs & 1)) * K + 2 * lane) = acc;
                acc = make_double2(0.0, 0.0);
            }
        }
    }
    pub();                                           // (the rows are in, the only stores come below)
    __syncthreads();
    if (fin_on) {
        const double *col = lds + (size_t)(fi * 16) * K + fk;
        // (the sixteen chunk sums requested together, added in chunk order: left to the compiler they
        // were fifteen LDS round trips one behind the other)
        double cv[16];
#pragma unroll
        for (int c = 0; c < 16; ++c)
            cv[c] = col[(size_t)c * K];
        __builtin_amdgcn_sched_barrier(0);
        double sum = cv[0];
#pragma unroll
        for (int c = 1; c < 16; ++c)
            sum += cv[c];
        mg.o.sstats[fidx] = sum * ek;
    }
    __syncthreads();
}

template <class Pub>
__device__ __forceinline__ void deferred_stats(const MergedArgs &mg, int vb, double *lds, const Pub &pub,
                                               int tid_bias)
{
    constexpr int W = kRegThreads / kWave;           // 8 waves
    const int tid = (int)threadIdx.x + tid_bias;     // (an opaque zero: deferred_helper)
    const int lane = tid & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int K = mg.K;
    const int kk = min(2 * lane, K - 2);             // this lane's pair of topics (K even)
    const bool k_on = 2 * lane < K;

    // columns of the words outside the batch: zero (lda.cpp:169) -- by the short-list workgroups
    // (the long-list ones end with two barriers), or by everybody when there are none
    const int zero_wgs = mg.n_short > 0 ? mg.n_short : mg.n_long;
    // (the first block's flags are requested NOW and looked at after the lists: behind them the
    // load would be one more latency at the end of the workgroup)
    const int zw0 = (vb * W + wid) * kWave;
    const bool zero_mine = vb < zero_wgs && zw0 < mg.V;
    const unsigned char flag0 = (zero_mine && zw0 + lane < mg.V) ? mg.active_flag[zw0 + lane] : (unsigned char)1;
    auto zero_columns = [&]() {
        if (!zero_mine)
            return;
        for (int w0 = zw0; w0 < mg.V; w0 += zero_wgs * W * kWave) {
            const int w = w0 + lane;
            const bool clear = w0 == zw0 ? flag0 == 0 : (w < mg.V && mg.active_flag[w] == 0);
            unsigned long long todo = __ballot(clear);
            while (todo) {                           // wave-uniform
                const int j = __builtin_ctzll(todo);
                todo &= todo - 1ull;
                if (k_on)
                    *reinterpret_cast<double2 *>(mg.o.sstats + (size_t)(w0 + j) * K + 2 * lane) =
                        make_double2(0.0, 0.0);
            }
        }
    };

    if (vb < mg.n_short) {
        // which class, which item of it: this wave's words [t, t + NW)
        int cls = 0, it = vb, base = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int items = (mg.cls_short[c] + W * kDeferShortNW[c] - 1) / (W * kDeferShortNW[c]);
            if (cls == c && it >= items) {
                it -= items;
                base += mg.cls_short[c];
                cls = c + 1;
            }
        }
        const int lr = 4 - cls, nw = 2 * (16 >> lr); // kDeferLog2R / kDeferShortNW
        const int t_end = base + (cls == 0 ? mg.cls_short[0] : cls == 1 ? mg.cls_short[1]
                                  : cls == 2 ? mg.cls_short[2] : mg.cls_short[3]);
        const int t = base + (it * W + wid) * nw;
        if (t < t_end)                               // wave-uniform (wave 0 always has words)
            deferred_short_wave(mg, t, t_end, lr, lane, kk, k_on, pub);
        zero_columns();
        return;
    }

    if (mg.n_short == 0)
        zero_columns();
    int cls = 0, it = vb - mg.n_short, base = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int items = (mg.cls_long[c] + kDeferLongLW[c] - 1) / kDeferLongLW[c];
        if (cls == c && it >= items) {
            it -= items;
            base += mg.cls_long[c];
            cls = c + 1;
        }
    }
    const int lw = cls == 0 ? 1 : 16 >> (5 - cls);   // kDeferLongLW: 1, 1, 2, 4
    const int t_end = base + (cls == 0 ? mg.cls_long[0] : cls == 1 ? mg.cls_long[1]
                              : cls == 2 ? mg.cls_long[2] : mg.cls_long[3]);
    deferred_long_group(mg, base + it * lw, t_end, 4 - cls, lds, tid, lane, wid, kk, k_on, pub);
}

// ---- deferred statistics: a stream of E-steps on an unchanged lambda -----------------------------
// (trlda_model_set_deferred_stats; reference: consecutive LDA::updateVariablesVI calls of a corpus
// pass, src/lda.cpp:160-220 -- the statistics of one call, :207-217, do not feed the next call.)
// The statistics of E-step i are workgroups of E-step i + 1's document launch: grid = documents of
// this step | preamble of the next step's batch (pre.nb) | statistics of the PREVIOUS step
// (n_short + n_long workgroups, merged_stats<true>).  The previous step's launch has ended -- its
// exp(psi(gamma)) rows and weights are ordinary memory -- so nobody waits for anybody; the helpers
// run on the CUs the documents leave free (56 of 256 at 200 documents) and a step is ONE launch of
// the documents' length.  Same sums in the same order as the kernel of its own
// (sstats_update2_kernel): bitwise the same statistics (tests/test_gpu_deferred.py).
// diagnostics (TRLDA_MERGED_STAMPS=1, tools/deferred_stamps.py): [s_memrealtime at the start, where
// it ran (XCC << 16 | HW_ID: SE, CU), s_memrealtime at the end] of every workgroup of a deferred
// launch -- statistics at rows [0, 1024), documents from 1024, the next batch's preamble from 2048
struct DeferredStamp {
    unsigned long long *row;
    __device__ __forceinline__ DeferredStamp(const MergedArgs &mg, int r) : row(nullptr)
    {
        if (mg.tstamps && threadIdx.x == 0) {        // launch-uniform test first
            row = mg.tstamps + 3 * (size_t)r;
            row[0] = __builtin_amdgcn_s_memrealtime();
            row[1] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 16) |
                     (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffffu);
        }
    }
    __device__ __forceinline__ void end()
    {
        if (row)
            row[2] = __builtin_amdgcn_s_memrealtime();
    }
};

// The helpers of a deferred launch take their work from a COUNTER: the launch has H helper
// workgroups behind the documents (as many as it has helper items, at most one per CU); a helper
// takes whatever item comes next -- first the pre.nb workgroups' worth
// of the next batch's preamble, then the n_short + n_long of the statistics.  The helpers that are
// resident from the start (the CUs the documents leave free) work through the list while the
// documents run; the ones dispatched when documents end take what is left.  Measured on the way
// here (profiles/r05_deferred_notes.txt): a workgroup per item pays ~1.2 us between its
// predecessor's end and its own first instruction (375 items on 56 CUs: a fifth of the shadow the
// documents cast); H persistent workgroups with a FIXED share each cannot spread over the CUs the
// documents free (49.8 against 41.8 us per step).  The next item is requested (one relaxed atomic by
// thread 0) before the current one is worked on, so its latency is not on anybody's path; the
// counter only grows -- the host passes the value it has at the start of the launch and adds the
// launch's items afterwards.
// ---- the statistics stage of a MERGED launch in the same 16-slot form (round 5) ---------------------
// merged_stats walks a list four rows at a time: the workgroups with the longest lists go through
// up to four dependent rounds of gathers after the documents' flag (flag -> end: median 2.6 us, the
// longest 4.7; with the M-step 3.6 / 5.6 -- profiles/r04_merged_stamps.txt), and the launch ends
// with them.  Here every wave is ONE round of at most 16 rows, the lists in classes by length as
// in the deferred stage, at most four words per wave (the M-step's exp(psi(lambda)) is two
// evaluations per lane and word, one after the other).  Before the flag: descriptors, documents,
// exp(psi(lambda)) and lambda' of the wave's words; after it: weights and rows together (one
// latency), the sums, the M-step.  Same sums in the same order: bitwise the statistics of
// merged_stats and of the kernel of its own.
constexpr int kMergedSlotNW[4] = {1, 2, 4, 4};   // words per wave by class (entries 9..16 | 5..8 | 3..4 | 1..2)

__host__ __device__ inline int merged_slot_short_items(const int (&c)[4])
{
    int n = 0;
    for (int i = 0; i < 4; ++i)
        n += (c[i] + 8 * kMergedSlotNW[i] - 1) / (8 * kMergedSlotNW[i]);
    return n;
}

__device__ __forceinline__ void merged_stats_slots(const MergedArgs &mg, int vb, double *lds)
{
    constexpr int W = kRegThreads / kWave;           // 8 waves
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int K = mg.K;
    const UpdateOut &o = mg.o;
    const int kk = min(2 * lane, K - 2);             // this lane's pair of topics (K even)
    const bool k_on = 2 * lane < K;
    const int n_stat = mg.n_short + mg.n_long;
    const unsigned int seen = merged_flag_load(mg, vb);          // (requested before everything else)

    // columns of the words outside the batch: zero (lda.cpp:169), written while the documents run
    if (mg.active_flag && o.sstats && !o.lambda) {   // launch-uniform
        for (int w0 = (vb * W + wid) * kWave; w0 < mg.V; w0 += n_stat * W * kWave) {
            const int w = w0 + lane;
            unsigned long long todo = __ballot(w < mg.V && mg.active_flag[w] == 0);
            while (todo) {                           // wave-uniform
                const int j = __builtin_ctzll(todo);
                todo &= todo - 1ull;
                if (k_on)
                    *reinterpret_cast<double2 *>(o.sstats + (size_t)(w0 + j) * K + 2 * lane) = make_double2(0.0, 0.0);
            }
        }
    }

    if (vb < mg.n_short) {
        int cls = 0, it = vb, base = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int items = (mg.cls_short[c] + W * kMergedSlotNW[c] - 1) / (W * kMergedSlotNW[c]);
            if (cls == c && it >= items) {
                it -= items;
                base += mg.cls_short[c];
                cls = c + 1;
            }
        }
        const int lr = 4 - cls, R = 1 << lr;
        const int nw = cls == 0 ? 1 : cls == 1 ? 2 : 4;          // kMergedSlotNW
        const int t_end = base + (cls == 0 ? mg.cls_short[0] : cls == 1 ? mg.cls_short[1]
                                  : cls == 2 ? mg.cls_short[2] : mg.cls_short[3]);
        const int t = base + (it * W + wid) * nw;
        // ---- before the flag: everything that does not depend on the documents
        const int sl = min(lane, 15), j = sl >> lr, u = sl & (R - 1);
        const bool w_on = j < nw && t + j < t_end;
        const int4 d = mg.desc[min(t + j, mg.N_short - 1)];     // (word, first entry, entries, 0)
        const bool on = lane < 16 && w_on && u < d.z;
        const int q = on ? d.y + u : 0;
        const int doc = on ? mg.wdoc[q] : -1;                   // -1: the zero row
        const int wword = w_on ? d.x : -1;
        double2 e2[8], lp[8];                        // (R >= 2: segments start at even slots)
#pragma unroll
        for (int s0 = 0; s0 < 16; s0 += 2) {
            e2[s0 >> 1] = make_double2(0.0, 0.0);
            lp[s0 >> 1] = make_double2(0.0, 0.0);
            if ((s0 & (R - 1)) == 0) {               // wave-uniform: a segment starts here
                const size_t ic = (size_t)max(__builtin_amdgcn_readlane(wword, s0), 0) * K + kk;
                e2[s0 >> 1] = *reinterpret_cast<const double2 *>(mg.eeb + ic);
                if (o.lambda_prime)                  // launch-uniform
                    lp[s0 >> 1] = *reinterpret_cast<const double2 *>(o.lambda_prime + ic);
            }
        }
        merged_wait_docs(mg, vb, seen);
        // ---- after it: the weights and the rows together
        const double tw = on ? __hip_atomic_load(const_cast<double *>(mg.tw_word) + q, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT)
                             : 0.0;
        double2 ev[16];
#pragma unroll
        for (int s0 = 0; s0 < 16; ++s0) {
            const long long row = (long long)__builtin_amdgcn_readlane(doc, s0) * K;
            ev[s0] = *reinterpret_cast<const double2 *>(mg.epg + row + kk);
        }
        const int tlo = __double2loint(tw), thi = __double2hiint(tw);
        double2 acc = make_double2(0.0, 0.0), ec = make_double2(0.0, 0.0), lc = make_double2(0.0, 0.0);
        double2 rs = make_double2(0.0, 0.0);
#pragma unroll
        for (int s0 = 0; s0 < 16; ++s0) {
            if ((s0 & 1) == 0 && (s0 & (R - 1)) == 0) {
                ec = e2[s0 >> 1];
                lc = lp[s0 >> 1];
            }
            const double tu = __hiloint2double(__builtin_amdgcn_readlane(thi, s0), __builtin_amdgcn_readlane(tlo, s0));
            acc.x = fma(tu, ev[s0].x, acc.x);        // (+0 * 0 past a list's end)
            acc.y = fma(tu, ev[s0].y, acc.y);
            if (((s0 + 1) & (R - 1)) == 0) {         // wave-uniform: the segment ends here
                const int w = __builtin_amdgcn_readlane(wword, s0);
                if (w >= 0 && k_on) {
                    const double2 lam = merged_update_pair(o, (size_t)w * K + 2 * lane,
                                                           make_double2(acc.x * ec.x, acc.y * ec.y), lc);
                    rs.x += lam.x;
                    rs.y += lam.y;
                }
                acc = make_double2(0.0, 0.0);
            }
        }
        if (o.partial) {                             // launch-uniform
            if (k_on)
                *reinterpret_cast<double2 *>(lds + wid * K + 2 * lane) = rs;
            __syncthreads();
            for (int k = tid; k < K; k += kRegThreads) {
                double sum = lds[k];
#pragma unroll
                for (int c = 1; c < W; ++c)
                    sum += lds[c * K + k];
                o.partial[(size_t)vb * K + k] = sum;
            }
        }
        return;
    }

    // ---- long lists: LW per workgroup (1 | 1 | 2 | 4 by class), wave `wid` walks chunks wid and
    // wid + 8 of each; thread (i, k) = (tid / TPW, tid % TPW) finishes topic k of list i
    int cls = 0, it = vb - mg.n_short, base = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int items = (mg.cls_long[c] + kDeferLongLW[c] - 1) / kDeferLongLW[c];
        if (cls == c && it >= items) {
            it -= items;
            base += mg.cls_long[c];
            cls = c + 1;
        }
    }
    const int lr = 4 - cls, R = 1 << lr;
    const bool two_pass = lr == 4;
    const int LW = cls == 0 ? 1 : 16 >> (5 - cls);   // kDeferLongLW: 1, 1, 2, 4
    const int t_end = base + (cls == 0 ? mg.cls_long[0] : cls == 1 ? mg.cls_long[1]
                              : cls == 2 ? mg.cls_long[2] : mg.cls_long[3]);
    const int t = base + it * LW;
    const int TPW = kRegThreads / LW;                // >= 128 >= K
    const int fi = tid / TPW, fk = tid % TPW;
    const int4 df = mg.desc[mg.N_short + min(t + fi, t_end - 1)];
    const bool fin_on = t + fi < t_end && fk < K;
    const size_t fidx = (size_t)df.x * K + min(fk, K - 1);
    const double ek = mg.eeb[fidx];
    const double lpk = o.lambda_prime ? o.lambda_prime[fidx] : 0.0;
    // the segments' documents (static): pass 0 and, for the longest class, pass 1
    int docs[2], qq[2];
    bool ons[2];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int sl = min(lane, 15), g = two_pass ? pass : sl >> lr, u = sl & (R - 1);
        const int i = g >> 1, h = g & 1;
        const bool w_on = t + i < t_end && (two_pass || pass == 0);
        const int4 d = mg.desc[mg.N_short + min(t + i, t_end - 1)];
        const int L = w_on ? d.z : 0;
        const int chunk = (L + 15) / 16;             // <= R
        const int c0 = min(L, (wid + 8 * h) * chunk);
        const int cl = min(L, c0 + chunk) - c0;
        ons[pass] = lane < 16 && u < cl;
        qq[pass] = ons[pass] ? d.y + c0 + u : 0;
        docs[pass] = ons[pass] ? mg.wdoc[qq[pass]] : -1;
    }
    merged_wait_docs(mg, vb, seen);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 0 || two_pass) {                 // block-uniform
            const double tw = ons[pass] ? __hip_atomic_load(const_cast<double *>(mg.tw_word) + qq[pass],
                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                        : 0.0;
            double2 ev[16];
#pragma unroll
            for (int s0 = 0; s0 < 16; ++s0) {
                const long long row = (long long)__builtin_amdgcn_readlane(docs[pass], s0) * K;
                ev[s0] = *reinterpret_cast<const double2 *>(mg.epg + row + kk);
            }
            const int tlo = __double2loint(tw), thi = __double2hiint(tw);
            double2 acc = make_double2(0.0, 0.0);
#pragma unroll
            for (int s0 = 0; s0 < 16; ++s0) {
                const double tu = __hiloint2double(__builtin_amdgcn_readlane(thi, s0),
                                                   __builtin_amdgcn_readlane(tlo, s0));
                acc.x = fma(tu, ev[s0].x, acc.x);
                acc.y = fma(tu, ev[s0].y, acc.y);
                if (((s0 + 1) & (R - 1)) == 0) {     // wave-uniform: the segment ends here
                    const int gs = two_pass ? pass : s0 >> lr;
                    if (k_on)
                        *reinterpret_cast<double2 *>(lds + (size_t)((gs >> 1) * 16 + wid + 8 * (gs & 1)) * K + 2 * lane) = acc;
                    acc = make_double2(0.0, 0.0);
                }
            }
        }
    }
    __syncthreads();
    double lam = 0.0;
    if (fin_on) {
        const double *col = lds + (size_t)(fi * 16) * K + fk;
        // (the sixteen chunk sums requested together, added in chunk order: left to the compiler they
        // were fifteen LDS round trips one behind the other)
        double cv[16];
#pragma unroll
        for (int c = 0; c < 16; ++c)
            cv[c] = col[(size_t)c * K];
        __builtin_amdgcn_sched_barrier(0);
        double sum = cv[0];
#pragma unroll
        for (int c = 1; c < 16; ++c)
            sum += cv[c];
        const double sv = sum * ek;
        if (o.sstats)
            o.sstats[fidx] = sv;
        if (o.lambda) {
            const double hat = o.eta + o.scale * sv;
            lam = o.lambda_prime ? o.omr * lpk + o.rho * hat : o.rho * hat;
            o.lambda[fidx] = lam;
            if (o.u_out)
                o.u_out[fidx] = exp_digamma_positive(lam);
        }
    }
    if (o.partial) {                                 // launch-uniform: this workgroup's row of lambda sums
        __syncthreads();                             // (the chunk sums have been read)
        if (fk < K)
            lds[fi * K + fk] = lam;                  // (0 for a list past the end)
        __syncthreads();
        if (tid < K) {
            double sum = lds[tid];
            for (int i = 1; i < LW; ++i)
                sum += lds[i * K + tid];
            o.partial[(size_t)vb * K + tid] = sum;
        }
    }
}

__device__ __forceinline__ void deferred_helper(const PreArgs &pre, const MergedArgs &mg, double *lds)
{
    __shared__ unsigned int next_item[2];
    const int n_items = pre.nb + mg.n_short + mg.n_long;
    // The first item: the helper's own index when ALL helpers are resident from the start (the
    // host launches exactly as many as the documents leave CUs free: first_static = H) -- a round
    // trip to the counter is ~1.8 us in front of a helper's first item; from the counter otherwise
    // (with item h for helpers that are only dispatched when the documents end, the items past the
    // resident ones waited for them: 38.3 against 32.9 us per step).
    int v = (int)blockIdx.x - pre.n_docs;
    if (mg.first_static == 0) {                      // launch-uniform
        if (threadIdx.x == 0)
            next_item[1] = __hip_atomic_fetch_add(mg.work_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                           mg.work_base;
        __syncthreads();
        v = (int)next_item[1];
    }
    for (int round = 0; v < n_items; ++round) {      // block-uniform
        unsigned int fetched = 0u;
        if (threadIdx.x == 0)                        // (needed at the end of the round)
            fetched = __hip_atomic_fetch_add(mg.work_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (published from inside the item, between its loads and its stores: PublishNext)
        const PublishNext pub{&next_item[round & 1], fetched, mg.work_base - (unsigned int)mg.first_static};
        // A zero the compiler cannot see through, added to every item's thread index: without it
        // everything in the items that does not depend on the item -- a few hundred instructions of
        // per-lane addresses, class tables, argument loads, and 25 registers spilled to make room for
        // them -- is hoisted in front of the loop, and a helper's first item started 3 us into the
        // launch (the documents: 0.2 us; profiles/r05_deferred_notes.txt).
        int zero = 0;
        asm volatile("" : "+v"(zero));
        if (v < pre.nb) {
            DeferredStamp st(mg, 2048 + min(v, 1023));
            docs_launch_preamble<true>(pre, lds, pre.n_docs + v, pub, zero);
            st.end();
        } else {
            DeferredStamp st(mg, min(v - pre.nb, 1023));
            deferred_stats(mg, v - pre.nb, lds, pub, zero);
            st.end();
        }
        __syncthreads();                             // (also: the next item reuses the LDS)
        v = (int)next_item[round & 1];
    }
}

template <int MODE>
__global__ __launch_bounds__(kRegThreads) void estep_docs_reg_deferred_kernel(DocKernelArgs a, PreArgs pre,
                                                                               MergedArgs mg)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    warm_kernel_arguments();                         // (first thing: every later argument load hits)
    const int bid = (int)blockIdx.x;
    if (bid >= pre.n_docs) {                         // block-uniform
        deferred_helper(pre, mg, lds);
        return;
    }
    DeferredStamp st(mg, 1024 + min(bid, 1023));
    if (a.docs_per_wg == 8)                          // launch-uniform: a wave per document (K <= 32)
        estep_docs_small_body(a, lds);
    else
        estep_docs_reg_body<MODE>(a, lds);
    st.end();
}

template <int KS>
__global__ __launch_bounds__(kRegThreads) void estep_docs_tiered_deferred_kernel(DocKernelArgs a, PreArgs pre,
                                                                                  int lds_rows, MergedArgs mg)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    warm_kernel_arguments();
    const int bid = (int)blockIdx.x;
    if (bid >= pre.n_docs) {                         // block-uniform
        deferred_helper(pre, mg, lds);
        return;
    }
    const int n = a.pad_meta[4 * (size_t)bid * a.meta_i4 + 1];
    if (a.docs_per_wg == 8 && bid >= a.small_block0) {   // block-uniform: K <= 32, eight short documents
        estep_docs_small_body(a, lds);
    } else if (a.meta_i4 == 2 && a.pad_meta[4 * ((size_t)bid * 2 + 1) + 1] > 1) {
        estep_docs_reg_body<0, true>(a, lds);        // one segment of a document split over CUs
    } else if (n <= 128) {
        estep_docs_reg_body<0>(a, lds);
    } else if (n <= 144) {
        estep_docs_reg_body<1>(a, lds);
    } else {
        const int4 meta = reinterpret_cast<const int4 *>(a.pad_meta)[(size_t)bid * a.meta_i4];
        estep_docs_wide_body<KS, true>(a, lds_rows, lds, meta.x, meta.z, meta.y);
    }
}

}  // namespace trlda
