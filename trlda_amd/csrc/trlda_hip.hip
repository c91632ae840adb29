// trlda_hip.hip -- C ABI (include/trlda_hip.h) over the gfx950 kernels in
// estep_kernels.h.  Host logic only: validation, device memory, launch sequences,
// and the control loops of OnlineLDA::updateParameters (reference
// src/onlinelda.cpp:53-111) and BatchLDA::updateParameters (src/batchlda.cpp:43-61).
//
// There is no CPU compute path in this file: if HIP cannot see a device, every
// compute entry point fails with TRLDA_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <time.h>
#include <unistd.h>
#include <vector>

#include "../../include/trlda_hip.h"
#include "estep_kernels.h"
#include "estep_wide.h"
#include "elbo_kernels.h"

namespace {

thread_local std::string g_error;

int fail(int code, const std::string &msg)
{
    g_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t err__ = (expr);                                                      \
        if (err__ != hipSuccess)                                                        \
            return fail(err__ == hipErrorNoDevice || err__ == hipErrorInvalidDevice     \
                            ? TRLDA_ERR_NO_DEVICE                                       \
                            : TRLDA_ERR_HIP,                                            \
                        std::string(#expr) + ": " + hipGetErrorString(err__));          \
    } while (0)

constexpr int kLdsBytes = 160 * 1024;   // LDS per workgroup on gfx950
#ifdef TRLDA_STAMPS
unsigned long long *g_stamp_buf = nullptr;
#endif
constexpr int kDenseThreads = 256;
constexpr int kMaxRowsumBlocks = 1025;   // 1024 blocks on large tables + the combined row

template <typename T>
int dev_alloc(T **p, size_t count)
{
    *p = nullptr;
    if (count == 0)
        count = 1;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T)));
    return TRLDA_OK;
}

int use_device(int device)
{
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess || n <= 0)
        return fail(TRLDA_ERR_NO_DEVICE,
                    "no HIP device available (libtrlda_hip has no CPU fallback)");
    if (device < 0 || device >= n)
        return fail(TRLDA_ERR_NO_DEVICE, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    return TRLDA_OK;
}

}  // namespace

struct trlda_batch {
    int device = 0;
    int V = 0, B = 0, max_n = 0;
    int64_t nnz = 0;
    int32_t *indptr = nullptr, *ids = nullptr, *cnts = nullptr;
    int32_t *order = nullptr;   // documents by decreasing length
    int32_t *wrank = nullptr;   // CSR position -> rank in word-major order
    int32_t *wptr = nullptr;    // V+1 word segment offsets
    int32_t *wdoc = nullptr;    // document of each word-major entry
    int32_t *active = nullptr;  // ids of the words that occur in the batch (ascending)
    int n_active = 0;
    int32_t *long_words = nullptr;   // words with more than kLongWord entries
    int n_long = 0;
    // per document, in `order`: (document, length, CSR offset, 0) and its first kRegMaxN word
    // ids padded to that length -- the register kernel's workgroup finds everything it needs
    // at an address that depends on its index only
    int32_t *pad_meta = nullptr;     // B x 4
    int32_t *pad_ids = nullptr;      // B x kRegMaxN
    std::vector<int32_t> sorted_len;   // host copy: document lengths in `order`
};

struct trlda_model {
    int device = 0;
    int K = 0, V = 0;
    hipStream_t stream = nullptr;
    int sstats_mode = TRLDA_SSTATS_SEGMENTED;
    int doc_threads = 0;
    int doc_kernel = 0;    // TRLDA_DOCS_*
    const char *last_doc_kernel = "";   // kernel that took most documents of the last E-step
    bool last_preamble_fused = false;
    bool split_preamble = false;        // never fuse kernels 1 and 2 (tests, comparisons)
    bool dense_preamble = false;   // true: exp E[log beta] for all V words, as the reference
    double *lambda = nullptr, *alpha = nullptr;
    double *eeb = nullptr, *psi_sum = nullptr, *partial = nullptr;
    unsigned int *counter = nullptr;
    // per-batch workspaces, grown on demand
    size_t cap_docs = 0, cap_nnz = 0;
    double *epg = nullptr, *tw_csr = nullptr, *tw_word = nullptr;
    // update_parameters workspaces
    double *lambda_prime = nullptr, *sstats = nullptr, *gamma = nullptr, *wordcounts = nullptr;
    size_t cap_gamma = 0;
    // timing: five events per E-step from a pool, resolved lazily (no host sync per step)
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;   // all events ever created
    size_t ev_used = 0;                // events recorded since the last collect
    double usec_sum[5] = {0, 0, 0, 0, 0};
    int64_t usec_cnt[5] = {0, 0, 0, 0, 0};
};

namespace {

int ensure_batch_workspace(trlda_model *m, const trlda_batch *b)
{
    if ((size_t)b->B > m->cap_docs) {
        if (m->epg)
            HIP_TRY(hipFree(m->epg));
        int rc = dev_alloc(&m->epg, (size_t)b->B * m->K);
        if (rc)
            return rc;
        m->cap_docs = (size_t)b->B;
    }
    if ((size_t)b->nnz > m->cap_nnz) {
        if (m->tw_csr)
            HIP_TRY(hipFree(m->tw_csr));
        if (m->tw_word)
            HIP_TRY(hipFree(m->tw_word));
        int rc = dev_alloc(&m->tw_csr, (size_t)b->nnz);
        if (rc)
            return rc;
        rc = dev_alloc(&m->tw_word, (size_t)b->nnz);
        if (rc)
            return rc;
        m->cap_nnz = (size_t)b->nnz;
    }
    return TRLDA_OK;
}

int ensure_update_workspace(trlda_model *m, int B)
{
    size_t KV = (size_t)m->K * m->V;
    if (!m->lambda_prime) {
        int rc = dev_alloc(&m->lambda_prime, KV);
        if (rc)
            return rc;
        rc = dev_alloc(&m->sstats, KV);
        if (rc)
            return rc;
        rc = dev_alloc(&m->wordcounts, (size_t)m->V);
        if (rc)
            return rc;
    }
    if ((size_t)B * m->K > m->cap_gamma) {
        if (m->gamma)
            HIP_TRY(hipFree(m->gamma));
        int rc = dev_alloc(&m->gamma, (size_t)B * m->K);
        if (rc)
            return rc;
        m->cap_gamma = (size_t)B * m->K;
    }
    return TRLDA_OK;
}

void collect_timing(trlda_model *m)
{
    if (m->ev_used == 0)
        return;
    if (hipEventSynchronize(m->ev_pool[m->ev_used - 1]) == hipSuccess) {
        for (size_t base = 0; base + 6 <= m->ev_used; base += 6)
            for (int i = 0; i < 5; ++i) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, m->ev_pool[base + i], m->ev_pool[base + i + 1]) ==
                    hipSuccess) {
                    m->usec_sum[i] += 1e3 * (double)ms;
                    m->usec_cnt[i] += 1;
                }
            }
    }
    m->ev_used = 0;
}

// next event of the current E-step's group of six (grows the pool on demand)
int stamp(trlda_model *m)
{
    if (m->ev_used == m->ev_pool.size()) {
        if (m->ev_pool.size() >= 6 * 8192)
            return fail(TRLDA_ERR_ARG, "timing: collect (trlda_model_get_timing) at least every "
                                       "8192 E-steps");
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        m->ev_pool.push_back(e);
    }
    HIP_TRY(hipEventRecord(m->ev_pool[m->ev_used++], m->stream));
    return TRLDA_OK;
}

template <int T>
int launch_docs(trlda_model *m, const trlda::DocKernelArgs &args, int B, size_t lds_bytes)
{
    auto kern = trlda::estep_docs_kernel<T>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3(B), dim3(T), lds_bytes, m->stream, args);
    HIP_TRY(hipGetLastError());
    return TRLDA_OK;
}

size_t docs_lds_bytes(int K, int Kp, int n_cap, int T)
{
    // beta[n_cap][Kp] | g[K] | e[K] | tw[n_cap] | cnt[n_cap] | part[max(T,K)] | wsum[T/64]
    size_t doubles = (size_t)n_cap * Kp + 2 * (size_t)K + 2 * (size_t)n_cap +
                     (size_t)std::max(T, K) + (size_t)(T / trlda::kWave);
    return doubles * sizeof(double);
}

// The E-step launch sequence on the model's stream (no synchronisation).
int estep_device(trlda_model *m, const trlda_batch *b, double *gamma_dev, double *sstats_dev,
                 int max_iter, double threshold, int32_t *iters_dev,
                 const double *gamma_in_dev = nullptr)
{
    using namespace trlda;
    const int K = m->K, V = m->V, B = b->B;
    const size_t KV = (size_t)K * V;
    if (b->V != V)
        return fail(TRLDA_ERR_SHAPE, "batch was created for a different vocabulary size");
    if (b->device != m->device)
        return fail(TRLDA_ERR_ARG, "batch and model live on different devices");
    int rc = ensure_batch_workspace(m, b);
    if (rc)
        return rc;
    const bool atomic = m->sstats_mode == TRLDA_SSTATS_ATOMIC;
    if (m->timing && (rc = stamp(m)))
        return rc;

    // 1. row sums of lambda, per block of words (lda.cpp:172).  Small tables: 64 blocks whose
    // partials every eeb block adds up itself (saves a launch).  Large ones (tens of MB and
    // more): up to 1024 blocks to fill HBM, then one small kernel combines the partials.
    const bool big = KV >= ((size_t)1 << 22);
    int G = std::min(big ? kMaxRowsumBlocks - 1 : trlda::kRowsumBlocks, std::max(1, V / 32));
    const double *partial_in = m->partial;
    // Small table and every document in the register-resident kernel's range: kernels 1 and
    // 2 become one launch and the topic factors exp(-psiSum) are applied by the document
    // kernel (estep_kernels.h, 2b)
    const bool fused = !big && B > 0 && m->doc_threads == 0 && m->doc_kernel == TRLDA_DOCS_AUTO &&
                       !m->split_preamble && K <= trlda::kRegMaxK && b->max_n <= trlda::kRegMaxN;
    m->last_preamble_fused = fused;
    if (fused) {
        constexpr int TP = 512;
        G = std::min(trlda::kRowsumBlocks, std::max(1, V / 32));
        int wpb = (V + G - 1) / G;
        G = (V + wpb - 1) / wpb;
        const bool dense = m->dense_preamble;
        const size_t total = dense ? KV : (size_t)K * (size_t)b->n_active;
        // G workgroups add up the row sums, the others fill exp(psi(lambda)): 256 in all, one
        // per CU (a 1024-thread workgroup of this kernel fills a CU's registers)
        const int GP = G + (int)std::max<size_t>(1, std::min<size_t>((total + TP - 1) / TP, (size_t)(256 * (1024 / TP) - G)));
        hipLaunchKernelGGL(preamble_fused_kernel<TP>, dim3(GP), dim3(TP), 0, m->stream, K, V, G, wpb,
                           total, m->lambda, m->partial, m->eeb, dense ? nullptr : b->active);
        HIP_TRY(hipGetLastError());
        if (m->timing && ((rc = stamp(m)) || (rc = stamp(m))))
            return rc;
    } else {
    {
        int wpb = (V + G - 1) / G;
        G = (V + wpb - 1) / wpb;
        hipLaunchKernelGGL(rowsum_partial_kernel<kDenseThreads>, dim3(G), dim3(kDenseThreads), 0,
                           m->stream, K, V, wpb, m->lambda, m->partial);
        HIP_TRY(hipGetLastError());
        if (G > trlda::kRowsumBlocks) {
            double *combined = m->partial + (size_t)G * K;       // row G of the same buffer
            hipLaunchKernelGGL(rowsum_combine_kernel<kDenseThreads>,
                               dim3((K + kDenseThreads / 8 - 1) / (kDenseThreads / 8)),
                               dim3(kDenseThreads), 0, m->stream, K, G, m->partial, combined);
            HIP_TRY(hipGetLastError());
            partial_in = combined;
            G = 1;
        }
    }
    if (m->timing && (rc = stamp(m)))
        return rc;

    // 2. psiSum + exp E[log beta] (lda.cpp:172-173), on the batch's active words unless the
    // dense preamble was asked for
    {
        constexpr int TE = 1024;
        const bool dense = m->dense_preamble;
        const size_t total = dense ? KV : (size_t)K * (size_t)b->n_active;
        size_t blocks = (total + TE - 1) / TE;
        int GE = (int)std::max<size_t>(1, std::min<size_t>(blocks, 256));
        size_t lds = (size_t)K * 9 * sizeof(double);
        if (lds > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(exp_elog_beta_kernel<TE>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(exp_elog_beta_kernel<TE>, dim3(GE), dim3(TE), lds, m->stream, K, total, G,
                           m->lambda, partial_in, m->psi_sum, m->eeb,
                           dense ? nullptr : b->active);
        HIP_TRY(hipGetLastError());
    }
    if (m->timing && (rc = stamp(m)))
        return rc;
    }

    // 3. per-document fixed point (lda.cpp:174-204)
    if (atomic)
        HIP_TRY(hipMemsetAsync(sstats_dev, 0, KV * sizeof(double), m->stream));  // lda.cpp:169
    if (B > 0) {
        DocKernelArgs a;
        a.K = K; a.B = B;
        a.stamps = nullptr;
#ifdef TRLDA_STAMPS
        {
            static unsigned long long *stamp_buf = nullptr;   // diagnostic build only
            if (!stamp_buf) {
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&stamp_buf), 65536 * 8 * 8));
                HIP_TRY(hipMemset(stamp_buf, 0, 65536 * 8 * 8));
            }
            a.stamps = stamp_buf;
            g_stamp_buf = stamp_buf;
        }
#endif
        a.indptr = b->indptr; a.ids = b->ids; a.cnts = b->cnts;
        a.eeb = m->eeb; a.alpha = m->alpha;
        a.gamma = gamma_dev; a.gamma_in = gamma_in_dev ? gamma_in_dev : gamma_dev;
        a.epg = m->epg; a.tw_csr = m->tw_csr;
        a.wrank = b->wrank; a.tw_word = m->tw_word;
        a.sstats_acc = atomic ? sstats_dev : nullptr;
        a.max_iter = max_iter; a.threshold = threshold; a.iters_out = iters_dev;
        a.partial = fused ? m->partial : nullptr;
        a.G = G;
        a.scale_out = fused ? m->psi_sum : nullptr;
        const int Kp = K | 1;

        // Documents are ordered by decreasing length and split into two runs:
        //   [B - n_reg, B)   K <= 128 and at most 192 words: slice in registers, both
        //                    orientations (estep_docs_reg_kernel)
        //   [0, B - n_reg)   everything else
        int n_reg = 0;
        if (m->doc_threads == 0 && K <= kRegMaxK && m->doc_kernel != TRLDA_DOCS_WIDE)
            while (n_reg < B && b->sorted_len[(size_t)(B - 1 - n_reg)] <= kRegMaxN)
                ++n_reg;

        // 128 < K <= 512, or K <= 128 with more than 192 words: registers in one orientation
        // (estep_wide.h); it takes every document the register tier does not.  Beyond 512
        // topics (or on request): the general kernel, slice in LDS when it fits, else streamed.
        const bool wide = m->doc_threads == 0 && K <= kWideMaxK && m->doc_kernel != TRLDA_DOCS_GENERAL;
        const int n_stream = B - n_reg;

        if (B - n_reg >= n_reg)
            m->last_doc_kernel = wide ? "estep_docs_wide_kernel" : "estep_docs_kernel";
        else
            m->last_doc_kernel = "estep_docs_reg_kernel";
        if (wide && B - n_reg > 0) {
            const int n_wide = B - n_reg;
            const int KS = (K + kWave - 1) / kWave;
            int jw = 0;
            switch (KS) {
            case 1: jw = wide_cfg<1>::JW; break;
            case 2: jw = wide_cfg<2>::JW; break;
            case 3: jw = wide_cfg<3>::JW; break;
            case 4: jw = wide_cfg<4>::JW; break;
            case 5: jw = wide_cfg<5>::JW; break;
            case 6: jw = wide_cfg<6>::JW; break;
            case 7: jw = wide_cfg<7>::JW; break;
            default: jw = wide_cfg<8>::JW; break;
            }
            // LDS rows for the words past the registers, as many as the longest document needs
            const size_t fixed = wide_lds_doubles(KS, 0) * sizeof(double);
            const int fit = (int)(((size_t)kLdsBytes - fixed) / ((size_t)(64 * KS + 1) * sizeof(double)));
            const int lds_rows = std::max(0, std::min(fit, b->max_n - kWideWaves * jw));
            const size_t lds_bytes = wide_lds_doubles(KS, lds_rows) * sizeof(double);
            a.n_cap = 0;
            a.Kp = 64 * KS;
            a.order = b->order;
#define TRLDA_LAUNCH_WIDE(KSV)                                                             \
    do {                                                                                   \
        auto kern = estep_docs_wide_kernel<KSV>;                                           \
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                  \
                                    hipFuncAttributeMaxDynamicSharedMemorySize,            \
                                    (int)lds_bytes));                                      \
        hipLaunchKernelGGL(kern, dim3(n_wide), dim3(kWideThreads), lds_bytes, m->stream, a, \
                           lds_rows);                                                      \
    } while (0)
            switch (KS) {
            case 1: TRLDA_LAUNCH_WIDE(1); break;
            case 2: TRLDA_LAUNCH_WIDE(2); break;
            case 3: TRLDA_LAUNCH_WIDE(3); break;
            case 4: TRLDA_LAUNCH_WIDE(4); break;
            case 5: TRLDA_LAUNCH_WIDE(5); break;
            case 6: TRLDA_LAUNCH_WIDE(6); break;
            case 7: TRLDA_LAUNCH_WIDE(7); break;
            default: TRLDA_LAUNCH_WIDE(8); break;
            }
#undef TRLDA_LAUNCH_WIDE
            HIP_TRY(hipGetLastError());
        } else if (n_stream > 0) {
            int T = m->doc_threads > 0 ? m->doc_threads : 256;
            size_t fixed = docs_lds_bytes(K, Kp, 0, T);
            int n_fit = fixed >= (size_t)kLdsBytes
                            ? 0
                            : (int)(((size_t)kLdsBytes - fixed) /
                                    ((size_t)(Kp + 2) * sizeof(double)));
            int gen_cap = std::min(b->max_n, n_fit);
            size_t lds_bytes = docs_lds_bytes(K, Kp, gen_cap, T);
            if (lds_bytes > (size_t)kLdsBytes)
                return fail(TRLDA_ERR_ARG,
                            "num_topics too large for the document kernel's LDS layout");
            a.n_cap = gen_cap;
            a.Kp = Kp;
            a.order = b->order;
            switch (T) {
            case 64: rc = launch_docs<64>(m, a, n_stream, lds_bytes); break;
            case 128: rc = launch_docs<128>(m, a, n_stream, lds_bytes); break;
            case 256: rc = launch_docs<256>(m, a, n_stream, lds_bytes); break;
            case 512: rc = launch_docs<512>(m, a, n_stream, lds_bytes); break;
            case 1024: rc = launch_docs<1024>(m, a, n_stream, lds_bytes); break;
            default: return fail(TRLDA_ERR_ARG, "doc_threads must be 64, 128, 256, 512 or 1024");
            }
            if (rc)
                return rc;
        }
        if (n_reg > 0) {
            a.n_cap = 0;
            a.Kp = K;
            // One launch; the variant follows the longest document of the tier (it comes first):
            // up to 128 words, up to 144 (all in registers still), up to 192 (LDS tail).  A
            // launch lasts as long as its longest document and the variants cost 33 / 36 / 42 us
            // at K = 100.  Running two variants on two streams was measured and lost: the event
            // fork/join costs more (~12 us) than it saves.
            // (With more documents than CUs it is throughput that counts, and there <2> wins over
            // <1>: it only charges the long documents for their tail.)
            const int longest = b->sorted_len[(size_t)(B - n_reg)];
            auto kern = longest <= 128                    ? estep_docs_reg_kernel<0>
                        : longest <= 144 && n_reg <= 256 ? estep_docs_reg_kernel<1>
                                                          : estep_docs_reg_kernel<2>;
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)kRegLdsBytes));
            a.order = b->order + (B - n_reg);
            a.pad_meta = b->pad_meta + (size_t)(B - n_reg) * 4;
            a.pad_ids = b->pad_ids + (size_t)(B - n_reg) * kRegMaxN;
            hipLaunchKernelGGL(kern, dim3(n_reg), dim3(kRegThreads), kRegLdsBytes, m->stream, a);
            HIP_TRY(hipGetLastError());
        }
    }
    if (m->timing && (rc = stamp(m)))
        return rc;

    // 4. sufficient statistics (lda.cpp:207-217)
    if (atomic) {
        size_t blocks = (KV + kDenseThreads - 1) / kDenseThreads;
        int G = (int)std::min<size_t>(blocks, 256 * 8);
        hipLaunchKernelGGL(finish_kernel<kDenseThreads>, dim3(G), dim3(kDenseThreads), 0, m->stream,
                           KV, m->eeb, sstats_dev);
    } else {
        // one wavefront per word; 16 words per workgroup for small K, 8 from K = 256 on
        // (measured: 7.1 vs 7.4 us at K = 100, 170 vs 145 us at K = 500)
#define TRLDA_LAUNCH_SSTATS(TS)                                                            \
    do {                                                                                   \
        constexpr int wpb = TS / kWave;                                                    \
        const int G_short = (V + wpb - 1) / wpb;                                           \
        size_t lds = (size_t)wpb * K * sizeof(double);                                     \
        auto kern = sstats_words_kernel<TS>;                                               \
        if (lds > 48 * 1024)                                                               \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),              \
                                        hipFuncAttributeMaxDynamicSharedMemorySize,        \
                                        (int)lds));                                        \
        hipLaunchKernelGGL(kern, dim3(G_short + b->n_long), dim3(TS), lds, m->stream, K, V, \
                           G_short, b->wptr, b->wdoc, b->long_words, m->tw_word, m->epg,   \
                           m->eeb, sstats_dev);                                            \
    } while (0)
        if (K >= 256)
            TRLDA_LAUNCH_SSTATS(512);
        else
            TRLDA_LAUNCH_SSTATS(1024);
#undef TRLDA_LAUNCH_SSTATS
    }
    HIP_TRY(hipGetLastError());
    // the sixth event follows the fifth at once: the interval between them is what two
    // event records cost on an otherwise idle stream position (which = 4)
    if (m->timing && ((rc = stamp(m)) || (rc = stamp(m))))
        return rc;
    return TRLDA_OK;
}

int blend_device(trlda_model *m, const double *lambda_prime, const double *sstats, double rho,
                 double eta, double scale)
{
    size_t KV = (size_t)m->K * m->V;
    size_t blocks = (KV + kDenseThreads - 1) / kDenseThreads;
    int G = (int)std::min<size_t>(blocks, 256 * 8);
    hipLaunchKernelGGL(trlda::blend_kernel<kDenseThreads>, dim3(G), dim3(kDenseThreads), 0,
                       m->stream, KV, rho, eta, scale, lambda_prime, sstats, m->lambda);
    HIP_TRY(hipGetLastError());
    return TRLDA_OK;
}

int wordcounts_device(trlda_model *m, const trlda_batch *b, double *wc)
{
    HIP_TRY(hipMemsetAsync(wc, 0, (size_t)m->V * sizeof(double), m->stream));
    if (b->nnz > 0) {
        size_t blocks = ((size_t)b->nnz + kDenseThreads - 1) / kDenseThreads;
        int G = (int)std::min<size_t>(blocks, 256 * 8);
        hipLaunchKernelGGL(trlda::wordcount_kernel<kDenseThreads>, dim3(G), dim3(kDenseThreads), 0,
                           m->stream, b->nnz, b->ids, b->cnts, wc);
        HIP_TRY(hipGetLastError());
    }
    return TRLDA_OK;
}

int tr_init_wc_device(trlda_model *m, const double *wc, const double *lambda_prime, double rho,
                      double eta, double coef)
{
    size_t KV = (size_t)m->K * m->V;
    size_t blocks = (KV + kDenseThreads - 1) / kDenseThreads;
    int G = (int)std::min<size_t>(blocks, 256 * 8);
    hipLaunchKernelGGL(trlda::tr_init_kernel<kDenseThreads>, dim3(G), dim3(kDenseThreads), 0,
                       m->stream, m->K, KV, rho, eta, coef, wc, lambda_prime, m->lambda);
    HIP_TRY(hipGetLastError());
    return TRLDA_OK;
}

int tr_init_device(trlda_model *m, const trlda_batch *b, const double *lambda_prime, double rho,
                   double eta, int num_documents)
{
    int rc = ensure_update_workspace(m, b->B);
    if (rc)
        return rc;
    rc = wordcounts_device(m, b, m->wordcounts);
    if (rc)
        return rc;
    // static_cast<double>(D) / B / K, evaluated in the reference's order (onlinelda.cpp:86)
    double coef = (double)num_documents / (double)b->B / (double)m->K;
    return tr_init_wc_device(m, m->wordcounts, lambda_prime, rho, eta, coef);
}

int check_model(const trlda_model *m)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "model is NULL");
    return use_device(m->device);
}

}  // namespace

// ===========================================================================
extern "C" {

const char *trlda_last_error(void) { return g_error.c_str(); }

int trlda_version(void) { return 100; }

int trlda_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

// ---- host RNG ---------------------------------------------------------------
//
// The reference draws lambda0 and every default gamma0 from libc rand() through
// Eigen::Random (src/utils.cpp:224-231, Eigen/src/Core/MathFunctions.h:439-446).  glibc's
// rand() is the TYPE_3 additive-feedback generator of random_r.c (x[i] = x[i-3] + x[i-31],
// output x >> 1) behind a lock that costs ~20 ns per call -- 2*10^6 calls per default gamma0 at
// K=100, B=200.  The same recurrence is reproduced here without the lock (the stream is
// checked against libc's in tests/test_boundary.py), and the logarithms -- glibc's own
// log(), so the values stay bit-identical -- are taken by a few threads over disjoint
// elements, each element accumulating its passes in order.
} // extern "C"

namespace {

struct GlibcRandom {
    uint32_t x[31];
    int f = 3, b = 0;
    void seed(unsigned int s)
    {
        // srandom_r, TYPE_3
        int32_t word = s == 0 ? 1 : (int32_t)s;
        x[0] = (uint32_t)word;
        for (int i = 1; i < 31; ++i) {
            const long hi = word / 127773, lo = word % 127773;
            long w = 16807 * lo - 2836 * hi;
            if (w < 0)
                w += 2147483647;
            word = (int32_t)w;
            x[i] = (uint32_t)word;
        }
        f = 3;
        b = 0;
        for (int i = 0; i < 310; ++i)
            (void)next();
    }
    inline uint32_t next()
    {
        x[f] += x[b];
        const uint32_t out = x[f] >> 1;
        if (++f == 31)
            f = 0;
        if (++b == 31)
            b = 0;
        return out;
    }
};

// ---- jump-ahead for the generator above ---------------------------------------------
// The unshifted sequence obeys s_n = s_{n-31} + s_{n-3} (mod 2^32): the window
// W_n = (s_{n-31} .. s_{n-1}) advances by a 31 x 31 companion matrix A over Z / 2^32, and
// A^N (square and multiply, cached per N) jumps N draws ahead.  sampleGamma's K*B*100 draws
// are consumed pass by pass (utils.cpp:224-231); with the jumps every host thread produces
// the draws of its own element range for all passes -- the same numbers in the same order of
// additions as one serial stream, so seeded trajectories stay bit-identical.
struct JumpMatrix {
    uint32_t a[31][31];
};

void jump_identity(JumpMatrix &m)
{
    std::memset(m.a, 0, sizeof(m.a));
    for (int i = 0; i < 31; ++i)
        m.a[i][i] = 1;
}

void jump_multiply(const JumpMatrix &x, const JumpMatrix &y, JumpMatrix &out)
{
    for (int i = 0; i < 31; ++i) {
        uint32_t row[31] = {0};
        for (int k = 0; k < 31; ++k) {
            const uint32_t xik = x.a[i][k];
            if (xik == 0)
                continue;
            for (int j = 0; j < 31; ++j)
                row[j] += xik * y.a[k][j];
        }
        std::memcpy(out.a[i], row, sizeof(row));
    }
}

const JumpMatrix &jump_power(uint64_t n)
{
    static std::mutex mu;
    static std::map<uint64_t, JumpMatrix> cache;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(n);
    if (it != cache.end())
        return it->second;
    JumpMatrix base, result, tmp;
    std::memset(base.a, 0, sizeof(base.a));
    for (int i = 0; i < 30; ++i)
        base.a[i][i + 1] = 1;                        // shift
    base.a[30][0] = 1;                               // s_n = s_{n-31} + s_{n-3}
    base.a[30][28] = 1;
    jump_identity(result);
    for (uint64_t e = n; e; e >>= 1) {
        if (e & 1) {
            jump_multiply(result, base, tmp);
            result = tmp;
        }
        jump_multiply(base, base, tmp);
        base = tmp;
    }
    if (cache.size() > 64)
        cache.clear();
    return cache.emplace(n, result).first->second;
}

// window (oldest first) <-> the circular buffer of GlibcRandom
void rng_to_window(const GlibcRandom &g, uint32_t (&w)[31])
{
    for (int j = 0; j < 31; ++j)
        w[j] = g.x[(g.f + j) % 31];
}
void window_to_rng(const uint32_t (&w)[31], GlibcRandom &g)
{
    g.f = 3;
    g.b = 0;
    for (int j = 0; j < 31; ++j)
        g.x[(3 + j) % 31] = w[j];
}
void jump_apply(const JumpMatrix &m, uint32_t (&w)[31])
{
    uint32_t out[31];
    for (int i = 0; i < 31; ++i) {
        uint32_t acc = 0;
        for (int j = 0; j < 31; ++j)
            acc += m.a[i][j] * w[j];
        out[i] = acc;
    }
    std::memcpy(w, out, sizeof(out));
}

GlibcRandom g_rng;

// A few persistent host threads for the gamma draw: starting 20 threads costs ~0.3 ms, as
// much as the draw itself.  run(n, job) executes job(0) on the caller and job(1..n-1) on the
// workers and returns when all are done.  One caller at a time (the Python surface holds the
// GIL across the call, as the reference does).
class HostPool {
public:
    ~HostPool()
    {
        {
            std::lock_guard<std::mutex> lock(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_)
            t.join();
    }
    void run(int n, const std::function<void(int)> &job)
    {
        std::lock_guard<std::mutex> serial(run_mu_);
        if (n <= 1) {
            job(0);
            return;
        }
        {
            std::lock_guard<std::mutex> lock(mu_);
            while ((int)workers_.size() < n - 1) {
                const int id = (int)workers_.size() + 1;
                workers_.emplace_back([this, id] { loop(id); });
            }
            job_ = &job;
            active_ = n;
            pending_ = n - 1;
            ++generation_;
        }
        cv_.notify_all();
        job(0);
        std::unique_lock<std::mutex> lock(mu_);
        done_.wait(lock, [this] { return pending_ == 0; });
        job_ = nullptr;
    }

private:
    void loop(int id)
    {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)> *job = nullptr;
            {
                std::unique_lock<std::mutex> lock(mu_);
                cv_.wait(lock, [&] { return stop_ || generation_ != seen; });
                if (stop_)
                    return;
                seen = generation_;
                if (id < active_)
                    job = job_;
            }
            if (job) {
                (*job)(id);
                std::lock_guard<std::mutex> lock(mu_);
                if (--pending_ == 0)
                    done_.notify_one();
            }
        }
    }
    std::mutex mu_, run_mu_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> workers_;
    const std::function<void(int)> *job_ = nullptr;
    uint64_t generation_ = 0;
    int active_ = 0, pending_ = 0;
    bool stop_ = false;
};

HostPool &host_pool()
{
    // On the heap and never destroyed: worker threads blocked on a condition variable must not
    // be joined from a static destructor at exit, and a child of fork() has no workers at all
    // -- it gets a pool of its own (the parent's object is left alone).
    static HostPool *pool = nullptr;
    static pid_t owner = 0;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!pool || owner != getpid()) {
        pool = new HostPool();
        owner = getpid();
    }
    return *pool;
}

struct RngInit {
    RngInit()
    {
        // module import seeds with the clock (python/src/module.cpp:356-359)
        timespec t;
        clock_gettime(CLOCK_REALTIME, &t);
        const unsigned int s = (unsigned int)((t.tv_nsec / 1000) * t.tv_sec);
        srand(s);
        g_rng.seed(s);
    }
} g_rng_init;

}  // namespace

extern "C" {

void trlda_seed(unsigned int seed)
{
    srand(seed);          // keep libc's own stream in step for anything else that uses it
    g_rng.seed(seed);
}

void trlda_sample_gamma(int m, int n, int k, double *out)
{
    const int64_t total = (int64_t)m * n;
    for (int64_t i = 0; i < total; ++i)
        out[i] = 0.0;
    if (total <= 0 || k <= 0)
        return;
    // out[i] = - sum_{p < k} log |u_{p, i}|, u_{p, i} the (p * total + i)-th draw of the stream
    // (utils.cpp:224-231).  Small requests: one thread, straight through the stream.
    unsigned int hw = std::thread::hardware_concurrency();
    // (the threads are persistent, host_pool(): their number is bounded by the work per thread)
    int64_t T = std::max<int64_t>(1, std::min<int64_t>({(int64_t)hw, (int64_t)64, total / 256}));
    if (total * k < (1 << 17))
        T = 1;
    if (const char *env = std::getenv("TRLDA_SAMPLE_THREADS"))   // tests: force a thread count
        T = std::max<int64_t>(1, std::min<int64_t>(std::atoi(env), total));
    if (T == 1) {
        for (int p = 0; p < k; ++p)
            for (int64_t i = 0; i < total; ++i) {
                const double u = -1.0 + 2.0 * (double)g_rng.next() / (double)2147483647;
                out[i] -= std::log(std::fabs(u));
            }
        return;
    }
    // Thread t owns the elements [t * len, min(total, (t + 1) * len)) in every pass: it starts
    // lo_t draws into the stream and, after the draws of a pass, jumps over the other threads'
    // share (total - its own length) to the next pass.  Per element the logs are added in pass
    // order, exactly as the serial loop does.
    const int64_t len = (total + T - 1) / T;
    T = (total + len - 1) / len;
    const int64_t last_len = total - (T - 1) * len;
    const JumpMatrix &hop = jump_power((uint64_t)len);                     // thread t -> t + 1
    const JumpMatrix &skip = jump_power((uint64_t)(total - len));          // pass p -> p + 1
    const JumpMatrix &skip_last = jump_power((uint64_t)(total - last_len));
    std::vector<GlibcRandom> start((size_t)T);
    {
        uint32_t w[31];
        rng_to_window(g_rng, w);
        for (int64_t t = 0; t < T; ++t) {
            window_to_rng(w, start[(size_t)t]);
            jump_apply(hop, w);
        }
    }
    GlibcRandom final_state;
    auto work = [&](int64_t t) {
        GlibcRandom g = start[(size_t)t];
        const int64_t lo = t * len, hi = std::min<int64_t>(total, lo + len);
        const JumpMatrix &sk = (t == T - 1) ? skip_last : skip;
        for (int p = 0; p < k; ++p) {
            for (int64_t i = lo; i < hi; ++i) {
                const double u = -1.0 + 2.0 * (double)g.next() / (double)2147483647;
                out[i] -= std::log(std::fabs(u));
            }
            if (p + 1 < k || t == T - 1) {
                if (p + 1 == k)
                    break;                           // the last thread ends where the stream ends
                uint32_t w[31];
                rng_to_window(g, w);
                jump_apply(sk, w);
                window_to_rng(w, g);
            }
        }
        if (t == T - 1)
            final_state = g;
    };
    host_pool().run((int)T, [&](int t) { work(t); });
    g_rng = final_state;
}

void trlda_sample_gamma_init(int m, int n, double *out)
{
    trlda_sample_gamma(m, n, 100, out);
    const int64_t total = (int64_t)m * n;
    for (int64_t i = 0; i < total; ++i)
        out[i] /= 100.;
}

// ---- device memory helpers ----------------------------------------------------

int trlda_dev_alloc(int device, size_t bytes, void **dev_out)
{
    if (!dev_out)
        return fail(TRLDA_ERR_ARG, "dev_out is NULL");
    int rc = use_device(device);
    if (rc)
        return rc;
    HIP_TRY(hipMalloc(dev_out, bytes ? bytes : 1));
    return TRLDA_OK;
}

int trlda_dev_free(int device, void *dev)
{
    int rc = use_device(device);
    if (rc)
        return rc;
    HIP_TRY(hipFree(dev));
    return TRLDA_OK;
}

int trlda_dev_upload(int device, void *dev_dst, const void *host_src, size_t bytes)
{
    int rc = use_device(device);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
    return TRLDA_OK;
}

int trlda_dev_download(int device, void *host_dst, const void *dev_src, size_t bytes)
{
    int rc = use_device(device);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
    return TRLDA_OK;
}

int trlda_dev_synchronize(int device)
{
    int rc = use_device(device);
    if (rc)
        return rc;
    HIP_TRY(hipDeviceSynchronize());
    return TRLDA_OK;
}

// ---- batches ------------------------------------------------------------------

int trlda_batch_create(trlda_batch **out, int device, int V, int B, const int32_t *indptr,
                       const int32_t *ids, const int32_t *cnts)
{
    if (!out)
        return fail(TRLDA_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (V <= 0 || B < 0 || !indptr)
        return fail(TRLDA_ERR_ARG, "bad batch dimensions");
    if (indptr[0] != 0)
        return fail(TRLDA_ERR_ARG, "indptr[0] must be 0");
    int max_n = 0;
    for (int d = 0; d < B; ++d) {
        if (indptr[d + 1] < indptr[d])
            return fail(TRLDA_ERR_ARG, "indptr must be non-decreasing");
        max_n = std::max(max_n, indptr[d + 1] - indptr[d]);
    }
    const int64_t nnz = indptr[B];
    if (nnz > 0 && (!ids || !cnts))
        return fail(TRLDA_ERR_ARG, "ids / cnts are NULL");
    for (int64_t i = 0; i < nnz; ++i)
        if (ids[i] < 0 || ids[i] >= V)
            return fail(TRLDA_ERR_WORD_ID, "word id outside [0, num_words)");

    int rc = use_device(device);
    if (rc)
        return rc;

    // word-major index: stable counting sort of the CSR positions by word id
    std::vector<int32_t> wptr((size_t)V + 1, 0), wrank((size_t)std::max<int64_t>(nnz, 1)),
        wdoc((size_t)std::max<int64_t>(nnz, 1)), order((size_t)std::max(B, 1));
    for (int64_t i = 0; i < nnz; ++i)
        ++wptr[(size_t)ids[i] + 1];
    for (int w = 0; w < V; ++w)
        wptr[(size_t)w + 1] += wptr[(size_t)w];
    {
        std::vector<int32_t> cursor(wptr.begin(), wptr.end() - 1);
        for (int d = 0; d < B; ++d)
            for (int32_t p = indptr[d]; p < indptr[d + 1]; ++p) {
                int32_t q = cursor[(size_t)ids[p]]++;
                wrank[(size_t)p] = q;
                wdoc[(size_t)q] = d;
            }
    }
    std::iota(order.begin(), order.begin() + B, 0);
    std::stable_sort(order.begin(), order.begin() + B, [&](int32_t x, int32_t y) {
        return indptr[x + 1] - indptr[x] > indptr[y + 1] - indptr[y];
    });

    trlda_batch *b = new trlda_batch();
    b->device = device; b->V = V; b->B = B; b->nnz = nnz; b->max_n = max_n;
    b->sorted_len.resize((size_t)B);
    for (int i = 0; i < B; ++i)
        b->sorted_len[(size_t)i] = indptr[order[(size_t)i] + 1] - indptr[order[(size_t)i]];
    auto up = [&](int32_t **dst, const int32_t *src, size_t count) -> int {
        int r = dev_alloc(dst, count);
        if (r)
            return r;
        if (count)
            HIP_TRY(hipMemcpy(*dst, src, count * sizeof(int32_t), hipMemcpyHostToDevice));
        return TRLDA_OK;
    };
    rc = up(&b->indptr, indptr, (size_t)B + 1);
    if (!rc) rc = up(&b->ids, ids, (size_t)nnz);
    if (!rc) rc = up(&b->cnts, cnts, (size_t)nnz);
    if (!rc) rc = up(&b->order, order.data(), (size_t)B);
    if (!rc) rc = up(&b->wrank, wrank.data(), (size_t)nnz);
    if (!rc) rc = up(&b->wptr, wptr.data(), (size_t)V + 1);
    if (!rc) rc = up(&b->wdoc, wdoc.data(), (size_t)nnz);
    {
        std::vector<int32_t> meta((size_t)std::max(B, 1) * 4, 0),
            pids((size_t)std::max(B, 1) * trlda::kRegMaxN, 0);
        for (int i = 0; i < B; ++i) {
            const int d = order[(size_t)i], p0 = indptr[d], n = indptr[d + 1] - p0;
            meta[(size_t)i * 4] = d;
            meta[(size_t)i * 4 + 1] = n;
            meta[(size_t)i * 4 + 2] = p0;
            // words past the document repeat its last id (rows that exist; masked by length)
            for (int j = 0; j < trlda::kRegMaxN; ++j)
                pids[(size_t)i * trlda::kRegMaxN + j] = n > 0 ? ids[p0 + std::min(j, n - 1)] : 0;
        }
        if (!rc) rc = up(&b->pad_meta, meta.data(), (size_t)B * 4);
        if (!rc) rc = up(&b->pad_ids, pids.data(), (size_t)B * trlda::kRegMaxN);
    }
    {
        std::vector<int32_t> active;
        for (int w = 0; w < V; ++w)
            if (wptr[(size_t)w + 1] > wptr[(size_t)w])
                active.push_back(w);
        b->n_active = (int)active.size();
        if (!rc) rc = up(&b->active, active.data(), active.size());
        std::vector<int32_t> longw;
        for (int w : active)
            if (wptr[(size_t)w + 1] - wptr[(size_t)w] > trlda::kLongWord)
                longw.push_back(w);
        b->n_long = (int)longw.size();
        if (!rc) rc = up(&b->long_words, longw.data(), longw.size());
    }
    if (rc) {
        trlda_batch_destroy(b);
        return rc;
    }
    *out = b;
    return TRLDA_OK;
}

int trlda_batch_destroy(trlda_batch *b)
{
    if (!b)
        return TRLDA_OK;
    if (hipSetDevice(b->device) == hipSuccess) {
        (void)hipFree(b->indptr); (void)hipFree(b->ids); (void)hipFree(b->cnts); (void)hipFree(b->order);
        (void)hipFree(b->wrank); (void)hipFree(b->wptr); (void)hipFree(b->wdoc);
        (void)hipFree(b->active); (void)hipFree(b->long_words);
        (void)hipFree(b->pad_meta); (void)hipFree(b->pad_ids);
    }
    delete b;
    return TRLDA_OK;
}

int trlda_batch_num_docs(const trlda_batch *b) { return b ? b->B : 0; }
int64_t trlda_batch_nnz(const trlda_batch *b) { return b ? b->nnz : 0; }
int trlda_batch_max_doc_len(const trlda_batch *b) { return b ? b->max_n : 0; }

// ---- model --------------------------------------------------------------------

int trlda_model_create(trlda_model **out, int device, int K, int V)
{
    if (!out)
        return fail(TRLDA_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (K <= 0 || V <= 0)
        return fail(TRLDA_ERR_ARG, "num_topics and num_words must be positive");
    int rc = use_device(device);
    if (rc)
        return rc;
    trlda_model *m = new trlda_model();
    m->device = device; m->K = K; m->V = V;
    size_t KV = (size_t)K * V;
    rc = dev_alloc(&m->lambda, KV);
    if (!rc) rc = dev_alloc(&m->eeb, KV);
    if (!rc) rc = dev_alloc(&m->alpha, (size_t)K);
    if (!rc) rc = dev_alloc(&m->psi_sum, 3 * (size_t)K);   // psi(row sums), the row sums, exp(-psi)
    if (!rc) rc = dev_alloc(&m->partial, (size_t)kMaxRowsumBlocks * K);
    if (!rc) rc = dev_alloc(&m->counter, 1);
    if (rc) {
        trlda_model_destroy(m);
        return rc;
    }
    HIP_TRY(hipMemset(m->counter, 0, sizeof(unsigned int)));
    // columns of words no batch has touched yet are never read for their value, but the
    // atomic-mode finish multiplies them by 0: keep them finite
    HIP_TRY(hipMemset(m->eeb, 0, KV * sizeof(double)));
    *out = m;
    return TRLDA_OK;
}

int trlda_model_destroy(trlda_model *m)
{
    if (!m)
        return TRLDA_OK;
    if (hipSetDevice(m->device) == hipSuccess) {
        (void)hipStreamSynchronize(m->stream);
        (void)hipFree(m->lambda); (void)hipFree(m->alpha); (void)hipFree(m->eeb); (void)hipFree(m->psi_sum);
        (void)hipFree(m->partial); (void)hipFree(m->counter); (void)hipFree(m->epg); (void)hipFree(m->tw_csr);
        (void)hipFree(m->tw_word); (void)hipFree(m->lambda_prime); (void)hipFree(m->sstats); (void)hipFree(m->gamma);
        (void)hipFree(m->wordcounts);
        for (auto &e : m->ev_pool)
            (void)hipEventDestroy(e);

    }
    delete m;
    return TRLDA_OK;
}

int trlda_model_set_stream(trlda_model *m, void *hip_stream)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(m->stream));
    m->stream = static_cast<hipStream_t>(hip_stream);
    return TRLDA_OK;
}

int trlda_model_set_sstats_mode(trlda_model *m, int mode)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "model is NULL");
    if (mode != TRLDA_SSTATS_SEGMENTED && mode != TRLDA_SSTATS_ATOMIC)
        return fail(TRLDA_ERR_ARG, "unknown sstats mode");
    m->sstats_mode = mode;
    return TRLDA_OK;
}

int trlda_model_set_dense_preamble(trlda_model *m, int dense)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "model is NULL");
    m->dense_preamble = dense != 0;
    return TRLDA_OK;
}

const char *trlda_model_last_doc_kernel(const trlda_model *m) { return m ? m->last_doc_kernel : ""; }

int trlda_model_last_preamble_fused(const trlda_model *m) { return m && m->last_preamble_fused; }

int trlda_model_set_split_preamble(trlda_model *m, int split)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "null model");
    m->split_preamble = split != 0;
    return TRLDA_OK;
}

int trlda_model_set_doc_kernel(trlda_model *m, int kind)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "null model");
    if (kind != TRLDA_DOCS_AUTO && kind != TRLDA_DOCS_GENERAL && kind != TRLDA_DOCS_WIDE)
        return fail(TRLDA_ERR_ARG, "doc_kernel must be TRLDA_DOCS_AUTO, _LDS or _WIDE");
    m->doc_kernel = kind;
    return TRLDA_OK;
}

int trlda_model_set_doc_threads(trlda_model *m, int threads)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "model is NULL");
    if (threads != 0 && threads != 64 && threads != 128 && threads != 256 && threads != 512 &&
        threads != 1024)
        return fail(TRLDA_ERR_ARG, "doc_threads must be 0, 64, 128, 256, 512 or 1024");
    m->doc_threads = threads;
    return TRLDA_OK;
}

int trlda_model_synchronize(trlda_model *m)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(m->stream));
    return TRLDA_OK;
}

int trlda_model_set_lambda(trlda_model *m, const double *host_lambda)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!host_lambda)
        return fail(TRLDA_ERR_ARG, "lambda is NULL");
    HIP_TRY(hipMemcpyAsync(m->lambda, host_lambda, (size_t)m->K * m->V * sizeof(double),
                           hipMemcpyHostToDevice, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return TRLDA_OK;
}

int trlda_model_get_lambda(trlda_model *m, double *host_lambda)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!host_lambda)
        return fail(TRLDA_ERR_ARG, "lambda is NULL");
    HIP_TRY(hipMemcpyAsync(host_lambda, m->lambda, (size_t)m->K * m->V * sizeof(double),
                           hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return TRLDA_OK;
}

int trlda_model_set_alpha(trlda_model *m, const double *host_alpha)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!host_alpha)
        return fail(TRLDA_ERR_ARG, "alpha is NULL");
    for (int k = 0; k < m->K; ++k)
        if (host_alpha[k] < 0.)
            return fail(TRLDA_ERR_VALUE, "Alpha should not be negative.");  // lda.h:147-159
    HIP_TRY(hipMemcpyAsync(m->alpha, host_alpha, (size_t)m->K * sizeof(double),
                           hipMemcpyHostToDevice, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return TRLDA_OK;
}

void *trlda_model_lambda_dev(trlda_model *m) { return m ? m->lambda : nullptr; }

int trlda_model_get_sstats(trlda_model *m, double *host_sstats)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!host_sstats)
        return fail(TRLDA_ERR_ARG, "sstats is NULL");
    if (!m->sstats)
        return fail(TRLDA_ERR_ARG, "no E-step has run through this model's own workspace yet");
    HIP_TRY(hipMemcpyAsync(host_sstats, m->sstats, (size_t)m->K * m->V * sizeof(double),
                           hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return TRLDA_OK;
}

int trlda_model_estep(trlda_model *m, const trlda_batch *b, double *gamma_dev, double *sstats_dev,
                      int max_iter, double threshold, int32_t *iters_dev)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!b || !sstats_dev || (b->B > 0 && !gamma_dev))
        return fail(TRLDA_ERR_ARG, "NULL batch / gamma / sstats");
    return estep_device(m, b, gamma_dev, sstats_dev, max_iter, threshold, iters_dev);
}

int trlda_model_estep_io(trlda_model *m, const trlda_batch *b, const double *gamma0_dev,
                         double *gamma_dev, double *sstats_dev, int max_iter, double threshold,
                         int32_t *iters_dev)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!b || !sstats_dev || (b->B > 0 && (!gamma_dev || !gamma0_dev)))
        return fail(TRLDA_ERR_ARG, "NULL batch / gamma / sstats");
    return estep_device(m, b, gamma_dev, sstats_dev, max_iter, threshold, iters_dev, gamma0_dev);
}

int trlda_model_estep_host(trlda_model *m, const trlda_batch *b, double *gamma, double *sstats,
                           int max_iter, double threshold, int32_t *iters_out)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!b || !sstats || (b->B > 0 && !gamma))
        return fail(TRLDA_ERR_ARG, "NULL batch / gamma / sstats");
    rc = ensure_update_workspace(m, b->B);
    if (rc)
        return rc;
    const size_t gbytes = (size_t)m->K * b->B * sizeof(double);
    const size_t sbytes = (size_t)m->K * m->V * sizeof(double);
    int32_t *iters_dev = nullptr;
    if (iters_out) {
        rc = dev_alloc(&iters_dev, (size_t)b->B);
        if (rc)
            return rc;
    }
    if (gbytes)
        HIP_TRY(hipMemcpyAsync(m->gamma, gamma, gbytes, hipMemcpyHostToDevice, m->stream));
    rc = estep_device(m, b, m->gamma, m->sstats, max_iter, threshold, iters_dev);
    if (!rc) {
        if (gbytes)
            HIP_TRY(hipMemcpyAsync(gamma, m->gamma, gbytes, hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipMemcpyAsync(sstats, m->sstats, sbytes, hipMemcpyDeviceToHost, m->stream));
        if (iters_out && b->B)
            HIP_TRY(hipMemcpyAsync(iters_out, iters_dev, (size_t)b->B * sizeof(int32_t),
                                   hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipStreamSynchronize(m->stream));
    }
    if (iters_dev)
        (void)hipFree(iters_dev);
    return rc;
}

// LDA::lowerBound, src/lda.cpp:297-360 (see csrc/elbo_kernels.h)
int trlda_model_lower_bound(trlda_model *m, const trlda_batch *b, double *gamma, double eta,
                            double factor, int max_iter, double threshold, double *bound_out)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!b || !bound_out || (b->B > 0 && !gamma))
        return fail(TRLDA_ERR_ARG, "NULL batch / gamma / bound");
    if (b->B <= 0)
        return fail(TRLDA_ERR_ARG, "the lower bound needs at least one document");
    rc = ensure_update_workspace(m, b->B);
    if (rc)
        return rc;
    const int K = m->K, V = m->V, B = b->B;
    const size_t KV = (size_t)K * V;
    const size_t gbytes = (size_t)K * B * sizeof(double);
    HIP_TRY(hipMemcpyAsync(m->gamma, gamma, gbytes, hipMemcpyHostToDevice, m->stream));
    rc = estep_device(m, b, m->gamma, m->sstats, max_iter, threshold, nullptr);   // :309
    if (rc)
        return rc;
    const int G = (int)std::min<size_t>((KV + kDenseThreads - 1) / kDenseThreads, 1024);
    double *out = nullptr;
    rc = dev_alloc(&out, 2 * (size_t)G + 2 * (size_t)B);
    if (rc)
        return rc;
    hipLaunchKernelGGL(trlda::elbo_dense_kernel<kDenseThreads>, dim3(G), dim3(kDenseThreads), 0,
                       m->stream, K, KV, eta, factor, m->lambda, m->psi_sum, m->sstats, out);
    const size_t lds = ((size_t)K + 4 * (kDenseThreads / trlda::kWave)) * sizeof(double);
    hipLaunchKernelGGL(trlda::elbo_docs_kernel<kDenseThreads>, dim3(B), dim3(kDenseThreads), lds,
                       m->stream, K, b->indptr, b->ids, b->cnts, m->lambda, m->psi_sum, m->alpha,
                       m->gamma, out + 2 * (size_t)G);
    std::vector<double> h(2 * (size_t)G + 2 * (size_t)B), lam_sum((size_t)K), alpha_h((size_t)K);
    hipError_t e1 = hipMemcpyAsync(h.data(), out, h.size() * sizeof(double), hipMemcpyDeviceToHost,
                                   m->stream);
    hipError_t e2 = hipMemcpyAsync(gamma, m->gamma, gbytes, hipMemcpyDeviceToHost, m->stream);
    hipError_t e3 = hipMemcpyAsync(alpha_h.data(), m->alpha, (size_t)K * sizeof(double),
                                   hipMemcpyDeviceToHost, m->stream);
    hipError_t e4 = hipStreamSynchronize(m->stream);
    (void)hipFree(out);
    HIP_TRY(e1); HIP_TRY(e2); HIP_TRY(e3); HIP_TRY(e4);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(lam_sum.data(), m->psi_sum + K, (size_t)K * sizeof(double),
                      hipMemcpyDeviceToHost));
    double pw_pb = 0.0, lg_lambda = 0.0, pz = 0.0, ptheta = 0.0;
    for (int g = 0; g < G; ++g) {
        pw_pb += h[2 * (size_t)g];
        lg_lambda += h[2 * (size_t)g + 1];
    }
    for (int d = 0; d < B; ++d) {
        pz += h[2 * (size_t)G + 2 * (size_t)d];
        ptheta += h[2 * (size_t)G + 2 * (size_t)d + 1];
    }
    double alpha_sum = 0.0, lg_alpha = 0.0, lg_lambda_sum = 0.0;
    for (int k = 0; k < K; ++k) {
        alpha_sum += alpha_h[(size_t)k];
        lg_alpha += std::lgamma(alpha_h[(size_t)k]);
        lg_lambda_sum += std::lgamma(lam_sum[(size_t)k]);
    }
    ptheta += (std::lgamma(alpha_sum) - lg_alpha) * B;                     // :355
    pw_pb += K * std::lgamma(V * eta) - lg_lambda_sum;                     // :356
    pw_pb -= (double)K * V * std::lgamma(eta) - lg_lambda;                 // :357
    *bound_out = pw_pb + factor * pz + factor * ptheta;                    // :359
    return TRLDA_OK;
}

int trlda_model_blend(trlda_model *m, const double *lambda_prime_dev, const double *sstats_dev,
                      double rho, double eta, double scale)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!lambda_prime_dev || !sstats_dev)
        return fail(TRLDA_ERR_ARG, "NULL lambda_prime / sstats");
    return blend_device(m, lambda_prime_dev, sstats_dev, rho, eta, scale);
}

int trlda_model_tr_init(trlda_model *m, const trlda_batch *b, const double *lambda_prime_dev,
                        double rho, double eta, int num_documents)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!b || !lambda_prime_dev || b->B <= 0)
        return fail(TRLDA_ERR_ARG, "NULL or empty batch / lambda_prime");
    return tr_init_device(m, b, lambda_prime_dev, rho, eta, num_documents);
}

int trlda_model_wordcounts(trlda_model *m, const trlda_batch *b, double *wordcounts_dev)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!b || !wordcounts_dev)
        return fail(TRLDA_ERR_ARG, "NULL batch / wordcounts");
    if (b->V != m->V)
        return fail(TRLDA_ERR_SHAPE, "batch was created for a different vocabulary size");
    return wordcounts_device(m, b, wordcounts_dev);
}

int trlda_model_tr_init_wc(trlda_model *m, const double *wordcounts_dev,
                           const double *lambda_prime_dev, double rho, double eta, double coef)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!wordcounts_dev || !lambda_prime_dev)
        return fail(TRLDA_ERR_ARG, "NULL wordcounts / lambda_prime");
    return tr_init_wc_device(m, wordcounts_dev, lambda_prime_dev, rho, eta, coef);
}

int trlda_model_copy_lambda(trlda_model *m, double *dst_dev)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!dst_dev)
        return fail(TRLDA_ERR_ARG, "NULL destination");
    HIP_TRY(hipMemcpyAsync(dst_dev, m->lambda, (size_t)m->K * m->V * sizeof(double),
                           hipMemcpyDeviceToDevice, m->stream));
    return TRLDA_OK;
}

int trlda_model_online_update(trlda_model *m, const trlda_batch *b, int num_documents, double eta,
                              int max_iter_tr, int max_iter_inference, double kappa, double tau,
                              double rho, int init_gamma, int update_lambda, double threshold,
                              int *update_count, double *rho_out, double *gamma_out)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!b || !update_count || !rho_out)
        return fail(TRLDA_ERR_ARG, "NULL batch / update_count / rho_out");
    if (b->B == 0) {                                         // onlinelda.cpp:54-56
        *rho_out = 1.0;
        return TRLDA_OK;
    }
    if (rho < 0.)                                            // onlinelda.cpp:59-66
        rho = std::pow(tau + (double)*update_count, -kappa);
    *rho_out = rho;

    if (update_lambda) {
        rc = ensure_update_workspace(m, b->B);
        if (rc)
            return rc;
        const int K = m->K, B = b->B;
        const size_t KV = (size_t)K * m->V;
        const size_t gbytes = (size_t)K * B * sizeof(double);
        std::vector<double> gamma0((size_t)K * B);
        const double scale = (double)num_documents / (double)B;

        HIP_TRY(hipMemcpyAsync(m->lambda_prime, m->lambda, KV * sizeof(double),
                               hipMemcpyDeviceToDevice, m->stream));
        auto fresh_gamma = [&]() -> int {                    // lda.cpp:135
            trlda_sample_gamma_init(K, B, gamma0.data());
            // gamma0 is reused by the next draw: finish the upload before returning
            HIP_TRY(hipMemcpyAsync(m->gamma, gamma0.data(), gbytes, hipMemcpyHostToDevice,
                                   m->stream));
            HIP_TRY(hipStreamSynchronize(m->stream));
            return TRLDA_OK;
        };
        if (max_iter_tr > 0) {
            rc = tr_init_device(m, b, m->lambda_prime, rho, eta, num_documents);
            for (int i = 0; !rc && i < max_iter_tr; ++i) {   // onlinelda.cpp:89-101
                if (!(i > 0 && init_gamma))
                    rc = fresh_gamma();
                if (!rc)
                    rc = estep_device(m, b, m->gamma, m->sstats, max_iter_inference, threshold,
                                      nullptr);
                if (!rc)
                    rc = blend_device(m, m->lambda_prime, m->sstats, rho, eta, scale);
            }
        } else {                                             // onlinelda.cpp:103-109
            rc = fresh_gamma();
            if (!rc)
                rc = estep_device(m, b, m->gamma, m->sstats, max_iter_inference, threshold,
                                  nullptr);
            if (!rc)
                rc = blend_device(m, m->lambda_prime, m->sstats, rho, eta, scale);
        }
        if (rc)
            return rc;
        if (gamma_out)
            HIP_TRY(hipMemcpyAsync(gamma_out, m->gamma, gbytes, hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipStreamSynchronize(m->stream));
    }
    ++*update_count;                                         // onlinelda.cpp:177
    return TRLDA_OK;
}

int trlda_model_batch_update(trlda_model *m, const trlda_batch *b, double eta, int max_epochs,
                             int max_iter_inference, int update_lambda, double threshold,
                             double *gamma_out)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!b)
        return fail(TRLDA_ERR_ARG, "NULL batch");
    if (b->B == 0)                                           // batchlda.cpp:44-46
        return TRLDA_OK;
    rc = ensure_update_workspace(m, b->B);
    if (rc)
        return rc;
    const int K = m->K, B = b->B;
    const size_t gbytes = (size_t)K * B * sizeof(double);
    std::vector<double> gamma0((size_t)K * B);
    for (int epoch = 0; epoch < max_epochs; ++epoch) {       // batchlda.cpp:48-61
        if (!update_lambda)
            continue;
        trlda_sample_gamma_init(K, B, gamma0.data());
        HIP_TRY(hipMemcpyAsync(m->gamma, gamma0.data(), gbytes, hipMemcpyHostToDevice, m->stream));
        HIP_TRY(hipStreamSynchronize(m->stream));
        rc = estep_device(m, b, m->gamma, m->sstats, max_iter_inference, threshold, nullptr);
        if (rc)
            return rc;
        // lambda = eta + sstats  ==  blend with rho = 1, scale = 1 (lambda' term is * 0)
        rc = blend_device(m, m->lambda, m->sstats, 1.0, eta, 1.0);
        if (rc)
            return rc;
    }
    if (gamma_out && update_lambda && max_epochs > 0)
        HIP_TRY(hipMemcpyAsync(gamma_out, m->gamma, gbytes, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return TRLDA_OK;
}

int trlda_model_cumulative_update(trlda_model *m, const trlda_batch *b, int max_epochs,
                                  int max_iter_inference, int update_lambda, double threshold,
                                  double *gamma_out)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!b)
        return fail(TRLDA_ERR_ARG, "NULL batch");
    if (b->B == 0)                                           // cumulativelda.cpp:50-52
        return TRLDA_OK;
    rc = ensure_update_workspace(m, b->B);
    if (rc)
        return rc;
    const int K = m->K, B = b->B;
    const size_t KV = (size_t)K * m->V;
    const size_t gbytes = (size_t)K * B * sizeof(double);
    // lambdaPrime = mLambda; mLambda = sampleGamma(K, V, 100) / 100   cumulativelda.cpp:57-60
    HIP_TRY(hipMemcpyAsync(m->lambda_prime, m->lambda, KV * sizeof(double),
                           hipMemcpyDeviceToDevice, m->stream));
    {
        std::vector<double> lam0(KV);
        trlda_sample_gamma_init(K, m->V, lam0.data());
        HIP_TRY(hipMemcpyAsync(m->lambda, lam0.data(), KV * sizeof(double), hipMemcpyHostToDevice,
                               m->stream));
        HIP_TRY(hipStreamSynchronize(m->stream));
    }
    std::vector<double> gamma0((size_t)K * B);
    bool ran = false;
    if (update_lambda) {
        for (int epoch = 0; epoch < max_epochs; ++epoch) {    // cumulativelda.cpp:62-71
            trlda_sample_gamma_init(K, B, gamma0.data());
            HIP_TRY(hipMemcpyAsync(m->gamma, gamma0.data(), gbytes, hipMemcpyHostToDevice,
                                   m->stream));
            HIP_TRY(hipStreamSynchronize(m->stream));
            rc = estep_device(m, b, m->gamma, m->sstats, max_iter_inference, threshold, nullptr);
            if (rc)
                return rc;
            size_t blocks = (KV + kDenseThreads - 1) / kDenseThreads;
            int G = (int)std::min<size_t>(blocks, 256 * 8);
            hipLaunchKernelGGL(trlda::accumulate_kernel<kDenseThreads>, dim3(G), dim3(kDenseThreads),
                               0, m->stream, KV, m->lambda_prime, m->sstats, m->lambda);
            HIP_TRY(hipGetLastError());
            ran = true;
        }
    }
    if (gamma_out && ran)
        HIP_TRY(hipMemcpyAsync(gamma_out, m->gamma, gbytes, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return TRLDA_OK;
}

// ---- one-shot host-pointer entry points ---------------------------------------

int trlda_estep(int K, int V, int B, const int32_t *indptr, const int32_t *ids,
                const int32_t *cnts, const double *lambda, const double *alpha, double *gamma,
                double *sstats, int max_iter, double threshold, int32_t *iters_out, int device)
{
    if (!lambda || !alpha || !sstats)
        return fail(TRLDA_ERR_ARG, "NULL lambda / alpha / sstats");
    trlda_model *m = nullptr;
    trlda_batch *b = nullptr;
    int rc = trlda_model_create(&m, device, K, V);
    if (!rc) rc = trlda_batch_create(&b, device, V, B, indptr, ids, cnts);
    if (!rc) rc = trlda_model_set_lambda(m, lambda);
    if (!rc) rc = trlda_model_set_alpha(m, alpha);
    if (!rc) rc = trlda_model_estep_host(m, b, gamma, sstats, max_iter, threshold, iters_out);
    trlda_batch_destroy(b);
    trlda_model_destroy(m);
    return rc;
}

int trlda_mstep_blend(int K, int V, double rho, double eta, double scale,
                      const double *lambda_prime, const double *sstats, double *lambda_out,
                      int device)
{
    if (!lambda_prime || !sstats || !lambda_out)
        return fail(TRLDA_ERR_ARG, "NULL lambda_prime / sstats / lambda_out");
    trlda_model *m = nullptr;
    int rc = trlda_model_create(&m, device, K, V);
    if (!rc) rc = ensure_update_workspace(m, 0);
    const size_t bytes = (size_t)K * V * sizeof(double);
    if (!rc) {
        hipError_t e1 = hipMemcpy(m->lambda_prime, lambda_prime, bytes, hipMemcpyHostToDevice);
        hipError_t e2 = hipMemcpy(m->sstats, sstats, bytes, hipMemcpyHostToDevice);
        if (e1 != hipSuccess || e2 != hipSuccess)
            rc = fail(TRLDA_ERR_HIP, "upload failed");
    }
    if (!rc) rc = blend_device(m, m->lambda_prime, m->sstats, rho, eta, scale);
    if (!rc) rc = trlda_model_get_lambda(m, lambda_out);
    trlda_model_destroy(m);
    return rc;
}

int trlda_tr_init(int K, int V, int B, int num_documents, double rho, double eta,
                  const int32_t *indptr, const int32_t *ids, const int32_t *cnts,
                  const double *lambda_prime, double *lambda_out, int device)
{
    if (!lambda_prime || !lambda_out)
        return fail(TRLDA_ERR_ARG, "NULL lambda_prime / lambda_out");
    if (B <= 0)
        return fail(TRLDA_ERR_ARG, "empty batch");
    trlda_model *m = nullptr;
    trlda_batch *b = nullptr;
    int rc = trlda_model_create(&m, device, K, V);
    if (!rc) rc = trlda_batch_create(&b, device, V, B, indptr, ids, cnts);
    if (!rc) rc = ensure_update_workspace(m, B);
    if (!rc && hipMemcpy(m->lambda_prime, lambda_prime, (size_t)K * V * sizeof(double),
                         hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(TRLDA_ERR_HIP, "upload failed");
    if (!rc) rc = tr_init_device(m, b, m->lambda_prime, rho, eta, num_documents);
    if (!rc) rc = trlda_model_get_lambda(m, lambda_out);
    trlda_batch_destroy(b);
    trlda_model_destroy(m);
    return rc;
}

#ifdef TRLDA_STAMPS
// diagnostic build only: copy out and clear the per-block segment cycle sums
extern "C" int trlda_debug_read_stamps(unsigned long long *host, int blocks)
{
    if (!g_stamp_buf)
        return -1;
    if (hipMemcpy(host, g_stamp_buf, (size_t)blocks * 64, hipMemcpyDeviceToHost) != hipSuccess)
        return -2;
    (void)hipMemset(g_stamp_buf, 0, 65536 * 8 * 8);
    return 0;
}
#endif

// ---- device special functions (test hook) -----------------------------------------------

int trlda_debug_digamma(int device, int n, double c, const double *x, double *psi, double *epsi,
                        double *epsi_lean, double *eminus)
{
    int rc = use_device(device);
    if (rc)
        return rc;
    if (n <= 0 || !x || !psi || !epsi || !epsi_lean || !eminus)
        return fail(TRLDA_ERR_ARG, "bad digamma table arguments");
    double *d = nullptr;
    rc = dev_alloc(&d, (size_t)n * 5);
    if (rc)
        return rc;
    const size_t bytes = (size_t)n * sizeof(double);
    HIP_TRY(hipMemcpy(d, x, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(trlda::digamma_table_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, n,
                       c, d, d + n, d + 2 * (size_t)n, d + 3 * (size_t)n, d + 4 * (size_t)n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(psi, d + n, bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(epsi, d + 2 * (size_t)n, bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(epsi_lean, d + 3 * (size_t)n, bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(eminus, d + 4 * (size_t)n, bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipFree(d));
    return TRLDA_OK;
}

int trlda_debug_fold16(int device, const double *in, double *out16, double *out4, double *out2)
{
    int rc = use_device(device);
    if (rc)
        return rc;
    if (!in || !out16 || !out4 || !out2)
        return fail(TRLDA_ERR_ARG, "bad fold table arguments");
    double *d = nullptr;
    rc = dev_alloc(&d, 64 * 16 + 3 * 64);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpy(d, in, 64 * 16 * sizeof(double), hipMemcpyHostToDevice));
    double *o = d + 64 * 16;
    hipLaunchKernelGGL(trlda::debug_fold16_kernel, dim3(1), dim3(64), 0, nullptr, d, o, o + 64,
                       o + 128);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out16, o, 64 * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out4, o + 64, 64 * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out2, o + 128, 64 * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipFree(d));
    return TRLDA_OK;
}

// ---- measurement ----------------------------------------------------------------

int trlda_model_set_timing(trlda_model *m, int enabled)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "model is NULL");
    if (m->timing)
        collect_timing(m);
    m->timing = enabled != 0;
    for (int i = 0; i < 5; ++i) {
        m->usec_sum[i] = 0;
        m->usec_cnt[i] = 0;
    }
    return TRLDA_OK;
}

int trlda_model_get_timing(trlda_model *m, int which, double *usec_sum, int64_t *count)
{
    if (!m || which < 0 || which > 4 || !usec_sum || !count)
        return fail(TRLDA_ERR_ARG, "bad timing query");
    collect_timing(m);
    *usec_sum = m->usec_sum[which];
    *count = m->usec_cnt[which];
    return TRLDA_OK;
}

}  // extern "C"
