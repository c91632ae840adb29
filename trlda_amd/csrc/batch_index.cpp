// batch_index.cpp -- a mini-batch's word-major index, built on the host (host only; see
// batch_index.h for what it is and host_common.h for why these files need no HIP).
#include "batch_index.h"

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>

#include "../../include/trlda_hip.h"
#include "host_common.h"
#include "index_params.h"

namespace trlda_host {

namespace {

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// documents of more than kSplitMinN words take several workgroups (segments)
inline int segments_of(int n)
{
    if (n <= trlda::kSplitMinN)
        return 1;
    const int c = (n + trlda::kSplitSegN - 1) / trlda::kSplitSegN;
    return c <= trlda::kSplitMaxSeg ? c : 1;
}

}  // namespace

int batch_index_plan(int V, int B, const int32_t *indptr, const int32_t *ids, const int32_t *cnts,
                     BatchIndex *x)
{
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
    x->V = V; x->B = B; x->max_n = max_n; x->nnz = nnz;

    // word-major segment offsets first: they give the number of active and long words, i.e.
    // the layout of the allocation
    std::vector<int32_t> &wptr = x->wptr;
    wptr.assign((size_t)V + 1, 0);
    for (int64_t i = 0; i < nnz; ++i)
        ++wptr[(size_t)ids[i] + 1];
    int n_active = 0, n_long = 0;
    int long_len = trlda::kLongWord;
    {
        // over[i]: words with more than kLongWord << i entries
        constexpr int kLevels = 16;
        int over[kLevels] = {0};
        for (int w = 0; w < V; ++w) {
            const int len = wptr[(size_t)w + 1];
            n_active += len > 0;
            for (int i = 0; i < kLevels && len > (trlda::kLongWord << i); ++i)
                ++over[i];
            wptr[(size_t)w + 1] += wptr[(size_t)w];
        }
        int level = 0;
        while (level + 1 < kLevels && over[level] > trlda::kLongWordsTarget)
            ++level;
        long_len = trlda::kLongWord << level;
        n_long = over[level];
        // (the longest lists are cut into segments, estep_kernels.h: the one-wave range stays short)
        if (long_len > trlda::kOneWaveMax) {
            long_len = trlda::kOneWaveMax;
            n_long = 0;
            for (int w = 0; w < V; ++w)
                n_long += wptr[(size_t)w + 1] - wptr[(size_t)w] > long_len;
        }
    }
    // the longest lists as segment tasks (estep_kernels.h, VeryLongArgs): segment length per batch
    int n_vl = 0, n_vl_tasks = 0, seg_len = trlda::kSegMin;
    {
        long long heavy = 0;
        for (int w = 0; w < V; ++w) {
            const int len = wptr[(size_t)w + 1] - wptr[(size_t)w];
            heavy += len > trlda::kSegMin ? len : 0;
        }
        static const long long seg_tasks = std::getenv("TRLDA_SEG_TASKS") ? std::atoll(std::getenv("TRLDA_SEG_TASKS"))
                                                                          : (long long)trlda::kSegTasks;
        while (seg_len < trlda::kSegMax && heavy / seg_len > seg_tasks)
            seg_len *= 2;
    }
    for (int w = 0; w < V; ++w) {
        const int len = wptr[(size_t)w + 1] - wptr[(size_t)w];
        if (len > seg_len) {
            ++n_vl;
            n_vl_tasks += (len + seg_len - 1) / seg_len;
        }
    }
    int n_wg = 0, n_xrows = 0;
    for (int d = 0; d < B; ++d) {
        const int c = segments_of(indptr[d + 1] - indptr[d]);
        n_wg += c;
        n_xrows += c > 1 ? c : 0;
    }
    if (n_xrows == 0)
        n_wg = 0;                                    // no split document: no second layout
    x->n_active = n_active; x->n_long = n_long; x->long_len = long_len;
    x->n_vl = n_vl; x->n_vl_tasks = n_vl_tasks; x->seg_len = seg_len;
    x->n_wg = n_wg; x->n_xrows = n_xrows;

    // layout (bytes, 256-aligned sections)
    size_t off = 0;
    auto section = [&](size_t bytes) {
        const size_t at = off;
        off = align256(off + std::max<size_t>(bytes, 4));
        return at;
    };
    const size_t nz = (size_t)nnz, Bz = (size_t)B;
    x->o_indptr = section((Bz + 1) * 4); x->o_ids = section(nz * 4); x->o_cnts = section(nz * 4);
    x->o_order = section(Bz * 4); x->o_wrank = section(nz * 4);
    x->o_wptr = section(((size_t)V + 1) * 4); x->o_wdoc = section(nz * 4);
    x->o_meta = section(Bz * 16); x->o_pids = section(Bz * trlda::kRegMaxN * 4);
    x->o_smeta = section((size_t)n_wg * 32); x->o_spids = section((size_t)n_wg * trlda::kRegMaxN * 4);
    x->o_active = section((size_t)n_active * 4); x->o_long = section((size_t)n_long * 4);
    x->o_flag = section((size_t)V); x->o_wc32 = section((size_t)V * 4);
    x->o_mdesc = section((size_t)n_active * 16); x->o_vlw = section((size_t)n_vl * 16);
    x->o_vlt = section((size_t)n_vl_tasks * 16); x->o_vltt = section((size_t)n_vl_tasks * 16);
    x->total = off;
    return TRLDA_OK;
}

void batch_index_fill(BatchIndex *x, const int32_t *indptr, const int32_t *ids, const int32_t *cnts, int cus,
                      char *h)
{
    const int V = x->V, B = x->B;
    const size_t nz = (size_t)x->nnz, Bz = (size_t)B;
    const int n_active = x->n_active, n_long = x->n_long, long_len = x->long_len;
    const int n_vl = x->n_vl, seg_len = x->seg_len, n_wg = x->n_wg;
    const std::vector<int32_t> &wptr = x->wptr;
    auto I = [&](size_t o) { return reinterpret_cast<int32_t *>(h + o); };

    std::memcpy(I(x->o_indptr), indptr, (Bz + 1) * 4);
    if (nz) {
        std::memcpy(I(x->o_ids), ids, nz * 4);
        std::memcpy(I(x->o_cnts), cnts, nz * 4);
    }
    std::memcpy(I(x->o_wptr), wptr.data(), ((size_t)V + 1) * 4);
    // stable counting sort of the CSR positions by word id, and the words' count sums
    bool wc32_ok = true, cnts_nonneg = true;
    {
        int32_t *wrank = I(x->o_wrank), *wdoc = I(x->o_wdoc), *wc32 = I(x->o_wc32);
        std::vector<int32_t> cursor(wptr.begin(), wptr.end() - 1);
        std::vector<int64_t> wsum((size_t)V, 0);
        for (int d = 0; d < B; ++d)
            for (int32_t p = indptr[d]; p < indptr[d + 1]; ++p) {
                const int32_t q = cursor[(size_t)ids[p]]++;
                wrank[p] = q;
                wdoc[q] = d;
                wsum[(size_t)ids[p]] += cnts[p];
                cnts_nonneg = cnts_nonneg && cnts[p] >= 0;
            }
        for (int w = 0; w < V; ++w) {
            wc32_ok = wc32_ok && wsum[(size_t)w] >= INT32_MIN && wsum[(size_t)w] <= INT32_MAX;
            wc32[w] = (int32_t)wsum[(size_t)w];
        }
    }
    x->wc32_ok = wc32_ok;
    x->cnts_nonneg = cnts_nonneg;
    int32_t *order = I(x->o_order);
    std::iota(order, order + B, 0);
    std::stable_sort(order, order + B, [&](int32_t a, int32_t b) {
        return indptr[a + 1] - indptr[a] > indptr[b + 1] - indptr[b];
    });

    x->sorted_len.resize(Bz);
    x->indptr_host.assign(indptr, indptr + Bz + 1);
    {
        int32_t *meta = I(x->o_meta), *pids = I(x->o_pids);
        for (int i = 0; i < B; ++i) {
            const int d = order[i], p0 = indptr[d], n = indptr[d + 1] - p0;
            x->sorted_len[(size_t)i] = n;
            meta[(size_t)i * 4] = d;
            meta[(size_t)i * 4 + 1] = n;
            meta[(size_t)i * 4 + 2] = p0;
            meta[(size_t)i * 4 + 3] = 0;
            // words past the document repeat its last id (rows that exist; masked by length)
            int32_t *row = pids + (size_t)i * trlda::kRegMaxN;
            const int m0 = std::min(n, trlda::kRegMaxN);
            for (int j = 0; j < m0; ++j)
                row[j] = ids[p0 + j];
            const int32_t fill = n > 0 ? ids[p0 + std::min(n, trlda::kRegMaxN) - 1] : 0;
            for (int j = m0; j < trlda::kRegMaxN; ++j)
                row[j] = fill;
        }
    }
    x->split_pays = false;
    if (n_wg > 0) {
        // Does splitting pay for THIS batch?  A launch lasts max(longest workgroup, all work /
        // CUs).  Per iteration, in microseconds at K = 100 (profiles/r03_length_sweep*.txt; only
        // the ratios matter): a document on one workgroup costs 1.0 + 0.0025 n up to 128 words,
        // 1.5 up to 144, 0.013 n up to 192 and 2.5 + 0.025 (n - 192) beyond; a segment 2.6
        // whatever its document's length -- 1.1 to 1.8 times the CU time of the unsplit form,
        // which is why a batch of 400-word documents that fills the chip anyway stays unsplit,
        // and a batch with a few long documents (or, like the reference's test_speed, very
        // uneven ones) does not.
        double sum_u = 0., max_u = 0., sum_s = 0., max_s = 0.;
        for (int i = 0; i < B; ++i) {
            const int n = indptr[order[i] + 1] - indptr[order[i]];
            const double cu = n <= 128 ? 1.0 + 0.0025 * n : n <= 144 ? 1.5 : n <= 192 ? 0.013 * n
                                                                        : 2.5 + 0.025 * (n - 192);
            const int c = segments_of(n);
            sum_u += cu; max_u = std::max(max_u, cu);
            sum_s += c > 1 ? 2.6 * c : cu; max_s = std::max(max_s, c > 1 ? 2.6 : cu);
        }
        x->split_pays = std::max(max_s, sum_s / cus) < 0.95 * std::max(max_u, sum_u / cus);
        int32_t *meta = I(x->o_smeta), *pids = I(x->o_spids);
        size_t w = 0;
        int xrow = 0;
        for (int i = 0; i < B; ++i) {
            const int d = order[i], p0 = indptr[d], n = indptr[d + 1] - p0;
            const int c = segments_of(n);
            const int base = n / c, rem = n % c;
            int start = 0;
            for (int sgm = 0; sgm < c; ++sgm, ++w) {
                const int len = base + (sgm < rem ? 1 : 0);
                int32_t *mm = meta + w * 8;
                mm[0] = d; mm[1] = len; mm[2] = p0 + start; mm[3] = 0;
                mm[4] = sgm; mm[5] = c; mm[6] = c > 1 ? xrow : 0; mm[7] = n;
                int32_t *row = pids + w * trlda::kRegMaxN;
                const int m0 = std::min(len, trlda::kRegMaxN);
                for (int j = 0; j < m0; ++j)
                    row[j] = ids[p0 + start + j];
                const int32_t fill = len > 0 ? ids[p0 + start + m0 - 1] : 0;
                for (int j = m0; j < trlda::kRegMaxN; ++j)
                    row[j] = fill;
                start += len;
            }
            if (c > 1)
                xrow += c;
        }
    }
    {
        int32_t *active = I(x->o_active), *longw = I(x->o_long);
        uint8_t *flag = reinterpret_cast<uint8_t *>(h + x->o_flag);
        int na = 0, nl = 0, longest = 0;
        for (int w = 0; w < V; ++w) {
            const int len = wptr[(size_t)w + 1] - wptr[(size_t)w];
            longest = std::max(longest, len);
            flag[w] = len > 0;
            if (len > 0)
                active[na++] = w;
            if (len > long_len)
                longw[nl++] = w;
        }
        x->long_host.assign(longw, longw + nl);
        x->max_list = longest;
        // the very long lists: equal segments of at most seg_len entries
        int32_t *vw = I(x->o_vlw), *vt = I(x->o_vlt);
        int j = 0, t = 0;
        x->vl_host.clear();
        x->vl_first.clear();
        for (int w = 0; w < V && n_vl > 0; ++w) {
            const int q0 = wptr[(size_t)w], len = wptr[(size_t)w + 1] - q0;
            if (len <= seg_len)
                continue;
            const int ns = (len + seg_len - 1) / seg_len;
            const int base = len / ns, rem = len % ns;
            vw[4 * j] = w; vw[4 * j + 1] = t; vw[4 * j + 2] = ns; vw[4 * j + 3] = 0;
            x->vl_host.push_back(w);
            x->vl_first.push_back(t);
            int start = 0;
            for (int sg = 0; sg < ns; ++sg, ++t) {
                const int sl = base + (sg < rem ? 1 : 0);
                vt[4 * t] = j; vt[4 * t + 1] = sg; vt[4 * t + 2] = q0 + start; vt[4 * t + 3] = sl;
                start += sl;
            }
            ++j;
        }
        x->vl_first.push_back(t);
        {
            // (counting sort of the tasks by sixteenth of the list their segment starts in, stable
            // in the word index)
            int32_t *vtt = I(x->o_vltt);
            int start[17] = {0};
            auto bucket = [&](int q) { return std::min(15, 16 * vt[4 * q + 1] / std::max(1, vw[4 * vt[4 * q] + 2])); };
            for (int q = 0; q < t; ++q)
                ++start[bucket(q) + 1];
            for (int i = 0; i < 16; ++i)
                start[i + 1] += start[i];
            for (int q = 0; q < t; ++q) {
                int32_t *e = vtt + 4 * (size_t)start[bucket(q)]++;
                e[0] = vt[4 * q]; e[1] = vt[4 * q + 1]; e[2] = vt[4 * q + 2]; e[3] = vt[4 * q + 3];
            }
        }
        // descriptors for the merged launch: counting sort by length, longest first, the short
        // lists (<= long_len entries) before the long ones
        int32_t *md = I(x->o_mdesc);
        const int n_short = n_active - n_long;
        std::vector<int32_t> at((size_t)long_len + 2, 0);        // at[len]: next slot of a short list of `len`
        for (int a = 0; a < na; ++a) {
            const int len = wptr[(size_t)active[a] + 1] - wptr[(size_t)active[a]];
            if (len <= long_len)
                ++at[(size_t)len];
        }
        int run = 0;
        for (int len = long_len; len >= 1; --len) {
            const int c = at[(size_t)len];
            at[(size_t)len] = run;
            run += c;
        }
        std::vector<int32_t> longs;
        for (int a = 0; a < na; ++a) {
            const int w = active[a], q0 = wptr[(size_t)w], len = wptr[(size_t)w + 1] - q0;
            if (len > long_len) {
                longs.push_back(w);
                continue;
            }
            int32_t *e = md + 4 * (size_t)at[(size_t)len]++;
            e[0] = w; e[1] = q0; e[2] = len; e[3] = 0;
        }
        for (int c = 0; c < 4; ++c)
            x->cls_short[c] = x->cls_long[c] = 0;
        for (int a = 0; a < na; ++a) {
            const int len = wptr[(size_t)active[a] + 1] - wptr[(size_t)active[a]];
            const int unit = len <= long_len ? len : (len + 15) / 16;       // a list, or a chunk of one
            const int c = unit > 8 ? 0 : unit > 4 ? 1 : unit > 2 ? 2 : 3;
            ++(len <= long_len ? x->cls_short : x->cls_long)[c];
        }
        std::stable_sort(longs.begin(), longs.end(), [&](int32_t a, int32_t b) {
            return wptr[(size_t)a + 1] - wptr[(size_t)a] > wptr[(size_t)b + 1] - wptr[(size_t)b];
        });
        for (size_t i = 0; i < longs.size(); ++i) {
            const int w = longs[i];
            int32_t *e = md + 4 * ((size_t)n_short + i);
            e[0] = w; e[1] = wptr[(size_t)w]; e[2] = wptr[(size_t)w + 1] - wptr[(size_t)w]; e[3] = 0;
        }
    }
}

}  // namespace trlda_host

// The index of a batch without a device (tests, tools/index_rate.py): info[0..31] = V, B, nnz, max_n,
// n_active, n_long, long_len, n_vl, n_vl_tasks, seg_len, n_wg, n_xrows, max_list, split_pays, wc32_ok,
// cnts_nonneg, cls_short[4], cls_long[4], total; info[32..] = the 19 section offsets in layout order,
// then `total` again (so that section i is [info[32 + i], info[33 + i]) up to its padding).
// buffer == NULL: plan only (counts, offsets); else `cap` >= total bytes are written.
extern "C" int trlda_debug_batch_index(int V, int B, const int32_t *indptr, const int32_t *ids,
                                       const int32_t *cnts, int cus, int64_t *info, void *buffer, size_t cap)
{
    using namespace trlda_host;
    if (!info)
        return fail(TRLDA_ERR_ARG, "info is NULL");
    BatchIndex x;
    int rc = batch_index_plan(V, B, indptr, ids, cnts, &x);
    if (rc)
        return rc;
    if (buffer) {
        if (cap < x.total)
            return fail(TRLDA_ERR_ARG, "buffer too small for the index");
        batch_index_fill(&x, indptr, ids, cnts, cus, static_cast<char *>(buffer));
    }
    const int64_t head[] = {x.V, x.B, x.nnz, x.max_n, x.n_active, x.n_long, x.long_len, x.n_vl, x.n_vl_tasks,
                            x.seg_len, x.n_wg, x.n_xrows, x.max_list, x.split_pays, x.wc32_ok, x.cnts_nonneg,
                            x.cls_short[0], x.cls_short[1], x.cls_short[2], x.cls_short[3],
                            x.cls_long[0], x.cls_long[1], x.cls_long[2], x.cls_long[3], (int64_t)x.total};
    for (int i = 0; i < 32; ++i)
        info[i] = i < (int)(sizeof(head) / sizeof(head[0])) ? head[i] : 0;
    const size_t offs[] = {x.o_indptr, x.o_ids, x.o_cnts, x.o_order, x.o_wrank, x.o_wptr, x.o_wdoc, x.o_meta,
                           x.o_pids, x.o_smeta, x.o_spids, x.o_active, x.o_long, x.o_flag, x.o_wc32, x.o_mdesc,
                           x.o_vlw, x.o_vlt, x.o_vltt, x.total};
    for (int i = 0; i < 20; ++i)
        info[32 + i] = (int64_t)offs[i];
    return TRLDA_OK;
}
