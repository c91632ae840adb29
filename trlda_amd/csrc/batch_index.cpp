// batch_index.cpp -- a mini-batch's word-major index, built on the host (host only; see
// batch_index.h for what it is and host_common.h for why these files need no HIP).
#include "batch_index.h"

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <time.h>

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

int batch_index_check_lengths(int V, int B, const int32_t *indptr, int *max_n_out, int *n_wg_out)
{
    if (V <= 0 || B < 0 || !indptr)
        return fail(TRLDA_ERR_ARG, "bad batch dimensions");
    if (indptr[0] != 0)
        return fail(TRLDA_ERR_ARG, "indptr[0] must be 0");
    int max_n = 0, n_wg = 0, n_xrows = 0;
    for (int d = 0; d < B; ++d) {
        const int n = indptr[d + 1] - indptr[d];
        if (indptr[d + 1] < indptr[d])
            return fail(TRLDA_ERR_ARG, "indptr must be non-decreasing");
        max_n = std::max(max_n, n);
        const int c = segments_of(n);
        n_wg += c;
        n_xrows += c > 1 ? c : 0;
    }
    *max_n_out = max_n;
    *n_wg_out = n_xrows ? n_wg : 0;
    return TRLDA_OK;
}

int batch_index_check_ids(int V, int64_t nnz, const int32_t *ids)
{
    // (the OR of the ids and of V - 1 - id: some id outside [0, V) sets the sign bit of one of them)
    uint32_t any = 0;
    const uint32_t last = (uint32_t)V - 1u;
    for (int64_t i = 0; i < nnz; ++i)
        any |= (uint32_t)ids[i] | (last - (uint32_t)ids[i]);
    if (any & 0x80000000u)
        return fail(TRLDA_ERR_WORD_ID, "word id outside [0, num_words)");
    return TRLDA_OK;
}

void batch_index_csr_offsets(int B, int64_t nnz, size_t *o_ids, size_t *o_cnts)
{
    *o_ids = align256(std::max<size_t>(((size_t)B + 1) * 4, 4));
    *o_cnts = *o_ids + align256(std::max<size_t>((size_t)nnz * 4, 4));
}

size_t batch_index_size_bound(int V, int B, int64_t nnz, int n_wg)
{
    const size_t nz = (size_t)nnz, Bz = (size_t)B, Vz = (size_t)V;
    const size_t n_active = std::min(Vz, nz), n_long = std::min(Vz, nz / (trlda::kLongWord + 1));
    const size_t n_vl = nz / (trlda::kSegMin + 1), n_tasks = nz / trlda::kSegMin + n_vl;
    const size_t sizes[] = {(Bz + 1) * 4, nz * 4, nz * 4, Bz * 4, nz * 4, (Vz + 1) * 4, nz * 4, Bz * 16,
                            Bz * trlda::kRegMaxN * 4, (size_t)n_wg * 32, (size_t)n_wg * trlda::kRegMaxN * 4,
                            n_active * 4, n_long * 4, Vz, Vz * 4, n_active * 16, n_vl * 16, n_tasks * 16,
                            n_tasks * 16};
    size_t total = 0;
    for (size_t b : sizes)
        total += align256(std::max<size_t>(b, 4));
    return total;
}

// Written for the host's time per mini-batch (round 6: 60 us of a 75 us trlda_batch_create were spent
// here, against 26 us of kernels per 200 documents): every pass over the entries or the vocabulary is
// branch-free where the data decide (a word is active or not with probability ~1/2: a mispredicted
// branch per word), the vocabulary is scanned twice in all, scratch lives in the thread, and the
// rare shapes (very long lists, a word whose counts overflow 32 bits, documents of more than 65 535
// words) take loops of their own.  What comes out is the index of rounds 1-5 bit for bit
// (tests/test_batch_index.py, tests/golden/f13_batch_index.json).
int batch_index_plan(int V, int B, const int32_t *indptr, const int32_t *ids, const int32_t *cnts,
                     BatchIndex *x)
{
    if (V <= 0 || B < 0 || !indptr)
        return fail(TRLDA_ERR_ARG, "bad batch dimensions");
    if (indptr[0] != 0)
        return fail(TRLDA_ERR_ARG, "indptr[0] must be 0");
    int max_n = 0, n_wg = 0, n_xrows = 0;
    for (int d = 0; d < B; ++d) {
        const int n = indptr[d + 1] - indptr[d];
        if (indptr[d + 1] < indptr[d])
            return fail(TRLDA_ERR_ARG, "indptr must be non-decreasing");
        max_n = std::max(max_n, n);
        const int c = segments_of(n);
        n_wg += c;
        n_xrows += c > 1 ? c : 0;
    }
    if (n_xrows == 0)
        n_wg = 0;                                    // no split document: no second layout
    const int64_t nnz = indptr[B];
    if (nnz > 0 && (!ids || !cnts))
        return fail(TRLDA_ERR_ARG, "ids / cnts are NULL");
    x->V = V; x->B = B; x->max_n = max_n; x->nnz = nnz;

    // word-major segment offsets first: they give the number of active and long words, i.e.
    // the layout of the allocation.  The histogram of the ids, and their validation on the way: an
    // id outside [0, V) (one unsigned comparison) is counted as V - 1 and reported after the loop
    std::vector<int32_t> &wptr = x->wptr;
    wptr.assign((size_t)V + 1, 0);
    {
        int32_t *cnt = wptr.data() + 1;
        const uint32_t last = (uint32_t)V - 1u;
        uint32_t worst = 0;
        for (int64_t i = 0; i < nnz; ++i) {
            const uint32_t id = (uint32_t)ids[i];
            worst = std::max(worst, id);
            ++cnt[std::min(id, last)];
        }
        if (nnz > 0 && worst > last)
            return fail(TRLDA_ERR_WORD_ID, "word id outside [0, num_words)");
    }
    // ONE scan of the vocabulary: offsets, active words, the longest list, the entries in lists of more
    // than kSegMin, and how many lists exceed kLongWord << i for every i (a list of `len` entries
    // exceeds the first floor(log2((len - 1) / kLongWord)) + 1 of them)
    constexpr int kLevels = 16;
    int n_active = 0, longest = 0;
    long long heavy = 0;
    int lvl[kLevels + 1] = {0};
    {
        int32_t run = 0;
        for (int w = 0; w < V; ++w) {
            const int len = wptr[(size_t)w + 1];
            n_active += len > 0;
            longest = std::max(longest, len);
            heavy += len > trlda::kSegMin ? len : 0;
            if (len > trlda::kLongWord) {            // (a few per cent of the words, or nearly all: predictable)
                const unsigned t = (unsigned)(len - 1) / (unsigned)trlda::kLongWord;
                ++lvl[std::min(32 - __builtin_clz(t), kLevels)];  // kLongWord << i < len for i < that
            }
            run += len;
            wptr[(size_t)w + 1] = run;
        }
    }
    int over[kLevels];
    for (int i = kLevels - 1, acc = 0; i >= 0; --i) {
        acc += lvl[i + 1];
        over[i] = acc;
    }
    int level = 0;
    while (level + 1 < kLevels && over[level] > trlda::kLongWordsTarget)
        ++level;
    int long_len = trlda::kLongWord << level, n_long = over[level];
    // (the longest lists are cut into segments, estep_kernels.h: the one-wave range stays short)
    if (long_len > trlda::kOneWaveMax) {
        long_len = trlda::kOneWaveMax;
        n_long = 0;
        for (int w = 0; w < V; ++w)
            n_long += wptr[(size_t)w + 1] - wptr[(size_t)w] > long_len;
    }
    // the longest lists as segment tasks (estep_kernels.h, VeryLongArgs): segment length per batch
    int n_vl = 0, n_vl_tasks = 0, seg_len = trlda::kSegMin;
    {
        static const long long seg_tasks = std::getenv("TRLDA_SEG_TASKS") ? std::atoll(std::getenv("TRLDA_SEG_TASKS"))
                                                                          : (long long)trlda::kSegTasks;
        while (seg_len < trlda::kSegMax && heavy / seg_len > seg_tasks)
            seg_len *= 2;
    }
    if (longest > seg_len) {                         // (else: no list is cut)
        for (int w = 0; w < V; ++w) {
            const int len = wptr[(size_t)w + 1] - wptr[(size_t)w];
            if (len > seg_len) {
                ++n_vl;
                n_vl_tasks += (len + seg_len - 1) / seg_len;
            }
        }
    }
    x->n_active = n_active; x->n_long = n_long; x->long_len = long_len;
    x->n_vl = n_vl; x->n_vl_tasks = n_vl_tasks; x->seg_len = seg_len;
    x->n_wg = n_wg; x->n_xrows = n_xrows;
    x->max_list = longest;

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

namespace {

// a document's (or a segment's) row of kRegMaxN word ids: its first ids, then its last id repeated
// (rows that exist; masked by length)
inline void padded_row(int32_t *row, const int32_t *src, int len)
{
    const int m0 = std::min(len, trlda::kRegMaxN);
    std::memcpy(row, src, (size_t)m0 * 4);
    const int32_t fill = len > 0 ? src[m0 - 1] : 0;
    for (int j = m0; j < trlda::kRegMaxN; ++j)
        row[j] = fill;
}

struct FillScratch {
    std::vector<int32_t> cursor, bins, alen, awords;
    std::vector<int64_t> pair;
};

}  // namespace

void batch_index_fill(BatchIndex *x, const int32_t *indptr, const int32_t *ids, const int32_t *cnts, int cus,
                      char *h)
{
    static thread_local FillScratch scratch;
    const int V = x->V, B = x->B;
    const size_t nz = (size_t)x->nnz, Bz = (size_t)B;
    const int n_active = x->n_active, n_long = x->n_long, long_len = x->long_len;
    const int n_vl = x->n_vl, seg_len = x->seg_len, n_wg = x->n_wg;
    const std::vector<int32_t> &wptr = x->wptr;
    auto I = [&](size_t o) { return reinterpret_cast<int32_t *>(h + o); };

    if (I(x->o_indptr) != indptr)
        std::memcpy(I(x->o_indptr), indptr, (Bz + 1) * 4);
    if (nz && I(x->o_ids) != ids)
        std::memcpy(I(x->o_ids), ids, nz * 4);
    if (nz && I(x->o_cnts) != cnts)
        std::memcpy(I(x->o_cnts), cnts, nz * 4);
    std::memcpy(I(x->o_wptr), wptr.data(), ((size_t)V + 1) * 4);
    // stable counting sort of the CSR positions by word id, and the words' count sums
    {
        int32_t *wrank = I(x->o_wrank), *wdoc = I(x->o_wdoc), *wc32 = I(x->o_wc32);
        std::vector<int32_t> &cursor = scratch.cursor;
        // (sum |cnt| below 2^31: no word's sum -- nor any partial sum -- leaves 32 bits; bounded first
        // by entries x the largest |cnt|, exactly only when that does not settle it)
        // (the OR of all counts: negative exactly when one of them is, and an upper bound of every one
        // of them when none is -- a reduction the baseline instruction set vectorises)
        int32_t any = 0;
        for (size_t p = 0; p < nz; ++p)
            any |= cnts[p];
        x->cnts_nonneg = any >= 0;
        int64_t mass = any >= 0 ? (int64_t)nz * any : (int64_t)INT32_MAX + 1;
        if (mass > INT32_MAX) {
            mass = 0;
            for (size_t p = 0; p < nz; ++p)
                mass += std::abs((int64_t)cnts[p]);
        }
        if (mass <= INT32_MAX) {
            // (a word's cursor and its running sum side by side: one line, one 8-byte load and store)
            std::vector<int64_t> &pair = scratch.pair;
            pair.resize((size_t)V);
            for (int w = 0; w < V; ++w)
                pair[(size_t)w] = (int64_t)(uint32_t)wptr[(size_t)w];       // low half: cursor; high half: sum
            for (int d = 0; d < B; ++d)
                for (int32_t p = indptr[d]; p < indptr[d + 1]; ++p) {
                    const size_t w = (size_t)ids[p];
                    const int64_t v = pair[w];
                    const int32_t q = (int32_t)(uint32_t)v;
                    pair[w] = v + 1 + (int64_t)((uint64_t)(int64_t)cnts[p] << 32);
                    wrank[p] = q;
                    wdoc[q] = d;
                }
            for (int w = 0; w < V; ++w)
                wc32[w] = (int32_t)(pair[(size_t)w] >> 32);
            x->wc32_ok = true;
        } else {
            std::memset(wc32, 0, (size_t)V * 4);
            cursor.assign(wptr.begin(), wptr.end() - 1);
            std::vector<int64_t> wsum((size_t)V, 0);
            for (int d = 0; d < B; ++d)
                for (int32_t p = indptr[d]; p < indptr[d + 1]; ++p) {
                    const int32_t q = cursor[(size_t)ids[p]]++;
                    wrank[p] = q;
                    wdoc[q] = d;
                    wsum[(size_t)ids[p]] += cnts[p];
                }
            bool ok = true;
            for (int w = 0; w < V; ++w) {
                ok = ok && wsum[(size_t)w] >= INT32_MIN && wsum[(size_t)w] <= INT32_MAX;
                wc32[w] = (int32_t)wsum[(size_t)w];
            }
            x->wc32_ok = ok;
        }
    }
    // documents by decreasing length, equal lengths in document order: a counting sort by length
    int32_t *order = I(x->o_order);
    if (x->max_n <= 65535) {
        std::vector<int32_t> &bins = scratch.bins;
        bins.assign((size_t)x->max_n + 2, 0);
        for (int d = 0; d < B; ++d)
            ++bins[(size_t)(indptr[d + 1] - indptr[d])];
        int32_t run = 0;
        for (int n = x->max_n; n >= 0; --n) {
            const int32_t c = bins[(size_t)n];
            bins[(size_t)n] = run;
            run += c;
        }
        for (int d = 0; d < B; ++d)
            order[bins[(size_t)(indptr[d + 1] - indptr[d])]++] = d;
    } else {
        std::iota(order, order + B, 0);
        std::stable_sort(order, order + B, [&](int32_t a, int32_t b) {
            return indptr[a + 1] - indptr[a] > indptr[b + 1] - indptr[b];
        });
    }

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
            padded_row(pids + (size_t)i * trlda::kRegMaxN, ids + p0, n);
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
                padded_row(pids + w * trlda::kRegMaxN, ids + p0 + start, len);
                start += len;
            }
            if (c > 1)
                xrow += c;
        }
    }
    {
        // the second scan of the vocabulary: flags, the active words (and their lengths, for the
        // descriptors below), the long words, the lengths' histogram -- a word is active or not about
        // as often as not: every word is written at the list's end and the end moves on for an active one
        int32_t *active = I(x->o_active), *longw = I(x->o_long);
        uint8_t *flag = reinterpret_cast<uint8_t *>(h + x->o_flag);
        // (the histogram of the lengths in four copies by the low bits of the word id, and from the
        // middle of the active list on in a second set of four: a counter that every other word
        // increments is a chain of store-to-load forwards, ~5 cycles a word)
        std::vector<int32_t> &alen = scratch.alen, &bins = scratch.bins;
        alen.resize((size_t)n_active + 1);
        const int nb = long_len + 2;                 // bins [len]: short lists of `len` entries; [long_len + 1]: longer
        bins.assign((size_t)nb * 8, 0);
        const int half = (n_active + 1) / 2;
        std::vector<int32_t> &awords = scratch.awords;
        awords.resize((size_t)n_active + 1);
        int na = 0, nl = 0;
        {
            // (local, restrict-qualified pointers: `flag` is a byte pointer, which the compiler must
            // otherwise assume to alias every vector's bookkeeping -- a reload of each per word)
            const int32_t *__restrict wp = wptr.data();
            int32_t *__restrict aw = awords.data(), *__restrict al = alen.data(), *__restrict hist = bins.data();
            uint8_t *__restrict fl = flag;
            const int cap = long_len + 1;
            for (int w = 0; w < V; ++w) {
                const int len = wp[w + 1] - wp[w];
                const bool on = len > 0;
                fl[w] = on;
                aw[na] = w;                          // (written for every word, kept when it is active: the
                al[na] = len;                        //  slot past the end exists, overwritten or unused)
                ++hist[((na >= half) * 4 + (w & 3)) * nb + (len < cap ? len : cap)];
                na += on;
                if (len > long_len)                  // (few words, or most of them: predictable either way)
                    longw[nl++] = w;
            }
        }
        std::memcpy(active, awords.data(), (size_t)n_active * 4);
        x->long_host.assign(longw, longw + nl);
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
        // lists (<= long_len entries) before the long ones; and the lists by length class
        // (entries of a short list, or of a sixteenth of a long one: 9..16 | 5..8 | 3..4 | 1..2)
        int32_t *md = I(x->o_mdesc);
        const int n_short = n_active - n_long;
        for (int c = 0; c < 4; ++c)
            x->cls_short[c] = x->cls_long[c] = 0;
        // cur[0 | 1][len]: the next slot of a short list of `len` entries in the first | second half of the
        // active list (two cursors: two chains of dependent increments instead of one)
        int32_t *cur0 = bins.data(), *cur1 = bins.data() + nb;
        {
            int32_t run = 0;
            for (int len = long_len; len >= 1; --len) {
                int32_t c0 = 0, c1 = 0;
                for (int q = 0; q < 4; ++q) {
                    c0 += bins[(size_t)(q * nb + len)];
                    c1 += bins[(size_t)((4 + q) * nb + len)];
                }
                x->cls_short[len > 8 ? 0 : len > 4 ? 1 : len > 2 ? 2 : 3] += c0 + c1;
                cur0[len] = run;                     // (rows 0 and 1 of `bins`: their counts for this and the
                cur1[len] = run + c0;                //  lengths still to come have been read)
                run += c0 + c1;
            }
        }
        std::vector<int32_t> longs;
        {
            const int32_t *__restrict wp = wptr.data(), *__restrict aw = awords.data(), *__restrict al = alen.data();
            int32_t *__restrict c0 = cur0, *__restrict c1 = cur1;
            int64_t *__restrict md2 = reinterpret_cast<int64_t *>(md);       // a descriptor: two 8-byte halves
            auto place = [&](int a, int32_t *__restrict cur) {
                const int w = aw[a], len = al[a];
                if (len > long_len)
                    return;
                const size_t slot = (size_t)cur[len]++;
                md2[2 * slot] = (int64_t)(uint32_t)w | ((int64_t)(uint32_t)wp[w] << 32);     // (word, first entry)
                md2[2 * slot + 1] = (int64_t)(uint32_t)len;                                  // (entries, 0)
            };
            for (int a = 0; a < half; ++a) {
                place(a, c0);
                if (half + a < na)
                    place(half + a, c1);
            }
        }
        if (n_long > 0)
            for (int a = 0; a < na; ++a) {
                const int len = alen[(size_t)a];
                if (len > long_len) {
                    longs.push_back(active[a]);
                    const int unit = (len + 15) / 16;
                    ++x->cls_long[unit > 8 ? 0 : unit > 4 ? 1 : unit > 2 ? 2 : 3];
                }
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

// microseconds per call of the index builder on this batch (tools/index_rate.py): `reps` calls into one
// buffer; mode 0: batch_index_plan alone, 1: plan + fill
extern "C" double trlda_debug_batch_index_rate(int V, int B, const int32_t *indptr, const int32_t *ids,
                                               const int32_t *cnts, int cus, int reps, int mode)
{
    using namespace trlda_host;
    BatchIndex probe;
    if (batch_index_plan(V, B, indptr, ids, cnts, &probe))
        return -1.0;
    std::vector<char> buf(probe.total);
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int r = 0; r < reps; ++r) {
        BatchIndex x;
        (void)batch_index_plan(V, B, indptr, ids, cnts, &x);
        if (mode)
            batch_index_fill(&x, indptr, ids, cnts, cus, buf.data());
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return ((t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3) / reps;
}
