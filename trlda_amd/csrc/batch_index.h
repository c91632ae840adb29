// batch_index.h -- a mini-batch's index, built on the host (host only; see host_common.h).
//
// The reference hands LDA::updateVariables a vector<vector<pair<int, int>>> (include/lda.h:21-23,
// python/src/ldainterface.cpp:152-190).  The device wants the flat CSR form of it plus what the
// kernels find their work by: the word-major order of the entries (a stable counting sort of the
// CSR positions by word: a word's entries in document order, the reference's order of additions,
// src/lda.cpp:207-213), the documents by decreasing length with their padded id rows, the active
// words and their lists by length class, the segment tasks of very long lists.  All of it lands
// in ONE buffer (256-byte aligned sections) that goes to the device in one copy.
//
// Two steps, so that the caller can take memory between them: batch_index_plan reads the batch
// once (validation, the histogram of the word ids, the scans over the vocabulary) and fixes the
// counts and the layout; batch_index_fill writes the sections.  Neither touches HIP: they run on
// the library's worker threads (trlda_hip.hip, trlda_batch_create) and under the sanitizers.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace trlda_host {

struct BatchIndex {
    int V = 0, B = 0, max_n = 0;
    int64_t nnz = 0;
    int n_active = 0, n_long = 0, long_len = 16;
    int n_vl = 0, n_vl_tasks = 0, seg_len = 256;
    int n_wg = 0, n_xrows = 0;
    int max_list = 0;
    bool split_pays = false, wc32_ok = true, cnts_nonneg = true;
    int cls_short[4] = {0, 0, 0, 0}, cls_long[4] = {0, 0, 0, 0};
    // byte offsets of the sections (trlda_batch's arrays of the same names)
    size_t o_indptr = 0, o_ids = 0, o_cnts = 0, o_order = 0, o_wrank = 0, o_wptr = 0, o_wdoc = 0, o_meta = 0,
           o_pids = 0, o_smeta = 0, o_spids = 0, o_active = 0, o_long = 0, o_flag = 0, o_wc32 = 0, o_mdesc = 0,
           o_vlw = 0, o_vlt = 0, o_vltt = 0, total = 0;
    // host copies the launch logic and the data-parallel paths read
    std::vector<int32_t> wptr;          // V + 1 word segment offsets
    std::vector<int32_t> sorted_len, indptr_host, long_host, vl_host, vl_first;
};

// What a caller can know before the index is planned (trlda_batch_create's own thread, while the
// index is built on another): the lengths' validation with the longest document and the split layout's
// workgroups; the word ids' range; where the CSR arrays lie in the buffer (their offsets depend on B and
// the number of entries alone) and an upper bound of the buffer's size.
int batch_index_check_lengths(int V, int B, const int32_t *indptr, int *max_n, int *n_wg);
int batch_index_check_ids(int V, int64_t nnz, const int32_t *ids);
void batch_index_csr_offsets(int B, int64_t nnz, size_t *o_ids, size_t *o_cnts);
size_t batch_index_size_bound(int V, int B, int64_t nnz, int n_wg);

// TRLDA_OK, or TRLDA_ERR_ARG / TRLDA_ERR_WORD_ID with the message set (host_common.h, fail)
int batch_index_plan(int V, int B, const int32_t *indptr, const int32_t *ids, const int32_t *cnts,
                     BatchIndex *x);
// writes x->total bytes at `h` (the gaps between sections are left as they are); `cus`: the device's
// compute units (whether splitting long documents pays depends on how full the chip is).  The three
// CSR arrays may already lie where they belong (indptr == h + o_indptr, ...): they are left alone then.
void batch_index_fill(BatchIndex *x, const int32_t *indptr, const int32_t *ids, const int32_t *cnts, int cus,
                      char *h);

}  // namespace trlda_host
