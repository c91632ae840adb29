// dp_kernels.h -- data-parallel E-step with FACTOR exchange.
//
// The reference adds the K x V statistics of its threads under `omp critical`
// (src/lda.cpp:211-217).  Across GPUs that sum is an all-reduce of K * V doubles per E-step:
// 5.6 / 80 / 400 MB at BASELINE.json's table sizes, 2 (N-1)/N times that over every GPU's links.
// But the statistics are a product of factors that are tiny next to the table,
//
//   sstats[k, w] = expElogbeta[k, w] * sum_{(d, w) in batch} (cnt_dw / phinorm_dw) expElogtheta[d, k]
//
// -- lambda (hence expElogbeta) is replicated, so a rank only has to tell the others the
// K numbers expElogtheta[d, :] of each of its documents and one weight per (document, word)
// entry: 8 (K + n_d) bytes per document instead of 8 K V per rank (0.3 MB instead of 5.6 MB per
// rank at K = 100, 200 documents of ~90 words; 2.5 instead of 400 MB at K = 500, 512 documents).
// Every rank then forms the statistics of the WHOLE mini-batch with the single-GPU kernel, which
// adds a word's entries in document order -- the reference's serial order: every rank performs
// the same additions on the same numbers, so the replicas stay bitwise equal and equal the
// one-GPU result up to the rounding of the document-kernel variant a shard selects (an
// all-reduce's order of additions depends on the rank count), and the fused M-step with carried
// row sums (estep_kernels.h, 4c) applies unchanged.
//
// Layout of the gathered buffer, `world` equal slots (ncclAllGather wants equal counts):
//
//   slot r = [ expElogtheta of rank r's documents, rows of K | weights of its entries, CSR order | pad ]
//            |<- tw_off = max_r(documents) * K doubles       ->|
//   slot size = a multiple of K doubles, so that a document's row is slot-relative row index
//
// factor_index_kernel: once per (mini-batch, cut points) -- the mini-batch keeps the result --
// one wavefront per document of the whole mini-batch: which rank, where its weights lie in the
// gathered buffer.  For every entry in word-major order (what the statistics kernel walks): the
// position of its weight in the gathered buffer and the row index of its document's
// expElogtheta there.  The statistics kernel then reads the gathered buffer directly
// (TwView, estep_kernels.h): per E-step the exchange is the all-gather and nothing else.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace trlda {

constexpr int kDpMaxWorld = 64;

struct DpCuts {                      // document cut points of the mini-batch, by value
    int32_t at[kDpMaxWorld + 1];
};

template <int T>
__global__ __launch_bounds__(T) void factor_index_kernel(
    int B, int world, const int32_t *__restrict__ indptr, const int32_t *__restrict__ wrank,
    DpCuts cuts /* world + 1 document cut points */, size_t slot, int slot_rows, size_t tw_off,
    int32_t *__restrict__ wsrc, int32_t *__restrict__ wrow)
{
    // one wavefront per document: its rank and its place in the rank's slot are wave-uniform,
    // its entries are consecutive in the slot (CSR order) and spread over the lanes
    const int lane = threadIdx.x & 63;
    const int d = (int)(((size_t)blockIdx.x * T + threadIdx.x) >> 6);
    if (d >= B)
        return;
    int r = 0;
    while (r + 1 < world && cuts.at[r + 1] <= d)     // (uniform: scalar loads of the argument)
        ++r;
    const int first = cuts.at[r];
    const int p0 = indptr[d], p1 = indptr[d + 1], pf = indptr[first];
    const size_t src = (size_t)r * slot + tw_off - (size_t)pf;
    const int row = r * slot_rows + (d - first);
    for (int p = p0 + lane; p < p1; p += 64) {
        const int q = wrank[p];
        wsrc[q] = (int32_t)(src + (size_t)p);
        wrow[q] = row;
    }
}

// ---------------------------------------------------------------------------
// DIRECT exchange of the factor slots (behind trlda_model_dp_direct_connect; RCCL's all-gather
// stays the default): xGMI is point to point, and a slot is a few hundred kB -- so every rank
// writes its slot straight into the gather buffers of its peers, which are mapped into this
// process through hipIpc handles, and tells them so with a per-source step counter in their
// buffer; no collective launch sits on the critical path of a step.
//
//   slot_push_kernel    this rank's slot -> every peer's buffer (same offset there), all peers
//                       at once, 16-byte stores; the end of the kernel is its release
//   slot_signal_wait_kernel   flags[rank] = step at every peer (system scope), then waits until
//                       every peer has signalled the same step here; the statistics kernel that
//                       follows in the stream starts with an acquire and reads the buffer
// Two buffers alternate by step parity: when a rank pushes step s + 2 into the buffer of step s,
// every peer has signalled s + 1, i.e. has finished the statistics of step s (stream order).
// ---------------------------------------------------------------------------
struct DpPeers {
    double *buf[kDpMaxWorld];                // base of rank r's exchange region in THIS process
    unsigned long long *flags[kDpMaxWorld];  // its kDpMaxWorld step counters
};

template <int T>
__global__ __launch_bounds__(T) void slot_push_kernel(DpPeers peers, int rank, int world,
                                                      size_t offset /* doubles: buffer parity + rank * slot */,
                                                      size_t count /* doubles, even */)
{
    const double2 *__restrict__ src = reinterpret_cast<const double2 *>(peers.buf[rank] + offset);
    const size_t n2 = count / 2;
    // blockIdx.y = which peer (skipping this rank), blockIdx.x strides over the slot
    int p = (int)blockIdx.y;
    if (p >= rank)
        ++p;
    if (p >= world)
        return;
    double2 *__restrict__ dst = reinterpret_cast<double2 *>(peers.buf[p] + offset);
    for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < n2; i += (size_t)gridDim.x * T)
        dst[i] = src[i];
    // the peer polls its counter while this device's later kernels run: the stores leave this
    // workgroup's XCD for the peer's memory here, not only at the kernel's end
    __threadfence_system();
}

__global__ void slot_signal_wait_kernel(DpPeers peers, int rank, int world, unsigned long long step,
                                        int *give_up)
{
    const int p = threadIdx.x;
    if (p < world && p != rank)
        __hip_atomic_store(peers.flags[p] + rank, step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (p < world && p != rank) {
        long long spins = 0;
        while (__hip_atomic_load(peers.flags[rank] + p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < step) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1ll << 26)) {             // a peer never came: give up, loudly
                *give_up = 1;
                break;
            }
        }
    }
}

}  // namespace trlda
