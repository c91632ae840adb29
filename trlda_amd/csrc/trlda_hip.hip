// trlda_hip.hip -- C ABI (include/trlda_hip.h) over the gfx950 kernels in
// estep_kernels.h.  Host logic only: validation, device memory, launch sequences,
// and the control loops of OnlineLDA::updateParameters (reference
// src/onlinelda.cpp:53-111) and BatchLDA::updateParameters (src/batchlda.cpp:43-61).
//
// There is no CPU compute path in this file: if HIP cannot see a device, every
// compute entry point fails with TRLDA_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <map>
#include <set>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <tuple>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <vector>

#include "../../include/trlda_hip.h"
#include "host_common.h"
#include "batch_index.h"
#include "estep_kernels.h"
#include "estep_wide.h"
#include "estep_merged.h"
#include "elbo_kernels.h"
#include "stream_kernels.h"
#include "eb_kernels.h"
#include "rng_kernels.h"
#include "dp_kernels.h"

namespace {

// frees a temporary device allocation on every path out of a function
struct DevTemp {
    void *p = nullptr;
    ~DevTemp()
    {
        if (p)
            (void)hipFree(p);
    }
};

using trlda_host::fail;                  // the thread-local message behind trlda_last_error()

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
// what a kernel may ask for dynamically: the rest belongs to its static __shared__ variables (a
// flag, a counter: 16 bytes in the kernels here).  Round 4: a tiered launch whose LDS rows came out
// at exactly 160 kB was refused by hipFuncSetAttribute (tests/fuzz_estep.py, seed 3)
constexpr int kLdsDynBytes = kLdsBytes - 256;
#ifdef TRLDA_STAMPS
unsigned long long *g_stamp_buf = nullptr;
#endif
constexpr int kDenseThreads = 256;
constexpr int kMaxRowsumBlocks = 1025;   // rows of trlda_model::partial (block partials of row sums)
constexpr int kUpdGroups = 64;           // rows the document kernel adds up itself (topic_scale_load)
constexpr int kCarryBlocks = 8;          // preamble_fused_kernel: workgroups adding up carried partials
constexpr int kUpdShortBlocks = 1024;    // sstats_update_kernel: blocks walking the short lists
constexpr int kUpdLongBlocks = 512;      //                       blocks walking the long lists (two per CU resident)
constexpr int kUpdSegBlocks = 2048;      //                       blocks walking the segments of the very long lists
constexpr int kUpdVlRows = 4096;         // very long words whose lambdas take a row of upd_partial each; a batch
                                         // with more of them keeps them in the one-block-per-word path
constexpr double kFusedRowsumFloor = 2e-3;   // psi(2e-3) = -500.6: exp(-psi(row sum)) stays finite

template <typename T>
int dev_alloc(T **p, size_t count)
{
    *p = nullptr;
    if (count == 0)
        count = 1;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T)));
    return TRLDA_OK;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel and size, not per launch
int ensure_dynamic_lds(const void *func, size_t bytes)
{
    static std::mutex mu;
    static std::map<const void *, size_t> granted;
    if (bytes <= 48 * 1024)
        return TRLDA_OK;
    std::lock_guard<std::mutex> lock(mu);
    auto it = granted.find(func);
    if (it != granted.end() && it->second >= bytes)
        return TRLDA_OK;
    HIP_TRY(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    granted[func] = bytes;
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
    // The index is built and uploaded by a worker thread of the library while trlda_batch_create's caller
    // goes on (round 6): what the caller's thread has established itself -- device, V, B, nnz, max_n, id --
    // is valid from the start, everything else once the ticket's state is kBuilt.  Every entry point that is
    // handed a batch waits for that first (batch_wait); a failed build (out of device memory) fails
    // the entry point with the build's status and message.
    // kQueued: nobody has started on it -- whoever needs the batch first (batch_wait) takes the build
    // from the queue and runs it on its own thread, so that "create, use at once" costs what it cost
    // when trlda_batch_create built the index itself; and a batch destroyed before anybody needed it
    // is never indexed at all (kCancelled).  The state lives in a ticket the queued job shares: the
    // job may outlive the batch.
    // kBuilding: a thread is at the index (host work only: batch_fill).  kFilled: the index lies in the
    // staging buffer; the upload -- allocation, one copy, two events -- is enqueued by a CALLER's thread
    // (kUploading: batch_upload), never by a worker: the next trlda_batch_create, the E-step the batch is
    // announced to, or its first user.
    enum { kBuilt = 0, kQueued = 1, kFailed = 2, kBuilding = 3, kCancelled = 4, kFilled = 5, kUploading = 6 };
    struct Ticket {
        std::atomic<int> state{kBuilt};
    };
    std::shared_ptr<Ticket> ticket = std::make_shared<Ticket>();
    void *slot = nullptr;              // the staging buffer that holds its CSR arrays, then its index, until the upload
    std::unique_ptr<trlda_host::BatchIndex> index;   // the layout, between batch_fill and batch_upload
    int cus = 256;                     // the device's compute units (whether splitting long documents pays)
    int build_rc = 0;
    std::string build_msg;
    bool destroy_when_built = false;   // trlda_batch_destroy came while a thread was at it: that thread destroys it
    // Every array below lives in ONE device allocation (`blob`), filled by ONE host-to-device
    // copy from pinned memory on the upload stream; `ready` marks the end of that copy and
    // `done` the last kernel that read the batch (recorded by every entry point that uses it),
    // so that creating, using and recycling batches never waits for the device.
    void *blob = nullptr;
    size_t blob_bytes = 0;
    hipEvent_t ready = nullptr, done = nullptr;
    bool used = false;
    // `done` is recorded when the batch is destroyed, on the stream that used it last (an event
    // record per use costs a barrier packet per launch sequence); a batch that moves to another
    // stream settles the first one on the spot
    hipStream_t last_stream = nullptr;
    bool last_owned = false;      // last_stream is a model's own stream (see live_own_streams)
    // the stream `done` was last recorded on (null: never recorded).  Round 4's fuzz run found a
    // caller's array changed under the library, and a watchpoint build put the write inside
    // hipEventQuery on a guard event whose recording stream had been destroyed
    // (profiles/r04_fuzz_watchpoint.txt).  The stand-alone probe (tools/probes/event_stream_probe.hip,
    // profiles/r04_event_stream_probe.txt) does NOT reproduce it -- 0 bytes written in all twelve
    // orders -- so the mechanism inside the runtime is inferred, not shown; the rule the library
    // keeps is the conservative one: an event is waited on, queried, recorded again or recycled
    // only while the stream that recorded it exists (stream_alive), and every event a model's own
    // stream has recorded is destroyed BEFORE that stream is (purge_stream_guards: cached
    // allocations' guards, spare events, live batches' `done`)
    hipStream_t done_on = nullptr;
    bool done_on_owned = false;
    bool ready_seen = false;      // the upload has been seen complete: no more waits
    uint64_t id = 0;            // unique per batch (an address can be reused by a later batch)
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
    uint8_t *active_flag = nullptr;   // V bytes: 1 for the words in `active`
    int32_t *long_words = nullptr;   // words with more than long_len entries
    int long_len = 16;               // (estep_kernels.h, kLongWord)
    int n_long = 0;
    // per document, in `order`: (document, length, CSR offset, 0) and its first kRegMaxN word
    // ids padded to that length -- the register kernel's workgroup finds everything it needs
    // at an address that depends on its index only
    int32_t *pad_meta = nullptr;     // B x 4
    int32_t *pad_ids = nullptr;      // B x kRegMaxN
    // the same per WORKGROUP when documents of more than kSplitMinN words are split into segments
    // of at most kSplitSegN (estep_docs_reg_body<0, true>): (document, segment length, CSR offset
    // of the segment, 0 | segment, segments, first exchange row of the document, document length)
    // and the segment's ids; only for batches that hold such a document
    int32_t *seg_meta = nullptr;     // n_wg x 8
    int32_t *seg_ids = nullptr;      // n_wg x kRegMaxN
    int n_wg = 0, n_xrows = 0;       // workgroups; exchange rows per iteration (sum of segments)
    bool split_pays = false;         // the launch is expected to end sooner with split documents
    std::vector<int32_t> sorted_len;   // host copy: document lengths in `order`
    std::vector<int32_t> indptr_host;  // host copy of indptr (data-parallel slot geometry)
    // data-parallel factor exchange: where each word-major entry's weight and document row lie
    // in the gathered buffer, for the cut points / geometry in dp_sig (built on first use)
    int32_t *dp_wsrc = nullptr, *dp_wrow = nullptr;
    std::vector<int64_t> dp_sig;
    // word-sharded M-step (data-parallel): host copies of the word-major offsets and of the long
    // words, and per world size the ranks' ranges of the vocabulary (word ids, positions in
    // `active`, positions in `long_words`: world + 1 cut points each), balanced by entries
    std::vector<int32_t> wptr_host, long_host;
    int ws_world = 0;
    std::vector<int32_t> ws_wcuts, ws_acuts, ws_lcuts;
    // sum of the counts per word (onlinelda.cpp:79-82), formed on the host while the batch is
    // indexed; valid unless a sum does not fit 32 bits (then the device adds them up)
    int32_t *wc32 = nullptr;
    bool wc32_ok = false;
    bool cnts_nonneg = true;        // no negative count: the statistics are >= 0 (see lambda_positive)
    // merged launch (estep_merged.h): (word, first entry, entries, 0) of the active words, first
    // the n_short words of at most long_len entries, then the n_long others, each group by
    // decreasing length (a wave's words of one round are then about equally long)
    int32_t *mdesc = nullptr;       // n_active x 4
    int n_short = 0;
    trlda_model *pending_in = nullptr;   // a model whose deferred statistics still read this batch
    // the lists by length class, longest first (estep_merged.h, deferred_stats): short lists of
    // 9..16 | 5..8 | 3..4 | 1..2 entries, long lists whose chunks (a sixteenth, rounded up) have
    // 9..16 | 5..8 | 3..4 | 1..2; meaningful when long_len == kLongWord and max_list <= 256
    int cls_short[4] = {0, 0, 0, 0}, cls_long[4] = {0, 0, 0, 0};
    int max_list = 0;               // entries of the longest word list (repeated ids within a document
                                    // are legal, lda.cpp:108: a list can be longer than B)
    // very long lists (estep_kernels.h, VeryLongArgs): words of more than seg_len entries, by word
    // id -- (word, first task, segments, 0) -- and their segment tasks (word index, segment, first
    // entry, entries)
    int32_t *vl_word = nullptr, *vl_task = nullptr;
    // the same tasks ordered by where in the document range their segment lies (all words' first
    // sixteenth, then the second ..): workgroups that run at the same time then gather rows of the
    // same few hundred documents -- L2 hits instead of 20 MB of rows streaming through 4 MB caches
    int32_t *vl_task_tiled = nullptr;
    int n_vl = 0, n_vl_tasks = 0, seg_len = 256;
    std::vector<int32_t> vl_host;   // host copy of the very long words' ids (ranks' slices)
    std::vector<int32_t> vl_first;  // ... and of their first tasks, + the total (n_vl + 1)
    std::vector<int32_t> ws_vcuts;  // word-sharded M-step: cut points in the very long words
};

// A data-parallel call in flight (dp_kernels.h): the model holds the whole mini-batch `b`, iterates
// the documents of `shard` = documents [cuts[rank], cuts[rank + 1]) of it, and exchanges factors.
struct DpContext {
    const trlda_batch *shard = nullptr;
    int rank = 0, world = 1;
    std::vector<int32_t> cuts;          // world + 1 document cut points of the whole mini-batch
    void *comm = nullptr;               // ncclComm_t, unless the model has an all-gather hook
    size_t slot = 0, tw_off = 0;        // doubles
    int doc_lo() const { return cuts[(size_t)rank]; }
    // Word-sharded M-step: after the factor exchange every rank forms statistics + M-step for ITS
    // range of the vocabulary only (batch->ws_*cuts) and the ranks exchange the lambda columns
    // they wrote, in place
    bool word_sharded = false;
};

struct trlda_model {
    int device = 0;
    int K = 0, V = 0;
    // A stream of the model's own (non-blocking): waits on other streams' events -- a batch's
    // upload -- then stay on the device; on the legacy null stream the runtime resolves them on
    // the host, which serialises the host behind the GPU.  trlda_model_set_stream replaces it.
    hipStream_t stream = nullptr, own_stream = nullptr;
    // The preamble of the batch announced as the next one, prepared by extra workgroups of this
    // call's document-kernel launch (estep_kernels.h, PreArgs) in alternating buffers; valid for
    // that batch while lambda has not been written (lambda_version).
    uint64_t lambda_version = 1;
    // (three of each: with deferred statistics the batch before this one still reads its buffer)
    double *eeb_pp[3] = {nullptr, nullptr, nullptr}, *partial_pp[3] = {nullptr, nullptr, nullptr};
    double *scale_pp[3] = {nullptr, nullptr, nullptr};   // 3 K each: the finished topic factors (PreArgs::c_out)
    // Deferred statistics (trlda_model_set_deferred_stats; estep_merged.h): in a stream of E-steps
    // on an unchanged lambda (trlda_model_estep_io_next) the statistics of a call (lda.cpp:207-217)
    // are not launched by that call -- they ride on the NEXT call's document launch, or are
    // launched as the kernel of their own by whatever touches the model next (flush_pending:
    // every entry point, trlda_model_flush, trlda_batch_destroy of the batch they read).  What the
    // pending statistics read is kept apart from what the next call writes: exp(psi(gamma)) rows
    // and weights in two alternating buffers, exp(psi(lambda)) of its batch in a third buffer.
    bool deferred_stats = false;
    struct {
        bool valid = false;
        const trlda_batch *batch = nullptr;
        double *sstats = nullptr;         // the caller's K x V
        double *epg = nullptr, *tw_word = nullptr;
        double *eeb = nullptr;
        int eeb_buf = -1;                 // index into eeb_pp, or -1: m->eeb
    } pending;
    double *dfr_epg_base[2] = {nullptr, nullptr}, *dfr_tw[2] = {nullptr, nullptr};
    size_t dfr_cap_docs[2] = {0, 0}, dfr_cap_tw[2] = {0, 0};
    int dfr_cur = 0;
    unsigned int defer_work_total = 0;    // what sync_counters[32] holds: the helpers' items so far
    bool last_deferred = false;           // the last E-step left its statistics pending
    bool last_carried = false;            // ... / its launch carried the call before's
    // Stream lanes (trlda_model_set_stream_lanes): the E-steps of a deferred stream
    // dealt in turn to two models of their own -- own streams, own workspaces, lambda and alpha
    // THIS model's -- so that the launches of consecutive calls may overlap on the device: the
    // documents of a launch end 1-2 us apart (a 129..144-word one 5 us after the others) and the
    // next launch's workgroups take the CUs as they come free.  Nothing of a lane's work is on this
    // model's stream until lanes_join() (every entry point except the next E-step of the stream).
    int lanes_wanted = 1;
    // 0: no lanes made yet; 2: two lanes whose streams were SEEN to run side by side with each other and
    // with the model's stream (lanes_ensure); 3: two lanes, not looked at (TRLDA_LANE_VERIFY=0); 1: no such
    // pair of streams was to be had from the runtime -- the stream goes one launch at a time
    int lane_state = 0;
    // ... and MEASURED (trlda_model_estep_io_ahead): what the probe kernels of lanes_ensure cannot see
    // -- round 5's failing case, a model in a process that had made and destroyed streams: 33.2 us per
    // step through two lanes against 30.5 through one -- shows in the stream's own launches.  Two
    // launches in flight means: a launch LASTS about two steps (51 us where a new one starts every
    // 26); lanes that do not overlap have launches of one step's length (30 us every 30-34).  After
    // kLaneCalAfter steps through the lanes, eight steps into a stretch, one launch of lane 0 is timed
    // (an event before it, one behind it) together with the four launches that follow it on that lane
    // (an event behind the last): launches_in_flight = duration / step interval; below 1.0 the lanes
    // are given up (lane_state 1), from 1.4 on kept, in between measured again.  Three events, no step taken out of the lanes, once per model (again
    // after trlda_model_set_stream); a window a join falls into is started again, at most eight
    // times.  TRLDA_LANE_CALIBRATE=0: never.
    struct {
        int phase = 0;                    // 0 counting | 2 the window | 3 waiting for the events | 4 done
        int n = 0;                        // lane 0's calls into the window
        int tries = 0;                    // windows broken by a join so far
        hipEvent_t e[3] = {nullptr, nullptr, nullptr};
        float us_launch = 0.f, us_step = 0.f;
        // (the window says something about the DEVICE only while the host keeps the lanes fed: the
        // host's own time between the window's first and last call is taken too, and a window in
        // which the calls came slower than 0.7 of the device's step -- a caller that makes its batches
        // as it goes, bench.py's value_end_to_end -- is no verdict: measured again, at most eight times)
        std::chrono::steady_clock::time_point host_t0;
        float host_us_step = 0.f;
        // The lanes are kept on two windows in a row that say so, and looked at AGAIN every
        // kLaneCalEvery steps for as long as they live: the failing case has starts in which the launches
        // overlap for a while (a window read 1.41) and then do not (52-57 us per step against 30.5
        // through one lane, profiles/r06_lanes_two_streams.txt) -- one good window is not a verdict
        // for the process's life.  A window below 1.0 ends the lanes whenever it comes.
        int keeps = 0;                    // windows in a row that kept the lanes
        long long next_check = 0;         // lane_steps at which the next window opens (phase 4)
        // What launches in flight cannot tell: two lanes whose launches DO overlap and are slow for it
        // (a start of the failing case: windows of 1.4-3.2 launches in flight, stretches of 52-57 us
        // per step against 30.5 through one lane).  So every look at the lanes begins with the thing
        // they are to beat, measured the same way: phase 6, kLaneSoloLead + kLaneSoloSteps calls in a
        // row on lane 0 alone (no join: the other lane simply gets nothing), events behind the
        // lead-in's launch and the last one -- a step through ONE lane.  Two lanes that are not 3 %
        // faster than that: a look that says no (as one with less than one launch in flight does); two of
        // the last four looks: given up.
        hipEvent_t s[2] = {nullptr, nullptr};
        // (the window's end is the LATER of the two lanes' last launches: a window timed on lane 0's
        // stream alone reads half the true step when the device runs that lane ahead of the other --
        // 28.9 us "per step" in a process whose lanes never beat 30.5: f, behind lane 1's last launch)
        hipEvent_t f = nullptr;
        int n1 = 0;                       // lane 1's launches into the window
        float us_solo = 0.f, host_us_solo = 0.f;
        std::chrono::steady_clock::time_point solo_t0;
        int lead = 0;                     // steps through both lanes before the window opens
        unsigned worse = 0;               // the last four looks, a bit each: two lanes did not pay
        // (the one-lane stretch is eighteen steps at one lane's speed: it begins kLaneSoloAfter calls
        // into a stretch, never in a caller's short stretches -- bench.py's 20-step regions between
        // fences get the window alone, as before, once a stretch has ended too early)
        bool solo_valid = false;          // s[0], s[1] are this look's
        int short_stretches = 0;
    } lane_cal;
    trlda_model *lane[2] = {nullptr, nullptr};
    trlda_model *lane_owner = nullptr;    // set in a lane: whose lambda / alpha it reads
    int lane_turn = 0;
    bool lanes_live = false;              // a lane holds work this model's stream has not waited for
    hipEvent_t lane_in[2] = {nullptr, nullptr}, lane_out[2] = {nullptr, nullptr};
    struct LaneRange {
        const char *lo = nullptr, *hi = nullptr;
    };
    // what the lane's launches since the last join write: the arrays of EVERY call of the stretch (with
    // deferred statistics a call's sstats are written by the lane's NEXT launch, and any launch since
    // the join may still be outstanding), each range once -- a streaming caller rotates a few sets
    std::vector<LaneRange> lane_writes[2];
    int lane_calls[2] = {0, 0};
    LaneRange prev_writes[3];             // the arrays of the stream's last call (whichever way it went)
    // timing (trlda_model_set_timing): a lane's launches run back to back on its stream, so one pair
    // of events per lane around a whole stretch of calls -- first call after a join to the join --
    // gives the launches' mean duration without an event between any two of them
    hipEvent_t lane_span[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    bool lane_span_open[2] = {false, false};
    int64_t lane_span_launches[2] = {0, 0}, lane_span_count = 0;
    double lane_span_us = 0.0;
    int64_t lane_steps = 0;               // E-steps that went through the lanes (tests)
    struct {
        bool valid = false;
        uint64_t batch_id = 0, version = 0;
        int buf = 0, G = 0;
        bool dense = false;
    } prefetch;
    double *eeb_cur = nullptr;          // the exp E[log beta] buffer of the E-step in flight
    int sstats_mode = TRLDA_SSTATS_SEGMENTED;
    int doc_threads = 0;
    int doc_kernel = 0;    // TRLDA_DOCS_*
    int small_k = -1;      // a wave per document at K <= 32 (estep_docs_small_body): -1 where it applies, 0 never,
                           // 1 asked for (trlda_model_set_doc_kernel(TRLDA_DOCS_SMALL / _REG); TRLDA_SMALL_K)
    const char *last_doc_kernel = "";   // kernel that took most documents of the last E-step
    bool last_preamble_fused = false;
    bool split_preamble = false;        // never fuse kernels 1 and 2 (tests, comparisons)
    bool dense_preamble = false;   // true: exp E[log beta] for all V words, as the reference
    double *lambda = nullptr, *alpha = nullptr;
    double *eeb = nullptr, *psi_sum = nullptr, *partial = nullptr;
    unsigned int *counter = nullptr;
    // Row sums carried from the kernels that wrote lambda: rs_full[k] = sum_w lambda[k, w]
    // (lda.cpp:172) is valid while rs_valid, and the next E-step skips its row-sum pass.
    // rs_static: the part over the words outside the current update's mini-batch.
    double *rs_full = nullptr, *rs_static = nullptr, *upd_partial = nullptr;
    bool rs_valid = false;
    // ... or they are still in pieces: block partials `carry_rows` (carry_n rows of K) plus
    // `carry_base` (K, or null).  Small tables add them up inside the next preamble launch
    // (preamble_fused_kernel's combine workgroups, 8 rows into carry_out); anything else
    // resolves them with rowsum_combine_wave_kernel first.
    bool carry_pending = false;
    const double *carry_rows = nullptr, *carry_base = nullptr;
    int carry_n = 0;
    double *carry_out = nullptr;        // kCarryBlocks x K
    // The preamble the last M-step kernel left behind (UpdateOut::u_out / group_rows): valid for
    // lambda_version `version`, for every word (`all`) or the active words of batch `batch_id`
    double *upd_groups = nullptr;       // kUpdGroups x K
    unsigned int *group_counter = nullptr;
    struct {
        bool valid = false, all = false;
        uint64_t version = 0, batch_id = 0;
        int n = 0;
        bool raw = false;               // the rows are block rows still to be added up (merged launch)
    } next_pre;
    // Big tables (K > 128): the M-step of a trust-region iteration left u = exp(psi(lambda)) of the
    // batch's words in `eeb` (sstats_update2_kernel<.., EMIT>): the next E-step on the same batch
    // and the same lambda needs no exp_elog_beta_kernel -- the single-orientation document kernel
    // applies the topic factors instead (estep_docs_wide_kernel<KS, true>)
    struct {
        bool valid = false;
        uint64_t batch_id = 0, version = 0;
    } u_left;
    bool big_emit = true;               // TRLDA_BIG_EMIT=0: exp_elog_beta_kernel in every iteration
    bool emit_next_preamble = true;     // trlda_model_set_fused_update(.. & 2 == 0)
    bool carry_rowsums = true;          // trlda_model_set_carry_rowsums (tests, comparisons)
    bool fused_update = true;           // trlda_model_set_fused_update: statistics + M-step in one pass
    bool keep_sstats = false;           // updates also leave the statistics (and all of lambda') behind
    // A lower bound on every row sum of lambda, kept on the host: the fused preamble forms
    // exp(-psi(row sum)), which overflows below 1.4e-3 (estep_kernels.h, 2b)
    double rs_floor = 0.0;
    int64_t d2h_bytes = 0;              // bytes copied to the host through this model (tests)
    bool lambda_exposed = false;        // trlda_model_lambda_dev was handed out: never trust rs_*
    // Every element of lambda is known to be > 0 (or NaN): set from the host copy when lambda is
    // uploaded, kept by the M-steps that provably keep it (mstep_keeps_positive), dropped by
    // everything else that writes lambda.  Only then may an M-step kernel also emit exp(psi(lambda))
    // with the call-free positive-argument form (psi.h, exp_digamma_positive).
    bool lambda_positive = false;
    bool pair_gathers = true;           // statistics kernel with two topics per lane (even K > 128)
    bool prefetch_next = true;          // trlda_model_set_prefetch: honour "next batch" announcements
    // per-batch workspaces, grown on demand
    size_t cap_docs = 0, cap_tw_csr = 0, cap_tw_word = 0;
    // Data-parallel factor exchange (DpContext below): the gathered factors of all ranks, the
    // statistics kernel's row index into them, the shard cut points on the device
    DpContext *dp = nullptr;            // set for the duration of a *_dp call
    double *dp_gather = nullptr;        // the buffer of the call in flight: one of the two below
    double *dp_gather_own = nullptr, *dp_gather_direct = nullptr;
    size_t cap_dp_gather = 0;
    int (*allgather_hook)(void *, const void *, void *, size_t, void *) = nullptr;
    void *allgather_ctx = nullptr;
    int (*allgatherv_hook)(void *, void *, const size_t *, int, int, void *) = nullptr;
    void *allgatherv_ctx = nullptr;
    bool word_sharding = true;          // trlda_model_set_word_sharding
    bool split_long_lists = true;       // TRLDA_SPLIT_LISTS=0: very long lists stay one workgroup's (comparisons)
    bool tiled_tasks = true;            // TRLDA_TILED_TASKS=0: segment tasks in word order (comparisons)
    bool last_word_sharded = false;     // the last *_dp call's M-steps were word-sharded
    // direct exchange (dp_kernels.h): this process's region [2 x world x max_slot doubles |
    // kDpMaxWorld step counters], exported through hipIpc; the peers' regions mapped here
    struct {
        void *region = nullptr;
        size_t max_slot = 0;                // doubles per slot the region was sized for
        int world = 0, rank = -1;
        bool connected = false;
        trlda::DpPeers peers{};
        std::vector<void *> opened;         // hipIpcOpenMemHandle results to close
        unsigned long long step = 0;
    } direct;
    double *epg = nullptr, *tw_csr = nullptr, *tw_word = nullptr;
    double *epg_base = nullptr;         // the allocation: K zeros (row -1 of epg), then epg
    // very long lists (VeryLongArgs): a row of K sums per segment task, a counter per word
    double *seg_partial = nullptr;
    size_t cap_seg_partial = 0;
    unsigned int *seg_counter = nullptr;
    size_t cap_seg_counter = 0;
    // merged launch (estep_merged.h): the statistics as workgroups of the document launch.
    // Two counters that only grow (documents done | topic factors finished) and what they have
    // been asked to reach so far; the finished topic factors of an in-launch combine
    // 0: never; 1 (default): where the statistics carry an M-step -- the update loops, where a
    // launch of their own costs 13 us against ~9 inside the document launch (round 4: 0.53 -> 0.50 ms
    // per update_parameters(max_iter_tr=10)); 2: also for plain E-steps, where the two forms take
    // the same time to 1 % (39.6 / 40.0 us per step) and the kernel of its own is the default
    int merged_launch = 1;              // trlda_model_set_merged_launch
    bool last_merged = false;
    unsigned int *sync_counters = nullptr;
    unsigned int docs_done_total = 0, c_ready_total = 0;
    // ... and a flag per waiter (statistics workgroups | document workgroups, 64 bytes apart) that
    // receives the number of the launch it may go on in (merged_epoch: only grows)
    unsigned int *sync_flags = nullptr;
    unsigned int merged_epoch = 0;
    unsigned long long *merged_stamps = nullptr;   // diagnostics, TRLDA_MERGED_STAMPS=1
    unsigned long long *deferred_stamps = nullptr; // ... of a deferred launch (3 x 3072)
    double *scale_comb = nullptr;       // 3 K
    // split documents: the exchange rows of a launch (NaN before it), the give-up flag
    double *xbuf = nullptr;
    size_t cap_xbuf = 0;
    // the give-up flag: one int in pinned, host-coherent memory that the kernels see through
    // `xerr` -- checking it after a synchronisation is a plain host read (check_split_exchange)
    int *xerr = nullptr;
    volatile int *xerr_host = nullptr;
    bool split_docs = true;             // trlda_model_set_split_docs
    int last_split_wgs = 0;             // workgroups of the last document launch beyond one per document
    // update_parameters workspaces
    double *lambda_prime = nullptr, *sstats = nullptr, *gamma = nullptr, *wordcounts = nullptr;
    size_t cap_gamma = 0, cap_lambda_prime = 0, cap_sstats = 0, cap_wordcounts = 0;
    double *stage[2] = {nullptr, nullptr};      // pinned host buffers for gamma0 draws
    size_t cap_stage[2] = {0, 0};
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    int stage_next = 0;
    // adaptive learning rate: running average of the updates (onlinelda.cpp:170), K x V
    double *ada_gradient = nullptr;
    double *reduce_out = nullptr;               // small buffer for block results of reductions
    // empirical-Bayes sums on their way to the host (trlda_model_online_eb_begin / _finish):
    // K + G + K doubles in pinned memory, complete when eb_event has passed
    struct {
        bool active = false, alpha = false, eta = false;
        int G = 0, B_total = 0;
        double *host = nullptr;
        size_t cap = 0;
        hipEvent_t event = nullptr;
    } eb;
    size_t cap_reduce = 0;
    int32_t *iters = nullptr;                   // per-document iteration counts (estep_host)
    size_t cap_iters = 0;
    // gamma0 of the NEXT fresh E-step, drawn ahead on a stream of its own while this call's
    // kernels run (fresh_gamma_device): two buffers alternate, `spec` says what is in flight
    bool draw_ahead = false;                    // trlda_model_set_draw_ahead / TRLDA_DRAW_AHEAD=1
    hipStream_t draw_stream = nullptr;
    hipEvent_t ev_main = nullptr, ev_draw = nullptr;
    double *gspec[2] = {nullptr, nullptr};
    size_t cap_gspec[2] = {0, 0};
    struct {
        bool valid = false;
        uint64_t token = 0;
        int which = 0;
        long long total = 0, lo = 0, hi = 0;
        bool in_launch = false;                 // drawn by a document launch of the model's stream: no event
    } spec;
    const double *gamma0_src = nullptr;         // the next E-step on m->gamma reads gamma0 here
    // ... or INSIDE this call's document launch (round 6; estep_merged.h, AuxArgs): extra workgroups of
    // a merged launch draw it on the CUs the documents leave free.  `aux_draw_req`: the fresh draw of
    // this call asks the call's next document launch to carry the draw of the next one (same shape);
    // that launch answers it or drops it.  TRLDA_DRAW_INLAUNCH=0 / trlda_model_set_draw_ahead(m, 0 | 1)
    // with bit 2 clear: every draw in its turn (or on the side stream).
    bool draw_inlaunch = true;
    struct {
        bool valid = false;
        long long total = 0;
    } aux_draw_req;
    int64_t inlaunch_draws = 0;                 // draws made inside a document launch so far (tests)
    unsigned int aux_work_total = 0;            // what sync_counters[48] holds: the auxiliary workgroups' items so far
    int64_t inlaunch_decays = 0;                // inactive-word passes made inside a document launch (tests)
    bool aux_decay = true;                      // trlda_model_set_aux_decay / TRLDA_AUX_DECAY=0
    uint32_t *rng_win2 = nullptr;               // scratch of the draw stream
    size_t cap_rng_win2 = 0;
    double *rng_vbuf2 = nullptr;
    size_t cap_rng_vbuf2 = 0;
    // device-side sampleGamma (rng_kernels.h): segment windows, |u| of a group of passes
    bool host_gamma_draw = false;               // true: the bit-exact host draw (glibc log)
    uint32_t *rng_win = nullptr;
    size_t cap_rng_win = 0;
    double *rng_vbuf = nullptr;
    size_t cap_rng_vbuf = 0;
    // timing: five events per E-step from a pool, resolved lazily (no host sync per step)
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;   // all events ever created
    size_t ev_used = 0;                // events recorded since the last collect
    double usec_sum[5] = {0, 0, 0, 0, 0};
    int64_t usec_cnt[5] = {0, 0, 0, 0, 0};
};

namespace {

// Per-device upload machinery: a non-blocking stream, two pinned staging buffers and a small
// cache of device allocations.  hipMalloc / hipFree cost tens of microseconds and hipFree waits
// for the whole device; a mini-batch lives for one update call.
struct UploadContext {
    std::mutex mu;
    hipStream_t stream = nullptr;
    // pinned staging buffers: one is a build's from stage_acquire until its copy is enqueued (`busy`),
    // and anybody's again once that copy has left (`ev`)
    struct Stage {
        void *host = nullptr;
        size_t cap = 0;
        hipEvent_t ev = nullptr;
        bool busy = false;
    };
    std::vector<std::unique_ptr<Stage>> stages;
    std::condition_variable stage_cv;
    struct Blob {
        void *ptr;
        size_t bytes;
        hipEvent_t done;      // last reader of the previous owner (may be null)
        hipStream_t on = nullptr;     // the stream it was recorded on, and whether that is a model's
        bool on_owned = false;        // own stream (purged when the model goes: purge_stream_guards)
    };
    std::vector<Blob> cache;
    size_t cached_bytes = 0;
    // spare events, each with the stream its last record sits on (null: never recorded, or the
    // upload stream, which lives as long as the process)
    struct Spare {
        hipEvent_t ev;
        hipStream_t on;
        bool on_owned;
    };
    std::vector<Spare> events;
    std::set<trlda_batch *> live;     // batches that exist (their `done` may sit on a model's stream)
    // indices the workers have finished and nobody has uploaded yet (uploads_drain); a batch that was
    // used or destroyed in the meantime is recognised by its ticket
    std::vector<std::pair<trlda_batch *, std::shared_ptr<trlda_batch::Ticket>>> filled;
    void spare(hipEvent_t ev, hipStream_t on = nullptr, bool owned = false)
    {
        if (ev)
            events.push_back(Spare{ev, owned ? on : nullptr, owned});
    }
};

UploadContext &upload_context(int device)
{
    static std::mutex mu;
    static std::map<std::pair<pid_t, int>, UploadContext *> all;   // never destroyed (see host_pool)
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_pair(getpid(), device);
    auto it = all.find(key);
    if (it == all.end())
        it = all.emplace(key, new UploadContext()).first;
    return *it->second;
}

constexpr size_t kBlobCacheMax = 32;   // (several indices are in the making at a time: stage_acquire)
constexpr size_t kBlobCacheBytes = (size_t)1 << 30;

int take_event(UploadContext &u, hipEvent_t *ev)
{
    if (!u.events.empty()) {                         // (none of them sits on a stream that is gone:
        *ev = u.events.back().ev;                    // purge_stream_guards)
        u.events.pop_back();
        return TRLDA_OK;
    }
    HIP_TRY(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    return TRLDA_OK;
}



// the models' own streams that still exist (a destroyed model has synchronised its stream:
// nothing of it can still be reading a batch)
struct LiveStreams {
    std::mutex mu;
    std::set<hipStream_t> own;
};
LiveStreams &live_own_streams()
{
    static LiveStreams *all = new LiveStreams();     // never destroyed (see host_pool)
    return *all;
}

// a model's own stream that has been destroyed must not be reached through an event it recorded
bool stream_alive(hipStream_t s, bool owned)
{
    if (!owned || !s)
        return true;                                 // the host's stream: alive by contract
    LiveStreams &ls = live_own_streams();
    std::lock_guard<std::mutex> lock(ls.mu);
    return ls.own.count(s) != 0;
}

// `done` <- everything the batch's last stream has been given so far.  True when `done` now holds
// a record that may be waited on (false: nothing to wait for -- never used, or the last stream is
// gone and its work with it)
bool batch_settle(trlda_batch *b)
{
    if (!b->used || !b->done)
        return false;
    if (!stream_alive(b->last_stream, b->last_owned))
        return false;                                // stream gone, its work complete
    if (b->done_on && !stream_alive(b->done_on, b->done_on_owned)) {
        // (cannot happen since purge_stream_guards replaces the `done` of every live batch when a
        // model's stream goes; kept as the second line: a fresh event, the old one dropped)
        hipEvent_t fresh = nullptr;
        if (hipEventCreateWithFlags(&fresh, hipEventDisableTiming) != hipSuccess)
            return false;
        b->done = fresh;
        b->done_on = nullptr;
    }
    (void)hipEventRecord(b->done, b->last_stream);
    b->done_on = b->last_stream;
    b->done_on_owned = b->last_owned;
    return true;
}

// a model's own stream is about to go (synchronised by the caller): drop the guards it recorded
void purge_stream_guards(int device, hipStream_t s)
{
    UploadContext &u = upload_context(device);
    std::lock_guard<std::mutex> lock(u.mu);
    for (UploadContext::Blob &c : u.cache)
        if (c.done && c.on_owned && c.on == s) {
            (void)hipEventDestroy(c.done);           // while its stream still exists
            c.done = nullptr;
            c.on = nullptr;
            c.on_owned = false;
        }
    // spare events whose last record sits on it (they were handed out and recorded
    // again after the stream had gone)
    size_t kept = 0;
    for (size_t i = 0; i < u.events.size(); ++i) {
        if (u.events[i].on_owned && u.events[i].on == s)
            (void)hipEventDestroy(u.events[i].ev);
        else
            u.events[kept++] = u.events[i];
    }
    u.events.resize(kept);
    // live batches: the stream is synchronised, so whatever it read of them is complete -- their
    // `done` gets a fresh, unrecorded event (batch_settle then has nothing to replace or leak)
    for (trlda_batch *b : u.live)
        if (b->done && b->done_on_owned && b->done_on == s) {
            hipEvent_t fresh = nullptr;
            if (hipEventCreateWithFlags(&fresh, hipEventDisableTiming) != hipSuccess)
                fresh = nullptr;
            (void)hipEventDestroy(b->done);
            b->done = fresh;
            b->done_on = nullptr;
            b->done_on_owned = false;
        }
}

// TRLDA_CALL_TIMES=1: host time of a lane call's sections, summed (trlda_debug_call_times): [0] the wait
// for the batch's index, [1] set-up, [2] the launch sequence (estep_device), [3] the rest; of [2]: [4] the
// batches' upload events, [5] the document launch itself, [6] the batches' reader marks; [7] calls
double g_call_us[8] = {0, 0, 0, 0, 0, 0, 0, 0};
const bool g_call_times = std::getenv("TRLDA_CALL_TIMES") != nullptr;
struct CallClock {
    std::chrono::steady_clock::time_point t;
    CallClock() { if (g_call_times) t = std::chrono::steady_clock::now(); }
    void to(int i)
    {
        if (!g_call_times)
            return;
        const auto n = std::chrono::steady_clock::now();
        g_call_us[i] += std::chrono::duration<double, std::micro>(n - t).count();
        t = n;
    }
};

// every reader of a batch: wait for its upload, and leave a mark behind
int batch_begin(trlda_model *m, const trlda_batch *b)
{
    CallClock clock;
    struct Done { CallClock &c; ~Done() { c.to(4); } } done{clock};
    if (b && b->ready && !b->ready_seen) {
        if (hipEventQuery(b->ready) == hipSuccess)
            const_cast<trlda_batch *>(b)->ready_seen = true;
        else
            HIP_TRY(hipStreamWaitEvent(m->stream, b->ready, 0));
    }
    return TRLDA_OK;
}
int batch_end(trlda_model *m, const trlda_batch *b)
{
    CallClock clock;
    struct Done { CallClock &c; ~Done() { c.to(6); } } done{clock};
    if (b && b->done) {
        trlda_batch *bb = const_cast<trlda_batch *>(b);
        if (bb->used && bb->last_stream != m->stream) {
            // second stream: it continues behind the first one's readers, so that one record
            // on it (at destruction) covers both
            if (batch_settle(bb))
                HIP_TRY(hipStreamWaitEvent(m->stream, bb->done, 0));
        }
        bb->last_stream = m->stream;
        bb->last_owned = m->stream == m->own_stream;
        bb->used = true;
    }
    return TRLDA_OK;
}

// grow-only device buffer: on failure the pointer is null AND the capacity is 0, so a later,
// smaller request allocates again instead of launching on a null pointer
template <typename T>
int grow(T **p, size_t *cap, size_t count)
{
    if (count <= *cap && *p)
        return TRLDA_OK;
    if (*p)
        (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    int rc = dev_alloc(p, count);
    if (rc)
        return rc;
    *cap = count;
    return TRLDA_OK;
}

int ensure_batch_workspace(trlda_model *m, const trlda_batch *b)
{
    // (row -1 of epg is zero: where the statistics stage of a merged launch points the entries
    // past the end of a list, estep_merged.h)
    int rc = TRLDA_OK;
    const size_t rows = (size_t)std::max(b->B, 1) + 1;
    if (rows * (size_t)m->K > m->cap_docs || !m->epg_base) {
        rc = grow(&m->epg_base, &m->cap_docs, rows * (size_t)m->K);
        m->epg = nullptr;
        if (!rc) {
            HIP_TRY(hipMemsetAsync(m->epg_base, 0, (size_t)m->K * sizeof(double), m->stream));
            m->epg = m->epg_base + m->K;
        }
    }
    if (!rc) rc = grow(&m->tw_csr, &m->cap_tw_csr, (size_t)std::max<int64_t>(b->nnz, 1));
    if (!rc) rc = grow(&m->tw_word, &m->cap_tw_word, (size_t)std::max<int64_t>(b->nnz, 1));
    return rc;
}

int ensure_update_workspace(trlda_model *m, int B)
{
    size_t KV = (size_t)m->K * m->V;
    int rc = grow(&m->lambda_prime, &m->cap_lambda_prime, KV);
    if (!rc) rc = grow(&m->sstats, &m->cap_sstats, KV);
    if (!rc) rc = grow(&m->wordcounts, &m->cap_wordcounts, (size_t)m->V);
    if (!rc) rc = grow(&m->gamma, &m->cap_gamma, (size_t)std::max(B, 1) * m->K);
    return rc;
}

// pinned staging for the gamma0 draws (two slots: the host may draw the next one while the
// previous upload is still in flight)
int ensure_gamma_staging(trlda_model *m, size_t count)
{
    for (int i = 0; i < 2; ++i) {
        if (count > m->cap_stage[i] || !m->stage[i]) {
            if (m->stage[i]) {
                (void)hipEventSynchronize(m->stage_ev[i]);
                (void)hipHostFree(m->stage[i]);
            }
            m->stage[i] = nullptr;
            m->cap_stage[i] = 0;
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&m->stage[i]),
                                  std::max<size_t>(count, 1) * sizeof(double), hipHostMallocDefault));
            m->cap_stage[i] = count;
            if (!m->stage_ev[i])
                HIP_TRY(hipEventCreateWithFlags(&m->stage_ev[i], hipEventDisableTiming));
        }
    }
    return TRLDA_OK;
}

// out_dev[total] = sampleGamma(total, 1, passes) / divisor on the device, from the host's libc
// stream, which is advanced by passes * total draws (defined below, after the generator)
int sample_gamma_on_device(trlda_model *m, long long total, int passes, double divisor, double *out_dev,
                           long long e_lo = 0, long long e_hi = -1, bool ahead = false);
// two levels of transposed jump matrices A^(d 16^l L) on the device (rng_kernels.h, AuxDrawArgs), or
// *out = nullptr when the cache is full (the caller then draws in its turn)
int rng_aux_matrices(int device, long long L, const uint32_t **out);

// gamma = sampleGamma(K, B, 100) / 100 (lda.cpp:135).  Default: drawn on the device from the same
// integer stream (rng_kernels.h).  host_gamma_draw: on the host (glibc's logarithm, bit for bit
// the reference's values), uploaded without blocking so that the draw of the next call can
// overlap the kernels of this one.
// The gamma0 of the next fresh E-step of the same shape, drawn while this call's kernels run:
// on a stream of the model's own, into the buffer the current call does not read, with the host
// stream advanced ahead of its turn (host_rng.cpp: whoever else touches the generator first
// cancels that, and the draw is repeated in its turn).  The draw stream waits for everything the
// main stream has been given so far -- the last reader of the target buffer is among it.
int draw_gamma_ahead(trlda_model *m, long long total, long long lo, long long hi)
{
    if (!m->draw_ahead || m->host_gamma_draw || hi <= lo)
        return TRLDA_OK;
    if (!m->draw_stream) {
        // (confining this stream to the ~50 CUs a 200-document launch leaves idle -- a document
        // workgroup needs a whole CU -- was measured: the draw then takes 230 us instead of 44 and
        // the call 174 us instead of 84)
        // (a priority of its own, TRLDA_DRAW_PRIORITY: the runtime hands a fifth stream of a priority a
        // hardware queue that is in use -- lanes_ensure)
        static const int prio = [] { const char *e = std::getenv("TRLDA_DRAW_PRIORITY"); return e ? std::atoi(e) : 0; }();
        if (prio != 0)
            HIP_TRY(hipStreamCreateWithPriority(&m->draw_stream, hipStreamNonBlocking, prio));
        else
            HIP_TRY(hipStreamCreateWithFlags(&m->draw_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&m->ev_main, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&m->ev_draw, hipEventDisableTiming));
    }
    const int which = m->gamma0_src == m->gspec[0] && m->gspec[0] ? 1 : 0;
    int rc = grow(&m->gspec[which], &m->cap_gspec[which], (size_t)(hi - lo));
    if (rc)
        return rc;
    HIP_TRY(hipEventRecord(m->ev_main, m->stream));
    HIP_TRY(hipStreamWaitEvent(m->draw_stream, m->ev_main, 0));
    const uint64_t token = trlda_host::rng_speculate_begin();
    rc = sample_gamma_on_device(m, total, 100, 100., m->gspec[which], lo, hi, /*ahead=*/true);
    if (rc) {
        trlda_host::rng_speculation_cancel();
        return rc;
    }
    HIP_TRY(hipEventRecord(m->ev_draw, m->draw_stream));
    m->spec.valid = true;
    m->spec.token = token;
    m->spec.which = which;
    m->spec.total = total; m->spec.lo = lo; m->spec.hi = hi;
    m->spec.in_launch = false;
    return TRLDA_OK;
}

// gamma0 = elements [lo, hi) of sampleGamma(total, 1, 100) / 100 for the E-step that follows on
// m->gamma: the draw made ahead if it is this one and still in turn, else drawn now
int device_gamma_now_or_ahead(trlda_model *m, long long total, long long lo, long long hi)
{
    m->gamma0_src = nullptr;
    bool have = false;
    if (m->spec.valid) {
        m->spec.valid = false;
        if (m->spec.total == total && m->spec.lo == lo && m->spec.hi == hi &&
            trlda_host::rng_speculation_claim(m->spec.token)) {
            if (!m->spec.in_launch)                  // (in_launch: an earlier kernel of this stream)
                HIP_TRY(hipStreamWaitEvent(m->stream, m->ev_draw, 0));
            m->gamma0_src = m->gspec[m->spec.which];
            have = true;
        }
        // (else: sample_gamma_on_device below cancels whatever is pending)
    }
    int rc = TRLDA_OK;
    if (!have)
        rc = sample_gamma_on_device(m, total, 100, 100., m->gamma, lo, hi);
    m->aux_draw_req.valid = false;
    if (!rc && m->draw_ahead)
        rc = draw_gamma_ahead(m, total, lo, hi);
    else if (!rc && m->draw_inlaunch && !m->host_gamma_draw && lo == 0 && hi == total) {
        // the next fresh gamma0 of this shape: by the call's next document launch, if it is one that
        // can carry it (estep_device)
        m->aux_draw_req.valid = true;
        m->aux_draw_req.total = total;
    }
    return rc;
}

int fresh_gamma_device(trlda_model *m, int B)
{
    if (m->dp) {
        // this rank's columns of the whole mini-batch's matrix; the stream moves on by all of it,
        // as in the single-process run (every rank draws from the same state: lda.cpp:135)
        const long long lo = (long long)m->K * m->dp->doc_lo(), n = (long long)m->K * m->dp->shard->B;
        if (!m->host_gamma_draw)
            return device_gamma_now_or_ahead(m, (long long)m->K * B, lo, lo + n);
        m->gamma0_src = nullptr;
        std::vector<double> full((size_t)m->K * B);
        trlda_sample_gamma_init(m->K, B, full.data());
        if (n > 0) {
            HIP_TRY(hipMemcpyAsync(m->gamma, full.data() + lo, (size_t)n * sizeof(double),
                                   hipMemcpyHostToDevice, m->stream));
            HIP_TRY(hipStreamSynchronize(m->stream));        // `full` goes out of scope
        }
        return TRLDA_OK;
    }
    if (!m->host_gamma_draw)
        return device_gamma_now_or_ahead(m, (long long)m->K * B, 0, (long long)m->K * B);
    m->gamma0_src = nullptr;
    const size_t count = (size_t)m->K * B;
    int rc = ensure_gamma_staging(m, count);
    if (rc)
        return rc;
    const int slot = m->stage_next;
    m->stage_next ^= 1;
    HIP_TRY(hipEventSynchronize(m->stage_ev[slot]));   // the slot's previous upload is done
    trlda_sample_gamma_init(m->K, B, m->stage[slot]);
    HIP_TRY(hipMemcpyAsync(m->gamma, m->stage[slot], count * sizeof(double), hipMemcpyHostToDevice,
                           m->stream));
    HIP_TRY(hipEventRecord(m->stage_ev[slot], m->stream));
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
    int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds_bytes);
    if (rc)
        return rc;
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

// What the statistics stage of an E-step leaves behind.
struct EstepOut {
    trlda::UpdateOut upd;       // sstats and / or the M-step (estep_kernels.h, 4c)
    bool active_only = false;   // walk the batch's active words only (fused path)
    int partial_rows = 0;       // out: rows written to upd.partial
    // in: the M-step may also leave the next E-step's preamble behind (exp(psi(lambda)) of the
    // words it writes + the row sums in few rows; next_base = the share of the words it does
    // not write, final at launch time, or nullptr).  out: it did, in `groups` rows
    bool emit_next = false;
    const double *next_base = nullptr;
    int groups = 0;
    bool raw_rows = false;      // out: the `groups` rows are block rows still to be added up (merged launch)
    // word-sharded M-step (set by estep_device for the statistics launch): this rank's slice --
    // positions [slice_lo, slice_lo + slice_n) of the active list (active_only) or word ids, and
    // its part [long_lo, long_lo + long_n) of the long words
    bool sliced = false;
    int slice_lo = 0, slice_n = 0, long_lo = 0, long_n = 0, vl_lo = 0, vl_n = 0;
    bool no_rows = false;       // out: no row sums were left behind (the next E-step adds lambda up)
    bool emit_u = false;        // in: big tables -- also leave exp(psi(lambda)) of the written words in eeb
    bool u_emitted = false;     // out: it did
    // in: the words OUTSIDE the batch, lambda = inact_a lambda + inact_b in place (the update call
    // without trust-region loop, onlinelda.cpp:103-109), may be written by auxiliary workgroups of
    // the document launch (estep_merged.h, AuxInactiveArgs); out: it was, the block rows of their
    // row sums are the inact_rows rows at inact_part
    bool inact_wanted = false;
    double inact_a = 0.0, inact_b = 0.0;
    int inact_rows = 0;
    const double *inact_part = nullptr;
    EstepOut() { upd = trlda::UpdateOut{}; }
    explicit EstepOut(double *sstats) : EstepOut() { upd.sstats = sstats; }
};

// statistics + M-step + row sums in one kernel: K <= 512, segmented mode
bool fused_update_available(const trlda_model *m)
{
    return m->fused_update && m->K <= trlda::kWideMaxK && m->sstats_mode == TRLDA_SSTATS_SEGMENTED;
}

// the column-slot streaming kernels (stream_kernels.h) cover K <= 512
bool stream_available(const trlda_model *m) { return m->K <= trlda::kWideMaxK; }

int combine_rowsums(trlda_model *m, const double *partial, int G, const double *base, double *out)
{
    constexpr int T = 256;
    hipLaunchKernelGGL(trlda::rowsum_combine_wave_kernel<T>, dim3((m->K + T / 64 - 1) / (T / 64)),
                       dim3(T), 0, m->stream, m->K, G, partial, base, out);
    HIP_TRY(hipGetLastError());
    return TRLDA_OK;
}

void invalidate_rowsums(trlda_model *m)               // called wherever lambda is (about to be) written
{
    m->rs_valid = false;
    m->carry_pending = false;
    ++m->lambda_version;
}

// carried row sums still in pieces -> rs_full
int resolve_carry(trlda_model *m)
{
    if (!m->carry_pending)
        return TRLDA_OK;
    int rc = combine_rowsums(m, m->carry_rows, m->carry_n, m->carry_base, m->rs_full);
    if (rc)
        return rc;
    m->carry_pending = false;
    m->rs_valid = true;
    return TRLDA_OK;
}

// something knows the row sums of the current lambda (and may be believed)
bool rowsums_carried(const trlda_model *m)
{
    return !m->lambda_exposed && m->carry_rowsums && (m->rs_valid || m->carry_pending);
}

template <int VEC>
int launch_rowsum_stream(trlda_model *m, const trlda::StreamGeom &g, double *partial)
{
    constexpr int T = trlda::kStreamThreads;
    const size_t lds = (size_t)g.cpb * m->K * sizeof(double);
    hipLaunchKernelGGL((trlda::rowsum_stream_kernel<T, VEC>), dim3(g.G), dim3(T), lds, m->stream,
                       m->K, m->V, g.P, g.cpb, m->lambda, partial);
    HIP_TRY(hipGetLastError());
    return TRLDA_OK;
}

// rs_full = row sums of lambda, from scratch (lda.cpp:172)
int rowsums_from_scratch(trlda_model *m)
{
    const trlda::StreamGeom g = trlda::stream_geometry(m->K, m->V);
    int rc = g.vec == 2 ? launch_rowsum_stream<2>(m, g, m->partial)
                        : launch_rowsum_stream<1>(m, g, m->partial);
    if (rc)
        return rc;
    return combine_rowsums(m, m->partial, g.G, nullptr, m->rs_full);
}

// lambda = omr * lambda' + rho * (eta + scale * s) with s >= 0 (no negative count in the batch):
// is every element of the result known to be > 0?  (lambda' = the model's own lambda, or its
// snapshot of it from the start of the call; anything else is not known.)
bool mstep_keeps_positive(const trlda_model *m, const trlda::UpdateOut &u, const trlda_batch *b)
{
    if (!b->cnts_nonneg || !(u.scale >= 0.) || !(u.rho >= 0.) || !(u.omr >= 0.))
        return false;
    // (rho eta > 0 alone does not do it: omr * lambda' with a caller's lambda that has elements
    // <= 0 can outweigh it -- ADVICE r4; the reference's digamma takes its reflection branch there,
    // exp_digamma_positive does not)
    const bool own = u.lambda_prime == m->lambda || (u.lambda_prime && u.lambda_prime == m->lambda_prime);
    if (u.rho * u.eta > 0.)
        return !u.lambda_prime || u.omr == 0. || (own && m->lambda_positive);
    return own && u.omr > 0. && m->lambda_positive;
}

using sstats_update_fn = void (*)(int, int, int, int, int, const int32_t *, const int32_t *, const int32_t *,
                                  const int32_t *, trlda::TwView, const double *, const double *,
                                  trlda::UpdateOut, trlda::VeryLongArgs);
template <int T, int NKB, int NH, bool EMIT>
sstats_update_fn sstats_update_entry()
{
    if constexpr (NH == 0)
        return trlda::sstats_update_kernel<T, NKB, EMIT>;
    else
        return trlda::sstats_update2_kernel<T, NKB, NH, EMIT>;
}

template <int T, int NKB, int NH>                    // NH = 0: one topic per lane (any K)
int launch_sstats_update(trlda_model *m, const trlda_batch *b, EstepOut &out)
{
    constexpr int W = T / trlda::kWave;
    const int K = m->K;
    // (word-sharded M-step: this rank's slice of the list / of the vocabulary and of the long words)
    const int N = out.sliced ? out.slice_n : out.active_only ? b->n_active : m->V;
    const int n_long = out.sliced ? out.long_n : b->n_long;
    const int G_short = std::max(1, std::min(kUpdShortBlocks, (N + W - 1) / W));
    const int G_long = std::min(kUpdLongBlocks, n_long);
    // the very long lists (of this rank's slice): segment tasks for whole workgroups
    trlda::VeryLongArgs vl{};
    const bool segments = b->n_vl > 0 && b->n_vl <= kUpdVlRows && m->split_long_lists;
    if (segments) {
        const int j_lo = out.sliced ? out.vl_lo : 0, j_n = out.sliced ? out.vl_n : b->n_vl;
        vl.j0 = j_lo;
        vl.t0 = b->vl_first[(size_t)j_lo];
        vl.n_words = j_n;
        vl.n_tasks = b->vl_first[(size_t)(j_lo + j_n)] - vl.t0;
        vl.G_seg = std::min(kUpdSegBlocks, vl.n_tasks);
        vl.seg_len = b->seg_len;
        // (a rank's slice: its words' tasks are contiguous in the by-word order; the whole batch:
        // the order that keeps concurrently running workgroups on the same documents)
        vl.task = reinterpret_cast<const int4 *>(out.sliced || !m->tiled_tasks ? b->vl_task : b->vl_task_tiled);
        vl.word = reinterpret_cast<const int4 *>(b->vl_word);
        int rc0 = grow(&m->seg_partial, &m->cap_seg_partial, std::max<size_t>((size_t)vl.n_tasks * K, 1));
        if (!rc0 && (size_t)b->n_vl > m->cap_seg_counter) {
            rc0 = grow(&m->seg_counter, &m->cap_seg_counter, (size_t)b->n_vl + 64);
            if (!rc0 && hipMemsetAsync(m->seg_counter, 0, m->cap_seg_counter * sizeof(unsigned int), m->stream) != hipSuccess)
                rc0 = fail(TRLDA_ERR_HIP, "hipMemsetAsync failed");
        }
        if (rc0)
            return rc0;
        vl.seg_partial = m->seg_partial;
        vl.seg_counter = m->seg_counter;
        vl.row_base = G_short + G_long;
    }
    vl.n_rows = G_short + G_long + vl.n_words;
    const size_t lds = (size_t)W * K * sizeof(double);
    // (the instantiation that also writes exp(psi(lambda)) needs more registers: only where asked)
    constexpr bool can_emit = NKB == 1 && NH <= 1;   // K <= 128
    // (exp(psi(.)) of the lambda it writes, in the form for positive arguments: only where the
    // result is known to be positive)
    const bool positive = out.upd.lambda && mstep_keeps_positive(m, out.upd, b);
    const bool emit = can_emit && out.emit_next && out.upd.lambda && out.upd.partial && positive;
    // (K > 128, two topics per lane: exp(psi(lambda)) only -- the row sums go the carried way)
    constexpr bool can_emit_u = NH == 2;
    const bool emit_u = can_emit_u && !can_emit && out.emit_u && out.upd.lambda && out.active_only && positive &&
                        !out.sliced;
    out.u_emitted = false;
    if (out.upd.lambda)
        m->lambda_positive = positive;
    sstats_update_fn kern = sstats_update_entry<T, NKB, NH, false>();
    if constexpr (can_emit || can_emit_u) {
        if (emit || emit_u)
            kern = sstats_update_entry<T, NKB, NH, true>();
    }
    int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds);
    if (rc)
        return rc;
    out.groups = 0;
    out.upd.u_out = nullptr;
    out.upd.group_rows = nullptr;
    out.upd.group_base = nullptr;
    out.upd.group_counter = nullptr;
    out.upd.group_size = 1;
    if (emit_u) {
        out.upd.u_out = m->eeb;
        out.u_emitted = true;
    }
    if (emit) {
        const int rows = vl.n_rows;
        out.upd.u_out = m->eeb;
        out.upd.group_rows = m->upd_groups;
        out.upd.group_base = out.next_base;
        out.upd.group_counter = m->group_counter;
        out.upd.group_size = (rows + kUpdGroups - 1) / kUpdGroups;
        out.groups = (rows + out.upd.group_size - 1) / out.upd.group_size;
    }
    // (data-parallel: expElogtheta rows of all ranks in the gathered buffer, dp_kernels.h)
    out.upd.w0 = (out.sliced && !out.active_only) ? out.slice_lo : 0;
    const int32_t *list = out.active_only ? b->active + (out.sliced ? out.slice_lo : 0) : nullptr;
    hipLaunchKernelGGL(kern, dim3(G_short + G_long + vl.G_seg), dim3(T), lds, m->stream, K, N, G_short,
                       n_long, b->long_len, list, b->wptr,
                       m->dp ? b->dp_wrow : b->wdoc, b->long_words + (out.sliced ? out.long_lo : 0),
                       m->dp ? trlda::TwView{m->dp_gather, b->dp_wsrc} : trlda::TwView{m->tw_word, nullptr},
                       m->dp ? m->dp_gather : m->epg, m->eeb_cur, out.upd, vl);
    HIP_TRY(hipGetLastError());
    out.partial_rows = vl.n_rows;
    return TRLDA_OK;
}

int sstats_update_device(trlda_model *m, const trlda_batch *b, EstepOut &out)
{
    const int K = m->K;
    // one wavefront per word; 16 words per workgroup for small K, 8 from K = 256 on
    // (measured: 7.1 vs 7.4 us at K = 100, 170 vs 145 us at K = 500)
    if (K % 2 == 0 && m->pair_gathers) {
        // a lane owns two adjacent topics: 16-byte gathers, 256 topics per pass over a list
#ifdef TRLDA_EMIT_T512
        // measurement variant: the instantiation that also emits exp(psi(lambda)) as 512-thread
        // workgroups at <= 80 VGPRs (three per CU) instead of 1024-thread ones at <= 64 (two per CU)
        if (K <= 128 && out.emit_next && out.upd.lambda && out.upd.partial &&
            mstep_keeps_positive(m, out.upd, b))
            return launch_sstats_update<512, 1, 1>(m, b, out);
#endif
        if (K <= 128)
            return launch_sstats_update<1024, 1, 1>(m, b, out);
        if (K < 256)
            return launch_sstats_update<1024, 1, 2>(m, b, out);
        return K <= 256 ? launch_sstats_update<512, 1, 2>(m, b, out)
                        : launch_sstats_update<512, 2, 2>(m, b, out);
    }
    const int NKB = (K + 127) / 128;
    if (K >= 256) {
        switch (NKB) {
        case 2: return launch_sstats_update<512, 2, 0>(m, b, out);
        case 3: return launch_sstats_update<512, 3, 0>(m, b, out);
        default: return launch_sstats_update<512, 4, 0>(m, b, out);
        }
    }
    return NKB == 1 ? launch_sstats_update<1024, 1, 0>(m, b, out)
                    : launch_sstats_update<1024, 2, 0>(m, b, out);
}

// Deferred statistics: the statistics an E-step left pending (trlda_model::pending), as the
// kernel of their own on the model's stream -- what the call itself would have launched, from the
// buffers the call kept apart.  Every entry point that is not the next E-step of the stream gets
// here first (check_model), so nobody ever sees a model with statistics outstanding.
int flush_pending(trlda_model *m)
{
    if (!m->pending.valid)
        return TRLDA_OK;
    const auto p = m->pending;
    m->pending.valid = false;
    const_cast<trlda_batch *>(p.batch)->pending_in = nullptr;
    double *epg = m->epg, *tw = m->tw_word, *eeb_cur = m->eeb_cur;
    m->epg = p.epg; m->tw_word = p.tw_word; m->eeb_cur = p.eeb;
    EstepOut out(p.sstats);
    int rc = sstats_update_device(m, p.batch, out);
    m->epg = epg; m->tw_word = tw; m->eeb_cur = eeb_cur;
    if (!rc && hipGetLastError() != hipSuccess)
        rc = fail(TRLDA_ERR_HIP, "the deferred statistics' launch failed");
    if (!rc)
        rc = batch_end(m, p.batch);
    return rc;
}

// the two buffers the documents of a deferred E-step write into, in turn
int ensure_deferred_workspace(trlda_model *m, const trlda_batch *b, int which)
{
    const size_t rows = (size_t)std::max(b->B, 1) + 1, K = (size_t)m->K;
    int rc = TRLDA_OK;
    if (rows * K > m->dfr_cap_docs[which] || !m->dfr_epg_base[which]) {
        rc = grow(&m->dfr_epg_base[which], &m->dfr_cap_docs[which], rows * K);
        if (!rc)                                     // row -1 is zero (estep_merged.h)
            HIP_TRY(hipMemsetAsync(m->dfr_epg_base[which], 0, K * sizeof(double), m->stream));
    }
    if (!rc)
        rc = grow(&m->dfr_tw[which], &m->dfr_cap_tw[which], (size_t)std::max<int64_t>(b->nnz, 1));
    return rc;
}

// the give-up flag of the exchanges that poll (split documents, the direct slot exchange)
int ensure_xerr(trlda_model *m)
{
    if (m->xerr)
        return TRLDA_OK;
    void *host = nullptr, *dev = nullptr;
    HIP_TRY(hipHostMalloc(&host, sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
    *static_cast<int *>(host) = 0;
    if (hipHostGetDevicePointer(&dev, host, 0) != hipSuccess) {
        (void)hipHostFree(host);
        return fail(TRLDA_ERR_HIP, "hipHostGetDevicePointer failed");
    }
    m->xerr_host = static_cast<volatile int *>(host);
    m->xerr = static_cast<int *>(dev);
    return TRLDA_OK;
}

// ---- data-parallel factor exchange (dp_kernels.h) -------------------------------------------
using nccl_allgather_fn = int (*)(const void *, void *, size_t, int, void *, hipStream_t);
constexpr int kNcclFloat64 = 8, kNcclSum = 0;       // ncclDataType_t / ncclRedOp_t (nccl.h)

void *rccl_symbol(const char *name)
{
    void *sym = dlsym(RTLD_DEFAULT, name);
    if (!sym) {
        const char *libs[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *lib : libs) {
            if (void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL)) {
                sym = dlsym(h, name);
                if (sym)
                    break;
            }
        }
    }
    return sym;
}

// slot geometry and buffers for a call on the whole mini-batch `b` cut at dp->cuts
int dp_prepare(trlda_model *m, const trlda_batch *b, DpContext *dp)
{
    const int K = m->K, world = dp->world;
    if (world < 1 || world > trlda::kDpMaxWorld || dp->rank < 0 || dp->rank >= world)
        return fail(TRLDA_ERR_ARG, "rank / world out of range (world <= 64)");
    if ((int)dp->cuts.size() != world + 1 || dp->cuts.front() != 0 || dp->cuts.back() != b->B)
        return fail(TRLDA_ERR_ARG, "doc_cuts must run from 0 to the mini-batch size in world + 1 steps");
    int max_docs = 0;
    int64_t max_nnz = 0;
    for (int r = 0; r < world; ++r) {
        const int lo = dp->cuts[(size_t)r], hi = dp->cuts[(size_t)r + 1];
        if (lo < 0 || hi < lo || hi > b->B)
            return fail(TRLDA_ERR_ARG, "doc_cuts must be non-decreasing and lie in [0, mini-batch size]");
        max_docs = std::max(max_docs, hi - lo);
        max_nnz = std::max<int64_t>(max_nnz, b->indptr_host[(size_t)hi] - b->indptr_host[(size_t)lo]);
    }
    const trlda_batch *sh = dp->shard;
    const int lo = dp->doc_lo(), hi = dp->cuts[(size_t)dp->rank + 1];
    if (!sh || sh->B != hi - lo || sh->nnz != b->indptr_host[(size_t)hi] - b->indptr_host[(size_t)lo])
        return fail(TRLDA_ERR_SHAPE, "the shard is not documents [doc_cuts[rank], doc_cuts[rank + 1]) of the mini-batch");
    dp->tw_off = (size_t)max_docs * K;
    dp->slot = (dp->tw_off + (size_t)max_nnz + (size_t)K - 1) / (size_t)K * (size_t)K;
    if (dp->slot / (size_t)K * (size_t)world > (size_t)INT32_MAX)
        return fail(TRLDA_ERR_ARG, "mini-batch too large for the factor exchange");
    if (dp->slot * (size_t)world > (size_t)INT32_MAX)
        return fail(TRLDA_ERR_ARG, "mini-batch too large for the factor exchange");
    int rc = TRLDA_OK;
    if (m->direct.connected && world > 1) {
        // the gather buffer is this step's half of the exported region (peers write into it)
        if (world != m->direct.world || dp->rank != m->direct.rank)
            return fail(TRLDA_ERR_ARG, "rank / world differ from trlda_model_dp_direct_connect's");
        if (dp->slot > m->direct.max_slot)
            return fail(TRLDA_ERR_ARG, "mini-batch needs larger slots than the direct exchange region has "
                                       "(trlda_model_dp_direct_alloc)");
        m->dp_gather_direct = static_cast<double *>(m->direct.region);   // (the half: per E-step)
    } else {
        m->dp_gather_direct = nullptr;
        rc = grow(&m->dp_gather_own, &m->cap_dp_gather, std::max<size_t>(dp->slot * (size_t)world, 1));
        if (rc)
            return rc;
    }
    m->dp_gather = m->dp_gather_direct ? m->dp_gather_direct : m->dp_gather_own;
    // the static index of this (mini-batch, cut points): kept by the batch
    std::vector<int64_t> sig;
    sig.push_back(K);
    sig.push_back((int64_t)dp->slot);
    sig.push_back((int64_t)dp->tw_off);
    for (int32_t c : dp->cuts)
        sig.push_back(c);
    trlda_batch *bb = const_cast<trlda_batch *>(b);
    if (bb->dp_sig != sig && b->nnz > 0) {
        if (!bb->dp_wsrc) {
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&bb->dp_wsrc), (size_t)b->nnz * sizeof(int32_t)));
            if (hipMalloc(reinterpret_cast<void **>(&bb->dp_wrow), (size_t)b->nnz * sizeof(int32_t)) != hipSuccess) {
                (void)hipFree(bb->dp_wsrc);
                bb->dp_wsrc = nullptr;
                return fail(TRLDA_ERR_HIP, "hipMalloc failed");
            }
        }
        if ((rc = batch_begin(m, b)))
            return rc;
        constexpr int T = 256;
        trlda::DpCuts cuts{};
        for (int r = 0; r <= world; ++r)
            cuts.at[r] = dp->cuts[(size_t)r];
        hipLaunchKernelGGL(trlda::factor_index_kernel<T>, dim3((b->B + T / 64 - 1) / (T / 64)), dim3(T), 0,
                           m->stream, b->B, world, b->indptr, b->wrank, cuts, dp->slot,
                           (int)(dp->slot / (size_t)K), dp->tw_off, bb->dp_wsrc, bb->dp_wrow);
        HIP_TRY(hipGetLastError());
        bb->dp_sig = sig;
    }
    // Word-sharded M-step: the ranks' ranges of the vocabulary, balanced by what a word costs the
    // statistics kernel (its entries, a constant per active word, a little for the others).  A
    // property of (mini-batch, world): every rank computes the same cut points.  Needs a transport
    // for ranges of unequal size: RCCL (grouped broadcasts) or the host's all-gather-v hook.
    dp->word_sharded = world > 1 && m->word_sharding && fused_update_available(m) &&
                       (m->allgatherv_hook || (dp->comm && !m->allgather_hook && !m->dp_gather_direct)) &&
                       (int)b->wptr_host.size() == m->V + 1;
    if (dp->word_sharded && bb->ws_world != world) {
        const int V = m->V;
        std::vector<int64_t> cost((size_t)V + 1, 0);
        for (int w = 0; w < V; ++w) {
            const int len = b->wptr_host[(size_t)w + 1] - b->wptr_host[(size_t)w];
            cost[(size_t)w + 1] = cost[(size_t)w] + (len > 0 ? 4 + len : 1);
        }
        bb->ws_wcuts.assign((size_t)world + 1, 0);
        bb->ws_acuts.assign((size_t)world + 1, 0);
        bb->ws_lcuts.assign((size_t)world + 1, 0);
        bb->ws_vcuts.assign((size_t)world + 1, 0);
        int w = 0, na = 0, nl = 0, nv = 0;
        for (int r = 1; r <= world; ++r) {
            const int64_t target = r == world ? cost[(size_t)V] : cost[(size_t)V] * r / world;
            while (w < V && (r == world || cost[(size_t)w + 1] <= target)) {
                const int len = b->wptr_host[(size_t)w + 1] - b->wptr_host[(size_t)w];
                na += len > 0;
                nl += len > b->long_len;
                nv += len > b->seg_len;
                ++w;
            }
            bb->ws_wcuts[(size_t)r] = w;
            bb->ws_acuts[(size_t)r] = na;
            bb->ws_lcuts[(size_t)r] = nl;
            bb->ws_vcuts[(size_t)r] = nv;
        }
        bb->ws_world = world;
    }
    return TRLDA_OK;
}

// Word-sharded M-step: every rank has written lambda[:, wcuts[rank] .. wcuts[rank + 1]) and
// receives the other ranks' ranges IN PLACE -- the replicas are bitwise equal afterwards, every
// column computed once, by its owner (src/onlinelda.cpp:99-100 / src/batchlda.cpp:60 across ranks).
// RCCL: one grouped launch of `world` broadcasts (ranges of unequal size; ncclAllGather wants equal
// counts); or the host's own transport (trlda_model_set_allgatherv).
using nccl_broadcast_fn = int (*)(const void *, void *, size_t, int, int, void *, hipStream_t);
using nccl_group_fn = int (*)();
int dp_exchange_lambda(trlda_model *m, const trlda_batch *b)
{
    DpContext *dp = m->dp;
    const int world = dp->world;
    const size_t K = (size_t)m->K;
    if (m->allgatherv_hook) {
        std::vector<size_t> offs((size_t)world + 1);
        for (int r = 0; r <= world; ++r)
            offs[(size_t)r] = (size_t)b->ws_wcuts[(size_t)r] * K;
        const int rc = m->allgatherv_hook(m->allgatherv_ctx, m->lambda, offs.data(), dp->rank, world, m->stream);
        if (rc != 0)
            return fail(TRLDA_ERR_HIP, "the all-gather-v hook failed with " + std::to_string(rc));
        return TRLDA_OK;
    }
    static nccl_broadcast_fn bcast = reinterpret_cast<nccl_broadcast_fn>(rccl_symbol("ncclBroadcast"));
    static nccl_group_fn gstart = reinterpret_cast<nccl_group_fn>(rccl_symbol("ncclGroupStart"));
    static nccl_group_fn gend = reinterpret_cast<nccl_group_fn>(rccl_symbol("ncclGroupEnd"));
    if (!bcast || !gstart || !gend)
        return fail(TRLDA_ERR_ARG, "ncclBroadcast / ncclGroupStart / ncclGroupEnd not found: load RCCL "
                                   "(librccl.so) into the process");
    int rc = gstart();
    for (int r = 0; r < world && rc == 0; ++r) {
        const size_t lo = (size_t)b->ws_wcuts[(size_t)r] * K, hi = (size_t)b->ws_wcuts[(size_t)r + 1] * K;
        if (hi > lo)
            rc = bcast(m->lambda + lo, m->lambda + lo, hi - lo, kNcclFloat64, r, dp->comm, m->stream);
    }
    const int rc_end = gend();
    if (rc != 0 || rc_end != 0)
        return fail(TRLDA_ERR_HIP, "ncclBroadcast (word-sharded M-step) failed with ncclResult_t " +
                                       std::to_string(rc != 0 ? rc : rc_end));
    return TRLDA_OK;
}

// all ranks' factors -> every rank (the statistics kernel reads them where they land)
int dp_exchange(trlda_model *m, const trlda_batch *)
{
    DpContext *dp = m->dp;
    double *mine = m->dp_gather + (size_t)dp->rank * dp->slot;
    if (m->dp_gather_direct && dp->world > 1) {
        // direct: this rank's slot into every peer's buffer, then signal and wait (dp_kernels.h)
        if (int rc = ensure_xerr(m))
            return rc;
        ++m->direct.step;                            // (estep_device chose this step's half)
        constexpr int T = 256;
        const size_t offset = (size_t)(m->direct.step & 1ull) * (size_t)dp->world * m->direct.max_slot +
                              (size_t)dp->rank * dp->slot;
        const size_t count = (dp->slot + 1) & ~(size_t)1;
        const unsigned gx = (unsigned)std::max<size_t>(1, std::min<size_t>((count / 2 + T - 1) / T, 32));
        hipLaunchKernelGGL(trlda::slot_push_kernel<T>, dim3(gx, (unsigned)(dp->world - 1)), dim3(T), 0,
                           m->stream, m->direct.peers, dp->rank, dp->world, offset, count);
        hipLaunchKernelGGL(trlda::slot_signal_wait_kernel, dim3(1), dim3(trlda::kDpMaxWorld), 0, m->stream,
                           m->direct.peers, dp->rank, dp->world, m->direct.step, m->xerr);
        HIP_TRY(hipGetLastError());
    } else if (m->allgather_hook) {
        const int rc = m->allgather_hook(m->allgather_ctx, mine, m->dp_gather, dp->slot, m->stream);
        if (rc != 0)
            return fail(TRLDA_ERR_HIP, "the all-gather hook failed with " + std::to_string(rc));
    } else if (dp->world > 1) {
        if (!dp->comm)
            return fail(TRLDA_ERR_ARG, "RCCL communicator is NULL");
        static nccl_allgather_fn fn = reinterpret_cast<nccl_allgather_fn>(rccl_symbol("ncclAllGather"));
        if (!fn)
            return fail(TRLDA_ERR_ARG, "ncclAllGather not found: load RCCL (librccl.so) into the process");
        const int rc = fn(mine, m->dp_gather, dp->slot, kNcclFloat64, dp->comm, m->stream);   // in place
        if (rc != 0)
            return fail(TRLDA_ERR_HIP, "ncclAllGather failed with ncclResult_t " + std::to_string(rc));
    }
    return TRLDA_OK;
}

// the small-table path: one launch for row sums + exp(psi(lambda)), topic factors applied by the
// register-resident document kernel (estep_kernels.h, 2b) -- what a batch and a model must be like
bool fused_preamble_possible(const trlda_model *m, const trlda_batch *docs)
{
    // (whatever the documents' lengths: every variant of the K <= 128 document launch applies
    // the topic factors, estep_docs_tiered_kernel)
    return (size_t)m->K * m->V < ((size_t)1 << 22) && docs->B > 0 && m->doc_threads == 0 &&
           m->doc_kernel == TRLDA_DOCS_AUTO && !m->split_preamble && m->K <= trlda::kRegMaxK &&
           !m->lambda_exposed;
}

// may the M-step kernel of an E-step on `docs` leave the next preamble behind?
bool can_emit_next(const trlda_model *m, const trlda_batch *docs)
{
    return m->emit_next_preamble && m->carry_rowsums && fused_preamble_possible(m, docs);
}

// The E-step launch sequence on the model's stream (no synchronisation).  `out` says what the
// statistics stage writes; an M-step in it (out.upd.lambda) needs fused_update_available().
int estep_device(trlda_model *m, const trlda_batch *b, double *gamma_dev, EstepOut &out,
                 int max_iter, double threshold, int32_t *iters_dev,
                 const double *gamma_in_dev = nullptr, const trlda_batch *next = nullptr)
{
    using namespace trlda;
    // data-parallel call: the preamble and the statistics cover the whole mini-batch `b`, the
    // document stage this rank's shard `db` of it
    DpContext *dp = m->dp;
    const trlda_batch *db = dp ? dp->shard : b;
    const int K = m->K, V = m->V, B = db->B;
    const size_t KV = (size_t)K * V;
    if (b->V != V || db->V != V)
        return fail(TRLDA_ERR_SHAPE, "batch was created for a different vocabulary size");
    if (m->eb.active)
        return fail(TRLDA_ERR_ARG, "an empirical-Bayes step is on its way (its alpha is not on the device "
                                   "yet): trlda_model_online_eb_finish first");
    if (b->device != m->device || db->device != m->device)
        return fail(TRLDA_ERR_ARG, "batch and model live on different devices");
    int rc = ensure_batch_workspace(m, b);
    if (!rc)
        rc = batch_begin(m, b);
    if (!rc && dp)
        rc = batch_begin(m, db);
    if (rc)
        return rc;
    const bool atomic = m->sstats_mode == TRLDA_SSTATS_ATOMIC;
    if (dp && atomic)
        return fail(TRLDA_ERR_ARG, "the factor exchange needs the segmented statistics mode");
    if (dp)
        next = nullptr;
    if (out.upd.lambda && !fused_update_available(m))
        return fail(TRLDA_ERR_ARG, "internal: fused M-step requested where it is not available");
    if (dp && m->dp_gather_direct && dp->world > 1) {
        // direct exchange: every E-step is a step of its own, in the other half of the region.
        // The step is COUNTED where the push is enqueued (dp_exchange): a rank whose call is
        // refused before that must not be a step ahead of its peers
        m->dp_gather = m->dp_gather_direct +
                       (size_t)((m->direct.step + 1ull) & 1ull) * (size_t)dp->world * m->direct.max_slot;
    }
    double *sstats_dev = out.upd.sstats;
    m->last_split_wgs = 0;
    m->last_merged = false;
    if (m->timing && (rc = stamp(m)))
        return rc;

    // 1. row sums of lambda (lda.cpp:172) -- unless the kernel that wrote lambda left them
    // behind (rs_valid).  Small tables: 64 blocks whose partials the consumers add up
    // themselves (saves a launch).  Large ones: the streaming kernel, then one small kernel
    // combines the block partials.
    const bool big = KV >= ((size_t)1 << 22);
    const bool trust = !m->lambda_exposed;
    const bool carried = rowsums_carried(m);
    int G = std::min(big ? kMaxRowsumBlocks - 1 : trlda::kRowsumBlocks, std::max(1, V / 32));
    const double *partial_in = m->partial;
    // Small table and every document in the register-resident kernel's range: kernels 1 and
    // 2 become one launch and the topic factors exp(-psiSum) are applied by the document
    // kernel (estep_kernels.h, 2b) -- as long as no row sum can be so small that exp(-psi(sum))
    // overflows (rs_floor: a bound the host keeps through every update)
    // (data-parallel: a property of the WHOLE mini-batch, which every rank holds -- the ranks must
    // agree on what the gathered factors mean: with the fused preamble a rank publishes c_k
    // expElogtheta and keeps exp(psi(lambda)) unnormalised, without it plain expElogtheta)
    const bool fused = fused_preamble_possible(m, b) && trust && m->rs_floor >= kFusedRowsumFloor;
    // Big table, single-orientation document kernel: the previous trust-region iteration's M-step
    // left exp(psi(lambda)) of THIS batch's words behind and the row sums came with it (carried):
    // no exp_elog_beta_kernel -- the document kernel applies the K topic factors
    const bool fused_big = !fused && big && carried && trust && !dp && !atomic && B > 0 && m->big_emit &&
                           m->u_left.valid && m->u_left.batch_id == b->id &&
                           m->u_left.version == m->lambda_version && m->rs_floor >= kFusedRowsumFloor &&
                           K > kRegMaxK && K <= kWideMaxK && m->doc_threads == 0 &&
                           m->doc_kernel == TRLDA_DOCS_AUTO && !m->dense_preamble;
    m->u_left.valid = false;
    m->last_preamble_fused = fused || fused_big;
    if (!fused && carried && (rc = resolve_carry(m)))   // everything else wants one row of sums
        return rc;
    m->eeb_cur = m->eeb;
    // this batch's preamble may have been left behind by the kernel that wrote lambda (the
    // M-step inside the statistics kernel: UpdateOut::u_out / group_rows) ...
    // Merged launch (estep_merged.h): how many workgroups the documents of this launch take, and
    // whether helper workgroups ride on it
    const size_t xcount_m = (size_t)db->n_xrows * (size_t)(max_iter + 1) * (size_t)K;
    const bool will_split = db->max_n > 128 && m->split_docs && db->n_wg > 0 && db->split_pays &&
                            max_iter > 0 && xcount_m * sizeof(double) <= ((size_t)256 << 20);
    // K <= 32: a wave per document, eight documents per workgroup (estep_kernels.h, estep_docs_small_body).
    // A throughput form: a lone wave takes 46-55 us for a document the eight-wave body finishes in 30, but a
    // CU holds eight such documents -- taken where the documents would not fit the chip one per CU
    // (K = 10: 512 documents 15.4 against 8.4 M docs/s, 6400 documents 38.5 against 9.7 M; 100
    // documents 3.7 against 5.6 M: profiles/r06_small_k.txt), or when asked for
    int cus_now = 256;
    (void)hipDeviceGetAttribute(&cus_now, hipDeviceAttributeMultiprocessorCount, m->device);
    // Documents of more than 128 words -- they lead the batch's sorted order -- keep a workgroup each, in
    // front of the waves' workgroups of the same (tiered) launch: a batch does not lose the form to its
    // long documents (CU-time: a long one 27-50 us, a short one 46 / 8 -- the form pays as long as most
    // are short; split documents keep their segments' workgroups, and the waves find their documents
    // through the batch's order and CSR arrays instead of the padded rows)
    int n_long = 0, n_long_wgs = 0;                  // ... and the workgroups they take (split: a segment each)
    if (db->max_n > 128)
        for (; n_long < B && db->sorted_len[(size_t)n_long] > 128; ++n_long) {
            const int n = db->sorted_len[(size_t)n_long];
            const int c = (n + trlda::kSplitSegN - 1) / trlda::kSplitSegN;
            n_long_wgs += (will_split && n > trlda::kSplitMinN && c <= trlda::kSplitMaxSeg) ? c : 1;   // (batch_index.cpp)
        }
    const bool small = (m->small_k > 0 || (m->small_k < 0 && B - n_long > cus_now)) && K <= 32 && B > 0 &&
                       (n_long == 0 || n_long * 2 <= B) && !atomic && m->doc_threads == 0 &&
                       m->doc_kernel == TRLDA_DOCS_AUTO;
    const int small_wgs = n_long_wgs + (B - n_long + 7) / 8;
    const int doc_wgs = small ? small_wgs : will_split ? db->n_wg : B;
    // (... a matter of speed: the helpers run UNDER the documents only when the documents leave
    // CUs free -- a document workgroup fills one.  Safety does not depend on it: estep_merged.h)
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->device);
    const bool merged_capable = m->merged_launch > 0 && fused && !atomic && !dp && B > 0 &&
                                doc_wgs <= std::min(trlda::kMergedMaxDocWgs, cus - 32);
    bool handed = fused && carried && m->carry_pending && m->next_pre.valid &&
                  m->next_pre.version == m->lambda_version &&
                  (m->next_pre.all || m->next_pre.batch_id == b->id);
    // (block rows still to be added up need this launch's combine workgroups: without them the
    // preamble launch below adds them up -- and fills exp(psi(lambda)) once more)
    if (handed && m->next_pre.raw && !merged_capable)
        handed = false;
    const bool comb = handed && m->next_pre.raw;
    if (handed) {
        partial_in = comb ? m->carry_rows : m->upd_groups;
        G = comb ? m->carry_n : m->next_pre.n;
        if (m->timing && ((rc = stamp(m)) || (rc = stamp(m))))
            return rc;
    }
    // ... or prepared under the previous call's document kernel
    const bool prefetched = fused && !carried && m->prefetch.valid && m->prefetch.batch_id == b->id &&
                            m->prefetch.version == m->lambda_version &&
                            m->prefetch.dense == m->dense_preamble;
    int cur_buf = -1;
    if (prefetched) {
        cur_buf = m->prefetch.buf;
        m->eeb_cur = m->eeb_pp[cur_buf];
        partial_in = m->partial_pp[cur_buf];
        G = m->prefetch.G;
        if (m->timing && ((rc = stamp(m)) || (rc = stamp(m))))
            return rc;
    }
    m->prefetch.valid = false;

    // Deferred statistics (trlda_model::pending).  launch_ok: this call is a plain E-step whose one
    // document launch can take helper workgroups; defer_self: its own statistics wait for the next
    // call; carry: its launch forms the statistics the call before left pending.  What cannot be
    // carried is launched as the kernel of its own, now -- before anything of this call overwrites
    // what it reads (the exp(psi(lambda)) buffer m->eeb, when this call fills it).
    auto stage_fits = [&](const trlda_batch *x) {
        return K % 2 == 0 && m->pair_gathers && x->long_len == trlda::kLongWord && x->B <= 256 &&
               x->max_list <= 256 && x->n_active > 0 && x->V == V && x->device == m->device;
    };
    const bool launch_ok = m->deferred_stats && fused && !dp && !atomic && !out.upd.lambda && sstats_dev &&
                           B > 0 && m->doc_threads == 0 && K <= kRegMaxK && m->doc_kernel == TRLDA_DOCS_AUTO &&
                           fused_update_available(m) && !comb;
    const bool defer_self = launch_ok && stage_fits(b);
    const bool writes_eeb = !(prefetched || handed);
    bool carry = false;
    if (m->pending.valid) {
        carry = launch_ok && stage_fits(m->pending.batch) && !(m->pending.eeb_buf < 0 && writes_eeb);
        if (!carry && (rc = flush_pending(m)))
            return rc;
    }
    m->last_deferred = false;
    m->last_carried = carry;

    if (fused && !prefetched && !handed) {
        constexpr int TP = 512;
        int wpb = 0, GC = 0;
        const double *carry_rows = nullptr, *carry_base = nullptr;
        int carry_n = 0;
        if (carried && m->carry_pending && (m->carry_n > trlda::kRowsumBlocks || m->carry_base)) {
            // block partials left by the statistics kernel: kCarryBlocks workgroups of this
            // launch add them up (with the inactive words' share) into carry_out
            G = 0;
            GC = kCarryBlocks;
            carry_rows = m->carry_rows; carry_base = m->carry_base; carry_n = m->carry_n;
        } else if (carried) {
            G = 0;                                   // no row-sum workgroups: all of them fill
        } else {
            G = std::min(trlda::kRowsumBlocks, std::max(1, V / 32));
            wpb = (V + G - 1) / G;
            G = (V + wpb - 1) / wpb;
        }
        const bool dense = m->dense_preamble;
        const size_t total = dense ? KV : (size_t)K * (size_t)b->n_active;
        // G workgroups add up the row sums, the others fill exp(psi(lambda)): 256 in all, one
        // per CU (a 1024-thread workgroup of this kernel fills a CU's registers)
        const int lead = G + GC;
        const int GP = lead + (int)std::max<size_t>(1, std::min<size_t>((total + TP - 1) / TP, (size_t)(256 * (1024 / TP) - lead)));
        hipLaunchKernelGGL(preamble_fused_kernel<TP>, dim3(GP), dim3(TP), 0, m->stream, K, V, G, wpb,
                           total, m->lambda, m->partial, m->eeb_cur, dense ? nullptr : b->active, GC,
                           carry_rows, carry_n, carry_base, m->carry_out);
        HIP_TRY(hipGetLastError());
        if (GC > 0) {
            partial_in = m->carry_out;
            G = kCarryBlocks;
            // lambda has not changed: its row sums are now these eight rows
            m->carry_rows = m->carry_out; m->carry_n = kCarryBlocks; m->carry_base = nullptr;
        } else if (carried && m->carry_pending) {
            partial_in = m->carry_rows;              // few enough rows for the document kernel
            G = m->carry_n;
        } else if (carried) {
            partial_in = m->rs_full;
            G = 1;
        }
        if (m->timing && ((rc = stamp(m)) || (rc = stamp(m))))
            return rc;
    } else if (!fused) {
    if (carried) {                                   // (resolved above)
        partial_in = m->rs_full;
        G = 1;
    } else if (big && stream_available(m)) {
        rc = rowsums_from_scratch(m);
        if (rc)
            return rc;
        partial_in = m->rs_full;
        G = 1;
    } else {
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
    // dense preamble was asked for -- or, with exp(psi(lambda)) left behind by the M-step, the K
    // topic factors only (psi_sum: psi of the row sums, the sums, exp(-psi))
    if (fused_big) {
        hipLaunchKernelGGL(trlda::topic_factors_kernel<256>, dim3((K + 255) / 256), dim3(256), 0, m->stream, K,
                           partial_in, m->psi_sum);
        HIP_TRY(hipGetLastError());
    } else {
        constexpr int TE = 1024;
        const bool dense = m->dense_preamble;
        const size_t total = dense ? KV : (size_t)K * (size_t)b->n_active;
        size_t blocks = (total + TE - 1) / TE;
        int GE = (int)std::max<size_t>(1, std::min<size_t>(blocks, 256));
        size_t lds = (size_t)K * 9 * sizeof(double);
        rc = ensure_dynamic_lds(reinterpret_cast<const void *>(exp_elog_beta_kernel<TE>), lds);
        if (rc)
            return rc;
        hipLaunchKernelGGL(exp_elog_beta_kernel<TE>, dim3(GE), dim3(TE), lds, m->stream, K, total, G,
                           m->lambda, partial_in, m->psi_sum, m->eeb_cur,
                           dense ? nullptr : b->active);
        HIP_TRY(hipGetLastError());
    }
    if (m->timing && (rc = stamp(m)))
        return rc;
    }

    // 3. per-document fixed point (lda.cpp:174-204)
    if (atomic) {
        if (!sstats_dev)
            return fail(TRLDA_ERR_ARG, "internal: atomic statistics need an sstats buffer");
        HIP_TRY(hipMemsetAsync(sstats_dev, 0, KV * sizeof(double), m->stream));  // lda.cpp:169
    }
    if (B > 0) {
        DocKernelArgs a;
        a.K = K; a.B = B;
        a.stamps = nullptr;
        a.meta_i4 = 1; a.xbuf = nullptr; a.xerr = nullptr;
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
        a.indptr = db->indptr; a.ids = db->ids; a.cnts = db->cnts;
        a.eeb = m->eeb_cur; a.alpha = m->alpha;
        a.gamma = gamma_dev; a.gamma_in = gamma_in_dev ? gamma_in_dev : gamma_dev;
        if (!gamma_in_dev && gamma_dev == m->gamma && m->gamma0_src) {
            a.gamma_in = m->gamma0_src;              // drawn ahead (device_gamma_now_or_ahead)
            m->gamma0_src = nullptr;
        }
        a.epg = m->epg; a.tw_csr = m->tw_csr;
        a.wrank = db->wrank; a.tw_word = m->tw_word;
        if (dp) {
            // straight into this rank's slot of the gathered buffer: expElogtheta rows, then the
            // weights in CSR order
            a.epg = m->dp_gather + (size_t)dp->rank * dp->slot;
            a.tw_word = a.epg + dp->tw_off;
            a.wrank = nullptr;
        }
        if (defer_self) {
            // kept apart from what the next call writes: its launch reads them (estep_merged.h)
            const int which = 1 - m->dfr_cur;
            if ((rc = ensure_deferred_workspace(m, db, which)))
                return rc;
            m->dfr_cur = which;
            a.epg = m->dfr_epg_base[which] + K;
            a.tw_word = m->dfr_tw[which];
        }
        a.sstats_acc = atomic ? sstats_dev : nullptr;
        a.max_iter = max_iter; a.threshold = threshold; a.iters_out = iters_dev;
        a.partial = (fused || fused_big) ? partial_in : nullptr;
        a.scale_in = fused_big ? m->psi_sum : prefetched ? m->scale_pp[cur_buf] : nullptr;
        a.done_counter = nullptr; a.scale_wait = nullptr; a.done_target = 0;
        a.go_flags = nullptr; a.n_go = 0; a.block0 = 0;
        a.docs_per_wg = small ? 8 : 1;
        a.small_block0 = small ? n_long_wgs : 0;
        a.small_first = small ? n_long : 0;
        a.epoch = m->merged_epoch + 1u;              // (of this launch, if it turns out to be merged)
        if (comb) {                                  // finished by workgroups of this launch
            a.scale_in = m->scale_comb;
            a.scale_wait = m->sync_flags + (size_t)trlda::kMergedMaxHelpers * trlda::kMergedFlagStride;
        }
        a.G = G;
        a.scale_out = fused ? m->psi_sum : nullptr;
        const int Kp = K | 1;

        // K <= 128: ONE launch for all documents, ordered by decreasing length -- the register
        // kernel when none has more than 128 words, else the tiered kernel, whose workgroups
        // take the variant their own document needs (estep_wide.h).  Everything else: below.
        int n_reg = 0;
        if (m->doc_threads == 0 && K <= kRegMaxK && m->doc_kernel == TRLDA_DOCS_AUTO)
            n_reg = B;

        // 128 < K <= 512, or K <= 128 with more than 192 words: registers in one orientation
        // (estep_wide.h); it takes every document the register tier does not.  Beyond 512
        // topics (or on request): the general kernel, slice in LDS when it fits, else streamed.
        const bool wide = m->doc_threads == 0 && K <= kWideMaxK && m->doc_kernel != TRLDA_DOCS_GENERAL;
        const int n_stream = B - n_reg;

        if (B - n_reg >= n_reg)
            m->last_doc_kernel = wide ? "estep_docs_wide_kernel" : "estep_docs_kernel";
        else
            m->last_doc_kernel = db->max_n <= 128 ? "estep_docs_reg_kernel" : "estep_docs_tiered_kernel";
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
            const int fit = (int)(((size_t)kLdsDynBytes - fixed) / ((size_t)(64 * KS + 1) * sizeof(double)));
            const int lds_rows = std::max(0, std::min(fit, db->max_n - kWideWaves * jw));
            const size_t lds_bytes = wide_lds_doubles(KS, lds_rows) * sizeof(double);
            a.n_cap = 0;
            a.Kp = 64 * KS;
            a.order = db->order;
#define TRLDA_LAUNCH_WIDE(KSV)                                                             \
    do {                                                                                   \
        auto kern = fused_big ? estep_docs_wide_kernel<KSV, true> : estep_docs_wide_kernel<KSV, false>; \
        if ((rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds_bytes)))     \
            return rc;                                                                     \
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
            int n_fit = fixed >= (size_t)kLdsDynBytes
                            ? 0
                            : (int)(((size_t)kLdsDynBytes - fixed) /
                                    ((size_t)(Kp + 2) * sizeof(double)));
            int gen_cap = std::min(db->max_n, n_fit);
            size_t lds_bytes = docs_lds_bytes(K, Kp, gen_cap, T);
            if (lds_bytes > (size_t)kLdsDynBytes)
                return fail(TRLDA_ERR_ARG,
                            "num_topics too large for the document kernel's LDS layout");
            a.n_cap = gen_cap;
            a.Kp = Kp;
            a.order = db->order;
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
            // One launch.  No document over 128 words: the register kernel.  Else the tiered
            // kernel: per workgroup up to 128 / 144 / 192 words in the register variants, beyond
            // that one orientation with the words past the registers in LDS rows (as many as fit)
            // or streamed from L2.  (Running two variants on two streams was measured and lost:
            // the event fork/join costs ~12 us; one launch per variant one behind the other made
            // a batch with a single 193-word document 2.5 times slower.)
            a.order = db->order;
            a.pad_meta = db->pad_meta;
            a.pad_ids = db->pad_ids;
            const bool tiered = db->max_n > 128;
            const int KS = (K + kWave - 1) / kWave;              // 1 or 2
            // Documents of more than 192 words: split over several workgroups that exchange K
            // sums per iteration (estep_docs_reg_body<0, true>) -- the batch brought the
            // per-workgroup layout; one row of K doubles per (segment, iteration), NaN before
            // the launch.  Not with an exchange buffer beyond 256 MB (max_iter in the thousands):
            // those batches keep one workgroup per document.
            int n_wgs = small ? small_wgs : n_reg;              // document workgroups of the launch
            const size_t xcount = (size_t)db->n_xrows * (size_t)(max_iter + 1) * (size_t)K;
            if (tiered && m->split_docs && db->n_wg > 0 && db->split_pays && max_iter > 0 &&
                xcount * sizeof(double) <= ((size_t)256 << 20)) {
                if ((rc = grow(&m->xbuf, &m->cap_xbuf, xcount)))
                    return rc;
                if ((rc = ensure_xerr(m)))
                    return rc;
                HIP_TRY(hipMemsetAsync(m->xbuf, 0xFF, xcount * sizeof(double), m->stream));   // NaN
                a.pad_meta = db->seg_meta;
                a.pad_ids = db->seg_ids;
                a.meta_i4 = 2;
                a.xbuf = m->xbuf;
                a.xerr = m->xerr;
                n_wgs = small ? small_wgs : db->n_wg;   // (the waves' workgroups behind the long documents' segments)
            }
            const bool split = a.meta_i4 == 2;
            m->last_split_wgs = small ? n_long_wgs - n_long : n_wgs - n_reg;
            if (small)
                m->last_doc_kernel = "estep_docs_small_body";
            int lds_rows = 0;
            size_t lds_bytes = kRegLdsBytes;
            // (LDS rows only for what the single-orientation body cannot hold in registers:
            // with split documents that is a document beyond the split range)
            if (tiered && db->max_n > (split ? kSplitSegN * kSplitMaxSeg : kRegMaxN)) {
                const int jw = KS == 1 ? wide_cfg<1>::JW : wide_cfg<2>::JW;
                const size_t fixed = wide_lds_doubles(KS, 0) * sizeof(double);
                const int fit = (int)(((size_t)kLdsDynBytes - fixed) / ((size_t)(64 * KS + 1) * sizeof(double)));
                lds_rows = std::max(0, std::min(fit, db->max_n - kWideWaves * jw));
                lds_bytes = std::max(lds_bytes, wide_lds_doubles(KS, lds_rows) * sizeof(double));
            }
            const void *kern_ptr = !tiered ? reinterpret_cast<const void *>(estep_docs_reg_kernel<0>)
                                   : KS == 1 ? reinterpret_cast<const void *>(estep_docs_tiered_kernel<1>)
                                             : reinterpret_cast<const void *>(estep_docs_tiered_kernel<2>);
            if ((rc = ensure_dynamic_lds(kern_ptr, lds_bytes)))
                return rc;
            // the next batch's preamble as extra workgroups of this launch (PreArgs): when the
            // caller announced it, it fits the same path, and lambda is not about to change
            PreArgs pre{};
            pre.n_docs = n_wgs;
            if (next && fused && n_reg == B && !out.upd.lambda && !atomic && next->V == V &&
                next->device == m->device && next->B > 0 && !m->lambda_exposed && m->prefetch_next) {
                if (!m->eeb_pp[0]) {
                    for (int i = 0; i < 3 && !rc; ++i) {
                        rc = dev_alloc(&m->eeb_pp[i], KV);
                        if (!rc) rc = dev_alloc(&m->partial_pp[i], (size_t)kRowsumBlocks * K);
                        if (!rc) rc = dev_alloc(&m->scale_pp[i], 3 * (size_t)K);
                        if (!rc && hipMemsetAsync(m->eeb_pp[i], 0, KV * sizeof(double), m->stream) != hipSuccess)
                            rc = fail(TRLDA_ERR_HIP, "hipMemsetAsync failed");
                    }
                    if (rc)
                        return rc;
                }
                if ((rc = batch_begin(m, next)))
                    return rc;
                // (not the buffer this batch reads, nor the one the carried statistics read)
                int nbuf = 0;
                while (nbuf == cur_buf || (carry && nbuf == m->pending.eeb_buf))
                    ++nbuf;
                const bool dense = m->dense_preamble;
                pre.K = K; pre.V = V;
                pre.G = std::min(kRowsumBlocks, std::max(1, V / 32));
                pre.wpb = (V + pre.G - 1) / pre.G;
                pre.G = (V + pre.wpb - 1) / pre.wpb;
                pre.total = dense ? KV : (size_t)K * (size_t)next->n_active;
                // ~8 elements per thread: at K = 100, B = 200 about 100 fill workgroups beside the 64
                // that add up the rows -- three rounds on the 56 CUs the documents leave free.
                // (Letting them start a few microseconds late, so as not to disturb the documents'
                // staging, was measured: every s_sleep step made the launch longer.)
                // (a deferred launch's helpers: 16 per thread, all loads in flight -- estep_kernels.h, FAT)
                const size_t per = (size_t)kRegThreads * (carry ? 16 : 8);
                pre.nb = pre.G + (int)std::max<size_t>(1, std::min<size_t>((pre.total + per - 1) / per, 448));
                pre.lambda = m->lambda;
                pre.partial = m->partial_pp[nbuf];
                pre.u = m->eeb_pp[nbuf];
                pre.active = dense ? nullptr : next->active;
                pre.c_out = m->scale_pp[nbuf];
                pre.c_counter = m->group_counter + kUpdGroups;
                m->prefetch.valid = true;
                m->prefetch.batch_id = next->id;
                m->prefetch.version = m->lambda_version;
                m->prefetch.buf = nbuf;
                m->prefetch.G = pre.G;
                m->prefetch.dense = dense;
            }
            // The statistics (and what rides on them: M-step, row sums, the next preamble) as
            // workgroups of this launch (estep_merged.h): small batches whose words' lists the
            // stage's waves hold in one piece, pairs of topics per lane
            // (a long list goes to the stage's sixteen chunks of at most 16 entries: 256 entries --
            // a list can be longer than B when ids repeat within documents)
            const bool merged_stats = !launch_ok &&
                                      merged_capable && n_wgs == doc_wgs && K % 2 == 0 && m->pair_gathers &&
                                      b->long_len == trlda::kLongWord && b->B <= 256 && b->max_list <= 256 &&
                                      b->n_active > 0 &&
                                      fused_update_available(m) &&
                                      (out.upd.lambda ? out.active_only
                                                      : sstats_dev != nullptr && m->merged_launch >= 2);
            m->last_merged = merged_stats;
            MergedArgs mg{};
            mg.first = n_wgs + pre.nb;
            if (comb) {
                mg.n_comb = trlda::kMergedComb;
                mg.c_rows = m->carry_rows; mg.c_n = m->carry_n; mg.c_base = m->carry_base;
                mg.c_out = m->scale_comb; mg.c_ready = m->sync_counters + 1;
                mg.c_target = m->c_ready_total + (unsigned int)trlda::kMergedComb;
                mg.c_flags = m->sync_flags + (size_t)trlda::kMergedMaxHelpers * trlda::kMergedFlagStride;
            }
            mg.epoch = a.epoch; mg.n_docs = n_wgs;

            {
                static const bool want_stamps = std::getenv("TRLDA_MERGED_STAMPS") != nullptr;
                if (want_stamps && !m->merged_stamps && dev_alloc(&m->merged_stamps, 3 * 1024) == TRLDA_OK)
                    (void)hipMemset(m->merged_stamps, 0, 3 * 1024 * sizeof(unsigned long long));
                mg.tstamps = want_stamps ? m->merged_stamps : nullptr;
            }
            mg.go_flags = m->sync_flags;
            mg.K = K; mg.V = V;
            if (merged_stats) {
                mg.N_short = b->n_short; mg.N_long = b->n_long;
                const int room = std::min(cus, trlda::kMergedMaxHelpers);
                mg.n_long = std::min(mg.N_long, room / 2);
                mg.n_short = mg.N_short > 0
                                 ? std::max(1, std::min((mg.N_short + 8 * trlda::kMergedNW - 1) / (8 * trlda::kMergedNW),
                                                        room - mg.n_long))
                                 : 0;
                // the stage in its 16-slot form (estep_merged.h, merged_stats_slots): a workgroup per
                // item of a length class, when the batch's items fit the stage's flags
                static const bool slots_off = [] {
                    const char *e = std::getenv("TRLDA_MERGED_SLOTS");
                    return e && std::atoi(e) == 0;
                }();
                for (int c = 0; c < 4; ++c) {
                    mg.cls_short[c] = b->cls_short[c];
                    mg.cls_long[c] = b->cls_long[c];
                }
                const int s_short = trlda::merged_slot_short_items(mg.cls_short);
                const int s_long = trlda::deferred_long_items(mg.cls_long);
                mg.slots = !slots_off && s_short + s_long <= trlda::kMergedMaxHelpers;
                if (mg.slots) {
                    mg.n_short = s_short;
                    mg.n_long = s_long;
                }
                mg.desc = reinterpret_cast<const int4 *>(b->mdesc);
                mg.wdoc = b->wdoc; mg.tw_word = m->tw_word; mg.epg = m->epg; mg.eeb = m->eeb_cur;
                mg.active_flag = (!out.upd.lambda && !out.active_only) ? b->active_flag : nullptr;
                mg.docs_done = m->sync_counters;
                mg.docs_target = m->docs_done_total + (unsigned int)n_wgs;
                a.done_counter = m->sync_counters;
                a.done_target = mg.docs_target;
                a.go_flags = m->sync_flags;
                a.n_go = mg.n_short + mg.n_long;
                // what the stage writes: as launch_sstats_update decides it for the kernel of its own
                const bool positive = out.upd.lambda && mstep_keeps_positive(m, out.upd, b);
                const bool emit = out.emit_next && out.upd.lambda && out.upd.partial && positive;
                if (out.upd.lambda)
                    m->lambda_positive = positive;
                out.upd.u_out = emit ? m->eeb : nullptr;
                out.upd.group_rows = nullptr; out.upd.group_base = nullptr;
                out.upd.group_counter = nullptr; out.upd.group_size = 1;
                out.groups = emit ? mg.n_short + mg.n_long : 0;
                out.raw_rows = emit;
                out.partial_rows = mg.n_short + mg.n_long;
                mg.o = out.upd;
            }
            const int n_help = mg.n_comb + mg.n_short + mg.n_long;
            if (carry) {
                // the statistics of the call before as workgroups of this launch (estep_merged.h,
                // deferred statistics): from the buffers that call kept apart, into its caller's array
                const trlda_batch *pb = m->pending.batch;
                MergedArgs dg{};
                dg.first = n_wgs + pre.nb;
                dg.K = K; dg.V = V;
                dg.N_short = pb->n_short; dg.N_long = pb->n_long;
                for (int c = 0; c < 4; ++c) {
                    dg.cls_short[c] = pb->cls_short[c];
                    dg.cls_long[c] = pb->cls_long[c];
                }
                dg.n_short = trlda::deferred_short_items(dg.cls_short);    // a workgroup per item
                dg.n_long = trlda::deferred_long_items(dg.cls_long);
                dg.desc = reinterpret_cast<const int4 *>(pb->mdesc);
                dg.wdoc = pb->wdoc; dg.tw_word = m->pending.tw_word; dg.epg = m->pending.epg;
                dg.eeb = m->pending.eeb;
                dg.active_flag = pb->active_flag;
                dg.o = trlda::UpdateOut{};
                dg.o.sstats = m->pending.sstats;
                {
                    static const bool want_stamps = std::getenv("TRLDA_MERGED_STAMPS") != nullptr;
                    if (want_stamps && !m->deferred_stamps && dev_alloc(&m->deferred_stamps, 3 * 3072) == TRLDA_OK)
                        (void)hipMemset(m->deferred_stamps, 0, 3 * 3072 * sizeof(unsigned long long));
                    dg.tstamps = want_stamps ? m->deferred_stamps : nullptr;
                }
                const void *dk = !tiered ? reinterpret_cast<const void *>(estep_docs_reg_deferred_kernel<0>)
                                 : KS == 1 ? reinterpret_cast<const void *>(estep_docs_tiered_deferred_kernel<1>)
                                           : reinterpret_cast<const void *>(estep_docs_tiered_deferred_kernel<2>);
                if ((rc = ensure_dynamic_lds(dk, lds_bytes)))
                    return rc;
                // helper workgroups: one per item, at most one per CU (estep_merged.h, deferred_helper:
                // they take their items from a counter)
                static const int helpers_env = [] {
                    const char *e = std::getenv("TRLDA_DEFER_HELPERS");
                    return e ? std::max(1, std::atoi(e)) : 0;
                }();
                const int n_items = pre.nb + dg.n_short + dg.n_long;
                // As many helpers as the documents leave CUs free: a helper that is only dispatched when
                // the documents end finds the list empty after a round trip to the counter, and the
                // launch waits for it (32.1 us per step with one helper per CU, 31.1 with 56:
                // profiles/r05_deferred_ab.txt) -- more of them only where the list is long for the free
                // CUs, and one per CU when the documents fill the chip (then everything runs behind them)
                const int free_cus = cus - n_wgs;
                const int wanted = free_cus >= 32 ? std::max(free_cus, (n_items + 5) / 6) : cus;
                const int H = std::min(n_items, helpers_env > 0 ? helpers_env : wanted);
                dg.work_counter = m->sync_counters + 32;        // (a cache line of its own)
                dg.work_base = m->defer_work_total;
                // every helper resident from the start: its first item is its own index, the counter
                // hands out the items from H on (one fetch per item in all: each helper's last one finds
                // nothing); else every item comes from the counter (n_items + H fetches)
                static const bool no_static = std::getenv("TRLDA_DEFER_STATIC0") != nullptr;   // (A/B)
                dg.first_static = (!no_static && free_cus >= 32 && H <= free_cus) ? H : 0;
                m->defer_work_total += (unsigned int)(dg.first_static ? n_items : n_items + H);
                const dim3 grid((unsigned)(n_wgs + H));
                CallClock launch_clock;
                if (!tiered)
                    hipLaunchKernelGGL(estep_docs_reg_deferred_kernel<0>, grid, dim3(kRegThreads), lds_bytes,
                                       m->stream, a, pre, dg);
                else if (KS == 1)
                    hipLaunchKernelGGL(estep_docs_tiered_deferred_kernel<1>, grid, dim3(kRegThreads), lds_bytes,
                                       m->stream, a, pre, lds_rows, dg);
                else
                    hipLaunchKernelGGL(estep_docs_tiered_deferred_kernel<2>, grid, dim3(kRegThreads), lds_bytes,
                                       m->stream, a, pre, lds_rows, dg);
                launch_clock.to(5);
                m->pending.valid = false;
                const_cast<trlda_batch *>(pb)->pending_in = nullptr;
                if ((rc = batch_end(m, pb)))
                    return rc;
            } else if (n_help > 0) {
                if ((rc = ensure_xerr(m)))
                    return rc;
                mg.xerr = m->xerr;
                a.xerr = m->xerr;
                // auxiliary workgroups (estep_merged.h, AuxArgs): the next fresh gamma0, when this call's
                // draw asked for it and the CUs the documents leave free take the workgroups in ONE round
                // of at most kAuxMaxChunks chunks each (~2 us a chunk under ~33 us of documents)
                trlda::AuxArgs aux{};
                constexpr int kAuxMaxChunks = 6;
                const int aux_room = cus - n_wgs - mg.n_comb - pre.nb;
                if (m->aux_draw_req.valid && aux_room >= 24 && merged_stats) {
                    const long long total = m->aux_draw_req.total;
                    const long long chunks = (total + trlda::kAuxChunk - 1) / trlda::kAuxChunk;
                    const int cpw = (int)((chunks + aux_room - 1) / aux_room);
                    const int n_draw = (int)((chunks + cpw - 1) / cpw);
                    // (which buffer: not the one this call's E-steps read their gamma0 from)
                    const int which = (a.gamma_in == m->gspec[0] && m->gspec[0]) ? 1 : 0;
                    const uint32_t *mt_stride = nullptr, *mt_total = nullptr;
                    if (cpw <= kAuxMaxChunks && n_draw < 256 &&
                        !(rc = rng_aux_matrices(m->device, (long long)cpw * trlda::kAuxChunk, &mt_stride)) &&
                        !(rc = rng_aux_matrices(m->device, total, &mt_total)) && mt_stride && mt_total &&
                        !(rc = grow(&m->gspec[which], &m->cap_gspec[which], (size_t)total))) {
                        aux.draw.n = n_draw;
                        aux.draw.passes = 100; aux.draw.cpw = cpw;
                        aux.draw.total = total; aux.draw.divisor = 100.;
                        aux.draw.out = m->gspec[which];
                        aux.draw.mt_stride = mt_stride; aux.draw.mt_total = mt_total;
                        // the host stream moves on ahead of its turn (host_rng.cpp: whoever touches the
                        // generator before the claim puts it back, and the draw is repeated in its turn)
                        const uint64_t token = trlda_host::rng_speculate_begin();
                        trlda_host::rng_current_window(aux.draw.w0.w);
                        trlda_host::rng_advance((uint64_t)(100 * total));
                        m->spec.valid = true; m->spec.token = token; m->spec.which = which;
                        m->spec.total = total; m->spec.lo = 0; m->spec.hi = total;
                        m->spec.in_launch = true;
                        ++m->inlaunch_draws;
                    }
                    if (rc)
                        return rc;
                }
                m->aux_draw_req.valid = false;
                // ... and the decay of the words outside the batch (K even: merged_stats), which the
                // caller would otherwise launch behind this kernel
                if (out.inact_wanted && m->aux_decay && merged_stats && aux_room >= 24 && out.inact_a != 0.0 &&
                    b->n_active < V) {
                    const trlda::StreamGeom g = trlda::stream_geometry(K, V);
                    if (g.vec == 2 && g.P <= 64) {
                        aux.inact.n_vb = g.G;
                        aux.inact.K = K; aux.inact.V = V; aux.inact.P = g.P; aux.inact.cpb = g.cpb;
                        aux.inact.a = out.inact_a; aux.inact.b = out.inact_b;
                        aux.inact.active_flag = b->active_flag;
                        aux.inact.lambda = m->lambda;
                        // (not rows [0, 64) of m->partial: this launch's documents may be adding up the
                        // row sums its preamble left there)
                        aux.inact.part_static = m->partial + (size_t)trlda::kStreamMaxBlocks * K;
                        out.inact_rows = g.G;
                        out.inact_part = aux.inact.part_static;
                        ++m->inlaunch_decays;
                    }
                }
                aux.n_items = aux.draw.n + aux.inact.n_vb;
                aux.n = std::min(aux.n_items, std::max(aux_room, 0));
                aux.work_counter = m->sync_counters + 48;        // (a cache line of its own)
                aux.work_base = m->aux_work_total;
                m->aux_work_total += (unsigned int)aux.n_items;  // (one fetch per item in all: each
                                                                 //  workgroup's last one finds nothing)
                mg.first += aux.n;
                const void *mk = !tiered ? reinterpret_cast<const void *>(estep_docs_reg_merged_kernel<0>)
                                 : KS == 1 ? reinterpret_cast<const void *>(estep_docs_tiered_merged_kernel<1>)
                                           : reinterpret_cast<const void *>(estep_docs_tiered_merged_kernel<2>);
                if ((rc = ensure_dynamic_lds(mk, lds_bytes)))
                    return rc;
                a.block0 = mg.n_comb;                // the topic-factor workgroups come first
                const dim3 grid((unsigned)(mg.first + n_help));
                if (!tiered)
                    hipLaunchKernelGGL(estep_docs_reg_merged_kernel<0>, grid, dim3(kRegThreads), lds_bytes,
                                       m->stream, a, pre, mg, aux);
                else if (KS == 1)
                    hipLaunchKernelGGL(estep_docs_tiered_merged_kernel<1>, grid, dim3(kRegThreads), lds_bytes,
                                       m->stream, a, pre, lds_rows, mg, aux);
                else
                    hipLaunchKernelGGL(estep_docs_tiered_merged_kernel<2>, grid, dim3(kRegThreads), lds_bytes,
                                       m->stream, a, pre, lds_rows, mg, aux);
                if (merged_stats)
                    m->docs_done_total += (unsigned int)n_wgs;
                m->c_ready_total += (unsigned int)mg.n_comb;
                ++m->merged_epoch;
            } else if (!tiered)
                hipLaunchKernelGGL(estep_docs_reg_kernel<0>, dim3(n_wgs + pre.nb), dim3(kRegThreads),
                                   lds_bytes, m->stream, a, pre);
            else if (KS == 1)
                hipLaunchKernelGGL(estep_docs_tiered_kernel<1>, dim3(n_wgs + pre.nb), dim3(kRegThreads),
                                   lds_bytes, m->stream, a, pre, lds_rows);
            else
                hipLaunchKernelGGL(estep_docs_tiered_kernel<2>, dim3(n_wgs + pre.nb), dim3(kRegThreads),
                                   lds_bytes, m->stream, a, pre, lds_rows);
            if (pre.nb > 0 && (rc = batch_end(m, next)))
                return rc;
            HIP_TRY(hipGetLastError());
        }
    }
    if (m->timing && (rc = stamp(m)))
        return rc;

    if (dp) {                                        // dp_kernels.h
        if ((rc = batch_end(m, db)) || (rc = dp_exchange(m, b)))
            return rc;
    }

    // 4. sufficient statistics (lda.cpp:207-217), with the M-step and the next row sums where
    // the caller asked for them
    if (atomic) {
        size_t blocks = (KV / 2 + kDenseThreads * 4 - 1) / (kDenseThreads * 4);
        int GF = (int)std::max<size_t>(1, std::min<size_t>(blocks, 256 * 8));
        hipLaunchKernelGGL((elementwise_stream_kernel<kDenseThreads, FinishOp>), dim3(GF),
                           dim3(kDenseThreads), 0, m->stream, KV, FinishOp{m->eeb_cur, sstats_dev});
    } else if (m->last_merged && !dp) {
        // (statistics: workgroups of the document launch above)
    } else if (defer_self) {
        // (statistics: workgroups of the NEXT call's document launch, or flush_pending)
        // (a batch remembers ONE model whose pending statistics read it: another model's go first)
        if (b->pending_in && b->pending_in != m && (rc = flush_pending(b->pending_in)))
            return rc;
        m->pending.valid = true;
        m->pending.batch = b;
        m->pending.sstats = sstats_dev;
        m->pending.epg = m->dfr_epg_base[m->dfr_cur] + K;
        m->pending.tw_word = m->dfr_tw[m->dfr_cur];
        m->pending.eeb = m->eeb_cur;
        m->pending.eeb_buf = prefetched ? cur_buf : -1;
        const_cast<trlda_batch *>(b)->pending_in = m;
        m->last_deferred = true;
    } else if (dp && dp->word_sharded && out.upd.lambda && !out.upd.sstats) {
        // Word-sharded M-step (data-parallel): statistics + M-step for THIS RANK's range of the
        // vocabulary only -- every rank holds all factors, so any rank can form any word's
        // statistics; a word's entries are added in document order by whoever owns it -- then the
        // ranks exchange the lambda columns they wrote.  No row sums and no exp(psi(lambda)) ride
        // along: the next E-step's preamble forms them from the exchanged lambda.
        EstepOut part = out;
        part.sliced = true;
        part.upd.partial = nullptr;
        part.emit_next = false;
        const int r = dp->rank;
        if (out.active_only) {
            part.slice_lo = b->ws_acuts[(size_t)r];
            part.slice_n = b->ws_acuts[(size_t)r + 1] - part.slice_lo;
        } else {
            part.slice_lo = b->ws_wcuts[(size_t)r];
            part.slice_n = b->ws_wcuts[(size_t)r + 1] - part.slice_lo;
        }
        part.long_lo = b->ws_lcuts[(size_t)r];
        part.long_n = b->ws_lcuts[(size_t)r + 1] - part.long_lo;
        part.vl_lo = b->ws_vcuts[(size_t)r];
        part.vl_n = b->ws_vcuts[(size_t)r + 1] - part.vl_lo;
        if ((rc = sstats_update_device(m, b, part)) || (rc = dp_exchange_lambda(m, b)))
            return rc;
        out.no_rows = true;
        out.groups = 0;
        out.partial_rows = 0;
        m->last_word_sharded = true;
    } else if (fused_update_available(m)) {
        rc = sstats_update_device(m, b, out);
        if (rc)
            return rc;
    } else {
        if (!sstats_dev)
            return fail(TRLDA_ERR_ARG, "internal: the statistics need an sstats buffer");
        // one wavefront per word; 16 words per workgroup for small K, 8 from K = 256 on
#define TRLDA_LAUNCH_SSTATS(TS)                                                            \
    do {                                                                                   \
        constexpr int wpb = TS / kWave;                                                    \
        const int G_short = (V + wpb - 1) / wpb;                                           \
        size_t lds = (size_t)wpb * K * sizeof(double);                                     \
        auto kern = sstats_words_kernel<TS>;                                               \
        if ((rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds)))           \
            return rc;                                                                     \
        hipLaunchKernelGGL(kern, dim3(G_short + b->n_long), dim3(TS), lds, m->stream, K, V, \
                           G_short, b->long_len, b->wptr, dp ? b->dp_wrow : b->wdoc, b->long_words, \
                           dp ? trlda::TwView{m->dp_gather, b->dp_wsrc}                    \
                              : trlda::TwView{m->tw_word, nullptr},                        \
                           dp ? m->dp_gather : m->epg, m->eeb_cur, sstats_dev);            \
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
    m->aux_draw_req.valid = false;                   // (a launch that could not carry it: drawn in its turn)
    return batch_end(m, b);
}

// sstats-only form (the E-step entry points)
int estep_device(trlda_model *m, const trlda_batch *b, double *gamma_dev, double *sstats_dev,
                 int max_iter, double threshold, int32_t *iters_dev,
                 const double *gamma_in_dev = nullptr)
{
    EstepOut out(sstats_dev);
    return estep_device(m, b, gamma_dev, out, max_iter, threshold, iters_dev, gamma_in_dev);
}

template <class Op>
int launch_elementwise(trlda_model *m, size_t total, const Op &op)
{
    size_t blocks = (total / 2 + kDenseThreads * 4 - 1) / (kDenseThreads * 4);
    int G = (int)std::max<size_t>(1, std::min<size_t>(blocks, 256 * 8));
    hipLaunchKernelGGL((trlda::elementwise_stream_kernel<kDenseThreads, Op>), dim3(G),
                       dim3(kDenseThreads), 0, m->stream, total, op);
    HIP_TRY(hipGetLastError());
    return TRLDA_OK;
}

int blend_device(trlda_model *m, const double *lambda_prime, const double *sstats, double rho,
                 double eta, double scale)
{
    invalidate_rowsums(m);
    m->lambda_positive = false;                 // (not tracked through this path)
    m->rs_floor = rho * m->V * eta;     // (1 - rho) lambda' >= 0 whatever lambda' is
    return launch_elementwise(m, (size_t)m->K * m->V,
                              trlda::BlendOp{rho, eta, scale, lambda_prime, sstats, m->lambda});
}

int wordcounts_device(trlda_model *m, const trlda_batch *b, double *wc)
{
    int rcb = batch_begin(m, b);
    if (rcb)
        return rcb;
    HIP_TRY(hipMemsetAsync(wc, 0, (size_t)m->V * sizeof(double), m->stream));
    if (b->nnz > 0) {
        size_t blocks = ((size_t)b->nnz + kDenseThreads - 1) / kDenseThreads;
        int G = (int)std::min<size_t>(blocks, 256 * 8);
        hipLaunchKernelGGL(trlda::wordcount_kernel<kDenseThreads>, dim3(G), dim3(kDenseThreads), 0,
                           m->stream, b->nnz, b->ids, b->cnts, wc);
        HIP_TRY(hipGetLastError());
    }
    return batch_end(m, b);
}

// The streaming pass over the words of lambda outside the batch (stream_kernels.h):
// block partials of the row sums go to m->partial rows [0, G) (inactive words) and
// [kStreamMaxBlocks, kStreamMaxBlocks + G) (active words, ACT_TRINIT only).
template <int ACT, bool SAVE_ALL>
int launch_inactive_update(trlda_model *m, double a, double b, double rho, double eta, double coef,
                           const uint8_t *flags, const double *wc, const double *src,
                           double *lambda_prime, int *G_out, const int32_t *wc32 = nullptr)
{
    constexpr int T = trlda::kStreamThreads;
    const trlda::StreamGeom g = trlda::stream_geometry(m->K, m->V);
    const size_t lds = (size_t)g.cpb * m->K * sizeof(double);
    double *ps = m->partial, *pa = m->partial + (size_t)trlda::kStreamMaxBlocks * m->K;
    if (g.vec == 2)
        hipLaunchKernelGGL((trlda::inactive_update_stream_kernel<T, 2, ACT, SAVE_ALL>), dim3(g.G),
                           dim3(T), lds, m->stream, m->K, m->V, g.P, g.cpb, a, b, rho, eta, coef,
                           flags, wc, wc32, src, m->lambda, lambda_prime, ps, pa);
    else
        hipLaunchKernelGGL((trlda::inactive_update_stream_kernel<T, 1, ACT, SAVE_ALL>), dim3(g.G),
                           dim3(T), lds, m->stream, m->K, m->V, g.P, g.cpb, a, b, rho, eta, coef,
                           flags, wc, wc32, src, m->lambda, lambda_prime, ps, pa);
    HIP_TRY(hipGetLastError());
    *G_out = g.G;
    return TRLDA_OK;
}

// lambda = (1 - rho) lambda' (+row) rho (eta + coef * wordcounts), lambda' a separate buffer
// (onlinelda.cpp:85-86); leaves the row sums of the new lambda behind
int tr_init_wc_device(trlda_model *m, const double *wc, const double *lambda_prime, double rho,
                      double eta, double coef)
{
    invalidate_rowsums(m);
    m->lambda_positive = false;                 // (not tracked through this path)
    m->rs_floor = rho * m->V * eta;
    if (stream_available(m)) {
        int G = 0;
        int rc = launch_inactive_update<trlda::ACT_TRINIT, false>(m, 0., 0., rho, eta, coef, nullptr,
                                                                  wc, lambda_prime, nullptr, &G);
        if (!rc)
            rc = combine_rowsums(m, m->partial + (size_t)trlda::kStreamMaxBlocks * m->K, G, nullptr,
                                 m->rs_full);
        if (!rc)
            m->rs_valid = true;
        return rc;
    }
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

// lambda was replaced from host memory: the carried row sums are void, and the smallest row
// sum (what decides whether the fused preamble is safe) is taken from the host copy.  Only
// small tables can use the fused preamble; for the others 0 ("unknown") is as good.
void note_host_lambda(trlda_model *m, const double *host_lambda)
{
    invalidate_rowsums(m);
    m->rs_floor = 0.0;
    m->lambda_positive = false;
    const size_t K = (size_t)m->K, V = (size_t)m->V;
    if (K * V >= ((size_t)1 << 22) || m->K > trlda::kRegMaxK) {
        // big tables: only "every element > 0" is needed (the M-step may then leave exp(psi(lambda))
        // behind through the call-free exp_digamma_positive, mstep_keeps_positive) -- one pass on a
        // few host threads while the copy to the device runs
        const size_t total = K * V;
        const int nt = (int)std::max<size_t>(1, std::min<size_t>(16, total >> 20));
        std::vector<char> ok((size_t)nt, 1);
        trlda_host::host_pool().run(nt, [&](int t) {
            const size_t i0 = total * (size_t)t / (size_t)nt, i1 = total * (size_t)(t + 1) / (size_t)nt;
            bool good = true;
            for (size_t i = i0; i < i1; ++i)
                good &= host_lambda[i] > 0.0;        // (NaN: false)
            ok[(size_t)t] = good;
        });
        bool positive = true;
        for (char c : ok)
            positive = positive && c;
        m->lambda_positive = positive;
        return;
    }
    std::vector<double> sum(K, 0.0);
    bool positive = true;
    for (size_t w = 0; w < V; ++w)
        for (size_t k = 0; k < K; ++k) {
            sum[k] += host_lambda[w * K + k];
            positive = positive && host_lambda[w * K + k] > 0.0;     // (NaN: false)
        }
    m->lambda_positive = positive;
    double lo = sum[0];
    for (size_t k = 1; k < K; ++k)
        lo = std::min(lo, sum[k]);
    m->rs_floor = lo > 0.0 ? 0.999 * lo : 0.0;   // NaN compares false -> 0
}

// after a synchronisation: did a segment of a split document give up waiting for its peers
// (estep_docs_reg_body<0, true>)?  Results of that launch are void.
int check_split_exchange(trlda_model *m)
{
    if (!m->xerr_host || !*m->xerr_host)
        return TRLDA_OK;
    *m->xerr_host = 0;
    return fail(TRLDA_ERR_HIP, "a wait inside a launch gave up: a document split over several workgroups "
                               "for one of its segments (trlda_model_set_split_docs(model, 0) keeps "
                               "every document on one workgroup), the statistics stage of a merged "
                               "launch for its documents or the documents for their topic factors "
                               "(trlda_model_set_merged_launch(model, 0)), or the direct slot exchange "
                               "for a peer's signal; the results of that call are void");
}

// wait for the model's stream; a call whose exchange gave up has no results
int sync_model(trlda_model *m)
{
    HIP_TRY(hipStreamSynchronize(m->stream));
    int rc = check_split_exchange(m);
    for (trlda_model *l : m->lane)                   // (joined before: the stream waited for theirs)
        if (l) {
            const int r = check_split_exchange(l);
            if (!rc)
                rc = r;
        }
    return rc;
}

constexpr int kLaneCalEvery = 1024;                  // (lane steps between two looks at the lanes: trlda_model::lane_cal)
constexpr int kLaneSoloLead = 2, kLaneSoloSteps = 16, kLaneDuoLead = 8, kLaneSoloAfter = 64;

// Stream lanes: what the lanes hold is brought to an end -- their pending statistics launched on
// their streams -- and the model's stream continues behind all of it.
int lanes_join(trlda_model *m)
{
    if (!m->lanes_live)
        return TRLDA_OK;
    if (m->lane_cal.phase == 2 || m->lane_cal.phase == 6) {
        // (a window with a join in it measures the join: again -- a caller whose stretches are
        // shorter than the windows never gets a measurement, and keeps its lanes; one whose stretches
        // end before the one-lane stretch can begin gets the window without it)
        auto &cal = m->lane_cal;
        const bool broken = cal.n > 0;
        if (cal.phase == 6 && !broken && ++cal.short_stretches >= 1) {
            cal.phase = 2;
            cal.lead = 0;
        }
        else if (cal.phase == 2 && cal.solo_valid)
            cal.phase = 6;                           // (the look starts again, with its one-lane stretch)
        cal.n = 0;
        cal.solo_valid = cal.solo_valid && cal.phase == 2;
        if (broken && ++cal.tries > 8) {
            cal.phase = 4;
            cal.next_check = m->lane_steps + kLaneCalEvery;
        }
    }
    m->lanes_live = false;
    m->lane_turn = 0;
    int rc = TRLDA_OK;
    for (int p = 0; p < 2; ++p) {
        trlda_model *l = m->lane[p];
        if (!l)
            continue;
        if (m->lane_span_open[p])                    // (the documents' launches: not the flush below)
            (void)hipEventRecord(m->lane_span[p][1], l->stream);
        const int r = flush_pending(l);
        if (!rc)
            rc = r;
        if (hipEventRecord(m->lane_out[p], l->stream) != hipSuccess ||
            hipStreamWaitEvent(m->stream, m->lane_out[p], 0) != hipSuccess) {
            // (no way to order the streams: wait here)
            (void)hipStreamSynchronize(l->stream);
        }
        m->lane_calls[p] = 0;
        m->lane_writes[p].clear();
    }
    return rc;
}

// keep_pending: the caller is the next E-step of a deferred stream (it decides itself whether its
// launch carries the pending statistics); everybody else finds none outstanding
// keep_lanes: the caller is the next E-step of a stream that runs through the lanes
int check_model(const trlda_model *m, bool keep_pending = false, bool keep_lanes = false)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "model is NULL");
    // (whatever another library of the process left in this thread's "last error" -- torch's
    // allocator probes stream capture and leaves "operation not permitted when stream is
    // capturing" behind -- is not an error of the launches this entry point is about to check)
    (void)hipGetLastError();
    int rc = use_device(m->device);
    if (!rc && m->pending.valid && !keep_pending)
        rc = flush_pending(const_cast<trlda_model *>(m));
    if (!rc && m->lanes_live && !keep_lanes)
        rc = lanes_join(const_cast<trlda_model *>(m));
    return rc;
}

}  // namespace

// ===========================================================================
extern "C" {

int trlda_version(void) { return 100; }

int trlda_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

} // extern "C"

namespace {

using trlda::RngSeedWindow;

__global__ void window_seed_kernel(long long S, RngSeedWindow w0, uint32_t *win)
{
    if (threadIdx.x < 31)
        win[(size_t)threadIdx.x * S] = w0.w[threadIdx.x];
}

// the matrices live on the device once per (process, device, segment length)
int rng_device_matrices(int device, int L, const uint32_t **out)
{
    static std::mutex mu;
    static std::map<std::tuple<pid_t, int, int>, uint32_t *> all;
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_tuple(getpid(), device, L);
    auto it = all.find(key);
    if (it == all.end()) {
        // row-major for window_level_kernel, then the transposes for window_direct_kernel (its 31
        // lanes of a window read one column entry each: consecutive words)
        std::vector<uint32_t> h = trlda_host::rng_level_matrices(L, trlda::kRngLevels);
        const size_t n = h.size();
        h.resize(2 * n);
        for (size_t mtx = 0; mtx < n / 961; ++mtx)
            for (int i = 0; i < 31; ++i)
                for (int j = 0; j < 31; ++j)
                    h[n + mtx * 961 + (size_t)j * 31 + i] = h[mtx * 961 + (size_t)i * 31 + j];
        uint32_t *d = nullptr;
        int rc = dev_alloc(&d, h.size());
        if (rc)
            return rc;
        HIP_TRY(hipMemcpy(d, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        it = all.emplace(key, d).first;
    }
    *out = it->second;
    return TRLDA_OK;
}

int rng_aux_matrices(int device, long long L, const uint32_t **out)
{
    static std::mutex mu;
    static std::map<std::tuple<pid_t, int, long long>, uint32_t *> all;
    constexpr size_t kMaxShapes = 64;                // (a stream of mini-batches has a few sizes)
    *out = nullptr;
    if (L <= 0 || L > 0x7fffffff)
        return TRLDA_OK;
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_tuple(getpid(), device, L);
    auto it = all.find(key);
    if (it == all.end()) {
        if (all.size() >= kMaxShapes)
            return TRLDA_OK;
        const std::vector<uint32_t> &h = trlda_host::rng_level_matrices((int)L, 2);
        std::vector<uint32_t> t(h.size());
        for (size_t mtx = 0; mtx < h.size() / 961; ++mtx)
            for (int i = 0; i < 31; ++i)
                for (int j = 0; j < 31; ++j)
                    t[mtx * 961 + (size_t)j * 31 + i] = h[mtx * 961 + (size_t)i * 31 + j];
        uint32_t *d = nullptr;
        int rc = dev_alloc(&d, t.size());
        if (rc)
            return rc;
        HIP_TRY(hipMemcpy(d, t.data(), t.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        it = all.emplace(key, d).first;
    }
    *out = it->second;
    return TRLDA_OK;
}

// elements [e_lo, e_hi) only (out_dev compact, e_hi - e_lo values); e_hi < 0: all of them.  The
// stream always advances by passes * total draws.
int sample_gamma_on_device(trlda_model *m, long long total, int passes, double divisor, double *out_dev,
                           long long e_lo, long long e_hi, bool ahead)
{
    if (total <= 0)
        return TRLDA_OK;
    if (!ahead)
        trlda_host::rng_speculation_cancel();        // a draw in its turn: nothing may be ahead of it
    // a draw made ahead runs on the draw stream with scratch of its own
    const hipStream_t stream = ahead ? m->draw_stream : m->stream;
    uint32_t *&win = ahead ? m->rng_win2 : m->rng_win;
    size_t &cap_win = ahead ? m->cap_rng_win2 : m->cap_rng_win;
    double *&vbuf = ahead ? m->rng_vbuf2 : m->rng_vbuf;
    size_t &cap_vbuf = ahead ? m->cap_rng_vbuf2 : m->cap_rng_vbuf;
    if (e_hi < 0)
        e_hi = total;
    constexpr int T = trlda::kRngThreads;
    const long long draws = total * passes;
    const int L = draws < trlda::kRngTinyDraws    ? trlda::kRngSegmentTiny
                  : draws < trlda::kRngSmallDraws ? trlda::kRngSegmentSmall
                                                  : trlda::kRngSegment;
    const long long S = (draws + L - 1) / L;
    // two-level windows for the small segment lengths (rng_kernels.h): the matrices produce the
    // windows of every kRngWalk-th segment, a walk along the recurrence the ones in between
    static const bool walk_off = [] { const char *e = std::getenv("TRLDA_RNG_WALK"); return e && e[0] == '0'; }();
    const bool walk = !walk_off && L != trlda::kRngSegment && S >= 4 * trlda::kRngWalk;
    const int Lm = walk ? L * trlda::kRngWalk : L;              // segment length of the matrix level
    const long long Sm = walk ? (S + trlda::kRngWalk - 1) / trlda::kRngWalk : S;
    const uint32_t *mats = nullptr;
    int rc = rng_device_matrices(m->device, Lm, &mats);
    // |u| of a group of passes: at most ~1 GB at a time -- whole blocks of kRngProductPasses passes (the
    // passes of a block are multiplied before their one logarithm, rng_kernels.h)
    long long group = std::max<long long>(1, std::min<long long>(passes, ((long long)1 << 27) / total));
    if (group < passes)
        group = std::max<long long>(trlda::kRngProductPasses, group / trlda::kRngProductPasses * trlda::kRngProductPasses);
    // small requests (a mini-batch's gamma0): the logarithms are summed where they are formed
    // (rng_kernels.h, draw_sum_kernel), fine windows in segment-major order
    const char *fused_env = std::getenv("TRLDA_RNG_FUSED");      // (read per call: the tests switch it)
    const bool fused_off = fused_env && fused_env[0] == '0';
    const bool fused = !fused_off && walk && L == trlda::kRngSegmentTiny && passes <= trlda::kRngFusedPasses &&
                       group == passes && e_hi > e_lo;
    if (!rc) rc = grow(&win, &cap_win, (size_t)(fused ? 32 : 31) * (size_t)S + (walk ? (size_t)31 * (size_t)Sm : 0));
    if (!rc && !fused) rc = grow(&vbuf, &cap_vbuf, (size_t)group * (size_t)total);
    if (rc)
        return rc;
    uint32_t *mwin = walk ? win + (size_t)(fused ? 32 : 31) * (size_t)S : win;  // where the matrix level writes
    RngSeedWindow w0;
    trlda_host::rng_current_window(w0.w);
    long long unit = 1;
    int levels = 0;
    while (levels < trlda::kRngLevels && unit < Sm) {
        unit *= 16;
        ++levels;
    }
    if (unit < Sm)
        return fail(TRLDA_ERR_ARG, "sampleGamma request too large for the device generator");
    if (Sm < 200000) {
        // few windows: the first three levels (4096 windows) straight from the seed window in one
        // launch, the levels above with one matrix-vector product per window each
        const uint32_t *mats_t = mats + (size_t)trlda::kRngLevels * 15 * 961;
        // (up to 16 384 windows: every level in that launch -- a window past the first 4096 costs
        // four products instead of one, and the launch of its own that the fourth level took goes)
        const bool all_direct = Sm <= 16384;
        const int direct = all_direct ? levels : std::min(levels, 3);
        const long long S0 = all_direct ? Sm : std::min<long long>(Sm, 4096);
        hipLaunchKernelGGL(trlda::window_direct_kernel<T>, dim3((unsigned)((S0 * 32 + T - 1) / T)), dim3(T),
                           0, stream, Sm, S0, direct, w0, mats_t, mwin);
        unit = 4096;
        for (int l = direct; l < levels; ++l, unit *= 16) {
            const long long lo = unit, hi = std::min<long long>(Sm, unit * 16);
            hipLaunchKernelGGL(trlda::window_level_coop_kernel<T>,
                               dim3((unsigned)(((hi - lo) * 32 + T - 1) / T)), dim3(T), 0, stream, Sm,
                               lo, hi, unit, mats_t + (size_t)l * 15 * 961, mwin);
        }
    } else {
        hipLaunchKernelGGL(window_seed_kernel, dim3(1), dim3(64), 0, stream, Sm, w0, mwin);
        unit = 1;
        for (int l = 0; l < levels; ++l, unit *= 16) {
            const long long lo = unit, hi = std::min<long long>(Sm, unit * 16);
            hipLaunchKernelGGL(trlda::window_level_kernel<T>, dim3((unsigned)((hi - lo + T - 1) / T)),
                               dim3(T), 0, stream, Sm, lo, hi, unit, mats + (size_t)l * 15 * 961,
                               mwin);
        }
    }
    if (walk) {
        constexpr int TW = 64;                       // few threads, long walks: spread them out
        const dim3 wgrid((unsigned)((Sm + TW - 1) / TW));
        if (fused)
            hipLaunchKernelGGL((trlda::window_walk_kernel<TW, trlda::kRngSegmentTiny, true>), wgrid, dim3(TW), 0,
                               stream, S, Sm, mwin, win);
        else if (L == trlda::kRngSegmentTiny)
            hipLaunchKernelGGL((trlda::window_walk_kernel<TW, trlda::kRngSegmentTiny>), wgrid, dim3(TW), 0,
                               stream, S, Sm, mwin, win);
        else
            hipLaunchKernelGGL((trlda::window_walk_kernel<TW, trlda::kRngSegmentSmall>), wgrid, dim3(TW), 0,
                               stream, S, Sm, mwin, win);
    }
    if (fused) {
        const long long chunks = (e_hi - e_lo + trlda::kRngSegmentTiny - 1) / trlda::kRngSegmentTiny;
        const bool one_segment = total % trlda::kRngSegmentTiny == 0 && e_lo % trlda::kRngSegmentTiny == 0;
        const int threads = (one_segment ? 1 : 2) * trlda::kRngFusedPasses;
        const size_t lds = (size_t)passes * 33 * sizeof(double) + (size_t)31 * threads * sizeof(uint32_t);
        hipLaunchKernelGGL((trlda::draw_sum_kernel<trlda::kRngSegmentTiny>), dim3((unsigned)chunks),
                           dim3(threads), lds, stream, S, total, e_lo, e_hi, passes, divisor, win, out_dev);
    }
    for (long long p0 = 0; !fused && p0 < passes; p0 += group) {
        const long long p1 = std::min<long long>(passes, p0 + group);
        const long long pos_lo = p0 * total, pos_hi = p1 * total;
        const long long seg_lo = pos_lo / L, seg_hi = (pos_hi + L - 1) / L;
        const dim3 dgrid((unsigned)((seg_hi - seg_lo + T - 1) / T));
        if (L == trlda::kRngSegment)
            hipLaunchKernelGGL((trlda::draw_abs_kernel<T, trlda::kRngSegment>), dgrid, dim3(T), 0,
                               stream, S, seg_lo, std::min(S, seg_hi), pos_lo, pos_hi, total, e_lo,
                               e_hi, win, vbuf);
        else if (L == trlda::kRngSegmentTiny)
            hipLaunchKernelGGL((trlda::draw_abs_kernel<T, trlda::kRngSegmentTiny>), dgrid, dim3(T), 0,
                               stream, S, seg_lo, std::min(S, seg_hi), pos_lo, pos_hi, total, e_lo,
                               e_hi, win, vbuf);
        else
            hipLaunchKernelGGL((trlda::draw_abs_kernel<T, trlda::kRngSegmentSmall>), dgrid, dim3(T), 0,
                               stream, S, seg_lo, std::min(S, seg_hi), pos_lo, pos_hi, total, e_lo,
                               e_hi, win, vbuf);
        if (e_hi > e_lo)
            hipLaunchKernelGGL(trlda::gamma_sum_kernel<T>,
                               dim3((unsigned)((e_hi - e_lo + T - 1) / T)), dim3(T), 0, stream, total,
                               e_lo, e_hi, (int)(p1 - p0), p0 == 0 ? 1 : 0,
                               p1 == passes ? divisor : 1.0, vbuf, out_dev);
    }
    HIP_TRY(hipGetLastError());
    // the host stream moves on by the same number of draws
    trlda_host::rng_advance((uint64_t)draws);
    return TRLDA_OK;
}

}  // namespace

extern "C" {

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

}  // extern "C"

namespace {

constexpr size_t kStageSlots = 24;

// what the ingestion did so far (trlda_debug_ingest_counters): [0] indices built by a worker, [1] taken over
// by their first user, [2] cancelled (destroyed unused), [3] built on the creating thread (no workers),
// [4] hipMalloc of a batch allocation, [5] hipFree of one, [6] waits for a staging buffer held by a
// build, [7] waits for a staging buffer's last upload
std::atomic<long long> g_ingest[8];

// workers that build and upload the indices (TRLDA_INDEX_THREADS, default 4 -- fewer on a small host;
// 0: trlda_batch_create builds on its caller's thread, as in rounds 1-5)
trlda_host::WorkQueue &index_queue()
{
    // (on the heap, per process: a child of fork() has none of the parent's threads -- host_pool's note)
    static trlda_host::WorkQueue *queue = nullptr;
    static pid_t owner = 0;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!queue || owner != getpid()) {
        int n = 4;
        if (const char *e = std::getenv("TRLDA_INDEX_THREADS"))
            n = std::max(0, std::min(std::atoi(e), 32));
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0 && n > 0)
            n = std::max(1, std::min(n, hw - 1));
        queue = new trlda_host::WorkQueue(n);
        owner = getpid();
        if (n > 0)                                   // (no index in the making while the runtime goes away)
            std::atexit([] { index_queue().wait_idle(); });
    }
    return *queue;
}

// publication of a build's end (state, destroy_when_built) and the consumers' wait for it
std::mutex g_build_mu;
std::condition_variable g_build_cv;

// TRLDA_CALL_TIMES=1: where a worker's build spends its time, ns summed over the builds
// (trlda_debug_call_times, [8..15]): [8] submit -> a worker has the job, [9] the grace period, [10] plan +
// fill, [11] the allocation (upload context's lock, cache, hipMalloc), [12] the stream calls, [13] the wait
// for the upload, [14] publication, [15] builds
std::atomic<long long> g_build_ns[8];
inline long long now_ns()
{
    return std::chrono::duration_cast<std::chrono::nanoseconds>(
               std::chrono::steady_clock::now().time_since_epoch()).count();
}

// a staging buffer of at least `bytes` that is nobody else's until stage_release
int uploads_drain(UploadContext &u);

int stage_acquire(UploadContext &u, size_t bytes, UploadContext::Stage **out)
{
    std::unique_lock<std::mutex> lock(u.mu);
    for (;;) {
        UploadContext::Stage *pick = nullptr, *late = nullptr;
        for (auto &sp : u.stages) {
            UploadContext::Stage *st = sp.get();
            if (st->busy)
                continue;
            if (st->ev && hipEventQuery(st->ev) != hipSuccess) {
                (void)hipGetLastError();
                late = late ? late : st;             // (its last upload is still on its way)
                continue;
            }
            if (!pick || (pick->cap < bytes && st->cap > pick->cap))
                pick = st;
            if (pick->cap >= bytes)
                break;                               // (large enough: no need to ask the others' events)
        }
        if (!pick && u.stages.size() < kStageSlots) {
            u.stages.emplace_back(new UploadContext::Stage());
            pick = u.stages.back().get();
        }
        if (!pick && late) {
            ++g_ingest[7];
            HIP_TRY(hipEventSynchronize(late->ev));
            pick = late;
        }
        if (!pick) {
            // every buffer holds an index that has not been uploaded: those that are finished are uploaded
            // now, by this thread (a caller that makes a whole corpus' batches before it uses the first);
            // if none is, a worker is still at one
            ++g_ingest[6];
            if (!u.filled.empty()) {
                lock.unlock();
                if (int rc = uploads_drain(u))
                    return rc;
                lock.lock();
            } else {
                u.stage_cv.wait(lock);
            }
            continue;
        }
        if (pick->cap < bytes) {
            if (pick->host)
                (void)hipHostFree(pick->host);
            pick->host = nullptr;
            pick->cap = 0;
            const size_t want = std::max<size_t>(bytes + bytes / 2, (size_t)1 << 20);
            HIP_TRY(hipHostMalloc(&pick->host, want, hipHostMallocDefault));
            pick->cap = want;
        }
        if (!pick->ev)
            HIP_TRY(hipEventCreateWithFlags(&pick->ev, hipEventDisableTiming));
        pick->busy = true;
        *out = pick;
        return TRLDA_OK;
    }
}

void stage_release(UploadContext &u, UploadContext::Stage *st)   // (u.mu held)
{
    st->busy = false;
    u.stage_cv.notify_all();
}

// A batch's index from its pinned staging buffer into its device allocation, as a kernel of the upload
// stream (16 bytes per lane straight over PCIe: pinned host memory is mapped into the device's address
// space).  Not hipMemcpyAsync: that call takes 2.7 us of host time -- or 31-42, for whole passes over a
// corpus at a time (profiles/r06_corpus_passes.txt: the end-to-end stream at 73-95 us per step instead of
// 40, one pass in three), whatever the runtime waits for inside it.  A launch is 3-4 us, every time.
__global__ __launch_bounds__(256) void blob_copy_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        dst[i] = src[i];
}

// The index of batch `b` from the CSR arrays at the head of its staging buffer, into that buffer
// (csrc/batch_index.cpp): host work only, NO HIP call -- on a worker thread, or on the thread of
// whoever needs the batch first.  (Round 6, first form: the workers also enqueued the upload.  Four
// threads calling hipMemcpyAsync / hipEventRecord on one stream beside the caller's launches: 32-175 us
// per build in those calls, the caller's launch sequence 13 us instead of 8 -- the runtime's locks;
// profiles/r06_e2e_trace.txt.  The uploads are now enqueued by the caller's thread, batch_upload.)
int batch_fill(trlda_batch *b)
{
    UploadContext::Stage *st = static_cast<UploadContext::Stage *>(b->slot);
    char *h = static_cast<char *>(st->host);
    size_t o_ids = 0, o_cnts = 0;
    trlda_host::batch_index_csr_offsets(b->B, b->nnz, &o_ids, &o_cnts);
    const int32_t *indptr = reinterpret_cast<const int32_t *>(h), *ids = reinterpret_cast<const int32_t *>(h + o_ids),
                  *cnts = reinterpret_cast<const int32_t *>(h + o_cnts);
    b->index.reset(new trlda_host::BatchIndex());
    trlda_host::BatchIndex &x = *b->index;
    long long t_mark = g_call_times ? now_ns() : 0;
    int rc = trlda_host::batch_index_plan(b->V, b->B, indptr, ids, cnts, &x);
    if (!rc && (x.total > st->cap || x.o_ids != o_ids || x.o_cnts != o_cnts))
        rc = fail(TRLDA_ERR_ARG, "internal: the index outgrew its staging buffer");
    if (rc) {
        UploadContext &u = upload_context(b->device);
        std::lock_guard<std::mutex> lock(u.mu);
        stage_release(u, st);
        b->slot = nullptr;
        return rc;
    }
    trlda_host::batch_index_fill(&x, indptr, ids, cnts, b->cus, h);
    if (g_call_times)
        g_build_ns[2] += now_ns() - t_mark;

    b->n_active = x.n_active; b->n_long = x.n_long; b->long_len = x.long_len;
    b->split_pays = x.split_pays;
    b->max_list = x.max_list;
    for (int c = 0; c < 4; ++c) {
        b->cls_short[c] = x.cls_short[c];
        b->cls_long[c] = x.cls_long[c];
    }
    b->sorted_len.swap(x.sorted_len);
    b->indptr_host.swap(x.indptr_host);
    b->wptr_host.swap(x.wptr);
    b->long_host.swap(x.long_host);
    b->vl_host.swap(x.vl_host);
    b->vl_first.swap(x.vl_first);
    return TRLDA_OK;
}

// ... and its way to the device: an allocation (from the cache when one fits), ONE copy on the upload
// stream, the events.  On the thread of a caller of the library (trlda_batch_create for the batches made
// before, the E-step a batch is announced to, its first user).
int batch_upload(trlda_batch *b)
{
    UploadContext &u = upload_context(b->device);
    UploadContext::Stage *st = static_cast<UploadContext::Stage *>(b->slot);
    const trlda_host::BatchIndex &x = *b->index;
    const size_t total = x.total;
    long long t_mark = g_call_times ? now_ns() : 0;
    auto mark = [&](int i) {
        if (!g_call_times)
            return;
        const long long n = now_ns();
        g_build_ns[i] += n - t_mark;
        t_mark = n;
    };
    auto give_up = [&](int rc) {
        std::lock_guard<std::mutex> lock(u.mu);
        stage_release(u, st);
        b->slot = nullptr;
        b->index.reset();
        return rc;
    };
    if (hipSetDevice(b->device) != hipSuccess)
        return give_up(fail(TRLDA_ERR_HIP, "hipSetDevice failed"));
    // Under the upload context's lock only what is shared: the allocation and two events.
    UploadContext::Blob blob{nullptr, 0, nullptr};
    bool blob_settled = false;
    {
        std::lock_guard<std::mutex> lock(u.mu);
        if (!u.stream && hipStreamCreateWithFlags(&u.stream, hipStreamNonBlocking) != hipSuccess) {
            stage_release(u, st);
            b->slot = nullptr;
            return fail(TRLDA_ERR_HIP, "hipStreamCreateWithFlags failed");
        }
        // (an allocation whose last reader has finished, if there is one: an upload into an allocation
        // that the previous call's kernels still read would have to wait for them; with a few
        // allocations in rotation the upload of call n + 1 runs under the kernels of call n)
        for (int pass = 0; pass < 2 && !blob.ptr; ++pass)
            for (size_t i = 0; i < u.cache.size(); ++i) {
                const UploadContext::Blob &c = u.cache[i];
                if (c.bytes < total || c.bytes > 4 * total + ((size_t)1 << 20))
                    continue;
                if (pass == 0 && c.done && stream_alive(c.on, c.on_owned) && hipEventQuery(c.done) != hipSuccess)
                    continue;
                if (pass == 1 && u.cache.size() < kBlobCacheMax / 2)
                    break;                           // rather a new allocation than a wait
                blob = c;
                blob_settled = pass == 0;            // (its last reader was SEEN to have finished: nothing to wait for)
                u.cached_bytes -= blob.bytes;
                u.cache.erase(u.cache.begin() + (long)i);
                break;
            }
        (void)hipGetLastError();                     // (hipErrorNotReady of the queries above)
        if (take_event(u, &b->ready) != TRLDA_OK || take_event(u, &b->done) != TRLDA_OK) {
            if (blob.ptr) {                          // (back where it came from)
                u.cache.push_back(blob);
                u.cached_bytes += blob.bytes;
            }
            stage_release(u, st);
            b->slot = nullptr;
            return fail(TRLDA_ERR_HIP, "hipEventCreate failed");
        }
    }
    if (!blob.ptr) {
        size_t want = (size_t)1 << 16;
        while (want < total)
            want <<= 1;
        ++g_ingest[4];
        hipError_t e = hipMalloc(&blob.ptr, want);
        if (e != hipSuccess)
            return give_up(fail(TRLDA_ERR_HIP, std::string("hipMalloc (batch): ") + hipGetErrorString(e)));
        blob.bytes = want;
    }
    b->blob = blob.ptr;
    b->blob_bytes = blob.bytes;
    mark(3);
    hipError_t err = hipSuccess;
    const bool guard = blob.done && stream_alive(blob.on, blob.on_owned);
    const bool guard_wait = guard && !blob_settled;
    if (guard_wait)                                  // the previous owner's last reader, where it may still run
        err = hipStreamWaitEvent(u.stream, blob.done, 0);
    // (else: that stream and its work are gone; the event is dropped)
    mark(4);
    static const bool copy_by_kernel = [] {          // (TRLDA_UPLOAD_COPY=memcpy: hipMemcpyAsync, for A/B)
        const char *e = std::getenv("TRLDA_UPLOAD_COPY");
        return !(e && e[0] == 'm');
    }();
    if (err == hipSuccess && copy_by_kernel) {
        // (the sections are 256-byte aligned and both buffers larger than the index: whole 16-byte words)
        const size_t n16 = (total + 15) / 16;
        // (sixteen workgroups: enough loads in flight for the PCIe link -- four take 67 us per step -- and
        // few enough to find free CUs beside the documents: 128 waited for them, 53-64 us per step)
        static const int copy_wgs = [] {
            const char *e = std::getenv("TRLDA_UPLOAD_COPY_WGS");
            return e ? std::max(1, std::atoi(e)) : 16;
        }();
        // (an index of megabytes -- a whole corpus as one batch -- is not in a tight pipeline: more loads in flight)
        const unsigned grid = (unsigned)std::min<size_t>((n16 + 255) / 256,
                                                         (size_t)copy_wgs * (total > ((size_t)4 << 20) ? 4 : 1));
        hipLaunchKernelGGL(blob_copy_kernel, dim3(grid), dim3(256), 0, u.stream, static_cast<uint4 *>(b->blob),
                           static_cast<const uint4 *>(st->host), n16);
        err = hipGetLastError();
    } else if (err == hipSuccess) {
        err = hipMemcpyAsync(b->blob, st->host, total, hipMemcpyHostToDevice, u.stream);
    }
    mark(5);
    if (err == hipSuccess)
        err = hipEventRecord(st->ev, u.stream);      // (the staging buffer is anybody's once this has passed)
    if (err == hipSuccess)
        err = hipEventRecord(b->ready, u.stream);
    if (err != hipSuccess) {
        (void)hipStreamSynchronize(u.stream);
        (void)hipFree(b->blob);
        b->blob = nullptr;
        return give_up(fail(TRLDA_ERR_HIP, std::string("batch upload: ") + hipGetErrorString(err)));
    }
    mark(4);
    std::lock_guard<std::mutex> lock(u.mu);
    if (guard)
        u.spare(blob.done, blob.on, blob.on_owned);
    stage_release(u, st);
    b->slot = nullptr;
    char *dv = static_cast<char *>(b->blob);
    auto D = [&](size_t o) { return reinterpret_cast<int32_t *>(dv + o); };
    b->indptr = D(x.o_indptr); b->ids = D(x.o_ids); b->cnts = D(x.o_cnts); b->order = D(x.o_order);
    b->wrank = D(x.o_wrank); b->wptr = D(x.o_wptr); b->wdoc = D(x.o_wdoc);
    b->pad_meta = D(x.o_meta); b->pad_ids = D(x.o_pids);
    b->seg_meta = x.n_wg ? D(x.o_smeta) : nullptr; b->seg_ids = x.n_wg ? D(x.o_spids) : nullptr;
    b->n_wg = x.n_wg; b->n_xrows = x.n_xrows;
    b->active = D(x.o_active); b->long_words = D(x.o_long);
    b->active_flag = reinterpret_cast<uint8_t *>(dv + x.o_flag);
    b->wc32 = D(x.o_wc32);
    b->wc32_ok = x.wc32_ok;
    b->cnts_nonneg = x.cnts_nonneg;
    b->mdesc = D(x.o_mdesc);
    b->n_short = x.n_active - x.n_long;
    b->vl_word = D(x.o_vlw); b->vl_task = D(x.o_vlt); b->vl_task_tiled = D(x.o_vltt);
    b->n_vl = x.n_vl; b->n_vl_tasks = x.n_vl_tasks; b->seg_len = x.seg_len;
    b->index.reset();
    u.live.insert(b);
    return TRLDA_OK;
}

// the end of a build's step (the index filled: kFilled; uploaded: kBuilt; either failed: kFailed): its
// status becomes the batch's, whoever waits is woken; a batch whose owner has let go of it in the
// meantime is destroyed here
void batch_publish(trlda_batch *b, int rc, int state_ok)
{
    bool destroy = false;
    {
        std::lock_guard<std::mutex> lock(g_build_mu);
        if (rc) {
            b->build_rc = rc;
            b->build_msg = trlda_last_error();
        }
        b->ticket->state.store(rc ? trlda_batch::kFailed : state_ok, std::memory_order_release);
        destroy = b->destroy_when_built;
        b->destroy_when_built = false;
    }
    g_build_cv.notify_all();
    if (destroy)
        (void)trlda_batch_destroy(b);
}

// kFilled -> kBuilt on this thread, if nobody else is at it; false: somebody else has it (or had)
bool batch_take_upload(trlda_batch *b, const std::shared_ptr<trlda_batch::Ticket> &ticket)
{
    int expect = trlda_batch::kFilled;
    if (!ticket->state.compare_exchange_strong(expect, trlda_batch::kUploading, std::memory_order_acq_rel))
        return false;
    batch_publish(b, batch_upload(b), trlda_batch::kBuilt);
    return true;
}

// every index the workers have finished: uploaded now, by this thread
int uploads_drain(UploadContext &u)
{
    std::vector<std::pair<trlda_batch *, std::shared_ptr<trlda_batch::Ticket>>> todo;
    {
        std::lock_guard<std::mutex> lock(u.mu);
        todo.swap(u.filled);
    }
    // (an entry whose batch was used or destroyed in the meantime: its ticket says so, the batch is not touched)
    for (auto &e : todo)
        (void)batch_take_upload(e.first, e.second);
    return TRLDA_OK;
}

// an announced batch: uploaded now if its index is there; true: it can be announced to a launch
bool batch_announced(const trlda_batch *cb)
{
    if (!cb)
        return false;
    trlda_batch *b = const_cast<trlda_batch *>(cb);
    if (b->ticket->state.load(std::memory_order_acquire) == trlda_batch::kFilled)
        (void)batch_take_upload(b, b->ticket);
    return b->ticket->state.load(std::memory_order_acquire) == trlda_batch::kBuilt;
}

// every entry point that is handed a batch: its index is there and on its way to the device (or the
// build's failure is the call's).  A build that no worker has started on yet is taken over by the caller.
int batch_wait(const trlda_batch *cb)
{
    if (!cb)
        return TRLDA_OK;
    trlda_batch *b = const_cast<trlda_batch *>(cb);
    std::atomic<int> &state = b->ticket->state;
    for (;;) {
        int st = state.load(std::memory_order_acquire);
        if (st == trlda_batch::kBuilt)
            return TRLDA_OK;
        if (st == trlda_batch::kFailed)
            return fail(b->build_rc, "the batch's index could not be built: " + b->build_msg);
        if (st == trlda_batch::kQueued) {
            int expect = trlda_batch::kQueued;
            if (state.compare_exchange_strong(expect, trlda_batch::kBuilding, std::memory_order_acq_rel)) {
                ++g_ingest[1];
                batch_publish(b, batch_fill(b), trlda_batch::kFilled);
            }
        } else if (st == trlda_batch::kFilled) {
            (void)batch_take_upload(b, b->ticket);
        } else if (st == trlda_batch::kBuilding || st == trlda_batch::kUploading) {
            std::unique_lock<std::mutex> lock(g_build_mu);
            g_build_cv.wait(lock, [&] { return state.load(std::memory_order_acquire) != st; });
        } else {
            return fail(TRLDA_ERR_ARG, "the batch was destroyed");
        }
    }
}

}  // namespace

extern "C" {

int trlda_batch_create(trlda_batch **out, int device, int V, int B, const int32_t *indptr,
                       const int32_t *ids, const int32_t *cnts)
{
    if (!out)
        return fail(TRLDA_ERR_ARG, "out is NULL");
    *out = nullptr;
    // On this thread: what can fail because of the arguments -- lengths, word ids (the reference has
    // undefined behaviour there: lda.cpp:108) -- and a copy of the CSR arrays into pinned memory, at
    // the place they have in the index (the caller's arrays are the caller's again on return).  The
    // index itself (csrc/batch_index.cpp) and the upload: on a worker thread.
    int max_n = 0, n_wg = 0;
    int rc = trlda_host::batch_index_check_lengths(V, B, indptr, &max_n, &n_wg);
    if (rc)
        return rc;
    const int64_t nnz = indptr[B];
    if (nnz > 0 && (!ids || !cnts))
        return fail(TRLDA_ERR_ARG, "ids / cnts are NULL");
    if ((rc = trlda_host::batch_index_check_ids(V, nnz, ids)))
        return rc;
    if ((rc = use_device(device)))
        return rc;
    UploadContext &u = upload_context(device);
    UploadContext::Stage *st = nullptr;
    if ((rc = stage_acquire(u, trlda_host::batch_index_size_bound(V, B, nnz, n_wg), &st)))
        return rc;
    {
        char *h = static_cast<char *>(st->host);
        size_t o_ids = 0, o_cnts = 0;
        trlda_host::batch_index_csr_offsets(B, nnz, &o_ids, &o_cnts);
        std::memcpy(h, indptr, ((size_t)B + 1) * 4);
        if (nnz) {
            std::memcpy(h + o_ids, ids, (size_t)nnz * 4);
            std::memcpy(h + o_cnts, cnts, (size_t)nnz * 4);
        }
    }
    trlda_batch *b = new trlda_batch();
    {
        static std::atomic<uint64_t> next_id{1};
        b->id = next_id.fetch_add(1);
    }
    b->device = device; b->V = V; b->B = B; b->nnz = nnz; b->max_n = max_n;
    {
        static std::atomic<int> cus_cache[64];
        int cus = device < 64 ? cus_cache[device].load(std::memory_order_relaxed) : 0;
        if (cus <= 0) {
            cus = 256;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
            if (device < 64)
                cus_cache[device].store(cus, std::memory_order_relaxed);
        }
        b->cus = cus;
    }
    b->slot = st;
    trlda_host::WorkQueue &queue = index_queue();
    if (queue.threads() <= 0) {
        ++g_ingest[3];
        rc = batch_fill(b);
        if (!rc)
            rc = batch_upload(b);
        if (rc) {                                    // (as in rounds 1-5: the failure is this call's)
            delete b;
            return rc;
        }
    } else {
        b->ticket->state.store(trlda_batch::kQueued, std::memory_order_release);
        std::shared_ptr<trlda_batch::Ticket> ticket = b->ticket;
        // (a worker leaves a fresh ticket alone for kGraceUs: a caller that uses its batch at once --
        // do_e_step / update_parameters on a list of tuples -- takes the build over itself and pays
        // what it paid when trlda_batch_create built the index, not a sleeping thread's wake-up on top:
        // 389 against 425-494 us per do_e_step; a pipeline makes its batches hundreds of us ahead)
        const auto start = std::chrono::steady_clock::now() + std::chrono::microseconds(40);
        const long long t_submit = g_call_times ? now_ns() : 0;
        UploadContext *up = &u;
        queue.submit([b, up, ticket, start, t_submit] {
            const long long t_got = g_call_times ? now_ns() : 0;
            while (std::chrono::steady_clock::now() < start &&
                   ticket->state.load(std::memory_order_acquire) == trlda_batch::kQueued)
                __builtin_ia32_pause();
            int expect = trlda_batch::kQueued;       // (taken over by its first user, or cancelled: nothing to do --
            if (ticket->state.compare_exchange_strong(expect, trlda_batch::kBuilding,   //  `b` may be gone)
                                                      std::memory_order_acq_rel)) {
                ++g_ingest[0];
                const long long t_go = g_call_times ? now_ns() : 0;
                const int rc_fill = batch_fill(b);
                const long long t_filled = g_call_times ? now_ns() : 0;
                // (A/B, TRLDA_INDEX_UPLOAD=worker: the worker enqueues the upload too, one worker at a time)
                static const bool by_worker = [] {
                    const char *e = std::getenv("TRLDA_INDEX_UPLOAD");
                    return e && e[0] == 'w';
                }();
                if (by_worker) {
                    static std::mutex upload_mu;
                    int rc_up = rc_fill;
                    if (!rc_up) {
                        std::lock_guard<std::mutex> one(upload_mu);
                        rc_up = batch_upload(b);
                    }
                    batch_publish(b, rc_up, trlda_batch::kBuilt);
                    if (g_call_times) {
                        g_build_ns[0] += t_got - t_submit;
                        g_build_ns[1] += t_go - t_got;
                        g_build_ns[6] += now_ns() - t_filled;
                        g_build_ns[7] += 1;
                    }
                    return;
                }
                batch_publish(b, rc_fill, trlda_batch::kFilled);
                if (!rc_fill) {                      // (`b` may be uploaded, used, even destroyed by now: the
                    std::lock_guard<std::mutex> lock(up->mu);   //  ticket tells whoever takes the entry)
                    up->filled.emplace_back(b, ticket);
                    up->stage_cv.notify_all();
                }
                if (g_call_times) {
                    g_build_ns[0] += t_got - t_submit;
                    g_build_ns[1] += t_go - t_got;
                    g_build_ns[6] += now_ns() - t_filled;
                    g_build_ns[7] += 1;
                }
            }
        });
        // the indices the workers have finished since the last call: uploaded here, on the caller's thread
        (void)uploads_drain(u);                      // (a failed upload is its batch's failure, not this call's)
    }
    *out = b;
    return TRLDA_OK;
}

int trlda_batch_destroy(trlda_batch *b)
{
    if (!b)
        return TRLDA_OK;
    for (;;) {
        // Nobody has started on its index (kQueued), or nobody on its upload (kFilled): that is never
        // done -- the queued job / the list of finished indices find the ticket cancelled.  A thread is
        // at one of the two: that thread destroys the batch when it is done.
        int st = b->ticket->state.load(std::memory_order_acquire);
        if (st == trlda_batch::kQueued || st == trlda_batch::kFilled) {
            if (!b->ticket->state.compare_exchange_strong(st, trlda_batch::kCancelled, std::memory_order_acq_rel))
                continue;
            ++g_ingest[2];
            UploadContext &u = upload_context(b->device);
            {
                std::lock_guard<std::mutex> lock(u.mu);
                if (b->slot)
                    stage_release(u, static_cast<UploadContext::Stage *>(b->slot));
            }
            delete b;
            return TRLDA_OK;
        }
        if (st == trlda_batch::kBuilding || st == trlda_batch::kUploading) {
            std::lock_guard<std::mutex> lock(g_build_mu);
            if (b->ticket->state.load(std::memory_order_acquire) == st) {
                b->destroy_when_built = true;
                return TRLDA_OK;
            }
            continue;
        }
        break;
    }
    // a model's deferred statistics still read this batch: they are launched first (the guard
    // event below then covers them)
    if (b->pending_in && hipSetDevice(b->device) == hipSuccess)
        (void)flush_pending(b->pending_in);
    if (b->dp_wsrc && hipSetDevice(b->device) == hipSuccess) {
        (void)hipFree(b->dp_wsrc);                   // (waits for the device)
        (void)hipFree(b->dp_wrow);
    }
    bool erased = false;
    if (b->blob && hipSetDevice(b->device) == hipSuccess) {
        UploadContext &u = upload_context(b->device);
        std::lock_guard<std::mutex> lock(u.mu);
        // recycle: whoever takes the allocation next waits (on the upload stream) for this
        // batch's last reader; a batch nobody read is guarded by its own upload
        const bool settled = batch_settle(b);
        // the guard: the last reader's record; a batch nobody read is guarded by its own upload
        // (`ready`, recorded on the upload stream, which lives as long as the process); a batch
        // whose last stream is gone needs none -- and its `done` event, if that stream recorded
        // it, is dropped rather than recycled
        u.live.erase(b);
        erased = true;
        hipEvent_t guard = b->used ? (settled ? b->done : nullptr) : b->ready;
        if (b->used)
            u.spare(b->ready);                       // (recorded on the upload stream only)
        else
            u.spare(b->done, b->done_on, b->done_on_owned);
        if (b->used && !settled && b->done && stream_alive(b->done_on, b->done_on_owned))
            u.spare(b->done, b->done_on, b->done_on_owned);   // never recorded, or on a stream that exists
        UploadContext::Blob blob{b->blob, b->blob_bytes, guard};
        if (b->used && settled) {
            blob.on = b->done_on;
            blob.on_owned = b->done_on_owned;
        }
        if (u.cache.size() < kBlobCacheMax && u.cached_bytes + b->blob_bytes <= kBlobCacheBytes) {
            u.cache.push_back(blob);
            u.cached_bytes += b->blob_bytes;
        } else {
            ++g_ingest[5];
            (void)hipFree(b->blob);                  // waits for the device: nothing reads it after
            u.spare(guard, blob.on, blob.on_owned);
        }
    }
    if (!erased) {
        // (whatever kept the branch above from running: no pointer to a deleted batch stays in the
        // upload context's list -- purge_stream_guards walks it)
        UploadContext &u = upload_context(b->device);
        std::lock_guard<std::mutex> lock(u.mu);
        u.live.erase(b);
    }
    delete b;
    return TRLDA_OK;
}

int trlda_batch_num_docs(const trlda_batch *b) { return b ? b->B : 0; }
int64_t trlda_batch_nnz(const trlda_batch *b) { return b ? b->nnz : 0; }
int trlda_batch_max_doc_len(const trlda_batch *b) { return b ? b->max_n : 0; }
// (what the index build establishes: waited for)
int trlda_batch_long_word_len(const trlda_batch *b) { return b && !batch_wait(b) ? b->long_len : 0; }
int trlda_batch_num_long_words(const trlda_batch *b) { return b && !batch_wait(b) ? b->n_long : 0; }
int trlda_batch_num_very_long_words(const trlda_batch *b) { return b && !batch_wait(b) ? b->n_vl : 0; }

// ---- model --------------------------------------------------------------------

}  // extern "C"

namespace {
// (stream_priority != 0: the model's own stream is created with that priority -- the lanes of
// another model, lanes_ensure)
// (lane_of: the model whose lambda and alpha the new one reads instead of allocating its own;
// adopt: a stream made by the caller that becomes the model's own -- lanes_ensure picks its streams)
int model_create(trlda_model **out, int device, int K, int V, int stream_priority, trlda_model *lane_of,
                 hipStream_t adopt = nullptr);
}  // namespace

extern "C" {

int trlda_model_create(trlda_model **out, int device, int K, int V)
{
    return model_create(out, device, K, V, 0, nullptr);
}

}  // extern "C"

namespace {

int model_create(trlda_model **out, int device, int K, int V, int stream_priority, trlda_model *lane_of,
                 hipStream_t adopt)
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
    if (const char *env = std::getenv("TRLDA_PAIR_GATHERS"))   // measurements
        m->pair_gathers = std::atoi(env) != 0;
    if (const char *env = std::getenv("TRLDA_DOC_KERNEL")) {   // measurements: force a document kernel
        const std::string v(env);
        m->doc_kernel = v == "wide" ? TRLDA_DOCS_WIDE : v == "general" ? TRLDA_DOCS_GENERAL
                                                                       : TRLDA_DOCS_AUTO;
    }
    size_t KV = (size_t)K * V;
    if (lane_of) {
        // a stream lane: lambda and alpha are its owner's (no allocation of its own); what a
        // lane does own is its exp E[log beta] buffers -- K x V doubles in `eeb` plus up to three of the
        // same in eeb_pp on small tables -- and the per-batch workspaces: ~4 x 8 K V bytes per lane
        // at most (K = 500, V = 100 000: 0.4 GB for `eeb`, no eeb_pp there), of 288 GB
        m->lambda = lane_of->lambda;
        m->alpha = lane_of->alpha;
        m->lane_owner = lane_of;
    } else {
        rc = dev_alloc(&m->lambda, KV);
        if (!rc) rc = dev_alloc(&m->alpha, (size_t)K);
    }
    if (!rc) rc = dev_alloc(&m->eeb, KV);
    if (!rc) rc = dev_alloc(&m->psi_sum, 3 * (size_t)K);   // psi(row sums), the row sums, exp(-psi)
    if (!rc) rc = dev_alloc(&m->partial, (size_t)kMaxRowsumBlocks * K);
    if (!rc) rc = dev_alloc(&m->counter, 1);
    if (!rc && adopt) {
        m->own_stream = adopt;
        m->stream = adopt;
        LiveStreams &ls = live_own_streams();
        std::lock_guard<std::mutex> lock(ls.mu);
        ls.own.insert(m->own_stream);
    } else if (!rc && !std::getenv("TRLDA_NULL_STREAM")) {
        hipError_t e = stream_priority != 0
                           ? hipStreamCreateWithPriority(&m->own_stream, hipStreamNonBlocking, stream_priority)
                           : hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking);
        if (e != hipSuccess && stream_priority != 0) {   // (a runtime without priorities: a plain stream)
            (void)hipGetLastError();
            e = hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking);
        }
        if (e != hipSuccess)
            rc = fail(TRLDA_ERR_HIP, "hipStreamCreateWithFlags failed");
        m->stream = m->own_stream;
        if (!rc) {
            LiveStreams &ls = live_own_streams();
            std::lock_guard<std::mutex> lock(ls.mu);
            ls.own.insert(m->own_stream);
        }
    }
    if (!rc) rc = dev_alloc(&m->rs_full, (size_t)K);
    if (!rc) rc = dev_alloc(&m->rs_static, (size_t)K);
    if (!rc) rc = dev_alloc(&m->upd_partial, (size_t)(kUpdShortBlocks + kUpdLongBlocks + kUpdVlRows) * K);
    if (!rc) rc = dev_alloc(&m->carry_out, (size_t)kCarryBlocks * K);
    if (!rc) rc = dev_alloc(&m->upd_groups, (size_t)kUpdGroups * K);
    if (!rc) {
        void *p = nullptr;
        // (+1: the row-sum workgroups' counter of a prefetched preamble)
        if (hipMalloc(&p, (kUpdGroups + 1) * sizeof(unsigned int)) != hipSuccess ||
            hipMemset(p, 0, (kUpdGroups + 1) * sizeof(unsigned int)) != hipSuccess)
            rc = fail(TRLDA_ERR_HIP, "hipMalloc failed");
        m->group_counter = static_cast<unsigned int *>(p);
    }
    if (!rc) rc = dev_alloc(&m->scale_comb, 3 * (size_t)K);
    if (!rc) rc = dev_alloc(&m->sync_counters, 64);          // [0], [1]: merged launch; [32]: deferred helpers
    if (!rc && hipMemset(m->sync_counters, 0, 64 * sizeof(unsigned int)) != hipSuccess)
        rc = fail(TRLDA_ERR_HIP, "hipMemset failed");
    {
        const size_t n_flags = (size_t)(trlda::kMergedMaxHelpers + trlda::kMergedMaxDocWgs) * trlda::kMergedFlagStride;
        if (!rc) rc = dev_alloc(&m->sync_flags, n_flags);
        if (!rc && hipMemset(m->sync_flags, 0, n_flags * sizeof(unsigned int)) != hipSuccess)
            rc = fail(TRLDA_ERR_HIP, "hipMemset failed");
    }
    // columns of words no batch has touched yet are never read for their value, but the
    // atomic-mode finish multiplies them by 0: keep them finite
    if (!rc && (hipMemset(m->counter, 0, sizeof(unsigned int)) != hipSuccess ||
                hipMemset(m->eeb, 0, KV * sizeof(double)) != hipSuccess ||
                hipStreamSynchronize(nullptr) != hipSuccess))    // (the model's stream does not
        rc = fail(TRLDA_ERR_HIP, "hipMemset failed");            //  wait for the null stream)
    if (rc) {
        trlda_model_destroy(m);
        return rc;
    }
    if (const char *env = std::getenv("TRLDA_SMALL_K"))          // 0: never a wave per document (A/B)
        m->small_k = env[0] == '0' ? 0 : -1;
    if (const char *env = std::getenv("TRLDA_AUX_DECAY"))        // 0: the streaming kernel behind the launch
        m->aux_decay = env[0] != '0';
    if (const char *env = std::getenv("TRLDA_DRAW_AHEAD")) {     // 0 / 1 / 2: trlda_model_set_draw_ahead
        m->draw_ahead = env[0] == '1';
        m->draw_inlaunch = env[0] == '2';
    }
    if (const char *env = std::getenv("TRLDA_SPLIT_LISTS"))
        m->split_long_lists = env[0] != '0';
    if (const char *env = std::getenv("TRLDA_TILED_TASKS"))
        m->tiled_tasks = env[0] != '0';
    if (const char *env = std::getenv("TRLDA_BIG_EMIT"))
        m->big_emit = env[0] != '0';
    if (const char *env = std::getenv("TRLDA_MERGED"))           // 0 = statistics always a launch of their own
        m->merged_launch = std::max(0, std::min(std::atoi(env), 2));
    *out = m;
    return TRLDA_OK;
}

}  // namespace

extern "C" {

int trlda_model_destroy(trlda_model *m)
{
    if (!m)
        return TRLDA_OK;
    if (hipSetDevice(m->device) == hipSuccess) {
        (void)flush_pending(m);                      // deferred statistics: into the caller's array
        (void)lanes_join(m);                         // ... and the lanes' (their streams end below)
        (void)hipStreamSynchronize(m->stream);
        for (int p = 0; p < 2; ++p) {
            if (m->lane[p])
                (void)trlda_model_destroy(m->lane[p]);
            m->lane[p] = nullptr;
            if (m->lane_in[p])
                (void)hipEventDestroy(m->lane_in[p]);
            if (m->lane_out[p])
                (void)hipEventDestroy(m->lane_out[p]);
            if (p == 0) {
                for (auto &e : m->lane_cal.e)
                    if (e)
                        (void)hipEventDestroy(e);
                for (auto &e : m->lane_cal.s)
                    if (e)
                        (void)hipEventDestroy(e);
                if (m->lane_cal.f)
                    (void)hipEventDestroy(m->lane_cal.f);
            }
            for (int q = 0; q < 2; ++q)
                if (m->lane_span[p][q])
                    (void)hipEventDestroy(m->lane_span[p][q]);
        }
        if (m->lane_owner) {                         // a lane: lambda and alpha are its owner's
            m->lambda = nullptr;
            m->alpha = nullptr;
        }
        if (m->draw_stream)
            (void)hipStreamSynchronize(m->draw_stream);
        (void)hipFree(m->lambda); (void)hipFree(m->alpha); (void)hipFree(m->eeb); (void)hipFree(m->psi_sum);
        (void)hipFree(m->xbuf);
        if (m->xerr_host)
            (void)hipHostFree(const_cast<int *>(m->xerr_host));
        if (m->eb.host)
            (void)hipHostFree(m->eb.host);
        if (m->eb.event)
            (void)hipEventDestroy(m->eb.event);
        (void)hipFree(m->partial); (void)hipFree(m->counter); (void)hipFree(m->epg_base); (void)hipFree(m->tw_csr);
        (void)hipFree(m->sync_counters); (void)hipFree(m->sync_flags); (void)hipFree(m->scale_comb);
        (void)hipFree(m->seg_partial); (void)hipFree(m->seg_counter);
        (void)hipFree(m->tw_word); (void)hipFree(m->dp_gather_own); (void)trlda_model_dp_direct_close(m); (void)hipFree(m->lambda_prime); (void)hipFree(m->sstats); (void)hipFree(m->gamma);
        (void)hipFree(m->wordcounts); (void)hipFree(m->rs_full); (void)hipFree(m->rs_static);
        (void)hipFree(m->upd_partial); (void)hipFree(m->ada_gradient); (void)hipFree(m->reduce_out);
        (void)hipFree(m->carry_out); (void)hipFree(m->upd_groups); (void)hipFree(m->group_counter);
        (void)hipFree(m->iters); (void)hipFree(m->rng_win); (void)hipFree(m->rng_vbuf);
        // a gamma0 drawn ahead that nobody will use: the host stream goes back to its turn
        if (m->spec.valid)
            trlda_host::rng_speculation_cancel_if(m->spec.token);
        // (events go before the streams that recorded them: batch_settle's note)
        if (m->draw_stream) {
            (void)hipStreamSynchronize(m->draw_stream);
            (void)hipEventDestroy(m->ev_main);
            (void)hipEventDestroy(m->ev_draw);
            (void)hipStreamDestroy(m->draw_stream);
        }
        (void)hipFree(m->gspec[0]); (void)hipFree(m->gspec[1]);
        (void)hipFree(m->rng_win2); (void)hipFree(m->rng_vbuf2);
        for (int i = 0; i < 2; ++i) {
            if (m->stage[i])
                (void)hipHostFree(m->stage[i]);
            if (m->stage_ev[i])
                (void)hipEventDestroy(m->stage_ev[i]);
        }
        for (auto &e : m->ev_pool)
            (void)hipEventDestroy(e);
        for (int i = 0; i < 3; ++i) {
            (void)hipFree(m->eeb_pp[i]); (void)hipFree(m->partial_pp[i]); (void)hipFree(m->scale_pp[i]);
        }
        for (int i = 0; i < 2; ++i) {
            (void)hipFree(m->dfr_epg_base[i]); (void)hipFree(m->dfr_tw[i]);
        }
        if (m->own_stream) {
            (void)hipStreamSynchronize(m->own_stream);
            // the recycled batch allocations whose last reader ran on this stream: their work is
            // complete, and their guard events must not outlive the stream (batch_settle's note)
            purge_stream_guards(m->device, m->own_stream);
            LiveStreams &ls = live_own_streams();
            std::lock_guard<std::mutex> lock(ls.mu);
            ls.own.erase(m->own_stream);
            (void)hipStreamDestroy(m->own_stream);
        }

    }
    if (m->pending.valid)
        const_cast<trlda_batch *>(m->pending.batch)->pending_in = nullptr;
    delete m;
    return TRLDA_OK;
}

int trlda_model_set_stream(trlda_model *m, void *hip_stream)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (int rc_sync = sync_model(m))
        return rc_sync;
    m->stream = static_cast<hipStream_t>(hip_stream);
    // (what was seen and measured of the lanes was against the old stream: measured again)
    if (m->lane_cal.phase != 0 || m->lane_state == 1) {
        m->lane_cal.phase = 0;
        m->lane_cal.n = 0;
        m->lane_cal.tries = 0;
        m->lane_cal.keeps = 0;
        m->lane_cal.worse = 0u;
        if (m->lane_state == 1)
            m->lane_state = m->lane[0] ? 2 : 0;
    }
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
    if (kind == TRLDA_DOCS_SMALL || kind == TRLDA_DOCS_REG) {
        // (the choice between the two bodies of the K <= 128 launch forms: everything around them --
        // fused preamble, merged / deferred statistics, lanes -- stays as it is)
        m->small_k = kind == TRLDA_DOCS_SMALL ? 1 : 0;
        m->doc_kernel = TRLDA_DOCS_AUTO;
        return TRLDA_OK;
    }
    if (kind != TRLDA_DOCS_AUTO && kind != TRLDA_DOCS_GENERAL && kind != TRLDA_DOCS_WIDE)
        return fail(TRLDA_ERR_ARG, "doc_kernel must be TRLDA_DOCS_AUTO, _GENERAL, _WIDE, _SMALL or _REG");
    if (kind == TRLDA_DOCS_AUTO)
        m->small_k = -1;
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
    return sync_model(m);
}

int trlda_model_last_split_workgroups(const trlda_model *m) { return m ? m->last_split_wgs : 0; }

int trlda_model_set_merged_launch(trlda_model *m, int enabled)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "model is NULL");
    m->merged_launch = enabled < 0 ? 0 : std::min(enabled, 2);
    return TRLDA_OK;
}

int trlda_model_last_merged(const trlda_model *m) { return m && m->last_merged ? 1 : 0; }

int trlda_model_set_split_docs(trlda_model *m, int enabled)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "model is NULL");
    m->split_docs = enabled != 0;
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
    note_host_lambda(m, host_lambda);     // while the copy runs
    if (int rc_sync = sync_model(m))
        return rc_sync;
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
    m->d2h_bytes += (int64_t)((size_t)m->K * m->V * sizeof(double));
    return sync_model(m);
}

int trlda_model_set_alpha(trlda_model *m, const double *host_alpha)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!host_alpha)
        return fail(TRLDA_ERR_ARG, "alpha is NULL");
    if (m->eb.active)
        return fail(TRLDA_ERR_ARG, "an empirical-Bayes step is on its way and would overwrite this alpha: "
                                   "trlda_model_online_eb_finish first");
    for (int k = 0; k < m->K; ++k)
        if (host_alpha[k] < 0.)
            return fail(TRLDA_ERR_VALUE, "Alpha should not be negative.");  // lda.h:147-159
    HIP_TRY(hipMemcpyAsync(m->alpha, host_alpha, (size_t)m->K * sizeof(double),
                           hipMemcpyHostToDevice, m->stream));
    if (int rc_sync = sync_model(m))
        return rc_sync;
    return TRLDA_OK;
}

void *trlda_model_lambda_dev(trlda_model *m)
{
    if (!m)
        return nullptr;
    // whoever holds this pointer may write lambda: nothing is known about its row sums any more
    invalidate_rowsums(m);
    m->lambda_positive = false;                 // (not tracked through this path)
    m->rs_floor = 0.0;
    m->lambda_exposed = true;
    return m->lambda;
}

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
    m->d2h_bytes += (int64_t)((size_t)m->K * m->V * sizeof(double));
    if (int rc_sync = sync_model(m))
        return rc_sync;
    return TRLDA_OK;
}

int trlda_model_estep(trlda_model *m, const trlda_batch *b, double *gamma_dev, double *sstats_dev,
                      int max_iter, double threshold, int32_t *iters_dev)
{
    // (the batches' indices are built on worker threads: trlda_batch_create)
    if (int rc_built = batch_wait(b))
        return rc_built;
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
    // (the batches' indices are built on worker threads: trlda_batch_create)
    if (int rc_built = batch_wait(b))
        return rc_built;
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!b || !sstats_dev || (b->B > 0 && (!gamma_dev || !gamma0_dev)))
        return fail(TRLDA_ERR_ARG, "NULL batch / gamma / sstats");
    return estep_device(m, b, gamma_dev, sstats_dev, max_iter, threshold, iters_dev, gamma0_dev);
}

int trlda_model_estep_io_next(trlda_model *m, const trlda_batch *b, const trlda_batch *next,
                              const double *gamma0_dev, double *gamma_dev, double *sstats_dev,
                              int max_iter, double threshold, int32_t *iters_dev)
{
    // (the batches' indices are built on worker threads: trlda_batch_create)
    if (int rc_built = batch_wait(b))
        return rc_built;
    if (next && !batch_announced(next))
        next = nullptr;                              // (not indexed yet: as good as not announced)
    // (the next E-step of a deferred stream: estep_device decides whether its launch carries the
    // statistics the call before left pending, or launches them first)
    int rc = check_model(m, true);
    if (rc)
        return rc;
    if (!b || !sstats_dev || (b->B > 0 && (!gamma_dev || !gamma0_dev))) {
        (void)flush_pending(m);
        return fail(TRLDA_ERR_ARG, "NULL batch / gamma / sstats");
    }
    EstepOut out(sstats_dev);
    rc = estep_device(m, b, gamma_dev, out, max_iter, threshold, iters_dev, gamma0_dev, next);
    if (rc && m->pending.valid)                      // a refused call leaves nothing outstanding
        (void)flush_pending(m);
    return rc;
}

int trlda_model_set_deferred_stats(trlda_model *m, int enabled)
{
    int rc = check_model(m);                         // (flushes what is pending)
    if (rc)
        return rc;
    m->deferred_stats = enabled != 0;
    return TRLDA_OK;
}

int trlda_model_flush(trlda_model *m) { return check_model(m); }

int trlda_model_last_deferred(const trlda_model *m)
{
    return m ? (m->last_deferred ? 1 : 0) | (m->last_carried ? 2 : 0) : 0;
}

}  // extern "C"

namespace {

// what a stream lane may take: any E-step on device arrays of a model that holds nothing a lane
// could not see -- no data-parallel context, no empirical-Bayes step on its way, and no row sums
// left behind by an M-step kernel (rowsums_carried: the model's own E-steps use those -- other
// sums than a lane would form from lambda, an ulp apart: no lanes until lambda is set anew, so
// that a stream through the lanes stays bitwise the one-lane stream).  Inside the lane the call
// is what it would have been on the model: with deferred statistics and announcements where they
// apply (small tables, K <= 128, <= 256 documents), the kernels of their own elsewhere.
bool lane_takes(const trlda_model *m, const trlda_batch *b)
{
    return m->lanes_wanted >= 2 && m->lane_state != 1 && !m->dp && !m->eb.active && !rowsums_carried(m) && b->B > 0 &&
           b->V == m->V && b->device == m->device;
}

// ---- which streams may be lanes ------------------------------------------------------------------
// Two launches overlap only if their streams sit on two hardware queues, and the runtime hands a
// stream a queue that is in use as soon as a priority's pool (four by default) is exhausted: two lanes
// on ONE queue -- or a lane on the queue of the caller's stream, whose event records then queue up
// behind the lane's launches -- are SLOWER than one lane (round 5: 31.9-35.3 us per step against 30.9;
// a process that had made other streams first got exactly that).  The assignment
// cannot be asked for or read; it can be seen: a kernel that waits ~40 us on each of two streams --
// they run side by side (40 us in all) or one behind the other (80).  lanes_ensure makes streams of the
// device's high priority until it holds two that run side by side with each other and with the
// model's stream; the ones it had to reject stay alive until it is done (each holds its queue's turn
// in the runtime's round robin) and go then.  No such pair among kLaneCandidates: no lanes for this
// model -- the stream goes one launch at a time, as fast as it ever was (trlda_model_lane_state).
__global__ void lane_probe_kernel(unsigned long long ticks)   // (s_memrealtime: 100 MHz)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks)
        __builtin_amdgcn_s_sleep(32);
}

constexpr int kLaneCandidates = 10;

// do kernels of streams a and b run side by side?  (ev: four timing events)
int streams_overlap(hipStream_t a, hipStream_t b, hipEvent_t (&ev)[4], bool *overlap)
{
    constexpr unsigned long long kTicks = 4000;      // 40 us
    for (int warm = 1; warm >= 0; --warm) {          // (the first launch on a new stream binds its queue)
        HIP_TRY(hipEventRecord(ev[0], a));
        hipLaunchKernelGGL(lane_probe_kernel, dim3(1), dim3(64), 0, a, warm ? 100ull : kTicks);
        HIP_TRY(hipEventRecord(ev[1], a));
        HIP_TRY(hipEventRecord(ev[2], b));
        hipLaunchKernelGGL(lane_probe_kernel, dim3(1), dim3(64), 0, b, warm ? 100ull : kTicks);
        HIP_TRY(hipEventRecord(ev[3], b));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventSynchronize(ev[1]));
        HIP_TRY(hipEventSynchronize(ev[3]));
    }
    float one = 0.f, span = 0.f;
    HIP_TRY(hipEventElapsedTime(&one, ev[0], ev[1]));
    HIP_TRY(hipEventElapsedTime(&span, ev[0], ev[3]));
    *overlap = span < 1.6f * std::max(one, 0.04f);   // (side by side: ~1.1 x; one behind the other: >= 2 x)
    if (!*overlap)
        return TRLDA_OK;
    // ... and a MARKER on a while b's kernel runs: two streams on one hardware queue may still run small
    // kernels side by side (no barrier between their dispatch packets), but an event record is a
    // barrier packet, and a barrier packet waits for everything in front of it in ITS queue -- which
    // is what costs the lanes their gain there (the model's stream records and waits for events
    // between the lanes' 50 us launches).  Recorded on an idle stream it is through in a few
    // microseconds; behind the other stream's kernel it takes the kernel's 40.
    HIP_TRY(hipEventRecord(ev[0], b));
    hipLaunchKernelGGL(lane_probe_kernel, dim3(1), dim3(64), 0, b, kTicks);
    HIP_TRY(hipEventRecord(ev[1], b));
    HIP_TRY(hipEventRecord(ev[2], a));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventSynchronize(ev[1]));
    HIP_TRY(hipEventSynchronize(ev[2]));
    float marker = 0.f;
    HIP_TRY(hipEventElapsedTime(&marker, ev[0], ev[2]));
    *overlap = marker < 0.6f * std::max(one, 0.04f);
    return TRLDA_OK;
}

int lanes_ensure(trlda_model *m)
{
    if (m->lane[0] && m->lane[1])
        return TRLDA_OK;
    if (m->lane_state == 1)
        return TRLDA_OK;
    // The lanes' streams get the device's HIGH priority: a pool of hardware queues nobody else in the
    // process normally draws from (TRLDA_LANE_PRIORITY: A/B; 0: the default priority)
    static const int lane_priority = [] {
        if (const char *e = std::getenv("TRLDA_LANE_PRIORITY"))
            return std::atoi(e);
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess)
            return 0;
        return greatest;
    }();
    static const bool verify = [] {
        const char *e = std::getenv("TRLDA_LANE_VERIFY");
        return !(e && e[0] == '0');
    }();
    auto make_stream = [&](hipStream_t *s) {
        hipError_t e = lane_priority != 0 ? hipStreamCreateWithPriority(s, hipStreamNonBlocking, lane_priority)
                                          : hipStreamCreateWithFlags(s, hipStreamNonBlocking);
        if (e != hipSuccess && lane_priority != 0) { // (a runtime without priorities: a plain stream)
            (void)hipGetLastError();
            e = hipStreamCreateWithFlags(s, hipStreamNonBlocking);
        }
        return e;
    };
    hipStream_t chosen[2] = {nullptr, nullptr};
    std::vector<hipStream_t> rejected;
    int found = 0, rc = TRLDA_OK;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    for (auto &e : ev)
        if (verify && hipEventCreate(&e) != hipSuccess)
            rc = fail(TRLDA_ERR_HIP, "hipEventCreate failed");
    // (the caller's stream is waited for first: the probe's kernels are timed against an idle device)
    if (!rc && verify && hipStreamSynchronize(m->stream) != hipSuccess)
        rc = fail(TRLDA_ERR_HIP, "hipStreamSynchronize failed");
    for (int c = 0; !rc && found < 2 && c < (verify ? kLaneCandidates : 2); ++c) {
        hipStream_t s = nullptr;
        if (make_stream(&s) != hipSuccess) {
            rc = fail(TRLDA_ERR_HIP, "hipStreamCreate failed");
            break;
        }
        bool ok = true;
        if (verify) {
            rc = streams_overlap(m->stream, s, ev, &ok);
            if (!rc && ok && found == 1)
                rc = streams_overlap(chosen[0], s, ev, &ok);
        }
        if (!rc && ok)
            chosen[found++] = s;
        else
            rejected.push_back(s);
    }
    for (hipStream_t s : rejected)
        (void)hipStreamDestroy(s);
    for (auto &e : ev)
        if (e)
            (void)hipEventDestroy(e);
    if (!rc && found < 2) {
        for (int p = 0; p < found; ++p)
            (void)hipStreamDestroy(chosen[p]);
        m->lane_state = 1;                           // no pair of queues: one launch at a time
        return TRLDA_OK;
    }
    for (int p = 0; !rc && p < 2; ++p) {
        trlda_model *l = nullptr;
        rc = model_create(&l, m->device, m->K, m->V, lane_priority, m, chosen[p]);
        if (rc)
            break;
        chosen[p] = nullptr;                         // (the lane's own now)
        // (the events first: a lane is published whole or not at all)
        if ((!m->lane_in[p] && hipEventCreateWithFlags(&m->lane_in[p], hipEventDisableTiming) != hipSuccess) ||
            (!m->lane_out[p] && hipEventCreateWithFlags(&m->lane_out[p], hipEventDisableTiming) != hipSuccess)) {
            (void)trlda_model_destroy(l);
            rc = fail(TRLDA_ERR_HIP, "hipEventCreateWithFlags failed");
            break;
        }
        m->lane[p] = l;
    }
    if (rc) {
        for (int p = 0; p < 2; ++p) {
            if (chosen[p])
                (void)hipStreamDestroy(chosen[p]);
            if (m->lane[p])
                (void)trlda_model_destroy(m->lane[p]);
            m->lane[p] = nullptr;
        }
        return rc;
    }
    m->lane_state = verify ? 2 : 3;
    return TRLDA_OK;
}

// a lane follows its owner's switches and what the owner knows about lambda (what a lane keeps
// about lambda -- a prefetched preamble -- is stamped with the version: a new one voids it)
void lane_follow(const trlda_model *m, trlda_model *l)
{
    l->sstats_mode = m->sstats_mode; l->doc_threads = m->doc_threads; l->doc_kernel = m->doc_kernel;
    l->small_k = m->small_k;
    l->split_preamble = m->split_preamble; l->dense_preamble = m->dense_preamble;
    l->pair_gathers = m->pair_gathers; l->prefetch_next = m->prefetch_next; l->split_docs = m->split_docs;
    l->merged_launch = m->merged_launch; l->split_long_lists = m->split_long_lists;
    l->tiled_tasks = m->tiled_tasks; l->fused_update = m->fused_update; l->carry_rowsums = m->carry_rowsums;
    l->emit_next_preamble = m->emit_next_preamble;
    l->deferred_stats = m->deferred_stats; l->big_emit = m->big_emit; l->keep_sstats = false;
    l->rs_floor = m->rs_floor; l->lambda_positive = m->lambda_positive;
    l->lambda_exposed = m->lambda_exposed;           // (the same choice of preamble as the model's own)
    l->lambda_version = m->lambda_version;
    // (not the per-launch event stamps of trlda_model_set_timing: events between a lane's launches
    // would change what they measure -- lane_span above)
}

// closed spans -> lane_span_us / lane_span_count (waits for the lanes' end events)
void lane_spans_collect(trlda_model *m)
{
    for (int p = 0; p < 2; ++p) {
        if (!m->lane_span_open[p] || m->lanes_live)
            continue;
        float ms = 0.f;
        if (hipEventSynchronize(m->lane_span[p][1]) == hipSuccess &&
            hipEventElapsedTime(&ms, m->lane_span[p][0], m->lane_span[p][1]) == hipSuccess) {
            m->lane_span_us += 1e3 * (double)ms;
            m->lane_span_count += m->lane_span_launches[p];
        }
        m->lane_span_open[p] = false;
        m->lane_span_launches[p] = 0;
    }
}

bool ranges_meet(const trlda_model::LaneRange &w, const void *p, size_t bytes)
{
    const char *lo = static_cast<const char *>(p);
    return w.lo && p && bytes && lo < w.hi && w.lo < lo + bytes;
}

}  // namespace

extern "C" {

int trlda_model_set_stream_lanes(trlda_model *m, int lanes)
{
    int rc = check_model(m);                         // (joins what the lanes hold)
    if (rc)
        return rc;
    if (lanes < 1 || lanes > 2)
        return fail(TRLDA_ERR_ARG, "stream lanes: 1 or 2");
    m->lanes_wanted = lanes;
    return TRLDA_OK;
}

long long trlda_model_lane_steps(const trlda_model *m) { return m ? (long long)m->lane_steps : 0; }

int trlda_model_lane_state(const trlda_model *m) { return m ? m->lane_state : 0; }

int trlda_model_lane_timing(const trlda_model *m, double *us_launch, double *us_step)
{
    if (!m || !us_launch || !us_step)
        return fail(TRLDA_ERR_ARG, "NULL model / output");
    *us_launch = m->lane_cal.us_launch;
    *us_step = m->lane_cal.us_step;
    return TRLDA_OK;
}

namespace {
// TRLDA_LANE_TRACE=1: host time of the phases of the first calls after a join (stderr)
struct LaneTrace {
    bool on;
    std::chrono::steady_clock::time_point t0;
    double us[6] = {0, 0, 0, 0, 0, 0};
    int n = 0;
    explicit LaneTrace(bool enabled) : on(enabled) { if (on) t0 = std::chrono::steady_clock::now(); }
    void mark()
    {
        if (!on || n >= 6)
            return;
        const auto t = std::chrono::steady_clock::now();
        us[n++] = std::chrono::duration<double, std::micro>(t - t0).count();
        t0 = t;
    }
};
}  // namespace

int trlda_model_estep_io_ahead(trlda_model *m, const trlda_batch *b, const trlda_batch *const *upcoming,
                               int n_upcoming, const double *gamma0_dev, double *gamma_dev,
                               double *sstats_dev, int max_iter, double threshold, int32_t *iters_dev)
{
    // (the batches' indices are built on worker threads: trlda_batch_create)
    CallClock call_clock;
    if (int rc_built = batch_wait(b))
        return rc_built;
    call_clock.to(0);
    if (!m)
        return fail(TRLDA_ERR_ARG, "model is NULL");
    if (n_upcoming < 0 || (n_upcoming > 0 && !upcoming))
        return fail(TRLDA_ERR_ARG, "upcoming batches: a count without a list");
    // An announced batch whose index is not there yet counts as not announced (the call that gets it
    // prepares its own preamble: the same results): an announcement is not worth taking a build over
    // from the workers for, let alone waiting for one.
    const trlda_batch *ready_up[2] = {nullptr, nullptr};
    for (int i_up = 0; i_up < n_upcoming && i_up < 2; ++i_up) {
        const trlda_batch *up = upcoming[i_up];
        if (batch_announced(up))                     // (its upload is enqueued here if its index is there)
            ready_up[i_up] = up;
    }
    if (n_upcoming > 2)
        n_upcoming = 2;
    upcoming = ready_up;
    const trlda_batch *next = n_upcoming > 0 ? upcoming[0] : nullptr;
    if (!b || !sstats_dev || (b->B > 0 && (!gamma_dev || !gamma0_dev)))
        return trlda_model_estep_io_next(m, b, next, gamma0_dev, gamma_dev, sstats_dev, max_iter,
                                         threshold, iters_dev);
    // A caller that hands consecutive calls the same arrays (or reads the last call's gamma as this
    // call's gamma0) has nothing to run side by side: one lane, as without the switch
    const size_t g_bytes = (size_t)b->B * m->K * sizeof(double), s_bytes = (size_t)m->K * m->V * sizeof(double);
    const size_t i_bytes = (size_t)b->B * sizeof(int32_t);
    auto meets_any = [&](const trlda_model::LaneRange *w, size_t n) {
        bool hit = false;
        for (size_t i = 0; i < n; ++i)
            hit = hit || ranges_meet(w[i], gamma_dev, g_bytes) || ranges_meet(w[i], gamma0_dev, g_bytes) ||
                  ranges_meet(w[i], sstats_dev, s_bytes) || ranges_meet(w[i], iters_dev, i_bytes);
        return hit;
    };
    const bool chained = meets_any(m->prev_writes, 3);
    {
        const char *g = reinterpret_cast<const char *>(gamma_dev), *st = reinterpret_cast<const char *>(sstats_dev),
                   *it = reinterpret_cast<const char *>(iters_dev);
        m->prev_writes[0] = {g, g ? g + g_bytes : g};
        m->prev_writes[1] = {st, st + s_bytes};
        m->prev_writes[2] = {it, it ? it + i_bytes : it};
    }
    if (chained || !lane_takes(m, b))
        return trlda_model_estep_io_next(m, b, next, gamma0_dev, gamma_dev, sstats_dev, max_iter,
                                         threshold, iters_dev);
    static const bool trace_on = std::getenv("TRLDA_LANE_TRACE") != nullptr;
    LaneTrace tr(trace_on && m->lane_calls[0] + m->lane_calls[1] < 3);
    // (statistics this model itself holds pending go first; what the lanes hold stays)
    int rc = check_model(m, false, true);
    if (!rc)
        rc = lanes_ensure(m);
    if (rc)
        return rc;
    if (m->lane_state == 1)                          // (no two streams that run side by side: lanes_ensure)
        return trlda_model_estep_io_next(m, b, next, gamma0_dev, gamma_dev, sstats_dev, max_iter,
                                         threshold, iters_dev);
    // the calibration (trlda_model::lane_cal): its verdict
    constexpr int kLaneCalAfter = 96, kLaneCalLaunches = 4;
    static const bool calibrate = [] {
        const char *e = std::getenv("TRLDA_LANE_CALIBRATE");
        return !(e && e[0] == '0');
    }();
    auto &cal = m->lane_cal;
    if (calibrate && cal.phase == 0 && m->lane_steps >= kLaneCalAfter) {
        for (auto &e : cal.e)
            if (!e && hipEventCreate(&e) != hipSuccess)
                cal.phase = 5;                       // (no events: no calibration, ever)
        for (auto &e : cal.s)
            if (!e && hipEventCreate(&e) != hipSuccess)
                cal.phase = 5;
        if (!cal.f && hipEventCreate(&cal.f) != hipSuccess)
            cal.phase = 5;
        if (cal.phase == 0) {
            cal.phase = 6;                           // (a look begins with its one-lane stretch)
            cal.n = 0;
            cal.solo_valid = false;
            cal.short_stretches = 0;
        }
    }
    if (calibrate && cal.phase == 4 && m->lane_state == 2 && m->lane_steps >= cal.next_check) {
        cal.phase = 6;                               // (the lanes are looked at again)
        cal.n = 0;
        cal.tries = 0;
        cal.solo_valid = false;
        cal.short_stretches = 0;
    }
    if (cal.phase == 3 && hipEventQuery(cal.e[2]) == hipSuccess && hipEventQuery(cal.f) == hipSuccess) {
        float launch = 0.f, steps = 0.f, steps1 = 0.f;
        cal.phase = 4;
        cal.next_check = m->lane_steps + kLaneCalEvery;
        if (hipEventElapsedTime(&launch, cal.e[0], cal.e[1]) == hipSuccess &&
            hipEventElapsedTime(&steps, cal.e[1], cal.e[2]) == hipSuccess &&
            hipEventElapsedTime(&steps1, cal.e[1], cal.f) == hipSuccess && steps > 0.f) {
            steps = std::max(steps, steps1);         // (both lanes' 2 x kLaneCalLaunches launches are through)
            cal.us_launch = 1e3f * launch;
            cal.us_step = 1e3f * steps / (2 * kLaneCalLaunches);
            // (TRLDA_LANE_CAL_HOST_SHARE: tests take every window for a verdict with 1e9)
            const char *he = std::getenv("TRLDA_LANE_CAL_HOST_SHARE");
            const float host_share = he ? (float)std::atof(he) : 0.7f;
            if (cal.host_us_step > host_share * cal.us_step) {   // the host did not keep the lanes fed: no verdict
                cal.phase = ++cal.tries > 8 ? 4 : 6;
                cal.n = 0;
                cal.us_launch = cal.us_step = 0.f;
            } else {
                // One launch is one sample, and launches differ (a batch with long documents lasts
                // longer than the step before and after it): below kDrop launches in flight the lanes
                // are given up, above kKeep kept, in between the window is taken again (the failing
                // case read 0.55-0.9 and once 1.33, the bench 1.7-2.2, log-normal lengths 1.34).
                // (TRLDA_LANE_CAL_MIN_IN_FLIGHT: tests make the lanes lose with 100, win with 0)
                const char *me = std::getenv("TRLDA_LANE_CAL_MIN_IN_FLIGHT");
                const float drop = me ? (float)std::atof(me) : 1.0f, keep = me ? drop : 1.4f;
                // ... and against the one-lane stretch in front of the window (no verdict from a
                // stretch the host did not keep fed, or when a test sets the bar)
                float solo_ms = 0.f;
                const bool solo_ok = !me && cal.solo_valid &&
                                     hipEventElapsedTime(&solo_ms, cal.s[0], cal.s[1]) == hipSuccess && solo_ms > 0.f;
                cal.solo_valid = false;
                cal.short_stretches = 0;
                cal.us_solo = solo_ok ? 1e3f * solo_ms / kLaneSoloSteps : 0.f;
                const bool solo_fed = solo_ok && cal.host_us_solo <= host_share * cal.us_solo;
                // A look says NO when the launches do not overlap (less than one in flight) or when two
                // lanes are not 3 % faster than one.  One look is one sample -- a stream with uploads
                // beside its launches has its hiccups: the lanes are given up when two of the last four
                // looks said no (a look that did is followed by the next at once), kept when two in a
                // row said yes.  (A process whose lanes pay now and then: half its looks say no.)
                const bool no = cal.us_launch < drop * cal.us_step || (solo_fed && cal.us_step >= 0.97f * cal.us_solo);
                cal.worse = ((cal.worse << 1) | (no ? 1u : 0u)) & 0xFu;
                if (no) {
                    cal.keeps = 0;
                    if (__builtin_popcount(cal.worse) >= 2) {   // nothing gained: one launch at a time
                        if ((rc = check_model(m, /*keep_pending=*/true)))   // (joins the lanes)
                            return rc;
                        m->lane_state = 1;
                        return trlda_model_estep_io_next(m, b, next, gamma0_dev, gamma_dev, sstats_dev, max_iter,
                                                         threshold, iters_dev);
                    }
                    cal.phase = 6;                   // (looked at again, at once)
                    cal.n = 0;
                } else if (!solo_fed && cal.us_launch < keep * cal.us_step) {   // (no one-lane figure, and between the bars)
                    cal.keeps = 0;
                    if (++cal.tries <= 8) {
                        cal.phase = 6;
                        cal.n = 0;
                    }
                } else if (++cal.keeps < 2) {
                    cal.phase = 6;
                    cal.n = 0;
                }
            }
        }
    }
    (void)hipGetLastError();                         // (hipErrorNotReady of the queries above)
    // (the one-lane stretch of a look: lane 0 takes every call -- once the stretch is long enough to hold it)
    const bool solo = cal.phase == 6 && (cal.n > 0 || m->lane_calls[0] + m->lane_calls[1] >= kLaneSoloAfter);
    const int p = solo ? 0 : m->lane_turn;
    trlda_model *l = m->lane[p], *o = m->lane[1 - p];
    lane_follow(m, l);
    tr.mark();
    // whatever the caller enqueued on the model's stream so far -- this call's gamma0, the last
    // reader of the arrays it writes, an upload of lambda -- comes first (nothing of the lanes is
    // on that stream: the record passes as soon as the caller's own work has)
    static const bool input_events = [] {
        const char *e = std::getenv("TRLDA_LANE_INPUT_EVENTS");
        return !(e && e[0] == '0');
    }();
    // (a stream with nothing outstanding has nothing to wait for: no record, no packet in its queue --
    // at the start of a stream of calls that is two hops between three hardware queues before the
    // first launch may begin)
    const bool first = !m->lanes_live;
    if ((input_events || first) && hipStreamQuery(m->stream) != hipSuccess) {
        (void)hipGetLastError();                     // (hipErrorNotReady is not an error)
        HIP_TRY(hipEventRecord(m->lane_in[p], m->stream));
        HIP_TRY(hipStreamWaitEvent(l->stream, m->lane_in[p], 0));
        if (first)                                   // (the other lane's first call may skip its own)
            HIP_TRY(hipStreamWaitEvent(o->stream, m->lane_in[p], 0));
    }
    // the OTHER lane's outstanding launches write arrays this call reads or writes (a caller that
    // hands the same gamma / sstats to consecutive calls): they go first -- correct, and serial
    tr.mark();
    // (a caller that hands every call arrays of its own lets the list grow: past kLaneRangeCap ranges
    // the other lane is waited for once, as if they met, and its list starts again)
    constexpr size_t kLaneRangeCap = 96;
    if (meets_any(m->lane_writes[1 - p].data(), m->lane_writes[1 - p].size()) ||
        m->lane_writes[1 - p].size() > kLaneRangeCap) {
        if ((rc = flush_pending(o)))
            return rc;
        HIP_TRY(hipEventRecord(m->lane_out[1 - p], o->stream));
        HIP_TRY(hipStreamWaitEvent(l->stream, m->lane_out[1 - p], 0));
        m->lane_writes[1 - p].clear();
    }
    if (m->timing && !m->lane_span_open[p]) {        // (the first call of a stretch on this lane)
        lane_spans_collect(m);
        for (int q = 0; q < 2 && !m->lane_span[p][q]; ++q)
            HIP_TRY(hipEventCreate(&m->lane_span[p][q]));
        HIP_TRY(hipEventRecord(m->lane_span[p][0], l->stream));
        m->lane_span_open[p] = true;
        m->lane_span_launches[p] = 0;
    }
    if (m->lane_span_open[p])
        ++m->lane_span_launches[p];
    m->lanes_live = true;
    // (the calibration's window opens eight steps into a stretch: the first launches after a join do
    // not overlap yet, and a stretch shorter than that is never measured)
    if (cal.phase == 2 && cal.n == 0 && cal.lead > 0)
        --cal.lead;
    const bool cal_first = cal.phase == 2 && p == 0 && cal.n == 0 && cal.lead == 0 &&
                           m->lane_calls[0] + m->lane_calls[1] >= 8;
    if (cal_first)
        HIP_TRY(hipEventRecord(cal.e[0], l->stream));
    EstepOut out(sstats_dev);
    call_clock.to(1);
    // (the batch this lane's NEXT launch will take: two ahead with the lanes in turn; the very next one
    // inside a one-lane stretch, whose last call hands over to the turns again)
    const bool solo_next = solo && cal.n + 1 < kLaneSoloLead + kLaneSoloSteps;
    rc = estep_device(l, b, gamma_dev, out, max_iter, threshold, iters_dev, gamma0_dev,
                      solo_next ? (n_upcoming > 0 ? upcoming[0] : nullptr)
                                : (n_upcoming > 1 ? upcoming[1] : nullptr));
    call_clock.to(2);
    tr.mark();
    if (tr.on)
        std::fprintf(stderr, "lane %d call %d of the stretch: set-up %.1f us, caller's stream %.1f us, "
                             "launch sequence %.1f us\n", p, m->lane_calls[p], tr.us[0], tr.us[1], tr.us[2]);
    if (rc) {                                        // a refused call leaves nothing outstanding
        (void)lanes_join(m);
        return rc;
    }
    {
        // (kept until the other lane has waited for this one or the lanes are joined: the sstats of
        // this call are written by the lane's next launch)
        std::vector<trlda_model::LaneRange> &w = m->lane_writes[p];
        auto note = [&w](const void *ptr, size_t bytes) {
            const char *lo = static_cast<const char *>(ptr);
            if (!lo || !bytes)
                return;
            for (const auto &r : w)
                if (r.lo == lo && r.hi == lo + bytes)
                    return;
            w.push_back(trlda_model::LaneRange{lo, lo + bytes});
        };
        note(gamma_dev, g_bytes);
        note(sstats_dev, s_bytes);
        note(iters_dev, i_bytes);
        ++m->lane_calls[p];
    }
    if (solo) {                                      // (after the launch: lane 0's stream's position)
        ++cal.n;
        if (cal.n == kLaneSoloLead) {
            HIP_TRY(hipEventRecord(cal.s[0], l->stream));
            cal.solo_t0 = std::chrono::steady_clock::now();
        } else if (cal.n == kLaneSoloLead + kLaneSoloSteps) {
            HIP_TRY(hipEventRecord(cal.s[1], l->stream));
            cal.host_us_solo = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() -
                                                                        cal.solo_t0).count() / kLaneSoloSteps;
            cal.phase = 2;                           // (then the window through both lanes, after a lead-in)
            cal.n = 0;
            cal.lead = kLaneDuoLead;
            cal.solo_valid = true;
        }
    }
    if (cal.phase == 2 && p == 1 && !solo && cal.n > 0 && ++cal.n1 == kLaneCalLaunches)
        HIP_TRY(hipEventRecord(cal.f, l->stream));   // (lane 1's last launch of the window)
    if (cal.phase == 2 && p == 0 && !solo) {         // (after lane 0's launch: its stream's position)
        if (cal_first) {
            HIP_TRY(hipEventRecord(cal.e[1], l->stream));
            cal.host_t0 = std::chrono::steady_clock::now();
            cal.n = 1;
            cal.n1 = 0;
        } else if (cal.n > 0 && ++cal.n == 1 + kLaneCalLaunches) {
            HIP_TRY(hipEventRecord(cal.e[2], l->stream));
            cal.host_us_step = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() -
                                                                        cal.host_t0).count() / (2 * kLaneCalLaunches);
            cal.phase = 3;
            cal.n = 0;
        }
    }
    m->lane_turn = solo ? 1 : 1 - p;                 // (after a one-lane stretch the other lane is next)
    ++m->lane_steps;
    m->last_deferred = l->last_deferred; m->last_carried = l->last_carried;
    m->last_doc_kernel = l->last_doc_kernel; m->last_preamble_fused = l->last_preamble_fused;
    m->last_split_wgs = l->last_split_wgs; m->last_merged = l->last_merged;
    call_clock.to(3);
    g_call_us[7] += 1.0;
    return TRLDA_OK;
}

int trlda_model_estep_corpus(trlda_model *m, int64_t n_docs, const int64_t *offsets, const int32_t *ids,
                             const int32_t *cnts, int batch_size, const double *gamma0_dev, double *gamma_dev,
                             double *const *sstats_ring, int n_ring, int max_iter, double threshold,
                             int32_t *iters_dev)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (n_docs < 0 || batch_size <= 0 || !offsets || !sstats_ring || n_ring < 3 ||
        (n_docs > 0 && (!gamma0_dev || !gamma_dev)))
        return fail(TRLDA_ERR_ARG, "corpus pass: NULL arrays, a batch size <= 0 or fewer than three statistics arrays");
    for (int r = 0; r < n_ring; ++r)
        if (!sstats_ring[r])
            return fail(TRLDA_ERR_ARG, "corpus pass: a NULL statistics array");
    const int64_t n_batches = (n_docs + batch_size - 1) / batch_size;
    // (a build is done ~170 us after its trlda_batch_create -- queue, grace period, index, upload calls: a
    // batch announced two steps before its E-step at 30-45 us per step has to be made eight steps ahead)
    static const int kAhead = [] {
        const char *e = std::getenv("TRLDA_CORPUS_AHEAD");
        return e ? std::max(1, std::min(16, std::atoi(e))) : 8;
    }();
    constexpr int kBehind = 4;
    const bool was_deferred = m->deferred_stats;
    const int was_lanes = m->lanes_wanted;
    m->deferred_stats = true;
    static const int corpus_lanes = [] {             // (A/B: TRLDA_CORPUS_LANES=1)
        const char *e = std::getenv("TRLDA_CORPUS_LANES");
        return e && e[0] == '1' ? 1 : 2;
    }();
    m->lanes_wanted = corpus_lanes;
    std::vector<trlda_batch *> batch((size_t)n_batches, nullptr);
    std::vector<int32_t> indptr;
    auto make = [&](int64_t i) {
        const int64_t d0 = i * batch_size, d1 = std::min<int64_t>(n_docs, d0 + batch_size);
        indptr.resize((size_t)(d1 - d0) + 1);
        for (int64_t d = d0; d <= d1; ++d) {
            const int64_t at = offsets[d] - offsets[d0];
            if (at < 0 || at > INT32_MAX)
                return fail(TRLDA_ERR_ARG, "corpus pass: offsets must not decrease, a mini-batch holds < 2^31 entries");
            indptr[(size_t)(d - d0)] = (int32_t)at;
        }
        return trlda_batch_create(&batch[(size_t)i], m->device, m->V, (int)(d1 - d0), indptr.data(),
                                  ids ? ids + offsets[d0] : nullptr, cnts ? cnts + offsets[d0] : nullptr);
    };
    for (int64_t i = 0; !rc && i < std::min<int64_t>(kAhead, n_batches); ++i)
        rc = make(i);
    for (int64_t i = 0; !rc && i < n_batches; ++i) {
        if (i + kAhead < n_batches)
            rc = make(i + kAhead);
        const trlda_batch *up[2] = {i + 1 < n_batches ? batch[(size_t)i + 1] : nullptr,
                                    i + 2 < n_batches ? batch[(size_t)i + 2] : nullptr};
        const int n_up = (up[0] != nullptr) + (up[0] && up[1]);
        const size_t at = (size_t)i * (size_t)batch_size;
        if (!rc)
            rc = trlda_model_estep_io_ahead(m, batch[(size_t)i], up, n_up, gamma0_dev + at * m->K,
                                            gamma_dev + at * m->K, sstats_ring[i % n_ring], max_iter, threshold,
                                            iters_dev ? iters_dev + at : nullptr);
        if (i >= kBehind) {                          // (its statistics rode on launch i - 2)
            (void)trlda_batch_destroy(batch[(size_t)(i - kBehind)]);
            batch[(size_t)(i - kBehind)] = nullptr;
        }
    }
    const int rc_flush = check_model(m);             // flushes, joins the lanes: all of it on the model's stream
    for (trlda_batch *b : batch)
        if (b)
            (void)trlda_batch_destroy(b);
    m->deferred_stats = was_deferred;
    m->lanes_wanted = was_lanes;
    return rc ? rc : rc_flush;
}

int trlda_model_set_prefetch(trlda_model *m, int enabled)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "model is NULL");
    m->prefetch_next = enabled != 0;
    m->prefetch.valid = false;
    return TRLDA_OK;
}

int trlda_model_estep_host(trlda_model *m, const trlda_batch *b, double *gamma, double *sstats,
                           int max_iter, double threshold, int32_t *iters_out)
{
    // (the batches' indices are built on worker threads: trlda_batch_create)
    if (int rc_built = batch_wait(b))
        return rc_built;
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
        rc = grow(&m->iters, &m->cap_iters, (size_t)std::max(b->B, 1));
        if (rc)
            return rc;
        iters_dev = m->iters;
    }
    m->gamma0_src = nullptr;                           // the caller's gamma, not one drawn ahead
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
        if (int rc_sync = sync_model(m))
            return rc_sync;
        m->d2h_bytes += (int64_t)(gbytes + sbytes);
        rc = check_split_exchange(m);
    }
    return rc;
}

// LDA::lowerBound, src/lda.cpp:297-360 (see csrc/elbo_kernels.h)
int trlda_model_lower_bound(trlda_model *m, const trlda_batch *b, double *gamma, double eta,
                            double factor, int max_iter, double threshold, double *bound_out)
{
    // (the batches' indices are built on worker threads: trlda_batch_create)
    if (int rc_built = batch_wait(b))
        return rc_built;
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
    m->gamma0_src = nullptr;                           // the caller's gamma, not one drawn ahead
    HIP_TRY(hipMemcpyAsync(m->gamma, gamma, gbytes, hipMemcpyHostToDevice, m->stream));
    rc = estep_device(m, b, m->gamma, m->sstats, max_iter, threshold, nullptr);   // :309
    if (rc)
        return rc;
    const int G = (int)std::min<size_t>((KV + kDenseThreads - 1) / kDenseThreads, 1024);
    rc = grow(&m->reduce_out, &m->cap_reduce, 2 * (size_t)G + 2 * (size_t)B);
    if (rc)
        return rc;
    double *out = m->reduce_out;
    hipLaunchKernelGGL(trlda::elbo_dense_kernel<kDenseThreads>, dim3(G), dim3(kDenseThreads), 0,
                       m->stream, K, KV, eta, factor, m->lambda, m->psi_sum, m->sstats, out);
    const size_t lds = ((size_t)K + 4 * (kDenseThreads / trlda::kWave)) * sizeof(double);
    hipLaunchKernelGGL(trlda::elbo_docs_kernel<kDenseThreads>, dim3(B), dim3(kDenseThreads), lds,
                       m->stream, K, b->indptr, b->ids, b->cnts, m->lambda, m->psi_sum, m->alpha,
                       m->gamma, out + 2 * (size_t)G);
    (void)batch_end(m, b);
    std::vector<double> h(2 * (size_t)G + 2 * (size_t)B), lam_sum((size_t)K), alpha_h((size_t)K);
    hipError_t e1 = hipMemcpyAsync(h.data(), out, h.size() * sizeof(double), hipMemcpyDeviceToHost,
                                   m->stream);
    hipError_t e2 = hipMemcpyAsync(gamma, m->gamma, gbytes, hipMemcpyDeviceToHost, m->stream);
    hipError_t e3 = hipMemcpyAsync(alpha_h.data(), m->alpha, (size_t)K * sizeof(double),
                                   hipMemcpyDeviceToHost, m->stream);
    hipError_t e4 = hipStreamSynchronize(m->stream);
    HIP_TRY(e1); HIP_TRY(e2); HIP_TRY(e3); HIP_TRY(e4);
    HIP_TRY(hipGetLastError());
    if (int rc_x = check_split_exchange(m))
        return rc_x;
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
    // (the batches' indices are built on worker threads: trlda_batch_create)
    if (int rc_built = batch_wait(b))
        return rc_built;
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!b || !lambda_prime_dev || b->B <= 0)
        return fail(TRLDA_ERR_ARG, "NULL or empty batch / lambda_prime");
    return tr_init_device(m, b, lambda_prime_dev, rho, eta, num_documents);
}

int trlda_model_wordcounts(trlda_model *m, const trlda_batch *b, double *wordcounts_dev)
{
    // (the batches' indices are built on worker threads: trlda_batch_create)
    if (int rc_built = batch_wait(b))
        return rc_built;
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

}  // extern "C"

namespace {

// rs_full = rs_static (or nothing) + the block partials the statistics kernel left behind
// the row sums of the lambda just written: `n` rows of block partials (+ base)
int carry_rowsums_from(trlda_model *m, const double *rows, int n, const double *base, double floor)
{
    invalidate_rowsums(m);
    m->rs_floor = floor;
    if ((size_t)m->K * m->V < ((size_t)1 << 22) && m->K <= trlda::kRegMaxK) {
        // small table: the next preamble launch adds the pieces up itself (no launch of its own)
        m->carry_pending = true;
        m->carry_rows = rows;
        m->carry_n = n;
        m->carry_base = base;
        return TRLDA_OK;
    }
    int rc = combine_rowsums(m, rows, n, base, m->rs_full);
    if (rc)
        return rc;
    m->rs_valid = true;
    return TRLDA_OK;
}

int finish_rowsums(trlda_model *m, const EstepOut &out, const double *base, double floor,
                   const trlda_batch *b = nullptr)
{
    if (out.no_rows) {
        // word-sharded M-step: every rank wrote a part of lambda and received the rest; the next
        // E-step adds the rows up from lambda itself (lda.cpp:172 as written)
        invalidate_rowsums(m);
        m->rs_floor = floor;
        m->next_pre.valid = false;
        return TRLDA_OK;
    }
    int rc = carry_rowsums_from(m, out.upd.partial, out.partial_rows, base, floor);
    // (carry_rowsums_from moved lambda_version on: what the kernel left behind belongs to the new one)
    m->next_pre.valid = !rc && out.groups > 0 && b && base == out.next_base;
    if (m->next_pre.valid) {
        m->next_pre.all = !out.active_only;
        m->next_pre.version = m->lambda_version;
        m->next_pre.batch_id = b->id;
        m->next_pre.n = out.groups;
        m->next_pre.raw = out.raw_rows;
    }
    return rc;
}

// rs_full for the E-step that follows, when nothing carried it here (big tables only: the
// small-table preamble adds up lambda itself)
int ensure_rowsums(trlda_model *m)
{
    if (rowsums_carried(m) || !stream_available(m))
        return TRLDA_OK;
    if ((size_t)m->K * m->V < ((size_t)1 << 22))
        return TRLDA_OK;
    int rc = rowsums_from_scratch(m);
    if (!rc)
        m->rs_valid = true;      // lambda has not changed since: they ARE its row sums
    return rc;
}

// OnlineLDA::updateParameters' lambda path with the statistics, the M-step and the next row
// sums in one kernel per trust-region iteration, restricted to the batch's active words
// (sstats_update_kernel; stream_kernels.h for the words outside the batch).
int online_update_fused(trlda_model *m, const trlda_batch *b, int num_documents, double eta,
                        int max_iter_tr, int max_iter_inference, double rho, int init_gamma,
                        double threshold)
{
    const int K = m->K, V = m->V, B = b->B;
    const size_t KV = (size_t)K * V;
    const double scale = (double)num_documents / (double)B;
    const double floor_after = (1. - rho) * m->rs_floor + rho * V * eta;   // sstats >= 0
    const bool keep = m->keep_sstats;     // adaptive rate: whole lambda' and the statistics stay
    int rc = TRLDA_OK, G = 0;

    EstepOut out;
    out.upd.omr = 1. - rho; out.upd.rho = rho; out.upd.eta = eta; out.upd.scale = scale;
    out.upd.lambda = m->lambda;
    out.upd.partial = m->upd_partial;

    if (max_iter_tr > 0) {
        // onlinelda.cpp:79-86 for the active words, the final value for all the others
        // (the words' count sums came with the batch, unless one of them overflows 32 bits)
        const int32_t *wc32 = b->wc32_ok ? b->wc32 : nullptr;
        rc = wc32 ? batch_begin(m, b) : wordcounts_device(m, b, m->wordcounts);
        const double coef = (double)num_documents / (double)B / (double)K;   // onlinelda.cpp:86
        if (!rc)
            rc = keep ? launch_inactive_update<trlda::ACT_TRINIT, true>(
                            m, 1. - rho, rho * eta, rho, eta, coef, b->active_flag, m->wordcounts,
                            m->lambda, m->lambda_prime, &G, wc32)
                      : launch_inactive_update<trlda::ACT_TRINIT, false>(
                            m, 1. - rho, rho * eta, rho, eta, coef, b->active_flag, m->wordcounts,
                            m->lambda, m->lambda_prime, &G, wc32);
        if (!rc) rc = combine_rowsums(m, m->partial, G, nullptr, m->rs_static);
        // (the initial step obeys the same bound on the row sums as the M-steps)
        if (!rc) rc = carry_rowsums_from(m, m->partial + (size_t)trlda::kStreamMaxBlocks * K, G,
                                         m->rs_static, floor_after);
        if (rc)
            return rc;
        out.upd.lambda_prime = m->lambda_prime;
        for (int i = 0; !rc && i < max_iter_tr; ++i) {       // onlinelda.cpp:89-101
            if (!(i > 0 && init_gamma))
                rc = fresh_gamma_device(m, B);               // lda.cpp:135
            const bool last = i + 1 == max_iter_tr;
            out.active_only = !(keep && last);
            out.upd.sstats = (keep && last) ? m->sstats : nullptr;
            // the next iteration's E-step runs on the same batch: its exp(psi(lambda)) and row
            // sums come out of this iteration's M-step (after the last one only when every word
            // was written: whatever batch comes next is covered)
            out.next_base = out.active_only ? m->rs_static : nullptr;
            out.emit_next = can_emit_next(m, b) && (!last || !out.active_only);
            // (big tables: the same for exp(psi(lambda)) alone, while the batch comes round again)
            // (K > 256: there the document kernel carries the factor multiply for nothing measurable
            // -- coefficients from scalar registers, factors in LDS; at K = 129 .. 256 it cost 1.9 % of
            // the document launches, more than exp_elog_beta_kernel takes at large batches)
            out.emit_u = !last && m->big_emit && m->carry_rowsums && K > 256;
            if (!rc)
                rc = estep_device(m, b, m->gamma, out, max_iter_inference, threshold, nullptr);
            if (!rc)
                rc = finish_rowsums(m, out, out.next_base, floor_after, b);
            if (!rc && out.u_emitted) {
                m->u_left.valid = true;
                m->u_left.batch_id = b->id;
                m->u_left.version = m->lambda_version;
            }
        }
        return rc;
    }

    // onlinelda.cpp:103-109: one E-step on the old lambda, then the M-step in place
    rc = fresh_gamma_device(m, B);
    if (!rc) rc = ensure_rowsums(m);
    if (rc)
        return rc;
    if (keep) {
        HIP_TRY(hipMemcpyAsync(m->lambda_prime, m->lambda, KV * sizeof(double),
                               hipMemcpyDeviceToDevice, m->stream));
        out.active_only = false;
        out.upd.sstats = m->sstats;
        out.upd.lambda_prime = m->lambda_prime;
        out.emit_next = can_emit_next(m, b);
        rc = estep_device(m, b, m->gamma, out, max_iter_inference, threshold, nullptr);
        if (!rc) rc = finish_rowsums(m, out, nullptr, floor_after, b);
        return rc;
    }
    out.active_only = true;
    out.upd.lambda_prime = m->lambda;                        // in place
    // the words outside the batch: lambda = (1 - rho) lambda + rho eta.  After the E-step's row-sum
    // stage (small tables: it reads all of lambda) -- by auxiliary workgroups of its document launch
    // where that is a merged one (the row sums were read by the kernel before it), else by the
    // streaming kernel behind it
    out.inact_wanted = true;
    out.inact_a = 1. - rho; out.inact_b = rho * eta;
    rc = estep_device(m, b, m->gamma, out, max_iter_inference, threshold, nullptr);
    const double *part_static = m->partial;
    if (!rc && out.inact_rows > 0) {
        G = out.inact_rows;
        part_static = out.inact_part;
    } else if (!rc)
        rc = launch_inactive_update<trlda::ACT_KEEP, false>(m, 1. - rho, rho * eta, rho, eta, 0.,
                                                            b->active_flag, nullptr, m->lambda,
                                                            nullptr, &G);
    if (!rc) rc = batch_end(m, b);                           // the pass above read its flags
    invalidate_rowsums(m);
    if (!rc) rc = combine_rowsums(m, part_static, G, nullptr, m->rs_static);
    if (!rc) rc = finish_rowsums(m, out, m->rs_static, floor_after);
    return rc;
}

}  // namespace

extern "C" {

int trlda_model_online_update(trlda_model *m, const trlda_batch *b, int num_documents, double eta,
                              int max_iter_tr, int max_iter_inference, double kappa, double tau,
                              double rho, int init_gamma, int update_lambda, double threshold,
                              int *update_count, double *rho_out, double *gamma_out)
{
    // (the batches' indices are built on worker threads: trlda_batch_create)
    if (int rc_built = batch_wait(b))
        return rc_built;
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!b || !update_count || !rho_out)
        return fail(TRLDA_ERR_ARG, "NULL batch / update_count / rho_out");
    if (b->V != m->V)
        return fail(TRLDA_ERR_SHAPE, "batch was created for a different vocabulary size");
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
        const double scale = (double)num_documents / (double)B;

        if (fused_update_available(m) && stream_available(m)) {
            rc = online_update_fused(m, b, num_documents, eta, max_iter_tr, max_iter_inference, rho,
                                     init_gamma, threshold);
        } else {
            // the plain sequence: lambda' = lambda; E-step -> statistics -> blend
            HIP_TRY(hipMemcpyAsync(m->lambda_prime, m->lambda, KV * sizeof(double),
                                   hipMemcpyDeviceToDevice, m->stream));
            if (max_iter_tr > 0) {
                rc = tr_init_device(m, b, m->lambda_prime, rho, eta, num_documents);
                for (int i = 0; !rc && i < max_iter_tr; ++i) {   // onlinelda.cpp:89-101
                    if (!(i > 0 && init_gamma))
                        rc = fresh_gamma_device(m, B);
                    if (!rc)
                        rc = estep_device(m, b, m->gamma, m->sstats, max_iter_inference, threshold,
                                          nullptr);
                    if (!rc)
                        rc = blend_device(m, m->lambda_prime, m->sstats, rho, eta, scale);
                }
            } else {                                             // onlinelda.cpp:103-109
                rc = fresh_gamma_device(m, B);
                if (!rc)
                    rc = estep_device(m, b, m->gamma, m->sstats, max_iter_inference, threshold,
                                      nullptr);
                if (!rc)
                    rc = blend_device(m, m->lambda_prime, m->sstats, rho, eta, scale);
            }
        }
        if (rc)
            return rc;
        if (gamma_out) {
            HIP_TRY(hipMemcpyAsync(gamma_out, m->gamma, gbytes, hipMemcpyDeviceToHost, m->stream));
            m->d2h_bytes += (int64_t)gbytes;
            if (int rc_sync = sync_model(m))
                return rc_sync;
        }
        // no synchronisation otherwise: the kernels of this call run while the host draws the
        // next call's gamma0; every getter synchronises the stream it copies on
    }
    ++*update_count;                                         // onlinelda.cpp:177
    return TRLDA_OK;
}

int trlda_model_batch_update(trlda_model *m, const trlda_batch *b, double eta, int max_epochs,
                             int max_iter_inference, int update_lambda, double threshold,
                             double *gamma_out)
{
    // (the batches' indices are built on worker threads: trlda_batch_create)
    if (int rc_built = batch_wait(b))
        return rc_built;
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!b)
        return fail(TRLDA_ERR_ARG, "NULL batch");
    if (b->V != m->V)
        return fail(TRLDA_ERR_SHAPE, "batch was created for a different vocabulary size");
    if (b->B == 0)                                           // batchlda.cpp:44-46
        return TRLDA_OK;
    rc = ensure_update_workspace(m, b->B);
    if (rc)
        return rc;
    const int K = m->K, B = b->B;
    const size_t KV = (size_t)K * m->V;
    const size_t gbytes = (size_t)K * B * sizeof(double);
    const bool fused = fused_update_available(m) && stream_available(m);
    bool first = true;
    for (int epoch = 0; epoch < max_epochs; ++epoch) {       // batchlda.cpp:48-61
        if (!update_lambda)
            continue;
        rc = fresh_gamma_device(m, B);
        if (rc)
            return rc;
        if (fused) {
            // lambda = eta + sstats (batchlda.cpp:60): the statistics kernel writes it for the
            // batch's words; the others are eta, written once per call
            rc = ensure_rowsums(m);
            EstepOut out;
            out.upd.omr = 0.; out.upd.rho = 1.; out.upd.eta = eta; out.upd.scale = 1.;
            out.upd.lambda = m->lambda;
            out.upd.lambda_prime = nullptr;
            out.upd.partial = m->upd_partial;
            out.upd.sstats = m->keep_sstats ? m->sstats : nullptr;
            out.active_only = !m->keep_sstats;
            // (first epoch: the share of the words outside the batch is not known yet)
            out.next_base = out.active_only ? m->rs_static : nullptr;
            out.emit_next = can_emit_next(m, b) && !(first && out.active_only) && epoch + 1 < max_epochs;
            if (!rc)
                rc = estep_device(m, b, m->gamma, out, max_iter_inference, threshold, nullptr);
            if (!rc && first && out.active_only) {
                int G = 0;
                rc = launch_inactive_update<trlda::ACT_KEEP, false>(m, 0., eta, 1., eta, 0.,
                                                                    b->active_flag, nullptr,
                                                                    m->lambda, nullptr, &G);
                if (!rc) rc = batch_end(m, b);
                invalidate_rowsums(m);
                if (!rc) rc = combine_rowsums(m, m->partial, G, nullptr, m->rs_static);
            }
            if (!rc)
                rc = finish_rowsums(m, out, out.active_only ? m->rs_static : nullptr, m->V * eta, b);
            first = false;
        } else {
            rc = estep_device(m, b, m->gamma, m->sstats, max_iter_inference, threshold, nullptr);
            if (!rc) {
                invalidate_rowsums(m);
                m->lambda_positive = false;                 // (not tracked through this path)
                m->rs_floor = m->V * eta;
                rc = launch_elementwise(m, KV, trlda::SetOp{eta, m->sstats, m->lambda});
            }
        }
        if (rc)
            return rc;
    }
    if (gamma_out && update_lambda && max_epochs > 0) {
        HIP_TRY(hipMemcpyAsync(gamma_out, m->gamma, gbytes, hipMemcpyDeviceToHost, m->stream));
        m->d2h_bytes += (int64_t)gbytes;
    }
    if (int rc_sync = sync_model(m))
        return rc_sync;
    return TRLDA_OK;
}

int trlda_model_cumulative_update(trlda_model *m, const trlda_batch *b, int max_epochs,
                                  int max_iter_inference, int update_lambda, double threshold,
                                  double *gamma_out)
{
    // (the batches' indices are built on worker threads: trlda_batch_create)
    if (int rc_built = batch_wait(b))
        return rc_built;
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!b)
        return fail(TRLDA_ERR_ARG, "NULL batch");
    if (b->V != m->V)
        return fail(TRLDA_ERR_SHAPE, "batch was created for a different vocabulary size");
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
    const double floor_prime = m->rs_floor;
    const bool prime_positive = m->lambda_positive;          // of lambda', which the M-steps add to
    if (m->host_gamma_draw) {
        std::vector<double> lam0(KV);
        trlda_sample_gamma_init(K, m->V, lam0.data());
        HIP_TRY(hipMemcpyAsync(m->lambda, lam0.data(), KV * sizeof(double), hipMemcpyHostToDevice,
                               m->stream));
        note_host_lambda(m, lam0.data());
        if (int rc_sync = sync_model(m))
            return rc_sync;
    } else {
        invalidate_rowsums(m);
        m->rs_floor = 0.0;
        rc = sample_gamma_on_device(m, (long long)KV, 100, 100., m->lambda);
        if (!rc && stream_available(m)) {
            // its row sums: carried to the first E-step, and their minimum decides whether the
            // fused small-table preamble is safe (K numbers come back)
            rc = rowsums_from_scratch(m);
            std::vector<double> rs((size_t)K);
            if (!rc) {
                HIP_TRY(hipMemcpyAsync(rs.data(), m->rs_full, (size_t)K * sizeof(double),
                                       hipMemcpyDeviceToHost, m->stream));
                if (int rc_sync = sync_model(m))
                    return rc_sync;
                m->d2h_bytes += (int64_t)K * sizeof(double);
                double lo = rs[0];
                for (int k = 1; k < K; ++k)
                    lo = std::min(lo, rs[(size_t)k]);
                m->rs_floor = lo > 0.0 ? 0.999 * lo : 0.0;
                m->rs_valid = true;
            }
        }
        if (rc)
            return rc;
    }
    const bool fused = fused_update_available(m) && stream_available(m);
    bool ran = false;
    // (the drawn lambda is positive; what mstep_keeps_positive asks about is lambda' + statistics)
    m->lambda_positive = prime_positive;
    if (update_lambda) {
        for (int epoch = 0; epoch < max_epochs; ++epoch) {    // cumulativelda.cpp:62-71
            rc = fresh_gamma_device(m, B);
            if (rc)
                return rc;
            if (fused) {
                // lambda = lambda' + sstats for every word (cumulativelda.cpp:70)
                rc = ensure_rowsums(m);
                EstepOut out;
                out.upd.omr = 1.; out.upd.rho = 1.; out.upd.eta = 0.; out.upd.scale = 1.;
                out.upd.lambda = m->lambda;
                out.upd.lambda_prime = m->lambda_prime;
                out.upd.partial = m->upd_partial;
                out.upd.sstats = m->keep_sstats ? m->sstats : nullptr;
                out.active_only = false;
                out.emit_next = can_emit_next(m, b);
                if (!rc)
                    rc = estep_device(m, b, m->gamma, out, max_iter_inference, threshold, nullptr);
                if (!rc)
                    rc = finish_rowsums(m, out, nullptr, floor_prime, b);
            } else {
                rc = estep_device(m, b, m->gamma, m->sstats, max_iter_inference, threshold, nullptr);
                if (!rc) {
                    invalidate_rowsums(m);
                    m->lambda_positive = false;                 // (not tracked through this path)
                    m->rs_floor = floor_prime;
                    rc = launch_elementwise(
                        m, KV, trlda::AccumulateOp{m->lambda_prime, m->sstats, m->lambda});
                }
            }
            if (rc)
                return rc;
            ran = true;
        }
    }
    if (gamma_out && ran) {
        HIP_TRY(hipMemcpyAsync(gamma_out, m->gamma, gbytes, hipMemcpyDeviceToHost, m->stream));
        m->d2h_bytes += (int64_t)gbytes;
    }
    if (int rc_sync = sync_model(m))
        return rc_sync;
    return TRLDA_OK;
}

// ---- multi-GPU composition over RCCL, no Python in between --------------------------------
//
// One process per GPU; documents shard across ranks, lambda is replicated.  The only exchange
// of the path is the sum of the K x V statistics where the reference has its `omp critical`
// reduction (src/lda.cpp:211-217) -- one ncclAllReduce on the model's stream -- plus one sum of
// V word counts for the trust-region initial step (src/onlinelda.cpp:79-82).  The host program
// owns the communicator (ncclCommInitRank) and hands it over as an opaque pointer; RCCL is
// looked up at run time in the process (the host already links it) or as librccl.so, so this
// library carries no link-time dependency on it.
} // extern "C"

namespace {

using nccl_allreduce_fn = int (*)(const void *, void *, size_t, int, int, void *, hipStream_t);

nccl_allreduce_fn rccl_allreduce()
{
    static nccl_allreduce_fn fn = reinterpret_cast<nccl_allreduce_fn>(rccl_symbol("ncclAllReduce"));
    return fn;
}

int allreduce_f64(trlda_model *m, void *comm, double *buf, size_t count)
{
    if (!comm)
        return fail(TRLDA_ERR_ARG, "RCCL communicator is NULL");
    nccl_allreduce_fn fn = rccl_allreduce();
    if (!fn)
        return fail(TRLDA_ERR_ARG, "ncclAllReduce not found: load RCCL (librccl.so) into the process");
    const int rc = fn(buf, buf, count, kNcclFloat64, kNcclSum, comm, m->stream);
    if (rc != 0)
        return fail(TRLDA_ERR_HIP, "ncclAllReduce failed with ncclResult_t " + std::to_string(rc));
    return TRLDA_OK;
}

// m->gamma = columns [doc_lo, doc_lo + Bl) of sampleGamma(K, B, 100) / 100 for a mini-batch of B
// documents (lda.cpp:135): gamma0 of the whole mini-batch from the (shared) host stream, this
// rank's columns kept -- every rank consumes the stream exactly as the single-process run does
int fresh_gamma_columns(trlda_model *m, int B, int doc_lo, int Bl, std::vector<double> &full)
{
    const int K = m->K;
    if (!m->host_gamma_draw)                                 // this rank's columns, on the device
        return device_gamma_now_or_ahead(m, (long long)K * B, (long long)K * doc_lo,
                                         (long long)K * (doc_lo + Bl));
    full.resize((size_t)K * B);
    trlda_sample_gamma_init(K, B, full.data());
    if (Bl > 0) {
        m->gamma0_src = nullptr;
        HIP_TRY(hipMemcpyAsync(m->gamma, full.data() + (size_t)K * doc_lo,
                               (size_t)K * Bl * sizeof(double), hipMemcpyHostToDevice, m->stream));
        if (int rc_sync = sync_model(m))
            return rc_sync;            // `full` is reused by the next draw
    }
    return TRLDA_OK;
}

int check_shard_range(const trlda_model *m, const trlda_batch *shard, int total_docs, int doc_lo)
{
    if (!shard)
        return fail(TRLDA_ERR_ARG, "NULL shard");
    if (shard->V != m->V)
        return fail(TRLDA_ERR_SHAPE, "batch was created for a different vocabulary size");
    if (total_docs < 0 || doc_lo < 0 || doc_lo + shard->B > total_docs)
        return fail(TRLDA_ERR_ARG, "shard [doc_lo, doc_lo + its size) must lie inside [0, total_docs)");
    return TRLDA_OK;
}

}  // namespace

extern "C" {

int trlda_model_allreduce_sstats(trlda_model *m, void *rccl_comm, double *sstats_dev)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!sstats_dev)
        return fail(TRLDA_ERR_ARG, "sstats is NULL");
    return allreduce_f64(m, rccl_comm, sstats_dev, (size_t)m->K * m->V);
}

int trlda_model_online_update_multi(trlda_model *m, const trlda_batch *shard, void *rccl_comm,
                                    int total_docs, int doc_lo, int num_documents, double eta,
                                    int max_iter_tr, int max_iter_inference, double kappa, double tau,
                                    double rho, int init_gamma, double threshold, int *update_count,
                                    double *rho_out)
{
    // (the batches' indices are built on worker threads: trlda_batch_create)
    if (int rc_built = batch_wait(shard))
        return rc_built;
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!update_count || !rho_out)
        return fail(TRLDA_ERR_ARG, "NULL update_count / rho_out");
    if ((rc = check_shard_range(m, shard, total_docs, doc_lo)))
        return rc;
    const int K = m->K, Bl = shard->B, B = total_docs;
    if (B == 0) {                                            // onlinelda.cpp:54-56
        *rho_out = 1.0;
        return TRLDA_OK;
    }
    if (rho < 0.)                                            // onlinelda.cpp:59-66
        rho = std::pow(tau + (double)*update_count, -kappa);
    *rho_out = rho;
    rc = ensure_update_workspace(m, Bl);
    if (rc)
        return rc;
    const size_t KV = (size_t)K * m->V;
    const double scale = (double)num_documents / (double)B;
    std::vector<double> full;
    auto fresh_gamma = [&]() -> int { return fresh_gamma_columns(m, B, doc_lo, Bl, full); };
    HIP_TRY(hipMemcpyAsync(m->lambda_prime, m->lambda, KV * sizeof(double), hipMemcpyDeviceToDevice,
                           m->stream));                      // lambdaPrime = mLambda  (:68)
    const int steps = max_iter_tr > 0 ? max_iter_tr : 1;
    if (max_iter_tr > 0) {                                   // onlinelda.cpp:79-86
        rc = wordcounts_device(m, shard, m->wordcounts);
        if (!rc) rc = allreduce_f64(m, rccl_comm, m->wordcounts, (size_t)m->V);
        const double coef = (double)num_documents / (double)B / (double)K;
        if (!rc) rc = tr_init_wc_device(m, m->wordcounts, m->lambda_prime, rho, eta, coef);
    }
    for (int i = 0; !rc && i < steps; ++i) {                 // onlinelda.cpp:89-101 / :103-109
        if (!(i > 0 && init_gamma))
            rc = fresh_gamma();
        if (!rc) rc = estep_device(m, shard, m->gamma, m->sstats, max_iter_inference, threshold, nullptr);
        if (!rc) rc = allreduce_f64(m, rccl_comm, m->sstats, KV);    // lda.cpp:211-217 across ranks
        if (!rc) rc = blend_device(m, m->lambda_prime, m->sstats, rho, eta, scale);
    }
    if (rc)
        return rc;
    ++*update_count;                                         // onlinelda.cpp:177
    return TRLDA_OK;
}

int trlda_model_allreduce(trlda_model *m, void *rccl_comm, double *buf_dev, size_t count)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!buf_dev && count)
        return fail(TRLDA_ERR_ARG, "buffer is NULL");
    return count ? allreduce_f64(m, rccl_comm, buf_dev, count) : TRLDA_OK;
}

// BatchLDA::updateParameters' lambda path (src/batchlda.cpp:43-61) over the ranks of rccl_comm:
// per epoch, this rank's documents from a fresh gamma, ONE all-reduce of the K x V statistics
// (src/lda.cpp:211-217 across ranks), lambda = eta + sstats on every rank.
int trlda_model_batch_update_multi(trlda_model *m, const trlda_batch *shard, void *rccl_comm,
                                   int total_docs, int doc_lo, double eta, int max_epochs,
                                   int max_iter_inference, int update_lambda, double threshold)
{
    // (the batches' indices are built on worker threads: trlda_batch_create)
    if (int rc_built = batch_wait(shard))
        return rc_built;
    int rc = check_model(m);
    if (rc)
        return rc;
    if ((rc = check_shard_range(m, shard, total_docs, doc_lo)))
        return rc;
    if (total_docs == 0)                                     // batchlda.cpp:44-46
        return TRLDA_OK;
    rc = ensure_update_workspace(m, shard->B);
    if (rc)
        return rc;
    const size_t KV = (size_t)m->K * m->V;
    std::vector<double> full;
    for (int epoch = 0; !rc && update_lambda && epoch < max_epochs; ++epoch) {   // batchlda.cpp:48-61
        rc = fresh_gamma_columns(m, total_docs, doc_lo, shard->B, full);
        if (!rc) rc = estep_device(m, shard, m->gamma, m->sstats, max_iter_inference, threshold, nullptr);
        if (!rc) rc = allreduce_f64(m, rccl_comm, m->sstats, KV);
        if (!rc) {
            invalidate_rowsums(m);
            m->lambda_positive = false;                 // (not tracked through this path)
            m->rs_floor = m->V * eta;
            rc = launch_elementwise(m, KV, trlda::SetOp{eta, m->sstats, m->lambda});   // :60
        }
    }
    if (rc)
        return rc;
    if (int rc_sync = sync_model(m))
        return rc_sync;
    return TRLDA_OK;
}

// LDA::updateVariables(documents, parameters) from a fresh random gamma (src/lda.cpp:119-138) for
// this rank's documents of a sharded mini-batch, gamma left on the device: what
// src/onlinelda.cpp:118-120 / src/batchlda.cpp:66-68 do before an alpha step when update_lambda
// is off.  No exchange: the statistics are not used.
int trlda_model_estep_resident_shard(trlda_model *m, const trlda_batch *shard, int total_docs,
                                     int doc_lo, int max_iter, double threshold)
{
    // (the batches' indices are built on worker threads: trlda_batch_create)
    if (int rc_built = batch_wait(shard))
        return rc_built;
    int rc = check_model(m);
    if (rc)
        return rc;
    if ((rc = check_shard_range(m, shard, total_docs, doc_lo)))
        return rc;
    if (total_docs == 0)
        return TRLDA_OK;
    rc = ensure_update_workspace(m, shard->B);
    std::vector<double> full;
    if (!rc) rc = fresh_gamma_columns(m, total_docs, doc_lo, shard->B, full);
    if (!rc && shard->B > 0)
        rc = estep_device(m, shard, m->gamma, m->sstats, max_iter, threshold, nullptr);
    return rc;
}

// ---- data parallelism with factor exchange (dp_kernels.h) ---------------------------------

// The direct exchange: a region of this model that the peers map (hipIpc) and write their slots
// into.  alloc -> the 64-byte handle goes to every peer by whatever means the host has ->
// connect with all handles (rank order).  Fine-grained device memory where the runtime exports
// it (the step counters are polled while kernels run), ordinary device memory otherwise.
int trlda_model_dp_direct_alloc(trlda_model *m, size_t max_slot_f64, int world, void *handle_out)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!handle_out || world < 1 || world > trlda::kDpMaxWorld || max_slot_f64 == 0)
        return fail(TRLDA_ERR_ARG, "bad dp_direct_alloc arguments");
    if ((rc = trlda_model_dp_direct_close(m)))
        return rc;
    max_slot_f64 = (max_slot_f64 + 1) & ~(size_t)1;
    const size_t bytes = 2 * (size_t)world * max_slot_f64 * sizeof(double) +
                         (size_t)trlda::kDpMaxWorld * sizeof(unsigned long long);
    // Fine-grained device memory or nothing: the step counters and the slots are polled and read
    // while kernels of OTHER devices write them, which ordinary (coarse-grained) device memory
    // does not promise to show before a kernel boundary.  No silent fallback: the
    // caller's ranks then stay on the all-gather together (ShardedOnlineLDA, bench.py).
    void *region = nullptr;
    hipIpcMemHandle_t handle;
    hipError_t e = hipExtMallocWithFlags(&region, bytes, hipDeviceMallocFinegrained);
    if (e == hipSuccess)
        e = hipIpcGetMemHandle(&handle, region);
    if (e != hipSuccess) {
        if (region)
            (void)hipFree(region);
        (void)hipGetLastError();
        return fail(TRLDA_ERR_HIP, std::string("the direct exchange needs fine-grained device memory that can "
                                               "be exported through hipIpc: ") + hipGetErrorString(e));
    }
    HIP_TRY(hipMemset(region, 0, bytes));
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the ABI hands out 64-byte handles");
    std::memcpy(handle_out, &handle, sizeof(handle));
    m->direct.region = region;
    m->direct.max_slot = max_slot_f64;
    m->direct.world = world;
    m->direct.step = 0;
    return TRLDA_OK;
}

int trlda_model_dp_direct_connect(trlda_model *m, int rank, int world, const void *handles)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!m->direct.region || world != m->direct.world || rank < 0 || rank >= world || !handles)
        return fail(TRLDA_ERR_ARG, "dp_direct_connect: alloc first, with the same world; rank in range");
    const size_t flag_off = 2 * (size_t)world * m->direct.max_slot * sizeof(double);
    for (int r = 0; r < world; ++r) {
        void *base = m->direct.region;
        if (r != rank) {
            hipIpcMemHandle_t h;
            std::memcpy(&h, static_cast<const char *>(handles) + (size_t)r * sizeof(h), sizeof(h));
            hipError_t e = hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) {
                (void)trlda_model_dp_direct_close(m);
                return fail(TRLDA_ERR_HIP, "hipIpcOpenMemHandle (rank " + std::to_string(r) +
                                               "): " + hipGetErrorString(e));
            }
            m->direct.opened.push_back(base);
        }
        m->direct.peers.buf[r] = static_cast<double *>(base);
        m->direct.peers.flags[r] = reinterpret_cast<unsigned long long *>(static_cast<char *>(base) + flag_off);
    }
    m->direct.rank = rank;
    m->direct.connected = true;
    return TRLDA_OK;
}

int trlda_model_dp_direct_close(trlda_model *m)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "model is NULL");
    if (m->direct.region || !m->direct.opened.empty()) {
        if (m->stream)
            (void)hipStreamSynchronize(m->stream);
        for (void *p : m->direct.opened)
            (void)hipIpcCloseMemHandle(p);
        m->direct.opened.clear();
        if (m->direct.region)
            (void)hipFree(m->direct.region);
    }
    m->direct.region = nullptr;
    m->direct.connected = false;
    m->direct.max_slot = 0;
    m->direct.world = 0;
    m->direct.rank = -1;
    m->dp_gather_direct = nullptr;
    return TRLDA_OK;
}

int trlda_model_set_allgather(trlda_model *m, trlda_allgather_fn fn, void *ctx)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "model is NULL");
    m->allgather_hook = fn;
    m->allgather_ctx = ctx;
    return TRLDA_OK;
}

int trlda_model_set_allgatherv(trlda_model *m, trlda_allgatherv_fn fn, void *ctx)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "model is NULL");
    m->allgatherv_hook = fn;
    m->allgatherv_ctx = ctx;
    return TRLDA_OK;
}

int trlda_model_set_word_sharding(trlda_model *m, int enabled)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "model is NULL");
    m->word_sharding = enabled != 0;
    return TRLDA_OK;
}

int trlda_model_last_word_sharded(const trlda_model *m) { return m && m->last_word_sharded ? 1 : 0; }

int trlda_model_set_split_lists(trlda_model *m, int enabled)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "model is NULL");
    m->split_long_lists = enabled != 0;
    return TRLDA_OK;
}

} // extern "C"

namespace {

int dp_enter(trlda_model *m, DpContext &dp, const trlda_batch *batch, const trlda_batch *shard,
             void *rccl_comm, int rank, int world, const int32_t *doc_cuts)
{
    if (!batch || !shard || !doc_cuts)
        return fail(TRLDA_ERR_ARG, "NULL batch / shard / doc_cuts");
    if (batch->V != m->V || shard->V != m->V)
        return fail(TRLDA_ERR_SHAPE, "batch was created for a different vocabulary size");
    if (world < 1 || world > trlda::kDpMaxWorld)
        return fail(TRLDA_ERR_ARG, "world must lie in [1, 64]");
    dp.shard = shard;
    dp.rank = rank;
    dp.world = world;
    dp.comm = rccl_comm;
    dp.cuts.assign(doc_cuts, doc_cuts + world + 1);
    m->last_word_sharded = false;
    int rc = dp_prepare(m, batch, &dp);
    if (!rc)
        m->dp = &dp;
    return rc;
}

}  // namespace

extern "C" {

int trlda_model_online_update_dp(trlda_model *m, const trlda_batch *batch, const trlda_batch *shard,
                                 void *rccl_comm, int rank, int world, const int32_t *doc_cuts,
                                 int num_documents, double eta, int max_iter_tr, int max_iter_inference,
                                 double kappa, double tau, double rho, int init_gamma, double threshold,
                                 int *update_count, double *rho_out)
{
    // (the batches' indices are built on worker threads: trlda_batch_create)
    if (int rc_built = batch_wait(batch))
        return rc_built;
    if (int rc_built = batch_wait(shard))
        return rc_built;
    int rc = check_model(m);
    if (rc)
        return rc;
    DpContext dp;
    if ((rc = dp_enter(m, dp, batch, shard, rccl_comm, rank, world, doc_cuts)))
        return rc;
    // the single-GPU call on the whole mini-batch; its document stage runs on the shard and an
    // exchange follows it (estep_device), everything else -- word counts, the initial step, the
    // fused M-steps with carried row sums -- is replicated work on replicated data
    rc = trlda_model_online_update(m, batch, num_documents, eta, max_iter_tr, max_iter_inference, kappa,
                                   tau, rho, init_gamma, 1, threshold, update_count, rho_out, nullptr);
    m->dp = nullptr;
    return rc;
}

int trlda_model_batch_update_dp(trlda_model *m, const trlda_batch *batch, const trlda_batch *shard,
                                void *rccl_comm, int rank, int world, const int32_t *doc_cuts,
                                double eta, int max_epochs, int max_iter_inference, int update_lambda,
                                double threshold)
{
    // (the batches' indices are built on worker threads: trlda_batch_create)
    if (int rc_built = batch_wait(batch))
        return rc_built;
    if (int rc_built = batch_wait(shard))
        return rc_built;
    int rc = check_model(m);
    if (rc)
        return rc;
    DpContext dp;
    if ((rc = dp_enter(m, dp, batch, shard, rccl_comm, rank, world, doc_cuts)))
        return rc;
    // the single-GPU call on the whole mini-batch, its document stage on the shard
    rc = trlda_model_batch_update(m, batch, eta, max_epochs, max_iter_inference, update_lambda,
                                  threshold, nullptr);
    m->dp = nullptr;
    return rc;
}

int trlda_model_estep_dp(trlda_model *m, const trlda_batch *batch, const trlda_batch *shard,
                         void *rccl_comm, int rank, int world, const int32_t *doc_cuts,
                         const double *gamma0_dev, double *gamma_dev, double *sstats_dev, int max_iter,
                         double threshold, int32_t *iters_dev, int mstep,
                         const double *lambda_prime_dev, double rho, double eta, double scale)
{
    // (the batches' indices are built on worker threads: trlda_batch_create)
    if (int rc_built = batch_wait(batch))
        return rc_built;
    if (int rc_built = batch_wait(shard))
        return rc_built;
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!gamma_dev)
        return fail(TRLDA_ERR_ARG, "gamma is NULL");
    if (!mstep && !sstats_dev)
        return fail(TRLDA_ERR_ARG, "nothing to write: no sstats buffer and no M-step");
    if (mstep && !fused_update_available(m))
        return fail(TRLDA_ERR_ARG, "the fused M-step is not available for this model (K > 512 or switched off)");
    DpContext dp;
    if ((rc = dp_enter(m, dp, batch, shard, rccl_comm, rank, world, doc_cuts)))
        return rc;
    EstepOut out(sstats_dev);
    if (mstep) {
        // lambda = (1 - rho) lambda' + rho (eta + scale * sstats) for every word
        // (onlinelda.cpp:99-100 / :107-108), in the statistics kernel
        out.upd.omr = 1. - rho; out.upd.rho = rho; out.upd.eta = eta; out.upd.scale = scale;
        out.upd.lambda = m->lambda;
        out.upd.lambda_prime = lambda_prime_dev ? lambda_prime_dev : m->lambda;
        out.upd.partial = m->upd_partial;
        out.active_only = false;
        out.emit_next = can_emit_next(m, batch);
        rc = ensure_update_workspace(m, 1);
        if (!rc) rc = ensure_rowsums(m);
    }
    if (!rc)
        rc = estep_device(m, batch, gamma_dev, out, max_iter, threshold, iters_dev, gamma0_dev);
    if (!rc && mstep)
        rc = finish_rowsums(m, out, nullptr, rho * m->V * eta, batch);
    m->dp = nullptr;
    return rc;
}

// ---- empirical Bayes / adaptive rate: device reductions, K-sized results ------------------

int trlda_model_set_keep_sstats(trlda_model *m, int keep)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "model is NULL");
    m->keep_sstats = keep != 0;
    return TRLDA_OK;
}

int trlda_model_set_carry_rowsums(trlda_model *m, int carry)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "model is NULL");
    m->carry_rowsums = carry != 0;
    return TRLDA_OK;
}

int trlda_model_set_next_preamble(trlda_model *m, int enabled)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "model is NULL");
    m->emit_next_preamble = enabled != 0;
    m->next_pre.valid = false;
    return TRLDA_OK;
}

int trlda_model_set_fused_update(trlda_model *m, int fused)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "model is NULL");
    m->fused_update = fused != 0;
    invalidate_rowsums(m);
    return TRLDA_OK;
}

int64_t trlda_model_d2h_bytes(const trlda_model *m) { return m ? m->d2h_bytes : 0; }

int trlda_model_set_draw_ahead(trlda_model *m, int enabled)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "model is NULL");
    // 0: every draw in its turn; 1: ahead on a side stream; 2 (the default): ahead inside the call's
    // document launch where that launch can carry it (estep_merged.h, AuxArgs), in its turn elsewhere
    m->draw_ahead = enabled == 1;
    m->draw_inlaunch = enabled == 2;
    m->aux_draw_req.valid = false;
    return TRLDA_OK;
}

long long trlda_model_inlaunch_draws(const trlda_model *m) { return m ? (long long)m->inlaunch_draws : 0; }

int trlda_model_set_aux_decay(trlda_model *m, int enabled)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "model is NULL");
    m->aux_decay = enabled != 0;
    return TRLDA_OK;
}

long long trlda_model_inlaunch_decays(const trlda_model *m) { return m ? (long long)m->inlaunch_decays : 0; }

int trlda_model_set_host_gamma_draw(trlda_model *m, int host)
{
    if (!m)
        return fail(TRLDA_ERR_ARG, "model is NULL");
    m->host_gamma_draw = host != 0;
    return TRLDA_OK;
}

int trlda_model_sample_gamma(trlda_model *m, int rows, int cols, int passes, double divisor,
                             double *out_dev)
{
    return trlda_model_sample_gamma_cols(m, rows, cols, 0, cols, passes, divisor, out_dev);
}

int trlda_model_sample_gamma_cols(trlda_model *m, int rows, int cols, int col_lo, int col_hi,
                                  int passes, double divisor, double *out_dev)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (rows < 0 || cols < 0 || passes < 0 || !out_dev || divisor == 0. || col_lo < 0 ||
        col_hi < col_lo || col_hi > cols)
        return fail(TRLDA_ERR_ARG, "bad sample_gamma arguments");
    return sample_gamma_on_device(m, (long long)rows * cols, passes, divisor, out_dev,
                                  (long long)rows * col_lo, (long long)rows * col_hi);
}

int trlda_model_estep_resident(trlda_model *m, const trlda_batch *b, int max_iter, double threshold)
{
    // (the batches' indices are built on worker threads: trlda_batch_create)
    if (int rc_built = batch_wait(b))
        return rc_built;
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!b)
        return fail(TRLDA_ERR_ARG, "NULL batch");
    if (b->B == 0)
        return TRLDA_OK;
    rc = ensure_update_workspace(m, b->B);
    if (!rc) rc = fresh_gamma_device(m, b->B);               // lda.cpp:135
    if (!rc) rc = estep_device(m, b, m->gamma, m->sstats, max_iter, threshold, nullptr);
    return rc;
}

int trlda_model_eb_gamma_stats(trlda_model *m, int B, const double *gamma_dev, double *out_host)
{
    if (B <= 0)
        return fail(TRLDA_ERR_ARG, "bad eb_gamma_stats arguments");
    return trlda_model_eb_gamma_stats_multi(m, nullptr, B, gamma_dev, out_host);
}

// the same sums over the documents of every rank: this rank's K numbers (zeros for a rank
// without documents), one all-reduce of K doubles (src/onlinelda.cpp:128 across ranks)
int trlda_model_eb_gamma_stats_multi(trlda_model *m, void *rccl_comm, int B, const double *gamma_dev,
                                     double *out_host)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (B < 0 || !out_host)
        return fail(TRLDA_ERR_ARG, "bad eb_gamma_stats arguments");
    if (B == 0) {
        rc = grow(&m->reduce_out, &m->cap_reduce, (size_t)m->K);
        if (rc)
            return rc;
        HIP_TRY(hipMemsetAsync(m->reduce_out, 0, (size_t)m->K * sizeof(double), m->stream));
        if (rccl_comm && (rc = allreduce_f64(m, rccl_comm, m->reduce_out, (size_t)m->K)))
            return rc;
        HIP_TRY(hipMemcpyAsync(out_host, m->reduce_out, (size_t)m->K * sizeof(double),
                               hipMemcpyDeviceToHost, m->stream));
        m->d2h_bytes += (int64_t)m->K * sizeof(double);
        if (int rc_sync = sync_model(m))
            return rc_sync;
        return TRLDA_OK;
    }
    const double *gamma = gamma_dev ? gamma_dev : m->gamma;
    if (!gamma || (!gamma_dev && (size_t)B * m->K > m->cap_gamma))
        return fail(TRLDA_ERR_ARG, "no gamma of that size is resident in the model");
    const int K = m->K;
    const int chunks = (B + trlda::kEbDocsPerBlock - 1) / trlda::kEbDocsPerBlock;
    rc = grow(&m->reduce_out, &m->cap_reduce, (size_t)(chunks + 1) * K);
    if (rc)
        return rc;
    constexpr int T = 256;
    const size_t lds = ((size_t)K + T / trlda::kWave + 1) * sizeof(double);
    auto kern = trlda::eb_gamma_kernel<T>;
    if ((rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds)))
        return rc;
    hipLaunchKernelGGL(kern, dim3(chunks), dim3(T), lds, m->stream, K, B, gamma, m->reduce_out);
    HIP_TRY(hipGetLastError());
    double *sum = m->reduce_out + (size_t)chunks * K;
    rc = combine_rowsums(m, m->reduce_out, chunks, nullptr, sum);
    if (!rc && rccl_comm)
        rc = allreduce_f64(m, rccl_comm, sum, (size_t)K);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpyAsync(out_host, sum, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, m->stream));
    m->d2h_bytes += (int64_t)K * sizeof(double);
    if (int rc_sync = sync_model(m))
        return rc_sync;
    return TRLDA_OK;
}

int trlda_model_eb_lambda_stats(trlda_model *m, double *sum_psi_lambda, double *rowsums_host)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!sum_psi_lambda || !rowsums_host)
        return fail(TRLDA_ERR_ARG, "bad eb_lambda_stats arguments");
    const int K = m->K;
    const size_t KV = (size_t)K * m->V;
    constexpr int T = 256;
    const int G = (int)std::max<size_t>(1, std::min<size_t>((KV + 4 * T - 1) / (4 * T), 2048));
    rc = grow(&m->reduce_out, &m->cap_reduce, (size_t)G + (size_t)K);
    if (rc)
        return rc;
    hipLaunchKernelGGL(trlda::eb_lambda_kernel<T>, dim3(G), dim3(T), 0, m->stream, KV, m->lambda,
                       m->reduce_out);
    HIP_TRY(hipGetLastError());
    // the row sums: carried by the kernel that wrote lambda, else added up now
    const double *rs = m->rs_full;
    if (rowsums_carried(m)) {
        rc = resolve_carry(m);
        if (rc)
            return rc;
    } else {
        if (stream_available(m)) {
            rc = rowsums_from_scratch(m);
            if (rc)
                return rc;
            if (!m->lambda_exposed)
                m->rs_valid = true;
        } else {
            int GR = std::min(kMaxRowsumBlocks - 1, std::max(1, m->V / 32));
            int wpb = (m->V + GR - 1) / GR;
            GR = (m->V + wpb - 1) / wpb;
            hipLaunchKernelGGL(trlda::rowsum_partial_kernel<kDenseThreads>, dim3(GR),
                               dim3(kDenseThreads), 0, m->stream, K, m->V, wpb, m->lambda, m->partial);
            HIP_TRY(hipGetLastError());
            rc = combine_rowsums(m, m->partial, GR, nullptr, m->rs_full);
            if (rc)
                return rc;
        }
    }
    std::vector<double> blocks((size_t)G);
    HIP_TRY(hipMemcpyAsync(blocks.data(), m->reduce_out, (size_t)G * sizeof(double),
                           hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipMemcpyAsync(rowsums_host, rs, (size_t)K * sizeof(double), hipMemcpyDeviceToHost,
                           m->stream));
    m->d2h_bytes += (int64_t)((size_t)G + K) * sizeof(double);
    if (int rc_sync = sync_model(m))
        return rc_sync;
    double total = 0.0;
    for (int g = 0; g < G; ++g)
        total += blocks[(size_t)g];
    *sum_psi_lambda = total;
    return TRLDA_OK;
}

// The empirical-Bayes steps of OnlineLDA::updateParameters (src/onlinelda.cpp:116-162) with ONE
// trip to the host: the device sums over gamma (and over the ranks, when a communicator is
// given) and over lambda are enqueued, K + G + K doubles come back in one synchronisation, the
// K-sized Newton steps run here (eb_steps.cpp), the new alpha goes back to the device.
int trlda_model_online_eb_begin(trlda_model *m, void *rccl_comm, int B_local, int B_total,
                                int update_alpha, int update_eta)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (B_local < 0 || B_total <= 0)
        return fail(TRLDA_ERR_ARG, "bad online_eb arguments");
    if (m->eb.active)
        return fail(TRLDA_ERR_ARG, "an empirical-Bayes step is already on its way: finish it first");
    if (!update_alpha && !update_eta)
        return TRLDA_OK;
    const int K = m->K;
    const size_t KV = (size_t)K * m->V;
    constexpr int T = 256;
    const int chunks = (std::max(B_local, 1) + trlda::kEbDocsPerBlock - 1) / trlda::kEbDocsPerBlock;
    const int G = (int)std::max<size_t>(1, std::min<size_t>((KV + 4 * T - 1) / (4 * T), 2048));
    // reduce_out: [gamma chunks | gamma sum (K)] [lambda blocks (G)]
    const size_t off_sum = (size_t)chunks * K, off_lam = off_sum + (size_t)K;
    rc = grow(&m->reduce_out, &m->cap_reduce, off_lam + (size_t)G);
    if (rc)
        return rc;
    const size_t hn = (size_t)K + (size_t)G + (size_t)K;
    if (m->eb.cap < hn) {
        if (m->eb.host)
            (void)hipHostFree(m->eb.host);
        m->eb.host = nullptr;
        m->eb.cap = 0;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&m->eb.host), hn * sizeof(double), hipHostMallocDefault));
        m->eb.cap = hn;
    }
    if (!m->eb.event)
        HIP_TRY(hipEventCreateWithFlags(&m->eb.event, hipEventDisableTiming));
    double *const host = m->eb.host;
    if (update_alpha) {
        double *sum = m->reduce_out + off_sum;
        if (B_local > 0) {
            if ((size_t)B_local * K > m->cap_gamma || !m->gamma)
                return fail(TRLDA_ERR_ARG, "no gamma of that size is resident in the model");
            const size_t lds = ((size_t)K + T / trlda::kWave + 1) * sizeof(double);
            auto kern = trlda::eb_gamma_kernel<T>;
            if ((rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds)))
                return rc;
            hipLaunchKernelGGL(kern, dim3(chunks), dim3(T), lds, m->stream, K, B_local, m->gamma,
                               m->reduce_out);
            HIP_TRY(hipGetLastError());
            rc = combine_rowsums(m, m->reduce_out, chunks, nullptr, sum);
        } else {
            HIP_TRY(hipMemsetAsync(sum, 0, (size_t)K * sizeof(double), m->stream));
        }
        if (!rc && rccl_comm)
            rc = allreduce_f64(m, rccl_comm, sum, (size_t)K);        // onlinelda.cpp:128 across ranks
        if (rc)
            return rc;
        HIP_TRY(hipMemcpyAsync(host, sum, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, m->stream));
        m->d2h_bytes += (int64_t)K * sizeof(double);
    }
    if (update_eta) {
        hipLaunchKernelGGL(trlda::eb_lambda_kernel<T>, dim3(G), dim3(T), 0, m->stream, KV, m->lambda,
                           m->reduce_out + off_lam);
        HIP_TRY(hipGetLastError());
        if (rowsums_carried(m)) {
            rc = resolve_carry(m);
        } else if (stream_available(m)) {
            rc = rowsums_from_scratch(m);
            if (!rc && !m->lambda_exposed)
                m->rs_valid = true;
        } else {
            int GR = std::min(kMaxRowsumBlocks - 1, std::max(1, m->V / 32));
            const int wpb = (m->V + GR - 1) / GR;
            GR = (m->V + wpb - 1) / wpb;
            hipLaunchKernelGGL(trlda::rowsum_partial_kernel<kDenseThreads>, dim3(GR), dim3(kDenseThreads), 0,
                               m->stream, K, m->V, wpb, m->lambda, m->partial);
            HIP_TRY(hipGetLastError());
            rc = combine_rowsums(m, m->partial, GR, nullptr, m->rs_full);
        }
        if (rc)
            return rc;
        HIP_TRY(hipMemcpyAsync(host + K, m->reduce_out + off_lam, (size_t)G * sizeof(double),
                               hipMemcpyDeviceToHost, m->stream));
        HIP_TRY(hipMemcpyAsync(host + K + G, m->rs_full, (size_t)K * sizeof(double), hipMemcpyDeviceToHost,
                               m->stream));
        m->d2h_bytes += (int64_t)((size_t)G + K) * sizeof(double);
    }
    HIP_TRY(hipEventRecord(m->eb.event, m->stream));
    m->eb.active = true;
    m->eb.alpha = update_alpha != 0;
    m->eb.eta = update_eta != 0;
    m->eb.G = G;
    m->eb.B_total = B_total;
    return TRLDA_OK;
}

int trlda_model_online_eb_pending(const trlda_model *m) { return m && m->eb.active ? 1 : 0; }

int trlda_model_online_eb_finish(trlda_model *m, double rho, double min_alpha, double min_eta,
                                 double *alpha_host, double *eta)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!alpha_host || !eta)
        return fail(TRLDA_ERR_ARG, "bad online_eb arguments");
    if (!m->eb.active)
        return TRLDA_OK;
    m->eb.active = false;
    const int K = m->K, G = m->eb.G;
    HIP_TRY(hipEventSynchronize(m->eb.event));                       // the one trip
    if ((rc = check_split_exchange(m)))                              // (the update before the sums)
        return rc;
    const double *host = m->eb.host;
    if (m->eb.alpha) {
        std::vector<double> next((size_t)K);
        rc = trlda_eb_online_alpha_step(K, alpha_host, host, (double)m->eb.B_total, rho, min_alpha,
                                        next.data());
        if (rc)
            return rc;
        std::memcpy(alpha_host, next.data(), (size_t)K * sizeof(double));
        // (staged through the pinned buffer's first K doubles: the caller's array may be gone
        // before the copy runs)
        std::memcpy(m->eb.host, alpha_host, (size_t)K * sizeof(double));
        HIP_TRY(hipMemcpyAsync(m->alpha, m->eb.host, (size_t)K * sizeof(double), hipMemcpyHostToDevice,
                               m->stream));
    }
    if (m->eb.eta) {
        double total = 0.0;
        for (int g = 0; g < G; ++g)
            total += host[(size_t)K + (size_t)g];
        *eta = trlda_eb_online_eta_step(*eta, total, host + K + G, K, m->V, rho, min_eta);
    }
    return TRLDA_OK;
}

int trlda_model_online_eb(trlda_model *m, void *rccl_comm, int B_local, int B_total, double rho,
                          int update_alpha, int update_eta, double min_alpha, double min_eta,
                          double *alpha_host, double *eta)
{
    if (!alpha_host || !eta)
        return fail(TRLDA_ERR_ARG, "bad online_eb arguments");
    const int rc = trlda_model_online_eb_begin(m, rccl_comm, B_local, B_total, update_alpha, update_eta);
    return rc ? rc : trlda_model_online_eb_finish(m, rho, min_alpha, min_eta, alpha_host, eta);
}

int trlda_model_adaptive_stats(trlda_model *m, double eta, double scale, double tau,
                               double *sq_norm_update, double *sq_norm_gradient)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!sq_norm_update || !sq_norm_gradient)
        return fail(TRLDA_ERR_ARG, "bad adaptive_stats arguments");
    if (!m->keep_sstats || !m->sstats || !m->lambda_prime)
        return fail(TRLDA_ERR_ARG, "adaptive_stats needs an update made with keep_sstats on");
    return trlda_model_adaptive_stats_dev(m, m->sstats, m->lambda_prime, eta, scale, tau, sq_norm_update,
                                          sq_norm_gradient);
}

int trlda_model_adaptive_stats_dev(trlda_model *m, const double *sstats_dev,
                                   const double *lambda_prime_dev, double eta, double scale, double tau,
                                   double *sq_norm_update, double *sq_norm_gradient)
{
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!sq_norm_update || !sq_norm_gradient || !sstats_dev || !lambda_prime_dev)
        return fail(TRLDA_ERR_ARG, "bad adaptive_stats arguments");
    const size_t KV = (size_t)m->K * m->V;
    if (!m->ada_gradient) {
        rc = dev_alloc(&m->ada_gradient, KV);                // mAdaGradient starts at zero
        if (rc)
            return rc;
        HIP_TRY(hipMemsetAsync(m->ada_gradient, 0, KV * sizeof(double), m->stream));
    }
    constexpr int T = 256;
    const int G = (int)std::max<size_t>(1, std::min<size_t>((KV + 4 * T - 1) / (4 * T), 2048));
    rc = grow(&m->reduce_out, &m->cap_reduce, 2 * (size_t)G);
    if (rc)
        return rc;
    hipLaunchKernelGGL(trlda::adaptive_kernel<T>, dim3(G), dim3(T), 0, m->stream, KV, eta, scale, tau,
                       sstats_dev, lambda_prime_dev, m->ada_gradient, m->reduce_out);
    HIP_TRY(hipGetLastError());
    std::vector<double> blocks(2 * (size_t)G);
    HIP_TRY(hipMemcpyAsync(blocks.data(), m->reduce_out, blocks.size() * sizeof(double),
                           hipMemcpyDeviceToHost, m->stream));
    m->d2h_bytes += (int64_t)(blocks.size() * sizeof(double));
    if (int rc_sync = sync_model(m))
        return rc_sync;
    double u2 = 0.0, g2 = 0.0;
    for (int g = 0; g < G; ++g) {
        u2 += blocks[2 * (size_t)g];
        g2 += blocks[2 * (size_t)g + 1];
    }
    *sq_norm_update = u2;
    *sq_norm_gradient = g2;
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
    if (!rc) rc = batch_wait(b);                     // (tr_init_device is not an entry point: it does not wait)
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

// Experiment (tools/graph_probe.py): one OnlineLDA::updateParameters call (src/onlinelda.cpp:
// 89-101: its 1 + 2 x max_iter_tr .. launches) recorded ONCE into a HIP graph and replayed, against
// the same call enqueued launch by launch.  The replay repeats the recorded call exactly -- the same
// gamma0 windows, the same batch -- so it is a measurement of what a graph could save, not a product
// path.  usec_out[0] = per call, direct; [1] = per replayed graph launch.
extern "C" int trlda_debug_graph_update(trlda_model *m, const trlda_batch *b, int num_documents, double eta,
                                        int max_iter_tr, int max_iter_inference, int reps, double *usec_out)
{
    // (the batches' indices are built on worker threads: trlda_batch_create)
    if (int rc_built = batch_wait(b))
        return rc_built;
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!b || !usec_out || reps < 1)
        return fail(TRLDA_ERR_ARG, "bad graph probe arguments");
    int count = 0;
    double rho = 0.;
    auto call = [&]() {
        return trlda_model_online_update(m, b, num_documents, eta, max_iter_tr, max_iter_inference, .7, 100.,
                                         0.01, 1, 1, 1e-3, &count, &rho, nullptr);
    };
    for (int i = 0; i < 3 && !rc; ++i)                            // allocations, attributes, caches
        rc = call();
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(m->stream));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t0 = now();
    for (int i = 0; i < reps && !rc; ++i)
        rc = call();
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(m->stream));
    usec_out[0] = std::chrono::duration<double, std::micro>(now() - t0).count() / reps;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    HIP_TRY(hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal));
    rc = call();
    hipError_t e = hipStreamEndCapture(m->stream, &graph);
    if (rc || e != hipSuccess) {
        (void)hipGetLastError();
        return fail(TRLDA_ERR_HIP, std::string("capture failed: ") + (rc ? trlda_last_error() : hipGetErrorString(e)));
    }
    HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i)
        HIP_TRY(hipGraphLaunch(exec, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    t0 = now();
    for (int i = 0; i < reps; ++i)
        HIP_TRY(hipGraphLaunch(exec, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    usec_out[1] = std::chrono::duration<double, std::micro>(now() - t0).count() / reps;
    (void)hipGraphExecDestroy(exec);
    (void)hipGraphDestroy(graph);
    return TRLDA_OK;
}

// Experiment, second form (round 5): what a PRODUCT path would have to do -- every call has other
// arguments (batch, grids, gamma0 windows, counters), so every call is captured anew and the
// instantiated graph is brought up to date with hipGraphExecUpdate (re-instantiated when that is
// refused), then launched.  Two batches in turn.  usec_out: [0] per call enqueued launch by launch,
// [1] per call through capture + update + launch, [2] host time of capture + update per call,
// [3] share of calls whose update was refused.
extern "C" int trlda_debug_graph_update2(trlda_model *m, const trlda_batch *b0, const trlda_batch *b1,
                                         int num_documents, double eta, int max_iter_tr, int max_iter_inference,
                                         int reps, double *usec_out)
{
    // (the batches' indices are built on worker threads: trlda_batch_create)
    if (int rc_built = batch_wait(b0))
        return rc_built;
    if (int rc_built = batch_wait(b1))
        return rc_built;
    int rc = check_model(m);
    if (rc)
        return rc;
    if (!b0 || !b1 || !usec_out || reps < 1)
        return fail(TRLDA_ERR_ARG, "bad graph probe arguments");
    int count = 0;
    double rho = 0.;
    auto call = [&](int i) {
        return trlda_model_online_update(m, (i & 1) ? b1 : b0, num_documents, eta, max_iter_tr, max_iter_inference,
                                         .7, 100., 0.01, 1, 1, 1e-3, &count, &rho, nullptr);
    };
    for (int i = 0; i < 4 && !rc; ++i)
        rc = call(i);
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(m->stream));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t0 = now();
    for (int i = 0; i < reps && !rc; ++i)
        rc = call(i);
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(m->stream));
    usec_out[0] = std::chrono::duration<double, std::micro>(now() - t0).count() / reps;
    hipGraphExec_t exec = nullptr;
    double host_us = 0.;
    int refused = 0;
    auto graph_call = [&](int i) -> int {
        auto h0 = now();
        hipGraph_t graph = nullptr;
        HIP_TRY(hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal));
        int r = call(i);
        hipError_t e = hipStreamEndCapture(m->stream, &graph);
        if (r || e != hipSuccess) {
            (void)hipGetLastError();
            return fail(TRLDA_ERR_HIP, std::string("capture failed: ") + (r ? trlda_last_error() : hipGetErrorString(e)));
        }
        bool fresh = exec == nullptr;
        if (exec) {
            hipGraphNode_t bad = nullptr;
            hipGraphExecUpdateResult res;
            if (hipGraphExecUpdate(exec, graph, &bad, &res) != hipSuccess) {
                (void)hipGetLastError();
                (void)hipGraphExecDestroy(exec);
                exec = nullptr;
                ++refused;
            }
        }
        if (!exec)
            HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        (void)fresh;
        host_us += std::chrono::duration<double, std::micro>(now() - h0).count();
        HIP_TRY(hipGraphLaunch(exec, m->stream));
        (void)hipGraphDestroy(graph);
        return TRLDA_OK;
    };
    for (int i = 0; i < 4 && !rc; ++i)
        rc = graph_call(i);
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(m->stream));
    host_us = 0.;
    refused = 0;
    t0 = now();
    for (int i = 0; i < reps && !rc; ++i)
        rc = graph_call(i);
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(m->stream));
    usec_out[1] = std::chrono::duration<double, std::micro>(now() - t0).count() / reps;
    usec_out[2] = host_us / reps;
    usec_out[3] = (double)refused / reps;
    if (exec)
        (void)hipGraphExecDestroy(exec);
    return TRLDA_OK;
}

// diagnostics: the s_memtime stamps of the model's last merged launch (3 x 1024 values)
extern "C" int trlda_debug_deferred_stamps(trlda_model *m, unsigned long long *host)
{
    if (!m || !host || !m->deferred_stamps)
        return TRLDA_ERR_ARG;
    HIP_TRY(hipStreamSynchronize(m->stream));
    HIP_TRY(hipMemcpy(host, m->deferred_stamps, 3 * 3072 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(m->deferred_stamps, 0, 3 * 3072 * sizeof(unsigned long long)));
    return TRLDA_OK;
}

extern "C" int trlda_debug_merged_stamps(trlda_model *m, unsigned long long *host)
{
    if (!m || !host || !m->merged_stamps)
        return TRLDA_ERR_ARG;
    HIP_TRY(hipStreamSynchronize(m->stream));
    HIP_TRY(hipMemcpy(host, m->merged_stamps, 3 * 1024 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return TRLDA_OK;
}

// diagnostics: copy out one of the model's intermediate buffers of the last E-step
//   0  exp(psi(lambda)) / exp E[log beta] as the last call's kernels read it (K x V)
//   1  the documents' exp E[log theta] rows (count values)
//   2  cnt / phinorm per entry in word order (count values)
//   3  the sstats buffer of the host entry points (K x V)
extern "C" int trlda_debug_peek(trlda_model *m, int which, double *host, size_t count)
{
    if (!m || !host)
        return TRLDA_ERR_ARG;
    const double *src = which == 0 ? m->eeb_cur : which == 1 ? m->epg : which == 2 ? m->tw_word
                        : which == 3 ? m->sstats : nullptr;
    if (!src)
        return TRLDA_ERR_ARG;
    HIP_TRY(hipStreamSynchronize(m->stream));
    HIP_TRY(hipMemcpy(host, src, count * sizeof(double), hipMemcpyDeviceToHost));
    return TRLDA_OK;
}

extern "C" void trlda_debug_call_times(double *out16)
{
    for (int i = 0; i < 8; ++i) {
        out16[i] = g_call_us[i];
        out16[8 + i] = i < 7 ? 1e-3 * (double)g_build_ns[i].load() : (double)g_build_ns[i].load();
    }
}

extern "C" void trlda_debug_ingest_counters(long long *out8)
{
    for (int i = 0; i < 8; ++i)
        out8[i] = g_ingest[i].load();
}

// tests: the first `bytes` bytes of a batch's device allocation (its index as the kernels see it), after
// its upload; state_out (may be NULL): the build ticket's state when the call arrived (trlda_batch::kQueued:
// this call took the build over; kBuilding: a worker had it; kBuilt: it was ready)
extern "C" int trlda_debug_batch_blob(const trlda_batch *b, void *host, size_t bytes, int *state_out)
{
    if (!b || !host)
        return TRLDA_ERR_ARG;
    if (state_out)
        *state_out = b->ticket->state.load(std::memory_order_acquire);
    if (int rc = batch_wait(b))
        return rc;
    if (bytes > b->blob_bytes)
        return fail(TRLDA_ERR_ARG, "more bytes than the batch's allocation holds");
    HIP_TRY(hipSetDevice(b->device));
    if (b->ready)
        HIP_TRY(hipEventSynchronize(b->ready));
    HIP_TRY(hipMemcpy(host, b->blob, bytes, hipMemcpyDeviceToHost));
    return TRLDA_OK;
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
    DevTemp guard;
    guard.p = d;
    const size_t bytes = (size_t)n * sizeof(double);
    HIP_TRY(hipMemcpy(d, x, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(trlda::digamma_table_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, n,
                       c, d, d + n, d + 2 * (size_t)n, d + 3 * (size_t)n, d + 4 * (size_t)n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(psi, d + n, bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(epsi, d + 2 * (size_t)n, bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(epsi_lean, d + 3 * (size_t)n, bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(eminus, d + 4 * (size_t)n, bytes, hipMemcpyDeviceToHost));
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
    DevTemp guard;
    guard.p = d;
    HIP_TRY(hipMemcpy(d, in, 64 * 16 * sizeof(double), hipMemcpyHostToDevice));
    double *o = d + 64 * 16;
    hipLaunchKernelGGL(trlda::debug_fold16_kernel, dim3(1), dim3(64), 0, nullptr, d, o, o + 64,
                       o + 128);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out16, o, 64 * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out4, o + 64, 64 * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out2, o + 128, 64 * sizeof(double), hipMemcpyDeviceToHost));
    return TRLDA_OK;
}

// (the text parser lives in text_docs.cpp, the generator in host_rng.cpp, the empirical-Bayes
// Newton steps in eb_steps.cpp: host-only translation units, also built under the sanitizers)

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
    if (m->lanes_live)
        (void)lanes_join(m);
    lane_spans_collect(m);
    m->lane_span_us = 0.0;
    m->lane_span_count = 0;
    return TRLDA_OK;
}

// E-steps that went through the lanes while timing was on: the mean duration of their launches
// (a lane's launches run back to back; two lanes' overlap)
int trlda_model_get_lane_timing(trlda_model *m, double *usec_sum, int64_t *launches)
{
    if (!m || !usec_sum || !launches)
        return fail(TRLDA_ERR_ARG, "bad timing query");
    int rc = check_model(m);                         // (joins: the spans end)
    if (rc)
        return rc;
    lane_spans_collect(m);
    *usec_sum = m->lane_span_us;
    *launches = m->lane_span_count;
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
