/*
 * trlda_hip.h -- C ABI of libtrlda_hip.so, the MI355X (gfx950) implementation of
 * trlda's per-document variational E-step and the lambda M-step around it.
 *
 * This is the drop-in boundary: plain pointers and sizes, status-code returns
 * (0 = ok, negative = error, message via trlda_last_error()), no exceptions and
 * no torch / Eigen / CPython types cross it.  Each entry point names the
 * reference interface it replaces (paths relative to the reference repository).
 *
 * Conventions (identical to the reference's in-memory layout):
 *   - every matrix is column-major fp64 (Eigen default; python/src/pyutils.cpp:22-26):
 *       lambda, sstats : K x V, element (k, w) at [k + K*w]
 *       gamma          : K x B, element (k, d) at [k + K*d]
 *   - a batch of documents is CSR int32: indptr[B+1], ids[nnz], cnts[nnz]; this is
 *     the flat form of LDA::Documents = vector<vector<pair<int,int>>>
 *     (include/lda.h:21-23) that PyList_ToDocuments builds
 *     (python/src/ldainterface.cpp:152-190).  Duplicate ids inside a document and
 *     zero counts are legal.  Word ids are validated (0 <= id < V); the reference
 *     has undefined behaviour there.
 *   - "host" pointers are ordinary memory; "dev" pointers are HIP device
 *     addresses on the model's device (hipMalloc'ed by trlda_dev_alloc or by
 *     anyone else, e.g. a torch tensor's data_ptr()).
 *   - functions taking a trlda_model enqueue work on the model's stream and
 *     return without synchronising unless they copy to host memory.  That stream is a
 *     non-blocking stream of the model's own until trlda_model_set_stream hands it
 *     another one: it does NOT order itself against the legacy null stream, so device
 *     buffers a caller fills or reads on a stream of its own must either be complete
 *     (trlda_model_synchronize / trlda_dev_synchronize) or that stream be the model's.
 *
 * There is NO CPU fallback behind this header: with no usable GPU every compute
 * entry point returns TRLDA_ERR_NO_DEVICE.
 */
#ifndef TRLDA_HIP_H
#define TRLDA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRLDA_OK               0
#define TRLDA_ERR_ARG         -1  /* bad argument (null pointer, negative size ...)       */
#define TRLDA_ERR_SHAPE       -2  /* TRLDA::Exception("... wrong dimensionality.")         */
#define TRLDA_ERR_WORD_ID     -3  /* word id outside [0, V)                                */
#define TRLDA_ERR_NO_DEVICE   -4  /* no HIP device / HIP runtime error at init             */
#define TRLDA_ERR_HIP         -5  /* a HIP call failed; see trlda_last_error()             */
#define TRLDA_ERR_VALUE       -6  /* TRLDA::Exception("... should not be negative.")       */

/* how sufficient statistics are accumulated (trlda_model_set_sstats_mode) */
#define TRLDA_SSTATS_SEGMENTED 0  /* per-word ordered sums via the batch's word-major index:
                                     bitwise reproducible, same addition order as the
                                     reference's serial loop (lda.cpp:207-213)              */
#define TRLDA_SSTATS_ATOMIC    1  /* fp64 global atomics from the document kernel           */

typedef struct trlda_model trlda_model;   /* device-resident lambda, alpha, workspaces */
typedef struct trlda_batch trlda_batch;   /* device-resident CSR batch + word-major index */

/* ---- library ---------------------------------------------------------- */

const char *trlda_last_error(void);       /* thread-local, never NULL */
int trlda_version(void);
int trlda_device_count(void);             /* number of HIP devices, 0 if none */

/* ---- host-side RNG: bit-compatible with the reference ------------------ */

/* trlda.seed(): srand(seed).  python/src/module.cpp:332-342 */
void trlda_seed(unsigned int seed);

/* The state of that stream (33 words), so that the ranks of a data-parallel job can adopt
 * rank 0's: the reference has one process and one stream (src/lda.cpp:71, :135); replicas that
 * draw "the same" gamma0 must share it. */
void trlda_rng_get_state(uint32_t *state33);
void trlda_rng_set_state(const uint32_t *state33);

/* sampleGamma(m, n, k): -sum_{i<k} log|U_i|, U = -1 + 2*rand()/RAND_MAX, k passes over
 * an m x n column-major matrix, libc rand() consumed in exactly the reference's order.
 * src/utils.cpp:224-231, Eigen/src/Core/MathFunctions.h:439-446.  Host memory. */
void trlda_sample_gamma(int m, int n, int k, double *out);

/* sampleGamma(m, n, 100) / 100.: the lambda init of LDA::LDA (src/lda.cpp:71) and the
 * default gamma init of LDA::updateVariables (src/lda.cpp:135).  Host memory. */
void trlda_sample_gamma_init(int m, int n, double *out);

/* ---- one-shot, host pointers ------------------------------------------- */

/*
 * LDA::updateVariables(documents, latents, parameters) with inferenceMethod == VI:
 * include/lda.h:109-112 -> src/lda.cpp:142-156 -> LDA::updateVariablesVI,
 * src/lda.cpp:160-220.  Called by python/src/ldainterface.cpp:367-370
 * (update_variables / do_e_step).
 *   gamma     in: initial gamma (K x B);  out: gamma after inference
 *   sstats    out: K x V sufficient statistics (already multiplied by exp E[log beta])
 *   iters_out optional (may be NULL): fixed-point iterations executed per document
 *   device    HIP device ordinal
 * Uploads, runs the HIP kernels, downloads, synchronises.
 */
int trlda_estep(int K, int V, int B,
                const int32_t *indptr, const int32_t *ids, const int32_t *cnts,
                const double *lambda, const double *alpha,
                double *gamma, double *sstats,
                int max_iter, double threshold, int32_t *iters_out, int device);

/* lambda = (1-rho) lambda' + rho (eta + scale * sstats): src/onlinelda.cpp:99-100 and
 * :108-109 (scale = D / B); src/batchlda.cpp:60 is the rho = 1, scale = 1 case. */
int trlda_mstep_blend(int K, int V, double rho, double eta, double scale,
                      const double *lambda_prime, const double *sstats,
                      double *lambda_out, int device);

/* Trust-region initial step, src/onlinelda.cpp:79-86:
 * lambda[:, w] = (1-rho) lambda'[:, w] + rho (eta + D/B/K * wordcounts[w]). */
int trlda_tr_init(int K, int V, int B, int num_documents, double rho, double eta,
                  const int32_t *indptr, const int32_t *ids, const int32_t *cnts,
                  const double *lambda_prime, double *lambda_out, int device);

/* ---- text corpora (host side) ---------------------------------------------------------
 *
 * The reference's corpus files (python/utils/load_documents.py:6-69): one document per line,
 * "<n> id:cnt id:cnt ...", the leading <n> ignored like line.split()[1:] does.  The file is
 * mapped and parsed by the library's host threads straight into CSR -- the form
 * trlda_batch_create takes: batch b of documents [d0, d1) is
 * indptr = offsets[d0 .. d1] - offsets[d0], ids + offsets[d0], cnts + offsets[d0].
 * Tokens that are not [+-]digits:[+-]digits within int32, and files with bare carriage
 * returns, fail with TRLDA_ERR_VALUE / TRLDA_ERR_ARG and a line number in trlda_last_error(). */
typedef struct trlda_docs trlda_docs;
int trlda_docs_from_text(const char *path, trlda_docs **out);
/* the same for text already in memory (a window of a file that does not fit: the Python loader
 * reads bounded chunks cut at line ends); `bytes` of `text`, the last line may lack its newline */
int trlda_docs_from_buffer(const char *text, size_t bytes, trlda_docs **out);
int64_t trlda_docs_num_docs(const trlda_docs *docs);
int64_t trlda_docs_nnz(const trlda_docs *docs);
const int64_t *trlda_docs_offsets(const trlda_docs *docs);   /* num_docs + 1 */
const int32_t *trlda_docs_ids(const trlda_docs *docs);       /* nnz */
const int32_t *trlda_docs_cnts(const trlda_docs *docs);      /* nnz */
int trlda_docs_destroy(trlda_docs *docs);

/* ---- device memory helpers (so a host program needs no HIP headers) ---- */

int trlda_dev_alloc(int device, size_t bytes, void **dev_out);
int trlda_dev_free(int device, void *dev);
int trlda_dev_upload(int device, void *dev_dst, const void *host_src, size_t bytes);
int trlda_dev_download(int device, void *host_dst, const void *dev_src, size_t bytes);
int trlda_dev_synchronize(int device);

/* ---- batches ------------------------------------------------------------ */

/* Validates (monotone indptr, 0 <= id < V), uploads the CSR arrays and builds the
 * word-major index (stable counting sort by word id) used by the segmented
 * sufficient-statistics kernel.  Replaces PyList_ToDocuments' deep copy
 * (python/src/ldainterface.cpp:152-190).
 *
 * On the caller's thread: the validation -- every error of the arguments is this call's -- and one
 * copy of the three arrays into pinned memory (they are the caller's again on return): ~8-12 us per
 * 200 documents instead of ~75.  The index (csrc/batch_index.cpp: host work only) is built on the
 * library's worker threads (TRLDA_INDEX_THREADS, default 4; 0: on this thread, as before round 6), up
 * to two dozen batches at a time.  The workers make NO HIP call: a finished index is uploaded --
 * an allocation, one copy on the library's upload stream (a 16-workgroup kernel that reads the pinned
 * buffer over PCIe; hipMemcpyAsync's call blocked for milliseconds now and then: TRLDA_UPLOAD_COPY=memcpy
 * for A/B), two events: ~10 us -- by a caller's thread,
 * whichever comes first: the next trlda_batch_create (which uploads what was finished since the last
 * one), the E-step the batch is announced to (`next`, `upcoming[]` below), or the batch's first user.
 * Every entry point that is handed the batch waits for the index -- or, when no worker has started on
 * it yet, does the work itself, so that "create, use at once" costs what it did.  A batch destroyed
 * before anybody used it is never indexed.  An index is there ~170 us after its trlda_batch_create
 * (a queue, 40 us in which a caller that uses its batch at once is left alone with it, ~80 us of work):
 * a stream makes its batches eight steps ahead of their use (bench.py, value_end_to_end).  A failed
 * upload (out of device memory) fails the batch's first use, with the build's status and message.
 * An announced batch whose index is not there yet counts as not announced.
 * (TRLDA_INDEX_UPLOAD=worker: the workers enqueue the uploads too, one at a time -- the first form of
 * round 6, kept for A/B: four threads in hipMemcpyAsync / hipEventRecord beside the caller's launches
 * spent 32-175 us per build in the runtime's locks.) */
int trlda_batch_create(trlda_batch **out, int device, int V, int B,
                       const int32_t *indptr, const int32_t *ids, const int32_t *cnts);
int trlda_batch_destroy(trlda_batch *batch);
int trlda_batch_num_docs(const trlda_batch *batch);
int64_t trlda_batch_nnz(const trlda_batch *batch);
int trlda_batch_max_doc_len(const trlda_batch *batch);
/* The number of entries above which a word's list of (document, weight) pairs is walked by a
 * whole workgroup instead of one wavefront in the statistics kernels: 16 << i, the smallest that
 * leaves at most 512 such words in this batch (csrc/estep_kernels.h, kLongWord); and how many
 * words that is. */
int trlda_batch_long_word_len(const trlda_batch *batch);
int trlda_batch_num_long_words(const trlda_batch *batch);
/* Lists of more than 256 .. 1024 entries (a word present in most documents of a large batch; the
 * length is chosen per batch so that there are about a thousand tasks) are cut into segments of at
 * most that many: a segment is a task for a whole workgroup of the statistics kernels, and
 * the workgroup that finishes a word's last segment adds the segments' sums up in segment order
 * (csrc/estep_kernels.h, VeryLongArgs) -- the kernel no longer lasts as long as its longest list
 * (K = 200, 12 500 documents: one list of 12 500 entries kept a workgroup busy for ~100 us).  The
 * number of such words in the batch; trlda_model_set_split_lists(model, 0) (or TRLDA_SPLIT_LISTS=0)
 * keeps every list with one workgroup, for comparisons: same results to rounding. */
int trlda_batch_num_very_long_words(const trlda_batch *batch);

/* ---- device-resident model ---------------------------------------------- */

/* State of TRLDA::LDA (include/lda.h:196-199: mAlpha, mEta, mLambda) kept in HBM.
 * lambda is NOT initialised here: the host mirror draws it with
 * trlda_sample_gamma_init (as src/lda.cpp:71 does) and uploads it. */
int trlda_model_create(trlda_model **out, int device, int K, int V);
int trlda_model_destroy(trlda_model *model);
/* Kernels of this model run on the host's stream from now on (default: a non-blocking stream of
 * the model's own).  The stream must outlive every trlda_batch that was used on it: a batch
 * records the event that guards its memory once, when it is destroyed, on the stream that used
 * it last. */
int trlda_model_set_stream(trlda_model *model, void *hip_stream /* hipStream_t */);
int trlda_model_set_sstats_mode(trlda_model *model, int mode);
/* exp E[log beta] (src/lda.cpp:173) is computed for the words that occur in the batch (the
 * only columns the path reads); dense = 1 fills all V columns like the reference.  The
 * outputs (gamma, sstats) are identical either way. */
int trlda_model_set_dense_preamble(trlda_model *model, int dense);
/* threads per document workgroup in the E-step kernel: 0 = auto, else 64..1024 (x64) */
int trlda_model_set_doc_threads(trlda_model *model, int threads);
/* which document kernels the E-step may use (all give the same results; tests and tuning):
 * AUTO = dual-orientation register kernel for K <= 128 and at most 192 words, else the
 * single-orientation register kernel up to K = 512, else the general (LDS / streaming)
 * kernel; GENERAL = never the single-orientation kernel; WIDE = the single-orientation
 * kernel for every document (K <= 512). */
#define TRLDA_DOCS_AUTO 0
#define TRLDA_DOCS_GENERAL 1
#define TRLDA_DOCS_WIDE 2
/* K <= 32: a WAVE per document of at most 128 words, eight documents per workgroup, no
 * LDS or barrier inside the fixed point (csrc/estep_kernels.h, estep_docs_small_body; round 6); longer
 * documents keep their workgroups -- one each, or a segment each where they are split -- in front of them
 * in the same launch (not where more than half of the documents are longer).  A
 * throughput form -- a document takes 46-55 us on one wave against 30 on eight, but a CU holds eight
 * of them: the default for batches with more short documents than the device has CUs (K = 10, 6400
 * documents: 38.5 against 9.7 M docs/s); _SMALL asks for it at any batch size, _REG for the
 * workgroup-per-document body of K <= 128.  Both leave everything around the body (fused preamble,
 * merged / deferred statistics, lanes) alone. */
#define TRLDA_DOCS_SMALL 3
#define TRLDA_DOCS_REG 4
int trlda_model_set_doc_kernel(trlda_model *model, int kind);
/* name (as a profiler lists it, without template arguments) of the document kernel that
 * took most documents of the model's last E-step; "" before the first */
const char *trlda_model_last_doc_kernel(const trlda_model *model);
/* Small tables whose whole batch fits the register-resident document kernel run the row
 * sums and exp(psi(lambda)) in one launch (preamble_fused_kernel) and apply the topic factors
 * exp(-psiSum) in the document kernel; results agree with the two-kernel preamble to a few
 * ulp.  split = 1 always uses the two kernels.  last_preamble_fused: what the last E-step did. */
int trlda_model_set_split_preamble(trlda_model *model, int split);
int trlda_model_last_preamble_fused(const trlda_model *model);
int trlda_model_synchronize(trlda_model *model);
/* K <= 128: a document of 193 .. 1024 words is iterated by ceil(n / 128) workgroups of the
 * document launch, one segment of its words each, which exchange K partial sums per iteration
 * through HBM (src/lda.cpp:189-193 is a sum over the document's words; every segment ends up with
 * bitwise the same gamma) -- a launch lasts as long as its slowest document, and one CU takes
 * ~270 us for 600 words where five take ~50.  enabled = 0: one workgroup per document whatever
 * its length (default 1).  Should a segment ever give up waiting for its peers, the next
 * trlda_model_synchronize / host-copying call fails with TRLDA_ERR_HIP. */
int trlda_model_set_split_docs(trlda_model *model, int enabled);
/* workgroups the model's last document launch used beyond one per document (0: nothing split;
 * a batch whose launch would not end sooner with split documents stays unsplit) */
int trlda_model_last_split_workgroups(const trlda_model *model);
/* Small tables, batches of at most 224 document workgroups (K <= 128 even, a mini-batch of 200):
 * the statistics (src/lda.cpp:207-217) with the M-step (src/onlinelda.cpp:99-100,
 * src/batchlda.cpp:60), the row sums the next E-step needs (src/lda.cpp:172) and its
 * exp(psi(lambda)) can be extra WORKGROUPS OF THE DOCUMENT LAUNCH that wait for a documents-done
 * counter, instead of a kernel of their own behind it: one launch per trust-region iteration
 * (csrc/estep_merged.h).  Same sums in the same order: bitwise the statistics of the stand-alone
 * kernel.  level 1 (default): where the statistics carry an M-step (the update entry points);
 * 2: plain E-steps too (the two forms take the same time there); 0, or TRLDA_MERGED=0 in the
 * environment: always the kernel of its own.  last_merged: what the model's last E-step did. */
int trlda_model_set_merged_launch(trlda_model *model, int level);
int trlda_model_set_split_lists(trlda_model *model, int enabled);
int trlda_model_last_merged(const trlda_model *model);

int trlda_model_set_lambda(trlda_model *model, const double *host_lambda /* K x V */);
int trlda_model_get_lambda(trlda_model *model, double *host_lambda /* K x V */);
int trlda_model_set_alpha(trlda_model *model, const double *host_alpha /* K */);
void *trlda_model_lambda_dev(trlda_model *model);   /* K x V fp64, device */
/* sufficient statistics of the last E-step run by trlda_model_online_update /
 * trlda_model_batch_update / trlda_model_estep_host (the model's own workspace): what the
 * adaptive learning rate of src/onlinelda.cpp:167-175 needs (lambdaHat = eta + D/B sstats). */
int trlda_model_get_sstats(trlda_model *model, double *host_sstats /* K x V */);

/*
 * Device-pointer E-step on the model's current lambda: src/lda.cpp:160-220.
 *   gamma_dev  K x B in/out;  sstats_dev K x V out;  iters_dev B int32 or NULL.
 * Launch sequence: row sums -> exp E[log beta] -> per-document
 * fixed point -> per-word sufficient statistics (x exp E[log beta]).
 */
int trlda_model_estep(trlda_model *model, const trlda_batch *batch,
                      double *gamma_dev, double *sstats_dev,
                      int max_iter, double threshold, int32_t *iters_dev);

/* Same, with the initial gamma read from a separate device array (K x B) that is left
 * untouched -- `ArrayXXd gamma = initialGamma` of src/lda.cpp:168 without a device copy. */
int trlda_model_estep_io(trlda_model *model, const trlda_batch *batch,
                         const double *gamma0_dev, double *gamma_dev, double *sstats_dev,
                         int max_iter, double threshold, int32_t *iters_dev);

/* The same, with the batch of the NEXT E-step announced.  Consecutive E-steps on an unchanged
 * lambda are independent of each other, and the document kernel of a 200-document batch leaves
 * 56 of the 256 CUs idle: extra workgroups of this call's document-kernel launch then prepare
 * `next`'s preamble (src/lda.cpp:172-173: row sums, exp(psi(lambda)) on its words) in alternate
 * buffers, and the call that later runs `next` starts with its document kernel.  Nothing is
 * skipped or shared between batches; a wrong or stale announcement (another batch comes, lambda
 * is written in between) just costs the wasted preparation.  Small tables (K <= 128, documents
 * of at most 192 words); elsewhere `next` is ignored.  trlda_model_set_prefetch(model, 0) makes
 * every call prepare its own preamble in a launch of its own. */
int trlda_model_estep_io_next(trlda_model *model, const trlda_batch *batch, const trlda_batch *next,
                              const double *gamma0_dev, double *gamma_dev, double *sstats_dev,
                              int max_iter, double threshold, int32_t *iters_dev);
int trlda_model_set_prefetch(trlda_model *model, int enabled);

/* Deferred statistics for a STREAM of E-steps on an unchanged lambda (a corpus pass of
 * LDA::updateVariablesVI calls, src/lda.cpp:160-220: the statistics of one call, :207-217, feed
 * nothing of the next).  With the switch on, trlda_model_estep_io_next returns once its document
 * launch is enqueued; the statistics of that call are formed by extra workgroups of the NEXT
 * trlda_model_estep_io_next call's document launch (on the CUs a 200-document batch leaves idle:
 * a step is then one launch), and are written into the `sstats_dev` THAT call was given.  They
 * are complete -- enqueued on the model's stream ahead of everything that follows -- after the
 * next trlda_model_estep_io_next call or after ANY other call on the model (trlda_model_flush,
 * trlda_model_synchronize, an update, trlda_model_destroy, trlda_batch_destroy of the batch): a
 * caller that reads `sstats_dev` on the stream after one E-step and before the next calls
 * trlda_model_flush first.  gamma and the iteration counts are written by the call itself, as
 * always.  The sums and their order are those of the kernel of its own: bitwise the same
 * statistics.  Same range as the announcement above (small tables, K <= 128 even, batches of at
 * most 256 documents); elsewhere, and for whatever cannot ride along, the statistics are launched
 * as their own kernel.  Off by default: the Python classes never turn it on.
 * trlda_model_last_deferred: bit 0 = the last E-step left its statistics pending, bit 1 = its
 * launch formed the statistics of the call before. */
int trlda_model_set_deferred_stats(trlda_model *model, int enabled);
int trlda_model_flush(trlda_model *model);
int trlda_model_last_deferred(const trlda_model *model);

/* Stream lanes: two E-steps of such a stream in flight at once.  The calls of a corpus pass of
 * LDA::updateVariablesVI (src/lda.cpp:160-220) on an unchanged lambda share nothing but lambda,
 * and a document launch does not end all at once: its documents finish 1-2 us apart (a
 * 129..144-word one 5 us after the others), the helper workgroups at a time of their own.  With
 * two lanes the library deals the stream's calls in turn to two streams of its own (each with the
 * buffers of a model: preamble, factors, pending statistics), so that the next call's workgroups
 * take the CUs as the current one's leave them.
 *   trlda_model_estep_io_ahead   trlda_model_estep_io_next with the batches of the following calls
 *       in order (upcoming[0] is the next call's, upcoming[1] the one after; n_upcoming may be 0):
 *       a lane prepares the preamble of ITS next call, which is two calls ahead.  Identical to
 *       trlda_model_estep_io_next(.., upcoming[0], ..) while lanes are off (the default) or
 *       lambda was last written by an update call (whose kernels leave lambda's row sums behind:
 *       E-steps on that lambda use those).  Any shape goes through the lanes: where deferred
 *       statistics and announcements apply (or are switched on) a call is one launch, elsewhere
 *       its preamble, document and statistics kernels -- whose tails and small launches then run
 *       under the other lane's documents (K = 100, 1600 documents: 188 against 211 us per step).
 *   visibility   gamma, the iteration counts and the statistics of a call that went through a lane
 *       are complete on the model's stream after trlda_model_flush or ANY other call on the model
 *       that is not the next trlda_model_estep_io_ahead -- not after the call itself.  What the
 *       caller enqueued on the model's stream before a call (its gamma0, the last reader of its
 *       output arrays) is waited for by the lane.
 *   outputs   calls in flight together must not share output arrays.  A call whose arrays meet
 *       those of the other lane's outstanding work waits for it (correct, and no faster than one
 *       lane): a streaming caller alternates between two sets of gamma / sstats arrays.
 * Results are those of the one-lane stream, bit for bit.  trlda_model_lane_steps: E-steps that went
 * through the lanes so far. */
int trlda_model_set_stream_lanes(trlda_model *model, int lanes /* 1 or 2 */);
int trlda_model_estep_io_ahead(trlda_model *model, const trlda_batch *batch,
                               const trlda_batch *const *upcoming, int n_upcoming,
                               const double *gamma0_dev, double *gamma_dev, double *sstats_dev,
                               int max_iter, double threshold, int32_t *iters_dev);
/* A corpus pass in ONE call (round 6): LDA::updateVariablesVI (src/lda.cpp:160-220) on an unchanged
 * lambda for every mini-batch of a corpus whose documents lie in HOST memory in CSR form -- `offsets`
 * (n_docs + 1, as trlda_docs_offsets), `ids`, `cnts`; mini-batch i = documents [i * batch_size,
 * min((i + 1) * batch_size, n_docs)) -- the reference's Python loop over do_e_step
 * (python/src/ldainterface.cpp:311-390) with the loop inside the library: batches are made eight ahead
 * of their E-step (trlda_batch_create: index on the worker threads; TRLDA_CORPUS_AHEAD), announced two ahead,
 * stepped through deferred statistics and two lanes (both switched on for the call and put back),
 * destroyed four steps later.  gamma0_dev / gamma_dev: K x n_docs on the device (a document's K
 * values contiguous; gamma0 is only read); the statistics of mini-batch i go to
 * sstats_ring[i % n_ring] (K x V each, n_ring >= 3: whoever wants every batch's statistics passes as
 * many arrays as there are batches); iters_dev: n_docs counts, or NULL.  Everything is complete on
 * the model's stream when the call returns (enqueued: trlda_model_synchronize to wait).  Bitwise the
 * results of the loop of trlda_model_estep_io calls. */
int trlda_model_estep_corpus(trlda_model *model, int64_t n_docs, const int64_t *offsets,
                             const int32_t *ids, const int32_t *cnts, int batch_size,
                             const double *gamma0_dev, double *gamma_dev, double *const *sstats_ring,
                             int n_ring, int max_iter, double threshold, int32_t *iters_dev);
long long trlda_model_lane_steps(const trlda_model *model);
/* What became of the lanes: 0 none made yet (they are made by the first call that goes through them);
 * 2 two lanes on streams that were SEEN to run side by side with each other and with the model's stream
 * -- the runtime hands out hardware queues that are in use once a priority's pool is exhausted, and two
 * lanes on one queue are slower than one lane, so the library makes streams until it holds such a pair
 * (a ~40 us probe kernel on each of two streams: 40 us in all or 80; a marker on one behind a kernel
 * on the other); 1 the lanes were given up -- no such pair was to be had, or the MEASUREMENT said so:
 * after 96 steps through the lanes one launch of a lane and the four that follow it are timed on the
 * device; two launches in flight means a launch LASTS about two steps (51 us where one starts every
 * 26), lanes that do not overlap have launches of one step's length: below 1.0 launches in flight a
 * look says NO (the window ends with the later of the two lanes' launches).  Where a stretch of calls is
 * long enough (64 calls without a flush) every look begins with what the lanes are to beat: eighteen
 * calls in a row on ONE lane, timed the same way -- two lanes that are not 3 % faster than that: NO.  When
 * two of the last four looks said no the lanes are dropped -- the stream of calls goes one launch at a
 * time, as without the switch; a look that said no is followed by the next at once; lanes that pay are
 * kept (on two looks in a row) and looked at AGAIN every 1024 steps for as long as they live: there are
 * process starts in which the launches overlap for a while and then do not.  A window during which the
 * host did not keep the lanes fed is no verdict.  3 two lanes, not looked at (TRLDA_LANE_VERIFY=0).
 * trlda_model_lane_timing: what the last measurement found,
 * microseconds (0: not measured yet). */
int trlda_model_lane_state(const trlda_model *model);
int trlda_model_lane_timing(const trlda_model *model, double *us_per_launch, double *us_per_step);
/* with trlda_model_set_timing on: the summed duration (HIP events on the lanes' streams, one pair per
 * lane around each stretch of calls between joins -- a lane's launches run back to back) and the
 * number of the document launches that went through the lanes; joins the lanes */
int trlda_model_get_lane_timing(trlda_model *model, double *usec_sum, int64_t *launches);

/* Host-pointer convenience around trlda_model_estep (uploads gamma0, downloads
 * gamma / sstats / iters, synchronises). */
int trlda_model_estep_host(trlda_model *model, const trlda_batch *batch,
                           double *gamma, double *sstats,
                           int max_iter, double threshold, int32_t *iters_out);

/* LDA::lowerBound (src/lda.cpp:297-360; python/src/ldainterface.cpp:420-470): an E-step on
 * the batch from gamma (K x B host, in: gamma0, out: gamma), then the variational lower
 * bound with the correction factor = num_documents / B of :302-303 (pass 1 for none).
 * phi is recomputed with the column of the word (the reference's :334 indexes the row; see
 * DESIGN.md "lower bound"). */
int trlda_model_lower_bound(trlda_model *model, const trlda_batch *batch, double *gamma,
                            double eta, double factor, int max_iter, double threshold,
                            double *bound_out);

/* model.lambda = (1-rho) lambda' + rho (eta + scale * sstats), all device pointers.
 * src/onlinelda.cpp:99-100, :108-109; src/batchlda.cpp:60 (rho = 1, scale = 1). */
int trlda_model_blend(trlda_model *model, const double *lambda_prime_dev,
                      const double *sstats_dev, double rho, double eta, double scale);

/* model.lambda = (1-rho) lambda' (+row) rho (eta + D/B/K * wordcounts): src/onlinelda.cpp:79-86 */
int trlda_model_tr_init(trlda_model *model, const trlda_batch *batch,
                        const double *lambda_prime_dev, double rho, double eta,
                        int num_documents);

/* The two halves of trlda_model_tr_init, for the multi-GPU composition where the word
 * counts of the whole mini-batch are the sum over ranks (src/onlinelda.cpp:79-82 is a sum
 * over documents): local counts -> all-reduce -> apply.
 *   wordcounts_dev  V fp64 (integer-valued, so the sum is exact in any order)
 *   coef            D / B_total / K, evaluated by the caller in that order (onlinelda.cpp:86) */
int trlda_model_wordcounts(trlda_model *model, const trlda_batch *batch, double *wordcounts_dev);
int trlda_model_tr_init_wc(trlda_model *model, const double *wordcounts_dev,
                           const double *lambda_prime_dev, double rho, double eta, double coef);

/* dst_dev[K x V] = model.lambda (device-to-device, on the model's stream): the
 * `lambdaPrime = mLambda` snapshot of src/onlinelda.cpp:68. */
int trlda_model_copy_lambda(trlda_model *model, double *dst_dev);

/*
 * OnlineLDA::updateParameters, lambda path: src/onlinelda.cpp:53-111, 177-179 (the
 * caller of python/src/onlineldainterface.cpp:204-256).  Runs the whole
 * trust-region loop on the device; gamma initial values are drawn on the host from
 * libc rand() at exactly the points the reference draws them (src/lda.cpp:135) so
 * trlda_seed() pins the trajectory.  threshold is the reference's fixed 0.001
 * unless overridden.  Returns rho through *rho_out and increments *update_count
 * (not for an empty batch: src/onlinelda.cpp:54-56, which returns 1.0).
 *   gamma_out  optional host K x B: gamma of the last E-step
 * Single GPU.  The multi-GPU composition (E-step -> RCCL all-reduce of sstats ->
 * blend) is done by the host mirror from the three calls above.
 * Returns without waiting for the device unless gamma_out is given (every getter
 * synchronises): the host's draw of the next gamma0 overlaps this call's kernels.
 */
int trlda_model_online_update(trlda_model *model, const trlda_batch *batch,
                              int num_documents, double eta,
                              int max_iter_tr, int max_iter_inference,
                              double kappa, double tau, double rho,
                              int init_gamma, int update_lambda, double threshold,
                              int *update_count, double *rho_out, double *gamma_out);

/*
 * BatchLDA::updateParameters, lambda path: src/batchlda.cpp:43-61 -- max_epochs x
 * { E-step from a fresh random gamma; lambda = eta + sstats }.
 */
int trlda_model_batch_update(trlda_model *model, const trlda_batch *batch, double eta,
                             int max_epochs, int max_iter_inference, int update_lambda,
                             double threshold, double *gamma_out);

/*
 * CumulativeLDA::updateParameters, lambda path: src/cumulativelda.cpp:49-72 -- lambda' =
 * lambda; lambda = sampleGamma(K, V, 100)/100 (drawn whether or not update_lambda is set,
 * like the reference); max_epochs x { E-step from a fresh random gamma; lambda = lambda' +
 * sstats }.
 */
int trlda_model_cumulative_update(trlda_model *model, const trlda_batch *batch, int max_epochs,
                                  int max_iter_inference, int update_lambda, double threshold,
                                  double *gamma_out);

/* ---- multi-GPU: documents shard across ranks, one RCCL all-reduce per E-step ---------------
 *
 * One process per GPU, lambda replicated, each rank holds a contiguous range of the
 * mini-batch's documents.  Documents are independent given lambda (src/lda.cpp:176-214 touches
 * only column i of gamma and ADDS into sstats), so the path's one exchange is the sum of the
 * K x V statistics where the reference has its omp-critical reduction (src/lda.cpp:211-217),
 * plus the V word counts of src/onlinelda.cpp:79-82.  `rccl_comm` is the host program's
 * ncclComm_t (from ncclCommInitRank), passed as an opaque pointer; ncclAllReduce is resolved
 * at run time from the process or librccl.so, and runs on the model's stream. */
/* sstats_dev[K x V] <- sum over ranks */
int trlda_model_allreduce_sstats(trlda_model *model, void *rccl_comm, double *sstats_dev);
/* OnlineLDA::updateParameters (src/onlinelda.cpp:53-111, 177-179) over the ranks of rccl_comm:
 *   shard       this rank's documents [doc_lo, doc_lo + trlda_batch_num_docs(shard)) of a
 *               mini-batch of total_docs
 * gamma0 is drawn for the whole mini-batch from the host stream on every rank (trlda_seed /
 * trlda_rng_set_state keep the ranks' streams equal) and this rank's columns are kept, so an
 * N-rank run reproduces the one-rank trajectory up to the summation order of the all-reduce.
 * Every rank applies the identical M-step: lambda stays replicated without a broadcast. */
int trlda_model_online_update_multi(trlda_model *model, const trlda_batch *shard, void *rccl_comm,
                                    int total_docs, int doc_lo, int num_documents, double eta,
                                    int max_iter_tr, int max_iter_inference, double kappa,
                                    double tau, double rho, int init_gamma, double threshold,
                                    int *update_count, double *rho_out);

/* BatchLDA::updateParameters, lambda path (src/batchlda.cpp:43-61), over the ranks of rccl_comm:
 * max_epochs x { this rank's documents from a fresh random gamma -- its columns of the whole
 * batch's draw, as above; ONE all-reduce of the K x V statistics (src/lda.cpp:211-217 across
 * ranks); lambda = eta + sstats on every rank (src/batchlda.cpp:60) }.  BASELINE config 4: 100 000
 * documents over 8 ranks, 80 MB per all-reduce.  gamma of the shard's documents stays on the
 * device for trlda_model_eb_gamma_stats_multi. */
int trlda_model_batch_update_multi(trlda_model *model, const trlda_batch *shard, void *rccl_comm,
                                   int total_docs, int doc_lo, double eta, int max_epochs,
                                   int max_iter_inference, int update_lambda, double threshold);
/* buf_dev[count] <- sum over ranks (fp64), on the model's stream: the K-vector reduction of
 * src/onlinelda.cpp:128 and any other sum a host composes across ranks */
int trlda_model_allreduce(trlda_model *model, void *rccl_comm, double *buf_dev, size_t count);
/* trlda_model_estep_resident for this rank's documents of a sharded mini-batch (gamma0 = its
 * columns of the whole mini-batch's draw; no exchange, the statistics are not used) */
int trlda_model_estep_resident_shard(trlda_model *model, const trlda_batch *shard, int total_docs,
                                     int doc_lo, int max_iter, double threshold);
/* trlda_model_eb_gamma_stats summed over the ranks' documents: out_host[k] = sum over ALL
 * documents of psi(gamma_dk) - psi(sum_k gamma_dk) (src/onlinelda.cpp:123-128,
 * src/batchlda.cpp:72-74); B = this rank's document count (0 allowed: the rank still takes part
 * in the all-reduce of K doubles).  rccl_comm NULL: this rank's sums only. */
int trlda_model_eb_gamma_stats_multi(trlda_model *model, void *rccl_comm, int B,
                                     const double *gamma_dev, double *out_host /* K */);

/* ---- multi-GPU with FACTOR exchange: an all-gather of 8 (K + n_d) bytes per document --------
 *
 * The same reduction point (src/lda.cpp:211-217), without moving K x V numbers.  lambda is
 * replicated, and sstats[k, w] = expElogbeta[k, w] * sum_{(d, w)} (cnt_dw / phinorm_dw) *
 * expElogtheta[d, k] (src/lda.cpp:207-217): a rank only has to publish, per document, the K
 * numbers expElogtheta[d, :] and one weight per (document, word) entry.  Every rank holds the
 * WHOLE mini-batch (`batch`: word lists, counts -- a few hundred kB) and iterates the documents
 * [doc_cuts[rank], doc_cuts[rank + 1]) of it (`shard`, a batch made of exactly those
 * documents); one ncclAllGather of equal slots follows the document stage, and every rank forms
 * the statistics of the whole mini-batch with the single-GPU kernel -- which adds the entries of
 * a word in document order, the reference's serial order.  Every rank performs the same
 * additions in the same order on the same numbers: the replicas of lambda stay bitwise equal
 * without a broadcast, and they equal the one-GPU result for the whole mini-batch up to the
 * rounding of the document-kernel variant a shard selects (~1e-15; bitwise when the shards and
 * the whole mini-batch select the same variant) -- no dependence on the rank count through an
 * order of summation.  The fused M-step with carried row sums applies unchanged.  Bytes received per rank and E-step: (world - 1) * slot * 8, slot ~
 * max_r(docs_r) * K + max_r(nnz_r): 2.1 MB at K = 100, 8 x 200 documents (the K x V all-reduce
 * moves 9.8 MB per rank); 17 MB at K = 500, 8 x 512 documents (700 MB).  Not the better choice
 * when documents outnumber words (BatchLDA on 8 x 12 500 documents, V = 50 000): compare
 * world * slot with 2 * K * V.
 *
 * `rccl_comm` as above (ncclAllGather, in place, on the model's stream); world = 1 needs none.
 * A host with a transport of its own installs trlda_model_set_allgather instead: the hook is
 * called with this rank's slot (`send` = recv + rank * count), the buffer of world * count
 * doubles and the model's hipStream_t, and must leave every rank's slot in `recv` in stream
 * order; a non-zero return fails the call. */
typedef int (*trlda_allgather_fn)(void *ctx, const void *send_dev, void *recv_dev,
                                  size_t count_f64, void *hip_stream);
int trlda_model_set_allgather(trlda_model *model, trlda_allgather_fn fn, void *ctx);
/* WORD-SHARDED M-STEP (default for world > 1 where a transport for it exists).  After the factor
 * exchange every rank holds every document's factors, so any rank can form any word's statistics:
 * rank r forms statistics + M-step (src/lda.cpp:207-217, src/onlinelda.cpp:99-100,
 * src/batchlda.cpp:60) for ITS contiguous range of the vocabulary only -- the ranges are balanced
 * by the words' entries, the same cut points on every rank -- and the ranks then exchange the
 * lambda columns they wrote, IN PLACE in the replicated table.  Every column is computed once, by
 * its owner, with a word's entries added in document order: the replicas are bitwise equal and
 * equal the one-GPU result.  Per rank and E-step the statistics stage shrinks by the world size
 * (8 x 200 documents, K = 100: 8 us instead of 38 us of kernels), and the second exchange moves
 * the lambda table once (5.6 MB received per rank at K = 100, V = 7000 -- half of what the
 * all-reduce of the statistics moves).  The next E-step's row sums and exp(psi(lambda)) are formed
 * from the exchanged lambda by its own preamble.
 * Transport: `rccl_comm` (one grouped launch of `world` ncclBroadcast calls: ranges of unequal
 * size), or the host's own -- the hook is called with the table, world + 1 offsets in doubles
 * (rank r wrote [offsets[r], offsets[r + 1])), this rank, the world size and the model's
 * hipStream_t, and must leave every range in place in stream order.  Without either (only the
 * equal-slot hook above, or the direct slot exchange) every rank forms the statistics of the
 * whole mini-batch as before.  set_word_sharding(model, 0) does the same by choice. */
typedef int (*trlda_allgatherv_fn)(void *ctx, void *table_dev, const size_t *offsets_f64 /* world + 1 */,
                                   int rank, int world, void *hip_stream);
int trlda_model_set_allgatherv(trlda_model *model, trlda_allgatherv_fn fn, void *ctx);
int trlda_model_set_word_sharding(trlda_model *model, int enabled);
int trlda_model_last_word_sharded(const trlda_model *model);
/* A DIRECT exchange of the slots, behind this switch (the all-gather above stays the default):
 * xGMI is point to point and a slot is a few hundred kB, so every rank writes its slot straight
 * into its peers' gather buffers -- device memory the peers export through hipIpc -- and signals
 * them with a step counter there; no collective launch sits on a step's critical path.
 *   alloc    a region for slots of up to max_slot_f64 doubles (two buffers, alternating per E-step,
 *            + the counters); handle_out receives the 64-byte hipIpcMemHandle_t to give to the peers
 *   connect  handles = world x 64 bytes in rank order (this rank's own is ignored): maps the
 *            peers' regions; the *_dp entry points then use the direct exchange for this world
 *   close    unmaps and frees (also done by trlda_model_destroy)
 * Every rank must make the same sequence of *_dp calls (they do: lambda is replicated).  A peer
 * that never signals makes the next trlda_model_synchronize fail instead of hanging. */
int trlda_model_dp_direct_alloc(trlda_model *model, size_t max_slot_f64, int world, void *handle_out);
int trlda_model_dp_direct_connect(trlda_model *model, int rank, int world, const void *handles);
int trlda_model_dp_direct_close(trlda_model *model);
/* OnlineLDA::updateParameters (src/onlinelda.cpp:53-111, 177-179) over `world` ranks: the
 * arguments of trlda_model_online_update, which it equals for world = 1.  gamma0 is this rank's
 * columns of the whole mini-batch's draw (the stream advances by all of it on every rank). */
int trlda_model_online_update_dp(trlda_model *model, const trlda_batch *batch,
                                 const trlda_batch *shard, void *rccl_comm, int rank, int world,
                                 const int32_t *doc_cuts /* world + 1 */, int num_documents,
                                 double eta, int max_iter_tr, int max_iter_inference, double kappa,
                                 double tau, double rho, int init_gamma, double threshold,
                                 int *update_count, double *rho_out);
/* BatchLDA::updateParameters (src/batchlda.cpp:43-61) with the factor exchange: the arguments of
 * trlda_model_batch_update, which it equals for world = 1.  Pays where the words outnumber the
 * documents' factors (world * slot < 2 K V); BASELINE config 4 is the opposite case
 * (trlda_model_batch_update_multi). */
int trlda_model_batch_update_dp(trlda_model *model, const trlda_batch *batch,
                                const trlda_batch *shard, void *rccl_comm, int rank, int world,
                                const int32_t *doc_cuts /* world + 1 */, double eta, int max_epochs,
                                int max_iter_inference, int update_lambda, double threshold);
/* One E-step (src/lda.cpp:160-220) over `world` ranks: gamma0_dev / gamma_dev hold this rank's
 * K x docs_r columns (gamma0_dev NULL: in place); sstats_dev (K x V, may be NULL with mstep)
 * receives the statistics of the whole mini-batch; iters_dev this rank's iteration counts or
 * NULL.  mstep != 0: the statistics kernel also writes lambda = (1 - rho) lambda' + rho (eta +
 * scale * sstats) for every word (src/onlinelda.cpp:99-100; lambda_prime_dev NULL: lambda'
 * is the current lambda) and leaves the row sums of the new lambda for the next E-step. */
int trlda_model_estep_dp(trlda_model *model, const trlda_batch *batch, const trlda_batch *shard,
                         void *rccl_comm, int rank, int world, const int32_t *doc_cuts,
                         const double *gamma0_dev, double *gamma_dev, double *sstats_dev,
                         int max_iter, double threshold, int32_t *iters_dev, int mstep,
                         const double *lambda_prime_dev, double rho, double eta, double scale);

/* ---- update loop: what stays on the device between its steps ----------------------------
 *
 * Inside one OnlineLDA update lambda' and rho are fixed and a word outside the mini-batch has
 * zero statistics (src/lda.cpp:169), so every M-step of the trust-region loop
 * (src/onlinelda.cpp:99-100) gives it the same value -- the one the initial step of :85-86
 * gives it too.  The update entry points above therefore write those words once per call and
 * run the loop on the batch's active words only, with the statistics (src/lda.cpp:207-217), the
 * M-step and the row sums the next E-step needs (src/lda.cpp:172) in ONE kernel per iteration.
 * The switches below exist for tests and before/after measurements; results do not depend on
 * them beyond summation order. */
/* fused = 0: separate statistics and M-step kernels over all V words (K > 512 always does) */
int trlda_model_set_fused_update(trlda_model *model, int fused);
/* Small tables (K <= 128, K * V < 2^22): while a lambda element is in its registers the M-step
 * kernel also writes exp(psi(.)) of it (src/lda.cpp:173's numerator) and combines the row sums
 * (src/lda.cpp:172) into a few rows, so the next E-step on the same words starts with its
 * document kernel: two launches per trust-region iteration instead of three.  enabled = 0: the
 * next E-step launches its preamble kernel as before (default 1). */
int trlda_model_set_next_preamble(trlda_model *model, int enabled);
/* carry = 0: every E-step adds up the rows of lambda again (src/lda.cpp:172 as written) */
int trlda_model_set_carry_rowsums(trlda_model *model, int carry);
/* keep = 1: the update entry points also leave the sufficient statistics of their last E-step
 * and the complete lambda' on the device (what src/onlinelda.cpp:167-175 reads); costs two
 * K x V writes per call. */
int trlda_model_set_keep_sstats(trlda_model *model, int keep);
/* gamma0 of the update entry points (sampleGamma(K, B, 100) / 100, src/lda.cpp:135) is drawn ON
 * THE DEVICE from the host's libc stream -- the same integers in the same order, hence the same
 * uniforms u, bit for bit -- and the host stream is advanced by the same number of draws.  The
 * device forms -sum_p log|u_p| as -sum over blocks of 25 passes of log(prod_p |u_p|) (one
 * logarithm per 25 draws; the product of 25 values > 2^-31 is a normal double): within 4e-15
 * relative of the host's sum -- which is itself 7e-16 off the exact value, the product form 2e-16
 * (csrc/rng_kernels.h).  host = 1: draw on the host instead, bit for bit the reference's values
 * (K * B * 100 glibc logarithms per call). */
int trlda_model_set_host_gamma_draw(trlda_model *model, int host);
/* The device draw of the NEXT fresh gamma0 of the same shape, made AHEAD of its turn while the current
 * call's kernels run, with the host stream advanced ahead of its turn.  Whatever else touches the
 * generator first -- trlda_seed, a host draw, another model, another shape -- puts the stream back
 * and the draw is repeated in its turn: the ORDER of draws is the reference's in every case
 * (src/lda.cpp:135, src/utils.cpp:224-231).
 *   2 (default)  inside the call's document launch: on small tables (K <= 128, <= 224 document
 *                workgroups) extra workgroups of the launch draw it on the CUs the documents leave
 *                free (csrc/rng_kernels.h, aux_draw_workgroup) -- no launch, no stream of its own;
 *                where the launch cannot carry it the draw is made in its turn;
 *   1            on a stream of the model's own (TRLDA_DRAW_AHEAD=1): gains 13 % at 1600 documents
 *                without trust-region loop, loses where the process has more streams than hardware
 *                queues (bench.py beside torch);
 *   0            every draw in its turn.
 * The values are bitwise the same in all three. */
int trlda_model_set_draw_ahead(trlda_model *model, int enabled);
/* gamma0 draws made inside a document launch so far (tests) */
long long trlda_model_inlaunch_draws(const trlda_model *model);
/* OnlineLDA::updateParameters without trust-region loop (src/onlinelda.cpp:103-109): the decay of the
 * words outside the mini-batch, lambda = (1 - rho) lambda + rho eta, by auxiliary workgroups of the
 * call's document launch (1, the default; small tables) or by the streaming kernel behind it (0).
 * Bitwise the same lambda and row sums either way. */
int trlda_model_set_aux_decay(trlda_model *model, int enabled);
long long trlda_model_inlaunch_decays(const trlda_model *model);   /* (tests) */
/* out_dev[rows x cols] = sampleGamma(rows, cols, passes) / divisor (src/utils.cpp:224-231) on
 * the device, as above (divisor 1 for the bare sum). */
int trlda_model_sample_gamma(trlda_model *model, int rows, int cols, int passes, double divisor,
                             double *out_dev);
/* The columns [col_lo, col_hi) of that matrix only (out_dev: rows x (col_hi - col_lo)); the
 * stream still advances by rows * cols * passes draws.  A data-parallel rank draws its own
 * documents' gamma0 out of the mini-batch's matrix this way, in step with the other ranks. */
int trlda_model_sample_gamma_cols(trlda_model *model, int rows, int cols, int col_lo, int col_hi,
                                  int passes, double divisor, double *out_dev);
/* bytes this model has copied to host memory so far (tests assert that the empirical-Bayes
 * steps move O(K), not O(K V)) */
int64_t trlda_model_d2h_bytes(const trlda_model *model);

/* LDA::updateVariables(documents, parameters) from a fresh random gamma (src/lda.cpp:119-138)
 * with gamma and the statistics left on the device: what src/onlinelda.cpp:118-120 and
 * src/batchlda.cpp:66-68 do before an alpha step when update_lambda is off. */
int trlda_model_estep_resident(trlda_model *model, const trlda_batch *batch, int max_iter,
                               double threshold);

/* out_host[k] = sum_d psi(gamma_dk) - psi(sum_k gamma_dk): the data term of the alpha gradient,
 * src/onlinelda.cpp:123-128, src/batchlda.cpp:72-74, src/cumulativelda.cpp:82-84.
 *   gamma_dev  K x B device array, or NULL: gamma of the model's last update / resident E-step */
int trlda_model_eb_gamma_stats(trlda_model *model, int B, const double *gamma_dev,
                               double *out_host /* K */);

/* *sum_psi_lambda = sum_kw psi(lambda_kw); rowsums_host[k] = sum_w lambda_kw (the host applies
 * psi to those K numbers): the data terms of the eta gradient, src/onlinelda.cpp:152-154,
 * src/batchlda.cpp:152. */
int trlda_model_eb_lambda_stats(trlda_model *model, double *sum_psi_lambda,
                                double *rowsums_host /* K */);

/* ---- empirical Bayes: the K- and scalar-sized host steps (csrc/eb_steps.cpp) --------------- */

/* alpha <- max(alpha - rho H^-1 g, min_alpha): one natural-gradient step, src/onlinelda.cpp:123-142;
 * psi_gamma_diff[k] = sum over the mini-batch's documents of psi(gamma_dk) - psi(sum_k gamma_dk)
 * (trlda_model_eb_gamma_stats), num_docs = the mini-batch's size.  Host memory, no GPU. */
int trlda_eb_online_alpha_step(int K, const double *alpha, const double *psi_gamma_diff,
                               double num_docs, double rho, double min_alpha, double *alpha_out);
/* eta <- max(eta - rho g / h, min_eta): src/onlinelda.cpp:147-162 (sums from
 * trlda_model_eb_lambda_stats) */
double trlda_eb_online_eta_step(double eta, double sum_psi_lambda, const double *rowsums, int K,
                                int V, double rho, double min_eta);
/* Newton steps with a step-halving line search on the lower bound: src/batchlda.cpp:81-141 ==
 * src/cumulativelda.cpp:90-150 (alpha), src/batchlda.cpp:147-205 (eta) */
int trlda_eb_alpha_line_search(int K, const double *alpha, const double *psi_gamma_diff,
                               double num_docs, int max_iter_alpha, double min_alpha,
                               double threshold, double *alpha_out);
double trlda_eb_eta_line_search(double eta, double sum_psi_lambda, const double *rowsums, int K,
                                int V, int max_iter_eta, double min_eta, double threshold);
/* `verbosity` of LDA::Parameters for the two line searches above (per calling thread): above 1 they
 * print their progress to stdout as the reference does (src/batchlda.cpp:78-88,120-123,155-165,
 * 184-187; src/cumulativelda.cpp:87-97,129-132). */
void trlda_eb_set_verbosity(int verbosity);
/* Both online steps after an update, with ONE synchronisation: the device sums over the gamma
 * the update left behind (B_local documents of this rank; summed over the ranks of rccl_comm
 * when that is not NULL) and over lambda, the host steps above, the new alpha back on the
 * device.  alpha_host (K) and *eta are read and updated; B_total = the mini-batch's size. */
int trlda_model_online_eb(trlda_model *model, void *rccl_comm, int B_local, int B_total, double rho,
                          int update_alpha, int update_eta, double min_alpha, double min_eta,
                          double *alpha_host, double *eta);
/* The same in two halves, so that the host can prepare the next mini-batch (parse, convert,
 * upload) while the device still works on this one: _begin enqueues the sums and their way back
 * to pinned host memory and returns; _finish waits for them, takes the steps and puts the new
 * alpha on the device.  Between the two the model runs no E-step (TRLDA_ERR_ARG): the caller
 * finishes before its next call -- trlda.models.OnlineLDA does, at the start of the next
 * update_parameters / do_e_step / lower_bound and in the alpha / eta getters.  _pending: 1 between
 * the halves. */
int trlda_model_online_eb_begin(trlda_model *model, void *rccl_comm, int B_local, int B_total,
                                int update_alpha, int update_eta);
int trlda_model_online_eb_finish(trlda_model *model, double rho, double min_alpha, double min_eta,
                                 double *alpha_host, double *eta);
int trlda_model_online_eb_pending(const trlda_model *model);
/* test hook: psi and psi' as eb_steps.cpp evaluates them */
void trlda_debug_host_psi(int n, const double *x, double *psi_out, double *psi1_out);

/* Adaptive learning rate, src/onlinelda.cpp:167-172, after an update made with keep_sstats on:
 * lambdaUpdate = (eta + scale * sstats) - lambda'; the model's running average
 * mAdaGradient = (1 - 1/tau) mAdaGradient + 1/tau lambdaUpdate (K x V, device, zero at first);
 * returns |lambdaUpdate|^2 and |mAdaGradient|^2.  eta: the value the update used. */
int trlda_model_adaptive_stats(trlda_model *model, double eta, double scale, double tau,
                               double *sq_norm_update, double *sq_norm_gradient);

/* The same with the statistics and lambda' given as device arrays (K x V each): for a host that
 * composes the data-parallel update itself (E-step -> all-reduce -> blend) and holds both. */
int trlda_model_adaptive_stats_dev(trlda_model *model, const double *sstats_dev,
                                   const double *lambda_prime_dev, double eta, double scale,
                                   double tau, double *sq_norm_update, double *sq_norm_gradient);

/* ---- test hook ----------------------------------------------------------- */

/* The device digamma (TRLDA::digamma, src/digamma.cpp:116-178, as compiled for gfx950)
 * evaluated at n host points: psi[i] = psi(x[i]); epsi[i] = exp(psi(x[i])) in the log-free
 * form the hot path uses (lda.cpp:173-174, :197 only ever need exp(psi)); epsi_lean[i] = the
 * call-free form for positive arguments that the M-step kernels use when they leave the next
 * E-step's exp(psi(lambda)) behind (bitwise the same except at the integers 1..10, where the
 * reference's exact harmonic branch and the regular form differ by a few ulp; for x <= 0 the
 * general form); eminus[i] = exp(psi(x[i]) - c)
 * (lda.cpp:173).  Lets the parity tests check the special functions against the reference's
 * table directly. */
int trlda_debug_digamma(int device, int n, double c, const double *x, double *psi, double *epsi,
                        double *epsi_lean, double *eminus);
/* test hook: the transposing wave reductions of csrc/estep_wide.h on a 64 x 16 table
 * (row = lane); out16 / out4 / out2 [64] = what each lane receives from the 16-, 4- and
 * 2-value folds of its row's leading values */
int trlda_debug_fold16(int device, const double *in, double *out16, double *out4, double *out2);
/* diagnostics: copy out one of the model's intermediate buffers of its last E-step (after
 * synchronising its stream).  which = 0: exp(psi(lambda)) / exp E[log beta] as the kernels read it
 * (K x V; only the batch's words are filled), 1: the documents' exp E[log theta] rows, 2: cnt /
 * phinorm per entry in word order, 3: the sstats buffer of the host entry points (K x V).
 * tests/fuzz_estep.py --passes uses it to say WHICH input of the statistics went wrong. */
int trlda_debug_peek(trlda_model *model, int which, double *host_out, size_t count);
/* diagnostics: the s_memrealtime stamps of the model's last merged launch, 3 x 1024 values
 * (TRLDA_MERGED_STAMPS=1; tools/merged_stamps.py) */
int trlda_debug_merged_stamps(trlda_model *model, unsigned long long *host_out);

/* ---- measurement --------------------------------------------------------- */

/* Average duration in microseconds (HIP events on the model's stream) of the
 * kernels launched by the most recent trlda_model_estep when timing is on:
 * which = 0 row sums (or the fused preamble, then 1 is empty), 1 exp E[log beta],
 * 2 per-document fixed point, 3 sufficient statistics, 4 nothing: two event records back to
 * back, i.e. the part of every other figure that is the events' own cost.  Timing adds
 * event records to the stream. */
int trlda_model_set_timing(trlda_model *model, int enabled);
int trlda_model_get_timing(trlda_model *model, int which, double *usec_sum, int64_t *count);

#ifdef __cplusplus
}
#endif
#endif /* TRLDA_HIP_H */
