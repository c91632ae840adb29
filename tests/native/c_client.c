/* A plain C client of the drop-in boundary (include/trlda_hip.h): no Python, no torch, no C++ --
 * what the reference's src/lda.cpp / src/onlinelda.cpp would link against.  Built with gcc and run
 * by tests/test_gpu_c_client.py on the GPU box.
 *
 *   one E-step through the one-shot entry (LDA::updateVariables, src/lda.cpp:142-220), then the
 *   resident form: model + batch handles, two OnlineLDA::updateParameters calls
 *   (src/onlinelda.cpp:53-111), lambda read back.
 *
 * Prints the size-independent invariants the test checks:
 *   sum(sstats) = sum(counts);  sum(gamma) = sum(counts) + B sum(alpha);
 *   sum(lambda) after an update = (1 - rho) sum(lambda') + rho (K V eta + D/B sum(counts)).
 *
 * With a file name as its argument it first runs a GOLDEN VECTOR of the compiled reference
 * (tests/golden/f1a_estep.npz, flattened by tests/test_gpu_c_client.py: documents, seeds of
 * lambda and gamma0, alpha, and the reference's gamma / sstats / the oracle's iteration counts at
 * max_iter = 20) through the one-shot entry AND through the handle API, and prints the largest
 * relative error of gamma and of the statistics for each -- values, not only mass balances. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "trlda_hip.h"

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ != TRLDA_OK) {                                                           \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, trlda_last_error());           \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

static double max_rel_err(const double *got, const double *want, size_t n, int *zeros_agree)
{
    double worst = 0.0;
    for (size_t i = 0; i < n; ++i) {
        if (want[i] == 0.0 || got[i] == 0.0) {       /* untouched words: exactly zero on both sides */
            if (want[i] != got[i])
                *zeros_agree = 0;
            continue;
        }
        const double e = fabs(got[i] - want[i]) / fabs(want[i]);
        worst = e > worst ? e : worst;
    }
    return worst;
}

/* file: int32 header {K, V, B, nnz, lambda_seed, gamma0_seed, max_iter, 0}, indptr[B + 1], ids[nnz],
 * cnts[nnz] (int32), then doubles alpha[K], gamma[K B], sstats[K V], then int32 iters[B] */
static int run_golden(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "cannot open %s\n", path);
        return 1;
    }
    int32_t h[8];
    if (fread(h, sizeof(int32_t), 8, f) != 8)
        return 1;
    const int K = h[0], V = h[1], B = h[2], nnz = h[3], max_iter = h[6];
    int32_t *indptr = malloc((size_t)(B + 1) * sizeof(int32_t));
    int32_t *ids = malloc((size_t)nnz * sizeof(int32_t)), *cnts = malloc((size_t)nnz * sizeof(int32_t));
    int32_t *iters_want = malloc((size_t)B * sizeof(int32_t)), *iters = malloc((size_t)B * sizeof(int32_t));
    double *alpha = malloc((size_t)K * sizeof(double));
    double *g_want = malloc((size_t)K * B * sizeof(double)), *s_want = malloc((size_t)K * V * sizeof(double));
    double *lambda = malloc((size_t)K * V * sizeof(double)), *gamma0 = malloc((size_t)K * B * sizeof(double));
    double *gamma = malloc((size_t)K * B * sizeof(double)), *sstats = malloc((size_t)K * V * sizeof(double));
    size_t ok = fread(indptr, sizeof(int32_t), (size_t)B + 1, f) == (size_t)B + 1;
    ok = ok && fread(ids, sizeof(int32_t), (size_t)nnz, f) == (size_t)nnz;
    ok = ok && fread(cnts, sizeof(int32_t), (size_t)nnz, f) == (size_t)nnz;
    ok = ok && fread(alpha, sizeof(double), (size_t)K, f) == (size_t)K;
    ok = ok && fread(g_want, sizeof(double), (size_t)K * B, f) == (size_t)K * B;
    ok = ok && fread(s_want, sizeof(double), (size_t)K * V, f) == (size_t)K * V;
    ok = ok && fread(iters_want, sizeof(int32_t), (size_t)B, f) == (size_t)B;
    fclose(f);
    if (!ok) {
        fprintf(stderr, "short read on %s\n", path);
        return 1;
    }
    trlda_seed((unsigned)h[4]);
    trlda_sample_gamma_init(K, V, lambda);            /* srand(seed); sampleGamma(K, V, 100) / 100 */
    trlda_seed((unsigned)h[5]);
    trlda_sample_gamma_init(K, B, gamma0);

    /* one shot, host pointers (LDA::updateVariables with latents, src/lda.cpp:142-220) */
    memcpy(gamma, gamma0, (size_t)K * B * sizeof(double));
    CHECK(trlda_estep(K, V, B, indptr, ids, cnts, lambda, alpha, gamma, sstats, max_iter, 1e-3, iters, 0));
    int zeros = 1;
    double ge = max_rel_err(gamma, g_want, (size_t)K * B, &zeros);
    double se = max_rel_err(sstats, s_want, (size_t)K * V, &zeros);
    printf("golden oneshot gamma_err %.3e sstats_err %.3e zeros_agree %d iters_equal %d\n", ge, se, zeros,
           memcmp(iters, iters_want, (size_t)B * sizeof(int32_t)) == 0);

    /* the handle API: resident model and batch */
    trlda_model *model = NULL;
    trlda_batch *batch = NULL;
    CHECK(trlda_model_create(&model, 0, K, V));
    CHECK(trlda_model_set_alpha(model, alpha));
    CHECK(trlda_model_set_lambda(model, lambda));
    CHECK(trlda_batch_create(&batch, 0, V, B, indptr, ids, cnts));
    memcpy(gamma, gamma0, (size_t)K * B * sizeof(double));
    memset(sstats, 0xff, (size_t)K * V * sizeof(double));
    memset(iters, 0xff, (size_t)B * sizeof(int32_t));
    CHECK(trlda_model_estep_host(model, batch, gamma, sstats, max_iter, 1e-3, iters));
    zeros = 1;
    ge = max_rel_err(gamma, g_want, (size_t)K * B, &zeros);
    se = max_rel_err(sstats, s_want, (size_t)K * V, &zeros);
    printf("golden handle gamma_err %.3e sstats_err %.3e zeros_agree %d iters_equal %d\n", ge, se, zeros,
           memcmp(iters, iters_want, (size_t)B * sizeof(int32_t)) == 0);

    /* a corpus pass on the fixed lambda as a STREAM: deferred statistics, two calls in flight
     * (trlda_model_set_stream_lanes, trlda_model_estep_io_ahead), device arrays, one set per call;
     * every call of the stream is the golden batch, so every call's results are the golden ones */
    {
        enum { CALLS = 5 };
        void *g0_dev = NULL, *g_dev[CALLS], *s_dev[CALLS], *it_dev[CALLS];
        const size_t gb = (size_t)K * B * sizeof(double), sb = (size_t)K * V * sizeof(double);
        CHECK(trlda_dev_alloc(0, gb, &g0_dev));
        CHECK(trlda_dev_upload(0, g0_dev, gamma0, gb));
        for (int c = 0; c < CALLS; ++c) {
            CHECK(trlda_dev_alloc(0, gb, &g_dev[c]));
            CHECK(trlda_dev_alloc(0, sb, &s_dev[c]));
            CHECK(trlda_dev_alloc(0, (size_t)B * sizeof(int32_t), &it_dev[c]));
        }
        CHECK(trlda_model_set_deferred_stats(model, 1));
        CHECK(trlda_model_set_stream_lanes(model, 2));
        const trlda_batch *upcoming[2] = {batch, batch};
        for (int c = 0; c < CALLS; ++c)
            CHECK(trlda_model_estep_io_ahead(model, batch, upcoming, c + 2 < CALLS ? 2 : CALLS - 1 - c,
                                             (const double *)g0_dev, (double *)g_dev[c], (double *)s_dev[c],
                                             max_iter, 1e-3, (int32_t *)it_dev[c]));
        const long long through = trlda_model_lane_steps(model);
        CHECK(trlda_model_synchronize(model));        /* joins the lanes */
        double ge_worst = 0.0, se_worst = 0.0;
        int its_ok = 1;
        zeros = 1;
        for (int c = 0; c < CALLS; ++c) {
            CHECK(trlda_dev_download(0, gamma, g_dev[c], gb));
            CHECK(trlda_dev_download(0, sstats, s_dev[c], sb));
            CHECK(trlda_dev_download(0, iters, it_dev[c], (size_t)B * sizeof(int32_t)));
            ge = max_rel_err(gamma, g_want, (size_t)K * B, &zeros);
            se = max_rel_err(sstats, s_want, (size_t)K * V, &zeros);
            ge_worst = ge > ge_worst ? ge : ge_worst;
            se_worst = se > se_worst ? se : se_worst;
            its_ok = its_ok && memcmp(iters, iters_want, (size_t)B * sizeof(int32_t)) == 0;
            CHECK(trlda_dev_free(0, g_dev[c]));
            CHECK(trlda_dev_free(0, s_dev[c]));
            CHECK(trlda_dev_free(0, it_dev[c]));
        }
        CHECK(trlda_dev_free(0, g0_dev));
        printf("golden lanes gamma_err %.3e sstats_err %.3e zeros_agree %d iters_equal %d through %lld of %d\n",
               ge_worst, se_worst, zeros, its_ok, through, (int)CALLS);
    }
    CHECK(trlda_batch_destroy(batch));
    CHECK(trlda_model_destroy(model));
    free(indptr); free(ids); free(cnts); free(iters_want); free(iters); free(alpha); free(g_want);
    free(s_want); free(lambda); free(gamma0); free(gamma); free(sstats);
    return 0;
}

int main(int argc, char **argv)
{
    enum { K = 50, V = 3000, B = 120, N = 80 };       /* N unique words per document */
    if (trlda_device_count() < 1) {
        fprintf(stderr, "no HIP device\n");
        return 2;
    }
    if (argc > 1 && run_golden(argv[1]) != 0)
        return 3;
    int32_t *indptr = malloc((B + 1) * sizeof(int32_t));
    int32_t *ids = malloc((size_t)B * N * sizeof(int32_t));
    int32_t *cnts = malloc((size_t)B * N * sizeof(int32_t));
    double total = 0.0;
    unsigned s = 12345u;
    for (int d = 0; d < B; ++d) {
        indptr[d] = d * N;
        /* N distinct ids: a stride walk over the vocabulary from a per-document offset */
        const int off = (int)((s = s * 1664525u + 1013904223u) % V);
        for (int j = 0; j < N; ++j) {
            ids[d * N + j] = (off + j * 37) % V;
            cnts[d * N + j] = 1 + (int)((s = s * 1664525u + 1013904223u) >> 30);
            total += cnts[d * N + j];
        }
    }
    indptr[B] = B * N;

    double *lambda = malloc((size_t)K * V * sizeof(double));
    double *gamma = malloc((size_t)K * B * sizeof(double));
    double *sstats = malloc((size_t)K * V * sizeof(double));
    double alpha[K];
    int32_t iters[B];
    trlda_seed(7);
    trlda_sample_gamma_init(K, V, lambda);            /* lda.cpp:71 */
    trlda_sample_gamma_init(K, B, gamma);             /* lda.cpp:135 */
    for (int k = 0; k < K; ++k)
        alpha[k] = 0.1;

    /* ---- one shot, host pointers ---- */
    CHECK(trlda_estep(K, V, B, indptr, ids, cnts, lambda, alpha, gamma, sstats, 20, 1e-3, iters, 0));
    double ssum = 0.0, gsum = 0.0;
    for (size_t i = 0; i < (size_t)K * V; ++i)
        ssum += sstats[i];
    for (size_t i = 0; i < (size_t)K * B; ++i)
        gsum += gamma[i];
    int itmin = iters[0], itmax = iters[0];
    for (int d = 1; d < B; ++d) {
        itmin = iters[d] < itmin ? iters[d] : itmin;
        itmax = iters[d] > itmax ? iters[d] : itmax;
    }
    printf("estep counts %.17g sstats %.17g gamma %.17g expect_gamma %.17g iters %d %d\n", total, ssum, gsum,
           total + B * K * 0.1, itmin, itmax);

    /* ---- resident: model + batch, two online updates ---- */
    trlda_model *model = NULL;
    trlda_batch *batch = NULL;
    CHECK(trlda_model_create(&model, 0, K, V));
    CHECK(trlda_model_set_alpha(model, alpha));
    CHECK(trlda_model_set_lambda(model, lambda));
    CHECK(trlda_batch_create(&batch, 0, V, B, indptr, ids, cnts));
    const double eta = 0.3;
    const int D = 100000;
    int count = 0;
    double lsum_prev = 0.0;
    for (size_t i = 0; i < (size_t)K * V; ++i)
        lsum_prev += lambda[i];
    for (int call = 0; call < 2; ++call) {
        double rho = 0.0;
        CHECK(trlda_model_online_update(model, batch, D, eta, call == 0 ? 3 : 0, 20, 0.7, 100.0, -1.0, 1, 1,
                                        1e-3, &count, &rho, NULL));
        CHECK(trlda_model_get_lambda(model, lambda));
        double lsum = 0.0;
        for (size_t i = 0; i < (size_t)K * V; ++i)
            lsum += lambda[i];
        const double expect = (1.0 - rho) * lsum_prev + rho * ((double)K * V * eta + (double)D / B * total);
        printf("update %d rho %.17g lambda %.17g expect %.17g count %d\n", call, rho, lsum, expect, count);
        lsum_prev = lsum;
    }
    /* an argument error comes back as a code and a message, not as a crash */
    const int rc = trlda_model_online_update(model, NULL, D, eta, 0, 20, 0.7, 100.0, -1.0, 1, 1, 1e-3, &count,
                                             NULL, NULL);
    printf("null batch -> %d (%s)\n", rc, trlda_last_error());
    CHECK(trlda_batch_destroy(batch));
    CHECK(trlda_model_destroy(model));
    free(indptr); free(ids); free(cnts); free(lambda); free(gamma); free(sstats);
    return 0;
}
