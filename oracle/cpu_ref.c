/*
 * oracle/cpu_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C (no Eigen, no OpenMP in the scalar entry points) restatement of the
 * one hot path of lucastheis/trlda that this repository accelerates: the
 * per-document variational E-step and the lambda M-step around it.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (trlda_amd -> libtrlda_hip.so) never does.
 *
 * Parity status: PINNED.  Every function below is checked in
 * tests/test_oracle.py against (a) the reference's own known-answer values
 * (utils_test.py:33-51, the digamma rows), (b) golden vectors produced by the
 * reference's unmodified C++ core compiled into oracle/_ref/ (recipe:
 * oracle/Makefile, driver: oracle/ref_shim.cpp, generator:
 * tests/golden/make_golden.py) and (c) M. Hoffman's onlineldavb.py imported
 * from /root/reference at fixture-generation time.
 *
 * All matrices are column-major fp64 (Eigen default, pyutils.cpp:22-26):
 *   lambda, sstats : K x V, element (k, w) at [k + K*w]
 *   gamma          : K x B, element (k, d) at [k + K*d]
 * Documents are CSR: indptr[B+1], ids[nnz], cnts[nnz] (int32), which is the
 * flat form of vector<vector<pair<int,int>>> (include/lda.h:21-23).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_EULER 0.577215664901532860606512090082402431

/* ------------------------------------------------------------------------
 * psi(x).  Follows src/digamma.cpp:116-178 (Cephes psi): reflection for
 * x <= 0, exact harmonic sum for integer x <= 10, upward recurrence to
 * s >= 10, then the 7-coefficient asymptotic series in 1/s^2.
 * ---------------------------------------------------------------------- */
static const double kPsiSeries[7] = {
     8.33333333333333333333E-2, -2.10927960927960927961E-2,
     7.57575757575757575758E-3, -4.16666666666666666667E-3,
     3.96825396825396825397E-3, -8.33333333333333333333E-3,
     8.33333333333333333333E-2,
};

double oracle_digamma(double x)
{
    double reflect = 0.0;
    int reflected = 0;

    if (x <= 0.0) {                                   /* digamma.cpp:123-144 */
        const double pi = 3.141592653589793238462643383279502884;
        double fl = floor(x);
        if (fl == x)
            return INFINITY;
        double frac = x - fl;
        if (frac != 0.5) {
            if (frac > 0.5) {
                fl += 1.0;
                frac = x - fl;
            }
            reflect = pi / tan(pi * frac);
        }
        reflected = 1;
        x = 1.0 - x;
    }

    double y;
    if (x <= 10.0 && x == floor(x)) {                 /* digamma.cpp:147-156 */
        int n = (int)x;
        y = 0.0;
        for (int i = 1; i < n; ++i)
            y += 1.0 / (double)i;
        y -= ORACLE_EULER;
    } else {
        double s = x, w = 0.0;                         /* digamma.cpp:158-163 */
        while (s < 10.0) {
            w += 1.0 / s;
            s += 1.0;
        }
        if (s < 1.0e17) {                              /* digamma.cpp:165-169 */
            double z = 1.0 / (s * s);
            double p = kPsiSeries[0];
            for (int i = 1; i <= 6; ++i)               /* polevl, :96-110 */
                p = p * z + kPsiSeries[i];
            y = z * p;
        } else {
            y = 0.0;
        }
        y = log(s) - (0.5 / s) - y - w;                /* digamma.cpp:171 */
    }
    if (reflected)
        y -= reflect;
    return y;
}

void oracle_digamma_array(const double *x, double *out, int64_t n)
{
    for (int64_t i = 0; i < n; ++i)
        out[i] = oracle_digamma(x[i]);
}

/* ------------------------------------------------------------------------
 * sampleGamma(m, n, k): utils.cpp:224-231.  k passes; each pass draws an
 * m x n matrix of U(-1,1) column-major from libc rand()
 * (Eigen/src/Core/MathFunctions.h:439-446) and subtracts log|u|.
 * ---------------------------------------------------------------------- */
void oracle_seed(unsigned int s) { srand(s); }        /* module.cpp:332-342 */

void oracle_sample_gamma(int m, int n, int k, double *out)
{
    int64_t total = (int64_t)m * n;
    for (int64_t i = 0; i < total; ++i)
        out[i] = 0.0;
    for (int pass = 0; pass < k; ++pass)
        for (int64_t i = 0; i < total; ++i) {
            double u = -1.0 + 2.0 * (double)rand() / (double)RAND_MAX;
            out[i] -= log(fabs(u));
        }
}

/* gamma0 / lambda0 as drawn by lda.cpp:71 and lda.cpp:135 */
void oracle_sample_gamma_init(int m, int n, double *out)
{
    oracle_sample_gamma(m, n, 100, out);
    int64_t total = (int64_t)m * n;
    for (int64_t i = 0; i < total; ++i)
        out[i] /= 100.;
}

/* ------------------------------------------------------------------------
 * exp(E[log beta]) preamble: lda.cpp:172-173.
 * ---------------------------------------------------------------------- */
void oracle_exp_elog_beta(int K, int V, const double *lambda, double *out)
{
    double *psi_sum = (double *)malloc(sizeof(double) * (size_t)K);
    for (int k = 0; k < K; ++k) {
        double s = 0.0;
        for (int w = 0; w < V; ++w)
            s += lambda[k + (int64_t)K * w];
        psi_sum[k] = oracle_digamma(s);
    }
    for (int w = 0; w < V; ++w)
        for (int k = 0; k < K; ++k) {
            int64_t i = k + (int64_t)K * w;
            out[i] = exp(oracle_digamma(lambda[i]) - psi_sum[k]);
        }
    free(psi_sum);
}

/* One document of lda.cpp:176-214.  beta_d is a caller-provided K*n scratch.
 * Writes the per-word weights cnt_j/phinorm_j into tw[n] and returns the
 * number of fixed-point iterations executed. */
static int estep_one_doc(int K, int n, const int32_t *ids, const int32_t *cnts,
                         const double *exp_elog_beta, const double *alpha,
                         double *gamma_d, double *epg /* exp(psi(gamma_d)) */,
                         double *beta_d, double *phinorm, double *last,
                         int max_iter, double threshold)
{
    for (int j = 0; j < n; ++j)                         /* lda.cpp:179-181 */
        memcpy(beta_d + (size_t)K * j, exp_elog_beta + (int64_t)K * ids[j],
               sizeof(double) * (size_t)K);

    for (int j = 0; j < n; ++j) {                       /* lda.cpp:183 */
        double s = 0.0;
        for (int k = 0; k < K; ++k)
            s += epg[k] * beta_d[k + (size_t)K * j];
        phinorm[j] = s + 1e-100;
    }

    int it = 0;
    for (; it < max_iter; ) {                           /* lda.cpp:185-204 */
        memcpy(last, gamma_d, sizeof(double) * (size_t)K);
        for (int k = 0; k < K; ++k)
            gamma_d[k] = 0.0;
        for (int j = 0; j < n; ++j) {                   /* :190-193 */
            double c = (double)cnts[j] / phinorm[j];
            const double *col = beta_d + (size_t)K * j;
            for (int k = 0; k < K; ++k)
                gamma_d[k] += c * col[k];
        }
        for (int k = 0; k < K; ++k) {                   /* :194-197 */
            gamma_d[k] *= epg[k];
            gamma_d[k] += alpha[k];
            epg[k] = exp(oracle_digamma(gamma_d[k]));
        }
        for (int j = 0; j < n; ++j) {                   /* :199 */
            double s = 0.0;
            for (int k = 0; k < K; ++k)
                s += epg[k] * beta_d[k + (size_t)K * j];
            phinorm[j] = s + 1e-100;
        }
        ++it;
        double change = 0.0;                            /* :202-203 */
        for (int k = 0; k < K; ++k)
            change += fabs(last[k] - gamma_d[k]);
        if (change / (double)K < threshold)
            break;
    }
    return it;
}

/*
 * LDA::updateVariablesVI, lda.cpp:160-220.
 * gamma: in = initial gamma (K x B), out = converged gamma.
 * sstats: out (K x V).  iters_out: optional per-document iteration counts.
 * Returns 0, or -1 for an out-of-range word id (the reference has UB there).
 */
int oracle_estep(int K, int V, int B, const int32_t *indptr, const int32_t *ids,
                 const int32_t *cnts, const double *lambda, const double *alpha,
                 double *gamma, double *sstats, int max_iter, double threshold,
                 int32_t *iters_out)
{
    int64_t nnz = indptr[B];
    for (int64_t i = 0; i < nnz; ++i)
        if (ids[i] < 0 || ids[i] >= V)
            return -1;

    int max_n = 0;
    for (int d = 0; d < B; ++d)
        if (indptr[d + 1] - indptr[d] > max_n)
            max_n = indptr[d + 1] - indptr[d];

    double *eeb = (double *)malloc(sizeof(double) * (size_t)K * V);
    double *epg = (double *)malloc(sizeof(double) * (size_t)K);
    double *last = (double *)malloc(sizeof(double) * (size_t)K);
    double *beta_d = (double *)malloc(sizeof(double) * (size_t)K * (max_n + 1));
    double *phinorm = (double *)malloc(sizeof(double) * (size_t)(max_n + 1));

    memset(sstats, 0, sizeof(double) * (size_t)K * V);  /* lda.cpp:169 */
    oracle_exp_elog_beta(K, V, lambda, eeb);            /* lda.cpp:172-173 */

    for (int d = 0; d < B; ++d) {
        int n = indptr[d + 1] - indptr[d];
        const int32_t *dids = ids + indptr[d];
        const int32_t *dcnt = cnts + indptr[d];
        double *g = gamma + (int64_t)K * d;
        for (int k = 0; k < K; ++k)                     /* lda.cpp:174 */
            epg[k] = exp(oracle_digamma(g[k]));
        int it = estep_one_doc(K, n, dids, dcnt, eeb, alpha, g, epg, beta_d,
                               phinorm, last, max_iter, threshold);
        if (iters_out)
            iters_out[d] = it;
        for (int j = 0; j < n; ++j) {                   /* lda.cpp:207-213 */
            double c = (double)dcnt[j] / phinorm[j];
            double *col = sstats + (int64_t)K * dids[j];
            for (int k = 0; k < K; ++k)
                col[k] += c * epg[k];
        }
    }
    for (int64_t i = 0; i < (int64_t)K * V; ++i)       /* lda.cpp:217 */
        sstats[i] *= eeb[i];

    free(eeb); free(epg); free(last); free(beta_d); free(phinorm);
    return 0;
}

/*
 * Same computation, documents spread over `nthreads` OpenMP threads without the
 * reference's critical section (lda.cpp:211): phase 1 runs the per-document fixed points
 * in parallel and keeps exp(psi(gamma_d)) and the per-word weights cnt/phinorm; phase 2
 * sums every word's contributions in document order (a counting sort by word id gives the
 * lists), so the result is identical to the serial function for any thread count.
 * Only used for the "all host cores" CPU baseline line of bench.py.
 */
int oracle_estep_mt(int K, int V, int B, const int32_t *indptr, const int32_t *ids,
                    const int32_t *cnts, const double *lambda, const double *alpha,
                    double *gamma, double *sstats, int max_iter, double threshold,
                    int32_t *iters_out, int nthreads)
{
    int64_t nnz = indptr[B];
    for (int64_t i = 0; i < nnz; ++i)
        if (ids[i] < 0 || ids[i] >= V)
            return -1;
    int max_n = 0;
    for (int d = 0; d < B; ++d)
        if (indptr[d + 1] - indptr[d] > max_n)
            max_n = indptr[d + 1] - indptr[d];
    if (nthreads < 1)
        nthreads = 1;

    size_t KV = (size_t)K * V;
    double *eeb = (double *)malloc(sizeof(double) * KV);
    double *psi_sum = (double *)malloc(sizeof(double) * (size_t)K);
    double *epg_all = (double *)malloc(sizeof(double) * (size_t)K * (B > 0 ? B : 1));
    double *tw = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));

    for (int k = 0; k < K; ++k) {
        double s = 0.0;
        for (int w = 0; w < V; ++w)
            s += lambda[k + (int64_t)K * w];
        psi_sum[k] = oracle_digamma(s);
    }
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int w = 0; w < V; ++w)
        for (int k = 0; k < K; ++k) {
            int64_t i = k + (int64_t)K * w;
            eeb[i] = exp(oracle_digamma(lambda[i]) - psi_sum[k]);
        }

#pragma omp parallel num_threads(nthreads)
    {
        double *last = (double *)malloc(sizeof(double) * (size_t)K);
        double *beta_d = (double *)malloc(sizeof(double) * (size_t)K * (max_n + 1));
        double *phinorm = (double *)malloc(sizeof(double) * (size_t)(max_n + 1));
#pragma omp for schedule(dynamic, 4)
        for (int d = 0; d < B; ++d) {
            int n = indptr[d + 1] - indptr[d];
            const int32_t *dids = ids + indptr[d];
            const int32_t *dcnt = cnts + indptr[d];
            double *g = gamma + (int64_t)K * d;
            double *epg = epg_all + (int64_t)K * d;
            for (int k = 0; k < K; ++k)
                epg[k] = exp(oracle_digamma(g[k]));
            int it = estep_one_doc(K, n, dids, dcnt, eeb, alpha, g, epg, beta_d, phinorm, last,
                                   max_iter, threshold);
            if (iters_out)
                iters_out[d] = it;
            for (int j = 0; j < n; ++j)
                tw[indptr[d] + j] = (double)dcnt[j] / phinorm[j];
        }
        free(last); free(beta_d); free(phinorm);
    }

    /* word-major lists (stable counting sort: document order within a word) */
    int32_t *wptr = (int32_t *)calloc((size_t)V + 1, sizeof(int32_t));
    int32_t *wpos = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
    int32_t *wdoc = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
    for (int64_t i = 0; i < nnz; ++i)
        ++wptr[ids[i] + 1];
    for (int w = 0; w < V; ++w)
        wptr[w + 1] += wptr[w];
    {
        int32_t *cur = (int32_t *)malloc(sizeof(int32_t) * (size_t)V);
        memcpy(cur, wptr, sizeof(int32_t) * (size_t)V);
        for (int d = 0; d < B; ++d)
            for (int p = indptr[d]; p < indptr[d + 1]; ++p) {
                int32_t q = cur[ids[p]]++;
                wpos[q] = p;
                wdoc[q] = d;
            }
        free(cur);
    }
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 64)
    for (int w = 0; w < V; ++w) {
        double *col = sstats + (int64_t)K * w;
        for (int k = 0; k < K; ++k)
            col[k] = 0.0;
        for (int q = wptr[w]; q < wptr[w + 1]; ++q) {
            double c = tw[wpos[q]];
            const double *epg = epg_all + (int64_t)K * wdoc[q];
            for (int k = 0; k < K; ++k)
                col[k] += c * epg[k];
        }
        for (int k = 0; k < K; ++k)
            col[k] *= eeb[k + (int64_t)K * w];
    }
    free(eeb); free(psi_sum); free(epg_all); free(tw); free(wptr); free(wpos); free(wdoc);
    return 0;
}

/* ------------------------------------------------------------------------
 * M-step pieces of OnlineLDA::updateParameters (onlinelda.cpp:53-111).
 * ---------------------------------------------------------------------- */

/* onlinelda.cpp:59-66 (non-adaptive branch) */
double oracle_learning_rate(double rho, double tau, double kappa, int update_count)
{
    if (rho < 0.)
        rho = pow(tau + (double)update_count, -kappa);
    return rho;
}

/* onlinelda.cpp:79-86: lambda = (1-rho) lambda' (+row) rho (eta + D/B/K * wordcounts) */
void oracle_tr_init(int K, int V, int B, int num_documents, double rho, double eta,
                    const int32_t *indptr, const int32_t *ids, const int32_t *cnts,
                    const double *lambda_prime, double *lambda_out)
{
    double *wc = (double *)calloc((size_t)V, sizeof(double));
    for (int64_t i = 0; i < indptr[B]; ++i)
        wc[ids[i]] += (double)cnts[i];
    for (int w = 0; w < V; ++w) {
        double add = rho * (eta + (double)num_documents / (double)B / (double)K * wc[w]);
        for (int k = 0; k < K; ++k) {
            int64_t i = k + (int64_t)K * w;
            lambda_out[i] = (1. - rho) * lambda_prime[i] + add;
        }
    }
    free(wc);
}

/* onlinelda.cpp:99-100 / :108-109: lambda = (1-rho) lambda' + rho (eta + D/B sstats) */
void oracle_mstep_blend(int K, int V, double rho, double eta, double scale,
                        const double *lambda_prime, const double *sstats,
                        double *lambda_out)
{
    for (int64_t i = 0; i < (int64_t)K * V; ++i) {
        double hat = eta + scale * sstats[i];
        lambda_out[i] = (1. - rho) * lambda_prime[i] + rho * hat;
    }
}

/*
 * OnlineLDA::updateParameters, lambda path only (onlinelda.cpp:53-111,
 * 177-179), default-gamma-init drawn from libc rand() exactly where the
 * reference draws it (lda.cpp:135).  lambda is updated in place; gamma_out
 * (K x B, optional) receives the last E-step's gamma.
 * Returns rho; *update_count is incremented unless the batch is empty.
 */
double oracle_online_update_parameters(
    int K, int V, int B, int num_documents, const int32_t *indptr,
    const int32_t *ids, const int32_t *cnts, double *lambda, const double *alpha,
    double eta, int max_iter_tr, int max_iter_inference, double kappa, double tau,
    double rho_in, int init_gamma, int update_lambda, double threshold,
    int *update_count, double *gamma_out)
{
    if (B == 0)
        return 1.0;                                      /* onlinelda.cpp:54-56 */

    double rho = oracle_learning_rate(rho_in, tau, kappa, *update_count);

    if (update_lambda) {
        size_t KV = (size_t)K * V;
        double *lambda_prime = (double *)malloc(sizeof(double) * KV);
        double *sstats = (double *)malloc(sizeof(double) * KV);
        double *gamma = (double *)malloc(sizeof(double) * (size_t)K * B);
        memcpy(lambda_prime, lambda, sizeof(double) * KV);
        double scale = (double)num_documents / (double)B;

        if (max_iter_tr > 0) {
            oracle_tr_init(K, V, B, num_documents, rho, eta, indptr, ids, cnts,
                           lambda_prime, lambda);
            for (int i = 0; i < max_iter_tr; ++i) {
                if (!(i > 0 && init_gamma))              /* onlinelda.cpp:91-95 */
                    oracle_sample_gamma_init(K, B, gamma);
                oracle_estep(K, V, B, indptr, ids, cnts, lambda, alpha, gamma,
                             sstats, max_iter_inference, threshold, NULL);
                oracle_mstep_blend(K, V, rho, eta, scale, lambda_prime, sstats, lambda);
            }
        } else {
            oracle_sample_gamma_init(K, B, gamma);
            oracle_estep(K, V, B, indptr, ids, cnts, lambda, alpha, gamma, sstats,
                         max_iter_inference, threshold, NULL);
            oracle_mstep_blend(K, V, rho, eta, scale, lambda_prime, sstats, lambda);
        }
        if (gamma_out)
            memcpy(gamma_out, gamma, sizeof(double) * (size_t)K * B);
        free(lambda_prime); free(sstats); free(gamma);
    }
    ++*update_count;                                     /* onlinelda.cpp:177 */
    return rho;
}

/*
 * BatchLDA::updateParameters, lambda path only (batchlda.cpp:43-61): per epoch a
 * full E-step from a fresh random gamma, then lambda = eta + sstats.
 */
double oracle_batch_update_parameters(
    int K, int V, int B, const int32_t *indptr, const int32_t *ids,
    const int32_t *cnts, double *lambda, const double *alpha, double eta,
    int max_epochs, int max_iter_inference, int update_lambda, double threshold,
    double *gamma_out)
{
    if (B == 0)
        return 1.;                                       /* batchlda.cpp:44-46 */
    size_t KV = (size_t)K * V;
    double *sstats = (double *)malloc(sizeof(double) * KV);
    double *gamma = (double *)malloc(sizeof(double) * (size_t)K * B);
    for (int epoch = 0; epoch < max_epochs; ++epoch) {
        if (update_lambda) {
            oracle_sample_gamma_init(K, B, gamma);
            oracle_estep(K, V, B, indptr, ids, cnts, lambda, alpha, gamma, sstats,
                         max_iter_inference, threshold, NULL);
            for (size_t i = 0; i < KV; ++i)              /* batchlda.cpp:60 */
                lambda[i] = eta + sstats[i];
        }
    }
    if (gamma_out)
        memcpy(gamma_out, gamma, sizeof(double) * (size_t)K * B);
    free(sstats); free(gamma);
    return 1.;
}

/* ------------------------------------------------------------------------
 * LDA::lowerBound given the E-step's outputs -- src/lda.cpp:304-360 (the
 * E-step itself, :309, is oracle_estep).  factor = numDocuments / B (:302-303).
 *
 * reference_indexing != 0 reproduces what the reference computes when built
 * the way its setup.py builds it (distutils passes -DNDEBUG, so Eigen's bounds
 * assertion is compiled out): lda.cpp:334 reads `psiLambda.row(id)` of the
 * K x V matrix where column id is meant, i.e. element c of "the row" is the
 * flat element id + c*K.  That mode exists to pin this restatement against
 * the compiled reference; reference_indexing == 0 is the intended formula
 * (phi_kj proportional to exp(E[log beta_k,w_j] + psi(gamma_k))) and is what
 * the product implements.  The two differ by about 1e-4 relative.
 * ---------------------------------------------------------------------- */
double oracle_lower_bound(int K, int V, int B, const int32_t *indptr, const int32_t *ids,
                          const int32_t *cnts, const double *lambda, const double *alpha,
                          double eta, const double *gamma, const double *sstats, double factor,
                          int reference_indexing)
{
    const size_t KV = (size_t)K * V;
    double *psi_lambda = (double *)malloc(sizeof(double) * KV);
    double *lambda_sum = (double *)calloc((size_t)K, sizeof(double));
    double *psi_lambda_sum = (double *)malloc(sizeof(double) * (size_t)K);
    double *psi_gamma = (double *)malloc(sizeof(double) * (size_t)K);
    double *phi = (double *)malloc(sizeof(double) * (size_t)K);
    for (size_t i = 0; i < KV; ++i) {                          /* :312-314 */
        psi_lambda[i] = oracle_digamma(lambda[i]);
        lambda_sum[i % K] += lambda[i];
    }
    for (int k = 0; k < K; ++k)
        psi_lambda_sum[k] = oracle_digamma(lambda_sum[k]);

    double pw_pb = 0.0;                                        /* :317 */
    for (size_t i = 0; i < KV; ++i)
        pw_pb += (eta + factor * sstats[i] - lambda[i]) * (psi_lambda[i] - psi_lambda_sum[i % K]);

    double pz = 0.0, ptheta = 0.0;
    for (int d = 0; d < B; ++d) {                              /* :325-351 */
        const double *g = gamma + (size_t)d * K;
        double gamma_sum = 0.0;
        for (int k = 0; k < K; ++k) {
            gamma_sum += g[k];
            psi_gamma[k] = oracle_digamma(g[k]);
        }
        const double psi_gamma_sum = oracle_digamma(gamma_sum);
        for (int p = indptr[d]; p < indptr[d + 1]; ++p) {
            const int id = ids[p];
            double mx = -HUGE_VAL;
            for (int k = 0; k < K; ++k) {                      /* :332-337 */
                const double pl = reference_indexing ? psi_lambda[(size_t)id + (size_t)k * K]
                                                     : psi_lambda[(size_t)id * K + k];
                phi[k] = pl - psi_lambda_sum[k] + psi_gamma[k];
                if (phi[k] > mx)
                    mx = phi[k];
            }
            double se = 0.0;                                   /* logSumExp, :338 */
            for (int k = 0; k < K; ++k)
                se += exp(phi[k] - mx);
            const double lse = mx + log(se);
            double tmp = 0.0;                                  /* :343-344 */
            for (int k = 0; k < K; ++k) {
                const double lp = phi[k] - lse, ph = exp(lp);
                tmp += (psi_gamma[k] - psi_gamma_sum) * ph - ph * lp;
            }
            pz += (double)cnts[p] * tmp;                       /* :347 */
        }
        for (int k = 0; k < K; ++k)                            /* :349-351 */
            ptheta += (alpha[k] - g[k]) * (psi_gamma[k] - psi_gamma_sum) + lgamma(g[k]);
        ptheta -= lgamma(gamma_sum);
    }
    double alpha_sum = 0.0, lg_alpha = 0.0, lg_lambda_sum = 0.0, lg_lambda = 0.0;
    for (int k = 0; k < K; ++k) {
        alpha_sum += alpha[k];
        lg_alpha += lgamma(alpha[k]);
        lg_lambda_sum += lgamma(lambda_sum[k]);
    }
    for (size_t i = 0; i < KV; ++i)
        lg_lambda += lgamma(lambda[i]);
    ptheta += (lgamma(alpha_sum) - lg_alpha) * B;              /* :355 */
    pw_pb += K * lgamma(V * eta) - lg_lambda_sum;              /* :356 */
    pw_pb -= (double)K * V * lgamma(eta) - lg_lambda;          /* :357 */
    free(psi_lambda); free(lambda_sum); free(psi_lambda_sum); free(psi_gamma); free(phi);
    return pw_pb + factor * pz + factor * ptheta;              /* :359 */
}
