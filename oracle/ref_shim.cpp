/*
 * oracle/ref_shim.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * extern "C" driver around the reference's UNMODIFIED C++ core.  It is compiled
 * together with /root/reference/code/trlda/src/{lda,onlinelda,batchlda,
 * cumulativelda,distribution,digamma,utils,zeta}.cpp (where they lie; nothing is
 * copied) into oracle/_ref/libtrlda_ref.so by oracle/Makefile.  The library is
 * used to (1) generate tests/golden/ and (2) pin oracle/cpu_ref.c, and may be
 * timed as bench.py's cpu_baseline (kind "reference").  It only exists where
 * /root/reference exists (this container); the .so travels to the GPU box.
 *
 * Array conventions match include/trlda_hip.h: column-major fp64, CSR int32.
 */
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "trlda/models"
#include "trlda/utils"

using Eigen::ArrayXd;
using Eigen::ArrayXXd;
using Eigen::Map;
using TRLDA::BatchLDA;
using TRLDA::CumulativeLDA;
using TRLDA::LDA;
using TRLDA::OnlineLDA;

static thread_local const char *g_last_error = "";

static LDA::Documents to_documents(int B, const int *indptr, const int *ids, const int *cnts)
{
    LDA::Documents docs(B);
    for (int d = 0; d < B; ++d) {
        docs[d].reserve(indptr[d + 1] - indptr[d]);
        for (int i = indptr[d]; i < indptr[d + 1]; ++i)
            docs[d].push_back(std::make_pair(ids[i], cnts[i]));
    }
    return docs;
}

extern "C" {

const char *ref_last_error() { return g_last_error; }

void ref_seed(unsigned int s) { srand(s); }

double ref_digamma(double x) { return TRLDA::digamma(x); }

double ref_polygamma(int n, double x) { return TRLDA::polygamma(n, x); }

void ref_sample_gamma(int m, int n, int k, double *out)
{
    ArrayXXd s = TRLDA::sampleGamma(m, n, k);
    std::memcpy(out, s.data(), sizeof(double) * (size_t)m * n);
}

/* ---- model handles ---------------------------------------------------- */

void *ref_online_create(int V, int K, int D, const double *alpha, double eta)
{
    try {
        ArrayXd a = Map<const ArrayXd>(alpha, K);
        return new OnlineLDA(V, D, a, eta);
    } catch (TRLDA::Exception &e) {
        g_last_error = e.message();
        return 0;
    }
}

void *ref_batch_create(int V, int K, const double *alpha, double eta)
{
    try {
        ArrayXd a = Map<const ArrayXd>(alpha, K);
        return new BatchLDA(V, a, eta);
    } catch (TRLDA::Exception &e) {
        g_last_error = e.message();
        return 0;
    }
}

void *ref_cumulative_create(int V, int K, const double *alpha, double eta)
{
    try {
        ArrayXd a = Map<const ArrayXd>(alpha, K);
        return new CumulativeLDA(V, a, eta);
    } catch (TRLDA::Exception &e) {
        g_last_error = e.message();
        return 0;
    }
}

void ref_model_destroy(void *h) { delete static_cast<LDA *>(h); }

void ref_model_get_lambda(void *h, double *out)
{
    ArrayXXd l = static_cast<LDA *>(h)->lambda();
    std::memcpy(out, l.data(), sizeof(double) * (size_t)l.size());
}

int ref_model_set_lambda(void *h, int K, int V, const double *in)
{
    try {
        static_cast<LDA *>(h)->setLambda(Map<const ArrayXXd>(in, K, V));
        return 0;
    } catch (TRLDA::Exception &e) {
        g_last_error = e.message();
        return -1;
    }
}

void ref_model_get_alpha(void *h, double *out)
{
    ArrayXd a = static_cast<LDA *>(h)->alpha();
    std::memcpy(out, a.data(), sizeof(double) * (size_t)a.size());
}

int ref_model_set_alpha(void *h, int K, const double *in)
{
    try {
        static_cast<LDA *>(h)->setAlpha(ArrayXd(Map<const ArrayXd>(in, K)));
        return 0;
    } catch (TRLDA::Exception &e) {
        g_last_error = e.message();
        return -1;
    }
}

double ref_model_get_eta(void *h) { return static_cast<LDA *>(h)->eta(); }

int ref_online_update_count(void *h) { return static_cast<OnlineLDA *>(h)->updateCount(); }

/* LDA::updateVariables(docs, latents, params) -- lda.cpp:142 -> :160-220.
 * gamma in/out (K x B); with use_latents == 0 the reference draws gamma0
 * itself from libc rand() (lda.cpp:119-138). */
int ref_model_estep(void *h, int B, const int *indptr, const int *ids, const int *cnts,
                    int use_latents, double *gamma, double *sstats, int max_iter,
                    double threshold)
{
    LDA *m = static_cast<LDA *>(h);
    try {
        LDA::Documents docs = to_documents(B, indptr, ids, cnts);
        LDA::Parameters p;
        p.maxIterInference = max_iter;
        p.threshold = threshold;
        std::pair<ArrayXXd, ArrayXXd> r;
        if (use_latents)
            r = m->updateVariables(docs, ArrayXXd(Map<const ArrayXXd>(gamma, m->numTopics(), B)), p);
        else
            r = m->updateVariables(docs, p);
        std::memcpy(gamma, r.first.data(), sizeof(double) * (size_t)r.first.size());
        std::memcpy(sstats, r.second.data(), sizeof(double) * (size_t)r.second.size());
        return 0;
    } catch (TRLDA::Exception &e) {
        g_last_error = e.message();
        return -1;
    }
}

/* LDA::lowerBound -- lda.cpp:297-360 (OnlineLDA overrides the default of num_documents,
 * onlinelda.cpp:184-191; is_online selects that virtual).  gamma0 is drawn from libc rand(). */
double ref_model_lower_bound(void *h, int is_online, int B, const int *indptr, const int *ids,
                             const int *cnts, int num_documents, int max_iter)
{
    try {
        LDA::Documents docs = to_documents(B, indptr, ids, cnts);
        LDA::Parameters p;
        p.maxIterInference = max_iter;
        if (is_online)
            return static_cast<OnlineLDA *>(h)->lowerBound(docs, p, num_documents);
        return static_cast<LDA *>(h)->lowerBound(docs, p, num_documents);
    } catch (TRLDA::Exception &e) {
        g_last_error = e.message();
        return 0.;
    }
}

/* OnlineLDA::updateParameters -- onlinelda.cpp:53-179, all kwargs of
 * onlineldainterface.cpp:204-256. */
double ref_online_update_parameters(void *h, int B, const int *indptr, const int *ids,
                                    const int *cnts, int max_iter_tr, int max_iter_inference,
                                    double kappa, double tau, double rho, int adaptive,
                                    int init_gamma, int update_lambda, int update_alpha,
                                    int update_eta, double min_alpha, double min_eta)
{
    OnlineLDA *m = static_cast<OnlineLDA *>(h);
    try {
        LDA::Documents docs = to_documents(B, indptr, ids, cnts);
        LDA::Parameters p;
        p.maxIterInference = 20;                       /* onlineldainterface.cpp:227 */
        p.maxIterTR = max_iter_tr;
        p.maxIterInference = max_iter_inference;
        p.kappa = kappa;
        p.tau = tau;
        p.rho = rho;
        p.adaptive = adaptive != 0;
        p.initGamma = init_gamma != 0;
        p.updateLambda = update_lambda != 0;
        p.updateAlpha = update_alpha != 0;
        p.updateEta = update_eta != 0;
        p.minAlpha = min_alpha;
        p.minEta = min_eta;
        return m->updateParameters(docs, p);
    } catch (TRLDA::Exception &e) {
        g_last_error = e.message();
        return -1.;
    }
}

/* BatchLDA::updateParameters -- batchlda.cpp:43-208 (lambda path :48-61). */
double ref_batch_update_parameters(void *h, int B, const int *indptr, const int *ids,
                                   const int *cnts, int max_epochs, int max_iter_inference,
                                   int update_lambda, int update_alpha, int update_eta)
{
    BatchLDA *m = static_cast<BatchLDA *>(h);
    try {
        LDA::Documents docs = to_documents(B, indptr, ids, cnts);
        LDA::Parameters p;
        p.maxEpochs = max_epochs;
        p.maxIterInference = max_iter_inference;
        p.updateLambda = update_lambda != 0;
        p.updateAlpha = update_alpha != 0;
        p.updateEta = update_eta != 0;
        return m->updateParameters(docs, p);
    } catch (TRLDA::Exception &e) {
        g_last_error = e.message();
        return -1.;
    }
}

/* BatchLDA::updateParameters with every kwarg of batchldainterface.cpp:126-172 */
double ref_batch_update_parameters_full(void *h, int B, const int *indptr, const int *ids,
                                        const int *cnts, int max_epochs, int max_iter_inference,
                                        int max_iter_alpha, int max_iter_eta, int update_lambda,
                                        int update_alpha, int update_eta, double min_alpha,
                                        double min_eta, double emp_bayes_threshold)
{
    BatchLDA *m = static_cast<BatchLDA *>(h);
    try {
        LDA::Documents docs = to_documents(B, indptr, ids, cnts);
        LDA::Parameters p;
        p.maxEpochs = max_epochs;
        p.maxIterInference = max_iter_inference;
        p.maxIterAlpha = max_iter_alpha;
        p.maxIterEta = max_iter_eta;
        p.updateLambda = update_lambda != 0;
        p.updateAlpha = update_alpha != 0;
        p.updateEta = update_eta != 0;
        p.minAlpha = min_alpha;
        p.minEta = min_eta;
        p.empBayesThreshold = emp_bayes_threshold;
        return m->updateParameters(docs, p);
    } catch (TRLDA::Exception &e) {
        g_last_error = e.message();
        return -1.;
    }
}

/* CumulativeLDA::updateParameters -- cumulativelda.cpp:49-153, kwargs of
 * cumulativeldainterface.cpp:115-160 */
double ref_cumulative_update_parameters(void *h, int B, const int *indptr, const int *ids,
                                        const int *cnts, int max_epochs, int max_iter_inference,
                                        int max_iter_alpha, int update_lambda, int update_alpha,
                                        double min_alpha, double emp_bayes_threshold,
                                        double inference_threshold)
{
    CumulativeLDA *m = static_cast<CumulativeLDA *>(h);
    try {
        LDA::Documents docs = to_documents(B, indptr, ids, cnts);
        LDA::Parameters p;
        p.maxEpochs = max_epochs;
        p.maxIterInference = max_iter_inference;
        p.maxIterAlpha = max_iter_alpha;
        p.updateLambda = update_lambda != 0;
        p.updateAlpha = update_alpha != 0;
        p.minAlpha = min_alpha;
        p.empBayesThreshold = emp_bayes_threshold;
        p.threshold = inference_threshold;
        return m->updateParameters(docs, p);
    } catch (TRLDA::Exception &e) {
        g_last_error = e.message();
        return -1.;
    }
}

} /* extern "C" */
