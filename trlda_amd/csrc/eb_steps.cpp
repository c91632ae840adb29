// eb_steps.cpp -- the K- and scalar-sized host arithmetic of the empirical-Bayes updates (host
// only; see host_common.h): the Newton / natural-gradient steps on alpha and eta of
// OnlineLDA::updateParameters (reference src/onlinelda.cpp:116-162) and the step-halving line
// searches of BatchLDA / CumulativeLDA (src/batchlda.cpp:64-205, src/cumulativelda.cpp:76-150).
// The sums over gamma and lambda they start from come from the device (csrc/eb_kernels.h).
//
// psi and psi' here are series of this library's own (upward recurrence to an argument where the
// asymptotic expansion is exact to the last bit), within 1e-15 of the reference's digamma
// (src/digamma.cpp) and polygamma(1, .) = zeta(2, .) (src/utils.cpp:107-111, src/zeta.cpp);
// log Gamma is libm's, as in the reference (src/utils.cpp:75-91).
#include <cmath>
#include <cstdio>
#include <vector>

#include "../../include/trlda_hip.h"
#include "host_common.h"

using trlda_host::fail;

namespace {

// psi(x), x > 0: psi(x) = psi(x + m) - sum_{i<m} 1/(x + i) up to s = x + m >= 10, then
// log s - 1/(2s) - sum_k B_2k / (2k s^2k)
double psi(double x)
{
    double s = x, w = 0.0;
    while (s < 10.0) {
        w += 1.0 / s;
        s += 1.0;
    }
    const double z = 1.0 / (s * s);
    double p = 8.33333333333333333333E-2;
    p = p * z - 2.10927960927960927961E-2;
    p = p * z + 7.57575757575757575758E-3;
    p = p * z - 4.16666666666666666667E-3;
    p = p * z + 3.96825396825396825397E-3;
    p = p * z - 8.33333333333333333333E-3;
    p = p * z + 8.33333333333333333333E-2;
    return std::log(s) - 0.5 / s - z * p - w;
}

// psi'(x) = zeta(2, x), x > 0: sum_{i<m} 1/(x + i)^2 up to s = x + m >= 20, then
// 1/s + 1/(2 s^2) + sum_k B_2k / s^(2k+1)
double psi1(double x)
{
    double s = x, w = 0.0;
    while (s < 20.0) {
        w += 1.0 / (s * s);
        s += 1.0;
    }
    const double z = 1.0 / (s * s);
    const double series = z * (1. / 6 + z * (-1. / 30 + z * (1. / 42 + z * (-1. / 30 + z * (5. / 66 + z * (
                                   -691. / 2730 + z * (7. / 6)))))));
    return w + (1.0 + 0.5 / s + series) / s;
}

double lngamma(double x)
{
    int sign = 0;
    return lgamma_r(x, &sign);
}

// L(alpha) = D (lgamma(sum alpha) - sum lgamma(alpha)) + sum_k psi_gamma_diff_k (alpha_k - 1)
double alpha_bound(int K, const double *alpha, const double *pgd, double D)
{
    double sum = 0.0, lg = 0.0, lin = 0.0;
    for (int k = 0; k < K; ++k) {
        sum += alpha[k];
        lg += lngamma(alpha[k]);
        lin += pgd[k] * (alpha[k] - 1.0);
    }
    return D * (lngamma(sum) - lg) + lin;
}

// gradient g, diagonal of the Hessian h and the constant c of the rank-one correction
// (onlinelda.cpp:123-138 == batchlda.cpp:86-98)
void alpha_newton_terms(int K, const double *alpha, const double *pgd, double D, std::vector<double> &g,
                        std::vector<double> &h, double *c)
{
    double sum = 0.0;
    for (int k = 0; k < K; ++k)
        sum += alpha[k];
    const double psi_sum = psi(sum), z = D * psi1(sum);
    double num = 0.0, den = 1.0 / z;
    for (int k = 0; k < K; ++k) {
        g[(size_t)k] = pgd[k] - D * (psi(alpha[k]) - psi_sum);
        h[(size_t)k] = -D * psi1(alpha[k]);
        num += g[(size_t)k] / h[(size_t)k];
        den += 1.0 / h[(size_t)k];
    }
    *c = num / den;
}

}  // namespace

extern "C" {

int trlda_eb_online_alpha_step(int K, const double *alpha, const double *psi_gamma_diff, double num_docs,
                               double rho, double min_alpha, double *alpha_out)
{
    if (K <= 0 || !alpha || !psi_gamma_diff || !alpha_out)
        return fail(TRLDA_ERR_ARG, "bad alpha step arguments");
    std::vector<double> g((size_t)K), h((size_t)K);
    double c = 0.0;
    alpha_newton_terms(K, alpha, psi_gamma_diff, num_docs, g, h, &c);
    for (int k = 0; k < K; ++k) {
        const double a = alpha[k] - rho * (g[(size_t)k] - c) / h[(size_t)k];
        alpha_out[k] = a < min_alpha ? min_alpha : a;              // onlinelda.cpp:140-141
    }
    return TRLDA_OK;
}

double trlda_eb_online_eta_step(double eta, double sum_psi_lambda, const double *rowsums, int K, int V,
                                double rho, double min_eta)
{
    double psi_rows = 0.0;
    for (int k = 0; k < K; ++k)
        psi_rows += psi(rowsums[k]);
    const double KV = (double)K * (double)V;
    const double g = sum_psi_lambda - V * psi_rows - KV * (psi(eta) - psi(V * eta));
    const double h = KV * (psi1(V * eta) - psi1(eta));
    const double e = eta - rho * g / h;                              // onlinelda.cpp:158-161
    return e < min_eta ? min_eta : e;
}

// `verbosity > 1` of the reference's line searches (src/batchlda.cpp:78-88,120-123,155-165,184-187,
// src/cumulativelda.cpp:87-97,129-132): progress on stdout in the reference's own words and
// std::cout's default format (six significant digits).  Per thread; the Python classes set it around
// the call from their `verbosity` argument.
static thread_local int g_verbosity = 0;
void trlda_eb_set_verbosity(int verbosity) { g_verbosity = verbosity; }
static void say(const char *what, double v)
{
    std::printf("\t%s: %g\n", what, v);
}

int trlda_eb_alpha_line_search(int K, const double *alpha, const double *psi_gamma_diff, double num_docs,
                               int max_iter_alpha, double min_alpha, double threshold, double *alpha_out)
{
    if (K <= 0 || !alpha || !psi_gamma_diff || !alpha_out)
        return fail(TRLDA_ERR_ARG, "bad alpha line search arguments");
    std::vector<double> cur(alpha, alpha + K), cand((size_t)K), g((size_t)K), h((size_t)K);
    const bool loud = g_verbosity > 1;
    if (loud)
        std::printf("Optimizing alpha...\n");
    double L = alpha_bound(K, cur.data(), psi_gamma_diff, num_docs), Lprime = L;
    for (int it = 0; it < max_iter_alpha; ++it) {                    // batchlda.cpp:81-141
        if (loud)
            say("Current function value", L);
        double c = 0.0;
        alpha_newton_terms(K, cur.data(), psi_gamma_diff, num_docs, g, h, &c);
        double rho = .2;
        for (int j = 0; j < 20; ++j) {
            bool below = false;
            for (int k = 0; k < K; ++k) {
                cand[(size_t)k] = cur[(size_t)k] - rho * (g[(size_t)k] - c) / h[(size_t)k];
                below = below || cand[(size_t)k] < min_alpha;
            }
            if (below) {
                rho /= 2.;
                continue;
            }
            Lprime = alpha_bound(K, cand.data(), psi_gamma_diff, num_docs);
            if (L <= Lprime) {
                if (loud) {
                    double g2 = 0.0;
                    for (int k = 0; k < K; ++k)
                        g2 += g[(size_t)k] * g[(size_t)k];
                    say("Step width", rho);
                    say("Gradient magnitude", std::sqrt(g2));
                }
                cur = cand;
                break;
            }
            rho /= 2.;
        }
        if (Lprime - L < threshold)
            break;
        L = Lprime;
    }
    for (int k = 0; k < K; ++k)
        alpha_out[k] = cur[(size_t)k];
    if (loud)
        std::fflush(stdout);
    return TRLDA_OK;
}

double trlda_eb_eta_line_search(double eta, double sum_psi_lambda, const double *rowsums, int K, int V,
                                int max_iter_eta, double min_eta, double threshold)
{
    double psi_rows = 0.0;
    for (int k = 0; k < K; ++k)
        psi_rows += psi(rowsums[k]);
    const double KV = (double)K * (double)V;
    const double c = sum_psi_lambda - V * psi_rows;
    auto bound = [&](double e) { return (e - 1) * c + K * lngamma(V * e) - KV * lngamma(e); };
    const bool loud = g_verbosity > 1;
    if (loud)
        std::printf("Optimizing eta...\n");
    double Lb = bound(eta), Lprime = Lb;
    for (int it = 0; it < max_iter_eta; ++it) {                      // batchlda.cpp:160-203
        if (loud)
            say("Current function value", Lb);
        const double g = c - KV * (psi(eta) - psi(V * eta));
        const double h = KV * (psi1(V * eta) - psi1(eta));
        double rho = .5;
        for (int j = 0; j < 20; ++j) {
            const double cand = eta - rho * g / h;
            if (cand < min_eta) {
                rho /= 2.;
                continue;
            }
            Lprime = bound(cand);
            if (Lb <= Lprime) {
                if (loud) {
                    say("Step width", rho);
                    say("Gradient", g);
                }
                eta = cand;
                break;
            }
            rho /= 2.;
        }
        if (Lprime - Lb < threshold)
            break;
        Lb = Lprime;
    }
    if (loud)
        std::fflush(stdout);
    return eta;
}

/* test hook: psi and psi' of this file at n points */
void trlda_debug_host_psi(int n, const double *x, double *psi_out, double *psi1_out)
{
    for (int i = 0; i < n; ++i) {
        psi_out[i] = psi(x[i]);
        psi1_out[i] = psi1(x[i]);
    }
}

} // extern "C"
