#include <math.h>
#include <quadmath.h>
#include <stdio.h>
#include <stdlib.h>
static double w_old(double x, double *rinv) {
    double q = x + 45.0;
    q = fma(q, x, 870.0); q = fma(q, x, 9450.0); q = fma(q, x, 63273.0); q = fma(q, x, 269325.0);
    q = fma(q, x, 723680.0); q = fma(q, x, 1172700.0); q = fma(q, x, 1026576.0); q = fma(q, x, 362880.0);
    const double P = q * x;
    double dP = fma(10.0, x, 405.0);
    dP = fma(dP, x, 6960.0); dP = fma(dP, x, 66150.0); dP = fma(dP, x, 379638.0); dP = fma(dP, x, 1346625.0);
    dP = fma(dP, x, 2894720.0); dP = fma(dP, x, 3518100.0); dP = fma(dP, x, 2053152.0); dP = fma(dP, x, 362880.0);
    const double s = x + 10.0;
    const double inv = 1.0 / (P * s);
    *rinv = P * inv;
    return (dP * s) * inv;
}
/* P = u (u+8)(u+14)(u+18)(u+20), u = x (x+9):  u^5 + 60 u^4 + 1388 u^3 + 15120 u^2 + 40320 u... computed below */
static double c4, c3, c2, c1, d3, d2, d1, d0;
static void coeffs(void) {
    /* expand (u)(u+8)(u+14)(u+18)(u+20) */
    double r[6] = {0, 1, 0, 0, 0, 0}; /* u */
    double roots[4] = {8, 14, 18, 20};
    int deg = 1;
    for (int i = 0; i < 4; ++i) {
        double n[7] = {0};
        for (int j = 0; j <= deg; ++j) { n[j + 1] += r[j]; n[j] += r[j] * roots[i]; }
        ++deg;
        for (int j = 0; j <= deg; ++j) r[j] = n[j];
    }
    /* r[5] u^5 + r[4] u^4 + ... + r[1] u */
    c4 = r[4]; c3 = r[3]; c2 = r[2]; c1 = r[1];
    d3 = 4 * r[4]; d2 = 3 * r[3]; d1 = 2 * r[2]; d0 = r[1];
    printf("P(u) = u^5 + %.0f u^4 + %.0f u^3 + %.0f u^2 + %.0f u;  dP/du = 5 u^4 + %.0f u^3 + %.0f u^2 + %.0f u + %.0f\n", c4, c3, c2, c1, d3, d2, d1, d0);
}
static double w_new(double x, double *rinv) {
    const double u = x * (x + 9.0);
    double q = u + c4;
    q = fma(q, u, c3); q = fma(q, u, c2); q = fma(q, u, c1);
    const double P = q * u;
    double dP = fma(5.0, u, d3);
    dP = fma(dP, u, d2); dP = fma(dP, u, d1); dP = fma(dP, u, d0);
    dP = dP * fma(2.0, x, 9.0);
    const double s = x + 10.0;
    const double inv = 1.0 / (P * s);
    *rinv = P * inv;
    return (dP * s) * inv;
}
int main(void) {
    coeffs();
    double worst_old = 0, worst_new = 0, worst_r_old = 0, worst_r_new = 0, xo = 0, xn = 0;
    srand(1);
    for (int i = 0; i < 4000000; ++i) {
        double e = -6.0 + 9.0 * rand() / RAND_MAX;   /* x from 1e-6 to 1e3 */
        double x = pow(10.0, e);
        __float128 ex = 0;
        for (int k = 0; k < 10; ++k) ex += 1.0Q / ((__float128)x + k);
        double r1, r2;
        double a = w_old(x, &r1), b = w_new(x, &r2);
        /* what matters: the ABSOLUTE error of w (it is an exponent) relative to 1, for x where exp(-w) is not 0 */
        double ea = fabs((double)((__float128)a - ex)), eb = fabs((double)((__float128)b - ex));
        if (x > 0.02) {           /* w < 50 */
            if (ea > worst_old) { worst_old = ea; xo = x; }
            if (eb > worst_new) { worst_new = eb; xn = x; }
        }
        double er1 = fabs(r1 * (x + 10.0) - 1.0), er2 = fabs(r2 * (x + 10.0) - 1.0);
        if (er1 > worst_r_old) worst_r_old = er1;
        if (er2 > worst_r_new) worst_r_new = er2;
    }
    printf("abs error of w for x > 0.02: old %.3g (at %.4g)  new %.3g (at %.4g); 1/s rel: old %.3g new %.3g\n", worst_old, xo, worst_new, xn, worst_r_old, worst_r_new);
    /* relative error of w (all x) */
    double ro = 0, rn = 0;
    for (int i = 0; i < 2000000; ++i) {
        double e = -10.0 + 14.0 * rand() / RAND_MAX;
        double x = pow(10.0, e);
        __float128 ex = 0;
        for (int k = 0; k < 10; ++k) ex += 1.0Q / ((__float128)x + k);
        double r1, r2;
        double a = w_old(x, &r1), b = w_new(x, &r2);
        double ea = fabs((double)(((__float128)a - ex) / ex)), eb = fabs((double)(((__float128)b - ex) / ex));
        if (ea > ro) ro = ea;
        if (eb > rn) rn = eb;
    }
    printf("rel error of w, x in [1e-10, 1e4]: old %.3g  new %.3g (ulp = 1.1e-16)\n", ro, rn);
    return 0;
}
