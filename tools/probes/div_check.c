// exhaustive: for every k in [0, 2^31), x = 2 k, d = 2147483647:
//   q0 = RN(x * y), r = fma(-q0, d, x), q1 = fma(r, y, q0)   with y = RN(1 / d)
// equals the IEEE quotient x / d ?
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <omp.h>
int main(void)
{
    const double d = 2147483647.0;
    const double y = 1.0 / d;
    long long bad = 0;
    #pragma omp parallel for reduction(+:bad) schedule(static)
    for (long long k = 0; k < (1LL << 31); ++k) {
        const double x = 2.0 * (double)k;
        const double q = x / d;
        const double q0 = x * y;
        const double r = fma(-q0, d, x);
        const double q1 = fma(r, y, q0);
        if (q1 != q) ++bad;
    }
    printf("y = %a; mismatches: %lld of %lld\n", y, bad, 1LL << 31);
    return bad != 0;
}
