// psi.h -- fp64 digamma for gfx950 device code.
//
// Same function as the reference's TRLDA::digamma(double) (src/digamma.cpp:116-178,
// Cephes psi): reflection for x <= 0, exact harmonic sum for integer x <= 10, upward
// recurrence w = sum 1/s until s >= 10, then log(s) - 0.5/s - z*P6(z) - w with
// z = 1/s^2 and the seven Bernoulli-series coefficients of src/digamma.cpp:44-52.
//
// Device shape: the data-dependent `while (s < 10)` becomes ten predicated steps so a
// wavefront's lanes stay converged and the ten independent fp64 divides pipeline; the
// additions into w happen in the reference's order (s = x, x+1, ...), so on the
// positive non-integer branch the only differences from the CPU result are the last-bit
// behaviour of the device log() and fused multiply-adds.
#pragma once
#include <hip/hip_runtime.h>

namespace trlda {

__device__ __forceinline__ double psi_series(double z)
{
    // polevl(z, A, 6), src/digamma.cpp:96-110
    double p = 8.33333333333333333333E-2;
    p = p * z + -2.10927960927960927961E-2;
    p = p * z + 7.57575757575757575758E-3;
    p = p * z + -4.16666666666666666667E-3;
    p = p * z + 3.96825396825396825397E-3;
    p = p * z + -8.33333333333333333333E-3;
    p = p * z + 8.33333333333333333333E-2;
    return p;
}

// x > 0 and not a small integer: the branch every gamma / lambda element takes
// when alpha, eta > 0.
__device__ __forceinline__ double psi_positive(double x)
{
    double s = x, w = 0.0;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        bool below = s < 10.0;
        double r = 1.0 / s;
        w = below ? w + r : w;
        s = below ? s + 1.0 : s;
    }
    double y = 0.0;
    if (s < 1.0e17) {
        double z = 1.0 / (s * s);
        y = z * psi_series(z);
    }
    return log(s) - (0.5 / s) - y - w;
}

__device__ __noinline__ double psi_rare(double x)
{
    // x <= 0 (src/digamma.cpp:123-144) or integer x <= 10 (:147-156)
    const double pi = 3.141592653589793238462643383279502884;
    const double euler = 0.577215664901532860606512090082402431;
    double reflect = 0.0;
    bool reflected = false;
    if (x <= 0.0) {
        double fl = floor(x);
        if (fl == x)
            return __builtin_huge_val();
        double frac = x - fl;
        if (frac != 0.5) {
            if (frac > 0.5) {
                fl += 1.0;
                frac = x - fl;
            }
            reflect = pi / tan(pi * frac);
        }
        reflected = true;
        x = 1.0 - x;
    }
    double y;
    if (x <= 10.0 && x == floor(x)) {
        int n = (int)x;
        y = 0.0;
        for (int i = 1; i < n; ++i)
            y += 1.0 / (double)i;
        y -= euler;
    } else {
        y = psi_positive(x);
    }
    return reflected ? y - reflect : y;
}

__device__ __forceinline__ double digamma(double x)
{
    if (__builtin_expect(x <= 0.0 || (x <= 10.0 && x == floor(x)), 0))
        return psi_rare(x);
    return psi_positive(x);
}

__device__ __forceinline__ double exp_digamma(double x) { return exp(digamma(x)); }

}  // namespace trlda
