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

// d = a b + c as the three-address instruction, whatever the register allocator thinks: left to
// itself the compiler forms Horner steps as v_fmac_f64 (d += a b) with a v_mov_b64 of the
// coefficient in front of each -- 15 moves in the exp(psi) stage of the document kernels, whose
// waves are bound by their instruction count.  fma3s: the addend from a scalar
// register pair (a literal coefficient: no vector register held for it).
__device__ __forceinline__ double fma3(double a, double b, double c)
{
#ifdef TRLDA_NO_FMA3
    return fma(a, b, c);
#else
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
#endif
}
__device__ __forceinline__ double fma3s(double a, double b, double c)
{
#ifdef TRLDA_NO_FMA3
    return fma(a, b, c);
#else
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
#endif
}

__device__ __forceinline__ double psi_series(double z)
{
    // polevl(z, A, 6), src/digamma.cpp:96-110 (coefficients :44-52), evaluated pairwise
    // (Estrin) so that the dependent chain is three fmas instead of six
    const double z2 = z * z, z4 = z2 * z2;
    const double c01 = fma(-8.33333333333333333333E-3, z, 8.33333333333333333333E-2);
    const double c23 = fma(-4.16666666666666666667E-3, z, 3.96825396825396825397E-3);
    const double c45 = fma(-2.10927960927960927961E-2, z, 7.57575757575757575758E-3);
    const double c6 = 8.33333333333333333333E-2;
    // A0 z^6 + A1 z^5 + ... + A6  with A0 = c6, A1,A2 = c45, A3,A4 = c23, A5,A6 = c01
    return fma(fma(c6, z2, c45), z4, fma(c23, z2, c01));
}

// 1/s for normal positive s: v_rcp_f64 seed (about 2^-26 relative) refined by two
// Newton steps, each an exact-residual fma pair -> error below one ulp.  Five
// instructions instead of the eleven of the IEEE-exact division expansion.
template <bool FAST>
__device__ __forceinline__ double rcp_pos(double s)
{
    if (!FAST)
        return 1.0 / s;
    double r = __builtin_amdgcn_rcp(s);
    r = fma(fma(-s, r, 1.0), r, r);
    r = fma(fma(-s, r, 1.0), r, r);
    return r;
}

// log(s) for normal s > 0 (psi only ever needs s >= 10).  frexp, fold the mantissa into
// [sqrt(1/2), sqrt(2)), log(m) = 2 atanh(u) with u = (m-1)/(m+1), |u| <= 0.1716: the odd
// series through u^21 (truncation < 1e-17), evaluated Estrin-style so that the dependent
// chain is 4 fmas deep instead of 10 -- a dependent fp64 op costs ~37 cycles on gfx950.
// About 35 instructions against ~100 for the general-purpose library log; the result is
// within 1 ulp of it (tests/test_gpu_parity.py::test_device_digamma_table).
__device__ __forceinline__ double log_normal(double s)
{
    int e;
    double m = frexp(s, &e);
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double f = m - 1.0;
    const double u = f * rcp_pos<true>(2.0 + f);
    const double z = u * u;
    const double z2 = z * z, z4 = z2 * z2, z8 = z4 * z4;
    // (three-address steps, the addend from a scalar register: no coefficient moves -- fma3s above)
    const double p01 = fma3s(2.0 / 3.0, z, 2.0);
    const double p23 = fma3s(2.0 / 7.0, z, 2.0 / 5.0);
    const double p45 = fma3s(2.0 / 11.0, z, 2.0 / 9.0);
    const double p67 = fma3s(2.0 / 15.0, z, 2.0 / 13.0);
    const double p89 = fma3s(2.0 / 19.0, z, 2.0 / 17.0);
    const double q0 = fma3(p23, z2, p01);
    const double q1 = fma3(p67, z2, p45);
    const double q2 = fma3(2.0 / 21.0, z2, p89);
    const double poly = fma3(q2, z8, fma3(q1, z4, q0));
    const double ed = (double)e;
    return fma(ed, 6.93147180369123816490e-01, fma(u, poly, ed * 1.90821492927058770002e-10));
}

template <bool FAST>
__device__ __forceinline__ double psi_positive_impl(double x)
{
    // psi(x) = psi(x + 10) - sum_{i<10} 1/(x + i).  The reference's
    // `while (s < 10) { w += 1/s; s += 1; }` (src/digamma.cpp:158-163) shifts by the
    // smallest m that reaches s >= 10; the identity holds for every m, and always taking
    // ten steps removes every data-dependent select from the hot path (s then lies in
    // [10, 20) for x < 10, where the series is even more accurate).  Laid out for
    // instruction-level parallelism -- a dependent fp64 op costs ~37 cycles on gfx950 -- the
    // ten reciprocals are independent and w is a pairwise sum.  Against the serial loop
    // this moves psi by a few 1e-16 relative.
    double ri[10];
#pragma unroll
    for (int i = 0; i < 10; ++i)
        ri[i] = rcp_pos<FAST>(x + (double)i);
    const double s = x + 10.0;
    const double w = (((ri[0] + ri[1]) + (ri[2] + ri[3])) + ((ri[4] + ri[5]) + (ri[6] + ri[7]))) +
                     (ri[8] + ri[9]);
    double y = 0.0;
    const double r = rcp_pos<FAST>(s);
    if (s < 1.0e17) {
        // z = 1/s^2 feeds a correction of at most 8.4e-4: r*r instead of a third
        // reciprocal changes psi by < 1e-19
        const double z = r * r;
        y = z * psi_series(z);
    }
    return ((FAST ? log_normal(s) : log(s)) - (0.5 * r) - y) - w;
}

// x > 0 and not a small integer: the branch every gamma / lambda element takes
// when alpha, eta > 0.  Arguments outside [1e-290, 1e290] (where the Newton residuals
// could over/underflow) take the exact-division variant.
__device__ __forceinline__ double psi_positive(double x)
{
    if (__builtin_expect(x > 1e-290 && x < 1e290, 1))
        return psi_positive_impl<true>(x);
    return psi_positive_impl<false>(x);
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

// exp(psi(x) - c) without the logarithm: with s = x + 10 and
//   psi(x) = log(s) - 1/(2s) - series(1/s^2) - sum_{i<10} 1/(x+i)      (digamma.cpp:158-171)
// the log comes straight back out of the exponential, exp(psi(x) - c) = s * exp(-(t + c)),
// t = 1/(2s) + series + sum >= 0.  One exp instead of log + exp -- the log is the longest
// dependency chain of psi -- and no cancellation between log(s) and t.  Both callers of the
// reference's digamma on the hot path only ever want exp(psi(.)) (lda.cpp:173-174, :197).
//
// 1/a + 1/(a + 1) = (2a + 1) / (a (a + 1)): one reciprocal for two terms of the recurrence
// (a^2 must not overflow: callers keep a below 1e150).  A couple of ulp.
__device__ __forceinline__ double rcp_pair(double a)
{
    return fma(2.0, a, 1.0) * rcp_pos<true>(fma(a, a, a));
}

// The ten recurrence terms as ONE rational function: with P(x) = x (x+1) ... (x+9),
//   sum_{i<10} 1/(x+i) = P'(x) / P(x).
// For x > 0 every coefficient of P and P' (Stirling numbers of the first kind, exact in fp64)
// and every Horner step is positive -- no cancellation: the quotient is within ~3 ulp of the
// exact sum, like the five reciprocal pairs it replaces (rcp_pair, kept for the rare branch) --
// and it is 23 instructions shorter: two Horner chains (in u = x (x + 9) since round 5: another six) and ONE reciprocal,
// 1 / (P s), which also yields 1/s = P / (P s).  The stage that evaluates this is bound by
// the instruction count of a single wave (~8.6 cycles per fp64 instruction).
// Needs x^11 finite: callers keep x below 1e25.
// exp(a) for a <= 0 (or NaN: some finite value -- every caller multiplies by a NaN then).  The
// library's exp spends two compares and three selects on the overflow / underflow ends; here the
// argument is clamped once (exp(-800) and everything below it round to 0 through v_ldexp_f64, as
// the reference's exp(-1e290) does) and the rest is the usual reduction a = n ln2 + r,
// |r| <= ln2 / 2, and 1 + r + r^2 q(r) with q the degree-9 Chebyshev fit of (e^r - 1 - r) / r^2
// (fit error 1e-16 of q = 1.3e-17 of the result; within 1.5 ulp of exp over the interval).
// SC: the polynomial's coefficients from scalar registers instead of vector ones -- 20 vector
// registers fewer.  For the kernels that are out of them: the single-orientation document kernel
// at K > 384 spilled 14-18 registers and reloaded six of them in every iteration's psi stage;
// with SC nothing spills and its launches are 13 % shorter (K = 500: 1518 -> 1318 us per 4096
// documents).  Not for the others: where the scalar registers are the scarce ones (the K <= 128
// kernels, above all with the statistics stage inside) the same switch costs 1.5 %
// (profiles/r04_scoef_ab.txt).
// ANY: the argument may be positive, infinite or NaN (exp(psi(x) - c) with a row sum's psi as c):
// clamped at both ends by selects -- a NaN stays one, +-inf come out as inf / 0 through v_ldexp_f64
template <bool SC = false, bool ANY = false>
__device__ __forceinline__ double exp_nonpos(double a)
{
    if constexpr (ANY) {
        a = a < -800.0 ? -800.0 : a;
        a = a > 720.0 ? 720.0 : a;
    } else {
        a = fmax(a, -800.0);
    }
    const double n = rint(a * 1.44269504088896340736);
    double r = fma(n, -6.93147180369123816490e-01, a);           // ln2 in two pieces
    r = fma(n, -1.90821492927058770002e-10, r);
#ifdef TRLDA_EXP_ESTRIN
    // pairwise: depth 6 instead of 11, three instructions more -- measured slower (30.5 against
    // 30.3 us for the register kernel, profiles/r04_psi_variants.txt): the count decides, not the depth
    const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
    const double a0 = fma3s(1.66666666666666685e-01, r, 5.00000000000000111e-01);
    const double a1 = fma3s(8.33333333333006500e-03, r, 4.16666666666241636e-02);
    const double a2 = fma3s(1.98412698630405450e-04, r, 1.38888889171967186e-03);
    const double a3 = fma3s(2.75572684803100238e-06, r, 2.48015213223686919e-05);
    const double a4 = fma3s(2.51003758325612340e-08, r, 2.76200758799833672e-07);
    const double b0 = fma3(a1, r2, a0), b1 = fma3(a3, r2, a2);
    double q = fma3(b1, r4, b0);
    q = fma3(a4, r8, q);
    const double p = fma3(r2, q, r) + 1.0;
#else
    double p;
    if constexpr (SC) {
        p = fma3s(r, 2.51003758325612340e-08, 2.76200758799833672e-07);
        p = fma3s(r, p, 2.75572684803100238e-06);
        p = fma3s(r, p, 2.48015213223686919e-05);
        p = fma3s(r, p, 1.98412698630405450e-04);
        p = fma3s(r, p, 1.38888889171967186e-03);
        p = fma3s(r, p, 8.33333333333006500e-03);
        p = fma3s(r, p, 4.16666666666241636e-02);
        p = fma3s(r, p, 1.66666666666666685e-01);
        p = fma3s(r, p, 5.00000000000000111e-01);
    } else {
        p = fma3(r, 2.51003758325612340e-08, 2.76200758799833672e-07);
        p = fma3(r, p, 2.75572684803100238e-06);
        p = fma3(r, p, 2.48015213223686919e-05);
        p = fma3(r, p, 1.98412698630405450e-04);
        p = fma3(r, p, 1.38888889171967186e-03);
        p = fma3(r, p, 8.33333333333006500e-03);
        p = fma3(r, p, 4.16666666666241636e-02);
        p = fma3(r, p, 1.66666666666666685e-01);
        p = fma3(r, p, 5.00000000000000111e-01);
    }
    p = fma(r, p, 1.0);
    p = fma(r, p, 1.0);
#endif
    return ldexp(p, (int)n);
}

// the same polynomial by Horner's rule with three-address steps, coefficients from scalar registers:
// six instructions where the pairwise form is seven plus three moves (the stage this is for is
// bound by its instruction count, not by the depth of the chain)
__device__ __forceinline__ double psi_series_horner(double z)
{
    double p = fma3s(8.33333333333333333333E-2, z, -2.10927960927960927961E-2);
    p = fma3s(p, z, 7.57575757575757575758E-3);
    p = fma3s(p, z, -4.16666666666666666667E-3);
    p = fma3s(p, z, 3.96825396825396825397E-3);
    p = fma3s(p, z, -8.33333333333333333333E-3);
    return fma3s(p, z, 8.33333333333333333333E-2);
}

// XF: Horner's rule in x (rounds 2-5a) -- for the one kernel that spills with the shorter form, the
// single-orientation document kernel at K > 448 (4 registers, 2.79 against 2.65 ms per
// update_parameters call at K = 500 / 512 documents)
template <bool ZERO_C = false, bool SC = false, bool XF = false>
__device__ __forceinline__ double exp_psi_regular(double x, double c)
{
    double P, dP;
#ifdef TRLDA_PSI_HORNER_X                            // (A/B: everywhere)
    constexpr bool in_x = true;
#else
    constexpr bool in_x = XF;
#endif
    if constexpr (in_x) {
        double q = x + 45.0;
        q = fma(q, x, 870.0);
        q = fma(q, x, 9450.0);
        q = fma(q, x, 63273.0);
        q = fma(q, x, 269325.0);
        q = fma(q, x, 723680.0);
        q = fma(q, x, 1172700.0);
        q = fma(q, x, 1026576.0);
        q = fma(q, x, 362880.0);
        P = q * x;
        dP = fma3s(10.0, x, 405.0);
        dP = fma(dP, x, 6960.0);
        dP = fma(dP, x, 66150.0);
        dP = fma(dP, x, 379638.0);
        dP = fma(dP, x, 1346625.0);
        dP = fma(dP, x, 2894720.0);
        dP = fma(dP, x, 3518100.0);
        dP = fma(dP, x, 2053152.0);
        dP = fma(dP, x, 362880.0);
    } else {
        // The factors pair up, (x + i)(x + 9 - i) = u + i (9 - i) with u = x (x + 9):
        //   P = u (u + 8)(u + 14)(u + 18)(u + 20) = u^5 + 60 u^4 + 1308 u^3 + 12176 u^2 + 40320 u,
        //   P' = dP/du (2x + 9)
        // -- thirteen instructions for the two where Horner's rule in x takes nineteen; every coefficient
        // and every step is positive for x > 0 as before (P'/P against the exact sum in binary128 over
        // 1e-10 .. 1e4: 8.5e-16 relative at worst, the form in x 1.0e-15; tools/probes/psi_u_form.c)
        const double u = x * (x + 9.0);
        double q = u + 60.0;
        q = fma(q, u, 1308.0);
        q = fma(q, u, 12176.0);
        q = fma(q, u, 40320.0);
        P = q * u;
        dP = fma3s(5.0, u, 240.0);
        dP = fma(dP, u, 3924.0);
        dP = fma(dP, u, 24352.0);
        dP = fma(dP, u, 40320.0);
        dP *= fma(2.0, x, 9.0);
    }
    const double s = x + 10.0;
    const double inv = rcp_pos<true>(P * s);
    const double r = P * inv;                        // 1 / s
    const double w = (dP * s) * inv;                 // P' / P
    const double z = r * r;
    if constexpr (ZERO_C) {
#ifndef TRLDA_LIBRARY_EXP
        const double yh = z * psi_series_horner(z);
        return s * exp_nonpos<SC>(-((fma(0.5, r, yh)) + w));     // psi(x) < log(x + 10): the argument is <= 0
#endif
    }
#ifdef TRLDA_GENERAL_LIBRARY_EXP
    const double y = z * psi_series(z);
    return s * exp(-((fma(0.5, r, y)) + w) - c);
#else
    const double y = z * psi_series_horner(z);
    return s * exp_nonpos<SC, true>(-((fma(0.5, r, y)) + w) - c);
#endif
}

// the form with five reciprocal pairs: valid up to 1e150 (the rational form needs x^11 finite)
__device__ __noinline__ double exp_psi_wide_range(double x, double c)
{
    double pi[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
        pi[i] = rcp_pair(x + (double)(2 * i));
    const double s = x + 10.0;
    const double w = ((pi[0] + pi[1]) + (pi[2] + pi[3])) + pi[4];
    const double r = rcp_pos<true>(s);
    double y = 0.0;
    if (s < 1.0e17) {
        const double z = r * r;
        y = z * psi_series(z);
    }
    return s * exp(-(((0.5 * r) + y) + w) - c);
}

// x <= 0, small integers, and everything outside (1e-290, 1e25)
__device__ __noinline__ double exp_digamma_rare(double x, double c)
{
    if (x >= 1e25 && x < 1e150)
        return exp_psi_wide_range(x, c);
    return exp(digamma(x) - c);
}

__device__ __forceinline__ double exp_digamma_minus(double x, double c)
{
    // the regular value is computed unconditionally so that the (rare-branch) test runs
    // beside the main dependency chain instead of in front of it
    const double v = exp_psi_regular(x, c);
    if (__builtin_expect(!(x > 1e-290 && x < 1e25) || (x <= 10.0 && x == floor(x)), 0))
        return exp_digamma_rare(x, c);
    return v;
}

// c = 0 (the document kernels, the fused preamble): the same with the short exponential
template <bool SC = false, bool XF = false>
__device__ __forceinline__ double exp_digamma(double x)
{
    const double v = exp_psi_regular<true, SC, XF>(x, 0.0);
    // (the integers 1 .. 10, where the reference takes the exact harmonic sum, src/digamma.cpp:147-156,
    // go through the regular form here as in exp_digamma_positive: psi(n) to a few ulp either way,
    // 2e-15 apart at most -- and three instructions fewer in the psi waves' stream: 0.2 us of the
    // document kernel.  TRLDA_INT_BRANCH puts the branch back.)
#ifdef TRLDA_INT_BRANCH
    if (__builtin_expect(!(x > 1e-290 && x < 1e25) || (x <= 10.0 && x == floor(x)), 0))
#else
    if (__builtin_expect(!(x > 1e-290 && x < 1e25), 0))
#endif
        return exp_digamma_rare(x, 0.0);
    return v;
}

// exp(psi(x)) for an argument that is KNOWN to be positive (or NaN) -- the lambda an M-step has
// just formed from a positive lambda' or a positive eta and non-negative statistics, which the
// host keeps track of (trlda_model::lambda_positive).  Selects only, no call and no branch: the
// general form's rare branch (x <= 0: reflection through tan, src/digamma.cpp:123-144) is a
// function call that alone puts a kernel at ~120 VGPRs, and even a branch to a table of the
// small integers' values costs ~35 -- the statistics kernel that also leaves the next preamble
// behind must stay at 64 to keep two workgroups per CU (estep_kernels.h, 4c / 4d).
//   0 < x < 1e-290   psi(x) = -1/x - ..., exp(psi(x)) = 0 like the reference's exp(-1e290)
//   x = 1 .. 10      the regular form (the reference takes the exact harmonic sum there,
//                    src/digamma.cpp:147-156; both are psi(n) to a few ulp: 2e-15 apart at most,
//                    tests/test_gpu_parity.py::test_device_digamma_table)
//   x >= 1e25        exp(psi(x)) = x - 1/2 + O(1/x) = x in fp64 (inf stays inf)
//   NaN              NaN
// XF: Horner's rule in x, as exp_psi_regular -- the stand-alone statistics kernels (at their register
// caps the shorter form spills more, and K = 500's update calls were 2-5 % slower with it)
template <bool XF = false>
__device__ __forceinline__ double exp_digamma_positive(double x)
{
#ifdef TRLDA_EXPT_NOEXP                              // timing experiment: results are wrong
    return x;
#endif
    const bool tiny = x < 1e-290, big = !(x < 1e25);
    double v = exp_psi_regular<true, false, XF>((tiny || big) ? 1.5 : x, 0.0);
    v = tiny ? 0.0 : v;
    return big ? x : v;
}

}  // namespace trlda
