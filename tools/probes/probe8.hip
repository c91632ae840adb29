// probe8 -- what one lone wavefront pays for exp(psi(x)) and its pieces (cycles per evaluation in
// a dependent chain of evaluations), Horner vs Estrin exponential.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/probe8.hip -o tools/probes/probe8
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../trlda_amd/csrc/psi.h"
using namespace trlda;

// the Estrin-scheme exponential that was tried in the psi stage (four dependent steps after the
// argument reduction instead of Horner's eleven) -- kept here, where it was measured
__device__ __forceinline__ double exp_short_chain(double x)
{
    const double n = rint(x * 1.44269504088896338700e+00);
    double r = fma(n, -6.93147180369123816490e-01, x);
    r = fma(n, -1.90821492927058770002e-10, r);
    const double r2 = r * r;
    const double p01 = r + 1.0, p23 = fma(r, 1.0 / 6.0, 0.5), p45 = fma(r, 1.0 / 120.0, 1.0 / 24.0);
    const double p67 = fma(r, 1.0 / 5040.0, 1.0 / 720.0), p89 = fma(r, 1.0 / 362880.0, 1.0 / 40320.0);
    const double pab = fma(r, 1.0 / 39916800.0, 1.0 / 3628800.0);
    const double pcd = fma(r, 1.0 / 6227020800.0, 1.0 / 479001600.0);
    const double r4 = r2 * r2;
    const double q0 = fma(p23, r2, p01), q1 = fma(p67, r2, p45), q2 = fma(pab, r2, p89);
    const double r8 = r4 * r4;
    return ldexp(fma(fma(pcd, r4, q2), r8, fma(q1, r4, q0)), (int)n);
}
__device__ __forceinline__ double exp_digamma_chain(double x)
{
    double pi[5];
    for (int i = 0; i < 5; ++i) pi[i] = rcp_pair(x + (double)(2 * i));
    const double s = x + 10.0, w = ((pi[0] + pi[1]) + (pi[2] + pi[3])) + pi[4], r = rcp_pos<true>(s);
    const double z = r * r, y = z * psi_series(z);
    return s * exp_short_chain(-(((0.5 * r) + y) + w));
}

template <int WHICH>
__global__ void chain(double *out, unsigned long long *cyc, double x0, int n)
{
    double x = x0 + threadIdx.x * 1e-3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        double v;
        if (WHICH == 0) v = exp_digamma(x);
        else if (WHICH == 1) v = exp_digamma_chain(x);
        else if (WHICH == 2) v = exp(-x);                 // library exp alone
        else if (WHICH == 3) v = exp_short_chain(-x);
        else if (WHICH == 4) v = ((rcp_pair(x) + rcp_pair(x + 2.0)) + (rcp_pair(x + 4.0) + rcp_pair(x + 6.0))) + rcp_pair(x + 8.0);
        else v = rcp_pos<true>(x);
        x = x0 + v * 1e-6;                                // the next evaluation depends on this one
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int WHICH>
void run(const char *name, int threads)
{
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 8);
    const int n = 2000;
    hipLaunchKernelGGL(chain<WHICH>, dim3(1), dim3(threads), 0, 0, out, cyc, 0.7, n);
    hipLaunchKernelGGL(chain<WHICH>, dim3(1), dim3(threads), 0, 0, out, cyc, 0.7, n);
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s %4d threads/CU  %8.1f cycles per evaluation\n", name, threads, (double)h / n);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int threads : {64, 256, 512}) {
        run<0>("exp_digamma (library exp)", threads);
        run<1>("exp_digamma_chain (Estrin)", threads);
        run<2>("library exp alone", threads);
        run<3>("exp_short_chain alone", threads);
        run<4>("five reciprocal pairs + sum", threads);
        run<5>("one rcp_pos", threads);
    }
    return 0;
}
