// rcp_probe -- how good is v_rcp_f64 on gfx950, raw and after one / two Newton steps?
// (psi.h's rcp_pos takes two; the reciprocal sits in the dependent chains of the document kernels'
// weights and exp(psi) stages.)  Error in ulp of the exact 1/x over 2^24 values per decade sample.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

__global__ void k(const double *x, double *r0, double *r1, double *r2, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double s = x[i];
    double r = __builtin_amdgcn_rcp(s);
    r0[i] = r;
    r = fma(fma(-s, r, 1.0), r, r);
    r1[i] = r;
    r = fma(fma(-s, r, 1.0), r, r);
    r2[i] = r;
}

int main()
{
    const int n = 1 << 22;
    std::vector<double> x(n), a(n), b(n), c(n);
    unsigned long long st = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        const double m = 1.0 + (double)(st >> 11) / 9007199254740992.0;       // [1, 2)
        x[i] = ldexp(m, (int)(st % 600) - 300);
    }
    double *dx, *d0, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
    hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double w0 = 0, w1 = 0, w2 = 0;
    long long ex1 = 0, ex2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = 1.0L / (long double)x[i];
        const double ref = (double)t;
        const double ulp = ldexp(1.0, ilogb(ref) - 52);
        w0 = fmax(w0, fabs((double)((long double)a[i] - t)) / ulp);
        w1 = fmax(w1, fabs((double)((long double)b[i] - t)) / ulp);
        w2 = fmax(w2, fabs((double)((long double)c[i] - t)) / ulp);
        ex1 += b[i] == ref;
        ex2 += c[i] == ref;
    }
    printf("v_rcp_f64: worst %.3g ulp; one Newton step: %.3f ulp (%.2f %% correctly rounded); two: %.3f ulp (%.2f %%)\n",
           w0, w1, 100.0 * ex1 / n, w2, 100.0 * ex2 / n);
    return 0;
}
