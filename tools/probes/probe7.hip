// Probe 7: accuracy of v_rcp_f64 and of one / two Newton steps on it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k(int n, const double *x, double *r0, double *r1, double *r2)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    double s = x[i];
    double r = __builtin_amdgcn_rcp(s); r0[i] = r;
    r = fma(fma(-s, r, 1.0), r, r); r1[i] = r;
    r = fma(fma(-s, r, 1.0), r, r); r2[i] = r;
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), a(n), b(n), c(n);
    unsigned long long st = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; double u = (st >> 11) * (1.0 / 9007199254740992.0); x[i] = std::exp((u - 0.5) * 40.0); }
    double *dx, *d0, *d1, *d2; CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&d0, n * 8)); CK(hipMalloc(&d1, n * 8)); CK(hipMalloc(&d2, n * 8));
    CK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, n, dx, d0, d1, d2); CK(hipDeviceSynchronize());
    CK(hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost));
    double e0 = 0, e1 = 0, e2 = 0; long exact2 = 0, exact1 = 0;
    for (int i = 0; i < n; ++i) {
        long double t = 1.0L / (long double)x[i];
        e0 = fmax(e0, (double)fabsl((a[i] - t) / t)); e1 = fmax(e1, (double)fabsl((b[i] - t) / t)); e2 = fmax(e2, (double)fabsl((c[i] - t) / t));
        exact1 += (b[i] == 1.0 / x[i]); exact2 += (c[i] == 1.0 / x[i]);
    }
    printf("v_rcp_f64 max rel err %.3e (2^%.1f); +1 Newton %.3e (%.1f%% correctly rounded); +2 Newton %.3e (%.1f%% correctly rounded)\n",
           e0, log2(e0), e1, 100.0 * exact1 / n, e2, 100.0 * exact2 / n);
    return 0;
}
