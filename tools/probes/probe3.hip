// Probe 3: the product-B / product-E inner loops in isolation (T threads, K=100, n=100, Kp=101).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NA>
__device__ __forceinline__ double lds_dot(const double *a, const double *b, int stride, int count)
{
    double acc[NA];
#pragma unroll
    for (int u = 0; u < NA; ++u) acc[u] = 0.0;
    int i = 0;
    for (; i + NA <= count; i += NA) {
        double x[NA], y[NA];
#pragma unroll
        for (int u = 0; u < NA; ++u) { x[u] = a[i + u]; y[u] = b[(i + u) * stride]; }
#pragma unroll
        for (int u = 0; u < NA; ++u) acc[u] = fma(x[u], y[u], acc[u]);
    }
#pragma unroll
    for (int u = 0; u < NA - 1; ++u)
        if (i + u < count) acc[u] = fma(a[i + u], b[(i + u) * stride], acc[u]);
#pragma unroll
    for (int w = NA / 2; w > 0; w >>= 1)
#pragma unroll
        for (int u = 0; u < w; ++u) acc[u] += acc[u + w];
    return acc[0];
}

template <int T, int NA>
__global__ __launch_bounds__(T) void prod(double *out, unsigned long long *t, int K, int n, int Kp, int reps)
{
    extern __shared__ double lds[];
    double *beta = lds, *tw = beta + n * Kp, *e = tw + n, *part = e + K;
    constexpr int W = T / 64;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < n * Kp; i += T) beta[i] = 1.0 + i * 1e-6;
    for (int i = tid; i < n; i += T) tw[i] = 1.0 + i * 1e-3;
    for (int i = tid; i < K; i += T) e[i] = 1.0 + i * 1e-3;
    __syncthreads();
    const int KB = (K + 63) / 64, JP = W / KB, kb = wid % KB, jp = wid / KB;
    const int k = kb * 64 + lane, JC = (n + JP - 1) / JP, j0 = min(n, jp * JC), j1 = min(n, j0 + JC);
    const int JB = (n + 63) / 64, KPn = W / JB, jb = wid % JB, kp = wid / JB;
    const int j = jb * 64 + lane, KC = (K + KPn - 1) / KPn, k0 = min(K, kp * KC), k1 = min(K, k0 + KC);
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        if (jp < JP && k < K) part[jp * K + k] = lds_dot<NA>(tw + j0, beta + j0 * Kp + k, Kp, j1 - j0);
        __syncthreads();
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        if (kp < KPn && j < n) part[kp * n + j] = lds_dot<NA>(e + k0, beta + j * Kp + k0, 1, k1 - k0);
        __syncthreads();
    }
    unsigned long long c2 = __builtin_amdgcn_s_memtime();
    out[tid] = part[tid % (K * JP)];
    if (tid == 0) { t[0] = c1 - c0; t[1] = c2 - c1; }
}

template <int T, int NA> int run(double *out, unsigned long long *t)
{
    const int K = 100, n = 100, Kp = 101, reps = 100;
    size_t lds = (size_t)(n * Kp + n + K + T) * 8;
    CK(hipFuncSetAttribute((const void *)prod<T, NA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((prod<T, NA>), dim3(1), dim3(T), lds, 0, out, t, K, n, Kp, reps); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((prod<T, NA>), dim3(1), dim3(T), lds, 0, out, t, K, n, Kp, reps); CK(hipDeviceSynchronize());
    unsigned long long h[2]; CK(hipMemcpy(h, t, 16, hipMemcpyDeviceToHost));
    printf("T=%4d NA=%d: product B %.0f cycles, product E %.0f cycles (incl. 1 barrier)\n", T, NA, h[0] / 100.0, h[1] / 100.0);
    return 0;
}

int main()
{
    double *out; unsigned long long *t;
    CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&t, 64));
    run<256, 8>(out, t); run<256, 4>(out, t); run<256, 2>(out, t); run<256, 1>(out, t);
    run<512, 8>(out, t); run<512, 4>(out, t);
    run<1024, 8>(out, t); run<1024, 4>(out, t); run<1024, 2>(out, t);
    return 0;
}
