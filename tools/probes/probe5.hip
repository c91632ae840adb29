// Probe 5: the compile-time-stride chunked dot product exactly as in the lean kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int STRIDE>
__device__ __forceinline__ double lds_dot_chunks(const double *__restrict__ wgt, const double *__restrict__ col, int chunks)
{
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
#ifdef ALIGNED_WGT
    wgt = (const double *)__builtin_assume_aligned(wgt, 16);
#endif
    for (int c = 0; c < chunks; ++c) {
        double x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { x[u] = wgt[u]; y[u] = col[u * STRIDE]; }
        wgt += 8; col += 8 * STRIDE;
        acc0 = fma(x[0], y[0], acc0); acc1 = fma(x[1], y[1], acc1); acc2 = fma(x[2], y[2], acc2); acc3 = fma(x[3], y[3], acc3);
        acc0 = fma(x[4], y[4], acc0); acc1 = fma(x[5], y[5], acc1); acc2 = fma(x[6], y[6], acc2); acc3 = fma(x[7], y[7], acc3);
    }
    return (acc0 + acc1) + (acc2 + acc3);
}

template <int T, int KP, int WHICH>
__global__ __launch_bounds__(T) void prod(double *out, unsigned long long *t, int K, int n, int reps, int active_waves)
{
    extern __shared__ double lds[];
    constexpr int W = T / 64;
    double *beta = lds, *tw = beta + (((n + 8) * KP + 1) & ~1), *e = tw + ((n + 8 + 1) & ~1), *part = e + K + 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (n + 8) * KP; i += T) beta[i] = 1.0 + i * 1e-6;
    for (int i = tid; i < n + 8; i += T) tw[i] = i < n ? 1.0 + i * 1e-3 : 0.0;
    for (int i = tid; i < K + 8; i += T) e[i] = i < K ? 1.0 + i * 1e-3 : 0.0;
    __syncthreads();
    const int KB = (K + 63) / 64, JP = W / KB, kb = wid % KB, jp = wid / KB;
    const int k_mine = kb * 64 + lane;
    const int JC = ((n + JP - 1) / JP + 7) & ~7, j0 = jp * JC;
    int chunks_b = min(JC, max(0, n - j0) + 7) / 8;
    const double *col_b = beta + j0 * KP + min(k_mine, K - 1), *wgt_b = tw + j0;
    const int JB = (n + 63) / 64, KPn = W / JB, jb = wid % JB, kp = wid / JB;
    const int j_mine = jb * 64 + lane;
    const int KC = ((K + KPn - 1) / KPn + 7) & ~7, k0 = kp * KC;
    int chunks_e = min(KC, max(0, K - k0) + 7) / 8;
    const double *col_e = beta + min(j_mine, n - 1) * KP + k0, *wgt_e = e + k0;
    if (wid >= active_waves) { chunks_b = 0; chunks_e = 0; }
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        if (WHICH == 0) {
            double s = lds_dot_chunks<KP>(wgt_b, col_b, chunks_b);
            if (k_mine < K) part[jp * K + k_mine] = s;
        } else {
            double s = lds_dot_chunks<1>(wgt_e, col_e, chunks_e);
            if (j_mine < n) part[kp * n + j_mine] = s;
        }
        __syncthreads();
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[tid] = part[tid % K];
    if (tid == 0) { t[0] = c1 - c0; t[1] = chunks_b; t[2] = chunks_e; }
}

template <int T, int KP, int WHICH> int run(double *out, unsigned long long *t, int active)
{
    const int K = 100, n = 100, reps = 200;
    size_t lds = (size_t)((n + 8) * KP + n + 8 + K + 8 + T + 8) * 8;
    CK(hipFuncSetAttribute((const void *)prod<T, KP, WHICH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((prod<T, KP, WHICH>), dim3(1), dim3(T), lds, 0, out, t, K, n, reps, active); CK(hipDeviceSynchronize()); }
    unsigned long long h[3]; CK(hipMemcpy(h, t, 24, hipMemcpyDeviceToHost));
    printf("T=%4d %s active_waves=%2d chunks(b,e)=(%llu,%llu): %.0f cycles per product+barrier\n", T, WHICH ? "E" : "B", active, h[1], h[2], h[0] / 200.0);
    return 0;
}

int main()
{
    double *out; unsigned long long *t;
    CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&t, 64));
    for (int act : {8, 4, 2, 1, 0}) { run<512, 101, 0>(out, t, act); }
    for (int act : {8, 4, 2, 1, 0}) { run<512, 101, 1>(out, t, act); }
    run<256, 101, 0>(out, t, 4); run<256, 101, 1>(out, t, 4);
    run<1024, 101, 0>(out, t, 16); run<1024, 101, 1>(out, t, 16);
    return 0;
}
