// Probe 4: decompose the cost of the padded dot-product loop (T threads, 32 MACs per thread).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: full (wgt broadcast from LDS + strided col from LDS + fma)
// MODE 1: col from LDS only (weight = constant register)
// MODE 2: wgt broadcast only (col value = register)
// MODE 3: fma only
template <int MODE>
__device__ __forceinline__ double dotp(const double *wgt, const double *col, int stride, int first, int last, int chunks, double reg)
{
    double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    for (int c = 0; c < chunks; ++c) {
        const int i0 = first + 8 * c;
        double x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = (MODE == 0 || MODE == 2) ? wgt[i0 + u] : reg + u;
            y[u] = (MODE == 0 || MODE == 1) ? col[min(i0 + u, last) * stride] : reg * u;
        }
        acc0 = fma(x[0], y[0], acc0); acc1 = fma(x[1], y[1], acc1); acc2 = fma(x[2], y[2], acc2); acc3 = fma(x[3], y[3], acc3);
        acc0 = fma(x[4], y[4], acc0); acc1 = fma(x[5], y[5], acc1); acc2 = fma(x[6], y[6], acc2); acc3 = fma(x[7], y[7], acc3);
    }
    return (acc0 + acc1) + (acc2 + acc3);
}

template <int T, int MODE, bool BARRIER>
__global__ __launch_bounds__(T) void prod(double *out, unsigned long long *t, int K, int n, int Kp, int reps, int chunks)
{
    extern __shared__ double lds[];
    double *beta = lds, *tw = beta + n * Kp, *part = tw + n + 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < n * Kp; i += T) beta[i] = 1.0 + i * 1e-6;
    for (int i = tid; i < n + 256; i += T) tw[i] = i < n ? 1.0 + i * 1e-3 : 0.0;
    __syncthreads();
    const int kb = wid & 1, jp = wid >> 1;
    const int k = kb * 64 + lane;
    const int j0 = jp * chunks * 8;
    double reg = 1.0 + lane * 1e-9;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; ++r) {
        double s = dotp<MODE>(tw, beta + min(k, K - 1), Kp, j0, n - 1, chunks, reg);
        if (k < K) part[jp * K + k] = s;
        reg += s * 1e-30;
        if (BARRIER) __syncthreads();
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[tid] = part[tid % K] + reg;
    if (tid == 0) t[0] = c1 - c0;
}

template <int T, int MODE, bool BARRIER> int run(double *out, unsigned long long *t, int chunks)
{
    const int K = 100, n = 100, Kp = 101, reps = 200;
    size_t lds = (size_t)(n * Kp + n + 256 + 2048) * 8;
    CK(hipFuncSetAttribute((const void *)prod<T, MODE, BARRIER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((prod<T, MODE, BARRIER>), dim3(1), dim3(T), lds, 0, out, t, K, n, Kp, reps, chunks); CK(hipDeviceSynchronize()); }
    unsigned long long h; CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost));
    printf("T=%4d MODE=%d barrier=%d chunks=%d: %.0f cycles per product (%.1f per MAC)\n", T, MODE, (int)BARRIER, chunks, h / 200.0, h / 200.0 / (8 * chunks));
    return 0;
}

int main()
{
    double *out; unsigned long long *t;
    CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&t, 64));
    run<512, 0, true>(out, t, 4); run<512, 1, true>(out, t, 4); run<512, 2, true>(out, t, 4); run<512, 3, true>(out, t, 4);
    run<512, 0, false>(out, t, 4); run<512, 3, false>(out, t, 4);
    run<256, 0, true>(out, t, 7); run<256, 3, true>(out, t, 7);
    run<1024, 0, true>(out, t, 2); run<1024, 3, true>(out, t, 2);
    run<64, 0, false>(out, t, 4); run<64, 1, false>(out, t, 4); run<64, 2, false>(out, t, 4); run<64, 3, false>(out, t, 4);
    return 0;
}
