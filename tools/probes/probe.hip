// Developer micro-probes for gfx950: clock, fp64 op costs, LDS latency/throughput (not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void clock_probe(double *out, unsigned long long *t, int iters)
{
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double a = threadIdx.x * 1e-3, b = 1.000001;
    for (int i = 0; i < iters; ++i) { a = fma(a, b, 1e-9); }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = c1 - c0; t[2 * blockIdx.x + 1] = r1 - r0; }
}

// dependent chain of OP per lane: cycles per op (latency); and NI independent chains (throughput)
template <int OP, int NI>
__global__ void op_probe(double *out, unsigned long long *t, int iters)
{
    double v[NI];
    for (int u = 0; u < NI; ++u) v[u] = 1.0 + threadIdx.x * 1e-3 + u * 0.1;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            if (OP == 0) v[u] = fma(v[u], 1.0000001, 1e-9);
            if (OP == 1) v[u] = __builtin_amdgcn_rcp(v[u]) + 0.5;
            if (OP == 2) v[u] = 1.0 / v[u] + 0.5;
            if (OP == 3) v[u] = log(v[u]) + 2.0;
            if (OP == 4) v[u] = exp(v[u] * 1e-3);
            if (OP == 5) v[u] = v[u] + 1.0;
        }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int u = 0; u < NI; ++u) s += v[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) t[blockIdx.x] = c1 - c0;
}

// LDS: dependent pointer chase (latency) and streaming reads (throughput per wave / per CU)
__global__ void lds_latency(int *out, unsigned long long *t, int iters)
{
    __shared__ int next[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) next[i] = (i * 17 + 5) & 1023;
    __syncthreads();
    int p = threadIdx.x;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) p = next[p];
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0) t[0] = c1 - c0;
}

template <int U>
__global__ void lds_stream(double *out, unsigned long long *t, int iters, int stride)
{
    extern __shared__ double buf[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) buf[i] = i * 1e-6;
    __syncthreads();
    double acc = 0;
    const int lane = threadIdx.x;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        double x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = buf[((i * U + u) * 64 + lane * stride) & 16383];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += x[u];
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) t[blockIdx.x] = c1 - c0;
}

__global__ void barrier_probe(unsigned long long *t, int iters)
{
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) __syncthreads();
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[0] = c1 - c0;
}

__global__ void gload_latency(const int *next, int *out, unsigned long long *t, int iters)
{
    int p = threadIdx.x;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) p = next[p];
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0) t[0] = c1 - c0;
}

int main()
{
    double *out; unsigned long long *t; int *iout;
    CK(hipMalloc(&out, 1 << 24)); CK(hipMalloc(&t, 1 << 16)); CK(hipMalloc(&iout, 1 << 16));
    std::vector<unsigned long long> h(8192);
    // warm
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(clock_probe, dim3(1024), dim3(256), 0, 0, out, t, 200000);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), t, 16, hipMemcpyDeviceToHost));
        printf("clock probe: %llu shader cycles / %llu ref ticks (100MHz) -> %.0f MHz ; fma dep chain %.2f cyc\n",
               h[0], h[1], 100.0 * h[0] / h[1], (double)h[0] / 200000);
    }
    // short kernel after idle: is the clock lower?
    hipLaunchKernelGGL(clock_probe, dim3(256), dim3(256), 0, 0, out, t, 20000);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), t, 16, hipMemcpyDeviceToHost));
    printf("short kernel clock: %.0f MHz\n", 100.0 * h[0] / h[1]);
    const char *names[] = {"fma_f64", "rcp_f64(+add)", "div_f64(+add)", "log(+add)", "exp(*mul)", "add_f64"};
#define RUNOP(OP) \
    hipLaunchKernelGGL((op_probe<OP, 1>), dim3(1), dim3(64), 0, 0, out, t, 2000); CK(hipDeviceSynchronize()); \
    CK(hipMemcpy(h.data(), t, 8, hipMemcpyDeviceToHost)); { double lat = h[0] / 2000.0; \
    hipLaunchKernelGGL((op_probe<OP, 8>), dim3(1), dim3(64), 0, 0, out, t, 2000); CK(hipDeviceSynchronize()); \
    CK(hipMemcpy(h.data(), t, 8, hipMemcpyDeviceToHost)); double thr = h[0] / 16000.0; \
    hipLaunchKernelGGL((op_probe<OP, 8>), dim3(1), dim3(256), 0, 0, out, t, 2000); CK(hipDeviceSynchronize()); \
    CK(hipMemcpy(h.data(), t, 8, hipMemcpyDeviceToHost)); double thr4 = h[0] / 16000.0; \
    hipLaunchKernelGGL((op_probe<OP, 8>), dim3(1), dim3(512), 0, 0, out, t, 2000); CK(hipDeviceSynchronize()); \
    CK(hipMemcpy(h.data(), t, 8, hipMemcpyDeviceToHost)); double thr8 = h[0] / 16000.0; \
    hipLaunchKernelGGL((op_probe<OP, 8>), dim3(1), dim3(1024), 0, 0, out, t, 2000); CK(hipDeviceSynchronize()); \
    CK(hipMemcpy(h.data(), t, 8, hipMemcpyDeviceToHost)); double thr16 = h[0] / 16000.0; \
    hipLaunchKernelGGL((op_probe<OP, 2>), dim3(1), dim3(64), 0, 0, out, t, 2000); CK(hipDeviceSynchronize()); \
    CK(hipMemcpy(h.data(), t, 8, hipMemcpyDeviceToHost)); double thr2c = h[0] / 4000.0; \
    printf("%-14s dep-chain %.1f ; 2 chains %.1f ; 8 chains 1 wave %.1f ; 1 wave/SIMD %.1f ; 2 waves/SIMD %.1f ; 4 waves/SIMD %.1f (cyc/op/wave)\n", names[OP], lat, thr2c, thr, thr4, thr8, thr16); }
    RUNOP(0) RUNOP(1) RUNOP(2) RUNOP(3) RUNOP(4) RUNOP(5)
    hipLaunchKernelGGL(lds_latency, dim3(1), dim3(64), 0, 0, iout, t, 4000); CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), t, 8, hipMemcpyDeviceToHost));
    printf("LDS dependent read latency: %.1f cyc\n", h[0] / 4000.0);
    for (int threads : {64, 256, 512, 1024}) for (int stride : {1, 101}) {
        CK(hipFuncSetAttribute((const void *)lds_stream<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        hipLaunchKernelGGL(lds_stream<8>, dim3(1), dim3(threads), 131072, 0, out, t, 1000, stride); CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), t, 8, hipMemcpyDeviceToHost));
        printf("LDS stream ds_read_b64 U=8 threads=%4d stride=%3d: %.1f cyc per wave-read ; %.1f B/clk/CU\n", threads, stride,
               h[0] / 8000.0, (double)threads * 8 * 8000 / h[0]);
    }
    for (int threads : {256, 1024}) {
        hipLaunchKernelGGL(barrier_probe, dim3(1), dim3(threads), 0, 0, t, 1000); CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), t, 8, hipMemcpyDeviceToHost));
        printf("__syncthreads threads=%d: %.1f cyc\n", threads, h[0] / 1000.0);
    }
    // global pointer chase: L2-resident (64 KB) and HBM-ish (256 MB)
    for (size_t n : {(size_t)16384, (size_t)64 << 20}) {
        std::vector<int> nx(n);
        for (size_t i = 0; i < n; ++i) nx[i] = (int)((i * 1048583ull + 12345) % n);
        int *d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, nx.data(), n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(gload_latency, dim3(1), dim3(64), 0, 0, d, iout, t, 2000); CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(gload_latency, dim3(1), dim3(64), 0, 0, d, iout, t, 2000); CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), t, 8, hipMemcpyDeviceToHost));
        printf("global dependent load latency, table %zu MB: %.0f cyc\n", n * 4 >> 20, h[0] / 2000.0);
        CK(hipFree(d));
    }
    return 0;
}
