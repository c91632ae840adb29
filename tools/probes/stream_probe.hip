// stream_probe.hip -- how fast the column-slot streaming kernels (csrc/stream_kernels.h) move a
// 400 MB table (K = 500, V = 100 000) for a few geometries; prints GB/s per variant.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/stream_probe.hip -o tools/probes/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../trlda_amd/csrc/stream_kernels.h"
using namespace trlda;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int T, int U, bool NT>
float run_rowsum(int K, int V, int Gmax, const double *lam, double *partial, int reps)
{
    const int P = K / 2, cpb = T / P;
    if (cpb < 1) return -1.f;
    long long G = (V + (long long)cpb * U - 1) / ((long long)cpb * U);
    if (G > Gmax) G = Gmax;
    const size_t lds = (size_t)cpb * K * sizeof(double);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((rowsum_stream_kernel<T, 2, U, NT>), dim3((int)G), dim3(T), lds, 0, K, V, P, cpb, lam, partial);
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((rowsum_stream_kernel<T, 2, U, NT>), dim3((int)G), dim3(T), lds, 0, K, V, P, cpb, lam, partial);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

__global__ void copy_kernel(size_t n2, const double2 *__restrict__ in, double2 *__restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += 4 * stride) {
        double2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = in[min(i + u * stride, n2 - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + u * stride < n2) out[i + u * stride] = v[u];
    }
}
__global__ void read_kernel(size_t n2, const double2 *__restrict__ in, double *__restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += 8 * stride) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = in[min(i + u * stride, n2 - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (i + u * stride < n2) ? v[u].x + v[u].y : 0.0;
    }
    if (acc == 12345.678) out[0] = acc;
}

int main()
{
    const int K = 500, V = 100000;
    const size_t KV = (size_t)K * V;
    double *lam, *lam2, *partial;
    CK(hipMalloc(&lam, KV * 8)); CK(hipMalloc(&lam2, KV * 8)); CK(hipMalloc(&partial, (size_t)4096 * K * 8));
    std::vector<double> h(KV, 1.0);
    CK(hipMemcpy(lam, h.data(), KV * 8, hipMemcpyHostToDevice));
    const double gb = KV * 8 / 1e9;
#define ROW(T, U, NT, G) { float ms = run_rowsum<T, U, NT>(K, V, G, lam, partial, 20); \
    printf("rowsum T=%4d U=%2d nt=%d Gmax=%4d  %8.1f us  %7.1f GB/s\n", T, U, (int)NT, G, ms * 1e3, gb / (ms * 1e-3)); }
    ROW(1024, 8, false, 512) ROW(1024, 8, false, 256) ROW(1024, 8, false, 1024) ROW(1024, 8, false, 2048)
    ROW(1024, 4, false, 512) ROW(1024, 4, false, 1024) ROW(1024, 4, false, 2048)
    ROW(1024, 16, false, 256) ROW(1024, 16, false, 512)
    ROW(1024, 8, true, 512) ROW(1024, 8, true, 1024) ROW(1024, 4, true, 2048)
    ROW(512, 8, false, 1024) ROW(512, 8, false, 2048) ROW(512, 4, false, 2048) ROW(512, 4, false, 4096)
    ROW(256, 8, false, 2048) ROW(256, 4, false, 4096) ROW(256, 8, true, 2048)
    // references: flat read and copy
    {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int G : {1024, 2048, 4096, 8192}) {
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(read_kernel, dim3(G), dim3(256), 0, 0, KV / 2, (const double2 *)lam, partial);
            hipEventRecord(a, 0);
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(read_kernel, dim3(G), dim3(256), 0, 0, KV / 2, (const double2 *)lam, partial);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
            printf("flat read  G=%5d x256            %8.1f us  %7.1f GB/s\n", G, ms * 1e3, gb / (ms * 1e-3));
        }
        for (int G : {2048, 8192}) {
            hipEventRecord(a, 0);
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(copy_kernel, dim3(G), dim3(256), 0, 0, KV / 2, (const double2 *)lam, (double2 *)lam2);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
            printf("flat copy  G=%5d x256            %8.1f us  %7.1f GB/s (read + write)\n", G, ms * 1e3, 2 * gb / (ms * 1e-3));
        }
        hipEventRecord(a, 0);
        for (int i = 0; i < 20; ++i) hipMemcpyAsync(lam2, lam, KV * 8, hipMemcpyDeviceToDevice, 0);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
        printf("hipMemcpy D2D                        %8.1f us  %7.1f GB/s (read + write)\n", ms * 1e3, 2 * gb / (ms * 1e-3));
    }
    return 0;
}
