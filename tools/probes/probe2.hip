// Probe 2: LDS read throughput without dependent chains; readlane-broadcast FMA cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int VEC>   // 1: b64, 2: b128 (double2)
__global__ void lds_tput(double *out, unsigned long long *t, int iters, int stride)
{
    extern __shared__ double buf[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) buf[i] = i * 1e-6;
    __syncthreads();
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        int base = ((i * 8 + wid) * 128) & 8191;
        if (VEC == 1) {
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = buf[base + u * 1024 % 8192 + lane * stride % 1024];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += x[u];
        } else {
            double2 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<double2 *>(&buf[(base + u * 1024 % 8192 + lane * 2 * stride % 1024) & ~1]);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += x[u].x + x[u].y;
        }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int u = 0; u < 8; ++u) s += acc[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) t[blockIdx.x] = c1 - c0;
}

// acc[u] += bcast(v, i) * y  : broadcast by v_readlane (SGPR operand) vs by LDS same-address read
template <int MODE>
__global__ void bcast_fma(double *out, unsigned long long *t, int iters)
{
    __shared__ double sh[64];
    if (threadIdx.x < 64) sh[threadIdx.x] = 1.0 + threadIdx.x * 1e-3;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    double v = 1.0 + lane * 1e-3;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double y = 1.0 + lane * 1e-6;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            double b;
            if (MODE == 0) {
                int lo = __builtin_amdgcn_readlane(__double2loint(v), (i * 8 + u) & 63);
                int hi = __builtin_amdgcn_readlane(__double2hiint(v), (i * 8 + u) & 63);
                b = __hiloint2double(hi, lo);
            } else {
                b = sh[(i * 8 + u) & 63];
            }
            acc[u] = fma(b, y, acc[u]);
        }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int u = 0; u < 8; ++u) s += acc[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) t[blockIdx.x] = c1 - c0;
}

int main()
{
    double *out; unsigned long long *t;
    CK(hipMalloc(&out, 1 << 24)); CK(hipMalloc(&t, 1 << 16));
    unsigned long long h[4];
    CK(hipFuncSetAttribute((const void *)lds_tput<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void *)lds_tput<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    for (int threads : {64, 256, 512, 1024}) {
        for (int stride : {1, 101}) {
            hipLaunchKernelGGL(lds_tput<1>, dim3(1), dim3(threads), 131072, 0, out, t, 2000, stride); CK(hipDeviceSynchronize());
            CK(hipMemcpy(h, t, 8, hipMemcpyDeviceToHost));
            printf("b64  threads=%4d stride=%3d: %.1f cyc/wave-read, %.1f B/clk/CU\n", threads, stride, h[0] / 16000.0, threads * 8.0 * 16000 / h[0]);
        }
        hipLaunchKernelGGL(lds_tput<2>, dim3(1), dim3(threads), 131072, 0, out, t, 2000, 1); CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, t, 8, hipMemcpyDeviceToHost));
        printf("b128 threads=%4d stride=  1: %.1f cyc/wave-read, %.1f B/clk/CU\n", threads, h[0] / 16000.0, threads * 16.0 * 16000 / h[0]);
    }
    for (int threads : {64, 256, 512, 1024}) {
        hipLaunchKernelGGL(bcast_fma<0>, dim3(1), dim3(threads), 0, 0, out, t, 2000); CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, t, 8, hipMemcpyDeviceToHost));
        double a = h[0] / 16000.0;
        hipLaunchKernelGGL(bcast_fma<1>, dim3(1), dim3(threads), 0, 0, out, t, 2000); CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, t, 8, hipMemcpyDeviceToHost));
        printf("broadcast+fma threads=%4d: readlane %.1f cyc/MAC/wave ; LDS-broadcast %.1f cyc/MAC/wave\n", threads, a, h[0] / 16000.0);
    }
    return 0;
}
