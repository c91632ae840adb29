// anyorder_probe -- can a kernel that FOLLOWS another one in the same stream start while the first
// still runs (hipExtAnyOrderLaunch: the AQL packet without its barrier bit), so that a dependency
// between them can be carried by a counter in memory instead of the kernel boundary?
//
//   A: one workgroup polls a flag (agent-scope loads) for at most ~20 ms, reports what it saw
//   B: sets the flag
// ordinary launches: B starts when A has ended -> A never sees the flag (control);
// B launched with hipExtAnyOrderLaunch: A sees it iff the two overlap;
// B on a second stream with no event between them (control: overlap is certain).
// Also: how long a dependent pair of empty kernels takes with and without the flag.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>

__global__ void wait_kernel(const int *flag, long long *out, long long limit_cycles)
{
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    long long spins = 0;
    int seen = 0;
    while (true) {
        seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ++spins;
        if (seen || (long long)__builtin_amdgcn_s_memtime() - t0 > limit_cycles)
            break;
        __builtin_amdgcn_s_sleep(8);
    }
    out[0] = seen;
    out[1] = spins;
    out[2] = (long long)__builtin_amdgcn_s_memtime() - t0;
}
__global__ void set_kernel(int *flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void empty_kernel(int *p) { if (p && threadIdx.x == 9999) *p = 1; }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main()
{
    int *flag; long long *out;
    CK(hipMalloc(&flag, 4)); CK(hipMalloc(&out, 24));
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const long long limit = 2000000;                 // s_memtime ticks (100 MHz): 20 ms
    const char *names[3] = {"ordinary launch (control)", "hipExtAnyOrderLaunch", "second stream, no event (control)"};
    for (int mode = 0; mode < 3; ++mode) {
        CK(hipMemset(flag, 0, 4)); CK(hipMemset(out, 0, 24));
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(wait_kernel, dim3(1), dim3(64), 0, s, flag, out, limit);
        if (mode == 0)
            hipLaunchKernelGGL(set_kernel, dim3(1), dim3(64), 0, s, flag);
        else if (mode == 1)
            hipExtLaunchKernelGGL(set_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, flag);
        else
            hipLaunchKernelGGL(set_kernel, dim3(1), dim3(64), 0, s2, flag);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        long long h[3];
        CK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
        printf("%-36s waiter saw the flag: %lld  (polls %lld, %.1f us)\n", names[mode], h[0], h[1], h[2] / 100.0);
    }
    // cost of a dependent pair of empty kernels, 2000 pairs
    for (int mode = 0; mode < 2; ++mode) {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 2000; ++i) {
            hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, s, flag);
            if (mode == 0)
                hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(1024), 0, s, flag);
            else
                hipExtLaunchKernelGGL(empty_kernel, dim3(256), dim3(1024), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, flag);
        }
        CK(hipDeviceSynchronize());
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("pair of empty kernels, second one %s: %.2f us per pair\n", mode ? "any-order" : "ordinary", us / 2000);
    }
    return 0;
}
