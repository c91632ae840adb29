// Probe: global -> LDS loads that bypass the registers (global_load_lds_dword / _dwordx4 on gfx950),
// the primitive a document workgroup would need to fetch the NEXT document's rows of exp(psi(lambda))
// into the idle transposition buffer while its registers are full.  Checks the layout
// (lane l's bytes land at M0-base + l * size), rows at an odd stride of doubles with dword loads, and
// times 128 rows x 100 doubles per workgroup against ordinary loads through registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

constexpr int K = 100, ROWS = 128, STRIDE = 129;

__global__ __launch_bounds__(512) void direct_kernel(const double *__restrict__ src, const int *__restrict__ ids,
                                                     double *__restrict__ out, int reps)
{
    extern __shared__ double tile[];                 // ROWS x STRIDE
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int rep = 0; rep < reps; ++rep) {
        for (int r = wid; r < ROWS; r += 8) {        // a wave per row: four dword loads of 64 lanes
            const char *row = reinterpret_cast<const char *>(src + (size_t)ids[(r + rep) % ROWS] * K);
            char *dst = reinterpret_cast<char *>(tile + (size_t)r * STRIDE);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int byte = c * 256 + lane * 4;
                if (byte < K * 8)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(row + byte),
                        (__attribute__((address_space(3))) void *)(dst + c * 256), 4, 0, 0);
            }
        }
        __builtin_amdgcn_s_waitcnt(0);               // vmcnt(0): the loads have landed in LDS
        __syncthreads();
    }
    for (int i = threadIdx.x; i < ROWS * K; i += 512)
        out[(size_t)blockIdx.x * ROWS * K + i] = tile[(size_t)(i / K) * STRIDE + i % K];
}

__global__ __launch_bounds__(512) void register_kernel(const double *__restrict__ src, const int *__restrict__ ids,
                                                       double *__restrict__ out, int reps)
{
    extern __shared__ double tile[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int rep = 0; rep < reps; ++rep) {
        double v0[16], v1[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const double *row = src + (size_t)ids[(wid + 8 * q + rep) % ROWS] * K;
            v0[q] = row[min(lane, K - 1)];
            v1[q] = row[min(lane + 64, K - 1)];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            tile[(size_t)(wid + 8 * q) * STRIDE + lane] = v0[q];
            if (lane + 64 < K)
                tile[(size_t)(wid + 8 * q) * STRIDE + lane + 64] = v1[q];
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < ROWS * K; i += 512)
        out[(size_t)blockIdx.x * ROWS * K + i] = tile[(size_t)(i / K) * STRIDE + i % K];
}

int main()
{
    const int V = 4000, WG = 200;
    std::vector<double> h((size_t)V * K);
    for (size_t i = 0; i < h.size(); ++i)
        h[i] = (double)i + 0.25;
    std::vector<int> ids(ROWS);
    for (int r = 0; r < ROWS; ++r)
        ids[r] = (r * 37 + 11) % V;
    double *src, *out;
    int *dids;
    hipMalloc(&src, h.size() * 8);
    hipMalloc(&out, (size_t)WG * ROWS * K * 8);
    hipMalloc(&dids, ROWS * 4);
    hipMemcpy(src, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dids, ids.data(), ROWS * 4, hipMemcpyHostToDevice);
    const size_t lds = (size_t)ROWS * STRIDE * 8;
    hipFuncSetAttribute(reinterpret_cast<const void *>(direct_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void *>(register_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    std::vector<double> got((size_t)ROWS * K);
    for (int which = 0; which < 2; ++which) {
        hipMemset(out, 0, (size_t)WG * ROWS * K * 8);
        if (which == 0)
            hipLaunchKernelGGL(direct_kernel, dim3(WG), dim3(512), lds, 0, src, dids, out, 1);
        else
            hipLaunchKernelGGL(register_kernel, dim3(WG), dim3(512), lds, 0, src, dids, out, 1);
        hipDeviceSynchronize();
        hipMemcpy(got.data(), out + (size_t)(WG - 1) * ROWS * K, got.size() * 8, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int r = 0; r < ROWS; ++r)
            for (int k = 0; k < K; ++k)
                bad += got[(size_t)r * K + k] != h[(size_t)ids[r] * K + k];
        printf("%s: %ld of %d elements wrong\n", which == 0 ? "LDS-direct loads (dword, row stride 129 doubles)" : "loads through registers", bad, ROWS * K);
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        const int reps = 50;
        for (int t = 0; t < 2; ++t) {
            hipEventRecord(a, 0);
            if (which == 0)
                hipLaunchKernelGGL(direct_kernel, dim3(WG), dim3(512), lds, 0, src, dids, out, reps);
            else
                hipLaunchKernelGGL(register_kernel, dim3(WG), dim3(512), lds, 0, src, dids, out, reps);
            hipEventRecord(b, 0);
            hipEventSynchronize(b);
        }
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        printf("   %d workgroups x %d fills of 128 x 100 doubles: %.2f us per fill\n", WG, reps, 1e3 * ms / reps);
    }
    return 0;
}
