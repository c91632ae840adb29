// xcu_probe -- what an exchange of K doubles per iteration between C workgroups costs when it goes
// through global memory (workgroups of one launch land on different XCDs round-robin, so nothing
// short of agent-scope -- sc1 -- accesses is visible across them).
//
// Every iteration each workgroup publishes 128 doubles and needs the 128 doubles of every other
// workgroup before it goes on (the shape of "one long document on several CUs": per-iteration
// partial sums of acc_k).  Two protocols, slots indexed by iteration (never reused in a launch):
//   A  data, then a per-workgroup flag (release), consumers poll the flags, then load the data
//   B  no flag: the buffer starts as a NaN sentinel and consumers poll the data itself
//   C  B with the loads of all rows in flight together (what the kernel does)
//   D  C with workgroup-scope accesses: the XCD's L2 is the meeting point -- right only when the
//      workgroups sit on ONE XCD (stride 8, if workgroups are dealt round-robin over the XCDs);
//      with stride 1 it is expected to time out or read stale rows: that is the measurement
//   E  D with the reader's L1 invalidated before every poll (buffer_inv sc0): the loads miss the CU's
//      L1 and are served by the XCD's L2, where a same-XCD writer's store already is
//   F  stores as in C (agent scope: through L2 to memory), every poll BOTH ways -- E's L2-level
//      load and C's agent-scope load in flight together; right wherever the workgroups sit, as fast
//      as E when they share an XCD
//   G  stores as in C, the poll: buffer_inv sc1 (L1 and the L2's possibly-stale lines), then loads that
//      may hit L2
// and, first, which XCD (HW_REG_XCC_ID) the workgroups of a launch land on.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/xcu_probe.hip -o tools/probes/xcu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

constexpr int K = 128, IT = 200;

// workgroup-scope accesses (sc0): served by the XCD's L2 -- coherent between the CUs of ONE XCD
// only, which is more than the scope promises and less than workgroups on different XCDs need
__device__ __forceinline__ void st_wg(double *p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ double ld_wg(const double *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void st_agent(double *p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ double ld_l2(const double *p)
{
    // invalidate this CU's L1, then a workgroup-scope load: L1 miss, L2 hit allowed
    __asm__ volatile("buffer_inv sc0" ::: "memory");
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ void where_kernel(int *xcc)
{
    if (threadIdx.x == 0)
        xcc[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15;   // HW_REG_XCC_ID[3:0]
}

template <int PROTO>
__global__ __launch_bounds__(512) void exchange(int C, int stride, double *buf /* IT x C x K */,
                                                unsigned int *flags /* IT x C */, double *out,
                                                unsigned long long *cyc, int *fail)
{
    // only every `stride`-th workgroup takes part (stride 8: all members on ONE XCD, if the
    // dispatcher deals workgroups round-robin over the 8 XCDs; stride 1: members on C XCDs)
    if (blockIdx.x % stride != 0 || (int)(blockIdx.x / stride) >= C)
        return;
    const int me = blockIdx.x / stride, tid = threadIdx.x;
    double acc = me + 1.0;
    __shared__ double sh[K];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < IT; ++it) {
        double *slot = buf + ((size_t)it * C + me) * K;
        if (tid < K) {
            if (PROTO == 3 || PROTO == 4)
                st_wg(slot + tid, acc + tid * 1e-3 + it);
            else
                st_agent(slot + tid, acc + tid * 1e-3 + it);      // never NaN
        }
        if (PROTO == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);                        // the stores have left
            __syncthreads();
            if (tid == 0)
                __hip_atomic_store(flags + (size_t)it * C + me, 1u, __ATOMIC_RELEASE,
                                   __HIP_MEMORY_SCOPE_AGENT);
            if (tid < C) {                                        // lane c polls workgroup c's flag
                int spins = 0;
                while (__hip_atomic_load(flags + (size_t)it * C + tid, __ATOMIC_ACQUIRE,
                                         __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                    if (++spins > (1 << 22)) { *fail = 1; break; }
                }
            }
            __syncthreads();
        }
        if (PROTO >= 2) {
            // B with all rows' loads in flight together; 2: agent scope, 3: workgroup scope (L2)
            double sum = 0.0;
            if (tid < K) {
                double v[8];
                unsigned need = (1u << C) - 1u;
                int spins = 0;
                while (need) {
                    double v2[8];
                    if (PROTO == 4 || PROTO == 5)
                        __asm__ volatile("buffer_inv sc0" ::: "memory");
                    if (PROTO == 6)
                        __asm__ volatile("buffer_inv sc1" ::: "memory");
                    for (int c = 0; c < 8; ++c)
                        if (need >> c & 1u) {
                            const double *p = buf + ((size_t)it * C + c) * K + tid;
                            v[c] = PROTO == 3 || PROTO >= 4 ? ld_wg(p) : ld_agent(p);
                            if (PROTO == 5)
                                v2[c] = ld_agent(p);
                        }
                    for (int c = 0; c < 8; ++c)
                        if (need >> c & 1u) {
                            if (PROTO == 5 && v[c] != v[c])
                                v[c] = v2[c];
                            if (v[c] == v[c])
                                need &= ~(1u << c);
                        }
                    if (++spins > (1 << 14)) { *fail = 1; break; }
                }
                for (int c = 0; c < C; ++c)
                    sum += v[c];
                sh[tid] = sum;
            }
            __syncthreads();
            if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                break;                                            // somebody gave up: all leave
            acc = sh[(tid + it) % K] * 1e-3 + me;
            continue;
        }
        // every thread k < K adds up column k over the C workgroups, in order
        double sum = 0.0;
        if (tid < K) {
            for (int c = 0; c < C; ++c) {
                const double *p = buf + ((size_t)it * C + c) * K + tid;
                double v = ld_agent(p);
                if (PROTO == 1) {
                    int spins = 0;
                    while (v != v) {                              // sentinel: not written yet
                        if (++spins > (1 << 22)) { *fail = 1; break; }
                        v = ld_agent(p);
                    }
                }
                sum += v;
            }
            sh[tid] = sum;
        }
        __syncthreads();
        acc = sh[(tid + it) % K] * 1e-3 + me;                     // the next round depends on this one
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
        cyc[me] = t1 - t0;
        out[me] = acc;
    }
}

int main()
{
    double *buf, *out;
    unsigned int *flags;
    unsigned long long *cyc;
    int *fail;
    const int CMAX = 8;
    hipMalloc(&buf, sizeof(double) * IT * CMAX * K);
    hipMalloc(&flags, sizeof(unsigned) * IT * CMAX);
    hipMalloc(&out, 8 * CMAX);
    hipMalloc(&cyc, 8 * CMAX);
    hipMalloc(&fail, 4);
    {
        int *xcc;
        hipMalloc(&xcc, 4 * 64);
        for (int rep = 0; rep < 3; ++rep) {
            int h[64];
            hipLaunchKernelGGL(where_kernel, dim3(rep == 1 ? 37 : 64), dim3(512), 0, 0, xcc);
            hipMemcpy(h, xcc, 4 * 64, hipMemcpyDeviceToHost);
            printf("XCC_ID of workgroups 0..31 (launch %d):", rep);
            for (int i = 0; i < 32; ++i)
                printf(" %d", h[i]);
            printf("\n");
        }
    }
    for (int proto = 0; proto < 7; ++proto)
        for (int stride : {1, 8})
            for (int C : {1, 2, 3, 5, 8}) {
                if (proto == 0 || proto == 1 || ((proto == 3 || proto == 4) && C > 1 && C < 8))
                    continue;                                     // (measured before; D is known to time out)
                hipMemset(buf, 0xFF, sizeof(double) * IT * CMAX * K);   // NaN sentinel
                hipMemset(flags, 0, sizeof(unsigned) * IT * CMAX);
                hipMemset(fail, 0, 4);
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                hipDeviceSynchronize();
                hipEventRecord(e0, 0);
                if (proto == 0)
                    hipLaunchKernelGGL(exchange<0>, dim3(C * stride), dim3(512), 0, 0, C, stride, buf, flags, out, cyc, fail);
                else if (proto == 1)
                    hipLaunchKernelGGL(exchange<1>, dim3(C * stride), dim3(512), 0, 0, C, stride, buf, flags, out, cyc, fail);
                else if (proto == 2)
                    hipLaunchKernelGGL(exchange<2>, dim3(C * stride), dim3(512), 0, 0, C, stride, buf, flags, out, cyc, fail);
                else if (proto == 3)
                    hipLaunchKernelGGL(exchange<3>, dim3(C * stride), dim3(512), 0, 0, C, stride, buf, flags, out, cyc, fail);
                else if (proto == 4)
                    hipLaunchKernelGGL(exchange<4>, dim3(C * stride), dim3(512), 0, 0, C, stride, buf, flags, out, cyc, fail);
                else if (proto == 6)
                    hipLaunchKernelGGL(exchange<6>, dim3(C * stride), dim3(512), 0, 0, C, stride, buf, flags, out, cyc, fail);
                else
                    hipLaunchKernelGGL(exchange<5>, dim3(C * stride), dim3(512), 0, 0, C, stride, buf, flags, out, cyc, fail);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                float ms = 0.f;
                hipEventElapsedTime(&ms, e0, e1);
                unsigned long long h[CMAX];
                int f = 0;
                hipMemcpy(h, cyc, 8 * C, hipMemcpyDeviceToHost);
                hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
                unsigned long long worst = 0;
                for (int c = 0; c < C; ++c)
                    worst = h[c] > worst ? h[c] : worst;
                fflush(stdout);
                printf("protocol %c  stride %d  C=%d workgroups: %6.0f s_memtime ticks, %6.0f ns per exchange "
                       "(launch %.1f us)%s\n", "ABCDEFG"[proto], stride, C, (double)worst / IT,
                       1e6 * ms / IT, 1e3 * ms, f ? "  [SPIN LIMIT HIT]" : "");
            }
    return 0;
}
