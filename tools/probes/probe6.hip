// Probe 6: latency of the psi pieces / exp / combine as used by the document kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../trlda_amd/csrc/psi.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using namespace trlda;

template <int WHICH>
__global__ void probe(double *out, unsigned long long *t, int iters)
{
    double x = 0.3 + threadIdx.x * 0.37;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        double v;
        if (WHICH == 0) v = psi_recurrence_piece<0, 4>(x);
        if (WHICH == 1) v = psi_recurrence_piece<8, 11>(x);
        if (WHICH == 2) v = psi_series_piece(x);
        if (WHICH == 3) v = psi_piece<4>(x, 0);
        if (WHICH == 4) v = psi_piece<4>(x, 3);
        if (WHICH == 5) v = exp(-x) ;
        if (WHICH == 6) v = exp_psi_from_pieces(x, -x);
        if (WHICH == 7) v = exp_psi_regular(x, 0.0);
        if (WHICH == 8) v = rcp_pos<true>(x);
        if (WHICH == 9) v = psi_is_regular(x) ? x * 1.0000001 : 0.5;
        x = x + v * 1e-9;   // dependent
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) t[0] = c1 - c0;
}

template <int WHICH> int run(const char *name, double *out, unsigned long long *t)
{
    for (int threads : {64, 512}) {
        hipLaunchKernelGGL(probe<WHICH>, dim3(1), dim3(threads), 0, 0, out, t, 1000); CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(probe<WHICH>, dim3(1), dim3(threads), 0, 0, out, t, 1000); CK(hipDeviceSynchronize());
        unsigned long long h; CK(hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost));
        printf("%-28s threads=%3d: %.0f cycles per dependent call\n", name, threads, h / 1000.0);
    }
    return 0;
}

int main()
{
    double *out; unsigned long long *t;
    CK(hipMalloc(&out, 1 << 16)); CK(hipMalloc(&t, 64));
    run<0>("recurrence<0,4>", out, t); run<1>("recurrence<8,11>", out, t); run<2>("series piece", out, t);
    run<3>("psi_piece<4>(x,0)", out, t); run<4>("psi_piece<4>(x,3)", out, t); run<5>("exp", out, t);
    run<6>("exp_psi_from_pieces", out, t); run<7>("exp_psi_regular (whole)", out, t); run<8>("rcp_pos", out, t);
    run<9>("psi_is_regular", out, t);
    return 0;
}
