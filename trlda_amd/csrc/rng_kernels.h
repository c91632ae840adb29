// rng_kernels.h -- sampleGamma (reference src/utils.cpp:224-231) on the device, from the SAME
// libc rand() stream the reference consumes.
//
// The reference draws lambda0 and every default gamma0 as -sum_{p<100} log|u_{p,i}| / 100 with
// u = -1 + 2 rand() / RAND_MAX, the (p * total + i)-th draw of glibc's TYPE_3 generator
// s_n = s_{n-31} + s_{n-3} (mod 2^32), output s_n >> 1 (see the host version in trlda_hip.hip).
// That is K * B * 100 draws and logarithms per update call -- 2 * 10^6 at K = 100, B = 200,
// 2 * 10^8 at K = 500, B = 4096 -- which the host's threads cannot produce as fast as the GPU
// consumes them.  The generator is linear: a window of 31 consecutive values advances N draws by
// a 31 x 31 matrix over Z / 2^32.  So the stream is cut into segments of kRngSegment draws:
//
//   window_level_kernel   the window at the start of every segment, W_s = A^(s L) W_0, built
//   / window_direct_kernel  radix 16: W_(d 16^l + r) = M[l][d] W_r, one matrix-vector product per
//                         segment and level (M[l][d] = A^(d 16^l L): 15 matrices per level,
//                         computed once on the host)
//   draw_abs_kernel       one thread per segment walks its L draws (window in LDS, a rotating
//                         index) and stores |u| of every draw -- the same integers, hence the
//                         same u, bit for bit, as the host stream
//   gamma_sum_kernel      out[i] = -sum over blocks of kRngProductPasses passes of
//                         log(prod_p |u_{p,i}|), and the final division by 100
//
// The logarithm of a PRODUCT (round 6): -sum_p log|u_p| is 100 logarithms per element -- 35 of
// the ~55 instructions a draw costs -- where the gamma0 of a mini-batch is 2 * 10^6 draws per
// update call.  |u| >= 1 / RAND_MAX > 2^-31, so the product of 25 draws is a normal number
// (> 2^-775) and -log(prod) needs ONE logarithm per 25 draws.  Against the exact sum the product
// form is the more accurate one (each multiplication rounds once, 25 x 1.1e-16 relative on the
// product = 2.8e-15 ABSOLUTE on its logarithm, where the reference's running sum rounds to
// ulp(100) / 2 = 7e-15 a hundred times): measured against 50-digit arithmetic the reference's own
// sum is off by up to 7.3e-16 relative, this form by 2.1e-16; the two differ by at most 1.4e-15
// over 2 * 10^5 elements (tests: < 4e-15 against the host draw, which is the reference's bit for
// bit).  The integers -- and the generator state after the call -- are the host stream's, exactly
// (the host advances it by the same number of draws).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "psi.h"

namespace trlda {

// draws per thread: 256 for large requests, 64 for small ones (a thread's draws are serial --
// ~130 instructions each at ~8.6 cycles for a wave that has little company -- so a request of
// 2 * 10^6 draws wants 31 000 threads, not 7 800)
constexpr int kRngSegment = 256;
constexpr int kRngSegmentSmall = 64;
constexpr long long kRngSmallDraws = (long long)1 << 25;
constexpr int kRngSegmentTiny = 32;   // a mini-batch of 200 documents at K = 100: 2 * 10^6 draws
constexpr long long kRngTinyDraws = (long long)1 << 22;
constexpr int kRngLevels = 8;         // 16^8 segments: more than any table needs
constexpr int kRngThreads = 256;
constexpr int kRngProductPasses = 25; // passes whose |u| are multiplied before ONE logarithm: 25 x 31 bits
                                      // of exponent stay a normal double (see above); blocks are counted
                                      // from pass 0, whatever groups of passes a request is worked off in

// u = -1 + 2 rand() / RAND_MAX (utils.cpp:224-231) from the generator's word v (rand() = v >> 1):
// the host's expression with its IEEE division -- which the compiler expands into a dozen
// instructions (scale, reciprocal, refinement, fix-up) per draw.  The divisor is a constant: with
// y = RN(1 / (2^31 - 1)), q0 = RN(x y), r = x - q0 d (exact, one fma) and q1 = RN(q0 + r y) the
// quotient is the correctly rounded one for EVERY x = 2 k, k < 2^31 -- checked exhaustively
// (tools/probes/div_check.c, tests/test_oracle.py::test_division_by_rand_max_is_exact) -- so the
// same u, bit for bit, in three instructions.
__device__ __forceinline__ double unit_draw(uint32_t v)
{
    const double d = 2147483647.0, y = 0x1.00000002p-31;        // y = RN(1 / d)
    const double x = 2.0 * (double)(v >> 1);
    const double q0 = x * y;
    const double r = fma(-q0, d, x);
    return -1.0 + fma(r, y, q0);
}

// windows are stored word-major: win[j * S + s], j < 31.
//   window_level_kernel   one level, one thread per window (961 serial multiply-adds: fine when
//                         there are hundreds of thousands of windows to keep the chip busy)
//   window_direct_kernel  every level in one launch, 32 lanes per window (below): for requests
//                         with few windows, where the chain of launches was most of the time
//                         (4 x 8.1 us of levels + 4.5 us of seeding at 31 000 windows)
template <int T>
__global__ __launch_bounds__(T) void window_level_kernel(long long S, long long lo, long long hi,
                                                         long long unit /* 16^level */,
                                                         const uint32_t *__restrict__ mats /* 15 x 31 x 31 */,
                                                         uint32_t *__restrict__ win)
{
    const long long s = lo + (long long)blockIdx.x * T + threadIdx.x;
    if (s >= hi)
        return;
    const int d = (int)(s / unit);                   // 1 .. 15
    const long long r = s - (long long)d * unit;
    const uint32_t *M = mats + (size_t)(d - 1) * 961;
    uint32_t w[31];
#pragma unroll
    for (int j = 0; j < 31; ++j)
        w[j] = win[(size_t)j * S + r];
#pragma unroll 1
    for (int i = 0; i < 31; ++i) {
        uint32_t acc = 0;
#pragma unroll
        for (int j = 0; j < 31; ++j)
            acc += M[i * 31 + j] * w[j];
        win[(size_t)i * S + s] = acc;
    }
}

// All levels in one launch, for requests with few windows (a chain of small dependent launches
// costs ~8 us each, whatever they do): 32 lanes form window s directly from the seed window,
// W_s = prod_l M[l][digit_l(s)] W_0 (powers of one matrix: any order), the 31 words handed
// around through an LDS line between the levels.
struct RngSeedWindow {
    uint32_t w[31];
};

template <int T>
__global__ __launch_bounds__(T) void window_direct_kernel(long long S /* stride of win */,
                                                          long long S0 /* windows [0, S0) */, int levels,
                                                          RngSeedWindow w0,
                                                          const uint32_t *__restrict__ mats_t,
                                                          uint32_t *__restrict__ win)
{
    __shared__ uint32_t line[2][T];                  // per 32 lanes: the window's 31 words
    const long long s = ((long long)blockIdx.x * T + threadIdx.x) / 32;
    const int i = threadIdx.x & 31, base = threadIdx.x & ~31;
    const int row = min(i, 30);
    uint32_t w = w0.w[row];
    long long rest = s;
    int buf = 0;
    for (int l = 0; l < levels; ++l) {               // (the same trip count in every thread)
        const int d = (int)(rest & 15);
        rest >>= 4;
        line[buf][threadIdx.x] = w;
        __syncthreads();
        if (d != 0) {                                // uniform over the 32 lanes of a window
            // (transposed matrices: the lanes of a window read consecutive words)
            const uint32_t *Mt = mats_t + ((size_t)l * 15 + (size_t)(d - 1)) * 961 + row;
            uint32_t acc = 0;
#pragma unroll
            for (int j = 0; j < 31; ++j)
                acc += Mt[j * 31] * line[buf][base + j];
            w = acc;
        }
        buf ^= 1;                                    // the next level writes the other line
    }
    if (s < S0 && i < 31)
        win[(size_t)i * S + s] = w;
}

// One level, 32 lanes per window, transposed matrices: W_s = M[level][s / unit] W_(s mod unit)
// for s in [lo, hi) -- ONE matrix-vector product per window, where the direct form spends one
// per level.  Requests with a few ten thousand windows take the first three levels (4096 windows)
// directly and the rest level by level: two or three launches instead of one with 4x the work.
template <int T>
__global__ __launch_bounds__(T) void window_level_coop_kernel(long long S, long long lo, long long hi,
                                                              long long unit,
                                                              const uint32_t *__restrict__ mats_t,
                                                              uint32_t *__restrict__ win)
{
    const long long s = lo + ((long long)blockIdx.x * T + threadIdx.x) / 32;
    const int i = threadIdx.x & 31;
    if (s >= hi || i >= 31)
        return;
    const int d = (int)(s / unit);                   // 1 .. 15
    const long long r = s - (long long)d * unit;
    const uint32_t *Mt = mats_t + (size_t)(d - 1) * 961 + i;
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 31; ++j)
        acc += Mt[j * 31] * win[(size_t)j * S + r];
    win[(size_t)i * S + s] = acc;
}

// Two-level windows (round 3): a matrix-vector product per window is 961 multiply-adds where
// moving a window L draws on by the recurrence itself is L additions.  So only every G-th window
// ("coarse": segments of G L draws) comes from the matrices -- the kernels above with the
// matrices of G L -- and thread c walks from coarse window c through its G fine windows in
// registers: after t steps the oldest word sits in register (t mod 31), so with the loops
// unrolled every index is a constant.  At 62 500 windows of 32 draws: 5 + 19 us of matrix
// products become ~3 (7 800 products) + ~3 (the walk).
constexpr int kRngWalk = 8;           // fine windows per coarse window

template <int T, int L, bool SEGMENT_MAJOR = false>      // (segment-major: for draw_sum_kernel below)
__global__ __launch_bounds__(T) void window_walk_kernel(long long S, long long Sc,
                                                        const uint32_t *__restrict__ cwin /* 31 x Sc */,
                                                        uint32_t *__restrict__ win /* 31 x S, or S x 32 */)
{
    const long long c = (long long)blockIdx.x * T + threadIdx.x;
    if (c >= Sc)
        return;
    uint32_t r[31];
#pragma unroll
    for (int j = 0; j < 31; ++j)
        r[j] = cwin[(size_t)j * Sc + c];
#pragma unroll
    for (int g = 0; g < kRngWalk; ++g) {
        const int rot = (g * L) % 31;                // register of the oldest word now
        const long long s = c * kRngWalk + g;
        if (s < S) {
            // (segment-major: a window is one 128-byte line.  Handing the words out through an LDS
            // tile so that a store instruction writes two whole lines instead of one word into each
            // of 64 was measured: 19.0 us against 8.0 -- 256 LDS round trips in a row; the stores
            // straight from the registers go out back to back)
            if constexpr (SEGMENT_MAJOR) {
                uint4 *line = reinterpret_cast<uint4 *>(win + (size_t)s * 32);   // eight 16-byte stores
#pragma unroll
                for (int j = 0; j < 32; j += 4)
                    line[j / 4] = make_uint4(r[(j + rot) % 31], r[(j + 1 + rot) % 31], r[(j + 2 + rot) % 31],
                                             j + 3 < 31 ? r[(j + 3 + rot) % 31] : 0u);
            } else {
#pragma unroll
                for (int j = 0; j < 31; ++j)
                    win[(size_t)j * S + s] = r[(j + rot) % 31];
            }
        }
        if (g + 1 < kRngWalk) {
#pragma unroll
            for (int t = 0; t < L; ++t) {            // s_n = s_(n-31) + s_(n-3): the oldest word
                const int f = (rot + t) % 31;        // becomes the newest
                r[f] += r[(f + 28) % 31];
            }
        }
    }
}

// positions [pos_lo, pos_hi) of the stream (a whole number of passes of `total` elements);
// vbuf[pos - pos_lo] = |u|.  A data-parallel rank needs the elements [e_lo, e_hi) of every
// pass only (its own documents' columns): segments that hold none of them do nothing (the
// others write all of their draws: vbuf is scratch).  A thread's draws are consecutive
// positions, so a wave's stores would be 64 separate 8-byte writes 8 L bytes apart: eight
// draws at a time go through an LDS tile and leave as 64-byte runs, eight per store instruction.
template <int T, int L>
__global__ __launch_bounds__(T) void draw_abs_kernel(long long S, long long seg_lo, long long seg_hi,
                                                     long long pos_lo, long long pos_hi, long long total,
                                                     long long e_lo, long long e_hi,
                                                     const uint32_t *__restrict__ win,
                                                     double *__restrict__ vbuf)
{
    __shared__ uint32_t x[31 * T];
    __shared__ double tile[T * 9];                   // per thread 8 values, rows padded to 9
    const int lane = threadIdx.x & 63;
    const long long s = seg_lo + (long long)blockIdx.x * T + threadIdx.x;
    const long long s_wave = s - lane;               // segment of the wave's lane 0
    bool on = s < seg_hi;
    const long long first = s * L;
    on = on && first < pos_hi && first + L > pos_lo;
    if (on) {
        // elements covered by the segment (it may wrap into the next pass once: total >= 1)
        const long long a = first % total, z = a + L;             // [a, z) modulo total
        on = z <= total ? (a < e_hi && z > e_lo)
                        : (a < e_hi || (z - total) > e_lo || L >= total);
    }
    const unsigned long long onmask = __ballot(on);
    if (onmask == 0)                                 // the whole wave has nothing to do
        return;
#pragma unroll
    for (int j = 0; j < 31; ++j)
        x[j * T + threadIdx.x] = on ? win[(size_t)j * S + s] : 0u;
    double *mine = tile + (size_t)threadIdx.x * 9;
    const double *wave_tile = tile + (size_t)(threadIdx.x - lane) * 9;
    int f = 0, b = 28;
    for (int q0 = 0; q0 < L; q0 += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint32_t v = x[f * T + threadIdx.x] + x[b * T + threadIdx.x];
            x[f * T + threadIdx.x] = v;
            f = f == 30 ? 0 : f + 1;
            b = b == 30 ? 0 : b + 1;
            // the host's u (unit_draw); |u| is a normal number in [4.6e-10, 1]
            mine[q] = fabs(unit_draw(v));
        }
        // (LDS traffic of one wave is in order: no barrier between the writes and these reads)
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int src = 8 * g + (lane >> 3);     // which thread's run, which of its 8 values
            const int o = lane & 7;
            const long long pos = (s_wave + src) * L + q0 + o;
            const bool src_on = (onmask >> src) & 1;
            if (src_on && pos >= pos_lo && pos < pos_hi)
                vbuf[pos - pos_lo] = wave_tile[src * 9 + o];
        }
    }
}

// draw_abs_kernel and gamma_sum_kernel in one (round 4), for requests of at most kRngFusedPasses
// passes whose fine windows come from the walk: a workgroup owns 32 consecutive ELEMENTS and all
// their passes.  Pass p of element i is draw p * total + i, so the workgroup needs, of every pass, the
// 32 draws from p * total + i0 on: those of one segment when that position is a multiple of the
// segment length, of two otherwise -- thread (h, p) walks segment h of pass p in full (a segment's
// draws come one after the other) and leaves |u| of the draws that are the workgroup's in an LDS
// tile; 32 lanes then multiply their element's passes block by block, subtract the blocks'
// logarithms and divide, as gamma_sum_kernel does: the same values in the same order, bit for bit,
// without the round trip of passes x total values through memory and without the launch.
// Windows in segment-major order here (win[s * 32 + j]: a thread's 31 words are one 124-byte run;
// threads of a wave are total / 32 segments apart).
constexpr int kRngFusedPasses = 128;

// acc - sum over the blocks of kRngProductPasses passes of log(prod |u|), passes [0, passes) at
// v[p * stride] (the FIRST of them is pass `p_first` of the request: blocks are counted from pass 0,
// so p_first is a multiple of the block length), divided by `divisor` (1: not at all).  The block's
// values are requested together and multiplied in pass order.
__device__ __forceinline__ double gamma_from_abs(const double *v, size_t stride, int passes, int p_first,
                                                 double acc, double divisor)
{
    constexpr int R = kRngProductPasses;
    (void)p_first;
    int q0 = 0;
    for (; q0 + R <= passes; q0 += R) {
        double x[R];
#pragma unroll
        for (int q = 0; q < R; ++q)
            x[q] = v[(size_t)(q0 + q) * stride];
        double prod = x[0];
#pragma unroll
        for (int q = 1; q < R; ++q)
            prod *= x[q];
        acc -= log_normal(prod);
    }
    if (q0 < passes) {
        double prod = v[(size_t)q0 * stride];
        for (int q = q0 + 1; q < passes; ++q)
            prod *= v[(size_t)q * stride];
        acc -= log_normal(prod);
    }
    return divisor != 1.0 ? acc / divisor : acc;
}

// The launch has kRngFusedPasses threads per workgroup when every pass's share of a chunk is ONE
// segment (total and e_lo multiples of the segment length), twice that otherwise; LDS (dynamic):
// 31 words of generator state per thread and the passes x 33 tile -- 42 kB at 100 passes, three
// workgroups per CU, so that the ~625 workgroups of a 200-document mini-batch are resident at once
// (a thread's 32 draws are ~4000 instructions one after the other: a second round doubles the launch).
template <int L>
__global__ __launch_bounds__(2 * kRngFusedPasses) void draw_sum_kernel(
    long long S, long long total, long long e_lo, long long e_hi, int passes, double divisor,
    const uint32_t *__restrict__ win /* S x 32 */, double *__restrict__ out)
{
    static_assert(L == 32, "an element chunk is one segment long");
    extern __shared__ __attribute__((aligned(16))) double draw_sum_lds[];
    const int T = (int)blockDim.x;                   // kRngFusedPasses or twice that
    double *vals = draw_sum_lds;                     // [pass][element of the chunk], rows padded to 33
    uint32_t *x = reinterpret_cast<uint32_t *>(vals + (size_t)passes * 33);
    const int p = threadIdx.x & (kRngFusedPasses - 1), h = threadIdx.x / kRngFusedPasses;
    const long long i0 = e_lo + (long long)blockIdx.x * L;       // first element of the chunk
    const int n_e = (int)min((long long)L, e_hi - i0);           // its elements
    const long long pos0 = (long long)p * total + i0;            // pass p's first draw of the chunk
    const long long s = pos0 / L + h;
    // draws [s L, s L + L) against [pos0, pos0 + n_e)
    const int e_first = (int)(s * L - pos0);                     // chunk element of the segment's draw 0
    const bool on = p < passes && s < S && e_first < n_e && e_first + L > 0;
    if (__ballot(on) == 0) {
        // (nothing in this wave: it still meets the others at the barrier below)
    } else {
        const uint32_t *w = win + (size_t)(on ? s : 0) * 32;
#pragma unroll
        for (int j = 0; j < 31; ++j)
            x[j * T + threadIdx.x] = on ? w[j] : 0u;
        int f = 0, b = 28;
#pragma unroll 8
        for (int q = 0; q < L; ++q) {
            const uint32_t v = x[f * T + threadIdx.x] + x[b * T + threadIdx.x];
            x[f * T + threadIdx.x] = v;
            f = f == 30 ? 0 : f + 1;
            b = b == 30 ? 0 : b + 1;
            const double u = unit_draw(v);
            const int e = e_first + q;
            if (on && e >= 0 && e < n_e)
                vals[p * 33 + e] = fabs(u);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < n_e)
        out[i0 - e_lo + threadIdx.x] = gamma_from_abs(vals + threadIdx.x, 33, passes, 0, 0.0, divisor);
}

// passes [p0, p1) of `total` elements each (p0 a multiple of kRngProductPasses), elements [e_lo, e_hi):
// out[i - e_lo] = (first ? 0 : out[i - e_lo]) - sum over the blocks of log(prod_p vbuf[p][i]); the last
// group divides by `divisor` (1 for none)
template <int T>
__global__ __launch_bounds__(T) void gamma_sum_kernel(long long total, long long e_lo, long long e_hi,
                                                      int passes, int first, double divisor,
                                                      const double *__restrict__ vbuf,
                                                      double *__restrict__ out)
{
    const long long i = e_lo + (long long)blockIdx.x * T + threadIdx.x;
    if (i >= e_hi)
        return;
    out += -e_lo;
    out[i] = gamma_from_abs(vbuf + i, (size_t)total, passes, 0, first ? 0.0 : out[i], divisor);
}

// ---------------------------------------------------------------------------------------------------
// The NEXT fresh gamma0 drawn inside a document launch (round 6; estep_merged.h, AuxArgs).
//
// An update call's document launch leaves ~50 CUs without a document for ~35 us, and the gamma0 of
// the next fresh E-step needs nothing but the generator's state -- so `n` extra workgroups of the
// launch draw it, each on its own: workgroup j owns `cpw` consecutive chunks of kAuxChunk = 4 x 31
// elements from element j * cpw * kAuxChunk on.  Nothing is handed from workgroup to workgroup:
//   1. the window at the workgroup's first element, W = A^(j stride) W0 (stride = cpw * kAuxChunk:
//      two radix-16 digits of j, the matrices of `stride`, 32 lanes);
//   2. the pass windows A^(p total) W, p < passes <= 128: V(d1) = M[1][d1] W (8 groups of 32 lanes),
//      then W_p = M[0][p & 15] V(p >> 4) (16 groups a round): ~passes + 8 matrix-vector products where
//      positions taken digit by digit would be six per window;
//   3. thread (p, h) = (tid & 127, tid >> 7) walks pass p from its window IN REGISTERS -- 31 steps
//      bring the rotating window back to where it was, so every register index is a constant:
//      h blocks of 31 skipped, then per chunk 31 draws (|u| into the LDS tile [pass][element]) and
//      93 skipped;
//   4. thread (e, g) multiplies block g of 25 passes of element e and takes its logarithm; thread e
//      subtracts the four and divides -- gamma_from_abs's sequence of operations, so the values are
//      bitwise those of draw_sum_kernel / gamma_sum_kernel.
// LDS: the tile (passes x 125 doubles), the pass windows (128 x 33 words), 8 x 32 + 32 words and
// 4 x 128 doubles: 122 kB at 100 passes, inside the document launch's 154 kB.
constexpr int kAuxBlock = 31;                     // draws per register block
constexpr int kAuxChunk = 4 * kAuxBlock;          // elements a workgroup finishes per step (4 threads a pass)
constexpr int kAuxTileStride = kAuxChunk + 1;
constexpr int kAuxMaxPasses = 4 * kRngProductPasses;   // four logarithm blocks, one per thread group

struct AuxDrawArgs {
    int n;                        // workgroups (0: no draw in this launch)
    int passes, cpw;              // passes <= kAuxMaxPasses; chunks per workgroup
    long long total;              // elements per pass (all of them are drawn)
    double divisor;
    double *out;                  // total doubles
    const uint32_t *mt_stride;    // 2 x 15 transposed matrices A^(d 16^l stride)
    const uint32_t *mt_total;     // 2 x 15 transposed matrices A^(d 16^l total)
    RngSeedWindow w0;             // the generator's window at the request's first draw
};

// acc_i = sum_j M[i][j] line[j] by lane `row` of a group of 32 (Mt: the transposed matrix, so the
// lanes of a group read consecutive words)
__device__ __forceinline__ uint32_t aux_matvec(const uint32_t *__restrict__ Mt, const uint32_t *line, int row)
{
    uint32_t m[31];
#pragma unroll
    for (int j = 0; j < 31; ++j)
        m[j] = Mt[j * 31 + row];
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < 31; ++j)
        acc += m[j] * line[j];
    return acc;
}

template <int T>
__device__ __forceinline__ void aux_draw_workgroup(const AuxDrawArgs &x, int j, double *lds)
{
    static_assert(T == 512, "four threads a pass, 128 passes");
    const int tid = threadIdx.x;
    const int row = min(tid & 31, 30), grp = tid >> 5;
    double *tile = lds;                                                        // passes x kAuxTileStride
    double *lgs = tile + (size_t)kAuxMaxPasses * kAuxTileStride;               // 4 x 128
    uint32_t *pwin = reinterpret_cast<uint32_t *>(lgs + 4 * 128);              // 128 x 33
    uint32_t *vwin = pwin + 128 * 33;                                          // 8 x 32
    uint32_t *base = vwin + 8 * 32;                                            // 32
    const long long e0 = (long long)j * x.cpw * kAuxChunk;
    // 1. W = A^(j stride) W0
    if (grp == 0) {                                  // (the 32 lanes of a group are one half of a wave:
        uint32_t w = x.w0.w[row];                    //  LDS traffic of a wave is in order, no barrier)
        const int d0 = j & 15, d1 = (j >> 4) & 15;
        if (d0) {
            base[tid] = w;
            __builtin_amdgcn_wave_barrier();
            w = aux_matvec(x.mt_stride + (size_t)(d0 - 1) * 961, base, row);
            __builtin_amdgcn_wave_barrier();
        }
        if (d1) {
            base[tid] = w;
            __builtin_amdgcn_wave_barrier();
            w = aux_matvec(x.mt_stride + (size_t)(15 + d1 - 1) * 961, base, row);
            __builtin_amdgcn_wave_barrier();
        }
        base[tid] = w;
    }
    __syncthreads();
    // 2. V(d1) = A^(16 d1 total) W, then W_p = A^((p & 15) total) V(p >> 4)
    if (grp < 8) {
        const uint32_t w = grp ? aux_matvec(x.mt_total + (size_t)(15 + grp - 1) * 961, base, row) : base[row];
        vwin[grp * 32 + (tid & 31)] = w;
    }
    __syncthreads();
    for (int p = grp; p < x.passes; p += T / 32) {
        const int d0 = p & 15, d1 = p >> 4;
        const uint32_t w = d0 ? aux_matvec(x.mt_total + (size_t)(d0 - 1) * 961, vwin + d1 * 32, row)
                              : vwin[d1 * 32 + row];
        if ((tid & 31) < 31)
            pwin[p * 33 + (tid & 31)] = w;
    }
    __syncthreads();
    // 3. the walk: thread (p, h)
    const int p = tid & 127, h = tid >> 7;
    const bool on = p < x.passes;
    uint32_t r[31];
#pragma unroll
    for (int q = 0; q < 31; ++q)
        r[q] = on ? pwin[p * 33 + q] : 0u;
    for (int s = 0; s < h; ++s) {
#pragma unroll
        for (int t = 0; t < kAuxBlock; ++t)          // s_n = s_(n-31) + s_(n-3): the oldest word becomes the newest
            r[t] += r[(t + 28) % 31];
    }
    for (int c = 0; c < x.cpw; ++c) {                // block-uniform
        const long long ec = e0 + (long long)c * kAuxChunk;
        if (ec >= x.total)
            break;
        const int n_e = (int)min((long long)kAuxChunk, x.total - ec);
#pragma unroll
        for (int t = 0; t < kAuxBlock; ++t) {
            r[t] += r[(t + 28) % 31];
            const int e = kAuxBlock * h + t;
            if (on && e < n_e)
                tile[(size_t)p * kAuxTileStride + e] = fabs(unit_draw(r[t]));
        }
        __syncthreads();
        // 4. thread (e, g): block g of the element's passes
        {
            const int e = tid & 127, g = tid >> 7;
            const int q0 = g * kRngProductPasses, qn = min(kRngProductPasses, x.passes - q0);
            if (e < n_e && qn > 0) {
                const double *v = tile + (size_t)q0 * kAuxTileStride + e;
                double prod;
                if (qn == kRngProductPasses) {
                    double xv[kRngProductPasses];
#pragma unroll
                    for (int q = 0; q < kRngProductPasses; ++q)
                        xv[q] = v[(size_t)q * kAuxTileStride];
                    prod = xv[0];
#pragma unroll
                    for (int q = 1; q < kRngProductPasses; ++q)
                        prod *= xv[q];
                } else {
                    prod = v[0];
                    for (int q = 1; q < qn; ++q)
                        prod *= v[(size_t)q * kAuxTileStride];
                }
                lgs[g * 128 + e] = log_normal(prod);
            }
        }
        // (the walk to the next chunk does not touch the tile: it runs under the others' logarithms)
        if (c + 1 < x.cpw) {
            for (int s = 0; s < 3; ++s) {
#pragma unroll
                for (int t = 0; t < kAuxBlock; ++t)
                    r[t] += r[(t + 28) % 31];
            }
        }
        __syncthreads();
        if (tid < n_e) {
            double acc = 0.0;
            const int nb = (x.passes + kRngProductPasses - 1) / kRngProductPasses;
            for (int g = 0; g < nb; ++g)             // (in block order, as gamma_from_abs subtracts them)
                acc -= lgs[g * 128 + tid];
            x.out[ec + tid] = x.divisor != 1.0 ? acc / x.divisor : acc;
        }
        // (the next chunk's draws write the tile: every thread has read its products above, behind
        // the barrier; lgs is rewritten only behind the next chunk's first barrier)
    }
}

}  // namespace trlda
