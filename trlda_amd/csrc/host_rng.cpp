// host_rng.cpp -- the host side of trlda's random numbers (host only; see host_common.h).
//
// The reference draws lambda0 and every default gamma0 from libc rand() through
// Eigen::Random (src/utils.cpp:224-231, Eigen/src/Core/MathFunctions.h:439-446).  glibc's
// rand() is the TYPE_3 additive-feedback generator of random_r.c (x[i] = x[i-3] + x[i-31],
// output x >> 1) behind a lock that costs ~20 ns per call -- 2*10^6 calls per default gamma0 at
// K=100, B=200.  The same recurrence is reproduced here without the lock (the stream is
// checked against libc's in tests/test_boundary.py), and the logarithms -- glibc's own
// log(), so the values stay bit-identical -- are taken by a few threads over disjoint
// elements, each element accumulating its passes in order.
#include "host_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <time.h>

#include "../../include/trlda_hip.h"

namespace trlda_host {

namespace {


struct GlibcRandom {
    uint32_t x[31];
    int f = 3, b = 0;
    void seed(unsigned int s)
    {
        // srandom_r, TYPE_3
        int32_t word = s == 0 ? 1 : (int32_t)s;
        x[0] = (uint32_t)word;
        for (int i = 1; i < 31; ++i) {
            const long hi = word / 127773, lo = word % 127773;
            long w = 16807 * lo - 2836 * hi;
            if (w < 0)
                w += 2147483647;
            word = (int32_t)w;
            x[i] = (uint32_t)word;
        }
        f = 3;
        b = 0;
        for (int i = 0; i < 310; ++i)
            (void)next();
    }
    inline uint32_t next()
    {
        x[f] += x[b];
        const uint32_t out = x[f] >> 1;
        if (++f == 31)
            f = 0;
        if (++b == 31)
            b = 0;
        return out;
    }
};

}  // namespace

// ---- jump-ahead for the generator above ---------------------------------------------
// The unshifted sequence obeys s_n = s_{n-31} + s_{n-3} (mod 2^32): the window
// W_n = (s_{n-31} .. s_{n-1}) advances by a 31 x 31 companion matrix A over Z / 2^32, and
// A^N (square and multiply, cached per N) jumps N draws ahead.  sampleGamma's K*B*100 draws
// are consumed pass by pass (utils.cpp:224-231); with the jumps every host thread produces
// the draws of its own element range for all passes -- the same numbers in the same order of
// additions as one serial stream, so seeded trajectories stay bit-identical.
void jump_identity(JumpMatrix &m)
{
    std::memset(m.a, 0, sizeof(m.a));
    for (int i = 0; i < 31; ++i)
        m.a[i][i] = 1;
}

void jump_multiply(const JumpMatrix &x, const JumpMatrix &y, JumpMatrix &out)
{
    for (int i = 0; i < 31; ++i) {
        uint32_t row[31] = {0};
        for (int k = 0; k < 31; ++k) {
            const uint32_t xik = x.a[i][k];
            if (xik == 0)
                continue;
            for (int j = 0; j < 31; ++j)
                row[j] += xik * y.a[k][j];
        }
        std::memcpy(out.a[i], row, sizeof(row));
    }
}

// Returned BY VALUE (3.8 kB): the cache below evicts, and a caller holds several powers at once
// and hands them to worker threads -- a reference into the map would dangle after an eviction.
// `half`: A^(n/2) when the caller has it (n even): one squaring instead of the whole ladder.
JumpMatrix jump_power(uint64_t n, const JumpMatrix *half)
{
    static std::mutex mu;
    static std::map<uint64_t, JumpMatrix> cache;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(n);
    if (it != cache.end())
        return it->second;
    JumpMatrix base, result, tmp;
    if (half && n % 2 == 0) {
        jump_multiply(*half, *half, result);
        if (cache.size() > 2048)
            cache.clear();
        cache.emplace(n, result);
        return result;
    }
    std::memset(base.a, 0, sizeof(base.a));
    for (int i = 0; i < 30; ++i)
        base.a[i][i + 1] = 1;                        // shift
    base.a[30][0] = 1;                               // s_n = s_{n-31} + s_{n-3}
    base.a[30][28] = 1;
    jump_identity(result);
    for (uint64_t e = n; e; e >>= 1) {
        if (e & 1) {
            jump_multiply(result, base, tmp);
            result = tmp;
        }
        jump_multiply(base, base, tmp);
        base = tmp;
    }
    if (cache.size() > 2048)                         // 8 MB
        cache.clear();
    cache.emplace(n, result);
    return result;
}

namespace {

// window (oldest first) <-> the circular buffer of GlibcRandom
void rng_to_window(const GlibcRandom &g, uint32_t (&w)[31])
{
    for (int j = 0; j < 31; ++j)
        w[j] = g.x[(g.f + j) % 31];
}
void window_to_rng(const uint32_t (&w)[31], GlibcRandom &g)
{
    g.f = 3;
    g.b = 0;
    for (int j = 0; j < 31; ++j)
        g.x[(3 + j) % 31] = w[j];
}
}  // namespace

void jump_apply(const JumpMatrix &m, uint32_t (&w)[31])
{
    uint32_t out[31];
    for (int i = 0; i < 31; ++i) {
        uint32_t acc = 0;
        for (int j = 0; j < 31; ++j)
            acc += m.a[i][j] * w[j];
        out[i] = acc;
    }
    std::memcpy(w, out, sizeof(out));
}

namespace {

GlibcRandom g_rng;

struct RngInit {
    RngInit()
    {
        // module import seeds with the clock (python/src/module.cpp:356-359)
        timespec t;
        clock_gettime(CLOCK_REALTIME, &t);
        const unsigned int s = (unsigned int)((t.tv_nsec / 1000) * t.tv_sec);
        srand(s);
        g_rng.seed(s);
    }
} g_rng_init;

}  // namespace

// ---- drawing ahead ---------------------------------------------------------------------
// A model may draw the gamma0 of its NEXT update while this one's kernels run (trlda_hip.hip,
// fresh_gamma_device): the stream is then advanced before its turn.  One such speculation can
// be outstanding; whoever touches the generator for anything else cancels it first -- the state
// goes back to where it was -- so the order of draws stays the reference's whatever comes next.
namespace {
struct Speculation {
    bool pending = false;
    uint64_t token = 0;
    GlibcRandom before;
} g_spec;
uint64_t g_spec_counter = 0;
}  // namespace

uint64_t rng_speculate_begin()
{
    rng_speculation_cancel();
    g_spec.pending = true;
    g_spec.before = g_rng;
    g_spec.token = ++g_spec_counter;
    return g_spec.token;
}

bool rng_speculation_claim(uint64_t token)
{
    if (!g_spec.pending || g_spec.token != token)
        return false;
    g_spec.pending = false;                          // the draw took its turn after all
    return true;
}

void rng_speculation_cancel_if(uint64_t token)
{
    if (g_spec.pending && g_spec.token == token)
        rng_speculation_cancel();
}

void rng_speculation_cancel()
{
    if (g_spec.pending) {
        g_rng = g_spec.before;
        g_spec.pending = false;
    }
}

void rng_current_window(uint32_t (&w)[31]) { rng_to_window(g_rng, w); }

void rng_advance(uint64_t draws)
{
    uint32_t w[31];
    rng_to_window(g_rng, w);
    jump_apply(jump_power(draws), w);
    window_to_rng(w, g_rng);
}

// M[l][d - 1] = A^(d 16^l L), l < kRngLevels, d = 1 .. 15 (31 x 31 words each), computed once
// per segment length L
const std::vector<uint32_t> &rng_level_matrices(int L, int levels)
{
    static std::mutex mu;
    static std::map<std::pair<int, int>, std::vector<uint32_t>> all;
    std::lock_guard<std::mutex> lock(mu);
    auto it = all.find(std::make_pair(L, levels));
    if (it != all.end())
        return it->second;
    std::vector<uint32_t> mats((size_t)levels * 15 * 961);
    JumpMatrix one = jump_power((uint64_t)L), cur, tmp;
    for (int l = 0; l < levels; ++l) {
        cur = one;
        for (int d = 1; d <= 15; ++d) {
            std::memcpy(mats.data() + ((size_t)l * 15 + (d - 1)) * 961, cur.a, sizeof(cur.a));
            jump_multiply(cur, one, tmp);            // A^((d + 1) 16^l L)
            cur = tmp;
        }
        one = cur;                                   // A^(16^(l + 1) L)
    }
    return all.emplace(std::make_pair(L, levels), std::move(mats)).first->second;
}

}  // namespace trlda_host

using namespace trlda_host;

extern "C" {

void trlda_seed(unsigned int seed)
{
    rng_speculation_cancel();
    srand(seed);          // keep libc's own stream in step for anything else that uses it
    g_rng.seed(seed);
}

// the generator's whole state: 31 words, then the two indices
void trlda_rng_get_state(uint32_t *state33)
{
    rng_speculation_cancel();
    std::memcpy(state33, g_rng.x, sizeof(g_rng.x));
    state33[31] = (uint32_t)g_rng.f;
    state33[32] = (uint32_t)g_rng.b;
}

void trlda_rng_set_state(const uint32_t *state33)
{
    rng_speculation_cancel();
    std::memcpy(g_rng.x, state33, sizeof(g_rng.x));
    g_rng.f = (int)(state33[31] % 31u);
    g_rng.b = (int)(state33[32] % 31u);
}

void trlda_sample_gamma(int m, int n, int k, double *out)
{
    rng_speculation_cancel();
    const int64_t total = (int64_t)m * n;
    for (int64_t i = 0; i < total; ++i)
        out[i] = 0.0;
    if (total <= 0 || k <= 0)
        return;
    // out[i] = - sum_{p < k} log |u_{p, i}|, u_{p, i} the (p * total + i)-th draw of the stream
    // (utils.cpp:224-231).  Small requests: one thread, straight through the stream.
    unsigned int hw = std::thread::hardware_concurrency();
    // (the threads are persistent, host_pool(): their number is bounded by the work per thread)
    // (more than 64 threads were measured on the 256-hardware-thread GPU box and lost: 1.0 ms
    // against 0.8 ms for K x B = 100 x 200, 100 against 56 ms for 500 x 4096)
    int64_t T = std::max<int64_t>(1, std::min<int64_t>({(int64_t)hw, (int64_t)64, total / 256}));
    if (total * k < (1 << 17))
        T = 1;
    if (const char *env = std::getenv("TRLDA_SAMPLE_THREADS"))   // tests: force a thread count
        T = std::max<int64_t>(1, std::min<int64_t>(std::atoi(env), total));
    if (T == 1) {
        for (int p = 0; p < k; ++p)
            for (int64_t i = 0; i < total; ++i) {
                const double u = -1.0 + 2.0 * (double)g_rng.next() / (double)2147483647;
                out[i] -= std::log(std::fabs(u));
            }
        return;
    }
    // Thread t owns the elements [t * len, min(total, (t + 1) * len)) in every pass: it starts
    // lo_t draws into the stream and, after the draws of a pass, jumps over the other threads'
    // share (total - its own length) to the next pass.  Per element the logs are added in pass
    // order, exactly as the serial loop does.
    const int64_t len = (total + T - 1) / T;
    T = (total + len - 1) / len;
    const int64_t last_len = total - (T - 1) * len;
    const JumpMatrix hop = jump_power((uint64_t)len);                      // thread t -> t + 1
    const JumpMatrix skip = jump_power((uint64_t)(total - len));           // pass p -> p + 1
    const JumpMatrix skip_last = jump_power((uint64_t)(total - last_len));
    // thread t starts hop^t into the stream: every thread applies the powers hop^(2^b) of the
    // set bits of t itself (a handful of 31 x 31 products) instead of the caller walking all T
    std::vector<JumpMatrix> hop_pow;
    for (int64_t span = 1; span < T; span <<= 1)
        hop_pow.push_back(span == 1 ? hop
                                    : jump_power((uint64_t)len * (uint64_t)span, &hop_pow.back()));
    uint32_t w0[31];
    rng_to_window(g_rng, w0);
    GlibcRandom final_state;
    auto work = [&](int64_t t) {
        GlibcRandom g;
        {
            uint32_t w[31];
            std::memcpy(w, w0, sizeof(w));
            for (size_t b = 0; b < hop_pow.size(); ++b)
                if ((t >> b) & 1)
                    jump_apply(hop_pow[b], w);
            window_to_rng(w, g);
        }
        const int64_t lo = t * len, hi = std::min<int64_t>(total, lo + len);
        const JumpMatrix &sk = (t == T - 1) ? skip_last : skip;
        for (int p = 0; p < k; ++p) {
            for (int64_t i = lo; i < hi; ++i) {
                const double u = -1.0 + 2.0 * (double)g.next() / (double)2147483647;
                out[i] -= std::log(std::fabs(u));
            }
            if (p + 1 < k || t == T - 1) {
                if (p + 1 == k)
                    break;                           // the last thread ends where the stream ends
                uint32_t w[31];
                rng_to_window(g, w);
                jump_apply(sk, w);
                window_to_rng(w, g);
            }
        }
        if (t == T - 1)
            final_state = g;
    };
    host_pool().run((int)T, [&](int t) { work(t); });
    g_rng = final_state;
}

void trlda_sample_gamma_init(int m, int n, double *out)
{
    trlda_sample_gamma(m, n, 100, out);
    const int64_t total = (int64_t)m * n;
    for (int64_t i = 0; i < total; ++i)
        out[i] /= 100.;
}

} // extern "C"
