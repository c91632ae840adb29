// Driver for the host-only translation units of libtrlda_hip.so (csrc/host_common.cpp,
// host_rng.cpp, text_docs.cpp, eb_steps.cpp) built with a plain C++ compiler under the
// sanitizers: trlda_amd/build.py --sanitize address|thread, run by tests/test_host_sanitize.py.
// Test infrastructure; GPU sanitizers are not available on this pool, the host side is what
// can be checked -- threads, mmap parsing, the jump-ahead cache, K-sized numerics.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../include/trlda_hip.h"
#include "../../trlda_amd/csrc/host_common.h"

static int failures = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++failures;                                                          \
        }                                                                        \
    } while (0)

static std::string corpus(unsigned seed, int docs, int max_words)
{
    std::string text;
    unsigned s = seed;
    auto next = [&]() { return s = s * 1664525u + 1013904223u; };
    for (int d = 0; d < docs; ++d) {
        const int n = (int)(next() >> 8) % (max_words + 1);
        text += std::to_string(n);
        for (int j = 0; j < n; ++j)
            text += " " + std::to_string((next() >> 8) % 1000000) + ":" + std::to_string((next() >> 8) % 90);
        text += (d % 7 == 3) ? "\r\n" : "\n";
    }
    return text;
}

static void parser_checks(bool threaded_callers)
{
    const std::string text = corpus(1, 4000, 60);
    trlda_docs *ref = nullptr;
    setenv("TRLDA_PARSE_THREADS", "1", 1);
    CHECK(trlda_docs_from_buffer(text.data(), text.size(), &ref) == TRLDA_OK);
    const int64_t n = trlda_docs_num_docs(ref), nnz = trlda_docs_nnz(ref);
    CHECK(n == 4000 && nnz > 0);
    for (const char *threads : {"2", "7", "64"}) {
        setenv("TRLDA_PARSE_THREADS", threads, 1);
        trlda_docs *d = nullptr;
        CHECK(trlda_docs_from_buffer(text.data(), text.size(), &d) == TRLDA_OK);
        CHECK(trlda_docs_num_docs(d) == n && trlda_docs_nnz(d) == nnz);
        CHECK(!std::memcmp(trlda_docs_offsets(d), trlda_docs_offsets(ref), (size_t)(n + 1) * 8));
        CHECK(!std::memcmp(trlda_docs_ids(d), trlda_docs_ids(ref), (size_t)nnz * 4));
        CHECK(!std::memcmp(trlda_docs_cnts(d), trlda_docs_cnts(ref), (size_t)nnz * 4));
        trlda_docs_destroy(d);
    }
    // the file form, a last line without its newline, nothing at all, refusals
    char path[] = "/tmp/trlda_sanitize_XXXXXX";
    const int fd = mkstemp(path);
    CHECK(fd >= 0);
    const std::string cut = text.substr(0, text.size() - 1);
    CHECK(write(fd, cut.data(), cut.size()) == (ssize_t)cut.size());
    close(fd);
    trlda_docs *d = nullptr;
    CHECK(trlda_docs_from_text(path, &d) == TRLDA_OK && trlda_docs_num_docs(d) == n);
    trlda_docs_destroy(d);
    unlink(path);
    CHECK(trlda_docs_from_text(path, &d) != TRLDA_OK && d == nullptr);
    CHECK(trlda_docs_from_buffer("", 0, &d) == TRLDA_OK && trlda_docs_num_docs(d) == 0);
    trlda_docs_destroy(d);
    const char *bad = "2 1:1 2:2\n2 3:1 oops\n";
    CHECK(trlda_docs_from_buffer(bad, std::strlen(bad), &d) == TRLDA_ERR_VALUE);
    CHECK(std::strstr(trlda_last_error(), "line 2") != nullptr);
    const char *cr = "1 1:1\r2 2:2\n";
    CHECK(trlda_docs_from_buffer(cr, std::strlen(cr), &d) == TRLDA_ERR_ARG);
    const char *big = "1 99999999999:1\n";
    CHECK(trlda_docs_from_buffer(big, std::strlen(big), &d) == TRLDA_ERR_VALUE);
    if (threaded_callers) {
        // several callers at once: the pool serialises them, results stay the same
        setenv("TRLDA_PARSE_THREADS", "4", 1);
        std::vector<std::thread> callers;
        for (int c = 0; c < 4; ++c)
            callers.emplace_back([&, c] {
                const std::string mine = corpus(10 + (unsigned)c, 1500, 30);
                trlda_docs *x = nullptr;
                if (trlda_docs_from_buffer(mine.data(), mine.size(), &x) != TRLDA_OK ||
                    trlda_docs_num_docs(x) != 1500)
                    __atomic_add_fetch(&failures, 1, __ATOMIC_RELAXED);
                trlda_docs_destroy(x);
            });
        for (auto &t : callers)
            t.join();
    }
    trlda_docs_destroy(ref);
}

static void rng_checks()
{
    // >= 100 distinct shapes: the jump-ahead cache fills, and (with many threads and lengths) evicts;
    // the threaded draw equals the serial one bit for bit, and leaves the stream at the same place
    for (int i = 0; i < 120; ++i) {
        const int m = 5 + 3 * i, n = 40 + (i * 37) % 211, k = 2 + i % 3;
        std::vector<double> serial((size_t)m * n), threaded((size_t)m * n);
        uint32_t s0[33], s1[33];
        setenv("TRLDA_SAMPLE_THREADS", "1", 1);
        trlda_seed(1000u + (unsigned)i);
        trlda_sample_gamma(m, n, k, serial.data());
        trlda_rng_get_state(s0);
        setenv("TRLDA_SAMPLE_THREADS", i % 2 ? "5" : "16", 1);
        trlda_seed(1000u + (unsigned)i);
        trlda_sample_gamma(m, n, k, threaded.data());
        trlda_rng_get_state(s1);
        CHECK(!std::memcmp(serial.data(), threaded.data(), serial.size() * 8));
        // (the state is a circular buffer: compare what comes next, not the raw words)
        double a[4], b[4];
        setenv("TRLDA_SAMPLE_THREADS", "1", 1);
        trlda_rng_set_state(s0);
        trlda_sample_gamma(2, 2, 1, a);
        trlda_rng_set_state(s1);
        trlda_sample_gamma(2, 2, 1, b);
        CHECK(!std::memcmp(a, b, sizeof(a)));
    }
    unsetenv("TRLDA_SAMPLE_THREADS");
    std::vector<double> big((size_t)300 * 400);
    trlda_seed(7);
    trlda_sample_gamma_init(300, 400, big.data());           // the default thread count
    for (double v : big)
        CHECK(std::isfinite(v) && v > 0.0);
}

// a draw made ahead of its turn: cancelled, the stream is where it was; claimed, it has moved on
static void speculation_checks()
{
    double a[4], b[4], c[8];
    setenv("TRLDA_SAMPLE_THREADS", "1", 1);
    trlda_seed(5);
    trlda_sample_gamma(2, 2, 1, a);
    trlda_seed(5);
    const uint64_t t1 = trlda_host::rng_speculate_begin();
    trlda_host::rng_advance(1000);
    trlda_sample_gamma(2, 2, 1, b);                  // any other use of the generator cancels it
    CHECK(!std::memcmp(a, b, sizeof(a)));
    CHECK(!trlda_host::rng_speculation_claim(t1));
    trlda_seed(5);
    trlda_sample_gamma(4, 2, 1, c);                  // eight draws: the last four are ...
    trlda_seed(5);
    const uint64_t t2 = trlda_host::rng_speculate_begin();
    trlda_host::rng_advance(4);
    CHECK(trlda_host::rng_speculation_claim(t2));    // ... what follows a claimed draw of four
    trlda_sample_gamma(2, 2, 1, b);
    CHECK(!std::memcmp(b, c + 4, sizeof(b)));
    const uint64_t t3 = trlda_host::rng_speculate_begin();
    trlda_host::rng_advance(77);
    trlda_host::rng_speculation_cancel_if(t3 + 1);   // somebody else's token: nothing happens
    CHECK(trlda_host::rng_speculation_claim(t3));
    unsetenv("TRLDA_SAMPLE_THREADS");
}

static void eb_checks()
{
    const int K = 37;
    std::vector<double> alpha((size_t)K), pgd((size_t)K), out((size_t)K), rows((size_t)K);
    for (int k = 0; k < K; ++k) {
        alpha[(size_t)k] = 0.05 + 0.01 * k;
        pgd[(size_t)k] = -40.0 - 3.0 * std::sin(k);
        rows[(size_t)k] = 200.0 + 10.0 * k;
    }
    CHECK(trlda_eb_online_alpha_step(K, alpha.data(), pgd.data(), 25., .1, 1e-6, out.data()) == TRLDA_OK);
    for (double v : out)
        CHECK(std::isfinite(v) && v >= 1e-6);
    CHECK(trlda_eb_alpha_line_search(K, alpha.data(), pgd.data(), 25., 10, 1e-6, 1e-8, out.data()) == TRLDA_OK);
    for (double v : out)
        CHECK(std::isfinite(v) && v >= 1e-6);
    const double e1 = trlda_eb_online_eta_step(.3, -90000., rows.data(), K, 900, .1, 1e-6);
    const double e2 = trlda_eb_eta_line_search(.3, -90000., rows.data(), K, 900, 20, 1e-6, 1e-8);
    CHECK(std::isfinite(e1) && e1 >= 1e-6 && std::isfinite(e2) && e2 >= 1e-6);
    double x[3] = {.01, 1.5, 11.}, p[3], p1[3];
    trlda_debug_host_psi(3, x, p, p1);
    CHECK(std::fabs(p1[2] - 0.09516633568168575) < 1e-12);
    CHECK(trlda_eb_online_alpha_step(0, alpha.data(), pgd.data(), 25., .1, 1e-6, out.data()) != TRLDA_OK);
}

// the mini-batch index (csrc/batch_index.cpp) into exactly-sized heap buffers -- a write past a
// section's end into the next allocation is what the sanitizer is here for -- from several threads
// at once through the work queue that trlda_batch_create builds it on
extern "C" int trlda_debug_batch_index(int V, int B, const int32_t *indptr, const int32_t *ids,
                                       const int32_t *cnts, int cus, int64_t *info, void *buffer, size_t cap);

static void index_checks()
{
    struct Shape { int V, B, len, dup; };
    const Shape shapes[] = {{7000, 200, 100, 0}, {30, 6, 9, 1}, {1, 4, 1, 1}, {50, 0, 0, 0}, {9000, 12, 700, 0},
                            {300, 3000, 60, 1}, {64, 20, 300, 1}, {256, 64, 64, 0}};
    trlda_host::WorkQueue queue(4);
    std::atomic<int> built{0};
    for (int rep = 0; rep < 3; ++rep)
        for (const Shape &sh : shapes)
            queue.submit([sh, rep, &built] {
                std::vector<int32_t> indptr(1, 0), ids, cnts;
                uint32_t state = 12345u + (uint32_t)sh.V * 7u + (uint32_t)rep;
                auto next = [&state] { state = state * 1664525u + 1013904223u; return state >> 8; };
                for (int d = 0; d < sh.B; ++d) {
                    const int n = sh.len ? (int)(next() % (uint32_t)(2 * sh.len)) : 0;
                    for (int j = 0; j < n; ++j) {
                        ids.push_back(sh.dup ? (int32_t)(next() % (uint32_t)sh.V)
                                             : (int32_t)((next() % (uint32_t)sh.V + (uint32_t)j) % (uint32_t)sh.V));
                        cnts.push_back((int32_t)(next() % 5u));
                    }
                    indptr.push_back((int32_t)ids.size());
                }
                int64_t info[64];
                CHECK(trlda_debug_batch_index(sh.V, sh.B, indptr.data(), ids.data(), cnts.data(), 256, info,
                                              nullptr, 0) == TRLDA_OK);
                std::vector<char> buf((size_t)info[24]);
                CHECK(trlda_debug_batch_index(sh.V, sh.B, indptr.data(), ids.data(), cnts.data(), 256, info,
                                              buf.data(), buf.size()) == TRLDA_OK);
                // the word-major ranks are a permutation of the entries
                const int32_t *wrank = reinterpret_cast<const int32_t *>(buf.data() + info[32 + 4]);
                std::vector<char> seen(ids.size(), 0);
                for (size_t p = 0; p < ids.size(); ++p) {
                    CHECK(wrank[p] >= 0 && (size_t)wrank[p] < ids.size() && !seen[(size_t)wrank[p]]);
                    seen[(size_t)wrank[p]] = 1;
                }
                ++built;
            });
    queue.wait_idle();
    CHECK(built.load() == 3 * (int)(sizeof(shapes) / sizeof(shapes[0])));
    int64_t info[64];
    const int32_t ip[2] = {0, 1}, bad_id[1] = {7}, one[1] = {1};
    CHECK(trlda_debug_batch_index(5, 1, ip, bad_id, one, 256, info, nullptr, 0) == TRLDA_ERR_WORD_ID);
    trlda_host::WorkQueue inline_queue(0);
    int ran = 0;
    inline_queue.submit([&ran] { ++ran; });
    CHECK(ran == 1);
}

int main(int argc, char **argv)
{
    const bool thread_mode = argc > 1 && !std::strcmp(argv[1], "threads");
    parser_checks(thread_mode);
    rng_checks();
    index_checks();
    if (!thread_mode) {
        speculation_checks();
        eb_checks();
    }
    if (failures) {
        std::fprintf(stderr, "%d check(s) failed\n", failures);
        return 1;
    }
    std::printf("HOST-SANITIZE-OK\n");
    return 0;
}
