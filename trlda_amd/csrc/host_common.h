// host_common.h -- what the host-only translation units of libtrlda_hip.so share: the
// thread-local error message behind trlda_last_error(), the persistent host thread pool, and
// the process's libc-compatible generator with its jump-ahead (host_rng.cpp).  Nothing here
// needs HIP: these files also build with a plain C++ compiler under the sanitizers
// (trlda_amd/build.py --sanitize, tests/test_host_sanitize.py).
#pragma once

#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace trlda_host {

// sets the calling thread's message (trlda_last_error()) and returns `code`
int fail(int code, const std::string &msg);

// A few persistent host threads (the gamma draw, the text parser): run(n, job) executes job(0)
// on the caller and job(1..n-1) on the workers and returns when all are done.
class HostPool {
public:
    HostPool();
    ~HostPool();
    void run(int n, const std::function<void(int)> &job);

private:
    struct Impl;
    Impl *impl_;
};
HostPool &host_pool();

// A queue of jobs worked off by a few threads of its own, in the order they were submitted (each
// thread takes the next job): the mini-batch index of trlda_batch_create is built here while the
// caller goes on (trlda_hip.hip).  Threads start with the first job; `threads` <= 0: submit()
// runs the job on the caller.  wait_idle(): every submitted job has finished.
class WorkQueue {
public:
    explicit WorkQueue(int threads);
    ~WorkQueue();
    int threads() const;
    void submit(std::function<void()> job);
    void wait_idle();

private:
    struct Impl;
    Impl *impl_;
};

// ---- the generator behind trlda_seed / trlda_sample_gamma (glibc's TYPE_3 rand()) ----------
struct JumpMatrix {
    uint32_t a[31][31];
};
void jump_multiply(const JumpMatrix &x, const JumpMatrix &y, JumpMatrix &out);
// A^n, the matrix that moves a 31-word window n draws on (by value: the cache behind it evicts)
JumpMatrix jump_power(uint64_t n, const JumpMatrix *half = nullptr);
void jump_apply(const JumpMatrix &m, uint32_t (&w)[31]);
// drawing ahead (host_rng.cpp): begin saves the state (cancelling any earlier speculation) and
// returns a token; claim(token) = that draw took its turn after all; cancel = the state goes
// back.  Every other use of the generator cancels first.
uint64_t rng_speculate_begin();
bool rng_speculation_claim(uint64_t token);
void rng_speculation_cancel();
void rng_speculation_cancel_if(uint64_t token);      // only if that one is the pending one
// the process generator: its window (oldest word first), and moving it on by `draws`
void rng_current_window(uint32_t (&w)[31]);
void rng_advance(uint64_t draws);
// M[l][d - 1] = A^(d 16^l L), l < levels, d = 1 .. 15 (31 x 31 words each), once per L
const std::vector<uint32_t> &rng_level_matrices(int L, int levels);

}  // namespace trlda_host
