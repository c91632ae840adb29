// host_common.cpp -- error message and host thread pool of libtrlda_hip.so (host only; see
// host_common.h).
#include "host_common.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <unistd.h>

#include "../../include/trlda_hip.h"

namespace trlda_host {

namespace {
thread_local std::string g_error;
}

int fail(int code, const std::string &msg)
{
    g_error = msg;
    return code;
}

// A few persistent host threads for the gamma draw and the text parser: starting 20 threads
// costs ~0.3 ms, as much as a draw itself.  One caller at a time (the Python surface holds the
// GIL across the call, as the reference does; other callers queue on run_mu_).
struct HostPool::Impl {
    ~Impl()
    {
        {
            std::lock_guard<std::mutex> lock(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_)
            t.join();
    }
    void run(int n, const std::function<void(int)> &job)
    {
        std::lock_guard<std::mutex> serial(run_mu_);
        if (n <= 1) {
            job(0);
            return;
        }
        {
            std::lock_guard<std::mutex> lock(mu_);
            while ((int)workers_.size() < n - 1) {
                const int id = (int)workers_.size() + 1;
                workers_.emplace_back([this, id] { loop(id); });
            }
            job_ = &job;
            active_ = n;
            pending_ = n - 1;
            ++generation_;
        }
        cv_.notify_all();
        job(0);
        std::unique_lock<std::mutex> lock(mu_);
        done_.wait(lock, [this] { return pending_ == 0; });
        job_ = nullptr;
    }
    void loop(int id)
    {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)> *job = nullptr;
            {
                std::unique_lock<std::mutex> lock(mu_);
                cv_.wait(lock, [&] { return stop_ || generation_ != seen; });
                if (stop_)
                    return;
                seen = generation_;
                if (id < active_)
                    job = job_;
            }
            if (job) {
                (*job)(id);
                std::lock_guard<std::mutex> lock(mu_);
                if (--pending_ == 0)
                    done_.notify_one();
            }
        }
    }
    std::mutex mu_, run_mu_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> workers_;
    const std::function<void(int)> *job_ = nullptr;
    uint64_t generation_ = 0;
    int active_ = 0, pending_ = 0;
    bool stop_ = false;
};

HostPool::HostPool() : impl_(new Impl()) {}
HostPool::~HostPool() { delete impl_; }
void HostPool::run(int n, const std::function<void(int)> &job) { impl_->run(n, job); }

HostPool &host_pool()
{
    // On the heap and never destroyed: worker threads blocked on a condition variable must not
    // be joined from a static destructor at exit, and a child of fork() has no workers at all
    // -- it gets a pool of its own (the parent's object is left alone).
    static HostPool *pool = nullptr;
    static pid_t owner = 0;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!pool || owner != getpid()) {
        pool = new HostPool();
        owner = getpid();
    }
    return *pool;
}

// ---- WorkQueue ---------------------------------------------------------------------------------
struct WorkQueue::Impl {
    explicit Impl(int n) : n_(n) {}
    ~Impl()
    {
        {
            std::lock_guard<std::mutex> lock(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_)
            t.join();
    }
    void submit(std::function<void()> job)
    {
        if (n_ <= 0) {
            job();
            return;
        }
        bool wake = false;
        {
            std::lock_guard<std::mutex> lock(mu_);
            if ((int)workers_.size() < n_ && (int)workers_.size() < (int)jobs_.size() + running_ + 1)
                workers_.emplace_back([this] { loop(); });     // (one more thread while there is a backlog)
            jobs_.push_back(std::move(job));
            pending_.store((int)jobs_.size(), std::memory_order_release);
            wake = sleepers_ > 0;
        }
        if (wake)
            cv_.notify_one();
    }
    void wait_idle()
    {
        std::unique_lock<std::mutex> lock(mu_);
        idle_.wait(lock, [this] { return jobs_.empty() && running_ == 0; });
    }
    void loop()
    {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lock(mu_);
                if (jobs_.empty() && !stop_) {
                    // (a stream of jobs a few microseconds apart: look again for a while before going
                    // to sleep -- waking a sleeper is a system call on the submitter's thread)
                    lock.unlock();
                    static const int spin_us = [] {
                        const char *e = std::getenv("TRLDA_QUEUE_SPIN_US");
                        return e ? std::atoi(e) : 150;
                    }();
                    const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us);
                    while (pending_.load(std::memory_order_acquire) == 0 && std::chrono::steady_clock::now() < until)
                        cpu_relax();
                    lock.lock();
                }
                ++sleepers_;
                cv_.wait(lock, [this] { return stop_ || !jobs_.empty(); });
                --sleepers_;
                if (stop_ && jobs_.empty())
                    return;
                job = std::move(jobs_.front());
                jobs_.erase(jobs_.begin());
                pending_.store((int)jobs_.size(), std::memory_order_release);
                ++running_;
            }
            job();
            {
                std::lock_guard<std::mutex> lock(mu_);
                --running_;
                if (jobs_.empty() && running_ == 0)
                    idle_.notify_all();
            }
        }
    }
    const int n_;
    std::mutex mu_;
    std::condition_variable cv_, idle_;
    std::vector<std::thread> workers_;
    std::vector<std::function<void()>> jobs_;
    std::atomic<int> pending_{0};                    // jobs_.size(), for the workers that look without the lock
    int running_ = 0, sleepers_ = 0;
    bool stop_ = false;
    static void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
};

WorkQueue::WorkQueue(int threads) : impl_(new Impl(threads)) {}
WorkQueue::~WorkQueue() { delete impl_; }
int WorkQueue::threads() const { return impl_->n_; }
void WorkQueue::submit(std::function<void()> job) { impl_->submit(std::move(job)); }
void WorkQueue::wait_idle() { impl_->wait_idle(); }

}  // namespace trlda_host

extern "C" const char *trlda_last_error(void) { return trlda_host::g_error.c_str(); }
