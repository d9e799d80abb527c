// work_pool.h -- a small persistent helper pool for host loops over many independent streams
// (dB finish of VU windows, queue copies of a group).  Host code only.
#ifndef CMHIP_WORK_POOL_H
#define CMHIP_WORK_POOL_H

#include <atomic>
#include <chrono>
#include <stdlib.h>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

// Used for the dB finish of many windows (log10 + sqrt per channel and stream: with thousands of
// streams per batch one host thread would take about as long as the GPU needs for the next block)
// and for the per-stream queue copies of a group.
// one step of a spin-wait, whatever the host's architecture
static inline void cmhip_cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__) || defined(__arm__)
    __asm__ __volatile__("yield" ::: "memory");
#else
    __asm__ __volatile__("" ::: "memory");
#endif
}

struct WorkPool {
    // Work is handed out in chunks of `chunk` items from a shared counter and the calling thread
    // works too, so a helper that the OS does not schedule in time (busy hosts, CPU quotas) costs
    // nothing: whoever runs takes the chunks.
    // Helpers that have just worked wait for the next job in a short spin before they go to sleep
    // (default 400 us, $CMHIP_POOL_SPIN_US): jobs that come every few hundred microseconds -- a
    // bench loop, a busy streaming host -- then never pay a futex wake-up, which on a loaded host
    // took longer than the job (a step of 0.35 ms grew to 0.40-0.46 ms).
    unsigned chunk = 64;
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::atomic<unsigned> generation{0}, active{0};
    unsigned sleepers = 0;                            // helpers inside cv_work.wait (under m)
    unsigned spin_us = 400;
    bool stop = false;
    void (*fn)(void *, unsigned, unsigned) = nullptr;
    void *arg = nullptr;
    unsigned total = 0;
    std::atomic<unsigned> next{0};

    explicit WorkPool(unsigned n)
    {
        if (const char *e = getenv("CMHIP_POOL_SPIN_US"))
            spin_us = (unsigned)atoi(e);
        for (unsigned i = 0; i < n; i++)
            workers.emplace_back([this] { loop(); });
    }
    ~WorkPool()
    {
        {
            std::lock_guard<std::mutex> g(m);
            stop = true;
            generation.fetch_add(1, std::memory_order_release);   // spinning helpers look at this
        }
        cv_work.notify_all();
        for (auto &t : workers)
            t.join();
    }
    void drain(void (*f)(void *, unsigned, unsigned), void *a, unsigned tot)
    {
        for (;;) {
            const unsigned lo = next.fetch_add(chunk, std::memory_order_relaxed);
            if (lo >= tot)
                return;
            f(a, lo, lo + chunk < tot ? lo + chunk : tot);
        }
    }
    // spins until pred() or `us` microseconds have passed; true if pred() held
    template <typename P>
    static bool spin_for(unsigned us, P pred)
    {
        if (pred())
            return true;
        if (!us)
            return false;
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            for (int i = 0; i < 64; i++) {
                if (pred())
                    return true;
                cmhip_cpu_relax();
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(us))
                return pred();
        }
    }
    void loop()
    {
        unsigned seen = 0;
        bool worked = false;                          // only a helper that has just worked spins
        for (;;) {
            void (*f)(void *, unsigned, unsigned);
            void *a;
            unsigned tot;
            if (!(worked && spin_for(spin_us, [&] { return generation.load(std::memory_order_acquire) != seen; }))) {
                std::unique_lock<std::mutex> g(m);
                sleepers++;
                cv_work.wait(g, [&] { return stop || generation.load(std::memory_order_relaxed) != seen; });
                sleepers--;
            }
            {
                std::lock_guard<std::mutex> g(m);     // the job's fields and `next` belong together
                if (stop)
                    return;
                seen = generation.load(std::memory_order_relaxed);
                // A helper that comes late finds every chunk taken: it must not join (the caller may
                // already have returned, and the next job will reset `next` under this lock -- a helper
                // still holding this job's bounds would then take chunks of the next one and drop them).
                // While chunks are left the caller is still inside run(), and it will see `active`.
                if (next.load(std::memory_order_relaxed) >= total)
                    continue;
                f = fn;
                a = arg;
                tot = total;
                active.fetch_add(1, std::memory_order_relaxed);
            }
            drain(f, a, tot);
            worked = true;
            if (active.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> g(m);
                cv_done.notify_one();
            }
        }
    }
    // start() hands a job to the helpers and returns; finish() takes whatever chunks are left itself and
    // waits for the helpers' last ones.  Between the two the caller does something else (the next launch);
    // run() is the two in one.  One job at a time: no start() before the finish() of the job before.
    void start(void (*f)(void *, unsigned, unsigned), void *a, unsigned tot, unsigned per_chunk = 64)
    {
        bool wake;
        {
            std::lock_guard<std::mutex> g(m);
            chunk = per_chunk ? per_chunk : 1;
            fn = f;
            arg = a;
            total = tot;
            next.store(0, std::memory_order_relaxed);
            generation.fetch_add(1, std::memory_order_release);
            wake = sleepers != 0;
        }
        if (wake)
            cv_work.notify_all();
    }
    void run(void (*f)(void *, unsigned, unsigned), void *a, unsigned tot, unsigned per_chunk = 64)
    {
        start(f, a, tot, per_chunk);
        finish();
    }
    void finish()
    {
        drain(fn, arg, total);                        // the caller works as well
        // chunks taken by helpers may still be running: they are short, so look before sleeping.
        // The last look is under the lock: a helper may be between finding chunks left and counting
        // itself in (it holds the lock there), and must be waited for like the others.
        spin_for(200, [&] { return active.load(std::memory_order_acquire) == 0; });
        std::unique_lock<std::mutex> g(m);
        cv_done.wait(g, [&] { return active.load(std::memory_order_acquire) == 0; });
    }
};

#endif
