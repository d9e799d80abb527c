// work_pool.h -- a small persistent helper pool for host loops over many independent streams
// (dB finish of VU windows, queue copies of a group).  Host code only.
#ifndef CMHIP_WORK_POOL_H
#define CMHIP_WORK_POOL_H

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

// Used for the dB finish of many windows (log10 + sqrt per channel and stream: with thousands of
// streams per batch one host thread would take about as long as the GPU needs for the next block)
// and for the per-stream queue copies of a group.
struct WorkPool {
    // Work is handed out in chunks of `chunk` items from a shared counter and the calling thread
    // works too, so a helper that the OS does not schedule in time (busy hosts, CPU quotas) costs
    // nothing: whoever runs takes the chunks.
    unsigned chunk = 64;
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    unsigned generation = 0, active = 0;
    bool stop = false;
    void (*fn)(void *, unsigned, unsigned) = nullptr;
    void *arg = nullptr;
    unsigned total = 0;
    std::atomic<unsigned> next{0};

    explicit WorkPool(unsigned n)
    {
        for (unsigned i = 0; i < n; i++)
            workers.emplace_back([this] { loop(); });
    }
    ~WorkPool()
    {
        {
            std::lock_guard<std::mutex> g(m);
            stop = true;
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
    void loop()
    {
        unsigned seen = 0;
        for (;;) {
            void (*f)(void *, unsigned, unsigned);
            void *a;
            unsigned tot;
            {
                std::unique_lock<std::mutex> g(m);
                cv_work.wait(g, [&] { return stop || generation != seen; });
                if (stop)
                    return;
                seen = generation;
                f = fn;
                a = arg;
                tot = total;
                active++;
            }
            drain(f, a, tot);
            {
                std::lock_guard<std::mutex> g(m);
                if (--active == 0)
                    cv_done.notify_one();
            }
        }
    }
    void run(void (*f)(void *, unsigned, unsigned), void *a, unsigned tot, unsigned per_chunk = 64)
    {
        {
            std::lock_guard<std::mutex> g(m);
            chunk = per_chunk ? per_chunk : 1;
            fn = f;
            arg = a;
            total = tot;
            next.store(0, std::memory_order_relaxed);
            generation++;
        }
        cv_work.notify_all();
        drain(f, a, tot);                             // the caller works as well
        std::unique_lock<std::mutex> g(m);            // chunks taken by helpers may still be running
        cv_done.wait(g, [&] { return active == 0; });
    }
};

#endif
