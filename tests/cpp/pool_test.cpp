// pool_test.cpp -- WorkPool (libcoolmic-dsp_amd/csrc/work_pool.h) without a GPU: many jobs of
// different sizes and chunkings from one caller (run(), and start() ... finish() with the caller away in between), with pauses on both sides of the helpers' spin
// window, every item counted exactly once.  Built by tests/test_work_pool.py (also under
// ThreadSanitizer where the toolchain has it).
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <thread>
#include <vector>

#include "work_pool.h"

struct Job {
    std::vector<std::atomic<unsigned>> *hits;
    unsigned salt;
};

int main(int argc, char **argv)
{
    const int jobs = argc > 1 ? atoi(argv[1]) : 3000;
    for (unsigned threads : {1u, 3u, 7u}) {
        WorkPool pool(threads);
        std::vector<std::atomic<unsigned>> hits(5000);
        unsigned long long total = 0, expect = 0;
        for (int j = 0; j < jobs; j++) {
            const unsigned n = (unsigned)(1 + (j * 7919u) % 4999u);
            const unsigned chunk = (unsigned)(1 + (j * 31u) % 97u);
            for (unsigned i = 0; i < n; i++)
                hits[i].store(0, std::memory_order_relaxed);
            Job job = {&hits, (unsigned)j};
            auto body = [](void *p, unsigned lo, unsigned hi) {
                Job *q = (Job *)p;
                for (unsigned i = lo; i < hi; i++)
                    (*q->hits)[i].fetch_add(1 + (q->salt & 1u), std::memory_order_relaxed);
            };
            if (j % 3 == 2) {                         // the two-phase form: the caller is away in between
                pool.start(body, &job, n, chunk);
                if (j % 12 == 2)
                    std::this_thread::sleep_for(std::chrono::microseconds(30));
                pool.finish();
            } else {
                pool.run(body, &job, n, chunk);
            }
            for (unsigned i = 0; i < n; i++) {
                const unsigned h = hits[i].load(std::memory_order_relaxed);
                if (h != 1 + ((unsigned)j & 1u)) {
                    printf("job %d item %u counted %u times (threads %u)\n", j, i, h, threads);
                    return 1;
                }
                total += h;
            }
            expect += (unsigned long long)n * (1 + ((unsigned)j & 1u));
            if (j % 500 == 250)                       // longer than the spin window: helpers go to sleep
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
            else if (j % 100 == 50)                   // inside it
                std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
        if (total != expect) {
            printf("threads %u: %llu != %llu\n", threads, total, expect);
            return 1;
        }
    }
    printf("pool ok\n");
    return 0;
}
