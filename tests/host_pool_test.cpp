// CPU test of nanopolish_amd/csrc/np_pool.h (tests/test_host_pool.py compiles and runs it): loops from several submitting threads at
// once, posted loops with a completion callback, drain, a pool without workers.
#include <algorithm>
#include <cstdio>
#include <numeric>
#include <thread>
#include <vector>
#include "np_pool.h"

using np_shim::Pool;

static int check(bool ok, const char* what) { if (!ok) { fprintf(stderr, "FAILED: %s\n", what); return 1; } return 0; }

int main()
{
    int bad = 0;
    for (int nt = 0; nt <= 7; nt += 7) {
        Pool pool(nt);
        // (1) one loop, every index exactly once
        std::vector<int> hit(100003, 0);
        pool.run((int)hit.size(), 97, [&](int i) { hit[i] += 1; });
        bad += check(std::accumulate(hit.begin(), hit.end(), 0LL) == (long long)hit.size() && *std::min_element(hit.begin(), hit.end()) == 1, "run covers every index once");
        // (2) four submitters at once
        std::vector<std::vector<long long> > out(4, std::vector<long long>(20000, 0));
        std::vector<std::thread> th;
        for (int t = 0; t < 4; ++t) th.push_back(std::thread([&, t]() { for (int rep = 0; rep < 5; ++rep) pool.run(20000, 64, [&, t](int i) { out[t][i] += i * (t + 1); }); }));
        for (auto& x : th) x.join();
        for (int t = 0; t < 4; ++t) for (int i = 0; i < 20000; ++i) if (out[t][i] != 5LL * i * (t + 1)) { bad += check(false, "concurrent submitters"); t = 4; break; }
        // (3) posted loops: completion callback runs once, after all indices; drain waits for them
        std::atomic<int> sum(0), after(0);
        for (int k = 0; k < 6; ++k) pool.post(5000, 50, [&](int) { sum.fetch_add(1); }, [&]() { after.fetch_add(1); });
        pool.drain();
        bad += check(sum.load() == 30000 && after.load() == 6, "posted loops finish before drain returns");
        // (5) loops the submitter does not take part in run on workers only; per-worker tasks run on the worker they name
        if (nt > 0) {
            std::vector<int> who(4000, -2);
            pool.run(4000, 16, [&](int i) { who[i] = pool.current_worker(); }, false);
            bool all_workers = true;
            for (int i = 0; i < 4000; ++i) all_workers = all_workers && who[i] >= 0 && who[i] < nt;
            bad += check(all_workers, "run(participate = false) stays on the workers");
            std::vector<int> ran(nt, -1);
            for (int w = 0; w < nt; ++w) pool.post_to(w, [&, w]() { ran[w] = pool.current_worker(); });
            pool.drain();
            bool own = true;
            for (int w = 0; w < nt; ++w) own = own && ran[w] == w;
            bad += check(own, "post_to runs on the named worker");
        }
        bad += check(pool.current_worker() == -1, "the submitting thread is no worker");
        // (6) worker indices belong to a pool: a worker of ANOTHER pool is no worker of this one (two pipelines in one process)
        if (nt > 0) {
            Pool other(2);
            std::vector<int> seen_by_pool(8, 0), seen_by_other(8, 0);
            other.run(8, 1, [&](int i) { seen_by_pool[i] = pool.current_worker(); seen_by_other[i] = other.current_worker(); }, false);
            bool ok = true;
            for (int i = 0; i < 8; ++i) ok = ok && seen_by_pool[i] == -1 && seen_by_other[i] >= 0 && seen_by_other[i] < 2;
            bad += check(ok, "a worker of another pool is not a worker of this pool");
        }
        // (7) urgent loops: submitted while a long loop runs, they go first -- the workers leave the long loop between chunks -- and both loops
        //     still cover every index exactly once
        {
            std::vector<int> slow(60000, 0), fast(6000, 0);
            std::atomic<int> slow_done_when_fast_ended(-1), slow_count(0);
            std::thread a([&]() { pool.run((int)slow.size(), 8, [&](int i) { volatile double x = 0; for (int q = 0; q < 400; ++q) x += q * 1e-3; slow[i] += 1; slow_count.fetch_add(1); }); });
            while (slow_count.load() < 200) std::this_thread::yield();
            std::thread b([&]() { for (int rep = 0; rep < 3; ++rep) pool.run((int)fast.size(), 8, [&](int i) { fast[i] += 1; }, false, true); slow_done_when_fast_ended = slow_count.load(); });
            a.join(); b.join();
            bad += check(*std::min_element(slow.begin(), slow.end()) == 1 && *std::max_element(slow.begin(), slow.end()) == 1, "the long loop still covers every index once");
            bad += check(*std::min_element(fast.begin(), fast.end()) == 3 && *std::max_element(fast.begin(), fast.end()) == 3, "urgent loops cover every index once each");
            if (nt > 0) bad += check(slow_done_when_fast_ended.load() < (int)slow.size(), "urgent loops finish before the long loop they interrupted");
        }
        // (4) empty loops
        pool.run(0, 8, [&](int) { bad += 1; });
        int called = 0;
        pool.post(0, 8, [&](int) { bad += 1; }, [&]() { called += 1; });
        bad += check(called == 1, "empty posted loop still completes");
    }
    if (!bad) printf("host pool ok\n");
    return bad ? 1 : 0;
}
