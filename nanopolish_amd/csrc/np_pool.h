// np_pool.h -- the host worker pool of the batched reference-side bindings (np_batch_dropin.cpp).
//
// Why not OpenMP: the reference drives its per-record work with `#pragma omp parallel for` on the CALLER's thread team
// (src/common/nanopolish_bam_processor.cpp:99).  A binding that packs batch k+1, waits for batch k on the device and builds the
// result maps of batch k-1 needs those three to run AT THE SAME TIME, on threads that are not the caller's: an OpenMP region
// blocks its caller until the team is done, and two regions started from two threads oversubscribe the machine.  This pool is a
// fixed set of std::threads serving chunked loops from any number of submitting threads; a submitter that waits for its loop
// works on it too.  C++11 (the reference's language level), header-only, no dependency on the device library: tests/test_host_pool.py
// compiles and runs it on the CPU.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace np_shim {

class Pool {
public:
    // n_threads worker threads (>= 0; with 0 every loop runs on its submitter)
    explicit Pool(int n_threads) : own_(n_threads > 0 ? n_threads : 0), stop_(false)
    {
        for (int i = 0; i < n_threads; ++i) workers_.push_back(std::thread(&Pool::worker, this, i));
    }
    ~Pool()
    {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_.notify_all();
        for (size_t i = 0; i < workers_.size(); ++i) workers_[i].join();
    }
    int threads() const { return (int)workers_.size(); }
    // index of the calling thread among THIS pool's workers (0 .. threads() - 1), -1 for any other thread -- a worker of another pool included
    int current_worker() const { const Who& w = who(); return w.pool == this ? w.idx : -1; }

    // fn(i) for every i in [0, n), in chunks of `chunk` consecutive indices; returns when all of them have run.  The calling thread
    // takes chunks too.  Loops submitted concurrently from several threads share the workers, oldest first.
    // participate = false: the submitter only waits -- every index then runs on a WORKER (what the binding's result builder wants:
    // the heap blocks a worker allocates are later freed by the same worker, post_to).  Without workers the submitter runs the loop.
    // urgent = true: the loop goes to the FRONT of the queue -- the workers finish it before they return to older loops (what the tail of a
    // pipeline wants: its loops release what the stages behind it wait for).
    void run(int n, int chunk, const std::function<void(int)>& fn, bool participate = true, bool urgent = false)
    {
        if (n <= 0) return;
        if (chunk < 1) chunk = 1;
        std::shared_ptr<Job> j = std::make_shared<Job>(n, chunk, fn);
        if (!workers_.empty() && (n > chunk || !participate)) {
            j->urgent = urgent;
            if (urgent) urgent_open_.fetch_add(1);
            { std::lock_guard<std::mutex> g(m_); if (urgent) jobs_.push_front(j); else jobs_.push_back(j); }
            cv_.notify_all();
        }
        if (participate || workers_.empty()) work_on(*j);
        std::unique_lock<std::mutex> g(j->m);
        while (j->done < j->n_chunks) j->cv.wait(g);
    }

    // the same loop without waiting: `after` (may be empty) runs on the thread that finishes the last chunk.  fn and after are copied.
    void post(int n, int chunk, const std::function<void(int)>& fn, const std::function<void()>& after)
    {
        if (n <= 0) { if (after) after(); return; }
        if (chunk < 1) chunk = 1;
        std::shared_ptr<Job> j = std::make_shared<Job>(n, chunk, fn);      // (kept alive by the threads that work on it)
        j->after = after; j->posted = true;
        { std::lock_guard<std::mutex> g(m_); posted_ += 1; if (!workers_.empty()) jobs_.push_back(j); }
        if (workers_.empty()) { work_on(*j); return; }
        cv_.notify_all();
    }

    // fn runs once on worker `w`, ahead of any shared loop that worker would take next (no completion callback; drain() waits for it)
    void post_to(int w, const std::function<void()>& fn)
    {
        if (w < 0 || w >= (int)workers_.size()) { fn(); return; }
        { std::lock_guard<std::mutex> g(m_); posted_ += 1; own_[w].push_back(fn); }
        cv_.notify_all();
    }

    // blocks until every posted loop has finished (users call it before the data their loops touch goes away); helps meanwhile
    void drain()
    {
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> g(m_);
                if (posted_ == 0) return;
                for (size_t i = 0; i < jobs_.size(); ++i) if (jobs_[i]->next.load() < jobs_[i]->n_chunks) { j = jobs_[i]; break; }
                if (!j) { idle_.wait(g); continue; }          // every chunk is handed out: wait for the threads that run them
            }
            work_on(*j);
        }
    }

private:
    struct Job {
        const int n, chunk, n_chunks;
        std::function<void(int)> fn;
        std::function<void()> after;
        std::atomic<int> next;
        bool posted;
        bool urgent = false;                // queued ahead of the others; workers leave a non-urgent loop for it between chunks
        std::atomic<bool> closed{false};    // (urgent) its last chunk has been handed out
        int done;                       // chunks finished (under m)
        std::mutex m; std::condition_variable cv;
        Job(int n_, int chunk_, const std::function<void(int)>& f) : n(n_), chunk(chunk_), n_chunks((n_ + chunk_ - 1) / chunk_), fn(f), next(0), posted(false), done(0) {}
    };

    // takes chunks of j until none is left; returns the number of chunks this thread ran
    void work_on(Job& j)
    {
        int mine = 0;
        const bool worker = who().pool == this;
        for (;;) {
            if (worker && !j.urgent && urgent_open_.load(std::memory_order_relaxed) > 0) break;      // an urgent loop waits: back to the queue's front
            const int c = j.next.fetch_add(1);
            if (c >= j.n_chunks) {
                if (j.urgent && !j.closed.exchange(true)) urgent_open_.fetch_sub(1);
                break;
            }
            const int lo = c * j.chunk, hi = lo + j.chunk < j.n ? lo + j.chunk : j.n;
            for (int i = lo; i < hi; ++i) j.fn(i);
            ++mine;
        }
        if (mine) {
            bool last = false;
            { std::lock_guard<std::mutex> g(j.m); j.done += mine; last = j.done == j.n_chunks; }
            if (last) {
                if (j.after) j.after();
                j.cv.notify_all();
                if (j.posted) { std::lock_guard<std::mutex> g(m_); posted_ -= 1; idle_.notify_all(); }
            }
        }
    }

    struct Who { const Pool* pool; int idx; };
    static Who& who() { static thread_local Who w = {nullptr, -1}; return w; }

    void worker(int idx)
    {
        who().pool = this; who().idx = idx;
        for (;;) {
            std::shared_ptr<Job> j;
            std::function<void()> mine;
            {
                std::unique_lock<std::mutex> g(m_);
                for (;;) {
                    if (!own_[idx].empty()) { mine = own_[idx].front(); own_[idx].pop_front(); break; }
                    while (!jobs_.empty() && jobs_.front()->next.load() >= jobs_.front()->n_chunks) jobs_.pop_front();   // nothing left to hand out
                    if (!jobs_.empty()) { j = jobs_.front(); break; }
                    if (stop_) return;
                    cv_.wait(g);
                }
            }
            if (mine) {
                mine();
                std::lock_guard<std::mutex> g(m_);
                posted_ -= 1; idle_.notify_all();
            } else
                work_on(*j);
        }
    }

    std::vector<std::thread> workers_;
    std::vector<std::deque<std::function<void()> > > own_;      // per-worker tasks (post_to), under m_
    std::deque<std::shared_ptr<Job> > jobs_;
    std::mutex m_;
    std::condition_variable cv_, idle_;
    int posted_ = 0;                    // posted loops not finished yet
    std::atomic<int> urgent_open_{0};   // urgent loops that still have chunks to hand out
    bool stop_;
};

} // namespace np_shim
