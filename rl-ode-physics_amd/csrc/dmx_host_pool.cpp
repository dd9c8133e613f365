// dmx_host_pool.cpp -- a small persistent pool of host threads behind dmx_parallel_for (dmx_batch_priv.hpp).
// The exact (pair-bearing) tick's host bookkeeping -- joint lists, level schedules, staging -- is independent per
// island; it is spread over the host's cores several times per tick, so the threads are created once and parked.
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace dmx {

namespace {

class HostPool {
    std::vector<std::thread> workers;
    std::mutex m, run_m;
    std::condition_variable cv_work, cv_done;
    const std::function<void(int)> *job = nullptr;
    unsigned long generation = 0;
    int participants = 0, pending = 0;
    bool stop = false;

    void worker(int id)
    {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(int)> *f = nullptr;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_work.wait(lk, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
                if (id < participants) f = job;
            }
            if (!f) continue;
            (*f)(id);
            std::lock_guard<std::mutex> lk(m);
            if (--pending == 0) cv_done.notify_one();
        }
    }

public:
    ~HostPool()
    {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv_work.notify_all();
        for (auto &t : workers) t.join();
    }
    // f(t) for t in [0, nt): t = 0 on the calling thread, the rest on pool threads; returns when all are done
    void run(int nt, const std::function<void(int)> &f)
    {
        if (nt <= 1) { f(0); return; }
        std::lock_guard<std::mutex> one_at_a_time(run_m);
        {
            std::lock_guard<std::mutex> lk(m);
            while ((int)workers.size() < nt - 1) {
                const int id = (int)workers.size() + 1;
                workers.emplace_back([this, id] { worker(id); });
            }
            job = &f; participants = nt; pending = nt - 1; generation++;
        }
        cv_work.notify_all();
        f(0);
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return pending == 0; });
        job = nullptr; participants = 0;
    }
};

}  // namespace

void host_pool_run(int nt, const std::function<void(int)> &f)
{
    static HostPool pool;
    pool.run(nt, f);
}

}  // namespace dmx
