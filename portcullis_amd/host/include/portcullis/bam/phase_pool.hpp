// PhasePool: a fixed set of worker threads for work that comes in phases (inflate / walk / fill / deflate): a phase is `n`
// independent tasks, run(n, fn) returns when all of them are done.  Used by BamReader's parallel decoders, BamWriter and
// BamFilter.
#pragma once

#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace portcullis {
namespace bam {

class PhasePool {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cvWork, cvDone;
    std::function<void(size_t)> fn;
    size_t nTasks = 0, nextTask = 0, pending = 0;
    uint64_t generation = 0;
    bool stop = false;

    void loop() {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cvWork.wait(lk, [&] { return stop || (generation != seen && nextTask < nTasks); });
            if (stop) return;
            seen = generation;
            while (nextTask < nTasks) {
                const size_t t = nextTask++;
                lk.unlock();
                fn(t);
                lk.lock();
                if (--pending == 0) cvDone.notify_all();
            }
        }
    }

public:
    explicit PhasePool(int n) {
        for (int i = 0; i < n; i++) threads.emplace_back([this] { loop(); });
    }
    ~PhasePool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cvWork.notify_all();
        for (auto& t : threads) t.join();
    }
    size_t size() const { return threads.size(); }
    void run(size_t n, std::function<void(size_t)> f) {
        if (n == 0) return;
        if (threads.empty() || n == 1) {
            for (size_t t = 0; t < n; t++) f(t);
            return;
        }
        std::unique_lock<std::mutex> lk(mu);
        fn = std::move(f);
        nTasks = n;
        nextTask = 0;
        pending = n;
        generation++;
        cvWork.notify_all();
        cvDone.wait(lk, [&] { return pending == 0; });
    }
};

}  // namespace bam
}  // namespace portcullis
