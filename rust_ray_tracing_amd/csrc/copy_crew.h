// copy_crew.h -- a crew of threads that copies one buffer at a time, each member a slice (host C++ only, no HIP: the sanitizer builds of
// tests/cpp/ exercise it under TSan / ASan).  Used by the staged host -> device upload of scene_device.hip: the caller posts a
// (source, destination, length), copies slice 0 itself and returns when every helper has finished its slice.  Helpers are started
// once and spin (yield) between posts -- a post every few hundred microseconds for a few tens of milliseconds per scene; spawning
// threads per chunk cost 50 us each.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <exception>
#include <thread>
#include <vector>

namespace mipt {

class CopyCrew {
  public:
    explicit CopyCrew(int threads = 4) : want_(threads < 1 ? 1 : threads) {}
    CopyCrew(const CopyCrew &) = delete;
    CopyCrew &operator=(const CopyCrew &) = delete;
    ~CopyCrew() { stop(); }
    void start() {                                             // idempotent; with fewer helpers than asked for the shares follow the real size
        if (started_) return;
        const uint64_t g0 = gen_.load(std::memory_order_relaxed);      // a helper waits for the first post AFTER its start (a restarted crew's counter is not 0)
        try { for (int t = 1; t < want_; t++) th_.emplace_back([this, t, g0]() { worker(t, g0); }); }
        catch (const std::exception &) {}
        n_ = 1 + (int)th_.size();
        started_ = true;
    }
    int size() const { return n_; }
    void copy(const void *src, void *dst, size_t len) {        // blocks until all of [src, src + len) is at dst
        if (!started_) start();
        src_ = (const char *)src; dst_ = (char *)dst; len_ = len;
        done_.store(0, std::memory_order_relaxed);
        gen_.fetch_add(1, std::memory_order_release);           // publishes src_ / dst_ / len_
        part(0);
        while (done_.load(std::memory_order_acquire) < n_ - 1) std::this_thread::yield();
    }
    void stop() {
        if (th_.empty()) return;
        quit_.store(true, std::memory_order_relaxed);
        gen_.fetch_add(1, std::memory_order_release);
        for (auto &x : th_) x.join();
        th_.clear();
        n_ = 1; started_ = false; quit_.store(false);
    }

  private:
    void part(int t) const {
        const size_t b = len_ / (size_t)n_ * (size_t)t, e = t == n_ - 1 ? len_ : len_ / (size_t)n_ * (size_t)(t + 1);
        if (e > b) memcpy(dst_ + b, src_ + b, e - b);
    }
    void worker(int t, uint64_t seen) {
        for (;;) {
            uint64_t g;
            while ((g = gen_.load(std::memory_order_acquire)) == seen) std::this_thread::yield();
            if (quit_.load(std::memory_order_relaxed)) return;
            seen = g;
            part(t);
            done_.fetch_add(1, std::memory_order_release);
        }
    }
    const int want_;
    int n_ = 1;
    bool started_ = false;
    std::vector<std::thread> th_;
    std::atomic<uint64_t> gen_{0};
    std::atomic<int> done_{0};
    std::atomic<bool> quit_{false};
    const char *src_ = nullptr;
    char *dst_ = nullptr;
    size_t len_ = 0;
};

} // namespace mipt
