// host_stage.h — parallel staging copies for the host-buffer API (engine.hip search_host_pipelined).
// A caller's queries live in pageable memory; `hipMemcpyAsync` from pageable memory is staged by the runtime on the calling thread,
// one thread's memcpy bandwidth for 100 MB per 32 768 x 768 call — the copy, not the search, then bounds a single synchronous caller
// (bench.py host_api_pcie_inclusive.single_caller: 9.5 ms per call against 7.1 ms per resident launch).  With a StagePool the call
// copies each chunk into the pipe's pinned buffer with several threads and hands the DMA engine pinned memory, like the slots of the
// coalescing path do with their callers' own threads (engine.hip CoSlot).  Host code only; no HIP types here so that
// tests/cxx/host_stage_test.cpp can exercise it without a device.
#pragma once
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace cosdev {

class StagePool {
  public:
    explicit StagePool(unsigned workers) {
        for (unsigned i = 0; i < workers; i++) th_.emplace_back([this, i] { run(i); });
    }
    ~StagePool() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_work_.notify_all();
        for (auto &t : th_) t.join();
    }
    StagePool(const StagePool &) = delete;
    StagePool &operator=(const StagePool &) = delete;
    unsigned workers() const { return (unsigned)th_.size(); }

    // dst[0, bytes) = src[0, bytes): the range is cut into workers() + 1 slices (multiples of 4 KiB), the caller copies the last one.
    // One copy at a time per pool (callers of different handles take turns).
    void copy(void *dst, const void *src, size_t bytes) {
        const size_t parts = th_.size() + 1;
        if (th_.empty() || bytes < (size_t)(64u << 10) * parts) { // not worth a wake-up
            memcpy(dst, src, bytes);
            return;
        }
        std::unique_lock<std::mutex> turn(turn_mu_);
        const size_t slice = ((bytes + parts - 1) / parts + 4095) & ~(size_t)4095;
        {
            std::lock_guard<std::mutex> g(mu_);
            dst_ = (char *)dst;
            src_ = (const char *)src;
            bytes_ = bytes;
            slice_ = slice;
            pending_ = (unsigned)th_.size();
            epoch_++;
        }
        cv_work_.notify_all();
        const size_t off = slice * th_.size();
        if (off < bytes) memcpy((char *)dst + off, (const char *)src + off, bytes - off);
        std::unique_lock<std::mutex> g(mu_);
        cv_done_.wait(g, [this] { return pending_ == 0; });
    }

  private:
    void run(unsigned i) {
        uint64_t seen = 0;
        for (;;) {
            char *d;
            const char *s;
            size_t n = 0, off;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_work_.wait(g, [&] { return stop_ || epoch_ != seen; });
                if (stop_) return;
                seen = epoch_;
                off = slice_ * i;
                d = dst_ + off;
                s = src_ + off;
                if (off < bytes_) n = bytes_ - off < slice_ ? bytes_ - off : slice_;
            }
            if (n) memcpy(d, s, n);
            {
                std::lock_guard<std::mutex> g(mu_);
                if (--pending_ == 0) cv_done_.notify_all();
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_, turn_mu_;
    std::condition_variable cv_work_, cv_done_;
    char *dst_ = nullptr;
    const char *src_ = nullptr;
    size_t bytes_ = 0, slice_ = 0;
    unsigned pending_ = 0;
    uint64_t epoch_ = 0;
    bool stop_ = false;
};

} // namespace cosdev
