// callers_bench.cpp — a native host harness for the reference's calling pattern: N concurrent callers (rayon workers / actix
// handlers, indexes/mod.rs:268-271, tests/rps-test.py:426-455), each submitting ONE `query-batch` of B queries at a time through the
// C ABI's cos_search_batch, which fuses them (cos_index_set_coalescing).  Python threads cannot play this part: 128 of them spend
// their time handing the interpreter lock around, and the launch rate measures the interpreter.  Not product code: bench.py loads
// this as a separate shared object and passes it the library's entry point and an index handle.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

extern "C" {
typedef int32_t (*search_fn)(void *ix, const float *queries, uint32_t B, uint32_t top_k, uint32_t *out_ids, float *out_scores, uint32_t *out_counts,
                             int32_t *out_status);

// queries: [n_sets][B][dim] host floats; caller j submits set j % n_sets, `reps` times.  Returns the number of failed calls;
// *out_seconds = wall time from the common start to the last caller's return; caller 0's last answer is copied out for the check.
int32_t run_callers(void *fn, void *ix, const float *queries, uint32_t n_sets, uint32_t B, uint32_t dim, uint32_t top_k, uint32_t n_callers, uint32_t reps,
                    double *out_seconds, uint32_t *out_ids0, float *out_scores0, uint32_t *out_counts0) {
    search_fn search = (search_fn)fn;
    std::atomic<uint32_t> ready{0}, failed{0};
    std::atomic<bool> go{false};
    std::vector<std::thread> th;
    th.reserve(n_callers);
    for (uint32_t j = 0; j < n_callers; j++) {
        th.emplace_back([&, j]() {
            std::vector<uint32_t> ids((size_t)B * top_k), cnt(B);
            std::vector<float> sc((size_t)B * top_k);
            const float *q = queries + (size_t)(j % n_sets) * B * dim;
            ready.fetch_add(1);
            while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
            for (uint32_t r = 0; r < reps; r++)
                if (search(ix, q, B, top_k, ids.data(), sc.data(), cnt.data(), nullptr) != 0) failed.fetch_add(1);
            if (j == 0) {
                if (out_ids0) memcpy(out_ids0, ids.data(), ids.size() * 4);
                if (out_scores0) memcpy(out_scores0, sc.data(), sc.size() * 4);
                if (out_counts0) memcpy(out_counts0, cnt.data(), cnt.size() * 4);
            }
        });
    }
    while (ready.load() < n_callers) std::this_thread::yield();
    const auto t0 = std::chrono::steady_clock::now();
    go.store(true, std::memory_order_release);
    for (auto &t : th) t.join();
    *out_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return (int32_t)failed.load();
}
}
