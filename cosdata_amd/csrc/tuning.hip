// tuning.hip — the knob registry behind cos_tuning_set / cos_tuning_get (tuning.h says what it is for).  Host code only.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "engine_internal.h"
#include "tuning.h"

namespace cosdev {
namespace {

const char *const NAMES[TUNE_COUNT] = {
    "walk_chain_min_b", "walk_order_min_b", "walk_side_min_b", "walk_split_levels", "walk_table_cols", "walk_table_min_b",
    "walk_table_max_bytes", "walk_table_gemm", "host_pipeline_min_b", "flat_unfused", "flat_tile_kernel", "flat_pf", "flat_fp4",
    "bm25_blocks", "sparse_layout", "walk_pb", "walk_pb_upper", "walk_lat", "walk_lat4", "walk_small_table_tk", "walk_lat_warm",
    "walk_lat_la", "walk_lat4_e", "finalize_fast", "shardset_force_rccl", "build_profile",
    "walk_merge_min", "walk_adj_mag", "walk_table_rule_c", "walk_table_after_sort", "walk_upper_lds_pad", "walk_r2", "finalize_wide_max_b", "flat_fp4_w8",
};
std::atomic<long long> g_values[TUNE_COUNT];
std::once_flag g_once;

int find_key(const char *name, size_t len) {
    for (int k = 0; k < TUNE_COUNT; k++)
        if (strlen(NAMES[k]) == len && strncmp(NAMES[k], name, len) == 0) return k;
    return -1;
}

void init_once() {
    std::call_once(g_once, [] {
        for (auto &v : g_values) v.store(TUNE_UNSET, std::memory_order_relaxed);
        // the one environment variable of the library: COS_TUNING="name=value,name=value" (unknown names and malformed items are skipped)
        const char *e = getenv("COS_TUNING");
        while (e && *e) {
            const char *end = strchr(e, ',');
            const size_t item = end ? (size_t)(end - e) : strlen(e);
            const char *eq = (const char *)memchr(e, '=', item);
            if (eq) {
                const int k = find_key(e, (size_t)(eq - e));
                char *stop = nullptr;
                const long long v = strtoll(eq + 1, &stop, 10);
                if (k >= 0 && stop != eq + 1) g_values[k].store(v, std::memory_order_relaxed);
            }
            e = end ? end + 1 : nullptr;
        }
    });
}

} // namespace

long long tune(TuneKey k) {
    init_once();
    return g_values[k].load(std::memory_order_relaxed);
}

} // namespace cosdev

using namespace cosdev;

extern "C" int32_t cos_tuning_set(const char *name, int64_t value) {
    if (!name) return cos_fail(COS_ERR_INVALID, "null knob name");
    const int k = find_key(name, strlen(name));
    if (k < 0) return cos_fail(COS_ERR_INVALID, "unknown tuning knob '%s'", name);
    if (value == TUNE_UNSET) return cos_fail(COS_ERR_INVALID, "INT64_MIN is the 'unset' marker: use cos_tuning_clear");
    init_once();
    g_values[k].store(value, std::memory_order_relaxed);
    return COS_OK;
}

extern "C" int32_t cos_tuning_clear(const char *name) {
    init_once();
    if (!name) { // every knob back to its built-in default
        for (auto &v : g_values) v.store(TUNE_UNSET, std::memory_order_relaxed);
        return COS_OK;
    }
    const int k = find_key(name, strlen(name));
    if (k < 0) return cos_fail(COS_ERR_INVALID, "unknown tuning knob '%s'", name);
    g_values[k].store(TUNE_UNSET, std::memory_order_relaxed);
    return COS_OK;
}

extern "C" int32_t cos_tuning_get(const char *name, int64_t *out_value, int32_t *out_is_set) {
    if (!name || !out_value || !out_is_set) return cos_fail(COS_ERR_INVALID, "null argument");
    const int k = find_key(name, strlen(name));
    if (k < 0) return cos_fail(COS_ERR_INVALID, "unknown tuning knob '%s'", name);
    const long long v = tune((TuneKey)k);
    *out_is_set = v != TUNE_UNSET;
    *out_value = v == TUNE_UNSET ? 0 : v;
    return COS_OK;
}
