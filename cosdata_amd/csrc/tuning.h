// tuning.h — the library's experiment knobs in ONE registry (not a reference interface).
//
// Through round 4 the kernels' launch policies read 28 process environment variables, each where it was needed.  A library the Rust
// host links must not change kernels on what happens to be in the process environment, so round 5 moved every surviving knob here:
//   * cos_tuning_set / cos_tuning_get (include/cosdata_hip.h) are the interface — profiling scripts and the parity tests that
//     compare two implementations of one operator call them;
//   * the ONLY environment variable the library reads is COS_TUNING="name=value,name=value", parsed once on first use, so that a
//     script can still steer an unmodified process (rocprofv3 runs);
//   * a knob nobody set reports TUNE_UNSET and the call site applies its measured default.  None of them changes a result.
#pragma once
#include <cstdint>

namespace cosdev {

enum TuneKey : int {
    TUNE_WALK_CHAIN_MIN_B,    // launches of at least this many queries take part in the walk chain (0 = always, 4294967295 = never); read at cos_index_create
    TUNE_WALK_ORDER_MIN_B,    // ... are split and locality-ordered (0 = one launch, arrival order); read at cos_index_create
    TUNE_WALK_SIDE_MIN_B,     // ... walk on the workspace's own low-priority stream (0 = the caller's stream); read at cos_index_create
    TUNE_WALK_SPLIT_LEVELS,   // bit l set = cut the locality-ordered walk after level l (default: the lowest level whose rows fit 64 MB)
    TUNE_WALK_TABLE_COLS,     // overrides cos_index_set_walk_table's max_cols
    TUNE_WALK_TABLE_MIN_B,    // overrides its min_queries
    TUNE_WALK_TABLE_MAX_BYTES,// budget of a handle's per-workspace tables (default 48 GiB)
    TUNE_WALK_TABLE_GEMM,     // 0 = the tile kernel (flat_codes_gemm_i8), 1 = the query-resident kernel (level_table_areg); default 1 where it exists
    TUNE_HOST_PIPELINE_MIN_B, // host calls of at least this many queries run as a chunk pipeline (default 8192; 0 = never)
    TUNE_FLAT_UNFUSED,        // 1 = exhaustive scans take the score-matrix path
    TUNE_FLAT_TILE_KERNEL,    // 1 = quaternary and u8 scans take the 256 x 128 tile kernel instead of the query-resident ones
    TUNE_FLAT_PF,             // k panels prefetched by the tile kernel of cos_flat_search_batch (1..3; default 1 — the level table's tile GEMM is fixed at 2)
    TUNE_FLAT_FP4,            // 0 = quaternary scans multiply i8 digits, 1 = e2m1 digits on the f8f6f4 MFMA (default 1: the library is built for gfx950 only)
    TUNE_BM25_BLOCKS,         // workgroups of bm25_score_kernel (default 8192)
    TUNE_SPARSE_LAYOUT,       // cos_sparse_create: 0 = u32 id + u8 key per posting, 1 = packed u32 (default: packed when ids fit 24 bits)
    TUNE_WALK_PB,             // 4 | 8 code rows in flight per wave (u8, 513..1024 dims), every launch
    TUNE_WALK_PB_UPPER,       // ... the upper range of a split walk only
    TUNE_WALK_LAT,            // overrides cos_index_set_latency_mode's max_queries
    TUNE_WALK_LAT4,           // overrides cos_index_set_latency_waves' max_queries
    TUNE_WALK_SMALL_TABLE_TK, // 0 = small launches with a level table keep the latency kernels
    TUNE_WALK_LAT_WARM,       // 0 = the one-wave latency kernel does not warm the top levels
    TUNE_WALK_LAT_LA,         // 1..4: lookahead window of the one-wave latency kernel
    TUNE_WALK_LAT4_E,         // entries consumed per round by the four-wave latency kernel
    TUNE_FINALIZE_FAST,       // 0 = the general finalize kernel alone
    TUNE_SHARDSET_FORCE_RCCL, // 1 = a world of one still goes through the RCCL exchange (tests)
    TUNE_BUILD_PROFILE,       // 1 = cos_index_build prints its phase times to stderr
    TUNE_WALK_MERGE_MIN,      // table levels: winners from which an expansion takes the ranked merge (0 = never; default 4 up to ef 64, 3 above)
    TUNE_WALK_ADJ_MAG,        // 0 = row levels gather the winners' norms from mags[] instead of reading them beside the adjacency (LevelDev::adj_mag); 2 = beside the adjacency at every ef and launch size (default 1: up to ef = 2 x the level's neighbour count, launches of 4096 queries or more)
    TUNE_WALK_TABLE_RULE_C,   // the automatic level-table rule's constant: a level takes part while it holds <= c x ef_search x neighbors_count nodes (default 10 up to ef = 1.5 x neighbors_count, 8 up to 2 x, 6 above)
    TUNE_WALK_TABLE_AFTER_SORT, // 0 = the level-table GEMM of a big launch never waits for the previous walk's order sort, 2 = always (default 1: tables of 2^30 entries or more)
    TUNE_WALK_UPPER_LDS_PAD,  // bytes of unused dynamic LDS added per wave on the upper range of a split walk (occupancy experiment; default 0; negative = a launch of table levels only keeps the full LDS layout)
    TUNE_WALK_R2,             // 0 = ef 65..128 walks with the 256-key pool of ef 129..256 (default 1: a 128-key pool)
    TUNE_FINALIZE_WIDE_MAX_B, // launches of at most this many queries finalize with eight waves per query (default 1024; 0 = never)
    TUNE_FLAT_FP4_W8,         // the FP4 scan of 768-dim quaternary codes: 0 = one wave per SIMD (flat_scan_q2_fp4), 1..4 = two waves per SIMD, flat_scan_q2_fp4_w8<ORDER = value - 1> (default 4: even deal, alternating accumulators, 128-column tiles)
    TUNE_COUNT
};

constexpr long long TUNE_UNSET = INT64_MIN;

long long tune(TuneKey k);                                   // TUNE_UNSET when nobody set it
inline long long tune_or(TuneKey k, long long dflt) {        // the value, or the call site's default
    const long long v = tune(k);
    return v == TUNE_UNSET ? dflt : v;
}

} // namespace cosdev
