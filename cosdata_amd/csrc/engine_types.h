// engine_types.h — POD structs passed by value to the gfx950 kernels.
#pragma once
#include <stdint.h>
#include "device_common.h"

namespace cosdev {

// One HNSW level in HBM (mirrors prob_node.rs:97-109 as flat arrays, slot order preserved).
struct LevelDev {
    const u32 *adj_vec;  // [n][M] neighbour's VECTOR ROW (root = row N), ROW_EMPTY = null slot
    const u32 *adj_node; // [n][M] neighbour's node index within this level (level 0: == adj_vec)
    const u32 *node_vec; // [n]    node index -> vector row (level 0: nullptr, identity)
    const u32 *child;    // [n]    node index one level down (level 0: nullptr)
    u32 n;
    u32 M;
    u32 root_idx;
    // |v| of every SCANNED neighbour slot, beside the adjacency: adj_mag[node * mag_stride + slot] = mags[adj_vec[node * M + slot]]
    // (mag_stride = min(M, shortlist_size); empty slots hold 1).  An expansion on a row level then gets its winners' norms with the
    // adjacency row it fetches anyway (coalesced, one more line or two per expansion) instead of one 4-byte gather per winner, each
    // of which pulled a whole line from a 50 MB array (12 % of the lower range's HBM traffic at 12.5M x 1024).  Written when a
    // graph is committed (engine.hip ensure_adj_mags); nullptr while a graph is being built or after the root changed: the walk then
    // gathers mags[] as before.  Same values, same bits.
    const float *adj_mag;
    u32 mag_stride;
    // pseudo-root component (metadata-filtered search): nodes are replicas, not vectors
    const u32 *node_id;   // [n] internal (replica) id of a node; nullptr on the base graph (id = vector row * id_stride)
    const u32 *node_meta; // [n] row of the node's metadata dimensions in IndexDev::mbits
};

struct IndexDev {
    const uint8_t *codes; // [N+1][row_stride] device code layout, row N = root
    const float *mags;    // [N+1]
    const float *raw;     // [N][raw_stride_f] raw f32 (rerank)
    const float *raw_mags; // [N] sqrt(sequential sum x*x) of raw rows
    u64 row_stride;       // bytes
    u64 raw_stride;       // floats
    u32 n;                // vectors (root row index)
    u32 dim;
    u32 metric;
    u32 storage;
    u32 num_layers;
    u32 shortlist;
    u32 visited_mode;
    u32 nchunks;   // 16-byte chunks per code row (integer engines)
    u32 G;         // lanes per row (power of two, integer engines)
    u32 id_base;
    u32 id_stride; // internal id of vector row r = r * id_stride: max_replica_per_node on collections with a metadata schema
                   // (ids are reserved per embedding, collection.rs:445-468), 1 otherwise
    // metadata dimensions of the pseudo-root component's nodes (types.rs:106-147)
    const int32_t *mbits; // [n_meta][mdim]
    const float *mmags;   // [n_meta]
    u32 mdim;
    LevelDev lv[MAX_LEVELS];
};

struct WalkArgs {
    const uint8_t *qcodes; // query codes in the device layout, [*][row_stride]
    const float *qmags;
    const u32 *q_rows;   // optional: query b reads row q_rows[b] of qcodes/qmags (builder: corpus rows)
    const u32 *self_ids; // optional: id pre-inserted in the visited filter (default COS_QUERY_ID)
    // EXACT mode: per-query bitset [B][vis_words_per_query] (one bit per node of the level being walked) that is all-zero
    // between levels and launches, plus the undo log [B][vis_log_cap] of the words a level set: a level clears exactly
    // what it touched, so nothing is memset per launch (the slab would be B x N/8 bytes).
    u32 *vis_bits;
    u32 *vis_log;
    u32 vis_words_per_query;
    u32 vis_log_cap;     // >= ef * max(M) + 2
    u32 B;
    u32 ef;
    u32 keep;            // 100 (search) / 64 (indexing)
    // outputs: per (query, level) lists, top level first: slot = num_layers - level
    u32 *out_ids;        // [B][L+1][keep] internal ids (local)
    float *out_sims;     // [B][L+1][keep]
    u32 *out_nodes;      // optional [B][L+1][keep] node index within the level (builder)
    u32 *out_counts;     // [B][L+1]
    int32_t *out_status; // [B]
    // metadata-filtered search: the filters of query b are rows [f_off[b], f_off[b+1]) of f_dims[][mdim] (values -1/0/1)
    const int32_t *f_dims;
    const float *f_mags;
    const u32 *f_off;
    u64 *out_stats;      // [B][4]: evals, expansions, adj_bytes, rounds
    u64 *out_stats2;     // optional [B][4] (walk_kernel only): evals, expansions, adj_bytes of the LAST level range of a split walk
                         // (the whole walk when it is not split), evaluations served by the level table
    // Level table (big search launches over u8 codes; kernels_flat.hip launch_level_table, engine.hip ensure_level_table): for the
    // levels >= tab_level_min — few nodes, walked by every query — the dot product of every (query, node) pair is computed before
    // the walk as ONE exact-integer i8 MFMA GEMM, tab[q][tab_col0[level] + node] = dot_product_u8 `as f32` (x86_64.rs:22-66).
    // The walk of those levels reads 4 bytes (+ the node's norm, tab_mags: a few KB that stay in L2) per evaluation instead of
    // gathering and dotting a code row, and forms the cosine with the same product and quotient as on a row level
    // (cosine.rs:223-235); which nodes it visits, the lossy filter and every result are what they were (same bits: the integer
    // dot is exact either way).  nullptr = no table.
    const float *tab;
    const float *tab_mags; // [columns] |v| of the table nodes (Storage mag), same column index as tab
    u64 tab_stride;       // floats per query row
    u32 tab_level_min;
    u32 tab_col0[MAX_LEVELS];
    u32 no_self_seed;     // 1 = nothing is pre-inserted in a level's filter (delete_embedding's walks, vector_store.rs:1232-1248); walk_kernel only
    u32 smem_mmax;        // LDS layout of walk_kernel (set by launch_walk): the widest neighbour row among the levels THIS launch walks (its filter: 8 x M bytes) ...
    u32 smem_win_bytes;   // ... and the bytes of the window region: 3 x LA x 64 x 4 when a row level is walked, just the ranked merge's scratch when every
                          // level of the launch is a table level (the upper range of a split walk on the metric's shard: 6.6 -> 2.9 KB per wave, LDS no
                          // longer caps the resident waves)
    u32 merge_min;        // table levels: an expansion with at least this many winners past the screen inserts them by ONE ranked merge
                          // (walk_kernel.inc commit_merge) instead of one pool shift each; 0 = always the serial insert.  Set by launch_walk.
    // Locality-ordered walk (big search launches; kernels_order.hip, engine.hip run_search).  The walk of a launch is split
    // into launches of the same kernel over consecutive level ranges [level_first .. level_last]; between two of them the
    // launch's queries are sorted by an ORDER KEY and dealt to the XCDs in contiguous runs (workgroup b runs on XCD b % 8, each
    // XCD has its own L2), so the waves resident on one XCD walk the same region of the graph and find each other's rows in
    // cache.  The key a range leaves: the position, in a depth-first order of level_last's graph (order_rank, built once per
    // graph on the host), of the best node the query found on level_last.  A range that does not start at the top level reads
    // its entry node from entry0 (the child link of the previous range's best node).  Every query's walk is what it was: only
    // WHEN and WHERE it runs changes.  phase 0 = the whole walk in one launch, arrival order (everything else).
    u32 phase;             // 0 | 1 = this launch walks [level_first .. level_last]
    u32 level_first, level_last;
    u32 key_n;             // nodes of level_last (level_last >= 1); also the key of a failed query (sorts last)
    const u32 *order_rank; // [key_n] node of level_last -> position in the depth-first order
    const u32 *q_order;    // [B] workgroup -> query; nullptr = arrival order
    u32 *entry0;           // [B] entry node of the next range: read when level_first < L, written when level_last >= 1
    u32 *order_key;        // [B] written when level_last >= 1
    u32 *order_iota;       // [B] order_iota[b] = b (the values the sort carries), written with the key
};

// Parameter domains of the walk kernels.  walk_kernel (register pools of up to 64 x 16 keys, one lane per scanned slot) and the latency
// kernels hold ef <= WALK_FAST_MAX_EF and <= 64 scanned slots per node; everything else the reference accepts (hnsw/types.rs:10-17,
// config.toml:32) goes to walk_general_kernel (kernels_walk_general.hip), whose candidates live in ef x 8 bytes of one workgroup's LDS.
constexpr u32 WALK_FAST_MAX_EF = 1024u;
constexpr u32 WALK_GENERAL_MAX_EF = 16384u;

} // namespace cosdev
