/*
 * cosdata_hip.h — C ABI of the MI355X-native ANN query engine for cosdata's dense/hybrid
 * search path (libcosdata_hip.so, hand-written HIP kernels for gfx950).
 *
 * The reference (cosdata/cosdata @ 2025-09-19) has no FFI/plugin seam; this header introduces
 * it at the L3<->L2 line of SURVEY.md §1.  Every entry point names the reference interface it
 * stands behind (paths relative to the reference root).  INTEGRATION.md shows the Rust
 * `extern "C"` block and the call sites a maintainer would redirect.
 *
 * Conventions
 *   - plain pointers and sizes, POD structs with an explicit struct_size; no C++/torch types;
 *   - every function returns a cos_status (0 = ok) and never unwinds or aborts across the ABI;
 *     cos_last_error_string() gives a per-thread message for the last non-zero status;
 *   - host buffers are borrowed for the duration of the call only; the library owns all device
 *     memory behind the opaque handle (exception: COS_UPLOAD_BORROW_DEVICE);
 *   - search entry points are thread-safe on a shared handle (the reference calls search from
 *     rayon workers, indexes/mod.rs:268-271); uploads/builds are exclusive;
 *   - internal ids follow the reference: vectors 0..n-1 (collection.rs:451-468, sequential),
 *     root = 0xFFFFFFFF (vector_store.rs:47), query = 0xFFFFFFFE (indexes/hnsw/mod.rs:398).
 */
#ifndef COSDATA_HIP_H
#define COSDATA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COS_ABI_VERSION 1u

/* models/types.rs:462-468 DistanceMetric */
typedef enum { COS_METRIC_COSINE = 0, COS_METRIC_EUCLIDEAN = 1, COS_METRIC_HAMMING = 2, COS_METRIC_DOT = 3 } cos_metric;
/* quantization/mod.rs:20-25 StorageType; SubByte carries its resolution separately */
typedef enum { COS_STORAGE_U8 = 0, COS_STORAGE_SUBBYTE = 1, COS_STORAGE_F16 = 2, COS_STORAGE_F32 = 3 } cos_storage;

typedef enum {
    COS_OK = 0,
    COS_ERR_STORAGE_MISMATCH = 1, /* DistanceError::StorageMismatch  -> WaCustomError::QuantizationMismatch */
    COS_ERR_CALCULATION = 2,      /* DistanceError::CalculationError (zero norm; cosine.rs:228-232)          */
    COS_ERR_INVALID = 3,          /* bad params / sizes / ordering of an upload                              */
    COS_ERR_UNIMPLEMENTED = 4,    /* arms the reference leaves `unimplemented!()` or not yet on the GPU      */
    COS_ERR_HIP = 5,              /* HIP runtime error (message has the hipError string)                     */
    COS_ERR_NOT_READY = 6,        /* search before vectors + graph are resident                              */
    COS_ERR_NO_DEVICE = 7         /* no usable gfx950 device: the product path never falls back to the CPU   */
} cos_status;

#define COS_ROOT_ID 0xFFFFFFFFu
#define COS_QUERY_ID 0xFFFFFFFEu
#define COS_SLOT_EMPTY 0xFFFFFFFDu /* a null neighbour pointer (prob_node.rs:97) in flat graph uploads */

/* visited-set semantics of the walk */
#define COS_VISITED_REF 0u   /* PerformantFixedSet replica (models/fixedset.rs): 64*M-bit lossy filter -> ID parity */
#define COS_VISITED_EXACT 1u /* exact visited set (recall mode; not ID-identical to the reference) */

/* upload flags */
#define COS_UPLOAD_DEFAULT 0u
#define COS_UPLOAD_BORROW_DEVICE 1u /* raw f32 pointer is device memory the caller keeps alive; no copy */

/* HNSWHyperParams (indexes/hnsw/types.rs:10-17) + the index fields search needs
 * (indexes/hnsw/mod.rs:59-78) + config.search.shortlist_size (config.toml:32). */
typedef struct {
    uint32_t struct_size; /* = sizeof(cos_params) */
    uint32_t abi_version; /* = COS_ABI_VERSION */
    uint32_t dim;
    uint32_t metric;     /* cos_metric */
    uint32_t storage;    /* cos_storage */
    uint32_t resolution; /* SubByte bits (1..3) */
    float range_lo, range_hi; /* values_range */
    uint32_t num_layers;             /* levels 0..num_layers */
    uint32_t neighbors_count;        /* M  (power of two, <= 256) */
    uint32_t level0_neighbors_count; /* M0 (power of two, <= 256) */
    uint32_t ef_construction;        /* <= 16384: up to 1024 the fast walk kernels, above it walk_general_kernel */
    uint32_t ef_search;              /* the same domain; cos_index_set_ef_search moves it on a live handle */
    uint32_t shortlist_size;         /* config.toml [search]: any; more than 64 scanned slots per node walk with walk_general_kernel */
    uint32_t visited_mode; /* COS_VISITED_* */
    int32_t device;        /* HIP device ordinal */
    uint32_t id_base;      /* shard support: global id of local vector 0; output ids = id_base + local id */
    uint32_t reserved;
    uint64_t seed;         /* GPU builder: level draws + root vector */
} cos_params;

typedef struct cos_index cos_index; /* opaque: one immutable device-resident index snapshot (one shard) */

/* Per-batch counters written by the kernels; the roofline's ALGORITHMIC bytes come from these. */
typedef struct {
    uint64_t evals;        /* distance evaluations (walk) */
    uint64_t expansions;   /* popped + expanded nodes */
    uint64_t adj_bytes;    /* sum over expansions of min(M_level, shortlist_size) * 4: the scanned slots of the row */
    uint64_t rerank_rows;  /* raw f32 rows gathered by the exact rerank */
    float walk_ms;         /* HIP-event time of the walk kernel of the last call on this stream (0 if timing off) */
    float finalize_ms;
    float prep_ms;
    uint32_t reserved;
} cos_search_stats;

/* ---- lifecycle ------------------------------------------------------------------------------ */
/* api_service.rs:26 init_hnsw_index_for_collection -> HNSWIndex::new (indexes/hnsw/mod.rs:81). */
int32_t cos_index_create(const cos_params *params, cos_index **out);
int32_t cos_index_destroy(cos_index *ix);
const char *cos_last_error_string(void);
int32_t cos_device_count(int32_t *out);
/* Diagnostic (not a reference interface): empirical HBM ceilings for the roofline report, SURVEY.md 8(d).
 * kind 0 = streaming read, 1 = streaming copy (read+write bytes), 2 = random gather of row_bytes-sized rows
 * (the walk's access pattern; row_bytes multiple of 16, <= 1024).  Allocates buffer_bytes (x2 for copy),
 * times `iters` launches with HIP events, returns GB/s (1e9 B/s). */
int32_t cos_hbm_probe(int32_t device, uint32_t kind, uint64_t buffer_bytes, uint32_t row_bytes, uint32_t iters,
                      double *out_gbps);

/* ---- data upload (host side of the snapshot; the Rust host walks its graph once) ------------ */
/* Collection::index_embeddings' raw f32 values (collection.rs:368, get_raw_emb_by_internal_id) for
 * internal ids [0,n).  The library quantizes them on the device with the index's StorageType
 * (ScalarQuantization::quantize, quantization/scalar.rs:10-52) and keeps raw for the exact rerank. */
int32_t cos_index_upload_vectors(cos_index *ix, const float *raw, uint32_t n, uint32_t flags);
/* Re-uploading vectors starts a new collection snapshot: the graph, the pseudo-root component AND the metadata schema
 * (cos_index_enable_metadata: mdim, max_replicas_per_node, the node table) are dropped — ids go back to id = vector row — so a host
 * that filters must call cos_index_enable_metadata again before cos_index_upload_meta_nodes / cos_index_build_meta. */
/* create_root_node's random vector (vector_store.rs:30-36), id u32::MAX. */
int32_t cos_index_set_root(cos_index *ix, const float *root_raw);
/* One HNSW level as flat arrays exported from ProbNode (prob_node.rs:97-109):
 * node_ids[n_nodes] ascending with COS_ROOT_ID last; nbr_ids[n_nodes][M_level] in SLOT ORDER,
 * COS_SLOT_EMPTY for null slots.  Level 0 must hold every vector.  Child links are implied
 * (same id one level down, vector_store.rs:897-903). */
int32_t cos_index_upload_graph_level(cos_index *ix, uint32_t level, uint32_t n_nodes, const uint32_t *node_ids,
                                     const uint32_t *nbr_ids);
int32_t cos_index_level_count(const cos_index *ix, uint32_t level, uint32_t *n_nodes);
int32_t cos_index_download_graph_level(const cos_index *ix, uint32_t level, uint32_t *node_ids, uint32_t *nbr_ids);
/* quantized codes in the REFERENCE layout (SubByte plane-major) + mags, rows [0,n] (row n = root) */
int32_t cos_index_download_codes(const cos_index *ix, void *codes, float *mags);
int32_t cos_index_download_root(const cos_index *ix, float *root_raw);

/* Load the graph of an index the reference server persisted (its `<collection>/dense_hnsw/` directory: {k}.index node records,
 * nodes.ptr, prop.data — models/serializer/hnsw/node.rs:19-101, neighbors.rs:22-61, latest_node.rs:18-44,
 * models/file_persist.rs:58-108) into the handle: every level's node ids and neighbour ids in slot order (as
 * cos_index_upload_graph_level would receive them) plus the root's stored code.  The raw vectors must already be resident
 * (cos_index_upload_vectors: the files hold only quantized codes, the exact rerank needs raw f32); root_ptr_offset =
 * HNSWIndexData.root_vec_ptr_offset (indexes/hnsw/mod.rs:47-57, kept in LMDB by the host).  COS_LOAD_VERIFY_CODES also
 * compares every stored Storage with the device's own quantization of the uploaded vector (StorageMismatch on a difference).
 * Collections with a metadata schema are refused (Unimplemented).  Format parity is UNPINNED: no reference-written file
 * exists in this image; see cosdata_amd/csrc/ref_index_reader.hip. */
#define COS_LOAD_DEFAULT 0u
#define COS_LOAD_VERIFY_CODES 1u
int32_t cos_index_load_reference_dir(cos_index *ix, const char *dense_hnsw_dir, uint32_t root_ptr_offset, uint32_t flags);
/* Host-only views of the same directory (no device needed): node count per level, then one level as the flat arrays
 * cos_index_upload_graph_level takes (node_ids ascending, root last; nbr_ids [n][M_level], COS_SLOT_EMPTY = null slot). */
int32_t cos_reference_dir_level_counts(const char *dense_hnsw_dir, uint32_t num_layers, uint32_t neighbors_count,
                                       uint32_t level0_neighbors_count, uint32_t *level_counts /* [num_layers + 1] */);
int32_t cos_reference_dir_read_level(const char *dense_hnsw_dir, uint32_t num_layers, uint32_t neighbors_count,
                                     uint32_t level0_neighbors_count, uint32_t level, uint32_t *node_ids, uint32_t *nbr_ids);

/* vector_store::index_embeddings (vector_store.rs:714) on the device: builds every level for the
 * uploaded vectors with the reference's edge semantics, batch-synchronously. */
int32_t cos_index_build(cos_index *ix, uint32_t batch_size);
/* vector_store::index_embeddings called AGAIN on a live index (vector_store.rs:714-780; index_embedding :782-975 per vector): m more
 * vectors take the internal ids [n, n + m) (sequential: collection.rs:451-468) and are inserted into the RESIDENT graph — no rebuild —
 * by the schedule of cos_index_build continued at inserted = n: level draws from the same RNG stream (the draws a full build would
 * have given these ids), batches of min(batch_size, max(1, inserted / 4)) walking the snapshot that precedes them, the link kernels on
 * the link state cos_index_build left on the handle (similarity of every neighbour slot + every node's cached lowest slot,
 * prob_node.rs:108).  The graph equals a full build's when the earlier build ended on one of its batch boundaries, and always the
 * oracle's coso_index_append_vectors + coso_index_build_rounds_continue.
 *   raw, flags = 0:                        HOST pointer to the m new rows [m][dim]; the index owns its raw rows
 *   raw, flags = COS_UPLOAD_BORROW_DEVICE: DEVICE pointer to the caller's WHOLE grown table [n + m][dim] (rows [0, n) unchanged), borrowed
 *                                          from now on — for an index whose rows were uploaded with the same flag
 * COS_ERR_NOT_READY: no graph, or no link state to continue from (the graph was uploaded, its root replaced, or the state released).
 * Exclusive like every graph change (no search in flight); a failed append leaves the handle without a graph. */
int32_t cos_index_append(cos_index *ix, const float *raw, uint32_t m, uint32_t flags, uint32_t batch_size /* 0 = 4096 */);
/* vector_store::delete_embedding (vector_store.rs:1206-1400) for the internal ids of `ids`, one after the other in the given order: per id
 * one walk for the vector's own code on every level (ef 512, keep 100, filters not pre-seeded, :1232-1248); on every level whose list
 * holds the node, its neighbours drop their back edges (remove_neighbor_by_id, prob_node.rs:285-306), a neighbour left without any
 * neighbour is linked again from the walk's results re-scored against it (:1305-1357), and the node's own slots are emptied.  The
 * vector's rows stay resident (ids are never reused, collection.rs:451-468); no walk reaches the node again (exhaustive scans —
 * cos_flat_search_batch, cos_bruteforce_topk — still see the row: the host filters deleted ids there, as it owns the id map).
 * Needs the link state like cos_index_append; exclusive like every graph change.  oracle: coso_index_delete. */
int32_t cos_index_delete(cos_index *ix, const uint32_t *ids, uint32_t m);
/* The link state of an UPLOADED graph (cos_index_upload_graph_level, cos_index_load_reference_dir), as the reference has it after a reload:
 * slot similarities (persisted by the reference, serializer/hnsw/neighbors.rs:22-61; recomputed here on the resident codes) and every
 * node's cached lowest slot by the deserializer's rule (ProbNode::new_with_neighbors_and_versions, prob_node.rs:145-181: the first empty
 * slot, else the first strictly smallest similarity).  Afterwards cos_index_append / cos_index_delete work on the uploaded graph; appends
 * draw their levels from the seed's stream advanced past the resident vectors.  Every storage an index can have. */
int32_t cos_index_restore_link_state(cos_index *ix);
/* frees the link state (as large as the adjacency: 4 bytes per neighbour slot); cos_index_append then returns COS_ERR_NOT_READY */
int32_t cos_index_release_link_state(cos_index *ix);

/* ---- search --------------------------------------------------------------------------------- */
/* IndexOps::batch_search -> HNSWIndex::search_internal (indexes/mod.rs:260, indexes/hnsw/mod.rs:390):
 * quantize query -> ann_search (every level, ef_search) -> finalize_ann_results (exact f32 rerank).
 * queries [B][dim] raw f32; outputs [B][top_k] (internal id + id_base, cosine score), out_counts[B].
 * Like the reference's collect::<Result<_>>, any failing query fails the call (first error
 * returned); out_status[B] (optional) has the per-query cos_status. */
int32_t cos_search_batch(cos_index *ix, const float *queries, uint32_t B, uint32_t top_k, uint32_t *out_ids,
                         float *out_scores, uint32_t *out_counts, int32_t *out_status);
/* Same, all pointers DEVICE memory, enqueued on `stream` (hipStream_t) without synchronising.
 * Statuses land in d_out_status[B] (required).  One in-flight batch per stream; concurrent host
 * threads use distinct streams. */
int32_t cos_search_batch_device(cos_index *ix, const float *d_queries, uint32_t B, uint32_t top_k, uint32_t *d_out_ids,
                                float *d_out_scores, uint32_t *d_out_counts, int32_t *d_out_status, void *stream);
/* ann_search's raw per-level output (vector_store.rs:256): ids/sims [B][(num_layers+1)][100],
 * counts [B][num_layers+1], top level first.  Host buffers. */
int32_t cos_ann_search_batch(cos_index *ix, const float *queries, uint32_t B, uint32_t *out_ids, float *out_sims,
                             uint32_t *out_counts, int32_t *out_status);
/* Dynamic batching for the host API: concurrent cos_search_batch calls (rayon workers / request handlers) with
 * the same top_k are fused into one launch of up to max_queries queries.  The request that opens a launch issues it when it is
 * full, or — while fewer than two coalesced launches of the handle are in flight — window_us after its last arrival (at most 8
 * windows after its first); while two are in flight it keeps gathering and goes when one of them completes, so under load the
 * launches grow to what the callers offer and an idle device answers after one window.  Choose max_queries near HALF of what the
 * callers offer at once (64 callers x 256 queries: 8 192), so that two launches alternate — one on the device while the other is
 * copied in and out.  max_queries = 0 (default) turns it off.  Results are identical to un-coalesced calls.  Not a reference
 * interface. */
int32_t cos_index_set_coalescing(cos_index *ix, uint32_t max_queries, uint32_t window_us);
/* What the batching did since the last cos_index_set_coalescing (diagnostic; bench.py reports it next to the callers' rate). */
typedef struct cos_coalescing_stats {
    uint64_t launches;         /* coalesced launches issued */
    uint64_t queries;          /* queries they carried */
    uint64_t requests;         /* cos_search_batch calls they carried */
    uint64_t closed_full;      /* launches that left because max_queries was reached */
    uint64_t closed_quiet;     /* ... because window_us passed without an arrival */
    uint64_t closed_deadline;  /* ... because 8 windows passed since the first request */
    uint64_t solo_calls;       /* calls below max_queries that found every slot in flight and launched on their own */
} cos_coalescing_stats;
int32_t cos_index_coalescing_stats(cos_index *ix, cos_coalescing_stats *out);
int32_t cos_index_set_ef_search(cos_index *ix, uint32_t ef_search);
int32_t cos_index_set_visited_mode(cos_index *ix, uint32_t mode);
/* Latency mode.  A launch of at most max_queries queries (default COS_LATENCY_MODE_DEFAULT_MAX_B; 0 = never) runs the latency
 * variant of the walk where it applies (u8 / quaternary storage, ef <= 256, reference visited filter): the code rows of every
 * entry of the lookahead window are fetched speculatively in one go and committed in pop order.  Same results bit for bit;
 * it trades bandwidth (idle on a small launch) for dependent round trips.  The reference's caller submits one 256-query batch
 * at a time per worker (indexes/mod.rs:260-272); bigger launches (coalesced batches) keep the throughput kernel. */
/* Since round 4 a launch over u8 codes has a level table (cos_index_set_walk_table, on by default) and then takes the throughput kernel
 * at every size — with the table it is faster than the one-wave latency kernel from 256 to 2048 queries per launch — so this knob
 * now selects the kernel only for quaternary storage, or with the table switched off. */
#define COS_LATENCY_MODE_DEFAULT_MAX_B 2048u
int32_t cos_index_set_latency_mode(cos_index *ix, uint32_t max_queries);
/* Four waves per query (kernels_walk_lat4.hip) for the smallest launches — at most max_queries queries (default
 * COS_LATENCY_WAVES_DEFAULT_MAX_B: one or two client batches; 0 = never): every query gets a workgroup of four waves, each owning one
 * entry of the lookahead window for the speculative half of a round, and the commit is data-parallel over the window.  Same conditions
 * and same results, bit for bit, as the one-wave latency variant; it exists for the reference's literal unit of work, one
 * `query-batch = 256` per worker (indexes/mod.rs:260-272): 0.94 vs 1.05 ms per batch at ef 64, 1.87 vs 2.35 ms at ef 256 (1M x 768 u8);
 * from 1024 queries per launch on the one-wave kernel is faster (two of these workgroups fill a CU's registers). */
/* With a level table (u8, the default) the four-wave kernel is taken above ef_search 64 only (it reads the table too: 1.73 vs 2.20 ms
 * per batch at ef 256); up to ef 64 the throughput kernel with the table is faster (0.77 vs 0.82 ms). */
#define COS_LATENCY_WAVES_DEFAULT_MAX_B 512u
int32_t cos_index_set_latency_waves(cos_index *ix, uint32_t max_queries);
/* Locality order of big launches — at least min_queries queries (default COS_WALK_ORDER_DEFAULT_MIN_B; 0 = never).  Not a reference
 * interface: the reference answers each query on its own rayon task (indexes/mod.rs:260-272) and has no launch whose order could
 * matter; on the device the order decides which rows a query finds in cache.  Such a launch walks in two steps of the same kernel:
 * the levels from the top down to a KEY LEVEL in arrival order — the key level is the lowest level whose code rows take at most
 * 64 MB (1M x 768 u8: level 2; cos_index_walk_order_cuts reports it; COS_WALK_SPLIT overrides) — then the launch is sorted by the
 * position, in a depth-first order of the key level's graph, of the best node each query found there, and EVERY level below the
 * key level (not only level 0) runs with the sorted queries dealt to the XCDs in contiguous runs (workgroup b runs on XCD b % 8,
 * each XCD has its own L2), so the waves resident on an XCD walk neighbouring regions of the graph.  The rerank follows the same
 * order.  Every query's walk, and so every result, is bit for bit what it is without the order.
 * Applies at ef_search <= 256: wider beams are bound by their own serial work and the second launch's tail costs more than the order saves. */
#define COS_WALK_ORDER_DEFAULT_MIN_B 8192u
int32_t cos_index_set_walk_order(cos_index *ix, uint32_t min_queries);
/* The levels after which such a launch is cut (descending; the launch is re-sorted after each): by default ONE, the lowest level whose
 * code rows take at most 64 MB; none (*out_n = 0) if the graph has no level the order can use.  Diagnostic: bench.py reports it. */
int32_t cos_index_walk_order_cuts(cos_index *ix, uint32_t *out_levels, uint32_t cap, uint32_t *out_n);
/* Level table — launches of at least min_queries queries over u8 codes (default COS_WALK_TABLE_DEFAULT_MIN_B = every launch; 0 =
 * never).  With a table a launch takes the throughput kernel, except one client batch above ef_search 64, which takes the four-wave
 * latency kernel (it reads the table too); the one-wave latency kernel then only serves storages without a table.  Not a reference
 * interface.  The graph's top levels are small and walked by every query (1M vectors: levels >= 4 hold 5 300 nodes), so
 * for the levels >= L_t the launch first computes similarity(query, node) for EVERY node of those levels as one exact-integer i8 MFMA
 * GEMM (dot_product_u8's integer, the same `as f32` and the same division by |q| * |v|, cosine.rs:223-235), and the walk of those
 * levels reads the similarity (4 bytes) where it would have gathered and dotted a code row.  Which nodes a walk visits, the lossy
 * visited filter and every result are bit for bit what they are without the table.
 * L_t: with max_cols = COS_WALK_TABLE_AUTO (the default) a level takes part while it holds at most c x ef_search x neighbors_count
 * nodes (c = 8 up to ef_search 64, 6 above), level by level from the top (a level's walk evaluates a multiple of ef x M rows per
 * query; the table costs one GEMM column per node): 1M x 768 at ef 64 / M 32 -> levels >= 3 (20 903 columns); a 12.5M x 1024 shard at ef 128 / M 64 -> levels >= 4 (65 161
 * columns, +45 % QPS over a fixed 8 192).  An explicit max_cols caps the columns of all table levels together; 0 = no table.  The
 * operand follows ef_search (cos_index_set_ef_search rebuilds it on the next big launch).
 * A launch holds queries x columns x 4 bytes of table (32 768 x 20 928: 2.7 GB; 32 768 x 65 184: 8.5 GB); a handle's tables together
 * stay under 48 GiB (tuning knob walk_table_max_bytes): a launch whose workspace would exceed that walks without a table.
 * Device memory of the search side in general: every caller stream (cos_search_batch_device) and every host call in flight
 * (cos_search_batch: up to 32 leased pipes, a lone big call uses up to five workspaces) owns a workspace sized for its largest launch
 * — query codes, per-level result lists ((num_layers + 1) x 100 x 8 B per query: 8 KB at ten levels), statistics, the level table —
 * and keeps it until the handle is destroyed. */
#define COS_WALK_TABLE_DEFAULT_MIN_B 1u
#define COS_WALK_TABLE_DEFAULT_MAX_COLS 8192u /* kept for callers that want the round-4 mid-round default */
#define COS_WALK_TABLE_AUTO 0xFFFFFFFFu
int32_t cos_index_set_walk_table(cos_index *ix, uint32_t max_cols, uint32_t min_queries);
/* the table the next big launch would use: its lowest level and its columns (0, 0 = none).  Diagnostic: bench.py reports it. */
int32_t cos_index_walk_table_info(cos_index *ix, uint32_t *out_level_min, uint32_t *out_cols);
/* The last completed batch on `stream` split by dispatch, for the roofline report: a big launch is up to three
 * pieces of work with three different bounds — the level-table GEMM (MFMA), the level range above the cut (rows that stay in
 * L2 / the memory-side cache: a gather from cache) and the level range below it (HBM).  Times are HIP events on the streams the
 * kernels ran on (0 without cos_index_enable_timing); counters as in cos_search_stats. */
typedef struct {
    uint32_t struct_size;       /* in: sizeof(cos_walk_split) */
    uint32_t queries;
    uint32_t table_level_min;   /* 0 = the launch used no table */
    uint32_t table_cols;
    uint32_t cut_after_level;   /* 0 = the walk was one dispatch (everything is in `lower`) */
    uint32_t reserved;
    float table_ms;             /* level-table GEMM (+ the queries' code sums) */
    float upper_ms, sort_ms, lower_ms;
    double table_int8_ops;      /* 2 * queries * cols * padded dims */
    uint64_t table_evals;       /* evaluations the table served (all of them above the cut) */
    uint64_t upper_evals, upper_expansions, upper_adj_bytes; /* levels above the cut, table evaluations included */
    uint64_t lower_evals, lower_expansions, lower_adj_bytes; /* levels below the cut (or the whole walk) */
} cos_walk_split;
int32_t cos_index_last_walk_split(cos_index *ix, void *stream, cos_walk_split *out);
/* counters + kernel times of the last completed batch on `stream` (NULL = the host API's stream).  A lone big cos_search_batch call
 * runs as up to four chunk launches (each with its own workspace): with stream == NULL the figures then describe the LAST chunk only
 * (about a quarter of the call's queries; cos_walk_split.queries says how many) — whether a call is chunked depends on what else is
 * in flight on the handle, so callers that want whole-batch counters use cos_search_batch_device on their own stream. */
int32_t cos_index_enable_timing(cos_index *ix, int32_t on);
int32_t cos_index_last_stats(cos_index *ix, void *stream, cos_search_stats *out);
/* HIP-event times of the (up to 128 most recent) launches enqueued on `stream` since cos_index_enable_timing(1):
 * sums / extremes per kernel, read after the fact so the measured launches are never synchronised in between. */
typedef struct {
    uint32_t launches;
    float prep_ms_sum, walk_ms_sum, finalize_ms_sum;
    float walk_ms_min, walk_ms_max;
} cos_timing_summary;
int32_t cos_index_timing_summary(cos_index *ix, void *stream, cos_timing_summary *out);

/* ---- metadata-filtered search ------------------------------------------------ */
/* Collections with a metadata schema index every embedding several times (metadata/mod.rs:128-146, vector_store.rs:484-640):
 * a Base replica under the main root and one Metadata replica per field combination under the PSEUDO root, next to the pseudo
 * nodes created with the index (api_service.rs:135-210).  A search with a Filter starts at the pseudo root and walks that second
 * component once per QueryFilterDimensions with one shared visited filter per level (indexes/hnsw/mod.rs:413-423,
 * vector_store.rs:273-313); the distance depends on the kinds of node and query (distance/cosine.rs:36-102).  Schema -> dimension
 * encoding (metadata/schema.rs, query_filtering.rs) stays on the host; the device takes the numeric form.
 * Ids: an embedding reserves max_replicas_per_node consecutive internal ids (collection.rs:445-468): vector row r has base id
 * r * max_replicas_per_node (also in the BASE graph from now on), replica i = base + i; pseudo root = u32::MAX - 257, pseudo
 * nodes follow it.  Returned ids are replica ids; scores are the exact cosine against the embedding's raw vector. */
int32_t cos_index_enable_metadata(cos_index *ix, uint32_t mdim /* <= 64 */, uint32_t max_replicas_per_node);
/* node table of the pseudo-root component: ascending replica ids (pseudo root and pseudo nodes last), mbits [n_nodes][mdim] */
int32_t cos_index_upload_meta_nodes(cos_index *ix, uint32_t n_nodes, const uint32_t *node_ids, const int32_t *mbits);
/* one level of the component, same conventions as cos_index_upload_graph_level (the pseudo root is on every level) */
int32_t cos_index_upload_meta_graph_level(cos_index *ix, uint32_t level, uint32_t n_nodes, const uint32_t *node_ids,
                                          const uint32_t *nbr_ids);
/* The component BUILT on the device instead of uploaded: index_embedding for the pseudo nodes (ascending id), then for the
 * Metadata replicas (ascending id) of the node table — every level walked from the pseudo root with the node's own metadata
 * dimensions as the filter, keep 64, then create_node_edges with the refusal rules of vector_store.rs:1017-1041 (a Metadata replica
 * links to a pseudo node only on an exact match; two Metadata replicas whose dimensions disagree are not linked) — in the same
 * batch-synchronous schedule as cos_index_build.  max_levels[n_nodes] = the level each node was drawn (the reference draws with
 * thread_rng: replicas from generate_level_probs, pseudo nodes from pseudo_level_probs, metadata/mod.rs:182-209); rows whose
 * metadata dimensions are all zero are Base replicas and stay out of the component.  Replaces the node table and every level. */
int32_t cos_index_build_meta(cos_index *ix, uint32_t n_nodes, const uint32_t *node_ids, const int32_t *mbits, const uint8_t *max_levels,
                             uint32_t batch_size /* 0 = 1024 */);
/* the component back on the host: node count of a level; (node_ids ascending [n_l], nbr_ids [n_l][M_l], COS_SLOT_EMPTY = null slot) */
int32_t cos_index_meta_level_count(cos_index *ix, uint32_t level, uint32_t *out);
int32_t cos_index_download_meta_graph_level(cos_index *ix, uint32_t level, uint32_t *node_ids, uint32_t *nbr_ids);
/* HNSWIndex::search_internal with DenseSearchInput(query, Some(filter)): the filters of query b are rows
 * [filter_offsets[b], filter_offsets[b+1]) of filter_dims[][mdim] (values -1 / 0 / 1, filter_encoded_dimensions'
 * output).  Host buffers; error behaviour as cos_search_batch. */
int32_t cos_search_filtered_batch(cos_index *ix, const float *queries, uint32_t B, const uint32_t *filter_offsets,
                                  const int8_t *filter_dims, uint32_t top_k, uint32_t *out_ids, float *out_scores,
                                  uint32_t *out_counts, int32_t *out_status);
/* the filtered ann_search's per-level lists before finalisation (layout of cos_ann_search_batch) */
int32_t cos_ann_search_filtered_batch(cos_index *ix, const float *queries, uint32_t B, const uint32_t *filter_offsets,
                                      const int8_t *filter_dims, uint32_t *out_ids, float *out_sims, uint32_t *out_counts,
                                      int32_t *out_status);

/* ---- operators (L1 of SURVEY.md §1) --------------------------------------------------------- */
/* QuantizationMetric::quantize (models/types.rs:504) for n vectors; codes in the REFERENCE layout
 * (u8: dim bytes; SubByte: resolution planes x ceil(dim/8) bytes plane-major, plane 0 = MSB;
 * f16: dim x 2; f32: dim x 4), mags[n].  Host buffers. */
int32_t cos_quantize_batch(uint32_t storage, uint32_t resolution, uint32_t dim, float range_lo, float range_hi,
                           const float *x, uint32_t n, void *codes, float *mags);
size_t cos_code_bytes(uint32_t storage, uint32_t resolution, uint32_t dim);
/* "auto" quantization: HNSWIndex::sample_embedding + finalize_sampling (indexes/hnsw/mod.rs:202-351) over the
 * first sample_threshold embeddings: the range is the first of +-{.025,.05,.1,.2,.3,.4,.5} whose tail holds at
 * most clamp_margin_percent (config.toml:38, default 1.0) of all sampled values, else +-1.0.  Host buffer. */
int32_t cos_sample_values_range(const float *x, uint32_t n, uint32_t dim, float clamp_margin_percent, float *range_lo,
                                float *range_hi);
/* DistanceMetric::calculate (models/types.rs:469) for explicit (x_i, y_j) pairs of stored vectors
 * in the reference layout: out[p] = metric(x[pair_x[p]], y[pair_y[p]]); status[p] per pair. */
int32_t cos_distance_batch(uint32_t metric, uint32_t storage, uint32_t resolution, uint32_t dim, const void *x_codes,
                           const float *x_mags, uint32_t nx, const void *y_codes, const float *y_mags, uint32_t ny,
                           const uint32_t *pair_x, const uint32_t *pair_y, uint32_t n_pairs, float *out,
                           int32_t *status);
/* exact brute-force cosine top-k on the resident raw vectors (ground truth for recall@k):
 * MFMA f32 GEMM for candidate generation + reference-order re-score of the survivors. */
int32_t cos_bruteforce_topk(cos_index *ix, const float *queries, uint32_t B, uint32_t k, uint32_t *out_ids,
                            float *out_scores);

/* Exhaustive search over the index's QUANTIZED codes (u8 / quaternary) as an exact-integer i8 MFMA GEMM,
 * then the reference's finalisation on the full candidate set: sort by quantized score (desc, larger id
 * first), keep 5k, exact f32 rerank (vector_store.rs:404-445), top-k (k <= 12).  The result is what
 * ann_search + finalize_ann_results would return if the walk visited every node: config c3's flat scan and
 * the exhaustive per-shard mode.  Host buffers. */
typedef struct {
    float gemm_ms;          /* summed HIP-event time of the i8 MFMA GEMM launches */
    uint32_t gemm_launches;
    double int8_ops;        /* 2 * B * n * padded_dim */
    double code_bytes;      /* bytes of codes streamed from HBM (n * row bytes * query-tile rows) */
} cos_flat_stats;
int32_t cos_flat_search_batch(cos_index *ix, const float *queries, uint32_t B, uint32_t top_k, uint32_t *out_ids,
                              float *out_scores, uint32_t *out_counts, cos_flat_stats *stats /* optional */);

/* ---- hybrid (config c5) --------------------------------------------------------------------- */
typedef struct cos_bm25 cos_bm25;
/* TFIDFIndexRoot postings as CSR (models/tf_idf_index.rs, versioned_vec.rs:208-224): term hashes
 * ascending, offsets[T+1], (doc_id, stored tf) doc-id ascending within a term. */
int32_t cos_bm25_create(int32_t device, const uint32_t *term_hashes, const uint64_t *offsets, uint32_t n_terms,
                        const uint32_t *doc_ids, const float *tfs, uint32_t documents_count, cos_bm25 **out);
int32_t cos_bm25_destroy(cos_bm25 *b);
/* SparseAnnQueryBasic::search_bm25 (models/sparse_ann_query.rs:149) for B queries given as
 * pre-hashed terms (CSR q_offsets[B+1] into q_terms); outputs [B][top_k]. */
int32_t cos_bm25_search_batch(cos_bm25 *b, const uint32_t *q_terms, const uint32_t *q_offsets, uint32_t B,
                              uint32_t top_k, uint32_t *out_ids, float *out_scores, uint32_t *out_counts);
/* Same scoring with the results left in DEVICE memory (d_out_* : [B][top_k], [B][top_k], [B]) and the kernels enqueued on
 * `stream` (NULL = the handle's own stream, asynchronous): the BM25 half of a hybrid batch can run next to the dense half and
 * feed a fusion stage without a round trip through the host.  q_terms / q_offsets are host arrays (the term lookup and the
 * idf use host tables, like the reference's radix tree walk). */
int32_t cos_bm25_search_batch_device(cos_bm25 *b, const uint32_t *q_terms, const uint32_t *q_offsets, uint32_t B, uint32_t top_k,
                                     uint32_t *d_out_ids, float *d_out_scores, uint32_t *d_out_counts, void *stream);
/* RRF fusion of hybrid_search (api/vectordb/search/repo.rs:311-340) for B queries. */
int32_t cos_rrf_fuse_batch(const uint32_t *dense_ids, const uint32_t *dense_counts, uint32_t dense_stride,
                           const uint32_t *sparse_ids, const uint32_t *sparse_counts, uint32_t sparse_stride, uint32_t B,
                           float fusion_constant_k, uint32_t top_k, uint32_t *out_ids, float *out_scores,
                           uint32_t *out_counts);

/* repo::hybrid_search (api/vectordb/search/repo.rs:168-341) in one call: dense top_k*3 (walk + exact rerank) and BM25 top_k*3
 * run concurrently on two streams of the device both indexes live on, RRF fuses them there (fusion_constant_k: dtos.rs:10-12,
 * default 60); host buffers in, fused [B][top_k] out.  A failing dense query fails the call like cos_search_batch. */
int32_t cos_hybrid_search_batch(cos_index *ix, cos_bm25 *b, const float *queries, const uint32_t *q_terms, const uint32_t *q_offsets,
                                uint32_t B, uint32_t top_k, float fusion_constant_k, uint32_t *out_ids, float *out_scores,
                                uint32_t *out_counts);

/* ---- text -> BM25 terms (host code, no device needed) -------------------------- */
/* TFIDFIndex's text side: tokenize (indexes/tf_idf/mod.rs:288-308), STOPWORDS (:282-286), process_text (:310-360: tokens longer
 * than max_token_len bytes skipped, lowercase, stopwords, stem, xxhash32 seed 0, count per hash), compute_bm25_term_frequency
 * (:362-371), count_tokens (:373-389).  The stemmer is the reference's un-vendored git dependency: `stem` is the host's shim over
 * it — (ctx, token, token_len, out, out_cap) -> bytes written; NULL hashes the lowercased token unstemmed.  Terms come out by
 * ascending hash (the reference: FxHashMap order); *out_n = distinct terms (also set when cap is too small -> Invalid). */
typedef size_t (*cos_stem_fn)(void *ctx, const char *token, size_t token_len, char *out, size_t out_cap);
int32_t cos_text_process(const char *utf8, size_t len, uint32_t max_token_len, float average_document_length, float k1, float b,
                         cos_stem_fn stem, void *stem_ctx, uint32_t *out_hashes, float *out_tfs, uint32_t cap, uint32_t *out_n);
uint32_t cos_text_count_tokens(const char *utf8, size_t len, uint32_t max_token_len);
float cos_bm25_term_frequency(uint32_t count, uint32_t document_length, float average_document_length, float k1, float b);
uint32_t cos_xxhash32(const void *data, size_t len, uint32_t seed); /* twox-hash XxHash32 (indexes/tf_idf/mod.rs:343-345) */
/* The English Snowball stemmer ("Porter2") as a cos_stem_fn: pass it as `stem` to cos_text_process to get process_text's
 * `Stemmer::create()` + `stemmer.stem(&lower)` (indexes/tf_idf/mod.rs:317-340) without a host-side shim.  Restated from the
 * published algorithm the reference's git dependency (snowball-stemmer 0.1.0 @ dcbd7da, Cargo.lock:2571-2573) ports; pinned by
 * the algorithm's sample vocabulary, not by that crate (its source is not available).  ctx is ignored; returns the byte length
 * of the stem and writes at most out_cap bytes. */
size_t cos_stem_english(void *ctx, const char *token, size_t token_len, char *out, size_t out_cap);

/* ---- learned-sparse inverted index ----------------------------------------- */
typedef struct cos_sparse cos_sparse;
/* InvertedIndexRoot as CSR (models/inverted_index.rs): dims[n_dims] ascending; for dimension t and quantized key q in
 * [0, 2^bits) the vector ids vec_ids[key_offsets[t*(2^bits+1)+q] .. key_offsets[t*(2^bits+1)+q+1]) (one list per
 * (dimension, key), as InvertedIndexNode::insert files them, inverted_index.rs:176-200); posting ranges of consecutive
 * dimensions are contiguous.  Optional raw sparse vectors as CSR (row_offsets[n_vectors+1], dims ascending per row) for
 * finalize_sparse_ann_results' raw-value rerank; pass NULLs without it. */
int32_t cos_sparse_create(int32_t device, uint32_t quantization_bits, float values_upper_bound, const uint32_t *dims, uint32_t n_dims,
                          const uint64_t *key_offsets, const uint32_t *vec_ids, uint32_t n_vectors, const uint64_t *row_offsets,
                          const uint32_t *raw_dims, const float *raw_vals, cos_sparse **out);
/* InvertedIndex::insert for a whole collection (models/inverted_index.rs:168-200): vectors in id order, every (dimension, value) pair
 * pushes the id to the end of the list of (dimension, quantize(value)).  Host code, no device.  Call once with the three output arrays
 * NULL to get *n_dims (the posting count is row_offsets[n_vectors]), then with out_dims[*n_dims], out_key_offsets[*n_dims * (2^bits + 1)],
 * out_vec_ids[nnz]. */
int32_t cos_sparse_build_csr(uint32_t quantization_bits, float values_upper_bound, uint32_t n_vectors, const uint64_t *row_offsets,
                             const uint32_t *raw_dims, const float *raw_vals, uint32_t *out_dims, uint64_t *out_key_offsets,
                             uint32_t *out_vec_ids, uint32_t *n_dims);
/* cos_sparse_build_csr + cos_sparse_create; keep_raw != 0 keeps the raw vectors on the device for the raw-value rerank. */
int32_t cos_sparse_create_from_vectors(int32_t device, uint32_t quantization_bits, float values_upper_bound, uint32_t n_vectors,
                                       const uint64_t *row_offsets, const uint32_t *raw_dims, const float *raw_vals, int32_t keep_raw,
                                       cos_sparse **out);
int32_t cos_sparse_destroy(cos_sparse *s);
/* InvertedIndex::search_internal (indexes/inverted/mod.rs:278-331) -> SparseAnnQueryBasic::sequential_search
 * (models/sparse_ann_query.rs:68-147) for B queries given as CSR pairs (q_offsets[B+1] into q_dims / q_vals).
 * reranking_factor = 0: config.rerank_sparse_with_raw_values = false -> scores are the quantized similarities as f32;
 * otherwise the best top_k * reranking_factor candidates are re-scored with the raw values and sorted (<= 64 candidates).
 * Order: score descending, larger id first (the reference leaves the order of equal / unranked entries to its hash map). */
int32_t cos_sparse_search_batch(cos_sparse *s, const uint32_t *q_dims, const float *q_vals, const uint32_t *q_offsets, uint32_t B,
                                uint32_t top_k, float early_terminate_threshold, uint32_t reranking_factor, uint32_t *out_ids,
                                float *out_scores, uint32_t *out_counts);

/* Figures of the most recent cos_sparse_search_batch on this handle: HIP-event time of its kernels, and the postings the
 * reference's traversal visits for that batch (sparse_ann_query.rs:92-125: every list of a term from its first visited key on;
 * 4 B each — SURVEY.md §8d counts algorithmic bytes, not what the device layout happens to read). */
typedef struct cos_sparse_stats {
    float kernel_ms;
    uint32_t blocks;
    uint64_t postings_visited;
    uint64_t posting_bytes;
} cos_sparse_stats;
int32_t cos_sparse_last_stats(cos_sparse *s, cos_sparse_stats *out);
/* Device layout of the handle's postings: 0 = a u32 id + a u8 key per posting (5 B), 1 = one packed u32 per posting
 * (key << 24 | id + 1; what cos_sparse_create builds whenever the collection holds fewer than 2^24 - 16384 vectors: measured in
 * round 5 at 0.453 ms against 0.540 ms per 256-query batch; tuning knob sparse_layout = 0 keeps the 5-byte layout).  Same results
 * either way; a tuning choice, not a reference interface. */
int32_t cos_sparse_layout(cos_sparse *s, uint32_t *packed);

/* ---- multi-GPU helper ----------------------------------------------------------------------- */
/* S-way merge of per-shard top-k lists gathered by the caller's RCCL all-gather
 * (SURVEY.md §8e): in [S][B][k] -> out [B][k], total_cmp desc, larger id first on ties.
 * All pointers device memory; enqueued on stream. */
int32_t cos_merge_topk_device(const uint32_t *d_ids, const float *d_scores, const uint32_t *d_counts, uint32_t S,
                              uint32_t B, uint32_t k, uint32_t *d_out_ids, float *d_out_scores,
                              uint32_t *d_out_counts, int32_t device, void *stream);

/* Same merge over S packed per-shard records, each B*(2k+1) 4-byte words laid out
 * [ids B*k | scores B*k (f32 bits) | counts B] — the layout that lets the caller exchange a shard's
 * whole result with ONE all-gather per batch instead of three. */
int32_t cos_merge_topk_packed_device(const uint32_t *d_packed, uint32_t S, uint32_t B, uint32_t k,
                                     uint32_t *d_out_ids, float *d_out_scores, uint32_t *d_out_counts,
                                     int32_t device, void *stream);

/* ---- sharded serving --------------------------------------------------------- */
/* A shard set = the S id-range shards of one collection (cos_index handles, cos_params.id_base = first id of the shard) plus
 * the exchange step between them: ONE all-gather of the packed per-shard records (RCCL over xGMI) and the S-way merge.
 * It stands behind the same caller as cos_search_batch — IndexOps::batch_search (indexes/mod.rs:260-272), a single process —
 * so the Rust host gets merged global top-k lists from one call.  Two deployments:
 *   - every shard in this process (n_local == world_size, unique_id NULL): communicators via ncclCommInitAll, the per-batch
 *     collective is a group of S ncclAllGather calls, one per device; cos_shardset_search_batch is the entry point;
 *   - one process per GPU (n_local == 1 < world_size): rank `first_rank` of a world_size communicator built from the
 *     ncclUniqueId one process obtained with cos_shardset_unique_id and handed to the others; the per-batch entry point is
 *     cos_shardset_exchange_device (all-gather + merge enqueued on the caller's stream).
 * Shards that share one device are exchanged with device copies (no communicator can span them). */
typedef struct cos_shardset cos_shardset;
#define COS_SHARDSET_UNIQUE_ID_BYTES 128
int32_t cos_shardset_unique_id(uint8_t *out /* [COS_SHARDSET_UNIQUE_ID_BYTES] */);
int32_t cos_shardset_create(cos_index *const *shards, uint32_t n_local, uint32_t first_rank, uint32_t world_size,
                            const uint8_t *unique_id, cos_shardset **out);
int32_t cos_shardset_destroy(cos_shardset *ss); /* the cos_index handles stay owned by the caller */
/* queries [B][dim] raw f32 (host) -> merged global top-k over every shard: out_ids [B][top_k] global internal ids,
 * out_scores exact cosine, out_counts [B].  Order: score desc by total_cmp, larger id first on ties.  Any failing query on
 * any shard fails the call (first error), like cos_search_batch.  Thread-safe; batches are serialised per shard set. */
int32_t cos_shardset_search_batch(cos_shardset *ss, const float *queries, uint32_t B, uint32_t top_k, uint32_t *out_ids,
                                  float *out_scores, uint32_t *out_counts);
/* Process-per-GPU exchange, all pointers DEVICE memory, enqueued on `stream`: d_packed_local = this shard's packed record
 * (B*(2*top_k+1) words, written by cos_search_batch_device on the same stream) -> d_gathered [world_size][words] ->
 * merged d_out_* on every rank.  Every rank must call it the same number of times in the same order. */
int32_t cos_shardset_exchange_device(cos_shardset *ss, const uint32_t *d_packed_local, uint32_t B, uint32_t top_k,
                                     uint32_t *d_gathered, uint32_t *d_out_ids, float *d_out_scores, uint32_t *d_out_counts,
                                     void *stream);

/* ---- tuning knobs (experiments; NOT a reference interface) ------------------------------------------------------------------
 * Launch-policy knobs of the kernels — which of two implementations of one operator runs, launch sizes at which a policy
 * switches, prefetch depths.  None changes a result; the parity tests use them to compare implementations, the profiling
 * scripts to time them.  Names: cosdata_amd/csrc/tuning.h.  The library reads ONE environment variable,
 * COS_TUNING="name=value,name=value" (parsed once, on first use), so that an unmodified process can be steered under rocprofv3;
 * nothing else of the process environment influences a kernel.  A knob that is not set leaves the built-in, measured default.
 * Process-wide, thread-safe (relaxed atomics); knobs read at cos_index_create apply to handles created afterwards. */
int32_t cos_tuning_set(const char *name, int64_t value);
int32_t cos_tuning_clear(const char *name /* NULL = every knob */);
int32_t cos_tuning_get(const char *name, int64_t *out_value, int32_t *out_is_set);

#ifdef __cplusplus
}
#endif
#endif /* COSDATA_HIP_H */
