// engine_internal.h — host-side structures shared by engine.hip and builder.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cosdata_hip.h"
#include "engine_types.h"
#include "tuning.h"
#include "link_types.h"

namespace cosdev {
hipError_t launch_fill_i32(int32_t *p, u64 n, int32_t v, hipStream_t st);
hipError_t launch_edge_pairs(const u32 *adj_vec, const u32 *node_vec, u32 node0, u32 cn, u32 M, u32 *px, u32 *py, hipStream_t st);
hipError_t launch_edge_keys(const u32 *adj_vec, const float *sims, const int32_t *st_in, u32 node0, u32 cn, u32 M, u32 metric, int32_t *key, int32_t *fail, hipStream_t st);
hipError_t launch_low_cache(const u32 *adj_vec, const int32_t *key, u32 n, u32 M, int32_t kmin, int32_t kmax, uint8_t *low_idx, int32_t *low_key, hipStream_t st);
hipError_t launch_unlink(const LinkArgs &a, const u32 *dn, u32 *status, hipStream_t st);
hipError_t launch_index_pair_distances(int eng, const uint8_t *codes, const float *mags, u64 row_stride, u32 nchunks, u32 dim, u32 metric, const u32 *d_pair_x,
                                       const u32 *d_pair_y, u32 n_pairs, float *d_out, int32_t *d_status, hipStream_t st); // kernels_misc.hip
hipError_t launch_grow_rows(const u32 *src, u32 *dst, u32 old_n, u32 new_n, u32 M, u32 remap_from, u32 remap_to, u32 fill, hipStream_t st);
hipError_t launch_grow_bytes(const uint8_t *src, uint8_t *dst, u32 old_n, u32 new_n, hipStream_t st);
hipError_t launch_fill_adj_mag(const u32 *adj_vec, const float *mags, float *adj_mag, u32 n, u32 M, u32 slots, hipStream_t st);
hipError_t launch_link_round(const LinkArgs &a, u32 maxM, const u32 *pend, const u32 *pcount, u32 *next, u32 *next_count, u32 *evq, u32 *evq_count,
                             u32 count_ub, u32 round, hipStream_t st);
hipError_t launch_quantize_rows(int eng, const float *x, u64 x_stride, u32 n, u32 dim, float lo, float hi, uint8_t *codes,
                                u64 row_stride, float *mags, float *raw_mags, hipStream_t st);
hipError_t launch_walk(int eng, const IndexDev &ix, const WalkArgs &wa, u32 lat_max_B, u32 lat4_max_B, hipStream_t st);
bool walk_general_needed(const IndexDev &ix, u32 ef); // kernels_walk_general.hip: ef > 1024 or more than 64 scanned slots per node
int walk_kernel_kind(int eng, const IndexDev &ix, const WalkArgs &wa, u32 lat_max_B, u32 lat4_max_B, bool table_available); // 0 throughput | 1 one-wave | 4 four-wave latency kernel
hipError_t launch_walk_meta(int eng, const IndexDev &ix, const WalkArgs &wa, hipStream_t st);
hipError_t launch_walk_meta_index(int eng, const IndexDev &ix, const WalkArgs &wa, hipStream_t st);
hipError_t launch_finalize(const IndexDev &ix, const float *queries, u64 q_stride, const float *q_raw_mags, const u32 *walk_ids,
                           const float *walk_sims, const u32 *walk_counts, const int32_t *walk_status, u32 B, u32 top_k,
                           u32 *out_ids, float *out_scores, u32 *out_counts, int32_t *out_status, u64 *out_rerank_rows,
                           hipStream_t st, const u32 *q_order = nullptr, u32 *slow_list = nullptr);
// buffers of the locality order of one workspace's big launches (kernels_order.hip)
struct WalkOrder {
    u32 cap = 0;
    u32 *entry0 = nullptr;      // [cap] level-0 entry node per query
    u32 *order_key = nullptr;   // [cap]
    u32 *keys_sorted = nullptr; // [cap]
    u32 *iota = nullptr, *vals_sorted = nullptr, *q_order = nullptr; // [cap]
    void *tmp = nullptr;        // radix sort scratch
    size_t tmp_bytes = 0;
};
hipError_t walk_order_reserve(WalkOrder &o, u32 B); // synchronous (re)allocation
void walk_order_free(WalkOrder &o);
hipError_t launch_walk_order(WalkOrder &o, u32 B, u32 key_max, u32 num_xcd, hipStream_t st);
// level table of the walk (kernels_flat.hip; WalkArgs::tab)
hipError_t launch_level_table_gather(const uint8_t *codes, const float *mags, u64 row_stride, const u32 *node_vec, u32 n, u32 col0,
                                     uint8_t *tcodes, float *tmags, hipStream_t st);
hipError_t launch_code_sums(const uint8_t *codes, u64 row_stride, u32 n, u32 *sums, hipStream_t st);
hipError_t launch_level_table(int eng, const uint8_t *qcodes, const float *qmags, u32 *qsums, uint8_t *qdig, u32 B, const uint8_t *tcodes, const float *tmags,
                              const u32 *tcsums, u64 row_stride, u32 ncols, float *tab, u64 tab_stride, u32 n_cus, hipStream_t st);
bool level_table_eng_supported(int eng, u64 row_stride); // storages the level table exists for (kernels_flat.hip)
int32_t quantize_ref_layout(uint32_t storage, uint32_t res, uint32_t dim, const float *x, uint32_t n, void *codes, float *mags);
int32_t distance_ref_layout(uint32_t metric, uint32_t storage, uint32_t res, uint32_t dim, const void *x_codes, const float *x_mags, uint32_t nx,
                            const void *y_codes, const float *y_mags, uint32_t ny, const uint32_t *pair_x, const uint32_t *pair_y, uint32_t n_pairs,
                            float *out, int32_t *status);
} // namespace cosdev

using cosdev::u32;
using cosdev::u64;

int32_t cos_fail(int32_t code, const char *fmt, ...);
#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) return cos_fail(COS_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// scoped device allocation: temporaries of an entry point are released on every return path
struct DevBuf {
    void *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { release(); return hipMalloc(&p, bytes ? bytes : 1); }
    void release() { if (p) (void)hipFree(p); p = nullptr; }
    template <typename T> T *as() const { return (T *)p; }
};

// ------------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------------
struct LevelHost {
    std::vector<u32> node_ids; // ascending, root last
    std::vector<u32> nbr_ids;  // [n][M] internal ids / COS_SLOT_EMPTY
    u32 *d_adj_vec = nullptr, *d_adj_node = nullptr, *d_node_vec = nullptr, *d_child = nullptr;
    u32 *d_node_id = nullptr, *d_node_meta = nullptr; // pseudo-root component only (metadata-filtered search)
    float *d_adj_mag = nullptr; // [n][min(M, shortlist)] norms of the scanned neighbour slots (LevelDev::adj_mag), valid while cos_index::adj_mag_valid
    u32 n = 0, M = 0;
    u32 root_idx = 0;        // pseudo-root component: node index of the pseudo root (base graph: the root is the last node)
    bool host_valid = false; // node_ids/nbr_ids mirror the device arrays
};

// What cos_index_build leaves behind so that cos_index_append can CONTINUE it (builder.hip): the link state of every level — the
// similarity key of every neighbour slot and every node's cached (lowest index, lowest similarity), prob_node.rs:108, which is
// history (remove_neighbor_by_id does not refresh it) and cannot be recomputed from the graph —, the RNG stream after the last level
// draw, and how many vectors the graph holds.  As large as the adjacency itself; cos_index_release_link_state frees it.
struct LinkState {
    DevBuf key[cosdev::MAX_LEVELS], low_idx[cosdev::MAX_LEVELS], low_key[cosdev::MAX_LEVELS], owner[cosdev::MAX_LEVELS];
    uint64_t rng = 0;
    u32 n_built = 0;
    bool valid = false;
    void release() {
        for (int l = 0; l < cosdev::MAX_LEVELS; l++) { key[l].release(); low_idx[l].release(); low_key[l].release(); owner[l].release(); }
        valid = false;
    }
};

// EXACT-mode visited filters of one stream (WalkArgs::vis_bits / vis_log).  The bitset is zeroed once, when it is
// (re)allocated; afterwards every walk level undoes its own bits.
struct VisTab {
    u32 *bits = nullptr, *log = nullptr;
    size_t bits_cap = 0, log_cap = 0; // allocated u32 words
    bool zeroed = false;
    size_t zeroed_cap = 0;
};
// sizes/allocates the filter for B queries at beam width ef and fills wa.vis_*
int32_t vis_tab_prepare(VisTab &vt, const cos_index *ix, u32 B, u32 ef, hipStream_t st, cosdev::WalkArgs &wa);

struct Workspace {
    u32 capB = 0, cap_topk = 0;
    uint8_t *q_codes = nullptr;
    float *q_mags = nullptr, *q_raw_mags = nullptr;
    u32 *walk_ids = nullptr, *walk_counts = nullptr;
    float *walk_sims = nullptr;
    int32_t *walk_status = nullptr;
    u64 *stats = nullptr;       // [B][4]
    u64 *stats2 = nullptr;      // [B][4] WalkArgs::out_stats2
    float *tab = nullptr;       // level table of this workspace's big launches [capB][tab_stride] (WalkArgs::tab), grown on demand
    size_t tab_cap = 0;         // floats
    u32 *qsums = nullptr;       // [capB] code sums of the queries (the table GEMM's recentring term)
    uint8_t *qdig = nullptr;    // [capB][dims] quaternary codes: the queries' i8 digit rows (the table GEMM's resident operand)
    u32 *fin_flags = nullptr;   // [capB + 1] finalize_fast_kernel -> finalize_list_kernel hand-over: count, then the queries (kernels_walk.hip)
    u64 *rerank_rows = nullptr; // [B]
    VisTab vis; // EXACT mode visited filters
    cosdev::WalkOrder order; // locality order of big launches (cos_index::walk_order_min_B)
    // host-API staging (device)
    float *d_queries = nullptr;
    u32 *d_out_ids = nullptr, *d_out_counts = nullptr;
    float *d_out_scores = nullptr;
    int32_t *d_out_status = nullptr;
    // filtered search (cos_search_filtered_batch): the batch's filters [f_cap words: dims | norms | offsets], grown on demand — until
    // round 6 three hipMalloc / hipFree pairs per call (a hipFree drains the device)
    unsigned char *f_buf = nullptr;
    size_t f_cap = 0;
    // timing: a ring of event quadruples (before prep | after prep | after walk | after finalize), one per launch, so a
    // run of launches can be summarised afterwards without synchronising between them (cos_index_timing_summary)
    // + the inner marks of a big launch's walk: [4] before / [5] after the level-table GEMM (caller's stream), [6] after the upper
    // level range, [7] after the order sort (walk stream); lastSplit says which of them the last launch recorded
    static constexpr u32 EV_RING = 128;
    static constexpr u32 EV_PER = 8;
    std::vector<hipEvent_t> ev; // [EV_RING][EV_PER]
    u32 ev_count = 0;           // timed launches since timing was switched on (ring position = ev_count % EV_RING)
    hipEvent_t walk_done = nullptr; // recorded after this workspace's walk kernel (walk chain, see cos_index::chain_*)
    hipEvent_t last_range = nullptr; // recorded when this workspace's walk reaches its LAST level range (after the order sort; an unsplit walk: its start)
    hipStream_t walk_stream = nullptr; // low-priority stream big walks run on (cos_index::walk_side_min_B), created on first use
    hipEvent_t prep_done = nullptr, walk_fin = nullptr; // caller's stream -> walk stream -> finalize stream
    u32 lastB = 0;
    bool timed = false;
    bool last_tab = false, last_split = false; // the last launch used the level table / was cut into two level ranges
    u32 last_tab_cols = 0, last_tab_level_min = 0, last_cut_level = 0;
};

// The pseudo-root component of a collection with a metadata schema: pseudo nodes + Metadata replicas, a graph of
// its own next to the base graph.  Node table = level 0 of the component, ascending replica id.
struct MetaGraph {
    u32 mdim = 0;                 // metadata dimensions per node (0 = the index has no metadata schema)
    std::vector<u32> node_ids;    // [n_meta] ascending
    int32_t *d_mbits = nullptr;   // [n_meta][mdim]
    float *d_mmags = nullptr;     // [n_meta]
    std::vector<LevelHost> lv;    // [num_layers + 1]
};

// Device-side resources of ONE host-API call in flight (cos_search_batch, cos_ann_search_batch, cos_search_filtered_batch):
// s[0] is the call's stream on the simple path; big batches run as a chunk pipeline (engine.hip, search_host_pipelined):
// H2D of chunk i+1 on `sc` under the walk of chunk i on s[i & 1], finalize of chunk i on `sf` under the walk of chunk i+1.
struct HostPipe {
    static constexpr u32 MAX_CHUNKS = 4;
    hipStream_t s[2] = {nullptr, nullptr}, sc = nullptr, sf = nullptr;
    hipEvent_t ev_in[MAX_CHUNKS] = {}, ev_walk[MAX_CHUNKS] = {};
    char wkey[MAX_CHUNKS] = {}; // &wkey[i] = key of chunk i's Workspace in cos_index::ws
    float *d_q = nullptr;       // [B][dim] the call's queries
    u32 *d_ids = nullptr, *d_counts = nullptr;
    float *d_scores = nullptr;
    int32_t *d_status = nullptr;
    size_t cap_q = 0, cap_ids = 0, cap_scores = 0, cap_counts = 0, cap_status = 0; // elements, one capacity per buffer
};
static constexpr u32 COS_MAX_HOST_PIPES = 32; // concurrent host-API calls served at once; further callers wait for a pipe

struct cos_index {
    cos_params p;
    int eng = -1;
    u32 n = 0;
    bool have_vectors = false, have_root = false, raw_borrowed = false;
    float *d_raw = nullptr;
    float *d_raw_mags = nullptr;
    uint8_t *d_codes = nullptr;
    float *d_mags = nullptr;
    u64 row_stride = 0;
    u32 nchunks = 0, G = 1;
    std::vector<float> root_raw;
    std::vector<LevelHost> lv;
    u32 id_stride = 1; // internal id of vector row r = r * id_stride (max_replica_per_node with a metadata schema, else 1)
    MetaGraph meta;
    hipStream_t own_stream = nullptr; // uploads / builder stream (exclusive entry points)
    std::mutex mu;                    // guards the workspace + thread-stream maps, the timing flag, ef_search and visited_mode
    std::map<void *, Workspace *> ws;
    // host-API searches: a call leases one HostPipe (private streams + staging) for its duration from a bounded pool, so
    // concurrent callers neither serialise nor share buffers and a host that churns threads cannot grow device memory
    std::mutex pipe_mu;
    std::condition_variable pipe_cv;
    std::vector<struct HostPipe *> pipes_all, pipes_free;
    std::atomic<int> host_calls_active{0};
    Workspace *last_ws = nullptr; // most recent batch (cos_index_last_stats with stream == NULL)
    // host-API request coalescing (cos_index_set_coalescing)
    std::mutex co_mu;
    std::condition_variable co_cv;
    std::vector<struct CoSlot *> co_slots; // pooled launches-in-assembly (engine.hip, CoSlot)
    struct CoSlot *co_open = nullptr;      // the slot new requests join
    u32 co_max_queries = 0, co_window_us = 0;
    u32 co_inflight = 0;                  // coalesced launches issued and not yet complete (co_mu)
    cos_coalescing_stats co_stats{};      // since the last cos_index_set_coalescing (co_mu)
    bool timing = false;
    // Walk chain: a launch big enough to fill the chip several times over (>= chain_min_B queries) gains nothing from sharing
    // it with ANOTHER stream's walk — two co-running walks only stretch each other — but its prologue/epilogue (quantize,
    // finalize: short kernels) do hide well under a neighbour's walk.  So big walks of different streams are ordered one after
    // the other with an event, while everything else of a launch stays free to overlap.
    std::mutex chain_mu;
    hipEvent_t chain_ev = nullptr; // walk_done of the most recent chained walk
    // ... and its last_range: the next launch's level-table GEMM waits for it.  Issued freely on the caller's stream the GEMM lands
    // next to the previous walk's UPPER range and order sort — and the sort's first kernel (32 workgroups) cannot be placed while the
    // GEMM's 450-register waves hold every SIMD: the chip ran the GEMM alone for 3 ms per step and the lower range waited
    // (profiles/r06_step_timeline_before.txt).  Behind this event the GEMM shares the chip with the HBM-bound lower range instead.
    hipEvent_t chain_last_range_ev = nullptr;
    // (Round 5 also chained the two level ranges of a split walk separately, so that the upper range of launch i+1 — table levels,
    // instruction issue — co-ran with the lower range of launch i — HBM.  Measured: 7.08 against 7.12 ms per step at ef 64, 17.34
    // against 17.27 at ef 256 (profiles/r05_phase_chain_probe.jsonl): both ranges are limited by the queries in flight, and two
    // co-running ranges share the wave slots.  Removed.)
    u32 chain_min_B = 16384;
    // launches of at least this many queries split their walk in two (upper levels | level 0) and run level 0 in locality order
    // (kernels_order.hip); 0 = never.  Env COS_WALK_ORDER_MIN_B.
    u32 walk_order_min_B = COS_WALK_ORDER_DEFAULT_MIN_B;
    u32 n_cus = 256; // hipDeviceAttributeMultiprocessorCount of the handle's device (persistent kernels launch one workgroup per CU)
    u32 num_xcd = 8; // hipDeviceAttributeNumberOfXccs of the handle's device (workgroup b of a grid runs on XCD b % num_xcd)
    // the order keys' tables: position of every node of a key level in a depth-first order of that level's graph, and the key
    // levels themselves, descending (ensure_order_rank, engine.hip); rebuilt after the graph changes
    u32 *d_order_rank[cosdev::MAX_LEVELS] = {};
    u32 order_rank_n[cosdev::MAX_LEVELS] = {};
    std::vector<u32> order_levels; // empty = the graph has no level the order could use
    bool order_rank_valid = false;
    // Level table (WalkArgs::tab): launches of at least walk_table_min_B queries over u8 codes precompute the similarities to every
    // node of the levels >= table_level_min with one i8 MFMA GEMM; table_level_min = the lowest level such that the levels from it
    // to the top hold at most walk_table_max_cols nodes together (0 = no table).  Env COS_WALK_TABLE_COLS / COS_WALK_TABLE_MIN_B.
    u32 walk_table_max_cols = COS_WALK_TABLE_AUTO, walk_table_min_B = COS_WALK_TABLE_DEFAULT_MIN_B;
    size_t table_bytes_total = 0;    // tables held by this handle's workspaces (budget: get_workspace)
    bool adj_mag_valid = false;      // LevelHost::d_adj_mag follows the graph, the norms and the root (ensure_adj_mags)
    LinkState link;                  // cos_index_build -> cos_index_append (builder.hip); dropped with the graph
    bool level_table_valid = false;  // the arrays below follow the graph and walk_table_max_cols
    u32 table_level_min = 0, table_cols = 0;
    u64 table_built_for_key = 0;    // max_cols, or the automatic rule's per-level bound: a change rebuilds the operand
    u64 table_stride = 0;            // floats per query row (cols padded to 32)
    u32 table_col0[cosdev::MAX_LEVELS] = {};
    uint8_t *d_tcodes = nullptr;     // [table_cols][row_stride] code rows of the table's nodes, level by level from the top
    float *d_tmags = nullptr;        // [table_cols]
    u32 *d_tcsums = nullptr;         // [table_cols]
    // operands built for other keys of the SAME graph (ef_search changes re-select the table's levels while searches of the previous
    // ef may still be in flight: nothing a launch may be reading is freed before the graph itself is replaced)
    struct TableSet { u64 key; u32 level_min, cols; u64 stride; u32 col0[cosdev::MAX_LEVELS]; uint8_t *tcodes; float *tmags; u32 *tcsums; };
    std::vector<TableSet> table_sets;
    u32 walk_side_min_B = 4096; // launches of at least this many queries walk on the workspace's low-priority stream; 0 = never
    // launches of at most this many queries run the latency variant of the walk (kernels_walk_lat.hip) where it applies; 0 = never
    u32 lat_max_B = COS_LATENCY_MODE_DEFAULT_MAX_B;
    // ... and the smallest of them (one client batch) give every query four waves (kernels_walk_lat4.hip); 0 = never
    u32 lat4_max_B = COS_LATENCY_WAVES_DEFAULT_MAX_B;
    struct FlatWs *flat_ws = nullptr; // cos_flat_search_batch's buffers (kernels_flat.hip), created on first use under `mu`
};


struct FlatWs;
void cos_flat_ws_release(cos_index *ix);
void cos_coalesce_release(cos_index *ix);
cosdev::IndexDev cos_make_index_dev(const cos_index *ix);
cosdev::IndexDev cos_make_meta_dev(const cos_index *ix);
int32_t cos_set_device(const cos_index *ix);
int32_t cos_prepare_walk_plans(cos_index *ix); // order ranks + level-table operand of a freshly committed graph (engine.hip)
void cos_meta_free_levels(cos_index *ix); // device arrays + host lists of every level of the pseudo-root component
