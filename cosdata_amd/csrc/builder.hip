// builder.hip — vector_store::index_embeddings (vector_store.rs:714-1109) for a device-resident index.
//
// Batch-synchronous construction, entirely on the device: the ids of a batch run the reference's per-level walk
// (traverse_find_nearest with ef_construction, keep 64, visited pre-seeded with the new id) against the graph snapshot
// that precedes the batch — one wavefront per new vector, the same walk kernel the search path uses — then their edges are
// connected with the reference's exact edge semantics (create_node_edges / ProbNode::add_neighbor: replace-lowest slot,
// bidirectional accept-or-rollback, evictee drops its back edge, stale lowest cache) by the round-synchronous link kernels
// of kernels_link.hip.  The host only does bookkeeping: level draws, node numbering, the round loop.
// oracle/cosdata_oracle_hnsw.c:coso_index_build_rounds (ordered variant) is the CPU statement of the same schedule; the
// tests assert the two graphs are identical.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

#include "engine_internal.h"

using namespace cosdev;

namespace {

constexpr u32 NONE = 0xFFFFFFFFu;

inline uint64_t splitmix64(uint64_t &st) {
    uint64_t z = (st += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline float rand_f32(uint64_t &st) { return (float)(splitmix64(st) >> 40) * (1.0f / 16777216.0f); }

inline int32_t total_key(float v) {
    int32_t b;
    memcpy(&b, &v, 4);
    b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
    return b;
}
inline float metric_min(u32 metric) { return metric == COS_METRIC_COSINE ? -1.0f : -INFINITY; } // types.rs:435-446
inline float metric_max(u32 metric) { return metric == COS_METRIC_COSINE ? 2.0f : INFINITY; }   // types.rs:448-457
inline int32_t order_key(u32 metric, float v) {
    const int32_t k = total_key(v);
    return (metric == COS_METRIC_EUCLIDEAN || metric == COS_METRIC_HAMMING) ? ~k : k;
}

// pinned host staging
struct HostBuf {
    void *p = nullptr;
    ~HostBuf() { if (p) (void)hipHostFree(p); }
    hipError_t alloc(size_t bytes) { return hipHostMalloc(&p, bytes ? bytes : 1); }
    template <typename T> T *as() const { return (T *)p; }
};

} // namespace

namespace {

// generate_level_probs(4.0, L): 1 - 4^-n, n = L..0 (common.rs:421-429), and get_max_insert_level (common.rs:373-379) for one draw
std::vector<double> level_probs(u32 Ltop) {
    std::vector<double> pv(Ltop + 1);
    for (u32 k = 0; k <= Ltop; k++) {
        const int nn = (int)(Ltop - k);
        double r = 1.0, a = 4.0;
        for (int b = nn;;) { if (b & 1) r *= a; b /= 2; if (b == 0) break; a *= a; }
        pv[k] = 1.0 - 1.0 / r;
    }
    return pv;
}
inline uint8_t draw_level(uint64_t &rng, const std::vector<double> &pv, u32 Ltop) {
    const double x = (double)rand_f32(rng);
    u32 k = 0;
    while (k < Ltop && !(x >= pv[k])) k++;
    return (uint8_t)(Ltop - k);
}

// a failed build / append leaves the handle WITHOUT a graph (search returns NotReady) instead of a half-linked one
struct GraphGuard {
    cos_index *ix;
    bool armed = true;
    ~GraphGuard() {
        if (!armed) return;
        (void)hipStreamSynchronize(ix->own_stream);
        for (auto &l : ix->lv) {
            void *ptrs[] = {l.d_adj_vec, l.d_adj_node, l.d_node_vec, l.d_child, l.d_adj_mag};
            for (void *p : ptrs) if (p) (void)hipFree(p);
            l.d_adj_vec = l.d_adj_node = l.d_node_vec = l.d_child = nullptr;
            l.d_adj_mag = nullptr;
            l.n = 0;
            l.node_ids.clear();
            l.nbr_ids.clear();
            l.host_valid = false;
        }
        ix->link.release();
        ix->adj_mag_valid = false;
        ix->have_root = false;
    }
};

// LinkArgs over the handle's level arrays and link state
void fill_link_levels(cos_index *ix, LinkArgs &la, u32 &maxM) {
    maxM = 0;
    for (u32 l = 0; l <= ix->p.num_layers; l++) {
        LevelHost &H = ix->lv[l];
        LinkLevelDev &D = la.lv[l];
        D.adj_vec = H.d_adj_vec;
        D.adj_node = l == 0 ? H.d_adj_vec : H.d_adj_node;
        D.node_vec = l == 0 ? nullptr : H.d_node_vec;
        D.key = ix->link.key[l].as<int32_t>();
        D.low_idx = ix->link.low_idx[l].as<uint8_t>();
        D.low_key = ix->link.low_key[l].as<int32_t>();
        D.owner = ix->link.owner[l].as<u32>();
        D.M = H.M;
        maxM = std::max(maxM, H.M);
    }
}

// The batch loop of index_embeddings for the vectors [first, ix->n): batches of min(Bmax, max(1, inserted / 4)) ids run the reference
// walk against the snapshot that precedes the batch, then the round-synchronous link kernels connect them.  max_level is indexed by
// vector row - first; cursor[l] = the node index the next inserted vector takes on level l (nodes are in id order, the root last).
int32_t run_batches(cos_index *ix, u32 first, const uint8_t *max_level, std::vector<u32> &cursor, u32 Bmax) {
    const u32 n = ix->n, Ltop = ix->p.num_layers, L1 = Ltop + 1, metric = ix->p.metric;
    hipStream_t st = ix->own_stream;
    LinkArgs la;
    memset(&la, 0, sizeof(la));
    u32 maxM = 0;
    fill_link_levels(ix, la, maxM);
    if (maxM > 256) return cos_fail(COS_ERR_UNIMPLEMENTED, "more than 256 neighbour slots per node");

    // ---- batch workspace (device) + pinned staging -----------------------------------------------------------
    const u32 KEEP = (u32)KEEP_INDEX;
    DevBuf d_rows, d_out_ids, d_out_nodes, d_out_sims, d_out_counts, d_status, d_me, d_pend[2], d_cnt, d_evq;
    HostBuf h_rows, h_me, h_pend, h_status, h_cnt;
    const size_t pend_cap = (size_t)Bmax * L1;              // (level, batch position) entries
    const size_t evq_cap = pend_cap * 2 * KEEP;             // a node queues at most 2 evictions per candidate
    HIP_TRY(d_rows.alloc((size_t)Bmax * 4));
    HIP_TRY(d_out_ids.alloc((size_t)Bmax * L1 * KEEP * 4));
    HIP_TRY(d_out_nodes.alloc((size_t)Bmax * L1 * KEEP * 4));
    HIP_TRY(d_out_sims.alloc((size_t)Bmax * L1 * KEEP * 4));
    HIP_TRY(d_out_counts.alloc((size_t)Bmax * L1 * 4));
    HIP_TRY(d_status.alloc((size_t)Bmax * 4));
    HIP_TRY(d_me.alloc((size_t)Bmax * L1 * 4));
    HIP_TRY(d_pend[0].alloc(pend_cap * 4));
    HIP_TRY(d_pend[1].alloc(pend_cap * 4));
    HIP_TRY(d_cnt.alloc(4 * 4));                            // [0], [1]: pending counters (ping-pong); [2]: eviction queue length
    HIP_TRY(d_evq.alloc(evq_cap * 3 * 4));
    HIP_TRY(h_rows.alloc((size_t)Bmax * 4));
    HIP_TRY(h_me.alloc((size_t)Bmax * L1 * 4));
    HIP_TRY(h_pend.alloc(pend_cap * 4));
    HIP_TRY(h_status.alloc((size_t)Bmax * 4));
    HIP_TRY(h_cnt.alloc(4 * 4));
    VisTab vtab; // EXACT build mode: visited filters of the batch walks
    struct VisFree { VisTab &v; ~VisFree() { if (v.bits) (void)hipFree(v.bits); if (v.log) (void)hipFree(v.log); } } visfree{vtab};

    la.L1 = L1;
    la.z_nodes = d_out_nodes.as<u32>();
    la.z_sims = d_out_sims.as<float>();
    la.z_counts = d_out_counts.as<u32>();
    la.me = d_me.as<u32>();
    la.metric = metric;
    la.kmin = order_key(metric, metric_min(metric));
    la.kmax = order_key(metric, metric_max(metric));

    IndexDev dev = cos_make_index_dev(ix);
    u32 inserted = first, round = 0;
    u64 n_rounds = 0, n_batches = 0;
    const bool prof = cosdev::tune_or(cosdev::TUNE_BUILD_PROFILE, 0) != 0;
    double t_walk = 0, t_link = 0;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    u32 *cnt = d_cnt.as<u32>();
    while (inserted < n) {
        const double t0 = now();
        n_batches++;
        const u32 bs = std::min({Bmax, std::max(1u, inserted / 4u), n - inserted});
        // batch bookkeeping: rows to walk, node index of every (batch item, level), initial pending list (all levels together)
        u32 np = 0;
        for (u32 b = 0; b < bs; b++) {
            const u32 id = inserted + b;
            h_rows.as<u32>()[b] = id;
            for (u32 l = 0; l <= Ltop; l++) {
                u32 m = NONE;
                if (max_level[id - first] >= l) { m = cursor[l]++; h_pend.as<u32>()[np++] = (l << 16) | b; }
                h_me.as<u32>()[(size_t)b * L1 + l] = m;
            }
        }
        HIP_TRY(hipMemcpyAsync(d_rows.p, h_rows.p, (size_t)bs * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_me.p, h_me.p, (size_t)bs * L1 * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_pend[0].p, h_pend.p, (size_t)np * 4, hipMemcpyHostToDevice, st));
        h_cnt.as<u32>()[0] = np; h_cnt.as<u32>()[1] = 0; h_cnt.as<u32>()[2] = 0; h_cnt.as<u32>()[3] = 0;
        HIP_TRY(hipMemcpyAsync(cnt, h_cnt.p, 16, hipMemcpyHostToDevice, st));
        // 1. the reference walk of every new vector against the snapshot that precedes the batch (one wave each)
        WalkArgs wa;
        memset(&wa, 0, sizeof(wa));
        wa.qcodes = ix->d_codes;
        wa.qmags = ix->d_mags;
        wa.q_rows = d_rows.as<u32>();
        wa.self_ids = d_rows.as<u32>(); // the new node's id is pre-inserted in the visited filter (vector_store.rs:807)
        if (dev.visited_mode == COS_VISITED_EXACT) {
            const int32_t vrc = vis_tab_prepare(vtab, ix, Bmax, ix->p.ef_construction, st, wa);
            if (vrc) return vrc;
        }
        wa.B = bs;
        wa.ef = ix->p.ef_construction;
        wa.keep = KEEP;
        wa.out_ids = d_out_ids.as<u32>();
        wa.out_sims = d_out_sims.as<float>();
        wa.out_nodes = d_out_nodes.as<u32>();
        wa.out_counts = d_out_counts.as<u32>();
        wa.out_status = d_status.as<int32_t>();
        HIP_TRY(launch_walk(ix->eng, dev, wa, ix->lat_max_B, ix->lat4_max_B, st));
        HIP_TRY(hipMemcpyAsync(h_status.p, d_status.p, (size_t)bs * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st)); // a failed walk leaves its lower levels unwritten: never link from it
        for (u32 b = 0; b < bs; b++)
            if (h_status.as<int32_t>()[b] != COS_OK)
                return cos_fail(h_status.as<int32_t>()[b], "vector %u cannot be indexed (zero norm -> DistanceError::CalculationError)", inserted + b);
        t_walk += now() - t0;
        const double t1 = now();
        // 2. link rounds (kernels_link.hip).  Rounds are enqueued a few at a time; each reads the true pending count on the
        // device, so over-launching costs nothing but empty blocks.
        u32 pending_ub = np, cur = 0;
        while (pending_ub > 0) {
            const u32 chunk = pending_ub > 64 ? 2 : 4;
            for (u32 r = 0; r < chunk; r++) {
                if (++round > LINK_MAX_ROUND) { // claim tags would wrap: start a new tag epoch
                    for (u32 l = 0; l <= Ltop; l++) HIP_TRY(hipMemsetAsync(ix->link.owner[l].p, 0, (size_t)ix->lv[l].n * 4, st));
                    round = 1;
                }
                HIP_TRY(launch_link_round(la, maxM, d_pend[cur].as<u32>(), cnt + cur, d_pend[cur ^ 1].as<u32>(), cnt + (cur ^ 1), d_evq.as<u32>(), cnt + 2,
                                          pending_ub, round, st));
                cur ^= 1;
                n_rounds++;
            }
            HIP_TRY(hipMemcpyAsync(h_cnt.p, cnt, 16, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            pending_ub = h_cnt.as<u32>()[cur]; // the list the next round would read
        }
        t_link += now() - t1;
        inserted += bs;
    }
    HIP_TRY(hipStreamSynchronize(st));
    if (prof)
        fprintf(stderr, "[cos_index_build] vectors [%u, %u) batches=%llu link rounds=%llu walk %.2fs link %.2fs\n", first, n, (unsigned long long)n_batches,
                (unsigned long long)n_rounds, t_walk, t_link);
    return COS_OK;
}

} // namespace

extern "C" int32_t cos_index_build(cos_index *ix, uint32_t batch_size) {
    if (!ix) return cos_fail(COS_ERR_INVALID, "null index");
    if (!ix->have_vectors) return cos_fail(COS_ERR_NOT_READY, "upload vectors before building");
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    const u32 n = ix->n, Ltop = ix->p.num_layers, metric = ix->p.metric;
    const u32 Bmax = std::min<u32>(batch_size ? batch_size : 4096u, LINK_MAX_BATCH);
    hipStream_t st = ix->own_stream;
    GraphGuard guard{ix};

    ix->order_rank_valid = false; // the order key's table follows the graph (engine.hip, ensure_order_rank)
    ix->level_table_valid = false; // and so does the level table's operand (ensure_level_table)
    ix->adj_mag_valid = false;     // the builder's walks gather mags[]; the adjacency-side norms are written when the graph is committed
    ix->link.release();
    // ---- root + level draws (same RNG stream as the oracle builders) ----------------------------
    uint64_t rng = ix->p.seed ? ix->p.seed : 0x1234567ull;
    std::vector<float> root(ix->p.dim);
    for (u32 i = 0; i < ix->p.dim; i++) root[i] = ix->p.range_lo + rand_f32(rng) * (ix->p.range_hi - ix->p.range_lo); // vector_store.rs:30-36
    rc = cos_index_set_root(ix, root.data());
    if (rc) return rc;
    const std::vector<double> pv = level_probs(Ltop);
    std::vector<uint8_t> max_level(n);
    for (u32 id = 0; id < n; id++) max_level[id] = draw_level(rng, pv, Ltop);

    // ---- level skeletons: nodes of a level in ascending id, root last; every slot empty ------------------------
    for (u32 l = 0; l <= Ltop; l++) {
        LevelHost &H = ix->lv[l];
        void *old[] = {H.d_adj_vec, H.d_adj_node, H.d_node_vec, H.d_child, H.d_adj_mag};
        for (void *p : old) if (p) (void)hipFree(p);
        H.d_adj_vec = H.d_adj_node = H.d_node_vec = H.d_child = nullptr;
        H.d_adj_mag = nullptr;
        H.host_valid = false;
        H.nbr_ids.clear();
        H.node_ids.clear();
        for (u32 id = 0; id < n; id++)
            if (max_level[id] >= l) H.node_ids.push_back(id * ix->id_stride); // internal id of vector row `id`
        H.node_ids.push_back(COS_ROOT_ID);
        const u32 nl = (u32)H.node_ids.size(), M = H.M;
        if (M > 256) return cos_fail(COS_ERR_UNIMPLEMENTED, "more than 256 neighbour slots per node");
        HIP_TRY(hipMalloc((void **)&H.d_adj_vec, (size_t)nl * M * 4));
        HIP_TRY(hipMemsetAsync(H.d_adj_vec, 0xFF, (size_t)nl * M * 4, st));
        if (l > 0) {
            std::vector<u32> node_vec(nl), child(nl);
            const std::vector<u32> &D = ix->lv[l - 1].node_ids;
            for (u32 i = 0; i < nl; i++) {
                node_vec[i] = H.node_ids[i] == COS_ROOT_ID ? n : H.node_ids[i] / ix->id_stride;
                child[i] = (u32)(std::lower_bound(D.begin(), D.end(), H.node_ids[i]) - D.begin()); // same id one level down (vector_store.rs:897-903)
            }
            HIP_TRY(hipMalloc((void **)&H.d_adj_node, (size_t)nl * M * 4));
            HIP_TRY(hipMemsetAsync(H.d_adj_node, 0xFF, (size_t)nl * M * 4, st));
            HIP_TRY(hipMalloc((void **)&H.d_node_vec, (size_t)nl * 4));
            HIP_TRY(hipMemcpy(H.d_node_vec, node_vec.data(), (size_t)nl * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMalloc((void **)&H.d_child, (size_t)nl * 4));
            HIP_TRY(hipMemcpy(H.d_child, child.data(), (size_t)nl * 4, hipMemcpyHostToDevice));
        }
        H.n = nl;
        // link state: slot keys (all empty), cached lowest = (0, MetricResult::min) (prob_node.rs:140), claim tags
        HIP_TRY(ix->link.key[l].alloc((size_t)nl * M * 4));
        HIP_TRY(ix->link.low_idx[l].alloc(nl));
        HIP_TRY(ix->link.low_key[l].alloc((size_t)nl * 4));
        HIP_TRY(ix->link.owner[l].alloc((size_t)nl * 4));
        HIP_TRY(launch_fill_i32(ix->link.key[l].as<int32_t>(), (u64)nl * M, INT32_MIN, st));
        HIP_TRY(hipMemsetAsync(ix->link.low_idx[l].p, 0, nl, st));
        HIP_TRY(launch_fill_i32(ix->link.low_key[l].as<int32_t>(), nl, order_key(metric, metric_min(metric)), st));
        HIP_TRY(hipMemsetAsync(ix->link.owner[l].p, 0, (size_t)nl * 4, st));
    }
    HIP_TRY(hipStreamSynchronize(st));

    std::vector<u32> cursor(Ltop + 1, 0); // first node of each level not yet inserted (nodes are in id order)
    rc = run_batches(ix, 0, max_level.data(), cursor, Bmax);
    if (rc) return rc;
    guard.armed = false;
    ix->link.rng = rng;
    ix->link.n_built = n;
    ix->link.valid = ix->id_stride == 1u; // (collections with a metadata schema reserve ids per embedding: not appendable here)
    if (int32_t rc2 = cos_prepare_walk_plans(ix)) return rc2; // order ranks + level-table operand of the new graph, outside any search
    return COS_OK; // the id-format host copy is made on demand (cos_index_download_graph_level)
}

// The link state of an UPLOADED graph (cos_index_upload_graph_level, cos_index_load_reference_dir), as the reference has it after a
// reload: it persists every slot's similarity (serializer/hnsw/neighbors.rs:22-61) and recomputes a node's cached lowest slot when it
// deserializes the node (ProbNode::new_with_neighbors_and_versions, prob_node.rs:145-181).  Here the similarities are recomputed with
// the distance operator's kernel on the resident codes (the distance is symmetric in its bits), the caches follow the reload rule,
// and later appends draw their levels from the seed's stream advanced past the resident vectors (a graph built by the same-seed
// builder and uploaded continues with the draws a native build would make).  oracle: coso_index_restore_link_state.
extern "C" int32_t cos_index_restore_link_state(cos_index *ix) {
    if (!ix) return cos_fail(COS_ERR_INVALID, "null index");
    if (!ix->have_vectors || !ix->have_root) return cos_fail(COS_ERR_NOT_READY, "restore needs vectors, root and every graph level");
    for (auto &l : ix->lv) if (l.n == 0) return cos_fail(COS_ERR_NOT_READY, "restore needs vectors, root and every graph level");
    if (ix->id_stride != 1u || ix->meta.mdim != 0u) return cos_fail(COS_ERR_UNIMPLEMENTED, "link state of a collection with a metadata schema");
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    const u32 Ltop = ix->p.num_layers, metric = ix->p.metric;
    const int32_t kmin = order_key(metric, metric_min(metric)), kmax = order_key(metric, metric_max(metric));
    hipStream_t st = ix->own_stream;
    ix->link.release();
    struct Fail { cos_index *ix; bool armed = true; ~Fail() { if (armed) ix->link.release(); } } fail_guard{ix};
    constexpr u64 CHUNK_PAIRS = 1ull << 24; // 64 MB per pair array
    DevBuf px, py, sims, dst, d_fail;
    HIP_TRY(px.alloc(CHUNK_PAIRS * 4));
    HIP_TRY(py.alloc(CHUNK_PAIRS * 4));
    HIP_TRY(sims.alloc(CHUNK_PAIRS * 4));
    HIP_TRY(dst.alloc(CHUNK_PAIRS * 4));
    HIP_TRY(d_fail.alloc(4));
    HIP_TRY(hipMemsetAsync(d_fail.p, 0, 4, st));
    for (u32 l = 0; l <= Ltop; l++) {
        LevelHost &H = ix->lv[l];
        const u32 nl = H.n, M = H.M;
        HIP_TRY(ix->link.key[l].alloc((size_t)nl * M * 4));
        HIP_TRY(ix->link.low_idx[l].alloc(nl));
        HIP_TRY(ix->link.low_key[l].alloc((size_t)nl * 4));
        HIP_TRY(ix->link.owner[l].alloc((size_t)nl * 4));
        HIP_TRY(hipMemsetAsync(ix->link.owner[l].p, 0, (size_t)nl * 4, st));
        const u32 step = (u32)std::max<u64>(1, CHUNK_PAIRS / M);
        for (u32 n0 = 0; n0 < nl; n0 += step) {
            const u32 cn = std::min(step, nl - n0);
            HIP_TRY(launch_edge_pairs(H.d_adj_vec, l == 0 ? nullptr : H.d_node_vec, n0, cn, M, px.as<u32>(), py.as<u32>(), st));
            HIP_TRY(cosdev::launch_index_pair_distances(ix->eng, ix->d_codes, ix->d_mags, ix->row_stride, ix->nchunks, ix->p.dim, metric, px.as<u32>(), py.as<u32>(),
                                                       (u32)((u64)cn * M), sims.as<float>(), dst.as<int32_t>(), st));
            HIP_TRY(launch_edge_keys(H.d_adj_vec, sims.as<float>(), dst.as<int32_t>(), n0, cn, M, metric, ix->link.key[l].as<int32_t>(), d_fail.as<int32_t>(), st));
        }
        HIP_TRY(launch_low_cache(H.d_adj_vec, ix->link.key[l].as<int32_t>(), nl, M, kmin, kmax, ix->link.low_idx[l].as<uint8_t>(), ix->link.low_key[l].as<int32_t>(), st));
    }
    int32_t failed = 0;
    HIP_TRY(hipMemcpyAsync(&failed, d_fail.p, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (failed) return cos_fail(failed, "an edge of the uploaded graph has no similarity (zero norm -> DistanceError::CalculationError)");
    uint64_t rng = ix->p.seed ? ix->p.seed : 0x1234567ull;
    for (u64 i = 0; i < (u64)ix->p.dim + ix->n; i++) (void)rand_f32(rng); // the root's components, then one level draw per resident vector
    ix->link.rng = rng;
    ix->link.n_built = ix->n;
    ix->link.valid = true;
    fail_guard.armed = false;
    return COS_OK;
}

extern "C" int32_t cos_index_release_link_state(cos_index *ix) {
    if (!ix) return cos_fail(COS_ERR_INVALID, "null index");
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ix->own_stream));
    ix->link.release();
    return COS_OK;
}

// vector_store::index_embeddings called AGAIN on a live index (vector_store.rs:714-780; index_embedding :782-975 per vector): m more
// vectors take the internal ids [n, n + m) (sequential: collection.rs:451-468) and are inserted into the resident graph by the
// schedule of cos_index_build CONTINUED at inserted = n — level draws from the same RNG stream, batches of min(batch, max(1,
// inserted / 4)) walking the snapshot that precedes them, the round-synchronous link kernels on the link state cos_index_build left
// (slot similarities and lowest caches of EVERY node: a new vector's back edges evict from old nodes' rows).  oracle/
// cosdata_oracle_hnsw.c: coso_index_append_vectors + coso_index_build_rounds_continue is the CPU statement; tests/test_gpu_append.py
// asserts identical graphs.  Exclusive like every graph change: no search in flight.
extern "C" int32_t cos_index_append(cos_index *ix, const float *raw, uint32_t m, uint32_t flags, uint32_t batch_size) {
    if (!ix || !raw || m == 0) return cos_fail(COS_ERR_INVALID, "null/empty vectors");
    if (!ix->have_vectors || !ix->have_root) return cos_fail(COS_ERR_NOT_READY, "append needs a built index");
    for (auto &l : ix->lv) if (l.n == 0) return cos_fail(COS_ERR_NOT_READY, "append needs a built index");
    if (!ix->link.valid || ix->link.n_built != ix->n)
        return cos_fail(COS_ERR_NOT_READY, "no link state to continue from: the graph was uploaded (not built by cos_index_build on this handle), its root "
                                           "was replaced, or cos_index_release_link_state freed it");
    if (ix->id_stride != 1u || ix->meta.mdim != 0u) return cos_fail(COS_ERR_UNIMPLEMENTED, "append on a collection with a metadata schema");
    if ((u64)ix->n + m >= 0xFFFFFFF0ull) return cos_fail(COS_ERR_INVALID, "too many vectors");
    const bool borrow = (flags & COS_UPLOAD_BORROW_DEVICE) != 0;
    if (borrow != ix->raw_borrowed)
        return cos_fail(COS_ERR_INVALID, borrow ? "the index owns its raw rows: append host rows" : "the index borrows its raw rows: append with COS_UPLOAD_BORROW_DEVICE and the whole grown table");
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    const u32 n0 = ix->n, n1 = n0 + m, Ltop = ix->p.num_layers, metric = ix->p.metric;
    const u32 Bmax = std::min<u32>(batch_size ? batch_size : 4096u, LINK_MAX_BATCH);
    const u64 dim = ix->p.dim, rs = ix->row_stride;
    hipStream_t st = ix->own_stream;
    HIP_TRY(hipDeviceSynchronize()); // exclusive: whatever read the old arrays is done
    GraphGuard guard{ix};
    ix->order_rank_valid = false;
    ix->level_table_valid = false;
    ix->adj_mag_valid = false;
    ix->link.valid = false; // until the append is complete
    cos_flat_ws_release(ix); // cached sums of the stored codes, scan buffers sized for the old corpus

    // ---- 1. the vector tables grow: rows [0, n0) stay, rows [n0, n1) are the new vectors, the root and the pseudo nodes' vector move behind them
    {
        uint8_t *codes = nullptr;
        float *mags = nullptr, *raw_mags = nullptr, *new_raw = nullptr;
        struct Tmp { void **p[4]; ~Tmp() { for (auto q : p) if (q && *q) (void)hipFree(*q); } } tmp{{(void **)&codes, (void **)&mags, (void **)&raw_mags, (void **)&new_raw}};
        HIP_TRY(hipMalloc((void **)&codes, ((size_t)n1 + 2) * rs));
        HIP_TRY(hipMalloc((void **)&mags, ((size_t)n1 + 2) * 4));
        HIP_TRY(hipMalloc((void **)&raw_mags, ((size_t)n1 + 2) * 4));
        HIP_TRY(hipMemcpyAsync(codes, ix->d_codes, (size_t)n0 * rs, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(codes + (size_t)n1 * rs, ix->d_codes + (size_t)n0 * rs, 2 * rs, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(mags, ix->d_mags, (size_t)n0 * 4, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(mags + n1, ix->d_mags + n0, 8, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(raw_mags, ix->d_raw_mags, (size_t)n0 * 4, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemsetAsync(raw_mags + n1, 0, 8, st));
        const float *new_rows;
        if (borrow) {
            new_rows = raw + (size_t)n0 * dim; // raw = the caller's whole grown table (device)
        } else {
            HIP_TRY(hipMalloc((void **)&new_raw, (size_t)n1 * dim * 4));
            HIP_TRY(hipMemcpyAsync(new_raw, ix->d_raw, (size_t)n0 * dim * 4, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(new_raw + (size_t)n0 * dim, raw, (size_t)m * dim * 4, hipMemcpyHostToDevice, st));
            new_rows = new_raw + (size_t)n0 * dim;
        }
        HIP_TRY(launch_quantize_rows(ix->eng, new_rows, dim, m, ix->p.dim, ix->p.range_lo, ix->p.range_hi, codes + (size_t)n0 * rs, rs, mags + n0, raw_mags + n0, st));
        HIP_TRY(hipStreamSynchronize(st));
        std::swap(codes, ix->d_codes);
        std::swap(mags, ix->d_mags);
        std::swap(raw_mags, ix->d_raw_mags);
        if (borrow) ix->d_raw = const_cast<float *>(raw);
        else std::swap(new_raw, ix->d_raw);
        ix->n = n1; // (tmp frees the old arrays)
    }

    // ---- 2. level draws of the new ids: the draws a full build would have given them
    uint64_t rng = ix->link.rng;
    const std::vector<double> pv = level_probs(Ltop);
    std::vector<uint8_t> max_level(m);
    for (u32 i = 0; i < m; i++) max_level[i] = draw_level(rng, pv, Ltop);

    // ---- 3. every level grows: the new nodes (ascending id) go between the old nodes and the root, every slot empty; references to
    //         the root (vector row n0 -> n1, node index nl - 1 -> nl' - 1) are rewritten
    std::vector<u32> cursor(Ltop + 1, 0);
    std::vector<u32> add(Ltop + 1, 0);
    for (u32 l = 0; l <= Ltop; l++)
        for (u32 i = 0; i < m; i++) add[l] += max_level[i] >= l ? 1u : 0u;
    for (u32 l = 0; l <= Ltop; l++) {
        LevelHost &H = ix->lv[l];
        const u32 nl0 = H.n, nl1 = nl0 + add[l], M = H.M;
        cursor[l] = nl0 - 1; // the first new node takes the root's old index
        H.node_ids.pop_back();
        for (u32 i = 0; i < m; i++) if (max_level[i] >= l) H.node_ids.push_back(n0 + i);
        H.node_ids.push_back(COS_ROOT_ID);
        H.host_valid = false;
        H.nbr_ids.clear();
        if (H.d_adj_mag) { (void)hipFree(H.d_adj_mag); H.d_adj_mag = nullptr; }
        u32 *adj_vec = nullptr, *adj_node = nullptr, *node_vec = nullptr, *child = nullptr;
        DevBuf key, low_idx, low_key;
        struct Tmp { void **p[4]; ~Tmp() { for (auto q : p) if (q && *q) (void)hipFree(*q); } } tmp{{(void **)&adj_vec, (void **)&adj_node, (void **)&node_vec, (void **)&child}};
        HIP_TRY(hipMalloc((void **)&adj_vec, (size_t)nl1 * M * 4));
        HIP_TRY(launch_grow_rows(H.d_adj_vec, adj_vec, nl0, nl1, M, n0, n1, ROW_EMPTY, st)); // vector rows: the root is row n
        if (l > 0) {
            HIP_TRY(hipMalloc((void **)&adj_node, (size_t)nl1 * M * 4));
            HIP_TRY(launch_grow_rows(H.d_adj_node, adj_node, nl0, nl1, M, nl0 - 1, nl1 - 1, ROW_EMPTY, st)); // node indices: the root is the last node
            // node -> vector row and node -> node one level down: old nodes keep theirs, the root's follow it, the new nodes' are known
            std::vector<u32> nv(add[l] + 1), ch(add[l] + 1);
            const u32 below0 = ix->lv[l - 1].n - add[l - 1] - 1; // first new node index one level down (level l - 1 has grown already)
            u32 k = 0, kb = 0;
            for (u32 i = 0; i < m; i++) {
                if (max_level[i] >= l) { nv[k] = n0 + i; ch[k] = below0 + kb; k++; }
                if (max_level[i] >= l - 1) kb++;
            }
            nv[k] = n1;                       // the root's vector row
            ch[k] = ix->lv[l - 1].n - 1;      // ... and its node one level down
            HIP_TRY(hipMalloc((void **)&node_vec, (size_t)nl1 * 4));
            HIP_TRY(hipMalloc((void **)&child, (size_t)nl1 * 4));
            HIP_TRY(hipMemcpyAsync(node_vec, H.d_node_vec, (size_t)(nl0 - 1) * 4, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(child, H.d_child, (size_t)(nl0 - 1) * 4, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpy(node_vec + (nl0 - 1), nv.data(), nv.size() * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(child + (nl0 - 1), ch.data(), ch.size() * 4, hipMemcpyHostToDevice));
        }
        HIP_TRY(key.alloc((size_t)nl1 * M * 4));
        HIP_TRY(low_idx.alloc(nl1));
        HIP_TRY(low_key.alloc((size_t)nl1 * 4));
        HIP_TRY(launch_grow_rows(ix->link.key[l].as<u32>(), key.as<u32>(), nl0, nl1, M, 0xFFFFFFFFu, 0xFFFFFFFFu, (u32)INT32_MIN, st));
        HIP_TRY(launch_grow_rows(ix->link.low_key[l].as<u32>(), low_key.as<u32>(), nl0, nl1, 1, 0xFFFFFFFFu, 0xFFFFFFFFu, (u32)order_key(metric, metric_min(metric)), st));
        HIP_TRY(launch_grow_bytes(ix->link.low_idx[l].as<uint8_t>(), low_idx.as<uint8_t>(), nl0, nl1, st));
        HIP_TRY(ix->link.owner[l].alloc((size_t)nl1 * 4));
        HIP_TRY(hipMemsetAsync(ix->link.owner[l].p, 0, (size_t)nl1 * 4, st)); // a new tag epoch (run_batches starts at round 1)
        HIP_TRY(hipStreamSynchronize(st));
        std::swap(adj_vec, H.d_adj_vec);
        std::swap(adj_node, H.d_adj_node);
        std::swap(node_vec, H.d_node_vec);
        std::swap(child, H.d_child);
        std::swap(key.p, ix->link.key[l].p);
        std::swap(low_idx.p, ix->link.low_idx[l].p);
        std::swap(low_key.p, ix->link.low_key[l].p);
        H.n = nl1; // (tmp / the DevBufs free the old arrays)
    }
    // child links INTO a level that grew: level l's root entry was written above; the old nodes' children are old nodes (unchanged)

    // ---- 4. the new vectors, batch by batch
    rc = run_batches(ix, n0, max_level.data(), cursor, Bmax);
    if (rc) return rc;
    guard.armed = false;
    ix->link.rng = rng;
    ix->link.n_built = n1;
    ix->link.valid = true;
    if (int32_t rc2 = cos_prepare_walk_plans(ix)) return rc2;
    return COS_OK;
}

// delete_embedding (vector_store.rs:1206-1400) for the internal ids of `ids`, one after the other in the given order (the reference
// deletes inside a transaction one embedding at a time; a later delete's walk sees the graph the earlier ones left).  Per id: ONE walk
// launch for the vector's own code — every level, ef = 512, keep 100, nothing pre-inserted in the filters (:1232-1248) — then, per
// level where the walk's list holds the node, unlink_kernel: every neighbour drops its back edge, the node's slots are emptied.  A level
// on which a neighbour would be left without any neighbour (the reference links it again between the removals, :1305-1357) is run on
// the host in the reference's order over a small cache of the rows involved (rare: a node whose ONLY neighbour is the deleted one).
// The vector's rows stay in the tables; the node is unreachable (edges are symmetric by construction) and its id is never returned by a
// walk again.  Needs the link state (slot keys / lowest caches are kept consistent: a later append continues from them).
// oracle/cosdata_oracle_hnsw.c: coso_index_delete is the CPU statement; tests/test_gpu_append.py asserts identical graphs.
namespace {

struct RowCache { // host copies of the level rows a slow-path delete touches; written back at the end
    struct Row { std::vector<u32> adj; std::vector<int32_t> key; uint8_t low_idx; int32_t low_key; bool dirty = false; };
    cos_index *ix; u32 level; u32 M;
    std::map<u32, Row> rows;
    hipError_t err = hipSuccess;
    Row &get(u32 node) {
        auto it = rows.find(node);
        if (it != rows.end()) return it->second;
        Row &r = rows[node];
        LevelHost &H = ix->lv[level];
        r.adj.resize(M); r.key.resize(M);
        auto cp = [&](void *dst, const void *src, size_t n) { if (err == hipSuccess) err = hipMemcpy(dst, src, n, hipMemcpyDeviceToHost); };
        cp(r.adj.data(), (level == 0 ? H.d_adj_vec : H.d_adj_node) + (size_t)node * M, (size_t)M * 4);
        cp(r.key.data(), ix->link.key[level].as<int32_t>() + (size_t)node * M, (size_t)M * 4);
        cp(&r.low_idx, ix->link.low_idx[level].as<uint8_t>() + node, 1);
        cp(&r.low_key, ix->link.low_key[level].as<int32_t>() + node, 4);
        return r;
    }
    hipError_t flush(const std::vector<u32> &node_vec /* node -> vector row (level 0: identity) */) {
        LevelHost &H = ix->lv[level];
        for (auto &kv : rows) {
            if (!kv.second.dirty || err != hipSuccess) continue;
            const u32 node = kv.first;
            Row &r = kv.second;
            auto cp = [&](void *dst, const void *src, size_t n) { if (err == hipSuccess) err = hipMemcpy(dst, src, n, hipMemcpyHostToDevice); };
            if (level == 0) cp(H.d_adj_vec + (size_t)node * M, r.adj.data(), (size_t)M * 4);
            else {
                std::vector<u32> av(M);
                for (u32 j = 0; j < M; j++) av[j] = r.adj[j] == NONE ? NONE : node_vec[r.adj[j]];
                cp(H.d_adj_node + (size_t)node * M, r.adj.data(), (size_t)M * 4);
                cp(H.d_adj_vec + (size_t)node * M, av.data(), (size_t)M * 4);
            }
            cp(ix->link.key[level].as<int32_t>() + (size_t)node * M, r.key.data(), (size_t)M * 4);
            cp(ix->link.low_idx[level].as<uint8_t>() + node, &r.low_idx, 1);
            cp(ix->link.low_key[level].as<int32_t>() + node, &r.low_key, 4);
        }
        return err;
    }
};

// ProbNode::add_neighbor (prob_node.rs:210-283) on cached rows; returns the slot or -1
int add_neighbor_host(RowCache &rc, u32 self, u32 nbr, int32_t k, int32_t kmin, int32_t kmax) {
    RowCache::Row &r = rc.get(self);
    const u32 M = rc.M, li = r.low_idx;
    if (k <= r.low_key) return -1;
    const bool ok = r.adj[li] == NONE || k > r.key[li];
    u32 old = NONE;
    if (ok) { old = r.adj[li]; r.adj[li] = nbr; r.key[li] = k; }
    u32 nl = 0;
    int32_t nk = kmax;
    for (u32 j = 0; j < M; j++) {
        if (r.adj[j] == NONE) { nk = kmin; nl = j; break; }
        if (r.key[j] < nk) { nk = r.key[j]; nl = j; }
    }
    r.low_idx = (uint8_t)nl;
    r.low_key = nk;
    r.dirty = true;
    if (!ok) return -1;
    if (old != NONE) { // the evictee drops its back edge; its cache is NOT refreshed
        RowCache::Row &o = rc.get(old);
        for (u32 j = 0; j < M; j++) if (o.adj[j] == self) { o.adj[j] = NONE; o.key[j] = INT32_MIN; o.dirty = true; break; }
    }
    return (int)li;
}

} // namespace

extern "C" int32_t cos_index_delete(cos_index *ix, const uint32_t *ids, uint32_t m) {
    if (!ix || (!ids && m)) return cos_fail(COS_ERR_INVALID, "null argument");
    if (!ix->have_vectors || !ix->have_root) return cos_fail(COS_ERR_NOT_READY, "delete needs a built index");
    for (auto &l : ix->lv) if (l.n == 0) return cos_fail(COS_ERR_NOT_READY, "delete needs a built index");
    if (!ix->link.valid) return cos_fail(COS_ERR_NOT_READY, "no link state (the graph was uploaded, its root replaced, or cos_index_release_link_state freed it)");
    if (ix->id_stride != 1u || ix->meta.mdim != 0u) return cos_fail(COS_ERR_UNIMPLEMENTED, "delete on a collection with a metadata schema");
    for (u32 i = 0; i < m; i++) if (ids[i] >= ix->n) return cos_fail(COS_ERR_INVALID, "id %u is not a resident vector", ids[i]);
    if (m == 0) return COS_OK;
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize()); // exclusive: no search in flight
    const u32 Ltop = ix->p.num_layers, L1 = Ltop + 1, metric = ix->p.metric, KEEP = (u32)KEEP_SEARCH, EF = 512;
    hipStream_t st = ix->own_stream;
    LinkArgs la;
    memset(&la, 0, sizeof(la));
    u32 maxM = 0;
    fill_link_levels(ix, la, maxM);
    la.L1 = L1;
    la.metric = metric;
    const int32_t kmin = order_key(metric, metric_min(metric)), kmax = order_key(metric, metric_max(metric));
    DevBuf d_row, d_out_ids, d_out_nodes, d_out_sims, d_out_counts, d_status, d_dn, d_ust, d_px, d_py, d_ds, d_dst;
    HIP_TRY(d_row.alloc(4));
    HIP_TRY(d_out_ids.alloc((size_t)L1 * KEEP * 4));
    HIP_TRY(d_out_nodes.alloc((size_t)L1 * KEEP * 4));
    HIP_TRY(d_out_sims.alloc((size_t)L1 * KEEP * 4));
    HIP_TRY(d_out_counts.alloc((size_t)L1 * 4));
    HIP_TRY(d_status.alloc(4));
    HIP_TRY(d_dn.alloc((size_t)L1 * 4));
    HIP_TRY(d_ust.alloc((size_t)L1 * 4));
    HIP_TRY(d_px.alloc((size_t)KEEP * 4));
    HIP_TRY(d_py.alloc((size_t)KEEP * 4));
    HIP_TRY(d_ds.alloc((size_t)KEEP * 4));
    HIP_TRY(d_dst.alloc((size_t)KEEP * 4));
    VisTab vtab;
    struct VisFree { VisTab &v; ~VisFree() { if (v.bits) (void)hipFree(v.bits); if (v.log) (void)hipFree(v.log); } } visfree{vtab};
    std::vector<u32> h_ids((size_t)L1 * KEEP), h_nodes((size_t)L1 * KEEP), h_counts(L1), h_dn(L1), h_ust(L1);
    std::vector<std::vector<u32>> node_vec(L1); // slow path only: node -> vector row, fetched once per level
    // (the locality order stays a permutation of the level's nodes and the level table's operand holds code rows: both still valid)
    ix->adj_mag_valid = false;
    for (auto &l : ix->lv) { l.host_valid = false; l.nbr_ids.clear(); }

    for (u32 t = 0; t < m; t++) {
        const u32 id = ids[t];
        IndexDev dev = cos_make_index_dev(ix);
        for (u32 l = 0; l <= Ltop; l++) dev.lv[l].adj_mag = nullptr; // (removed slots would keep a stale norm: harmless, but not worth a refill per id)
        WalkArgs wa;
        memset(&wa, 0, sizeof(wa));
        HIP_TRY(hipMemcpyAsync(d_row.p, &id, 4, hipMemcpyHostToDevice, st));
        wa.qcodes = ix->d_codes;
        wa.qmags = ix->d_mags;
        wa.q_rows = d_row.as<u32>();
        wa.no_self_seed = 1u;
        if (dev.visited_mode == COS_VISITED_EXACT) {
            const int32_t vrc = vis_tab_prepare(vtab, ix, 1, EF, st, wa);
            if (vrc) return vrc;
        }
        wa.B = 1;
        wa.ef = EF;
        wa.keep = KEEP;
        wa.out_ids = d_out_ids.as<u32>();
        wa.out_sims = d_out_sims.as<float>();
        wa.out_nodes = d_out_nodes.as<u32>();
        wa.out_counts = d_out_counts.as<u32>();
        wa.out_status = d_status.as<int32_t>();
        HIP_TRY(launch_walk(ix->eng, dev, wa, 0, 0, st)); // (the throughput kernel: the latency kernels have no unseeded filter)
        int32_t wst = COS_OK;
        HIP_TRY(hipMemcpyAsync(&wst, d_status.p, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h_ids.data(), d_out_ids.p, h_ids.size() * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h_nodes.data(), d_out_nodes.p, h_nodes.size() * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h_counts.data(), d_out_counts.p, (size_t)L1 * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (wst != COS_OK) return cos_fail(wst, "vector %u cannot be walked for (zero norm -> DistanceError::CalculationError)", id);
        bool any = false;
        for (u32 l = 0; l <= Ltop; l++) { // list slot = num_layers - level
            const u32 slot = Ltop - l;
            h_dn[l] = NONE;
            for (u32 i = 0; i < h_counts[slot]; i++)
                if (h_ids[(size_t)slot * KEEP + i] == id) { h_dn[l] = h_nodes[(size_t)slot * KEEP + i]; any = true; break; }
        }
        if (!any) continue;
        HIP_TRY(hipMemcpyAsync(d_dn.p, h_dn.data(), (size_t)L1 * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(launch_unlink(la, d_dn.as<u32>(), d_ust.as<u32>(), st));
        HIP_TRY(hipMemcpyAsync(h_ust.data(), d_ust.p, (size_t)L1 * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (u32 l = 0; l <= Ltop; l++) {
            if (h_ust[l] != 1u) continue;
            // ---- the reference's own order for this level (vector_store.rs:1277-1369)
            const LevelHost &H = ix->lv[l];
            const u32 M = H.M, slot = Ltop - l, node = h_dn[l];
            if (l > 0 && node_vec[l].empty()) {
                node_vec[l].resize(H.n);
                HIP_TRY(hipMemcpy(node_vec[l].data(), H.d_node_vec, (size_t)H.n * 4, hipMemcpyDeviceToHost));
            }
            auto vec_row = [&](u32 nd) { return l == 0 ? nd : node_vec[l][nd]; };
            std::vector<u32> res; // the walk's results minus the node (swap_remove :1277; the order is irrelevant: re-sorted below)
            for (u32 i = 0; i < h_counts[slot]; i++) if (h_nodes[(size_t)slot * KEEP + i] != node) res.push_back(h_nodes[(size_t)slot * KEEP + i]);
            RowCache cache{ix, l, M};
            RowCache::Row &dr = cache.get(node);
            for (u32 j = 0; j < M; j++) {
                const u32 x = dr.adj[j];
                if (x == NONE) continue;
                RowCache::Row &xr = cache.get(x);
                bool removed = false, empty = true;
                for (u32 k = 0; k < M; k++) if (xr.adj[k] == node) { xr.adj[k] = NONE; xr.key[k] = INT32_MIN; xr.dirty = true; removed = true; break; }
                for (u32 k = 0; k < M; k++) if (xr.adj[k] != NONE) { empty = false; break; }
                if (!(removed && empty)) continue;
                std::vector<u32> cand, px, py;
                for (u32 r : res) if (r != x) { cand.push_back(r); px.push_back(vec_row(x)); py.push_back(vec_row(r)); }
                std::vector<float> sims(cand.size());
                std::vector<int32_t> dst(cand.size());
                if (!cand.empty()) {
                    HIP_TRY(hipMemcpyAsync(d_px.p, px.data(), px.size() * 4, hipMemcpyHostToDevice, st));
                    HIP_TRY(hipMemcpyAsync(d_py.p, py.data(), py.size() * 4, hipMemcpyHostToDevice, st));
                    HIP_TRY(cosdev::launch_index_pair_distances(ix->eng, ix->d_codes, ix->d_mags, ix->row_stride, ix->nchunks, ix->p.dim, metric, d_px.as<u32>(), d_py.as<u32>(),
                                                               (u32)cand.size(), d_ds.as<float>(), d_dst.as<int32_t>(), st));
                    HIP_TRY(hipMemcpyAsync(sims.data(), d_ds.p, sims.size() * 4, hipMemcpyDeviceToHost, st));
                    HIP_TRY(hipMemcpyAsync(dst.data(), d_dst.p, dst.size() * 4, hipMemcpyDeviceToHost, st));
                    HIP_TRY(hipStreamSynchronize(st));
                    for (int32_t e : dst) if (e != COS_OK) return cos_fail(e, "re-linking node %u after the delete of %u failed (DistanceError)", x, id);
                }
                std::vector<u32> ord(cand.size());
                for (u32 i = 0; i < ord.size(); i++) ord[i] = i;
                std::sort(ord.begin(), ord.end(), [&](u32 a, u32 b) { // descending; larger id (= node index) first on ties
                    const int32_t ka = order_key(metric, sims[a]), kb = order_key(metric, sims[b]);
                    return ka != kb ? ka > kb : cand[a] > cand[b];
                });
                u32 succ = 0; // create_node_edges (vector_store.rs:976-1074)
                for (u32 oi : ord) {
                    if (succ >= M) break;
                    const int32_t k = order_key(metric, sims[oi]);
                    const int r1 = add_neighbor_host(cache, x, cand[oi], k, kmin, kmax);
                    if (r1 < 0) continue;
                    const int r2 = add_neighbor_host(cache, cand[oi], x, k, kmin, kmax);
                    if (r2 >= 0) succ++;
                    else {
                        RowCache::Row &xr2 = cache.get(x);
                        if (xr2.adj[(u32)r1] == cand[oi]) { xr2.adj[(u32)r1] = NONE; xr2.key[(u32)r1] = INT32_MIN; xr2.dirty = true; } // remove_neighbor_by_index_and_id (an empty slot's key is EMPTY_KEY, like the link kernel leaves it)
                    }
                }
            }
            RowCache::Row &dr2 = cache.get(node);
            for (u32 j = 0; j < M; j++) { dr2.adj[j] = NONE; dr2.key[j] = INT32_MIN; }
            dr2.dirty = true;
            if (cache.err != hipSuccess) HIP_TRY(cache.err);
            HIP_TRY(cache.flush(node_vec[l]));
        }
    }
    HIP_TRY(hipStreamSynchronize(st));
    if (int32_t rc2 = cos_prepare_walk_plans(ix)) return rc2; // the adjacency-side norms of the changed rows
    return COS_OK;
}

// ================================================================================================================================
// The pseudo-root component of a collection with a metadata schema, built on the device (rounds 1-2 took it from the
// oracle and uploaded it).
//   insert order: pseudo nodes, then the Metadata replicas of every embedding      api_service.rs:135-210, vector_store.rs:484-640
//   index_embedding for such a node: every level walked from the pseudo root with the node's OWN metadata dimensions as the
//   query's filter (walk_meta_kernel<.., INDEXING>), keep 64, its id pre-inserted in the visited filter
//   create_node_edges with the two refusal rules of vector_store.rs:1017-1041     (link_kernel, LinkLevelDev::kind)
// Same batch-synchronous schedule as cos_index_build — batches of min(Bmax, max(1, inserted / 4)) nodes walk the snapshot that
// precedes them, then the round-synchronous claim / link / evict kernels connect them; oracle/cosdata_oracle_hnsw.c:
// coso_meta_build_rounds is the CPU statement and tests/test_gpu_meta.py asserts identical graphs.  The level of every node comes
// from the caller (the reference draws it with thread_rng; pseudo nodes from pseudo_level_probs, metadata/mod.rs:182-209).
// ================================================================================================================================
extern "C" int32_t cos_index_build_meta(cos_index *ix, uint32_t n_nodes, const uint32_t *node_ids, const int32_t *mbits, const uint8_t *max_levels,
                                        uint32_t batch_size) {
    if (!ix || !node_ids || !mbits || !max_levels || n_nodes == 0) return cos_fail(COS_ERR_INVALID, "null/empty node table");
    constexpr u32 PSEUDO_LO = 0xFFFFFEFEu, PSEUDO_HI = 0xFFFFFFFDu;
    u32 vmode;
    { std::lock_guard<std::mutex> g(ix->mu); vmode = ix->p.visited_mode; }
    if (vmode != COS_VISITED_REF) return cos_fail(COS_ERR_UNIMPLEMENTED, "the metadata component is built and searched with the reference's PerformantFixedSet filter only");
    int32_t rc = cos_index_upload_meta_nodes(ix, n_nodes, node_ids, mbits); // validates, uploads mbits / norms, drops the old levels
    if (rc) return rc;
    rc = cos_set_device(ix);
    if (rc) return rc;
    const u32 n = ix->n, Ltop = ix->p.num_layers, L1 = Ltop + 1, metric = ix->p.metric, md = ix->meta.mdim;
    // default batch 1024 (the base graph's: 4096): every Metadata replica links to the same few pseudo nodes, the rounds of a batch run ~1.4
    // nodes each whatever its size, and a round costs what its PENDING list costs — 600 k nodes: 27 s at 4096, 16.4 s at 1024 or 256
    const u32 Bmax = std::min<u32>(batch_size ? batch_size : 1024u, LINK_MAX_BATCH);
    hipStream_t st = ix->own_stream;
    auto is_pseudo = [&](u32 id) { return id >= PSEUDO_LO && id <= PSEUDO_HI; };
    const std::vector<u32> &tab = ix->meta.node_ids;
    if (!std::binary_search(tab.begin(), tab.end(), PSEUDO_LO)) return cos_fail(COS_ERR_INVALID, "the node table has no pseudo root (u32::MAX - 257)");
    for (u32 i = 0; i < n_nodes; i++)
        if (max_levels[i] > Ltop) return cos_fail(COS_ERR_INVALID, "node %u: level %u above num_layers %u", node_ids[i], max_levels[i], Ltop);
    // Metadata::from (types.rs:112-126): a node whose metadata dimensions are all zero is a Base replica — it lives under the main root
    std::vector<float> mmag(n_nodes);
    std::vector<uint8_t> kind(n_nodes);
    for (u32 i = 0; i < n_nodes; i++) {
        float acc = -0.0f;
        for (u32 j = 0; j < md; j++) { const float x = (float)mbits[(size_t)i * md + j]; acc = acc + x * x; }
        mmag[i] = sqrtf(acc);
        kind[i] = mmag[i] == 0.0f ? 0 : (is_pseudo(node_ids[i]) ? 1 : 2);
    }
    std::vector<u32> ord; // insertion order: pseudo nodes (ascending id, the root excluded), then the Metadata replicas (ascending id)
    for (int pass = 0; pass < 2; pass++)
        for (u32 i = 0; i < n_nodes; i++) {
            const bool ps = is_pseudo(node_ids[i]);
            if (node_ids[i] == PSEUDO_LO || ps != (pass == 0) || kind[i] == 0) continue;
            ord.push_back(i);
        }
    const u32 total = (u32)ord.size();

    struct Guard { // a failed build leaves the handle without the component (filtered search: NotReady)
        cos_index *ix;
        bool armed = true;
        ~Guard() { if (armed) { (void)hipStreamSynchronize(ix->own_stream); cos_meta_free_levels(ix); } }
    } guard{ix};

    // ---- level skeletons: the root + every node whose level reaches l, ascending id; every slot empty ------------------------------
    u32 maxM = 0;
    std::vector<DevBuf> key(L1), low_idx(L1), low_key(L1), owner(L1), d_kind(L1);
    std::vector<std::vector<u32>> tab_of(L1); // per level: node index -> row of the node table
    LinkArgs la;
    memset(&la, 0, sizeof(la));
    ix->meta.lv.resize(L1);
    for (u32 l = 0; l <= Ltop; l++) {
        LevelHost &H = ix->meta.lv[l];
        H.M = l == 0 ? ix->p.level0_neighbors_count : ix->p.neighbors_count;
        std::vector<u32> rows; // node-table rows of this level, ascending id (the table is ascending)
        for (u32 i = 0; i < n_nodes; i++)
            if (node_ids[i] == PSEUDO_LO || (kind[i] != 0 && max_levels[i] >= l)) rows.push_back(i);
        const u32 nl = (u32)rows.size(), M = H.M;
        maxM = std::max(maxM, M);
        H.node_ids.resize(nl);
        std::vector<u32> node_vec(nl), node_meta(nl), child(nl);
        std::vector<uint8_t> kd(nl);
        for (u32 i = 0; i < nl; i++) {
            const u32 id = node_ids[rows[i]];
            H.node_ids[i] = id;
            node_vec[i] = is_pseudo(id) ? n + 1 : id / ix->id_stride;
            node_meta[i] = rows[i];
            kd[i] = kind[rows[i]];
            if (id == PSEUDO_LO) { H.root_idx = i; kd[i] = 1; }
            if (l > 0) {
                const std::vector<u32> &D = ix->meta.lv[l - 1].node_ids;
                child[i] = (u32)(std::lower_bound(D.begin(), D.end(), id) - D.begin());
            }
        }
        tab_of[l] = std::move(rows);
        auto up = [&](u32 *&dst, const std::vector<u32> &src) -> hipError_t {
            hipError_t e = hipMalloc((void **)&dst, std::max<size_t>(src.size(), 1) * 4);
            return e == hipSuccess ? hipMemcpy(dst, src.data(), src.size() * 4, hipMemcpyHostToDevice) : e;
        };
        HIP_TRY(hipMalloc((void **)&H.d_adj_vec, (size_t)nl * M * 4));
        HIP_TRY(hipMalloc((void **)&H.d_adj_node, (size_t)nl * M * 4));
        HIP_TRY(hipMemsetAsync(H.d_adj_vec, 0xFF, (size_t)nl * M * 4, st));
        HIP_TRY(hipMemsetAsync(H.d_adj_node, 0xFF, (size_t)nl * M * 4, st));
        HIP_TRY(up(H.d_node_vec, node_vec));
        HIP_TRY(up(H.d_node_id, H.node_ids));
        HIP_TRY(up(H.d_node_meta, node_meta));
        if (l > 0) HIP_TRY(up(H.d_child, child));
        H.n = nl;
        H.host_valid = false;
        HIP_TRY(key[l].alloc((size_t)nl * M * 4));
        HIP_TRY(low_idx[l].alloc(nl));
        HIP_TRY(low_key[l].alloc((size_t)nl * 4));
        HIP_TRY(owner[l].alloc((size_t)nl * 4));
        HIP_TRY(d_kind[l].alloc(nl));
        HIP_TRY(launch_fill_i32(key[l].as<int32_t>(), (u64)nl * M, INT32_MIN, st));
        HIP_TRY(hipMemsetAsync(low_idx[l].p, 0, nl, st));
        HIP_TRY(launch_fill_i32(low_key[l].as<int32_t>(), nl, order_key(metric, metric_min(metric)), st));
        HIP_TRY(hipMemsetAsync(owner[l].p, 0, (size_t)nl * 4, st));
        HIP_TRY(hipMemcpy(d_kind[l].p, kd.data(), nl, hipMemcpyHostToDevice));
        LinkLevelDev &D = la.lv[l];
        D.adj_vec = H.d_adj_vec;
        D.adj_node = H.d_adj_node;
        D.node_vec = H.d_node_vec;
        D.key = key[l].as<int32_t>();
        D.low_idx = low_idx[l].as<uint8_t>();
        D.low_key = low_key[l].as<int32_t>();
        D.owner = owner[l].as<u32>();
        D.kind = d_kind[l].as<uint8_t>();
        D.M = M;
    }
    if (maxM > 256) return cos_fail(COS_ERR_UNIMPLEMENTED, "more than 256 neighbour slots per node");
    HIP_TRY(hipStreamSynchronize(st));

    const u32 KEEP = (u32)KEEP_INDEX;
    DevBuf d_rows, d_self, d_foff, d_fdims, d_fmags, d_qc, d_qm, d_out_ids, d_out_nodes, d_out_sims, d_out_counts, d_status, d_me, d_pend[2], d_cnt, d_evq;
    const size_t pend_cap = (size_t)Bmax * L1, evq_cap = pend_cap * 2 * KEEP;
    HIP_TRY(d_rows.alloc((size_t)Bmax * 4));
    HIP_TRY(d_self.alloc((size_t)Bmax * 4));
    HIP_TRY(d_foff.alloc(((size_t)Bmax + 1) * 4));
    HIP_TRY(d_fdims.alloc((size_t)Bmax * md * 4));
    HIP_TRY(d_fmags.alloc((size_t)Bmax * 4));
    HIP_TRY(d_out_ids.alloc((size_t)Bmax * L1 * KEEP * 4));
    HIP_TRY(d_out_nodes.alloc((size_t)Bmax * L1 * KEEP * 4));
    HIP_TRY(d_out_sims.alloc((size_t)Bmax * L1 * KEEP * 4));
    HIP_TRY(d_out_counts.alloc((size_t)Bmax * L1 * 4));
    HIP_TRY(d_status.alloc((size_t)Bmax * 4));
    HIP_TRY(d_me.alloc((size_t)Bmax * L1 * 4));
    HIP_TRY(d_pend[0].alloc(pend_cap * 4));
    HIP_TRY(d_pend[1].alloc(pend_cap * 4));
    HIP_TRY(d_cnt.alloc(4 * 4));
    HIP_TRY(d_evq.alloc(evq_cap * 3 * 4));
    std::vector<u32> h_rows(Bmax), h_self(Bmax), h_foff(Bmax + 1), h_me((size_t)Bmax * L1), h_pend(pend_cap);
    std::vector<int32_t> h_fdims((size_t)Bmax * md), h_status(Bmax);
    std::vector<float> h_fmags(Bmax);
    u32 h_cnt[4];

    la.L1 = L1;
    la.z_nodes = d_out_nodes.as<u32>();
    la.z_sims = d_out_sims.as<float>();
    la.z_counts = d_out_counts.as<u32>();
    la.me = d_me.as<u32>();
    la.metric = metric;
    la.kmin = order_key(metric, metric_min(metric));
    la.kmax = order_key(metric, metric_max(metric));

    IndexDev dev = cos_make_meta_dev(ix);
    dev.visited_mode = COS_VISITED_REF;
    u32 inserted = 0, round = 0;
    u32 *cnt = d_cnt.as<u32>();
    while (inserted < total) {
        const u32 bs = std::min({Bmax, std::max(1u, inserted / 4u), total - inserted});
        u32 np = 0;
        for (u32 b = 0; b < bs; b++) {
            const u32 t = ord[inserted + b], id = node_ids[t];
            h_rows[b] = is_pseudo(id) ? n + 1 : id / ix->id_stride;
            h_self[b] = id;
            h_foff[b] = b;
            for (u32 j = 0; j < md; j++) h_fdims[(size_t)b * md + j] = mbits[(size_t)t * md + j];
            h_fmags[b] = mmag[t];
            for (u32 l = 0; l <= Ltop; l++) {
                u32 m = NONE;
                if (max_levels[t] >= l) {
                    const std::vector<u32> &ids = ix->meta.lv[l].node_ids;
                    m = (u32)(std::lower_bound(ids.begin(), ids.end(), id) - ids.begin());
                    h_pend[np++] = (l << 16) | b;
                }
                h_me[(size_t)b * L1 + l] = m;
            }
        }
        h_foff[bs] = bs;
        HIP_TRY(hipMemcpyAsync(d_rows.p, h_rows.data(), (size_t)bs * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_self.p, h_self.data(), (size_t)bs * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_foff.p, h_foff.data(), ((size_t)bs + 1) * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_fdims.p, h_fdims.data(), (size_t)bs * md * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_fmags.p, h_fmags.data(), (size_t)bs * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_me.p, h_me.data(), (size_t)bs * L1 * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_pend[0].p, h_pend.data(), (size_t)np * 4, hipMemcpyHostToDevice, st));
        h_cnt[0] = np; h_cnt[1] = 0; h_cnt[2] = 0; h_cnt[3] = 0;
        HIP_TRY(hipMemcpyAsync(cnt, h_cnt, 16, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st)); // the staging vectors are pageable and rewritten by the next batch
        WalkArgs wa;
        memset(&wa, 0, sizeof(wa));
        wa.qcodes = ix->d_codes;
        wa.qmags = ix->d_mags;
        wa.q_rows = d_rows.as<u32>();
        wa.self_ids = d_self.as<u32>();
        wa.B = bs;
        wa.ef = ix->p.ef_construction;
        wa.keep = KEEP;
        wa.out_ids = d_out_ids.as<u32>();
        wa.out_sims = d_out_sims.as<float>();
        wa.out_nodes = d_out_nodes.as<u32>();
        wa.out_counts = d_out_counts.as<u32>();
        wa.out_status = d_status.as<int32_t>();
        wa.f_dims = d_fdims.as<int32_t>();
        wa.f_mags = d_fmags.as<float>();
        wa.f_off = d_foff.as<u32>();
        HIP_TRY(launch_walk_meta_index(ix->eng, dev, wa, st));
        HIP_TRY(hipMemcpyAsync(h_status.data(), d_status.p, (size_t)bs * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (u32 b = 0; b < bs; b++)
            if (h_status[b] != COS_OK)
                return cos_fail(h_status[b], "node %u of the metadata component cannot be indexed (status %d)", node_ids[ord[inserted + b]], h_status[b]);
        u32 pending_ub = np, cur = 0;
        while (pending_ub > 0) {
            const u32 chunk = pending_ub > 64 ? 2 : 4;
            for (u32 r = 0; r < chunk; r++) {
                if (++round > LINK_MAX_ROUND) {
                    for (u32 l = 0; l <= Ltop; l++) HIP_TRY(hipMemsetAsync(owner[l].p, 0, (size_t)ix->meta.lv[l].n * 4, st));
                    round = 1;
                }
                HIP_TRY(launch_link_round(la, maxM, d_pend[cur].as<u32>(), cnt + cur, d_pend[cur ^ 1].as<u32>(), cnt + (cur ^ 1), d_evq.as<u32>(), cnt + 2,
                                          pending_ub, round, st));
                cur ^= 1;
            }
            HIP_TRY(hipMemcpyAsync(h_cnt, cnt, 16, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            pending_ub = h_cnt[cur];
        }
        inserted += bs;
    }
    HIP_TRY(hipStreamSynchronize(st));
    guard.armed = false;
    return COS_OK;
}
