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

extern "C" int32_t cos_index_build(cos_index *ix, uint32_t batch_size) {
    if (!ix) return cos_fail(COS_ERR_INVALID, "null index");
    if (!ix->have_vectors) return cos_fail(COS_ERR_NOT_READY, "upload vectors before building");
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    const u32 n = ix->n, Ltop = ix->p.num_layers, L1 = Ltop + 1, metric = ix->p.metric;
    const u32 Bmax = std::min<u32>(batch_size ? batch_size : 4096u, LINK_MAX_BATCH);
    hipStream_t st = ix->own_stream;

    // a failed build leaves the handle WITHOUT a graph (search returns NotReady) instead of a half-linked one
    struct Guard {
        cos_index *ix;
        bool armed = true;
        ~Guard() {
            if (!armed) return;
            (void)hipStreamSynchronize(ix->own_stream);
            for (auto &l : ix->lv) {
                void *ptrs[] = {l.d_adj_vec, l.d_adj_node, l.d_node_vec, l.d_child};
                for (void *p : ptrs) if (p) (void)hipFree(p);
                l.d_adj_vec = l.d_adj_node = l.d_node_vec = l.d_child = nullptr;
                l.n = 0;
                l.node_ids.clear();
                l.nbr_ids.clear();
                l.host_valid = false;
            }
            ix->have_root = false;
        }
    } guard{ix};

    // ---- root + level draws (same RNG stream as the oracle builders) ----------------------------
    uint64_t rng = ix->p.seed ? ix->p.seed : 0x1234567ull;
    std::vector<float> root(ix->p.dim);
    for (u32 i = 0; i < ix->p.dim; i++) root[i] = ix->p.range_lo + rand_f32(rng) * (ix->p.range_hi - ix->p.range_lo); // vector_store.rs:30-36
    rc = cos_index_set_root(ix, root.data());
    if (rc) return rc;
    std::vector<double> pv(L1);
    for (u32 k = 0; k <= Ltop; k++) { // generate_level_probs(4.0, L): 1 - 4^-n, n = L..0 (common.rs:421-429)
        const int nn = (int)(Ltop - k);
        double r = 1.0, a = 4.0;
        for (int b = nn;;) { if (b & 1) r *= a; b /= 2; if (b == 0) break; a *= a; }
        pv[k] = 1.0 - 1.0 / r;
    }
    std::vector<uint8_t> max_level(n);
    for (u32 id = 0; id < n; id++) {
        const double x = (double)rand_f32(rng);
        u32 k = 0;
        while (k < Ltop && !(x >= pv[k])) k++; // get_max_insert_level (common.rs:373-379)
        max_level[id] = (uint8_t)(Ltop - k);
    }

    // ---- level skeletons: nodes of a level in ascending id, root last; every slot empty ------------------------
    u32 maxM = 0;
    std::vector<DevBuf> key(L1), low_idx(L1), low_key(L1), owner(L1); // link state, released when the build ends
    LinkArgs la;
    memset(&la, 0, sizeof(la));
    for (u32 l = 0; l <= Ltop; l++) {
        LevelHost &H = ix->lv[l];
        void *old[] = {H.d_adj_vec, H.d_adj_node, H.d_node_vec, H.d_child};
        for (void *p : old) if (p) (void)hipFree(p);
        H.d_adj_vec = H.d_adj_node = H.d_node_vec = H.d_child = nullptr;
        H.host_valid = false;
        H.nbr_ids.clear();
        H.node_ids.clear();
        for (u32 id = 0; id < n; id++)
            if (max_level[id] >= l) H.node_ids.push_back(id * ix->id_stride); // internal id of vector row `id`
        H.node_ids.push_back(COS_ROOT_ID);
        const u32 nl = (u32)H.node_ids.size(), M = H.M;
        maxM = std::max(maxM, M);
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
        HIP_TRY(key[l].alloc((size_t)nl * M * 4));
        HIP_TRY(low_idx[l].alloc(nl));
        HIP_TRY(low_key[l].alloc((size_t)nl * 4));
        HIP_TRY(owner[l].alloc((size_t)nl * 4));
        HIP_TRY(launch_fill_i32(key[l].as<int32_t>(), (u64)nl * M, INT32_MIN, st));
        HIP_TRY(hipMemsetAsync(low_idx[l].p, 0, nl, st));
        HIP_TRY(launch_fill_i32(low_key[l].as<int32_t>(), nl, order_key(metric, metric_min(metric)), st));
        HIP_TRY(hipMemsetAsync(owner[l].p, 0, (size_t)nl * 4, st));
        LinkLevelDev &D = la.lv[l];
        D.adj_vec = H.d_adj_vec;
        D.adj_node = l == 0 ? H.d_adj_vec : H.d_adj_node;
        D.node_vec = l == 0 ? nullptr : H.d_node_vec;
        D.key = key[l].as<int32_t>();
        D.low_idx = low_idx[l].as<uint8_t>();
        D.low_key = low_key[l].as<int32_t>();
        D.owner = owner[l].as<u32>();
        D.M = M;
    }
    if (maxM > 256) return cos_fail(COS_ERR_UNIMPLEMENTED, "more than 256 neighbour slots per node");
    HIP_TRY(hipStreamSynchronize(st));

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
    std::vector<u32> cursor(L1, 0); // first node of each level not yet inserted (nodes are in id order)
    u32 inserted = 0, round = 0;
    u64 n_rounds = 0, n_batches = 0;
    const bool prof = getenv("COS_BUILD_PROFILE") != nullptr;
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
                if (max_level[id] >= l) { m = cursor[l]++; h_pend.as<u32>()[np++] = (l << 16) | b; }
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
                    for (u32 l = 0; l <= Ltop; l++) HIP_TRY(hipMemsetAsync(owner[l].p, 0, (size_t)ix->lv[l].n * 4, st));
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
        fprintf(stderr, "[cos_index_build] n=%u batches=%llu link rounds=%llu walk %.2fs link %.2fs\n", n, (unsigned long long)n_batches,
                (unsigned long long)n_rounds, t_walk, t_link);
    guard.armed = false;
    return COS_OK; // the id-format host copy is made on demand (cos_index_download_graph_level)
}
