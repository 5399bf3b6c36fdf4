// builder.hip — vector_store::index_embeddings (vector_store.rs:714-1109) for a device-resident index.
//
// Batch-synchronous construction: the ids of a batch run the reference's per-level walk
// (traverse_find_nearest with ef_construction, keep 64, visited pre-seeded with the new id) on the
// GPU against the graph snapshot that precedes the batch — one wavefront per new vector, the same
// walk kernel the search path uses — then their edges are connected in id order with the
// reference's exact edge semantics (create_node_edges / ProbNode::add_neighbor: replace-lowest slot,
// bidirectional accept-or-rollback, evictee drops its back edge, stale lowest cache).  The link step
// is graph bookkeeping on the host mirror; the rows it dirtied are scattered back to HBM before the
// next batch.  oracle/cosdata_oracle_hnsw.c:coso_index_build_batched is the CPU statement of the same
// schedule; tests assert the two graphs are identical.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

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
inline float metric_min(u32 metric) { return metric == COS_METRIC_COSINE ? -1.0f : -INFINITY; }
inline float metric_max(u32 metric) { return metric == COS_METRIC_COSINE ? 2.0f : INFINITY; }

struct LevelBuild {
    u32 n = 0, M = 0;
    std::vector<u32> node_ids, node_vec, child;
    std::vector<u32> nbr;       // [n][M] node index / NONE
    std::vector<int32_t> key;   // [n][M] MetricResult order key of the slot's similarity; EMPTY_KEY for a null slot
    std::vector<uint8_t> low_idx;
    std::vector<int32_t> low_key; // cached (lowest_index, lowest_sim) of prob_node.rs:108, as an order key
    std::vector<u32> stamp;     // dirty marker per node (batch number)
    std::vector<u32> dirty;
    u32 cursor = 0;             // first node of this level not yet inserted
};
constexpr int32_t EMPTY_KEY = INT32_MIN; // below every real key: the first-minimum scan finds the first empty slot

inline int32_t order_key(u32 metric, float v) {
    const int32_t k = total_key(v);
    return (metric == COS_METRIC_EUCLIDEAN || metric == COS_METRIC_HAMMING) ? ~k : k;
}

inline void mark_dirty(LevelBuild &L, u32 node, u32 batch_no) {
    if (L.stamp[node] != batch_no) { L.stamp[node] = batch_no; L.dirty.push_back(node); }
}

// ProbNode::add_neighbor (prob_node.rs:210-283) on the host mirror; similarities are compared through their
// MetricResult order keys (types.rs:401-411).  Returns the slot or -1.
inline int add_neighbor(LevelBuild &L, int32_t kmin, int32_t kmax, u32 self, u32 nbr, int32_t dist, u32 batch_no) {
    const u32 M = L.M;
    const u32 lowest_idx = L.low_idx[self];
    if (dist <= L.low_key[self]) return -1;
    u32 *nb = &L.nbr[(size_t)self * M];
    int32_t *nk = &L.key[(size_t)self * M];
    const bool ok = nb[lowest_idx] == NONE || dist > nk[lowest_idx];
    u32 old = NONE;
    if (ok) { old = nb[lowest_idx]; nb[lowest_idx] = nbr; nk[lowest_idx] = dist; mark_dirty(L, self, batch_no); }
    // new lowest: the first empty slot if any (lowest_sim = MetricResult::min), else the first strictly-smallest similarity
    // (the scan starts from MetricResult::max — 2.0 for cosine — exactly like prob_node.rs:245-259: slots whose
    // similarity is not below it can never become the lowest; quantized "cosines" above 2 do occur)
    int32_t mn = kmax;
    u32 mi = 0;
    for (u32 j = 0; j < M; j++)
        if (nk[j] < mn) { mn = nk[j]; mi = j; }
    L.low_idx[self] = (uint8_t)mi;
    L.low_key[self] = mn == EMPTY_KEY ? kmin : mn;
    if (!ok) return -1;
    if (old != NONE) { // the evictee drops its back edge; its lowest cache is NOT refreshed (prob_node.rs:285-306)
        u32 *ob = &L.nbr[(size_t)old * M];
        for (u32 j = 0; j < M; j++)
            if (ob[j] == self) { ob[j] = NONE; L.key[(size_t)old * M + j] = EMPTY_KEY; mark_dirty(L, old, batch_no); break; }
    }
    return (int)lowest_idx;
}

// create_node_edges (vector_store.rs:976-1074)
inline void create_node_edges(LevelBuild &L, u32 metric, int32_t kmin, int32_t kmax, u32 node, const u32 *z_nodes, const float *z_sims, u32 zn, u32 batch_no) {
    u32 succ = 0;
    for (u32 i = 0; i < zn; i++) {
        if (succ >= L.M) break;
        const int32_t dk = order_key(metric, z_sims[i]);
        const int r = add_neighbor(L, kmin, kmax, node, z_nodes[i], dk, batch_no);
        if (r >= 0) {
            const int r2 = add_neighbor(L, kmin, kmax, z_nodes[i], node, dk, batch_no);
            if (r2 >= 0) succ++;
            else if (L.nbr[(size_t)node * L.M + (u32)r] == z_nodes[i]) { // remove_neighbor_by_index_and_id
                L.nbr[(size_t)node * L.M + (u32)r] = NONE;
                L.key[(size_t)node * L.M + (u32)r] = EMPTY_KEY;
                mark_dirty(L, node, batch_no);
            }
        }
    }
}

// scatter dirtied rows: packed holds neighbour NODE indices; the vector-row array is derived on the device
__global__ void scatter_rows2_kernel(u32 *__restrict__ adj_vec, u32 *__restrict__ adj_node, const u32 *__restrict__ node_vec,
                                     const u32 *__restrict__ rows, const u32 *__restrict__ packed, u32 n_rows, u32 M) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (u64)n_rows * M) return;
    const u32 r = (u32)(i / M), j = (u32)(i % M);
    const u32 nb = packed[i];
    const u64 o = (u64)rows[r] * M + j;
    if (adj_node) adj_node[o] = nb;
    adj_vec[o] = (nb == ROW_EMPTY || !node_vec) ? nb : node_vec[nb];
}

template <typename T>
hipError_t dmalloc(T *&p, size_t count) { return hipMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T)); }

} // namespace

extern "C" int32_t cos_index_build(cos_index *ix, uint32_t batch_size) {
    if (!ix) return cos_fail(COS_ERR_INVALID, "null index");
    if (!ix->have_vectors) return cos_fail(COS_ERR_NOT_READY, "upload vectors before building");
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    const u32 n = ix->n, Ltop = ix->p.num_layers, metric = ix->p.metric;
    const u32 Bmax = batch_size ? batch_size : 4096u;
    hipStream_t st = ix->own_stream;

    // ---- root + level draws (same RNG stream as the oracle builder) ----------------------------
    uint64_t rng = ix->p.seed ? ix->p.seed : 0x1234567ull;
    std::vector<float> root(ix->p.dim);
    for (u32 i = 0; i < ix->p.dim; i++) root[i] = ix->p.range_lo + rand_f32(rng) * (ix->p.range_hi - ix->p.range_lo); // vector_store.rs:30-36
    rc = cos_index_set_root(ix, root.data());
    if (rc) return rc;
    std::vector<double> pv(Ltop + 1);
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

    // ---- level skeletons -----------------------------------------------------------------------
    std::vector<LevelBuild> lb(Ltop + 1);
    for (u32 l = 0; l <= Ltop; l++) {
        LevelBuild &L = lb[l];
        L.M = ix->lv[l].M;
        for (u32 id = 0; id < n; id++)
            if (max_level[id] >= l) L.node_ids.push_back(id);
        L.node_ids.push_back(COS_ROOT_ID);
        L.n = (u32)L.node_ids.size();
        L.node_vec.resize(L.n);
        for (u32 i = 0; i < L.n; i++) L.node_vec[i] = L.node_ids[i] == COS_ROOT_ID ? n : L.node_ids[i];
        L.nbr.assign((size_t)L.n * L.M, NONE);
        L.key.assign((size_t)L.n * L.M, EMPTY_KEY);
        L.low_idx.assign(L.n, 0);                    // prob_node.rs:140
        L.low_key.assign(L.n, order_key(metric, metric_min(metric)));
        L.stamp.assign(L.n, 0);
        if (l > 0) {
            L.child.resize(L.n);
            const std::vector<u32> &D = lb[l - 1].node_ids;
            for (u32 i = 0; i < L.n; i++) L.child[i] = (u32)(std::lower_bound(D.begin(), D.end(), L.node_ids[i]) - D.begin());
        }
    }
    // device arrays at full size, every slot empty
    for (u32 l = 0; l <= Ltop; l++) {
        LevelHost &H = ix->lv[l];
        LevelBuild &L = lb[l];
        if (H.d_adj_vec) (void)hipFree(H.d_adj_vec);
        if (H.d_adj_node) (void)hipFree(H.d_adj_node);
        if (H.d_node_vec) (void)hipFree(H.d_node_vec);
        if (H.d_child) (void)hipFree(H.d_child);
        H.d_adj_vec = H.d_adj_node = H.d_node_vec = H.d_child = nullptr;
        H.host_valid = false;
        HIP_TRY(dmalloc(H.d_adj_vec, (size_t)L.n * L.M));
        HIP_TRY(hipMemsetAsync(H.d_adj_vec, 0xFF, (size_t)L.n * L.M * 4, st));
        if (l > 0) {
            HIP_TRY(dmalloc(H.d_adj_node, (size_t)L.n * L.M));
            HIP_TRY(hipMemsetAsync(H.d_adj_node, 0xFF, (size_t)L.n * L.M * 4, st));
            HIP_TRY(dmalloc(H.d_node_vec, L.n));
            HIP_TRY(hipMemcpyAsync(H.d_node_vec, L.node_vec.data(), (size_t)L.n * 4, hipMemcpyHostToDevice, st));
            HIP_TRY(dmalloc(H.d_child, L.n));
            HIP_TRY(hipMemcpyAsync(H.d_child, L.child.data(), (size_t)L.n * 4, hipMemcpyHostToDevice, st));
        }
        H.n = L.n;
    }
    HIP_TRY(hipStreamSynchronize(st));

    // ---- batch workspace -----------------------------------------------------------------------
    const u32 L1 = Ltop + 1, KEEP = (u32)KEEP_INDEX;
    u32 *d_rows = nullptr, *d_out_ids = nullptr, *d_out_nodes = nullptr, *d_out_counts = nullptr, *d_pack = nullptr;
    float *d_out_sims = nullptr;
    int32_t *d_status = nullptr;
    VisTab vtab; // EXACT build mode: visited filters of the batch walks
    size_t pack_cap = 0;
    // pinned staging: walk results down, packed dirty rows up
    u32 *h_rows = nullptr, *h_nodes = nullptr, *h_counts = nullptr, *h_pack = nullptr;
    float *h_sims = nullptr;
    int32_t *h_status = nullptr;
    size_t h_pack_cap = 0;
    auto cleanup = [&]() {
        void *ptrs[] = {d_rows, d_out_ids, d_out_nodes, d_out_sims, d_out_counts, d_status, d_pack, vtab.bits, vtab.log};
        for (void *p : ptrs) if (p) (void)hipFree(p);
        void *hp[] = {h_rows, h_nodes, h_sims, h_counts, h_status, h_pack};
        for (void *p : hp) if (p) (void)hipHostFree(p);
    };
#define BUILD_TRY(expr)                                                                                                    \
    do {                                                                                                                   \
        hipError_t _be = (expr);                                                                                           \
        if (_be != hipSuccess) { cleanup(); HIP_TRY(_be); }                                                                \
    } while (0)
    BUILD_TRY(dmalloc(d_rows, Bmax));
    BUILD_TRY(dmalloc(d_out_ids, (size_t)Bmax * L1 * KEEP));
    BUILD_TRY(dmalloc(d_out_nodes, (size_t)Bmax * L1 * KEEP));
    BUILD_TRY(dmalloc(d_out_sims, (size_t)Bmax * L1 * KEEP));
    BUILD_TRY(dmalloc(d_out_counts, (size_t)Bmax * L1));
    BUILD_TRY(dmalloc(d_status, Bmax));
    BUILD_TRY(hipHostMalloc((void **)&h_rows, (size_t)Bmax * 4));
    BUILD_TRY(hipHostMalloc((void **)&h_nodes, (size_t)Bmax * L1 * KEEP * 4));
    BUILD_TRY(hipHostMalloc((void **)&h_sims, (size_t)Bmax * L1 * KEEP * 4));
    BUILD_TRY(hipHostMalloc((void **)&h_counts, (size_t)Bmax * L1 * 4));
    BUILD_TRY(hipHostMalloc((void **)&h_status, (size_t)Bmax * 4));
#undef BUILD_TRY

    IndexDev dev = cos_make_index_dev(ix);

    u32 inserted = 0, batch_no = 0;
    const bool prof = getenv("COS_BUILD_PROFILE") != nullptr;
    double t_walk = 0, t_link = 0, t_up = 0;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    while (inserted < n) {
        const double t0 = now();
        batch_no++;
        const u32 bs = std::min({Bmax, std::max(1u, inserted / 4u), n - inserted});
        for (u32 b = 0; b < bs; b++) h_rows[b] = inserted + b;
        hipError_t e = hipMemcpyAsync(d_rows, h_rows, (size_t)bs * 4, hipMemcpyHostToDevice, st);
        WalkArgs wa;
        memset(&wa, 0, sizeof(wa));
        wa.qcodes = ix->d_codes;
        wa.qmags = ix->d_mags;
        wa.q_rows = d_rows;
        wa.self_ids = d_rows; // the new node's id is pre-inserted in the visited filter (vector_store.rs:807)
        if (dev.visited_mode == COS_VISITED_EXACT) {
            const int32_t vrc = vis_tab_prepare(vtab, ix, Bmax, ix->p.ef_construction, st, wa);
            if (vrc) { cleanup(); return vrc; }
        }
        wa.B = bs;
        wa.ef = ix->p.ef_construction;
        wa.keep = KEEP;
        wa.out_ids = d_out_ids;
        wa.out_sims = d_out_sims;
        wa.out_nodes = d_out_nodes;
        wa.out_counts = d_out_counts;
        wa.out_status = d_status;
        if (e == hipSuccess) e = launch_walk(ix->eng, dev, wa, st);
        if (e == hipSuccess) e = hipMemcpyAsync(h_nodes, d_out_nodes, (size_t)bs * L1 * KEEP * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(h_sims, d_out_sims, (size_t)bs * L1 * KEEP * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(h_counts, d_out_counts, (size_t)bs * L1 * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(h_status, d_status, (size_t)bs * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { cleanup(); HIP_TRY(e); }
        for (u32 b = 0; b < bs; b++)
            if (h_status[b] != COS_OK) {
                cleanup();
                return cos_fail(h_status[b], "vector %u cannot be indexed (zero norm -> DistanceError::CalculationError)", inserted + b);
            }

        const double t1 = now();
        t_walk += t1 - t0;
        // connect edges level by level, new nodes in id order (create_node_edges, vector_store.rs:923-936).  Levels are
        // independent graphs: level 0 (3/4 of the work) and the upper levels are linked by two host threads.
        const int32_t kmin = order_key(metric, metric_min(metric)), kmax = order_key(metric, metric_max(metric));
        auto link_levels = [&](u32 l_begin, u32 l_end) {
            for (u32 l = l_begin; l < l_end; l++) {
                LevelBuild &L = lb[l];
                L.dirty.clear();
                const u32 slot = Ltop - l;
                // the link is a chain of random row accesses (neighbour slots + keys of every candidate): software-prefetch the
                // rows of the batch nodes a few positions ahead while the current one is being connected
                auto prefetch_rows = [&](u32 b) {
                    if (b >= bs || max_level[inserted + b] < l) return;
                    const size_t base = ((size_t)b * L1 + slot) * KEEP;
                    const u32 zn = h_counts[(size_t)b * L1 + slot];
                    for (u32 i = 0; i < zn; i++) {
                        const size_t o = (size_t)h_nodes[base + i] * L.M;
                        for (u32 c = 0; c < L.M; c += 16) { // 64-byte lines
                            __builtin_prefetch(&L.nbr[o + c], 1, 1);
                            __builtin_prefetch(&L.key[o + c], 1, 1);
                        }
                        __builtin_prefetch(&L.low_key[h_nodes[base + i]], 1, 1);
                        __builtin_prefetch(&L.low_idx[h_nodes[base + i]], 1, 1);
                        __builtin_prefetch(&L.stamp[h_nodes[base + i]], 1, 1);
                    }
                };
                for (u32 b = 0; b < 3 && b < bs; b++) prefetch_rows(b);
                for (u32 b = 0; b < bs; b++) {
                    prefetch_rows(b + 3);
                    const u32 id = inserted + b;
                    if (max_level[id] < l) continue;
                    const u32 me = L.cursor++;
                    const size_t base = ((size_t)b * L1 + slot) * KEEP;
                    create_node_edges(L, metric, kmin, kmax, me, &h_nodes[base], &h_sims[base], h_counts[(size_t)b * L1 + slot], batch_no);
                }
            }
        };
        if (Ltop >= 1 && bs >= 64) {
            std::thread upper(link_levels, 1u, Ltop + 1);
            link_levels(0u, 1u);
            upper.join();
        } else
            link_levels(0u, Ltop + 1);
        const double tl1 = now();
        t_link += tl1 - t1;
        // scatter the dirtied adjacency rows back to HBM (node indices up; vector rows derived on the device)
        for (u32 l = 0; l <= Ltop; l++) {
            LevelBuild &L = lb[l];
            const u32 nd = (u32)L.dirty.size();
            if (nd == 0) continue;
            const size_t need = (size_t)nd * L.M + nd;
            if (need > pack_cap) { if (d_pack) (void)hipFree(d_pack); d_pack = nullptr; pack_cap = need * 2; e = dmalloc(d_pack, pack_cap); if (e != hipSuccess) { cleanup(); HIP_TRY(e); } }
            if (need > h_pack_cap) {
                if (h_pack) (void)hipHostFree(h_pack);
                h_pack = nullptr;
                h_pack_cap = need * 2;
                e = hipHostMalloc((void **)&h_pack, h_pack_cap * 4);
                if (e != hipSuccess) { cleanup(); HIP_TRY(e); }
            }
            memcpy(h_pack, L.dirty.data(), (size_t)nd * 4); // [row ids | packed rows]
            u32 *pk = h_pack + nd;
            for (u32 i = 0; i < nd; i++) memcpy(pk + (size_t)i * L.M, &L.nbr[(size_t)L.dirty[i] * L.M], (size_t)L.M * 4); // NONE == ROW_EMPTY
            e = hipMemcpyAsync(d_pack, h_pack, need * 4, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) {
                const u64 total = (u64)nd * L.M;
                hipLaunchKernelGGL(scatter_rows2_kernel, dim3((u32)((total + 255) / 256)), dim3(256), 0, st, ix->lv[l].d_adj_vec,
                                   l == 0 ? (u32 *)nullptr : ix->lv[l].d_adj_node, l == 0 ? (const u32 *)nullptr : (const u32 *)ix->lv[l].d_node_vec,
                                   d_pack, d_pack + nd, nd, L.M);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipStreamSynchronize(st); // staging buffers are reused by the next level
            if (e != hipSuccess) { cleanup(); HIP_TRY(e); }
        }
        t_up += now() - tl1;
        inserted += bs;
    }
    cleanup();
    if (prof) fprintf(stderr, "[cos_index_build] n=%u batches=%u walk+copy %.2fs link %.2fs upload %.2fs\n", n, batch_no, t_walk, t_link, t_up);

    // ---- host copy of the graph in the id format (cos_index_download_graph_level) ---------------
    for (u32 l = 0; l <= Ltop; l++) {
        LevelHost &H = ix->lv[l];
        LevelBuild &L = lb[l];
        H.node_ids = L.node_ids;
        H.nbr_ids.resize((size_t)L.n * L.M);
        for (size_t i = 0; i < (size_t)L.n * L.M; i++) H.nbr_ids[i] = L.nbr[i] == NONE ? COS_SLOT_EMPTY : L.node_ids[L.nbr[i]];
        H.host_valid = true;
    }
    return COS_OK;
}
