// engine.hip — host side of libcosdata_hip.so: the opaque index handle, uploads, per-stream
// workspaces and the C ABI of include/cosdata_hip.h.  No CPU compute path exists here: every
// numeric result is produced by the gfx950 kernels; without a device the calls fail loudly.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "engine_internal.h"

using namespace cosdev;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
int32_t cos_fail(int32_t code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

extern "C" const char *cos_last_error_string(void) { return g_err.c_str(); }

extern "C" int32_t cos_device_count(int32_t *out) {
    if (!out) return cos_fail(COS_ERR_INVALID, "null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *out = 0; return cos_fail(COS_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *out = n;
    return COS_OK;
}

static u32 pow2ceil(u32 v) { u32 p = 1; while (p < v) p <<= 1; return p; }

int32_t cos_set_device(const cos_index *ix) {
    HIP_TRY(hipSetDevice(ix->p.device));
    return COS_OK;
}

static bool graph_ready(const cos_index *ix) {
    if (!ix->have_vectors || !ix->have_root) return false;
    for (auto &l : ix->lv) if (l.n == 0) return false;
    return true;
}

IndexDev cos_make_index_dev(const cos_index *ix) {
    IndexDev d;
    memset(&d, 0, sizeof(d));
    d.codes = ix->d_codes;
    d.mags = ix->d_mags;
    d.raw = ix->d_raw;
    d.raw_mags = ix->d_raw_mags;
    d.row_stride = ix->row_stride;
    d.raw_stride = ix->p.dim;
    d.n = ix->n;
    d.dim = ix->p.dim;
    d.metric = ix->p.metric;
    d.storage = ix->p.storage;
    d.num_layers = ix->p.num_layers;
    d.shortlist = ix->p.shortlist_size;
    d.visited_mode = ix->p.visited_mode;
    d.nchunks = ix->nchunks;
    d.G = ix->G;
    d.id_base = ix->p.id_base;
    d.id_stride = ix->id_stride;
    d.mbits = ix->meta.d_mbits;
    d.mmags = ix->meta.d_mmags;
    d.mdim = ix->meta.mdim;
    for (u32 l = 0; l <= ix->p.num_layers; l++) {
        const LevelHost &h = ix->lv[l];
        d.lv[l].adj_vec = h.d_adj_vec;
        d.lv[l].adj_node = l == 0 ? h.d_adj_vec : h.d_adj_node;
        d.lv[l].node_vec = l == 0 ? nullptr : h.d_node_vec;
        d.lv[l].child = l == 0 ? nullptr : h.d_child;
        d.lv[l].n = h.n;
        d.lv[l].M = h.M;
        d.lv[l].root_idx = h.n ? h.n - 1 : 0;
        d.lv[l].adj_mag = (ix->adj_mag_valid && cosdev::tune_or(cosdev::TUNE_WALK_ADJ_MAG, 1) != 0) ? h.d_adj_mag : nullptr;
        d.lv[l].mag_stride = std::min(h.M, ix->p.shortlist_size);
    }
    return d;
}

extern "C" int32_t cos_index_create(const cos_params *p, cos_index **out) {
    if (!p || !out) return cos_fail(COS_ERR_INVALID, "null argument");
    *out = nullptr;
    if (p->struct_size != sizeof(cos_params) || p->abi_version != COS_ABI_VERSION)
        return cos_fail(COS_ERR_INVALID, "cos_params size/version mismatch (%u/%u, expected %zu/%u)", p->struct_size, p->abi_version,
                    sizeof(cos_params), COS_ABI_VERSION);
    if (p->dim == 0) return cos_fail(COS_ERR_INVALID, "dim == 0");
    auto pow2 = [](u32 v) { return v && !(v & (v - 1)); };
    if (!pow2(p->neighbors_count) || !pow2(p->level0_neighbors_count) || p->neighbors_count > 256 || p->level0_neighbors_count > 256)
        return cos_fail(COS_ERR_INVALID, "neighbors_count / level_0_neighbors_count must be powers of two <= 256 (PerformantFixedSet, fixedset.rs)");
    if (p->num_layers + 1 > (u32)MAX_LEVELS || (p->num_layers + 1) * KEEP_SEARCH > 2048)
        return cos_fail(COS_ERR_UNIMPLEMENTED, "num_layers > 15 not supported on the device");
    // (more than 64 scanned neighbour slots per node and ef > 1024 walk with walk_general_kernel: kernels_walk_general.hip)
    if (p->ef_search > cosdev::WALK_GENERAL_MAX_EF || p->ef_construction > cosdev::WALK_GENERAL_MAX_EF)
        return cos_fail(COS_ERR_UNIMPLEMENTED, "ef > %u not supported on the device (the candidates of a level live in one workgroup's LDS)", cosdev::WALK_GENERAL_MAX_EF);
    if (p->metric != COS_METRIC_COSINE && p->metric != COS_METRIC_DOT)
        return cos_fail(COS_ERR_UNIMPLEMENTED, "device walk implements cosine and dot-product metrics");
    int eng;
    u64 row_stride;
    u32 nchunks = 0, G = 1;
    switch (p->storage) {
    case COS_STORAGE_U8:
        eng = ENG_U8; row_stride = ((u64)p->dim + 15) & ~15ull; nchunks = (u32)(row_stride / 16); G = std::min(64u, pow2ceil(nchunks));
        // (more than 2048 dimensions: walk_general_kernel, any number of chunks per row)
        break;
    case COS_STORAGE_SUBBYTE:
        // one 16 B chunk holds every plane of 128 / 64 / 32 dims for binary / quaternary / octal
        if (p->resolution == 1) { eng = ENG_Q1; nchunks = (p->dim + 127) / 128; }
        else if (p->resolution == 2) { eng = ENG_Q2; nchunks = (p->dim + 63) / 64; }
        else if (p->resolution == 3) { eng = ENG_Q3; nchunks = (p->dim + 31) / 32; }
        else return cos_fail(COS_ERR_CALCULATION, "SubByte resolution %u: the reference's distance arms return CalculationError (cosine.rs:147-154)", p->resolution);
        row_stride = (u64)nchunks * 16; G = std::min(64u, pow2ceil(nchunks));
        // (more than 64 chunks per row: walk_general_kernel)
        break;
    case COS_STORAGE_F32:
        if (p->metric == COS_METRIC_DOT) return cos_fail(COS_ERR_STORAGE_MISMATCH, "DotProductDistance has no FullPrecisionFP arm (dotproduct.rs:20-64)");
        eng = ENG_F32; row_stride = ((u64)p->dim * 4 + 15) & ~15ull; G = 2;
        break;
    case COS_STORAGE_F16:
        eng = ENG_F16; row_stride = ((u64)p->dim * 2 + 15) & ~15ull; G = 1;
        break;
    default: return cos_fail(COS_ERR_UNIMPLEMENTED, "storage kind %u not supported on the device yet", p->storage);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return cos_fail(COS_ERR_NO_DEVICE, "no HIP device visible; the GPU path has no CPU fallback");
    if (p->device < 0 || p->device >= ndev) return cos_fail(COS_ERR_INVALID, "device %d out of range (%d visible)", p->device, ndev);
    cos_index *ix = new cos_index();
    // experiments (tuning.h): 0 = always / 4294967295 = never; 0 = one launch, arrival order; 0 = walks stay on the caller's stream
    ix->chain_min_B = (u32)cosdev::tune_or(cosdev::TUNE_WALK_CHAIN_MIN_B, ix->chain_min_B);
    ix->walk_order_min_B = (u32)cosdev::tune_or(cosdev::TUNE_WALK_ORDER_MIN_B, ix->walk_order_min_B);
    ix->walk_side_min_B = (u32)cosdev::tune_or(cosdev::TUNE_WALK_SIDE_MIN_B, ix->walk_side_min_B);
    ix->p = *p;
    ix->eng = eng;
    ix->row_stride = row_stride;
    ix->nchunks = nchunks;
    ix->G = G;
    ix->lv.resize(p->num_layers + 1);
    for (u32 l = 0; l <= p->num_layers; l++) ix->lv[l].M = l == 0 ? p->level0_neighbors_count : p->neighbors_count;
    if (hipSetDevice(p->device) != hipSuccess || hipStreamCreateWithFlags(&ix->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ix;
        return cos_fail(COS_ERR_HIP, "cannot create stream on device %d", p->device);
    }
    // the locality order deals sorted queries to the XCDs (kernels_order.hip): the count is the device's, not a constant; a part that
    // does not report it (or reports one XCD) simply gets no dealing — the order then only sorts
    int nx = 0;
    if (hipDeviceGetAttribute(&nx, hipDeviceAttributeNumberOfXccs, p->device) == hipSuccess && nx >= 1 && nx <= 64) ix->num_xcd = (u32)nx;
    else { (void)hipGetLastError(); ix->num_xcd = 1; }
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, p->device) == hipSuccess && ncu >= 1) ix->n_cus = (u32)ncu;
    else (void)hipGetLastError();
    *out = ix;
    return COS_OK;
}

// the pseudo-root component as the walk kernels see it: every level (0 included) maps node -> id / vector row / metadata row
IndexDev cos_make_meta_dev(const cos_index *ix) {
    IndexDev d = cos_make_index_dev(ix);
    for (u32 l = 0; l <= ix->p.num_layers; l++) {
        const LevelHost &h = ix->meta.lv[l];
        d.lv[l].adj_vec = h.d_adj_vec;
        d.lv[l].adj_node = h.d_adj_node;
        d.lv[l].node_vec = h.d_node_vec;
        d.lv[l].child = h.d_child;
        d.lv[l].node_id = h.d_node_id;
        d.lv[l].node_meta = h.d_node_meta;
        d.lv[l].n = h.n;
        d.lv[l].M = h.M;
        d.lv[l].root_idx = h.root_idx;
    }
    return d;
}

static void free_level(LevelHost &l) {
    if (l.d_adj_vec) (void)hipFree(l.d_adj_vec);
    if (l.d_adj_node) (void)hipFree(l.d_adj_node);
    if (l.d_node_vec) (void)hipFree(l.d_node_vec);
    if (l.d_child) (void)hipFree(l.d_child);
    if (l.d_node_id) (void)hipFree(l.d_node_id);
    if (l.d_node_meta) (void)hipFree(l.d_node_meta);
    if (l.d_adj_mag) (void)hipFree(l.d_adj_mag);
    l.d_adj_mag = nullptr;
    l.d_node_id = l.d_node_meta = nullptr;
    l.d_adj_vec = l.d_adj_node = l.d_node_vec = l.d_child = nullptr;
    l.n = 0;
    l.node_ids.clear();
    l.nbr_ids.clear();
    l.host_valid = false;
}
static void free_ws(Workspace *w) {
    void *ptrs[] = {w->fin_flags, w->stats2, w->tab, w->qsums, w->qdig, w->q_codes, w->q_mags, w->q_raw_mags, w->walk_ids, w->walk_counts, w->walk_sims, w->walk_status, w->stats,
                    w->rerank_rows, w->vis.bits, w->vis.log, w->d_queries, w->d_out_ids, w->d_out_counts, w->d_out_scores, w->d_out_status, w->f_buf};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    cosdev::walk_order_free(w->order);
    for (auto &e : w->ev) if (e) (void)hipEventDestroy(e);
    if (w->walk_done) (void)hipEventDestroy(w->walk_done);
    if (w->walk_fin) (void)hipEventDestroy(w->walk_fin);
    if (w->last_range) (void)hipEventDestroy(w->last_range);
    if (w->prep_done) (void)hipEventDestroy(w->prep_done);
    if (w->walk_stream) (void)hipStreamDestroy(w->walk_stream);
    delete w;
}

void cos_meta_free_levels(cos_index *ix) {
    for (auto &l : ix->meta.lv) free_level(l);
}

// the whole pseudo-root component: levels, node table, metadata dimensions, and the id stride it imposed on the base graph.
// After this the handle is a collection without a metadata schema again (cos_index_enable_metadata re-creates row n + 1).
static void reset_meta(cos_index *ix) {
    for (auto &l : ix->meta.lv) free_level(l);
    if (ix->meta.d_mbits) (void)hipFree(ix->meta.d_mbits);
    if (ix->meta.d_mmags) (void)hipFree(ix->meta.d_mmags);
    ix->meta.d_mbits = nullptr;
    ix->meta.d_mmags = nullptr;
    ix->meta.node_ids.clear();
    ix->meta.mdim = 0;
    ix->id_stride = 1;
}

static void free_pipe(HostPipe *hp) {
    void *ptrs[] = {hp->d_q, hp->d_ids, hp->d_counts, hp->d_scores, hp->d_status};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (hipEvent_t e : hp->ev_in) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : hp->ev_walk) if (e) (void)hipEventDestroy(e);
    hipStream_t sts[] = {hp->s[0], hp->s[1], hp->sc, hp->sf};
    for (hipStream_t st : sts) if (st) (void)hipStreamDestroy(st);
    delete hp;
}

extern "C" int32_t cos_index_destroy(cos_index *ix) {
    if (!ix) return COS_OK;
    (void)hipSetDevice(ix->p.device);
    (void)hipDeviceSynchronize();
    for (auto &kv : ix->ws) free_ws(kv.second);
    cos_flat_ws_release(ix);
    for (HostPipe *hp : ix->pipes_all) free_pipe(hp);
    cos_coalesce_release(ix);
    for (auto &l : ix->lv) free_level(l);
    for (auto &l : ix->meta.lv) free_level(l);
    if (ix->meta.d_mbits) (void)hipFree(ix->meta.d_mbits);
    if (ix->meta.d_mmags) (void)hipFree(ix->meta.d_mmags);
    if (ix->d_raw && !ix->raw_borrowed) (void)hipFree(ix->d_raw);
    if (ix->d_raw_mags) (void)hipFree(ix->d_raw_mags);
    if (ix->d_codes) (void)hipFree(ix->d_codes);
    if (ix->d_mags) (void)hipFree(ix->d_mags);
    for (u32 *p : ix->d_order_rank) if (p) (void)hipFree(p);
    for (auto &t : ix->table_sets) { // level-table operands (the current one is among them)
        if (t.tcodes) (void)hipFree(t.tcodes);
        if (t.tmags) (void)hipFree(t.tmags);
        if (t.tcsums) (void)hipFree(t.tcsums);
    }
    if (ix->own_stream) (void)hipStreamDestroy(ix->own_stream);
    delete ix;
    return COS_OK;
}

// ------------------------------------------------------------------------------------------------
// uploads
// ------------------------------------------------------------------------------------------------
extern "C" int32_t cos_index_upload_vectors(cos_index *ix, const float *raw, uint32_t n, uint32_t flags) {
    if (!ix || !raw || n == 0) return cos_fail(COS_ERR_INVALID, "null/empty vectors");
    if (n >= 0xFFFFFFF0u) return cos_fail(COS_ERR_INVALID, "too many vectors");
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    if (ix->d_raw && !ix->raw_borrowed) (void)hipFree(ix->d_raw);
    if (ix->d_raw_mags) (void)hipFree(ix->d_raw_mags);
    if (ix->d_codes) (void)hipFree(ix->d_codes);
    if (ix->d_mags) (void)hipFree(ix->d_mags);
    ix->d_raw = nullptr; ix->d_raw_mags = nullptr; ix->d_codes = nullptr; ix->d_mags = nullptr;
    ix->have_vectors = false;
    cos_flat_ws_release(ix); // cached sums of the stored codes, scan buffers sized for the old corpus
    ix->have_root = false;
    for (auto &l : ix->lv) free_level(l); // a graph refers to vector rows: new vectors invalidate it
    ix->order_rank_valid = false;
    ix->level_table_valid = false;
    ix->adj_mag_valid = false;
    ix->link.release();
    reset_meta(ix);                       // ... and so do the pseudo-root component, its node table and the id stride
    const u64 dim = ix->p.dim;
    struct Rollback { // a failed upload leaves the handle empty instead of half-populated
        cos_index *ix;
        bool armed = true;
        ~Rollback() {
            if (!armed) return;
            if (ix->d_raw && !ix->raw_borrowed) (void)hipFree(ix->d_raw);
            if (ix->d_raw_mags) (void)hipFree(ix->d_raw_mags);
            if (ix->d_codes) (void)hipFree(ix->d_codes);
            if (ix->d_mags) (void)hipFree(ix->d_mags);
            ix->d_raw = nullptr; ix->d_raw_mags = nullptr; ix->d_codes = nullptr; ix->d_mags = nullptr;
            ix->raw_borrowed = false;
        }
    } rollback{ix};
    if (flags & COS_UPLOAD_BORROW_DEVICE) {
        // the caller's producer (e.g. a torch kernel on another stream) may still be writing the buffer: our streams
        // are non-blocking, so drain the device once before the quantize kernel reads it
        HIP_TRY(hipDeviceSynchronize());
        ix->d_raw = const_cast<float *>(raw);
        ix->raw_borrowed = true;
    } else {
        HIP_TRY(hipMalloc(&ix->d_raw, (size_t)n * dim * 4));
        HIP_TRY(hipMemcpy(ix->d_raw, raw, (size_t)n * dim * 4, hipMemcpyHostToDevice));
        ix->raw_borrowed = false;
    }
    // rows [0, n) = vectors, row n = the root, row n + 1 = the vector shared by the pseudo nodes of a metadata collection
    HIP_TRY(hipMalloc(&ix->d_raw_mags, ((size_t)n + 2) * 4));
    HIP_TRY(hipMalloc(&ix->d_codes, ((size_t)n + 2) * ix->row_stride));
    HIP_TRY(hipMalloc(&ix->d_mags, ((size_t)n + 2) * 4));
    HIP_TRY(hipMemsetAsync(ix->d_codes + (size_t)n * ix->row_stride, 0, 2 * ix->row_stride, ix->own_stream));
    HIP_TRY(hipMemsetAsync(ix->d_mags + n, 0, 8, ix->own_stream));
    HIP_TRY(launch_quantize_rows(ix->eng, ix->d_raw, dim, n, ix->p.dim, ix->p.range_lo, ix->p.range_hi, ix->d_codes, ix->row_stride,
                                 ix->d_mags, ix->d_raw_mags, ix->own_stream));
    HIP_TRY(hipStreamSynchronize(ix->own_stream));
    rollback.armed = false;
    ix->n = n;
    ix->have_vectors = true;
    ix->have_root = false;
    return COS_OK;
}

extern "C" int32_t cos_index_set_root(cos_index *ix, const float *root_raw) {
    if (!ix || !root_raw) return cos_fail(COS_ERR_INVALID, "null argument");
    if (!ix->have_vectors) return cos_fail(COS_ERR_NOT_READY, "upload vectors before the root");
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    ix->root_raw.assign(root_raw, root_raw + ix->p.dim);
    DevBuf tmp; // [dim] staged root + 1 float for the unused raw-norm output
    HIP_TRY(tmp.alloc(((size_t)ix->p.dim + 1) * 4));
    float *d_tmp = tmp.as<float>(), *d_dummy = d_tmp + ix->p.dim;
    HIP_TRY(hipMemcpy(d_tmp, root_raw, (size_t)ix->p.dim * 4, hipMemcpyHostToDevice));
    HIP_TRY(launch_quantize_rows(ix->eng, d_tmp, ix->p.dim, 1, ix->p.dim, ix->p.range_lo, ix->p.range_hi,
                                 ix->d_codes + (size_t)ix->n * ix->row_stride, ix->row_stride, ix->d_mags + ix->n, d_dummy, ix->own_stream));
    HIP_TRY(hipStreamSynchronize(ix->own_stream));
    ix->have_root = true;
    // the root is a node of every level: the level-table operand holds a gathered COPY of its code row, norm and code sum, which a
    // second set_root on a live graph would leave stale (the row levels would use the new row) — the next search regathers
    ix->level_table_valid = false;
    ix->adj_mag_valid = false; // ... and the adjacency-side norms hold the old root's |v| wherever the root is a neighbour
    if (ix->link.valid && graph_ready(ix)) ix->link.release(); // the slot similarities of edges to the root were computed with the old root
    return COS_OK;
}

// id -> vector row (root = row n)
static inline u32 row_of(const cos_index *ix, u32 id) { return id == COS_ROOT_ID ? ix->n : id / ix->id_stride; }

static int32_t push_level_to_device(cos_index *ix, u32 level) {
    LevelHost &L = ix->lv[level];
    const u32 n = (u32)L.node_ids.size(), M = L.M;
    std::vector<u32> adj_vec((size_t)n * M), adj_node, node_vec, child;
    if (level > 0) { adj_node.resize((size_t)n * M); node_vec.resize(n); }
    for (u32 i = 0; i < n; i++) {
        if (level > 0) node_vec[i] = row_of(ix, L.node_ids[i]);
        for (u32 j = 0; j < M; j++) {
            u32 id = L.nbr_ids[(size_t)i * M + j];
            if (id == COS_SLOT_EMPTY) {
                adj_vec[(size_t)i * M + j] = ROW_EMPTY;
                if (level > 0) adj_node[(size_t)i * M + j] = ROW_EMPTY;
                continue;
            }
            auto it = std::lower_bound(L.node_ids.begin(), L.node_ids.end(), id);
            if (it == L.node_ids.end() || *it != id) return cos_fail(COS_ERR_INVALID, "level %u: neighbour id %u of node %u is not a node of this level", level, id, L.node_ids[i]);
            adj_vec[(size_t)i * M + j] = row_of(ix, id);
            if (level > 0) adj_node[(size_t)i * M + j] = (u32)(it - L.node_ids.begin());
        }
    }
    if (L.d_adj_vec) (void)hipFree(L.d_adj_vec);
    if (L.d_adj_node) (void)hipFree(L.d_adj_node);
    if (L.d_node_vec) (void)hipFree(L.d_node_vec);
    L.d_adj_vec = L.d_adj_node = L.d_node_vec = nullptr;
    HIP_TRY(hipMalloc(&L.d_adj_vec, adj_vec.size() * 4));
    HIP_TRY(hipMemcpy(L.d_adj_vec, adj_vec.data(), adj_vec.size() * 4, hipMemcpyHostToDevice));
    if (level > 0) {
        HIP_TRY(hipMalloc(&L.d_adj_node, adj_node.size() * 4));
        HIP_TRY(hipMemcpy(L.d_adj_node, adj_node.data(), adj_node.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(&L.d_node_vec, node_vec.size() * 4));
        HIP_TRY(hipMemcpy(L.d_node_vec, node_vec.data(), node_vec.size() * 4, hipMemcpyHostToDevice));
    }
    L.n = n;
    L.host_valid = true;
    ix->order_rank_valid = false; // the order key's table follows the graph (ensure_order_rank)
    ix->level_table_valid = false;
    ix->adj_mag_valid = false;
    ix->link.release();           // an uploaded level carries no slot similarities / lowest caches to continue from
    return COS_OK;
}

// child links: same id one level down (vector_store.rs:897-903)
static int32_t resolve_children(cos_index *ix, u32 level) {
    if (level == 0 || level > ix->p.num_layers) return COS_OK;
    LevelHost &L = ix->lv[level], &D = ix->lv[level - 1];
    if (L.node_ids.empty() || D.node_ids.empty()) return COS_OK;
    std::vector<u32> child(L.node_ids.size());
    for (size_t i = 0; i < L.node_ids.size(); i++) {
        auto it = std::lower_bound(D.node_ids.begin(), D.node_ids.end(), L.node_ids[i]);
        if (it == D.node_ids.end() || *it != L.node_ids[i]) return cos_fail(COS_ERR_INVALID, "node %u of level %u is missing on level %u", L.node_ids[i], level, level - 1);
        child[i] = (u32)(it - D.node_ids.begin());
    }
    if (L.d_child) (void)hipFree(L.d_child);
    L.d_child = nullptr;
    HIP_TRY(hipMalloc(&L.d_child, child.size() * 4));
    HIP_TRY(hipMemcpy(L.d_child, child.data(), child.size() * 4, hipMemcpyHostToDevice));
    return COS_OK;
}

extern "C" int32_t cos_index_upload_graph_level(cos_index *ix, uint32_t level, uint32_t n_nodes, const uint32_t *node_ids,
                                                const uint32_t *nbr_ids) {
    if (!ix || !node_ids || !nbr_ids) return cos_fail(COS_ERR_INVALID, "null argument");
    if (!ix->have_vectors) return cos_fail(COS_ERR_NOT_READY, "upload vectors before the graph");
    if (level > ix->p.num_layers || n_nodes == 0) return cos_fail(COS_ERR_INVALID, "bad level / empty level");
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    for (u32 i = 0; i < n_nodes; i++) {
        if (i && node_ids[i] <= node_ids[i - 1]) return cos_fail(COS_ERR_INVALID, "level %u: node_ids must be strictly ascending", level);
        if (node_ids[i] != COS_ROOT_ID && (node_ids[i] % ix->id_stride != 0 || node_ids[i] / ix->id_stride >= ix->n))
            return cos_fail(COS_ERR_INVALID, "level %u: node id %u is not the base id of a resident vector (ids are vector row x %u)", level, node_ids[i], ix->id_stride);
    }
    if (node_ids[n_nodes - 1] != COS_ROOT_ID) return cos_fail(COS_ERR_INVALID, "level %u: the root (0xFFFFFFFF) must be the last node", level);
    if (level == 0 && n_nodes != ix->n + 1) return cos_fail(COS_ERR_INVALID, "level 0 must hold every vector plus the root (%u nodes, expected %u)", n_nodes, ix->n + 1);
    LevelHost &L = ix->lv[level];
    L.node_ids.assign(node_ids, node_ids + n_nodes);
    L.nbr_ids.assign(nbr_ids, nbr_ids + (size_t)n_nodes * L.M);
    rc = push_level_to_device(ix, level);
    if (rc) { free_level(L); return rc; }
    rc = resolve_children(ix, level);
    if (rc == COS_OK) rc = resolve_children(ix, level + 1);
    if (rc == COS_OK && graph_ready(ix)) rc = cos_prepare_walk_plans(ix); // the last level of a graph: its order / level tables now, not under a search
    return rc;
}

extern "C" int32_t cos_index_level_count(const cos_index *ix, uint32_t level, uint32_t *n_nodes) {
    if (!ix || !n_nodes || level > ix->p.num_layers) return cos_fail(COS_ERR_INVALID, "bad argument");
    *n_nodes = ix->lv[level].n;
    return COS_OK;
}

// A level built on the device has no id-format host copy until somebody asks for one: node indices -> internal ids.
static int32_t materialize_host_level(cos_index *ix, u32 level) {
    LevelHost &L = ix->lv[level];
    if (L.host_valid) return COS_OK;
    if (L.n == 0 || L.node_ids.size() != L.n || !L.d_adj_vec) return cos_fail(COS_ERR_NOT_READY, "level %u not resident", level);
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    std::vector<u32> adj((size_t)L.n * L.M);
    HIP_TRY(hipMemcpy(adj.data(), level == 0 ? L.d_adj_vec : L.d_adj_node, adj.size() * 4, hipMemcpyDeviceToHost));
    L.nbr_ids.resize(adj.size());
    for (size_t i = 0; i < adj.size(); i++) L.nbr_ids[i] = adj[i] == ROW_EMPTY ? COS_SLOT_EMPTY : L.node_ids[adj[i]]; // level 0: node index = vector row
    L.host_valid = true;
    return COS_OK;
}

extern "C" int32_t cos_index_download_graph_level(const cos_index *ix_, uint32_t level, uint32_t *node_ids, uint32_t *nbr_ids) {
    cos_index *ix = const_cast<cos_index *>(ix_);
    if (!ix || level > ix->p.num_layers) return cos_fail(COS_ERR_INVALID, "bad argument");
    {
        std::lock_guard<std::mutex> g(ix->mu);
        const int32_t rc = materialize_host_level(ix, level);
        if (rc) return rc;
    }
    const LevelHost &L = ix->lv[level];
    if (node_ids) memcpy(node_ids, L.node_ids.data(), L.node_ids.size() * 4);
    if (nbr_ids) memcpy(nbr_ids, L.nbr_ids.data(), L.nbr_ids.size() * 4);
    return COS_OK;
}

extern "C" size_t cos_code_bytes(uint32_t storage, uint32_t resolution, uint32_t dim) {
    switch (storage) {
    case COS_STORAGE_U8: return dim;
    case COS_STORAGE_SUBBYTE: return (size_t)resolution * ((dim + 7) / 8);
    case COS_STORAGE_F16: return (size_t)dim * 2;
    case COS_STORAGE_F32: return (size_t)dim * 4;
    default: return 0;
    }
}

// device layout -> reference layout for one row
static void row_to_reference_layout(int eng, u32 dim, const uint8_t *dev_row, uint8_t *ref_row) {
    if (eng == ENG_U8) memcpy(ref_row, dev_row, dim);
    else if (eng == ENG_F32) memcpy(ref_row, dev_row, (size_t)dim * 4);
    else if (eng == ENG_F16) memcpy(ref_row, dev_row, (size_t)dim * 2);
    else if (eng == ENG_Q1) memcpy(ref_row, dev_row, (dim + 7) / 8); // one plane, already contiguous
    else if (eng == ENG_Q2) { // [chunk][plane][8 B] -> plane-major
        const u32 pb = (dim + 7) / 8;
        for (u32 p = 0; p < 2; p++)
            for (u32 b = 0; b < pb; b++) ref_row[(size_t)p * pb + b] = dev_row[(size_t)(b / 8) * 16 + p * 8 + (b % 8)];
    } else { // Q3: [chunk][plane][4 B] (+4 B pad) -> plane-major
        const u32 pb = (dim + 7) / 8;
        for (u32 p = 0; p < 3; p++)
            for (u32 b = 0; b < pb; b++) ref_row[(size_t)p * pb + b] = dev_row[(size_t)(b / 4) * 16 + p * 4 + (b % 4)];
    }
}

// reference layout -> device layout for one row (inverse of row_to_reference_layout); dev_row is zero-filled by the caller
static void row_to_device_layout(int eng, u32 dim, const uint8_t *ref_row, uint8_t *dev_row) {
    if (eng == ENG_U8) memcpy(dev_row, ref_row, dim);
    else if (eng == ENG_F32) memcpy(dev_row, ref_row, (size_t)dim * 4);
    else if (eng == ENG_F16) memcpy(dev_row, ref_row, (size_t)dim * 2);
    else if (eng == ENG_Q1) memcpy(dev_row, ref_row, (dim + 7) / 8);
    else if (eng == ENG_Q2) {
        const u32 pb = (dim + 7) / 8;
        for (u32 p = 0; p < 2; p++)
            for (u32 b = 0; b < pb; b++) dev_row[(size_t)(b / 8) * 16 + p * 8 + (b % 8)] = ref_row[(size_t)p * pb + b];
    } else {
        const u32 pb = (dim + 7) / 8;
        for (u32 p = 0; p < 3; p++)
            for (u32 b = 0; b < pb; b++) dev_row[(size_t)(b / 4) * 16 + p * 4 + (b % 4)] = ref_row[(size_t)p * pb + b];
    }
}

// The root as the reference STORED it (a quantized Storage + mag, read from prop.data by ref_index_reader.hip): written
// straight into row N.  The raw f32 root is not known on this path (cos_index_download_root returns zeros); the walk only
// ever reads the root's code and norm, and the rerank never sees the root (common.rs:397).
int32_t cos_set_root_code(cos_index *ix, const uint8_t *ref_code, float mag) {
    if (!ix->have_vectors) return cos_fail(COS_ERR_NOT_READY, "upload vectors before the root");
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    std::vector<uint8_t> dev(ix->row_stride, 0);
    row_to_device_layout(ix->eng, ix->p.dim, ref_code, dev.data());
    if (ix->eng == ENG_U8) {
        // u8 storage: Storage::UnsignedByte::mag IS sqrt(sum of squares of the bytes as f32) (scalar.rs:25-26).  The walk's unscaled
        // quotient (device_common.h div_rn_unscaled) relies on norms of that form (0, or 1 <= mag < 2^28): a stored value that is not
        // finite or outside [1, 255 sqrt(dim)] — a damaged file — is replaced by the norm recomputed from the code row, so that
        // every kernel (throughput, latency, single-distance) divides by the same, valid number.
        const float hi = 255.0f * sqrtf((float)ix->p.dim) * 1.0001f;
        if (!(mag == 0.0f || (mag >= 1.0f && mag <= hi))) { // (NaN fails every comparison)
            uint32_t ss = 0;
            for (u32 i = 0; i < ix->p.dim; i++) ss += (uint32_t)ref_code[i] * (uint32_t)ref_code[i];
            mag = sqrtf((float)ss);
        }
    }
    HIP_TRY(hipMemcpy(ix->d_codes + (size_t)ix->n * ix->row_stride, dev.data(), dev.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ix->d_mags + ix->n, &mag, 4, hipMemcpyHostToDevice));
    ix->root_raw.assign(ix->p.dim, 0.0f);
    ix->have_root = true;
    ix->level_table_valid = false;
    ix->adj_mag_valid = false;
    return COS_OK;
}

int32_t cos_push_level_ids(cos_index *ix, u32 level, std::vector<u32> &&node_ids, std::vector<u32> &&nbr_ids) {
    return cos_index_upload_graph_level(ix, level, (u32)node_ids.size(), node_ids.data(), nbr_ids.data());
}

extern "C" int32_t cos_index_download_codes(const cos_index *ix, void *codes, float *mags) {
    if (!ix || !ix->have_vectors) return cos_fail(COS_ERR_NOT_READY, "no vectors resident");
    HIP_TRY(hipSetDevice(ix->p.device));
    const size_t rows = (size_t)ix->n + 1;
    if (codes) {
        std::vector<uint8_t> dev(rows * ix->row_stride);
        HIP_TRY(hipMemcpy(dev.data(), ix->d_codes, dev.size(), hipMemcpyDeviceToHost));
        const size_t cb = cos_code_bytes(ix->p.storage, ix->p.resolution, ix->p.dim);
        for (size_t r = 0; r < rows; r++) row_to_reference_layout(ix->eng, ix->p.dim, dev.data() + r * ix->row_stride, (uint8_t *)codes + r * cb);
    }
    if (mags) HIP_TRY(hipMemcpy(mags, ix->d_mags, rows * 4, hipMemcpyDeviceToHost));
    return COS_OK;
}

extern "C" int32_t cos_index_download_root(const cos_index *ix, float *root_raw) {
    if (!ix || !root_raw) return cos_fail(COS_ERR_INVALID, "null argument");
    if (!ix->have_root) return cos_fail(COS_ERR_NOT_READY, "no root");
    memcpy(root_raw, ix->root_raw.data(), (size_t)ix->p.dim * 4);
    return COS_OK;
}

static int32_t ensure_level_table(cos_index *ix);
static u32 walk_table_min_B(const cos_index *ix);
extern "C" int32_t cos_index_set_ef_search(cos_index *ix, uint32_t ef) {
    if (!ix) return cos_fail(COS_ERR_INVALID, "null index");
    if (ef > cosdev::WALK_GENERAL_MAX_EF) return cos_fail(COS_ERR_UNIMPLEMENTED, "ef > %u not supported on the device", cosdev::WALK_GENERAL_MAX_EF);
    std::lock_guard<std::mutex> g(ix->mu); // searches snapshot ef / visited mode under the same lock (run_search); the graph state is read under it too
    const bool live = graph_ready(ix);
    if (live)
        if (int32_t rc = cos_set_device(ix)) return rc; // (nothing changed yet)
    ix->p.ef_search = ef;
    // the automatic level-table rule depends on ef: the operand of the new rule is gathered HERE, by the caller that changes the
    // knob, not inside the next search (where the gather and its device synchronisation ran under ix->mu in front of every
    // concurrent searcher); operands are cached per (graph, rule), so going back and forth costs nothing the second time.
    // Best effort: the table is an accelerator, not a result — a gather that fails (out of memory is already absorbed inside
    // ensure_level_table) leaves the new ef in force, the next search walks without a table or retries the gather, and the
    // setter reports success: an error here would say "ef unchanged" while it has changed.
    if (live && walk_table_min_B(ix)) (void)ensure_level_table(ix);
    return COS_OK;
}
extern "C" int32_t cos_index_set_visited_mode(cos_index *ix, uint32_t mode) {
    if (!ix || mode > 1) return cos_fail(COS_ERR_INVALID, "bad visited mode");
    std::lock_guard<std::mutex> g(ix->mu);
    ix->p.visited_mode = mode;
    return COS_OK;
}
extern "C" int32_t cos_index_set_latency_mode(cos_index *ix, uint32_t max_queries) {
    if (!ix) return cos_fail(COS_ERR_INVALID, "null");
    std::lock_guard<std::mutex> g(ix->mu);
    ix->lat_max_B = max_queries;
    return COS_OK;
}

extern "C" int32_t cos_index_set_latency_waves(cos_index *ix, uint32_t max_queries) {
    if (!ix) return cos_fail(COS_ERR_INVALID, "null");
    std::lock_guard<std::mutex> g(ix->mu);
    ix->lat4_max_B = max_queries;
    return COS_OK;
}
extern "C" int32_t cos_index_set_walk_order(cos_index *ix, uint32_t min_queries) {
    if (!ix) return cos_fail(COS_ERR_INVALID, "null");
    std::lock_guard<std::mutex> g(ix->mu);
    ix->walk_order_min_B = min_queries;
    return COS_OK;
}
extern "C" int32_t cos_index_enable_timing(cos_index *ix, int32_t on) {
    if (!ix) return cos_fail(COS_ERR_INVALID, "null");
    std::lock_guard<std::mutex> g(ix->mu);
    if (on && !ix->timing) for (auto &kv : ix->ws) kv.second->ev_count = 0; // a new measurement window
    ix->timing = on != 0;
    return COS_OK;
}

// ------------------------------------------------------------------------------------------------
// workspaces (one per stream; same-stream work is ordered, so reuse is safe)
// ------------------------------------------------------------------------------------------------
template <typename T>
static hipError_t regrow(T *&p, size_t count) {
    if (p) (void)hipFree(p);
    p = nullptr;
    return hipMalloc((void **)&p, count * sizeof(T));
}

int32_t vis_tab_prepare(VisTab &vt, const cos_index *ix, u32 B, u32 ef, hipStream_t st, WalkArgs &wa) {
    const u32 M = std::max(ix->p.level0_neighbors_count, ix->p.neighbors_count);
    // a level inserts its entry node plus at most M neighbours per pop, and pops at most ef times
    const u64 log_cap = (u64)ef * M + 2;
    u32 max_nodes = 0;
    for (const LevelHost &l : ix->lv) max_nodes = std::max(max_nodes, l.n);
    const u32 words = (max_nodes + 31) / 32 + 1;
    if (log_cap > 0xFFFFFFFFull) return cos_fail(COS_ERR_INVALID, "ef too large");
    const size_t need_bits = (size_t)B * words, need_log = (size_t)B * log_cap;
    if (need_bits > vt.bits_cap || need_log > vt.log_cap) {
        HIP_TRY(hipStreamSynchronize(st));
        if (need_bits > vt.bits_cap) {
            if (vt.bits) HIP_TRY(hipFree(vt.bits));
            vt.bits = nullptr;
            vt.bits_cap = 0;
            HIP_TRY(hipMalloc((void **)&vt.bits, need_bits * 4));
            vt.bits_cap = need_bits;
        }
        if (need_log > vt.log_cap) {
            if (vt.log) HIP_TRY(hipFree(vt.log));
            vt.log = nullptr;
            vt.log_cap = 0;
            HIP_TRY(hipMalloc((void **)&vt.log, need_log * 4));
            vt.log_cap = need_log;
        }
    }
    // the per-query stride may change (other B / graph), which is harmless: the whole allocation is all-zero between launches
    if (!vt.zeroed || vt.zeroed_cap != vt.bits_cap) {
        HIP_TRY(hipMemsetAsync(vt.bits, 0, vt.bits_cap * 4, st));
        vt.zeroed = true;
        vt.zeroed_cap = vt.bits_cap;
    }
    wa.vis_bits = vt.bits;
    wa.vis_log = vt.log;
    wa.vis_words_per_query = words;
    wa.vis_log_cap = (u32)log_cap;
    return COS_OK;
}

// the workspace registered under `key` (a caller's stream, or a HostPipe slot), grown to B queries; `st` is the stream the
// workspace's previous launches ran on (drained before a buffer is replaced)
// The tables behind the order keys of big launches (WalkArgs::order_rank): every node of a key level -> its position in a
// depth-first order of that level's graph.  Consecutive positions are graph neighbours or a short backtrack apart, so queries
// whose best key-level nodes have close positions walk the same region of the levels below — a one-dimensional locality order
// that needs nothing but the graph (the vectors' geometry is in its edges).
// Key levels (the walk is cut AFTER each of them): by default ONE, the lowest level whose code rows take at most 64 MB — a
// quarter of the 256 MB memory-side cache, so the levels walked in arrival order stay cache-resident whatever the order, and the
// key is as fine as that allows (c2: level 2, 62 500 nodes x 768 B = 48 MB, levels 1 and 0 walked in order; a 12.5M x 1024
// shard: level 4, 48 829 nodes).  Every further cut costs the tail of one more launch (~0.3 ms per 32768 queries) and measured
// slower: profiles/archive/r03_locality_probe_split_levels.jsonl (cuts after 1 | 2 | 3 | 4 | 3,1 | 4,1 | 5,1 | 4,2,1 | 5,3,1).
// COS_WALK_SPLIT=a,b,.. overrides (descending levels; "0" = no split).
// Built on the host from the adjacency, once per graph: milliseconds for levels this small (<= 2^20 nodes).  Caller holds ix->mu.
static int32_t ensure_order_rank(cos_index *ix) {
    if (ix->order_rank_valid) return COS_OK;
    for (u32 l = 0; l < cosdev::MAX_LEVELS; l++) {
        if (ix->d_order_rank[l]) (void)hipFree(ix->d_order_rank[l]);
        ix->d_order_rank[l] = nullptr;
        ix->order_rank_n[l] = 0;
    }
    ix->order_levels.clear();
    const u32 Ltop = ix->p.num_layers;
    auto usable = [&](u32 l) { return l >= 1 && l <= Ltop && ix->lv[l].n > 1 && ix->lv[l].n <= (1u << 20) && ix->lv[l].d_adj_node; };
    std::vector<u32> want;
    if (const long long mask = cosdev::tune(cosdev::TUNE_WALK_SPLIT_LEVELS); mask != cosdev::TUNE_UNSET) { // experiments: bit l = cut after level l
        for (u32 l = Ltop; l >= 1; l--)
            if ((mask >> l) & 1) want.push_back(l);
    } else {
        for (u32 l = 1; l <= Ltop; l++)
            if (usable(l) && (size_t)ix->lv[l].n * ix->row_stride <= ((size_t)64 << 20)) { want.push_back(l); break; }
    }
    u32 prev = Ltop + 1;
    for (u32 l : want) { // descending, distinct, usable; a level that is not is skipped, not an error
        if (!usable(l) || l >= prev) continue;
        ix->order_levels.push_back(l);
        prev = l;
    }
    if (!ix->order_levels.empty()) HIP_TRY(hipDeviceSynchronize()); // a build or an upload on another stream may still be writing the adjacency
    for (u32 kl : ix->order_levels) {
        const LevelHost &H = ix->lv[kl];
        const u32 n = H.n, M = H.M;
        std::vector<u32> adj((size_t)n * M), rank(n, 0u), stack;
        HIP_TRY(hipMemcpy(adj.data(), H.d_adj_node, adj.size() * 4, hipMemcpyDeviceToHost));
        std::vector<uint8_t> seen(n, 0);
        u32 next = 0;
        auto dfs_from = [&](u32 start) {
            stack.push_back(start);
            while (!stack.empty()) {
                const u32 v = stack.back();
                stack.pop_back();
                if (seen[v]) continue;
                seen[v] = 1;
                rank[v] = next++;
                const u32 *row = &adj[(size_t)v * M];
                for (u32 j = M; j-- > 0;) { // pushed last = visited first: slot order
                    const u32 u = row[j];
                    if (u != cosdev::ROW_EMPTY && u < n && !seen[u]) stack.push_back(u);
                }
            }
        };
        dfs_from(n - 1); // the root (last node of every level): where every walk of the level starts
        for (u32 v = 0; v < n; v++) if (!seen[v]) dfs_from(v);
        HIP_TRY(hipMalloc((void **)&ix->d_order_rank[kl], (size_t)n * 4));
        HIP_TRY(hipMemcpy(ix->d_order_rank[kl], rank.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        ix->order_rank_n[kl] = n;
    }
    ix->order_rank_valid = true;
    return COS_OK;
}

// The level table's per-graph operand (WalkArgs::tab, cosdata_hip.h cos_index_set_walk_table): which levels it covers and the code
// rows, norms and code sums of their nodes, level by level from the top, gathered on the device.  Caller holds ix->mu.
static void clear_current_table(cos_index *ix) {
    ix->d_tcodes = nullptr;
    ix->d_tmags = nullptr;
    ix->d_tcsums = nullptr;
    ix->table_cols = 0;
    ix->table_level_min = 0;
    ix->table_stride = 0;
}
// every operand of the handle (the graph changed, or the handle dies): exclusive entry points only
static void free_level_tables(cos_index *ix) {
    for (auto &t : ix->table_sets) {
        if (t.tcodes) (void)hipFree(t.tcodes);
        if (t.tmags) (void)hipFree(t.tmags);
        if (t.tcsums) (void)hipFree(t.tcsums);
    }
    ix->table_sets.clear();
    clear_current_table(ix);
}
static u32 walk_table_max_cols(const cos_index *ix) { // COS_WALK_TABLE_AUTO = sized from ef_search and neighbors_count (ensure_level_table)
    const long long env = cosdev::tune_or(cosdev::TUNE_WALK_TABLE_COLS, -1);
    return env >= 0 ? (u32)std::min<long long>(env, 1ll << 20) : ix->walk_table_max_cols;
}
static u32 walk_table_min_B(const cos_index *ix) {
    const long long env = cosdev::tune_or(cosdev::TUNE_WALK_TABLE_MIN_B, -1);
    return env >= 0 ? (u32)std::min<long long>(env, 0xFFFFFFFFll) : ix->walk_table_min_B;
}
static int32_t ensure_level_table(cos_index *ix) {
    const u32 max_cols = walk_table_max_cols(ix);
    // Which levels pay.  A table column costs one GEMM column per launch (~2.6 ps per query and column at K = 1024), a level walked
    // from rows costs ~0.11 ns per evaluation, and a level's walk evaluates 0.2-0.4 x ef x M rows per query: a level of n nodes pays
    // while n < ~40 x that.  Measured on both ends — 1M x 768 at ef 64, M 32: level 4 (3 917 nodes) pays, level 3 (15 570) does not;
    // one 12.5M x 1024 shard at ef 128, M 64: level 5 (12 225) +24 % QPS, level 4 (48 870) another +17 %, both far above a fixed
    // 8 192 columns (profiles/r04_c4_table_cols_probe.jsonl) — the automatic rule is n_level <= c x ef_search x neighbors_count, level
    // by level from the top; an explicit max_cols caps the columns of all table levels together instead.  Round 4-5: c = 8 up to ef 64
    // and 6 above: the wider pools of ef > 64 make a walk spend its time ranking and inserting, so a row it does not fetch saves less
    // (1M x 768, M 32: at ef 64 level 3 = 7.6 x ef x M pays, 7.53 -> 7.13 ms per 32 768 queries; at ef 256 level 2 = 7.7 x ef x M
    // costs 5.4 ms of GEMM and saves nothing, level 3 = 1.9 x pays; profiles/r04_table_level_rule_probe.jsonl).
    const bool automatic = max_cols == COS_WALK_TABLE_AUTO;
    // Round 6 (profiles/r06_rule_probe_*.jsonl): what decides is how many rows an expansion still evaluates, i.e. how far the level's
    // filter (64 x M bits) is from saturation after ef pops — ef relative to M, not ef alone.  12.5M x 1024 with M 64: level 4 (48 870
    // nodes) at ef 80 / 96 / 112 (9.5 / 8.0 / 6.8 x ef x M) +25 % / +25 % / +25 % QPS, 14 evaluations per expansion; 1M x 768 with M 32:
    // level 2 (62 869 nodes) at ef 128 / 256 (15.3 / 7.7 x) -20 % / -15 %, 3-5 evaluations per expansion.  c = 10 up to ef = 1.5 M, 8 up
    // to 2 M, 6 above (M 32: the old rule — 8 up to ef 64, 6 above — except c = 10 up to ef 48).
    const u32 efs = ix->p.ef_search, Mup = ix->p.neighbors_count;
    const u64 c_rule = (u64)cosdev::tune_or(cosdev::TUNE_WALK_TABLE_RULE_C, 2u * efs <= 3u * Mup ? 10 : (efs <= 2u * Mup ? 8 : 6));
    const u64 per_level = automatic ? std::min<u64>(c_rule * ix->p.ef_search * ix->p.neighbors_count, 1u << 20) : (u64)(1u << 20);
    const u64 total_cap = automatic ? (u64)(1u << 20) : (u64)max_cols;
    const u64 key = automatic ? (0x8000000000000000ull | per_level) : (u64)max_cols;
    if (ix->level_table_valid && ix->table_built_for_key == key) return COS_OK;
    if (!ix->level_table_valid) { // a new graph: the old graph's operands go (graph changes are exclusive: no search is in flight)
        HIP_TRY(hipDeviceSynchronize());
        free_level_tables(ix);
    }
    // (from here on every path leaves a definite state for this (graph, key): a table, or "none" — which is also what an operand
    // that could not be allocated leaves, see below)
    ix->level_table_valid = true;
    ix->table_built_for_key = key;
    clear_current_table(ix);
    for (const auto &t : ix->table_sets)
        if (t.key == key) { // built before for this graph (ef_search went back and forth)
            ix->table_level_min = t.level_min;
            ix->table_cols = t.cols;
            ix->table_stride = t.stride;
            memcpy(ix->table_col0, t.col0, sizeof(t.col0));
            ix->d_tcodes = t.tcodes;
            ix->d_tmags = t.tmags;
            ix->d_tcsums = t.tcsums;
            return COS_OK;
        }
    if (!cosdev::level_table_eng_supported(ix->eng, ix->row_stride) || max_cols == 0) return COS_OK; // u8 codes; quaternary codes of 128 ... 1024 dims
    const u32 Ltop = ix->p.num_layers;
    u32 cols = 0, lmin = 0;
    for (u32 l = Ltop; l >= 1; l--) { // level 0 (every vector) never takes part
        if (!ix->lv[l].d_node_vec || ix->lv[l].n == 0 || ix->lv[l].n > per_level || (u64)cols + ix->lv[l].n > total_cap) break;
        cols += ix->lv[l].n;
        lmin = l;
    }
    if (lmin == 0) return COS_OK;
    HIP_TRY(hipDeviceSynchronize()); // a build or an upload on another stream may still be writing the levels
    uint8_t *tcodes = nullptr;
    float *tmags = nullptr;
    u32 *tcsums = nullptr;
    auto build = [&]() -> hipError_t { // (a failure leaves the handle without a table for this key — same results — and frees what it had allocated)
        hipError_t e = hipMalloc((void **)&tcodes, (size_t)cols * ix->row_stride);
        if (e == hipSuccess) e = hipMalloc((void **)&tmags, (size_t)cols * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&tcsums, (size_t)cols * 4);
        u32 c0 = 0;
        for (u32 l = Ltop; l >= lmin && e == hipSuccess; l--) {
            ix->table_col0[l] = c0;
            e = cosdev::launch_level_table_gather(ix->d_codes, ix->d_mags, ix->row_stride, ix->lv[l].d_node_vec, ix->lv[l].n, c0, tcodes, tmags, nullptr);
            c0 += ix->lv[l].n;
        }
        if (e == hipSuccess) e = cosdev::launch_code_sums(tcodes, ix->row_stride, cols, tcsums, nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        return e;
    };
    if (const hipError_t e = build(); e != hipSuccess) {
        (void)hipDeviceSynchronize();
        if (tcodes) (void)hipFree(tcodes);
        if (tmags) (void)hipFree(tmags);
        if (tcsums) (void)hipFree(tcsums);
        if (e == hipErrorOutOfMemory) { // the table is an optimisation: short of HBM the walk reads rows (same results)
            (void)hipGetLastError();
            return COS_OK;
        }
        ix->level_table_valid = false; // anything else is a real failure: reported, and retried by the next search
        HIP_TRY(e);
    }
    ix->d_tcodes = tcodes;
    ix->d_tmags = tmags;
    ix->d_tcsums = tcsums;
    ix->table_cols = cols;
    ix->table_level_min = lmin;
    ix->table_stride = ((u64)cols + 31) / 32 * 32;
    cos_index::TableSet t;
    t.key = key;
    t.level_min = lmin;
    t.cols = cols;
    t.stride = ix->table_stride;
    memcpy(t.col0, ix->table_col0, sizeof(t.col0));
    t.tcodes = ix->d_tcodes;
    t.tmags = ix->d_tmags;
    t.tcsums = ix->d_tcsums;
    ix->table_sets.push_back(t);
    return COS_OK;
}

extern "C" int32_t cos_index_set_walk_table(cos_index *ix, uint32_t max_cols, uint32_t min_queries) {
    if (!ix) return cos_fail(COS_ERR_INVALID, "null argument");
    if (max_cols > (1u << 20) && max_cols != COS_WALK_TABLE_AUTO) return cos_fail(COS_ERR_INVALID, "max_cols must be <= 2^20 (or COS_WALK_TABLE_AUTO)");
    std::lock_guard<std::mutex> g(ix->mu);
    ix->walk_table_max_cols = max_cols;
    ix->walk_table_min_B = min_queries;
    return COS_OK;
}

extern "C" int32_t cos_index_walk_table_info(cos_index *ix, uint32_t *out_level_min, uint32_t *out_cols) {
    if (!ix || !out_level_min || !out_cols) return cos_fail(COS_ERR_INVALID, "null argument");
    *out_level_min = *out_cols = 0;
    if (!graph_ready(ix)) return cos_fail(COS_ERR_NOT_READY, "index needs vectors, root and every graph level");
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(ix->mu);
    if (walk_table_min_B(ix) == 0) return COS_OK;
    if ((rc = ensure_level_table(ix))) return rc;
    *out_level_min = ix->table_level_min;
    *out_cols = ix->table_cols;
    return COS_OK;
}

// The per-graph tables of big launches (locality order ranks, level-table operand) built when a graph is committed — the end of
// cos_index_build, the upload of a graph's last level (which is also how cos_index_load_reference_dir commits) — instead of inside the first big search, where
// the host-side DFS and the device synchronisations ran under ix->mu and held up every concurrent searcher.  The lazy calls in
// get_workspace stay as a fall-back (knobs changed after the commit).
// LevelDev::adj_mag of every level: the norm of each scanned neighbour slot next to the adjacency.  Caller holds ix->mu; graph
// changes are exclusive (no search in flight).  Out of memory leaves the walk on its mags[] gathers (same results).
static int32_t ensure_adj_mags(cos_index *ix) {
    if (ix->adj_mag_valid) return COS_OK;
    if (!graph_ready(ix) || ix->meta.mdim != 0u) return COS_OK; // (collections with a metadata schema keep the gathers: their walks are walk_meta_kernel's)
    if (std::min(ix->p.neighbors_count, ix->p.shortlist_size) > 64u || std::min(ix->p.level0_neighbors_count, ix->p.shortlist_size) > 64u)
        return COS_OK; // (walk_general_kernel's indexes: it gathers the norms)
    HIP_TRY(hipDeviceSynchronize()); // a build or an upload on another stream may still be writing the adjacency
    for (u32 l = 0; l <= ix->p.num_layers; l++) {
        LevelHost &H = ix->lv[l];
        if (H.d_adj_mag) (void)hipFree(H.d_adj_mag);
        H.d_adj_mag = nullptr;
    }
    for (u32 l = 0; l <= ix->p.num_layers; l++) {
        LevelHost &H = ix->lv[l];
        const u32 slots = std::min(H.M, ix->p.shortlist_size);
        const hipError_t e = hipMalloc((void **)&H.d_adj_mag, (size_t)H.n * slots * 4);
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            for (u32 k = 0; k <= l; k++) { if (ix->lv[k].d_adj_mag) (void)hipFree(ix->lv[k].d_adj_mag); ix->lv[k].d_adj_mag = nullptr; }
            return COS_OK; // adj_mag_valid stays false: the next graph change tries again
        }
        HIP_TRY(e);
        HIP_TRY(cosdev::launch_fill_adj_mag(H.d_adj_vec, ix->d_mags, H.d_adj_mag, H.n, H.M, slots, ix->own_stream));
    }
    HIP_TRY(hipStreamSynchronize(ix->own_stream));
    ix->adj_mag_valid = true;
    return COS_OK;
}

int32_t cos_prepare_walk_plans(cos_index *ix) {
    std::lock_guard<std::mutex> g(ix->mu);
    if (int32_t rc = ensure_adj_mags(ix)) return rc;
    if (ix->walk_order_min_B)
        if (int32_t rc = ensure_order_rank(ix)) return rc;
    if (walk_table_min_B(ix))
        if (int32_t rc = ensure_level_table(ix)) return rc;
    return COS_OK;
}

extern "C" int32_t cos_index_walk_order_cuts(cos_index *ix, uint32_t *out_levels, uint32_t cap, uint32_t *out_n) {
    if (!ix || !out_n || (cap && !out_levels)) return cos_fail(COS_ERR_INVALID, "null argument");
    *out_n = 0;
    if (!graph_ready(ix)) return cos_fail(COS_ERR_NOT_READY, "index needs vectors, root and every graph level");
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    std::lock_guard<std::mutex> g(ix->mu);
    if ((rc = ensure_order_rank(ix))) return rc;
    for (u32 l : ix->order_levels) {
        if (*out_n < cap) out_levels[*out_n] = l;
        (*out_n)++;
    }
    return COS_OK;
}

static int32_t get_workspace(cos_index *ix, void *key, hipStream_t st, u32 B, u32 top_k, bool host_api, Workspace **out) {
    std::lock_guard<std::mutex> g(ix->mu);
    Workspace *&w = ix->ws[key];
    if (!w) w = new Workspace();
    const u32 L1 = ix->p.num_layers + 1;
    if (B > w->capB) {
        u32 cap = std::max(B, 64u);
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(regrow(w->q_codes, (size_t)cap * ix->row_stride));
        HIP_TRY(regrow(w->q_mags, cap));
        HIP_TRY(regrow(w->q_raw_mags, cap));
        HIP_TRY(regrow(w->walk_ids, (size_t)cap * L1 * KEEP_SEARCH));
        HIP_TRY(regrow(w->walk_sims, (size_t)cap * L1 * KEEP_SEARCH));
        HIP_TRY(regrow(w->walk_counts, (size_t)cap * L1));
        HIP_TRY(regrow(w->walk_status, cap));
        HIP_TRY(regrow(w->stats, (size_t)cap * 4));
        HIP_TRY(regrow(w->stats2, (size_t)cap * 4));
        HIP_TRY(regrow(w->qsums, cap));
        if (ix->eng == ENG_Q2) HIP_TRY(regrow(w->qdig, (size_t)cap * (ix->row_stride / 16) * 64)); // the table GEMM's digit rows of the queries
        HIP_TRY(regrow(w->fin_flags, (size_t)cap + 1));
        HIP_TRY(regrow(w->rerank_rows, cap));
        HIP_TRY(regrow(w->d_queries, (size_t)cap * ix->p.dim));
        HIP_TRY(regrow(w->d_out_counts, cap));
        HIP_TRY(regrow(w->d_out_status, cap));
        w->capB = cap;
        w->cap_topk = 0;
    }
    if (!ix->adj_mag_valid && B >= 1024u) // (a root replaced on a live graph; small launches keep their gathers until a big one pays for the refill)
        if (int32_t rc = ensure_adj_mags(ix)) return rc;
    if (ix->walk_order_min_B && B >= ix->walk_order_min_B && ix->p.ef_search <= 256u) { // wider beams keep the single launch (run_search)
        if (w->order.cap < w->capB) {
            HIP_TRY(hipStreamSynchronize(st));
            HIP_TRY(cosdev::walk_order_reserve(w->order, w->capB));
        }
        if (int32_t rc = ensure_order_rank(ix)) return rc;
    }
    if (const u32 tmin = walk_table_min_B(ix); tmin && (B >= tmin || B <= ix->lat4_max_B)) { // from min_queries on, and the four-wave latency kernel's launches
        if (int32_t rc = ensure_level_table(ix)) return rc;
        const size_t need = (size_t)w->capB * ix->table_stride;
        if (need > w->tab_cap) {
            // every workspace (one per caller stream, up to five per leased host pipe) holds the table of its own launch in flight:
            // a handle's tables together stay under a budget (tuning knob walk_table_max_bytes, default 48 GiB) — a workspace that would
            // exceed it simply walks without a table (same results)
            const size_t budget = (size_t)cosdev::tune_or(cosdev::TUNE_WALK_TABLE_MAX_BYTES, (long long)48 << 30);
            HIP_TRY(hipStreamSynchronize(st));
            if (w->tab) HIP_TRY(hipFree(w->tab));
            ix->table_bytes_total -= w->tab_cap * 4;
            w->tab = nullptr;
            w->tab_cap = 0;
            if (ix->table_bytes_total + need * 4 <= budget) {
                if (const hipError_t e = hipMalloc((void **)&w->tab, need * 4); e == hipSuccess) {
                    w->tab_cap = need;
                    ix->table_bytes_total += need * 4;
                } else if (e == hipErrorOutOfMemory) { // HBM is short: this workspace walks without a table (same results)
                    (void)hipGetLastError();
                    w->tab = nullptr;
                } else
                    HIP_TRY(e);
            }
        }
    }
    if (host_api && (size_t)top_k * w->capB > (size_t)w->cap_topk * w->capB) {
        HIP_TRY(regrow(w->d_out_ids, (size_t)w->capB * top_k));
        HIP_TRY(regrow(w->d_out_scores, (size_t)w->capB * top_k));
        w->cap_topk = top_k;
    }
    if (w->ev.empty()) {
        w->ev.assign((size_t)Workspace::EV_RING * Workspace::EV_PER, nullptr);
        for (auto &e : w->ev) HIP_TRY(hipEventCreate(&e));
    }
    if (!w->walk_done) HIP_TRY(hipEventCreateWithFlags(&w->walk_done, hipEventDisableTiming));
    if (!w->walk_fin) HIP_TRY(hipEventCreateWithFlags(&w->walk_fin, hipEventDisableTiming));
    if (!w->last_range) HIP_TRY(hipEventCreateWithFlags(&w->last_range, hipEventDisableTiming));
    *out = w;
    return COS_OK;
}

// quantize -> walk on `st`, then (finalize) on `st_fin` (the same stream unless the caller pipelines chunks: then `st_fin` waits
// for the walk through an event); all buffers device memory.  `chain` = take part in the walk chain (big launches of
// different streams one after the other); the chunks of one pipelined host call co-run instead.
static int32_t run_search(cos_index *ix, Workspace *w, const float *d_queries, u32 B, u32 top_k, u32 *d_out_ids, float *d_out_scores,
                          u32 *d_out_counts, int32_t *d_out_status, bool do_finalize, hipStream_t st, hipStream_t st_fin = nullptr,
                          hipEvent_t walk_ev = nullptr, bool chain = true) {
    IndexDev dev = cos_make_index_dev(ix);
    bool timed;
    u32 ef, lat_max_B, lat4_max_B, order_min_B, n_keys = 0, key_level[cosdev::MAX_LEVELS], key_n[cosdev::MAX_LEVELS];
    const u32 *order_rank[cosdev::MAX_LEVELS];
    u32 tab_level_min = 0, tab_cols = 0, tab_col0[cosdev::MAX_LEVELS] = {}, tab_min_B = 0;
    u64 tab_stride = 0;
    const uint8_t *tcodes = nullptr;
    const float *tmags = nullptr;
    const u32 *tcsums = nullptr;
    { // one consistent snapshot of the knobs cos_index_set_* may change from another thread
        std::lock_guard<std::mutex> g(ix->mu);
        order_min_B = ix->order_rank_valid && !ix->order_levels.empty() ? ix->walk_order_min_B : 0u;
        for (u32 l : ix->order_levels) {
            key_level[n_keys] = l;
            key_n[n_keys] = ix->order_rank_n[l];
            order_rank[n_keys++] = ix->d_order_rank[l];
        }
        tab_min_B = walk_table_min_B(ix);
        if (tab_min_B && ix->level_table_valid && ix->table_level_min && w->tab && (size_t)B * ix->table_stride <= w->tab_cap) {
            tab_level_min = ix->table_level_min;
            tab_cols = ix->table_cols;
            tab_stride = ix->table_stride;
            memcpy(tab_col0, ix->table_col0, sizeof(tab_col0));
            tcodes = ix->d_tcodes;
            tmags = ix->d_tmags;
            tcsums = ix->d_tcsums;
        }
        timed = ix->timing;
        ef = ix->p.ef_search;
        lat_max_B = ix->lat_max_B;
        lat4_max_B = ix->lat4_max_B;
        dev.visited_mode = ix->p.visited_mode;
    }
    // Norms beside the adjacency (LevelDev::adj_mag) cost two more lines per window entry and save one line per winner: they pay while an
    // expansion still finds several winners, i.e. while the level's filter (64 x M bits) is not saturated by the ef pops of the level.
    // Measured (profiles/r06_adjmag_probe_*.jsonl): 1M x 768, M0 64 — ef 64: 7.8 evaluations per expansion, lower range 2.64 -> 2.42 ms;
    // ef 256: 2.35 per expansion, 5.10 -> 5.31 ms; 12.5M x 1024, M0 256 / M 64, ef 128: 9.8 per expansion, 26.5 -> 24.3 ms.
    // ... and while the launch is bound by bandwidth at all: one client batch is a chain of dependent round trips per query, where two more
    // loads and an LDS round trip per window entry only add latency (single batch of c2: 407 k QPS without, 396 k with).
    if (cosdev::tune_or(cosdev::TUNE_WALK_ADJ_MAG, 1) != 2) // (2 = at every ef and launch size: experiments)
        for (u32 l = 0; l <= dev.num_layers; l++)
            if (ef > 2u * dev.lv[l].M || B < 4096u) dev.lv[l].adj_mag = nullptr;
    WalkArgs wa;
    memset(&wa, 0, sizeof(wa));
    if (dev.visited_mode == COS_VISITED_EXACT) {
        int32_t rc = vis_tab_prepare(w->vis, ix, B, ef, st, wa);
        if (rc) return rc;
    }
    // which kernel walks this launch decides whether the level table is worth its GEMM: the throughput kernel (big launches, from
    // walk_table_min_B queries) and the four-wave latency kernel (one client batch) read it, the one-wave latency kernel does not
    // (a launch outside the fast kernels' domain — walk_general_kernel — walks every level from rows, in one launch, in arrival order)
    const bool general = cosdev::walk_general_needed(dev, ef);
    const bool ordered = !general && order_min_B && B >= order_min_B && n_keys > 0 && w->order.cap >= B && ef <= 256u;
    if (general) tab_level_min = 0;
    if (tab_level_min) {
        WalkArgs probe;
        memset(&probe, 0, sizeof(probe));
        probe.B = B;
        probe.ef = ef;
        const int kind = ordered ? 0 : cosdev::walk_kernel_kind(ix->eng, dev, probe, lat_max_B, lat4_max_B, true);
        // (the four-wave kernel reads the table of u8 codes only)
        if (!((kind == 4 && ix->eng == ENG_U8) || (kind == 0 && (B >= tab_min_B || B <= lat4_max_B)))) tab_level_min = 0;
    }
    hipEvent_t *ev = &w->ev[(size_t)(w->ev_count % Workspace::EV_RING) * Workspace::EV_PER];
    if (timed) HIP_TRY(hipEventRecord(ev[0], st));
    HIP_TRY(launch_quantize_rows(ix->eng, d_queries, ix->p.dim, B, ix->p.dim, ix->p.range_lo, ix->p.range_hi, w->q_codes, ix->row_stride,
                                 w->q_mags, w->q_raw_mags, st));
    // Level table: on the caller's stream, i.e. BEFORE the walk takes its place in the walk chain — the GEMM of this launch runs next
    // to the previous launch's walk.  (Round 5 also issued it inside the chain, right in front of its own walk, where it shares the
    // chip with nobody: 7.07-7.12 against 6.69 ms per step — the overlap hides 0.4 ms.  profiles/r05_table_gemm_in_chain_probe_not_kept.jsonl)
    const bool chained = chain && B >= ix->chain_min_B;
    if (tab_level_min) {
        // (engine_internal.h, chain_last_range_ev.)  Only a GEMM long enough to outlast the previous walk's upper range blocks its sort: the
        // shard's 65 161 columns x 32 768 queries (3.1 ms alone; gate: 33.3 -> 32.6 ms per step) do, c2's 20 903 (0.85 ms; 6.22 -> 6.29) do not
        // (profiles/r06_inflight_probe.txt); knob: 0 = never, 2 = always
        const long long gate = cosdev::tune_or(cosdev::TUNE_WALK_TABLE_AFTER_SORT, 1);
        if (chained && (gate == 2 || (gate == 1 && (u64)tab_cols * B >= (1ull << 30)))) {
            std::lock_guard<std::mutex> g(ix->chain_mu);
            if (ix->chain_last_range_ev && ix->chain_last_range_ev != w->last_range) HIP_TRY(hipStreamWaitEvent(st, ix->chain_last_range_ev, 0));
        }
        if (timed) HIP_TRY(hipEventRecord(ev[4], st));
        HIP_TRY(cosdev::launch_level_table(ix->eng, w->q_codes, w->q_mags, w->qsums, w->qdig, B, tcodes, tmags, tcsums, ix->row_stride, tab_cols, w->tab,
                                           tab_stride, ix->n_cus, st));
        if (timed) HIP_TRY(hipEventRecord(ev[5], st));
    }
    if (timed) HIP_TRY(hipEventRecord(ev[1], st));
    wa.qcodes = w->q_codes;
    wa.qmags = w->q_mags;
    wa.B = B;
    wa.ef = ef;
    wa.keep = KEEP_SEARCH;
    wa.out_ids = w->walk_ids;
    wa.out_sims = w->walk_sims;
    wa.out_counts = w->walk_counts;
    wa.out_status = w->walk_status;
    wa.out_stats = w->stats;
    wa.out_stats2 = w->stats2;
    HIP_TRY(hipMemsetAsync(w->stats2, 0, (size_t)B * 32, st)); // the latency kernels do not write it
    if (tab_level_min) {
        wa.tab = w->tab;
        wa.tab_mags = tmags;
        wa.tab_stride = tab_stride;
        wa.tab_level_min = tab_level_min;
        memcpy(wa.tab_col0, tab_col0, sizeof(tab_col0));
    }
    // Big walks run on the workspace's own LOW-PRIORITY stream (ordered after the quantize and before the finalize of `st` by
    // events): the short kernels around a walk — the neighbouring launch's quantize and, above all, its finalize, which used to
    // take 11-16 ms instead of 1.2 when its waves queued behind 32 768 walk waves of the next launch — are dispatched ahead of
    // the walk's pending workgroups.  The caller sees the same stream order.
    hipStream_t sw = st;
    if (ix->walk_side_min_B && B >= ix->walk_side_min_B) {
        if (!w->walk_stream) {
            int lo = 0, hi = 0;
            HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi)); // numerically largest = lowest priority
            HIP_TRY(hipStreamCreateWithPriority(&w->walk_stream, hipStreamNonBlocking, lo));
            HIP_TRY(hipEventCreateWithFlags(&w->prep_done, hipEventDisableTiming));
        }
        sw = w->walk_stream;
        HIP_TRY(hipEventRecord(w->prep_done, st));
        HIP_TRY(hipStreamWaitEvent(sw, w->prep_done, 0));
    }
    // Big launches walk in locality order (kernels_order.hip): the levels down to the first key level in arrival order, then, after
    // every key level, a sort of the launch by the key that level left and the levels below with the sorted queries dealt to the
    // XCDs.  Same walks, same results.
    // Beam widths above 256 keep the single launch: the cut costs the tail of one more launch, a fixed ~3 % of the walk whatever
    // the ef, while what the order saves shrinks as the walk turns from memory-bound to bound by its own serial work (c2: +16 % at
    // ef 64, +3.6 % at 128, +2.6 % at 256, +1 % at 512; nothing measurable at ef 512 on the uniform corpus or on a 12.5M shard:
    // profiles/archive/r03_order_probe_*.jsonl and the two r03_final_bench_default_* lines, taken with and without this rule).
    auto walk = [&](hipStream_t s) -> int32_t {
        if (!ordered) {
            if (chained) HIP_TRY(hipEventRecord(w->last_range, s));
            HIP_TRY(launch_walk(ix->eng, dev, wa, lat_max_B, lat4_max_B, s));
            return COS_OK;
        }
        wa.phase = 1;
        wa.entry0 = w->order.entry0;
        wa.order_key = w->order.order_key;
        wa.order_iota = w->order.iota;
        u32 first = dev.num_layers;
        for (u32 i = 0; i <= n_keys; i++) {
            const bool last = i == n_keys;
            wa.level_first = first;
            wa.level_last = last ? 0u : key_level[i];
            wa.key_n = last ? 0u : key_n[i];
            wa.order_rank = last ? nullptr : order_rank[i];
            if (last && chained) HIP_TRY(hipEventRecord(w->last_range, s));
            HIP_TRY(launch_walk(ix->eng, dev, wa, 0, 0, s));
            if (last) break;
            if (timed && i == 0) HIP_TRY(hipEventRecord(ev[6], s));
            HIP_TRY(cosdev::launch_walk_order(w->order, B, key_n[i], ix->num_xcd, s));
            if (timed && i == 0) HIP_TRY(hipEventRecord(ev[7], s));
            wa.q_order = w->order.q_order;
            first = key_level[i] - 1;
        }
        return COS_OK;
    };
    if (chained) { // walk chain (engine_internal.h): wait for the previous big walk, whichever stream it ran on
        std::lock_guard<std::mutex> g(ix->chain_mu);
        if (ix->chain_ev && ix->chain_ev != w->walk_done) HIP_TRY(hipStreamWaitEvent(sw, ix->chain_ev, 0));
        if (timed) HIP_TRY(hipEventRecord(ev[1], sw)); // the kernel's own duration: after the wait
        if (int32_t rc = walk(sw)) return rc;
        HIP_TRY(hipEventRecord(w->walk_done, sw));
        ix->chain_ev = w->walk_done;
        ix->chain_last_range_ev = w->last_range;
    } else {
        if (timed && sw != st) HIP_TRY(hipEventRecord(ev[1], sw));
        if (int32_t rc = walk(sw)) return rc;
    }
    if (timed) HIP_TRY(hipEventRecord(ev[2], sw));
    hipStream_t sf = st_fin ? st_fin : st;
    if (sf != sw) { // finalize on another stream than the walk's, ordered after it
        hipEvent_t we = walk_ev ? walk_ev : w->walk_fin;
        HIP_TRY(hipEventRecord(we, sw));
        HIP_TRY(hipStreamWaitEvent(sf, we, 0));
    }
    if (do_finalize) {
        HIP_TRY(launch_finalize(dev, d_queries, ix->p.dim, w->q_raw_mags, w->walk_ids, w->walk_sims, w->walk_counts, w->walk_status, B, top_k,
                                d_out_ids, d_out_scores, d_out_counts, d_out_status, w->rerank_rows, sf, ordered ? w->order.q_order : nullptr, w->fin_flags));
    }
    if (timed) { HIP_TRY(hipEventRecord(ev[3], sf)); w->ev_count++; }
    w->lastB = B;
    w->timed = timed;
    w->last_tab = tab_level_min != 0;
    w->last_tab_cols = tab_cols;
    w->last_tab_level_min = tab_level_min;
    w->last_split = ordered;
    w->last_cut_level = ordered ? key_level[0] : 0u;
    { std::lock_guard<std::mutex> g(ix->mu); ix->last_ws = w; }
    return COS_OK;
}

static int32_t check_search_args(cos_index *ix, const void *q, u32 B, u32 top_k) {
    if (!ix || !q) return cos_fail(COS_ERR_INVALID, "null argument");
    if (!graph_ready(ix)) return cos_fail(COS_ERR_NOT_READY, "index needs vectors, root and every graph level before search");
    if (top_k == 0 || top_k > 1024) return cos_fail(COS_ERR_INVALID, "top_k must be in [1, 1024]");
    if (B == 0) return cos_fail(COS_ERR_INVALID, "empty batch");
    return cos_set_device(ix);
}

extern "C" int32_t cos_search_batch_device(cos_index *ix, const float *d_queries, uint32_t B, uint32_t top_k, uint32_t *d_out_ids,
                                           float *d_out_scores, uint32_t *d_out_counts, int32_t *d_out_status, void *stream) {
    int32_t rc = check_search_args(ix, d_queries, B, top_k);
    if (rc) return rc;
    if (!d_out_ids || !d_out_scores || !d_out_counts || !d_out_status) return cos_fail(COS_ERR_INVALID, "null output");
    Workspace *w;
    rc = get_workspace(ix, stream, (hipStream_t)stream, B, top_k, false, &w);
    if (rc) return rc;
    return run_search(ix, w, d_queries, B, top_k, d_out_ids, d_out_scores, d_out_counts, d_out_status, true, (hipStream_t)stream);
}

// host API: every call leases a HostPipe (private streams, staging and, through it, private workspaces) from the handle's
// bounded pool, so concurrent callers (rayon workers, indexes/mod.rs:268-271) neither serialise nor share staging buffers, and
// a host that churns threads (pool restarts, per-request threads) re-uses the same few pipes instead of growing device memory.
// The pipes belong to the handle and die with it (cos_index_destroy).
struct PipeLease {
    cos_index *ix;
    HostPipe *hp = nullptr;
    explicit PipeLease(cos_index *ix_) : ix(ix_) {}
    PipeLease(const PipeLease &) = delete;
    PipeLease &operator=(const PipeLease &) = delete;
    int32_t acquire() {
        std::unique_lock<std::mutex> lk(ix->pipe_mu);
        for (;;) {
            if (!ix->pipes_free.empty()) { hp = ix->pipes_free.back(); ix->pipes_free.pop_back(); break; }
            if (ix->pipes_all.size() < COS_MAX_HOST_PIPES) {
                HostPipe *n = new HostPipe();
                hipError_t e = hipStreamCreateWithFlags(&n->s[0], hipStreamNonBlocking);
                if (e != hipSuccess) { delete n; return cos_fail(COS_ERR_HIP, "hipStreamCreateWithFlags: %s", hipGetErrorString(e)); }
                ix->pipes_all.push_back(n);
                hp = n;
                break;
            }
            ix->pipe_cv.wait(lk); // COS_MAX_HOST_PIPES calls are in flight: wait for one to finish
        }
        ix->host_calls_active.fetch_add(1);
        return COS_OK;
    }
    ~PipeLease() {
        if (!hp) return;
        ix->host_calls_active.fetch_sub(1);
        { std::lock_guard<std::mutex> g(ix->pipe_mu); ix->pipes_free.push_back(hp); }
        ix->pipe_cv.notify_one();
    }
};

template <typename T>
static hipError_t grow_elems(T *&p, size_t &cap, size_t need) { // one capacity per buffer; a failed allocation leaves (null, 0)
    if (need <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc((void **)&p, need * sizeof(T));
    if (e == hipSuccess) cap = need;
    return e;
}

static int32_t report_status(const int32_t *status, u32 B) {
    for (u32 b = 0; b < B; b++)
        if (status[b] != COS_OK) return cos_fail(status[b], "query %u failed with status %d (zero-norm vector -> DistanceError::CalculationError)", b, status[b]);
    return COS_OK;
}

// one launch for a contiguous host batch, on the leased pipe's stream
static int32_t search_host_simple(cos_index *ix, HostPipe *hp, const float *queries, uint32_t B, uint32_t top_k, uint32_t *out_ids, float *out_scores,
                                  uint32_t *out_counts, int32_t *out_status) {
    hipStream_t st = hp->s[0];
    Workspace *w;
    int32_t rc = get_workspace(ix, (void *)st, st, B, top_k, true, &w);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(w->d_queries, queries, (size_t)B * ix->p.dim * 4, hipMemcpyHostToDevice, st));
    rc = run_search(ix, w, w->d_queries, B, top_k, w->d_out_ids, w->d_out_scores, w->d_out_counts, w->d_out_status, true, st);
    if (rc) { (void)hipDeviceSynchronize(); return rc; } // the walk may be running on the workspace's side stream
    std::vector<int32_t> status(B);
    hipError_t e = hipMemcpyAsync(out_ids, w->d_out_ids, (size_t)B * top_k * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(out_scores, w->d_out_scores, (size_t)B * top_k * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(out_counts, w->d_out_counts, (size_t)B * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(status.data(), w->d_out_status, (size_t)B * 4, hipMemcpyDeviceToHost, st);
    const hipError_t es = hipStreamSynchronize(st); // always drained before the host buffers go out of scope
    HIP_TRY(e);
    HIP_TRY(es);
    if (out_status) memcpy(out_status, status.data(), (size_t)B * 4);
    return report_status(status.data(), B);
}

// A big host batch with nobody else on the device: the call's own copies would sit in front of and behind its one walk
// (H2D of 100 MB ahead of a 32 768 x 768 launch, finalize + D2H after it).  So the batch runs as <= 4 chunks:
//   sc        H2D chunk 0 | H2D chunk 1 | H2D chunk 2 | H2D chunk 3              (straight from the caller's buffer)
//   s[0]                   quantize+walk 0             | quantize+walk 2
//   s[1]                                 quantize+walk 1             | quantize+walk 3
//   sf                                                   finalize 0 | finalize 1 | finalize 2 | finalize 3 | D2H
// chunk i+1 travels while chunk i is walked, chunk i is reranked while chunk i+1 is walked, and neighbouring walks co-run (the
// next one fills the wave slots the previous one drains; they are not put in the walk chain).  Results are those of one
// launch: queries are independent.  When other host calls are in flight the same overlap already happens ACROSS calls (their
// copies hide under this call's chained walk), and whole-batch launches are the better shape — then the simple path is used.
// (Round 4 wrote a threaded pinned staging of the chunks, round 5 measured it: one caller 3.26 M QPS with 4 or 8 helper threads against
// 3.44 M for this plain pageable copy — profiles/r05_candidates_host_stage.txt — and removed it.)
static int32_t search_host_pipelined(cos_index *ix, HostPipe *hp, const float *queries, uint32_t B, uint32_t top_k, uint32_t *out_ids,
                                     float *out_scores, uint32_t *out_counts, int32_t *out_status) {
    const size_t dim = ix->p.dim;
    const u32 chunk = (((B + HostPipe::MAX_CHUNKS - 1) / HostPipe::MAX_CHUNKS) + 255u) & ~255u;
    const u32 nch = (B + chunk - 1) / chunk;
    if (!hp->s[1]) HIP_TRY(hipStreamCreateWithFlags(&hp->s[1], hipStreamNonBlocking));
    if (!hp->sc) HIP_TRY(hipStreamCreateWithFlags(&hp->sc, hipStreamNonBlocking));
    if (!hp->sf) HIP_TRY(hipStreamCreateWithFlags(&hp->sf, hipStreamNonBlocking));
    for (u32 i = 0; i < HostPipe::MAX_CHUNKS; i++) {
        if (!hp->ev_in[i]) HIP_TRY(hipEventCreateWithFlags(&hp->ev_in[i], hipEventDisableTiming));
        if (!hp->ev_walk[i]) HIP_TRY(hipEventCreateWithFlags(&hp->ev_walk[i], hipEventDisableTiming));
    }
    HIP_TRY(grow_elems(hp->d_q, hp->cap_q, (size_t)B * dim));
    HIP_TRY(grow_elems(hp->d_ids, hp->cap_ids, (size_t)B * top_k));
    HIP_TRY(grow_elems(hp->d_scores, hp->cap_scores, (size_t)B * top_k));
    HIP_TRY(grow_elems(hp->d_counts, hp->cap_counts, (size_t)B));
    HIP_TRY(grow_elems(hp->d_status, hp->cap_status, (size_t)B));
    struct Drain { // whatever happens, nothing of this call is still running when its buffers are handed back
        HostPipe *hp;
        bool ok = false;
        ~Drain() {
            hipStream_t sts[] = {hp->sc, hp->s[0], hp->s[1], hp->sf};
            for (hipStream_t st : sts) if (st) (void)hipStreamSynchronize(st);
            // big chunks walk on their workspace's own low-priority stream: on an error return that walk may have been launched
            // without `sf` ever waiting for it, so the error path drains the device before the pipe's buffers are handed back
            if (!ok) (void)hipDeviceSynchronize();
        }
    } drain{hp};
    for (u32 i = 0; i < nch; i++) {
        const u32 c0 = i * chunk, cb = std::min(chunk, B - c0);
        hipStream_t st = hp->s[i & 1];
        Workspace *w;
        int32_t rc = get_workspace(ix, (void *)&hp->wkey[i], st, cb, top_k, false, &w);
        if (rc) return rc;
        float *dq = hp->d_q + (size_t)c0 * dim;
        const float *src = queries + (size_t)c0 * dim;
        HIP_TRY(hipMemcpyAsync(dq, src, (size_t)cb * dim * 4, hipMemcpyHostToDevice, hp->sc));
        HIP_TRY(hipEventRecord(hp->ev_in[i], hp->sc));
        HIP_TRY(hipStreamWaitEvent(st, hp->ev_in[i], 0));
        rc = run_search(ix, w, dq, cb, top_k, hp->d_ids + (size_t)c0 * top_k, hp->d_scores + (size_t)c0 * top_k, hp->d_counts + c0, hp->d_status + c0,
                        true, st, hp->sf, hp->ev_walk[i], false);
        if (rc) return rc;
    }
    std::vector<int32_t> status(B);
    hipError_t e = hipMemcpyAsync(out_ids, hp->d_ids, (size_t)B * top_k * 4, hipMemcpyDeviceToHost, hp->sf);
    if (e == hipSuccess) e = hipMemcpyAsync(out_scores, hp->d_scores, (size_t)B * top_k * 4, hipMemcpyDeviceToHost, hp->sf);
    if (e == hipSuccess) e = hipMemcpyAsync(out_counts, hp->d_counts, (size_t)B * 4, hipMemcpyDeviceToHost, hp->sf);
    if (e == hipSuccess) e = hipMemcpyAsync(status.data(), hp->d_status, (size_t)B * 4, hipMemcpyDeviceToHost, hp->sf);
    const hipError_t es = hipStreamSynchronize(hp->sf);
    HIP_TRY(e);
    HIP_TRY(es);
    drain.ok = true;
    if (out_status) memcpy(out_status, status.data(), (size_t)B * 4);
    return report_status(status.data(), B);
}

static u32 host_pipeline_min_B() { // tuning knob host_pipeline_min_b: experiments (0 = never chunk)
    return (u32)cosdev::tune_or(cosdev::TUNE_HOST_PIPELINE_MIN_B, 8192);
}

static int32_t search_host_once(cos_index *ix, const float *queries, uint32_t B, uint32_t top_k, uint32_t *out_ids, float *out_scores,
                                uint32_t *out_counts, int32_t *out_status) {
    PipeLease lease(ix);
    int32_t rc = lease.acquire();
    if (rc) return rc;
    const u32 min_B = host_pipeline_min_B();
    if (min_B && B >= min_B && ix->host_calls_active.load() == 1)
        return search_host_pipelined(ix, lease.hp, queries, B, top_k, out_ids, out_scores, out_counts, out_status);
    return search_host_simple(ix, lease.hp, queries, B, top_k, out_ids, out_scores, out_counts, out_status);
}

// ------------------------------------------------------------------------------------------------
// Request coalescing for the host API (dynamic batching).  The walk kernel gives every query one
// wavefront, so a 256-CU chip wants thousands of queries per launch, while the reference's callers
// (rayon workers, actix handlers: indexes/mod.rs:268-271, tests/rps-test.py) submit a few hundred at a
// time from many threads.  With coalescing on, concurrent cos_search_batch calls that use the same top_k
// are fused: the first caller becomes the leader, waits `window_us` for followers (or until `max_queries`
// are pending), runs ONE launch for all of them and hands every caller its own slice.  Results are
// identical to un-coalesced calls (queries are independent); errors stay per request.
// ------------------------------------------------------------------------------------------------
// A slot = one launch being assembled: pinned staging for up to `cap` queries and their results.  Callers reserve a range under
// co_mu, copy their own queries into the slot's pinned buffer IN PARALLEL (128 callers x 786 KB: the copies of a group used to be
// issued one after the other by the leader, each a pageable H2D of its own with its pinning overhead), and copy their own results out
// the same way; the leader — the request that opened the slot — only waits for the slot to close, issues ONE H2D / launch / D2H and
// wakes the others.  Slots are pooled per handle; a slot is recycled when its last request has copied its results out.
// Round 5 (64 callers ran at 3.2 M QPS, 128 at 0.75-0.87 M):
//   * the leader and the followers of a slot slept on ONE condition variable under the handle-wide co_mu, and every arrival
//     notified all of them: 64 arrivals x 63 sleeping followers, each wake-up a round trip through co_mu, in front of the very
//     arrivals the leader was waiting for — which then came in a trickle, and the quiet window closed the slot with a fraction of
//     them.  Now the leader alone is notified by an arrival (`cv_leader`, with co_mu), the followers sleep on the slot's own
//     mutex / `cv_done`, and a request leaves the slot through an atomic counter;
//   * WHEN a slot launches follows the device, not only the clock: while two launches of the handle are in flight a third would
//     only queue behind them, so the open slot keeps gathering requests (up to max_queries) and the completion of a launch wakes
//     its leader; with fewer than two in flight it launches when it is full or `window_us` after its last arrival (at most 8
//     windows after its first).  Under load batches grow to what the callers offer; an idle device answers after one window.
// 128 callers: 0.87 -> 3.86 M QPS, 256 callers: 1.13 -> 4.14 M (profiles/r05_callers_probe_*.jsonl).  When every caller fits ONE launch
// (64 callers x 256 with max_queries 16 384: 2.5-2.8 M) the device idles while the next launch is copied in; max_queries = half of
// what the callers offer keeps two launches alternating (64 callers, max_queries 8 192: 3.2 M).  Tried and not kept: (i) closing a
// slot as soon as it holds half of the queries of all requests then inside cos_search_batch — the instantaneous count of callers
// "inside" swings with every return, slots closed early, 128 / 256 callers fell to 3.3 / 3.1 M; (ii) running a launch that has the
// device to itself as the chunk pipeline of a lone big host call — 64 callers 2.47 M against 2.77 M, 32 callers 2.0 against 2.5 M.
struct CoSlot {
    float *pin_q = nullptr;     // [cap][dim]
    unsigned char *pin_out = nullptr; // ids [cap][k] | scores [cap][k] | counts [cap] | status [cap]
    u32 cap = 0, cap_k = 0;
    u32 top_k = 0;
    u32 reserved = 0, n_req = 0;           // queries reserved, requests in the slot (co_mu)
    std::atomic<u32> left{0};              // requests that have not left yet
    std::atomic<u32> copied{0};            // requests whose queries are in pin_q
    bool closed = false;                   // (co_mu)
    bool done = false;                     // (mu)
    int32_t rc = COS_OK;
    std::string err;
    std::condition_variable cv_leader;     // with co_mu: `reserved` grew / the slot closed / a launch of the handle completed
    std::mutex mu;                         // the followers' own lock: nobody but the slot's requests ever takes it
    std::condition_variable cv_done;       // with mu: `done`
};

static void co_slot_free(CoSlot *sl) {
    if (sl->pin_q) (void)hipHostFree(sl->pin_q);
    if (sl->pin_out) (void)hipHostFree(sl->pin_out);
    delete sl;
}
void cos_coalesce_release(cos_index *ix) { // cos_index_destroy
    for (CoSlot *sl : ix->co_slots) co_slot_free(sl);
    ix->co_slots.clear();
    ix->co_open = nullptr;
}

// an idle slot with room for max_q queries of top_k results (caller holds co_mu)
static CoSlot *co_slot_get(cos_index *ix, u32 max_q, u32 top_k) {
    CoSlot *sl = nullptr;
    for (CoSlot *c : ix->co_slots)
        if (c->left.load(std::memory_order_acquire) == 0 && !c->closed && c->n_req == 0) { sl = c; break; }
    if (!sl) {
        if (ix->co_slots.size() >= 8) return nullptr; // every slot is in flight: the caller launches on its own
        sl = new CoSlot();
        ix->co_slots.push_back(sl);
    }
    if (sl->cap < max_q) {
        if (sl->pin_q) (void)hipHostFree(sl->pin_q);
        sl->pin_q = nullptr;
        sl->cap = 0;
        if (hipHostMalloc((void **)&sl->pin_q, (size_t)max_q * ix->p.dim * 4, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        sl->cap = max_q;
        sl->cap_k = 0;
    }
    if (sl->cap_k < top_k) {
        if (sl->pin_out) (void)hipHostFree(sl->pin_out);
        sl->pin_out = nullptr;
        sl->cap_k = 0;
        if (hipHostMalloc((void **)&sl->pin_out, (size_t)sl->cap * top_k * 8 + (size_t)sl->cap * 8, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        sl->cap_k = top_k;
    }
    sl->top_k = top_k;
    sl->reserved = sl->n_req = 0;
    sl->left.store(0);
    sl->copied.store(0);
    sl->closed = sl->done = false;
    sl->rc = COS_OK;
    sl->err.clear();
    return sl;
}

// the leader's launch: pinned queries -> device, search, results -> pinned
static int32_t co_slot_run(cos_index *ix, CoSlot *sl) {
    const u32 total = sl->reserved, top_k = sl->top_k;
    PipeLease lease(ix);
    int32_t rc = lease.acquire();
    if (rc) return rc;
    hipStream_t st = lease.hp->s[0];
    u32 *p_ids = (u32 *)sl->pin_out;
    float *p_sc = (float *)(p_ids + (size_t)sl->cap * top_k);
    u32 *p_cnt = (u32 *)(p_sc + (size_t)sl->cap * top_k);
    int32_t *p_st = (int32_t *)(p_cnt + sl->cap);
    Workspace *w;
    rc = get_workspace(ix, (void *)st, st, total, top_k, true, &w);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(w->d_queries, sl->pin_q, (size_t)total * ix->p.dim * 4, hipMemcpyHostToDevice, st));
    rc = run_search(ix, w, w->d_queries, total, top_k, w->d_out_ids, w->d_out_scores, w->d_out_counts, w->d_out_status, true, st);
    if (rc) { (void)hipDeviceSynchronize(); return rc; } // the walk may be running on the workspace's side stream
    hipError_t e = hipMemcpyAsync(p_ids, w->d_out_ids, (size_t)total * top_k * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(p_sc, w->d_out_scores, (size_t)total * top_k * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(p_cnt, w->d_out_counts, (size_t)total * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(p_st, w->d_out_status, (size_t)total * 4, hipMemcpyDeviceToHost, st);
    const hipError_t es = hipStreamSynchronize(st);
    HIP_TRY(e);
    HIP_TRY(es);
    return COS_OK;
}

extern "C" int32_t cos_index_set_coalescing(cos_index *ix, uint32_t max_queries, uint32_t window_us) {
    if (!ix) return cos_fail(COS_ERR_INVALID, "null index");
    std::lock_guard<std::mutex> g(ix->co_mu);
    ix->co_max_queries = max_queries;
    ix->co_window_us = window_us;
    ix->co_stats = cos_coalescing_stats{};
    return COS_OK;
}

extern "C" int32_t cos_index_coalescing_stats(cos_index *ix, cos_coalescing_stats *out) {
    if (!ix || !out) return cos_fail(COS_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> g(ix->co_mu);
    *out = ix->co_stats;
    return COS_OK;
}

extern "C" int32_t cos_search_batch(cos_index *ix, const float *queries, uint32_t B, uint32_t top_k, uint32_t *out_ids, float *out_scores,
                                    uint32_t *out_counts, int32_t *out_status) {
    int32_t rc = check_search_args(ix, queries, B, top_k);
    if (rc) return rc;
    if (!out_ids || !out_scores || !out_counts) return cos_fail(COS_ERR_INVALID, "null output");
    const size_t dim = ix->p.dim;
    CoSlot *sl = nullptr;
    u32 off = 0, max_q = 0, window = 0;
    bool leader = false;
    {
        std::unique_lock<std::mutex> lk(ix->co_mu);
        max_q = ix->co_max_queries;
        window = ix->co_window_us;
        if (max_q != 0 && B < max_q) {
            sl = ix->co_open;
            if (sl && (sl->closed || sl->top_k != top_k || sl->reserved + B > max_q || sl->reserved + B > sl->cap)) {
                sl->closed = true; // full (or another top_k): its leader launches it as it is; a new slot opens
                sl->cv_leader.notify_one();
                sl = nullptr;
            }
            if (!sl) {
                sl = co_slot_get(ix, max_q, top_k);
                ix->co_open = sl;
                leader = sl != nullptr;
            }
            if (sl) {
                off = sl->reserved;
                sl->reserved += B;
                sl->n_req++;
                sl->left.fetch_add(1, std::memory_order_relaxed);
                if (sl->reserved >= max_q) { sl->closed = true; ix->co_open = nullptr; }
                if (!leader) sl->cv_leader.notify_one(); // the leader alone: the followers sleep on the slot's own cv_done
            } else
                ix->co_stats.solo_calls++;
        }
    }
    if (!sl) return search_host_once(ix, queries, B, top_k, out_ids, out_scores, out_counts, out_status);

    memcpy(sl->pin_q + (size_t)off * dim, queries, (size_t)B * dim * 4); // every caller stages its own queries, in parallel
    sl->copied.fetch_add(1, std::memory_order_release);
    if (leader) {
        u32 n_req;
        {
            std::unique_lock<std::mutex> lk(ix->co_mu);
            // Closing rule (see the struct): full -> now; fewer than two launches of the handle in flight -> `window` after the last
            // arrival, at most eight windows after the first; two or more in flight -> keep gathering, re-examined whenever a launch
            // completes (and every few windows, so that a lost wake-up costs time, not liveness).
            const auto t0 = std::chrono::steady_clock::now();
            const auto hard = t0 + std::chrono::microseconds(8ull * window);
            auto quiet = t0 + std::chrono::microseconds(window);
            u32 seen = sl->reserved;
            int why = 0; // 0 full, 1 quiet window, 2 eight windows
            while (!sl->closed) {
                const auto now = std::chrono::steady_clock::now();
                if (ix->co_inflight < 2u) {
                    if (now >= hard) { why = 2; break; }
                    if (now >= quiet) { why = 1; break; }
                    sl->cv_leader.wait_until(lk, std::min(quiet, hard));
                } else
                    sl->cv_leader.wait_until(lk, now + std::chrono::microseconds(4ull * std::max(window, 50u)));
                if (sl->reserved != seen) { seen = sl->reserved; quiet = std::chrono::steady_clock::now() + std::chrono::microseconds(window); }
            }
            sl->closed = true;
            if (ix->co_open == sl) ix->co_open = nullptr;
            n_req = sl->n_req;
            ix->co_inflight++;
            ix->co_stats.launches++;
            ix->co_stats.queries += sl->reserved;
            ix->co_stats.requests += n_req;
            if (why == 0) ix->co_stats.closed_full++;
            else if (why == 1) ix->co_stats.closed_quiet++;
            else ix->co_stats.closed_deadline++;
        }
        while (sl->copied.load(std::memory_order_acquire) < n_req) std::this_thread::yield(); // followers still copying (microseconds)
        const int32_t r = co_slot_run(ix, sl);
        const std::string e = r ? std::string(cos_last_error_string()) : std::string();
        {
            std::lock_guard<std::mutex> lk(ix->co_mu);
            ix->co_inflight--;
            if (ix->co_open) ix->co_open->cv_leader.notify_one(); // the slot that was gathering behind two launches may go now
        }
        {
            std::lock_guard<std::mutex> lk(sl->mu);
            sl->rc = r;
            sl->err = e;
            sl->done = true;
        }
        sl->cv_done.notify_all();
    } else {
        std::unique_lock<std::mutex> lk(sl->mu);
        while (!sl->done) sl->cv_done.wait(lk);
    }
    // every request takes its own slice out of the pinned result buffer and judges its own queries
    const int32_t slot_rc = sl->rc;
    std::string slot_err = slot_rc != COS_OK ? sl->err : std::string();
    int32_t my_rc = COS_OK;
    u32 bad_q = 0;
    if (slot_rc == COS_OK) {
        const u32 *p_ids = (const u32 *)sl->pin_out;
        const float *p_sc = (const float *)(p_ids + (size_t)sl->cap * top_k);
        const u32 *p_cnt = (const u32 *)(p_sc + (size_t)sl->cap * top_k);
        const int32_t *p_st = (const int32_t *)(p_cnt + sl->cap);
        memcpy(out_ids, p_ids + (size_t)off * top_k, (size_t)B * top_k * 4);
        memcpy(out_scores, p_sc + (size_t)off * top_k, (size_t)B * top_k * 4);
        memcpy(out_counts, p_cnt + off, (size_t)B * 4);
        if (out_status) memcpy(out_status, p_st + off, (size_t)B * 4);
        for (u32 b = 0; b < B; b++)
            if (p_st[off + b] != COS_OK) { my_rc = p_st[off + b]; bad_q = b; break; } // the reference fails the whole request of THIS caller (collect::<Result<_>>)
    }
    if (sl->left.fetch_sub(1, std::memory_order_acq_rel) == 1) { // the last request out: back in the pool
        std::lock_guard<std::mutex> lk(ix->co_mu);
        sl->n_req = 0;
        sl->closed = false;
    }
    if (slot_rc != COS_OK) return cos_fail(slot_rc, "%s", slot_err.c_str()); // HIP errors etc. hit every request of the launch
    if (my_rc != COS_OK) return cos_fail(my_rc, "query %u failed with status %d (zero-norm vector -> DistanceError::CalculationError)", bad_q, my_rc);
    return COS_OK;
}

extern "C" int32_t cos_ann_search_batch(cos_index *ix, const float *queries, uint32_t B, uint32_t *out_ids, float *out_sims,
                                        uint32_t *out_counts, int32_t *out_status) {
    int32_t rc = check_search_args(ix, queries, B, 1);
    if (rc) return rc;
    if (!out_ids || !out_sims || !out_counts) return cos_fail(COS_ERR_INVALID, "null output");
    PipeLease lease(ix); // a leased pipe like cos_search_batch: thread-safe on a shared handle
    rc = lease.acquire();
    if (rc) return rc;
    hipStream_t st = lease.hp->s[0];
    Workspace *w;
    rc = get_workspace(ix, (void *)st, st, B, 1, true, &w);
    if (rc) return rc;
    const u32 L1 = ix->p.num_layers + 1;
    HIP_TRY(hipMemcpyAsync(w->d_queries, queries, (size_t)B * ix->p.dim * 4, hipMemcpyHostToDevice, st));
    rc = run_search(ix, w, w->d_queries, B, 1, nullptr, nullptr, nullptr, nullptr, false, st);
    if (rc) return rc;
    std::vector<int32_t> status(B);
    HIP_TRY(hipMemcpyAsync(out_ids, w->walk_ids, (size_t)B * L1 * KEEP_SEARCH * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_sims, w->walk_sims, (size_t)B * L1 * KEEP_SEARCH * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_counts, w->walk_counts, (size_t)B * L1 * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(status.data(), w->walk_status, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (out_status) memcpy(out_status, status.data(), (size_t)B * 4);
    for (u32 b = 0; b < B; b++)
        if (status[b] != COS_OK) return cos_fail(status[b], "query %u failed with status %d", b, status[b]);
    return COS_OK;
}

extern "C" int32_t cos_index_last_stats(cos_index *ix, void *stream, cos_search_stats *out) {
    if (!ix || !out) return cos_fail(COS_ERR_INVALID, "null argument");
    memset(out, 0, sizeof(*out));
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    Workspace *w = nullptr;
    {
        std::lock_guard<std::mutex> g(ix->mu);
        if (stream) {
            auto it = ix->ws.find(stream);
            if (it != ix->ws.end()) w = it->second;
        } else
            w = ix->last_ws;
        if (!w) return cos_fail(COS_ERR_NOT_READY, "no batch has run on this stream");
    }
    if (w->lastB == 0) return cos_fail(COS_ERR_NOT_READY, "no batch has run on this stream");
    if (w->timed && w->ev_count) {
        hipEvent_t *ev = &w->ev[(size_t)((w->ev_count - 1) % Workspace::EV_RING) * Workspace::EV_PER];
        HIP_TRY(hipEventSynchronize(ev[3]));
        HIP_TRY(hipEventElapsedTime(&out->prep_ms, ev[0], ev[1]));
        HIP_TRY(hipEventElapsedTime(&out->walk_ms, ev[1], ev[2]));
        HIP_TRY(hipEventElapsedTime(&out->finalize_ms, ev[2], ev[3]));
    } else {
        HIP_TRY(hipDeviceSynchronize());
    }
    std::vector<u64> st((size_t)w->lastB * 4), rr(w->lastB);
    HIP_TRY(hipMemcpy(st.data(), w->stats, st.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(rr.data(), w->rerank_rows, rr.size() * 8, hipMemcpyDeviceToHost));
    for (u32 b = 0; b < w->lastB; b++) {
        out->evals += st[(size_t)b * 4 + 0];
        out->expansions += st[(size_t)b * 4 + 1];
        out->adj_bytes += st[(size_t)b * 4 + 2];
        out->reserved += (uint32_t)st[(size_t)b * 4 + 3]; // walk rounds (lookahead windows issued)
        out->rerank_rows += rr[b];
    }
    return COS_OK;
}

extern "C" int32_t cos_index_last_walk_split(cos_index *ix, void *stream, cos_walk_split *out) {
    if (!ix || !out) return cos_fail(COS_ERR_INVALID, "null argument");
    if (out->struct_size != sizeof(cos_walk_split)) return cos_fail(COS_ERR_INVALID, "cos_walk_split.struct_size mismatch");
    memset(out, 0, sizeof(*out));
    out->struct_size = sizeof(cos_walk_split);
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    Workspace *w = nullptr;
    {
        std::lock_guard<std::mutex> g(ix->mu);
        if (stream) {
            auto it = ix->ws.find(stream);
            if (it != ix->ws.end()) w = it->second;
        } else
            w = ix->last_ws;
    }
    if (!w || w->lastB == 0) return cos_fail(COS_ERR_NOT_READY, "no batch has run on this stream");
    out->queries = w->lastB;
    out->table_level_min = w->last_tab ? w->last_tab_level_min : 0u; // of THAT launch (the handle's may have moved on with ef_search)
    out->table_cols = w->last_tab ? w->last_tab_cols : 0u;
    out->cut_after_level = w->last_cut_level;
    if (w->timed && w->ev_count) {
        hipEvent_t *ev = &w->ev[(size_t)((w->ev_count - 1) % Workspace::EV_RING) * Workspace::EV_PER];
        HIP_TRY(hipEventSynchronize(ev[3]));
        if (w->last_tab) HIP_TRY(hipEventElapsedTime(&out->table_ms, ev[4], ev[5]));
        if (w->last_split) {
            HIP_TRY(hipEventElapsedTime(&out->upper_ms, ev[1], ev[6]));
            HIP_TRY(hipEventElapsedTime(&out->sort_ms, ev[6], ev[7]));
            HIP_TRY(hipEventElapsedTime(&out->lower_ms, ev[7], ev[2]));
        } else
            HIP_TRY(hipEventElapsedTime(&out->lower_ms, ev[1], ev[2]));
    } else {
        HIP_TRY(hipDeviceSynchronize());
    }
    const u32 kdims = (u32)((ix->row_stride + 63) / 64 * 64);
    out->table_int8_ops = 2.0 * (double)w->lastB * (double)out->table_cols * (double)kdims;
    std::vector<u64> st((size_t)w->lastB * 4), s2((size_t)w->lastB * 4);
    HIP_TRY(hipMemcpy(st.data(), w->stats, st.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(s2.data(), w->stats2, s2.size() * 8, hipMemcpyDeviceToHost));
    u64 te = 0, tx = 0, ta = 0;
    for (u32 b = 0; b < w->lastB; b++) {
        te += st[(size_t)b * 4 + 0];
        tx += st[(size_t)b * 4 + 1];
        ta += st[(size_t)b * 4 + 2];
        out->lower_evals += s2[(size_t)b * 4 + 0];
        out->lower_expansions += s2[(size_t)b * 4 + 1];
        out->lower_adj_bytes += s2[(size_t)b * 4 + 2];
        out->table_evals += s2[(size_t)b * 4 + 3];
    }
    out->upper_evals = te - out->lower_evals;
    out->upper_expansions = tx - out->lower_expansions;
    out->upper_adj_bytes = ta - out->lower_adj_bytes;
    return COS_OK;
}

extern "C" int32_t cos_index_timing_summary(cos_index *ix, void *stream, cos_timing_summary *out) {
    if (!ix || !out) return cos_fail(COS_ERR_INVALID, "null argument");
    memset(out, 0, sizeof(*out));
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    Workspace *w = nullptr;
    {
        std::lock_guard<std::mutex> g(ix->mu);
        auto it = ix->ws.find(stream);
        if (it != ix->ws.end()) w = it->second;
    }
    if (!w || w->ev_count == 0) return cos_fail(COS_ERR_NOT_READY, "no timed batch has run on this stream");
    const u32 n = std::min(w->ev_count, Workspace::EV_RING);
    out->launches = n;
    out->walk_ms_min = 1e30f;
    for (u32 i = 0; i < n; i++) {
        hipEvent_t *ev = &w->ev[(size_t)((w->ev_count - 1 - i) % Workspace::EV_RING) * Workspace::EV_PER];
        float a = 0, b = 0, c = 0;
        HIP_TRY(hipEventSynchronize(ev[3]));
        HIP_TRY(hipEventElapsedTime(&a, ev[0], ev[1]));
        HIP_TRY(hipEventElapsedTime(&b, ev[1], ev[2]));
        HIP_TRY(hipEventElapsedTime(&c, ev[2], ev[3]));
        out->prep_ms_sum += a;
        out->walk_ms_sum += b;
        out->finalize_ms_sum += c;
        out->walk_ms_min = std::min(out->walk_ms_min, b);
        out->walk_ms_max = std::max(out->walk_ms_max, b);
    }
    return COS_OK;
}

// ------------------------------------------------------------------------------------------------
// metadata-filtered search: the pseudo-root component and cos_search_filtered_batch
// ------------------------------------------------------------------------------------------------
static constexpr u32 PSEUDO_LO = 0xFFFFFEFEu, PSEUDO_HI = 0xFFFFFFFDu; // u32::MAX - 257 (pseudo_root_id, metadata/mod.rs:217) ..= u32::MAX - 2

extern "C" int32_t cos_index_enable_metadata(cos_index *ix, uint32_t mdim, uint32_t max_replicas_per_node) {
    if (!ix || mdim == 0 || mdim > 64 || max_replicas_per_node == 0) return cos_fail(COS_ERR_INVALID, "metadata dimensions must be in [1, 64] and max_replicas_per_node >= 1");
    if (!ix->have_vectors) return cos_fail(COS_ERR_NOT_READY, "upload vectors before enabling the metadata component");
    // (walk_meta_kernel keeps walk_kernel's domain: register pools, one lane per scanned slot)
    if (ix->p.ef_construction > cosdev::WALK_FAST_MAX_EF || std::min(ix->p.neighbors_count, ix->p.shortlist_size) > 64u ||
        std::min(ix->p.level0_neighbors_count, ix->p.shortlist_size) > 64u ||
        (ix->nchunks != 0u && (ix->nchunks + ix->G - 1u) / ix->G > (ix->eng == ENG_U8 ? 2u : 1u)))
        return cos_fail(COS_ERR_UNIMPLEMENTED, "metadata collections: ef_construction <= %u, at most 64 scanned neighbour slots per node, u8 rows of at most 2048 / SubByte rows of at most 64 x 16 bytes", cosdev::WALK_FAST_MAX_EF);
    if ((u64)ix->n * max_replicas_per_node >= PSEUDO_LO) return cos_fail(COS_ERR_INVALID, "replica ids would run into the reserved id range");
    // a SHARD of a metadata collection (round 6): the shard's embeddings are rows [0, n) here and embeddings [e0, e0 + n) of the
    // collection, so its replica ids are id_base + row x max_replicas + i with id_base = e0 x max_replicas — a multiple of max_replicas
    if (ix->p.id_base % max_replicas_per_node != 0)
        return cos_fail(COS_ERR_INVALID, "id_base %u of a metadata collection's shard must be a multiple of max_replicas_per_node %u (first embedding x replicas)", ix->p.id_base, max_replicas_per_node);
    if ((u64)ix->p.id_base + (u64)ix->n * max_replicas_per_node >= PSEUDO_LO) return cos_fail(COS_ERR_INVALID, "replica ids would run into the reserved id range");
    for (auto &l : ix->lv)
        if (l.n) return cos_fail(COS_ERR_INVALID, "enable the metadata component before uploading / building the base graph: it changes the id of every vector (row x max_replicas)");
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    ix->id_stride = max_replicas_per_node;
    ix->meta.mdim = mdim;
    ix->meta.lv.resize(ix->p.num_layers + 1);
    for (u32 l = 0; l <= ix->p.num_layers; l++) ix->meta.lv[l].M = l == 0 ? ix->p.level0_neighbors_count : ix->p.neighbors_count;
    // the pseudo nodes' vector: all zeros (pseudo_node_vector, metadata/mod.rs:211-214), quantized like any other vector -> row n + 1
    DevBuf z;
    HIP_TRY(z.alloc(((size_t)ix->p.dim + 1) * 4));
    HIP_TRY(hipMemsetAsync(z.p, 0, ((size_t)ix->p.dim + 1) * 4, ix->own_stream));
    HIP_TRY(launch_quantize_rows(ix->eng, z.as<float>(), ix->p.dim, 1, ix->p.dim, ix->p.range_lo, ix->p.range_hi,
                                 ix->d_codes + ((size_t)ix->n + 1) * ix->row_stride, ix->row_stride, ix->d_mags + ix->n + 1, z.as<float>() + ix->p.dim, ix->own_stream));
    HIP_TRY(hipStreamSynchronize(ix->own_stream));
    return COS_OK;
}

extern "C" int32_t cos_index_upload_meta_nodes(cos_index *ix, uint32_t n_nodes, const uint32_t *node_ids, const int32_t *mbits) {
    if (!ix || !node_ids || !mbits || n_nodes == 0) return cos_fail(COS_ERR_INVALID, "null/empty node table");
    if (!ix->meta.mdim || !ix->have_vectors) return cos_fail(COS_ERR_NOT_READY, "cos_index_enable_metadata first");
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    const u32 md = ix->meta.mdim;
    for (u32 i = 0; i < n_nodes; i++) {
        if (i && node_ids[i] <= node_ids[i - 1]) return cos_fail(COS_ERR_INVALID, "node ids must be strictly ascending");
        const bool pseudo = node_ids[i] >= PSEUDO_LO && node_ids[i] <= PSEUDO_HI;
        if (!pseudo && node_ids[i] / ix->id_stride >= ix->n) return cos_fail(COS_ERR_INVALID, "replica id %u has no resident vector", node_ids[i]);
    }
    std::vector<float> mags(n_nodes);
    for (u32 i = 0; i < n_nodes; i++) { // Metadata::from (types.rs:112-126): sqrt of the sequential sum of squares
        float acc = -0.0f;
        for (u32 j = 0; j < md; j++) { const float x = (float)mbits[(size_t)i * md + j]; acc = acc + x * x; }
        mags[i] = sqrtf(acc);
    }
    if (ix->meta.d_mbits) (void)hipFree(ix->meta.d_mbits);
    if (ix->meta.d_mmags) (void)hipFree(ix->meta.d_mmags);
    ix->meta.d_mbits = nullptr; ix->meta.d_mmags = nullptr;
    for (auto &l : ix->meta.lv) free_level(l);
    HIP_TRY(hipMalloc((void **)&ix->meta.d_mbits, (size_t)n_nodes * md * 4));
    HIP_TRY(hipMalloc((void **)&ix->meta.d_mmags, (size_t)n_nodes * 4));
    HIP_TRY(hipMemcpy(ix->meta.d_mbits, mbits, (size_t)n_nodes * md * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ix->meta.d_mmags, mags.data(), (size_t)n_nodes * 4, hipMemcpyHostToDevice));
    ix->meta.node_ids.assign(node_ids, node_ids + n_nodes);
    return COS_OK;
}

extern "C" int32_t cos_index_upload_meta_graph_level(cos_index *ix, uint32_t level, uint32_t n_nodes, const uint32_t *node_ids, const uint32_t *nbr_ids) {
    if (!ix || !node_ids || !nbr_ids || n_nodes == 0) return cos_fail(COS_ERR_INVALID, "null/empty level");
    if (!ix->meta.mdim || !ix->have_vectors || ix->meta.node_ids.empty()) return cos_fail(COS_ERR_NOT_READY, "cos_index_enable_metadata + cos_index_upload_meta_nodes first");
    if (level > ix->p.num_layers) return cos_fail(COS_ERR_INVALID, "bad level");
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    LevelHost &L = ix->meta.lv[level];
    const u32 M = L.M, n = ix->n;
    const std::vector<u32> &tab = ix->meta.node_ids;
    if (node_ids[0] != PSEUDO_LO && !std::binary_search(node_ids, node_ids + n_nodes, PSEUDO_LO)) return cos_fail(COS_ERR_INVALID, "level %u: the pseudo root (u32::MAX - 257) must be a node of every level", level);
    std::vector<u32> adj_vec((size_t)n_nodes * M), adj_node((size_t)n_nodes * M), node_vec(n_nodes), node_meta(n_nodes);
    u32 root_idx = 0;
    for (u32 i = 0; i < n_nodes; i++) {
        if (i && node_ids[i] <= node_ids[i - 1]) return cos_fail(COS_ERR_INVALID, "level %u: node_ids must be strictly ascending", level);
        auto it = std::lower_bound(tab.begin(), tab.end(), node_ids[i]);
        if (it == tab.end() || *it != node_ids[i]) return cos_fail(COS_ERR_INVALID, "level %u: node %u is not in the node table", level, node_ids[i]);
        node_meta[i] = (u32)(it - tab.begin());
        const bool pseudo = node_ids[i] >= PSEUDO_LO && node_ids[i] <= PSEUDO_HI;
        if (!pseudo && node_ids[i] / ix->id_stride >= n) return cos_fail(COS_ERR_INVALID, "level %u: replica id %u has no resident vector", level, node_ids[i]);
        node_vec[i] = pseudo ? n + 1 : node_ids[i] / ix->id_stride;
        if (node_ids[i] == PSEUDO_LO) root_idx = i;
    }
    for (u32 i = 0; i < n_nodes; i++)
        for (u32 j = 0; j < M; j++) {
            const u32 id = nbr_ids[(size_t)i * M + j];
            if (id == COS_SLOT_EMPTY) { adj_vec[(size_t)i * M + j] = adj_node[(size_t)i * M + j] = ROW_EMPTY; continue; }
            const uint32_t *it = std::lower_bound(node_ids, node_ids + n_nodes, id);
            if (it == node_ids + n_nodes || *it != id) return cos_fail(COS_ERR_INVALID, "level %u: neighbour id %u of node %u is not a node of this level", level, id, node_ids[i]);
            adj_node[(size_t)i * M + j] = (u32)(it - node_ids);
            adj_vec[(size_t)i * M + j] = node_vec[it - node_ids];
        }
    free_level(L);
    L.M = M;
    auto up = [&](u32 *&dst, const std::vector<u32> &src) -> hipError_t {
        hipError_t e = hipMalloc((void **)&dst, std::max<size_t>(src.size(), 1) * 4);
        return e == hipSuccess ? hipMemcpy(dst, src.data(), src.size() * 4, hipMemcpyHostToDevice) : e;
    };
    std::vector<u32> ids(node_ids, node_ids + n_nodes);
    HIP_TRY(up(L.d_adj_vec, adj_vec));
    HIP_TRY(up(L.d_adj_node, adj_node));
    HIP_TRY(up(L.d_node_vec, node_vec));
    HIP_TRY(up(L.d_node_id, ids));
    HIP_TRY(up(L.d_node_meta, node_meta));
    L.node_ids = std::move(ids);
    L.n = n_nodes;
    L.root_idx = root_idx;
    // child links: the same replica one level down; resolved for this level and the one above once both are resident
    for (u32 lv = level; lv <= std::min(level + 1, ix->p.num_layers); lv++) {
        if (lv == 0) continue;
        LevelHost &U = ix->meta.lv[lv], &D = ix->meta.lv[lv - 1];
        if (U.n == 0 || D.n == 0) continue;
        std::vector<u32> child(U.n);
        for (u32 i = 0; i < U.n; i++) {
            auto it = std::lower_bound(D.node_ids.begin(), D.node_ids.end(), U.node_ids[i]);
            if (it == D.node_ids.end() || *it != U.node_ids[i]) return cos_fail(COS_ERR_INVALID, "node %u of level %u is missing on level %u", U.node_ids[i], lv, lv - 1);
            child[i] = (u32)(it - D.node_ids.begin());
        }
        if (U.d_child) (void)hipFree(U.d_child);
        U.d_child = nullptr;
        HIP_TRY(up(U.d_child, child));
    }
    return COS_OK;
}

static bool meta_ready(const cos_index *ix) {
    if (!ix->meta.mdim || ix->meta.lv.size() != ix->p.num_layers + 1) return false;
    for (u32 l = 0; l <= ix->p.num_layers; l++)
        if (ix->meta.lv[l].n == 0 || (l > 0 && !ix->meta.lv[l].d_child)) return false;
    return ix->have_vectors;
}

// quantize -> filtered walk (pseudo-root component) -> [finalize]; host buffers.  per-level lists when out_lvl_* are given.
static int32_t search_filtered_host(cos_index *ix, const float *queries, u32 B, const u32 *f_off, const int8_t *f_dims, u32 top_k, u32 *out_ids,
                                    float *out_scores, u32 *out_counts, int32_t *out_status, u32 *lvl_ids, float *lvl_sims, u32 *lvl_counts) {
    if (!ix || !queries || !f_off || !f_dims || B == 0) return cos_fail(COS_ERR_INVALID, "bad argument");
    if (!meta_ready(ix)) return cos_fail(COS_ERR_NOT_READY, "the metadata component needs its node table and every graph level before a filtered search");
    u32 ef, vmode;
    { std::lock_guard<std::mutex> g(ix->mu); ef = ix->p.ef_search; vmode = ix->p.visited_mode; }
    if (vmode != COS_VISITED_REF) return cos_fail(COS_ERR_UNIMPLEMENTED, "filtered search implements the reference's PerformantFixedSet filter only");
    if (ef > cosdev::WALK_FAST_MAX_EF) return cos_fail(COS_ERR_UNIMPLEMENTED, "filtered search: ef_search <= %u", cosdev::WALK_FAST_MAX_EF);
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    const u32 md = ix->meta.mdim, nf = f_off[B], L1 = ix->p.num_layers + 1;
    for (u32 b = 0; b < B; b++)
        if (f_off[b + 1] < f_off[b] || f_off[b + 1] == f_off[b]) return cos_fail(COS_ERR_INVALID, "query %u has no filter (an unfiltered query goes through cos_search_batch)", b);
    std::vector<int32_t> fd((size_t)nf * md);
    std::vector<float> fm(nf);
    for (u32 f = 0; f < nf; f++) { // Metadata::from(&QueryFilterDimensions) (types.rs:128-147)
        float acc = -0.0f;
        for (u32 j = 0; j < md; j++) { const int32_t v = f_dims[(size_t)f * md + j]; fd[(size_t)f * md + j] = v; const float x = (float)v; acc = acc + x * x; }
        fm[f] = sqrtf(acc);
    }
    PipeLease lease(ix);
    rc = lease.acquire();
    if (rc) return rc;
    hipStream_t st = lease.hp->s[0];
    struct Drain { hipStream_t st; ~Drain() { (void)hipStreamSynchronize(st); } } drain{st};
    Workspace *w;
    rc = get_workspace(ix, (void *)st, st, B, top_k ? top_k : 1, true, &w);
    if (rc) return rc;
    // the batch's filters in the workspace's own buffer, ONE copy: [dims nf x md | norms nf | offsets B + 1]
    const size_t f_words = fd.size() + fm.size() + (size_t)B + 1;
    if (f_words * 4 > w->f_cap) {
        HIP_TRY(hipStreamSynchronize(st));
        if (w->f_buf) (void)hipFree(w->f_buf);
        w->f_buf = nullptr;
        w->f_cap = 0;
        HIP_TRY(hipMalloc((void **)&w->f_buf, f_words * 4 * 2));
        w->f_cap = f_words * 4 * 2;
    }
    std::vector<u32> fpack(f_words);
    memcpy(fpack.data(), fd.data(), fd.size() * 4);
    memcpy(fpack.data() + fd.size(), fm.data(), fm.size() * 4);
    memcpy(fpack.data() + fd.size() + fm.size(), f_off, ((size_t)B + 1) * 4);
    HIP_TRY(hipMemcpyAsync(w->f_buf, fpack.data(), f_words * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(w->d_queries, queries, (size_t)B * ix->p.dim * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(launch_quantize_rows(ix->eng, w->d_queries, ix->p.dim, B, ix->p.dim, ix->p.range_lo, ix->p.range_hi, w->q_codes, ix->row_stride, w->q_mags,
                                 w->q_raw_mags, st));
    IndexDev dev = cos_make_meta_dev(ix);
    dev.visited_mode = COS_VISITED_REF;
    WalkArgs wa;
    memset(&wa, 0, sizeof(wa));
    wa.qcodes = w->q_codes;
    wa.qmags = w->q_mags;
    wa.B = B;
    wa.ef = ef;
    wa.keep = KEEP_SEARCH;
    wa.out_ids = w->walk_ids;
    wa.out_sims = w->walk_sims;
    wa.out_counts = w->walk_counts;
    wa.out_status = w->walk_status;
    wa.f_dims = (const int32_t *)w->f_buf;
    wa.f_mags = (const float *)(w->f_buf + fd.size() * 4);
    wa.f_off = (const u32 *)(w->f_buf + (fd.size() + fm.size()) * 4);
    HIP_TRY(launch_walk_meta(ix->eng, dev, wa, st));
    std::vector<int32_t> status(B);
    if (lvl_ids) {
        HIP_TRY(hipMemcpyAsync(lvl_ids, w->walk_ids, (size_t)B * L1 * KEEP_SEARCH * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(lvl_sims, w->walk_sims, (size_t)B * L1 * KEEP_SEARCH * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(lvl_counts, w->walk_counts, (size_t)B * L1 * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(status.data(), w->walk_status, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    } else {
        HIP_TRY(launch_finalize(dev, w->d_queries, ix->p.dim, w->q_raw_mags, w->walk_ids, w->walk_sims, w->walk_counts, w->walk_status, B, top_k, w->d_out_ids,
                                w->d_out_scores, w->d_out_counts, w->d_out_status, w->rerank_rows, st));
        HIP_TRY(hipMemcpyAsync(out_ids, w->d_out_ids, (size_t)B * top_k * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(out_scores, w->d_out_scores, (size_t)B * top_k * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(out_counts, w->d_out_counts, (size_t)B * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(status.data(), w->d_out_status, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    if (out_status) memcpy(out_status, status.data(), (size_t)B * 4);
    for (u32 b = 0; b < B; b++)
        if (status[b] != COS_OK) return cos_fail(status[b], "filtered query %u failed with status %d", b, status[b]);
    return COS_OK;
}

extern "C" int32_t cos_search_filtered_batch(cos_index *ix, const float *queries, uint32_t B, const uint32_t *filter_offsets, const int8_t *filter_dims,
                                             uint32_t top_k, uint32_t *out_ids, float *out_scores, uint32_t *out_counts, int32_t *out_status) {
    if (!out_ids || !out_scores || !out_counts || top_k == 0 || top_k > 1024) return cos_fail(COS_ERR_INVALID, "bad argument");
    return search_filtered_host(ix, queries, B, filter_offsets, filter_dims, top_k, out_ids, out_scores, out_counts, out_status, nullptr, nullptr, nullptr);
}

extern "C" int32_t cos_ann_search_filtered_batch(cos_index *ix, const float *queries, uint32_t B, const uint32_t *filter_offsets, const int8_t *filter_dims,
                                                 uint32_t *out_ids, float *out_sims, uint32_t *out_counts, int32_t *out_status) {
    if (!out_ids || !out_sims || !out_counts) return cos_fail(COS_ERR_INVALID, "null output");
    return search_filtered_host(ix, queries, B, filter_offsets, filter_dims, 0, nullptr, nullptr, nullptr, out_status, out_ids, out_sims, out_counts);
}

// ------------------------------------------------------------------------------------------------
// QuantizationMetric::quantize operator
// ------------------------------------------------------------------------------------------------
extern "C" int32_t cos_quantize_batch(uint32_t storage, uint32_t resolution, uint32_t dim, float lo, float hi, const float *x, uint32_t n,
                                      void *codes, float *mags) {
    if (!x || !codes || !mags || dim == 0 || n == 0) return cos_fail(COS_ERR_INVALID, "bad argument");
    int eng;
    u64 row_stride;
    if (storage == COS_STORAGE_U8) { eng = ENG_U8; row_stride = ((u64)dim + 15) & ~15ull; }
    else if (storage == COS_STORAGE_SUBBYTE && resolution == 1) { eng = ENG_Q1; row_stride = (u64)((dim + 127) / 128) * 16; }
    else if (storage == COS_STORAGE_SUBBYTE && resolution == 2) { eng = ENG_Q2; row_stride = (u64)((dim + 63) / 64) * 16; }
    else if (storage == COS_STORAGE_SUBBYTE && resolution == 3) { eng = ENG_Q3; row_stride = (u64)((dim + 31) / 32) * 16; }
    else if (storage == COS_STORAGE_F32) { eng = ENG_F32; row_stride = ((u64)dim * 4 + 15) & ~15ull; }
    else if (storage == COS_STORAGE_F16 || (storage == COS_STORAGE_SUBBYTE && resolution >= 1 && resolution <= 8)) { eng = -1; row_stride = 0; }
    else return cos_fail(COS_ERR_INVALID, "unknown storage kind / resolution");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return cos_fail(COS_ERR_NO_DEVICE, "no HIP device visible; the GPU path has no CPU fallback");
    if (eng < 0) return quantize_ref_layout(storage, resolution, dim, x, n, codes, mags);
    DevBuf bx, bm, bc;
    HIP_TRY(bx.alloc((size_t)n * dim * 4));
    HIP_TRY(bm.alloc((size_t)n * 4));
    HIP_TRY(bc.alloc((size_t)n * row_stride));
    HIP_TRY(hipMemcpy(bx.p, x, (size_t)n * dim * 4, hipMemcpyHostToDevice));
    HIP_TRY(launch_quantize_rows(eng, bx.as<float>(), dim, n, dim, lo, hi, bc.as<uint8_t>(), row_stride, bm.as<float>(), nullptr, 0));
    std::vector<uint8_t> dev((size_t)n * row_stride);
    HIP_TRY(hipMemcpy(dev.data(), bc.p, dev.size(), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(mags, bm.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    const size_t cb = cos_code_bytes(storage, resolution, dim);
    for (size_t r = 0; r < n; r++) row_to_reference_layout(eng, dim, dev.data() + r * row_stride, (uint8_t *)codes + r * cb);
    return COS_OK;
}


// ---- the pseudo-root component back to the host (what cos_index_build_meta built, or what was uploaded) ------------------------------
extern "C" int32_t cos_index_meta_level_count(cos_index *ix, uint32_t level, uint32_t *out) {
    if (!ix || !out) return cos_fail(COS_ERR_INVALID, "null argument");
    if (level > ix->p.num_layers || ix->meta.lv.size() != ix->p.num_layers + 1) return cos_fail(COS_ERR_INVALID, "bad level / no metadata component");
    *out = ix->meta.lv[level].n;
    return COS_OK;
}

extern "C" int32_t cos_index_download_meta_graph_level(cos_index *ix, uint32_t level, uint32_t *node_ids, uint32_t *nbr_ids) {
    if (!ix || !node_ids || !nbr_ids) return cos_fail(COS_ERR_INVALID, "null argument");
    if (level > ix->p.num_layers || ix->meta.lv.size() != ix->p.num_layers + 1) return cos_fail(COS_ERR_INVALID, "bad level / no metadata component");
    const LevelHost &L = ix->meta.lv[level];
    if (L.n == 0 || !L.d_adj_node) return cos_fail(COS_ERR_NOT_READY, "level %u of the metadata component is not resident", level);
    int32_t rc = cos_set_device(ix);
    if (rc) return rc;
    std::vector<u32> adj((size_t)L.n * L.M);
    HIP_TRY(hipMemcpy(adj.data(), L.d_adj_node, adj.size() * 4, hipMemcpyDeviceToHost));
    for (u32 i = 0; i < L.n; i++) {
        node_ids[i] = L.node_ids[i];
        for (u32 j = 0; j < L.M; j++) {
            const u32 nb = adj[(size_t)i * L.M + j];
            nbr_ids[(size_t)i * L.M + j] = nb == ROW_EMPTY ? COS_SLOT_EMPTY : L.node_ids[nb];
        }
    }
    return COS_OK;
}
