// kernels_hybrid.hip — BM25 scoring over CSR postings and reciprocal-rank fusion (config c5).
//   SparseAnnQueryBasic::search_bm25   models/sparse_ann_query.rs:149-233
//   get_idf                            models/sparse_ann_query.rs:298-302 (ln_1p on the HOST libm, like the reference)
//   RRF fusion of hybrid_search        api/vectordb/search/repo.rs:311-340
//
// BM25 is HBM-bound streaming work: every posting (8 B: doc id + stored tf) of every query term is read
// once, coalesced.  The reference's document-at-a-time heap merge is restated as a tiled term-at-a-time
// accumulation that produces bit-identical f32 sums: the doc-id space is cut into tiles of TILE ids; inside
// a tile the query's terms are applied in ASCENDING TERM-HASH order (the documented order, oracle/…bm25.c)
// with a workgroup barrier between terms, so each document's score is p0, then +p1, then +p2 … exactly like
// the sequential merge.  Postings of one term are distinct documents, so lanes never collide inside a term.
// The 512 result buckets (doc_id % 512, strictly-greater score wins, first seen = smallest id on ties) are
// an order-independent max over the key (score, ~doc_id): one 64-bit LDS/global atomic max.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "engine_internal.h"

using namespace cosdev;

namespace {

constexpr u32 BUCKETS = 512;  // sparse_ann_query.rs:154
constexpr u32 TILE = 8192;    // doc ids per LDS accumulator tile (32 KB of f32)
constexpr u32 MAX_QTERMS = 64;
constexpr u32 DIR_MIN = 256;  // posting lists longer than this get a tile directory; shorter ones are scanned whole per tile
constexpr u32 NO_DIR = 0xFFFFFFFFu;
constexpr int PU = 8;       // postings per thread per chunk

struct QueryTerms { // per query, terms ascending by hash, only those that have a posting list
    u64 begin[MAX_QTERMS];
    u64 end[MAX_QTERMS];
    float idf[MAX_QTERMS];
    u32 dir[MAX_QTERMS]; // row of the term in the tile directory, NO_DIR for short lists
    u32 n;
};

__device__ __forceinline__ u64 lower_bound_doc(const u32 *__restrict__ docs, u64 lo, u64 hi, u32 key) {
    while (lo < hi) {
        u64 mid = lo + (hi - lo) / 2;
        if (docs[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// UNTOUCHED marks a document no term has reached yet: a NaN bit pattern that tf * idf and the sums of such products cannot take
// (cos_bm25_create rejects non-finite stored term frequencies; idf is finite), so the accumulator itself says whether the first
// posting assigns (p0) or a later one adds (+ p1 ...).  It replaces a separate bitmap whose bits were set with LDS atomics: the
// postings of a dense term are consecutive documents, so up to 32 lanes of a wave hit the SAME bitmap word per instruction and the
// hardware serialises same-address atomics — that, not HBM or the barriers, held the kernel at ~1.2 ms per batch on c5 (a
// barrier-free variant with wave-owned 2048-document tiles measured the same 1.3 ms with the bitmap and 0.75 ms without it,
// against 0.66 ms for this one: dropped).
constexpr u32 UNTOUCHED = 0xFFFFFFFFu;

// apply one chunk (PU postings per lane, all of ONE term: distinct documents, so the PU read-modify-writes of a lane and those of
// the other lanes never touch the same slot and the reads can all be issued before the first write)
template <u32 N>
__device__ __forceinline__ void bm25_apply_chunk(float *acc, u32 d0, float idf, const u32 (&dv)[PU], const float (&tv)[PU], u32 mask) {
    float old[PU];
    bool ok[PU];
#pragma unroll
    for (int u = 0; u < PU; u++) {
        const u32 slot = dv[u] - d0; // out of range (another tile of a short list) wraps to >= N
        ok[u] = ((mask >> u) & 1u) && dv[u] >= d0 && slot < N;
        old[u] = acc[slot & (N - 1)];
    }
#pragma unroll
    for (int u = 0; u < PU; u++) {
        if (ok[u]) {
            const float p = __fmul_rn(tv[u], idf); // tf * head.idf
            acc[(dv[u] - d0) & (N - 1)] = __float_as_uint(old[u]) != UNTOUCHED ? __fadd_rn(old[u], p) : p;
        }
    }
}

// grid = B * splits blocks: block (q, s) owns the tiles s, s+splits, s+2*splits, ...
// Tile directory: for every posting list longer than DIR_MIN, tile_dir[row][t] = offset (relative to the list's begin) of the
// first posting whose doc id is >= t * TILE, t = 0 .. n_tiles.  It replaces the two ~17-step binary searches every
// (query, tile, term) step used to make — the kernel was latency-bound on them at 0.16 of the HBM roof.
//
// Software pipeline.  A block walks a flat sequence of CHUNKS — (tile, term, PU * 256 consecutive postings of the term's slice
// of the tile) — and always has the NEXT chunk's postings in flight (registers) while it applies the current one to the LDS
// accumulators, across term barriers and tile flushes alike.  Before, a thread's 4 loads were issued, waited for and applied, so
// a CU had ~16 KB in flight half of the time: 2.5 TB/s is what Little's law gives for that at ~1.5 us of loaded HBM latency
// (profiles/archive/r02_c5_hybrid_1M_tile_directory.json).  Now 2 x 16 KB per block, 4 blocks per CU.

struct Bm25Cursor { // block-uniform
    u32 tile, t;
    u64 base, e; // postings [base, min(base + PU * 256, e)) of term t's slice of the tile
    bool valid;
};

__global__ __launch_bounds__(256) void bm25_score_kernel(const u32 *__restrict__ docs, const float *__restrict__ tfs,
                                                         const QueryTerms *__restrict__ qts, u32 n_docs, const u32 *__restrict__ tile_dir,
                                                         u64 *__restrict__ buckets /*[B][512]*/, const u32 *__restrict__ order, u32 splits) {
    __shared__ float acc[TILE]; // UNTOUCHED (a NaN pattern no score can take) until a term reaches the document
    __shared__ u64 lb[BUCKETS];
    // 1-D grid, heaviest queries first: block id -> (rank in the host's descending-postings order, split).  Query sizes are
    // heavy-tailed (a few Zipf-head terms decide everything), so the blocks of the heaviest queries must not be the last to start.
    const u32 q = order[blockIdx.x / splits];
    const u32 split = blockIdx.x % splits;
    const QueryTerms *qt = &qts[q];
    const u32 nt = qt->n;
    if (nt == 0) return;
    const u32 n_tiles = (n_docs + TILE - 1) / TILE;
    if (split >= n_tiles) return;
    for (u32 i = threadIdx.x; i < BUCKETS; i += blockDim.x) lb[i] = 0ull;
    for (u32 i = threadIdx.x; i < TILE; i += blockDim.x) acc[i] = __uint_as_float(UNTOUCHED);

    auto slice = [&](u32 tile, u32 t, u64 &b, u64 &e) { // term t's postings inside the tile
        const u32 dr = qt->dir[t];
        b = qt->begin[t];
        e = qt->end[t];
        if (dr != NO_DIR) { // straight from the directory
            const u32 *row = tile_dir + (u64)dr * (n_tiles + 1);
            e = b + row[tile + 1];
            b = b + row[tile];
        } // else: a short list is scanned whole and filtered by range (no search at all)
    };
    auto advance = [&](const Bm25Cursor &c) -> Bm25Cursor {
        Bm25Cursor n = c;
        if (c.base + (u64)PU * 256 < c.e) { n.base = c.base + (u64)PU * 256; return n; }
        if (c.t + 1 < nt) n.t = c.t + 1;
        else { n.t = 0; n.tile = c.tile + splits; }
        n.valid = n.tile < n_tiles;
        if (n.valid) slice(n.tile, n.t, n.base, n.e);
        return n;
    };
    // Every load is issued unconditionally (masked lanes read posting 0 and drop it): a fixed number of loads per chunk lets the
    // compiler wait for exactly the older chunk (s_waitcnt vmcnt(2 * PU)) while the newer one stays in flight; with predicated
    // loads it had to drain the queue (vmcnt(0)) before touching the current chunk.
    // The loaded registers are not touched here (the lane mask travels separately): any use would wait for the data.
    auto fetch = [&](const Bm25Cursor &c, u32 (&dv)[PU], float (&tv)[PU]) -> u32 {
        u32 mask = 0;
#pragma unroll
        for (int u = 0; u < PU; u++) {
            const u64 i = c.base + threadIdx.x + (u64)u * 256;
            const bool in = c.valid && i < c.e;
            const u64 ii = in ? i : 0ull;
            dv[u] = docs[ii];
            tv[u] = tfs[ii];
            mask |= (in ? 1u : 0u) << u;
        }
        return mask;
    };
    // apply chunk c (registers dv/tv); nx = the chunk after it (already in flight)
    auto apply = [&](const Bm25Cursor &c, const Bm25Cursor &nx, const u32 (&dv)[PU], const float (&tv)[PU], const u32 mask) {
        const u32 d0 = c.tile * TILE;
        const float idf = qt->idf[c.t];
        bm25_apply_chunk<TILE>(acc, d0, idf, dv, tv, mask);
        const bool term_done = !nx.valid || nx.tile != c.tile || nx.t != c.t;
        const bool tile_done = !nx.valid || nx.tile != c.tile;
        if (term_done) __syncthreads(); // a document's score is p0, then + p1, then + p2 ... in term order
        if (tile_done) {
            for (u32 slot = threadIdx.x; slot < TILE; slot += blockDim.x) {
                const float v = acc[slot];
                if (__float_as_uint(v) != UNTOUCHED) {
                    const u32 doc = d0 + slot;
                    const u64 key = pack_key(simkey(v), ~doc); // larger score, then smaller doc id
                    atomicMax((unsigned long long *)&lb[doc % BUCKETS], (unsigned long long)key);
                    acc[slot] = __uint_as_float(UNTOUCHED);
                }
            }
            __syncthreads();
        }
    };

    Bm25Cursor cur;
    cur.tile = split; cur.t = 0; cur.valid = true;
    slice(cur.tile, 0, cur.base, cur.e);
    u32 da[PU], db[PU];
    float ta[PU], tb[PU];
    u32 ma = fetch(cur, da, ta), mb;
    __syncthreads();
    for (;;) { // ping-pong between the two register sets: no copies, so nothing waits on the chunk in flight
        const Bm25Cursor n1 = advance(cur);
        mb = fetch(n1, db, tb);
        apply(cur, n1, da, ta, ma);
        if (!n1.valid) break;
        const Bm25Cursor n2 = advance(n1);
        ma = fetch(n2, da, ta);
        apply(n1, n2, db, tb, mb);
        if (!n2.valid) break;
        cur = n2;
    }
    // a query's ~100 blocks all fold into the same 512 global buckets: look before the atomic (a stale read only costs a
    // redundant atomicMax, never a lost one) — a bucket's running maximum is raised ~ln(blocks) times, not `blocks` times
    for (u32 i = threadIdx.x; i < BUCKETS; i += blockDim.x) {
        const u64 v = lb[i];
        if (v && v > __hip_atomic_load(&buckets[(u64)q * BUCKETS + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax((unsigned long long *)&buckets[(u64)q * BUCKETS + i], (unsigned long long)v);
    }
}

// one wave per query: 512 buckets -> sort by (score desc, larger id first) -> top k
__global__ __launch_bounds__(64) void bm25_topk_kernel(const u64 *__restrict__ buckets, u32 B, u32 top_k, u32 *__restrict__ out_ids,
                                                       float *__restrict__ out_scores, u32 *__restrict__ out_counts) {
    const int lane = threadIdx.x;
    const u32 q = blockIdx.x;
    if (q >= B) return;
    u64 k[8];
    u32 cnt = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const u64 v = buckets[(u64)q * BUCKETS + (u32)lane * 8 + r];
        k[r] = v ? pack_key((u32)(v >> 32), ~(u32)v) : 0ull; // back to (score, doc id)
        cnt += v != 0ull;
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) cnt += (u32)__shfl_xor((int)cnt, m, 64);
    bitonic_sort_desc<8>(k, lane);
    const u32 n = cnt < top_k ? cnt : top_k;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const u32 e = (u32)lane * 8 + r;
        if (e < n) {
            out_ids[(u64)q * top_k + e] = (u32)k[r];
            out_scores[(u64)q * top_k + e] = simkey_inv((u32)(k[r] >> 32));
        }
    }
    if (lane == 0) out_counts[q] = n;
}

// RRF: one wave per query.  score(id) = [last dense occurrence: 1/(rank+k+eps)] then += each sparse occurrence.
template <int R>
__global__ __launch_bounds__(64) void rrf_kernel(const u32 *__restrict__ dense_ids, const u32 *__restrict__ dense_counts, u32 dense_stride,
                                                 const u32 *__restrict__ sparse_ids, const u32 *__restrict__ sparse_counts, u32 sparse_stride, u32 B,
                                                 float kc, u32 top_k, u32 *__restrict__ out_ids, float *__restrict__ out_scores,
                                                 u32 *__restrict__ out_counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32 *ids = (u32 *)smem_raw; // [nd + ns]
    const int lane = threadIdx.x;
    const u32 q = blockIdx.x;
    if (q >= B) return;
    const u32 nd = dense_counts[q], ns = sparse_counts[q], n = nd + ns;
    for (u32 i = lane; i < n; i += 64) ids[i] = i < nd ? dense_ids[(u64)q * dense_stride + i] : sparse_ids[(u64)q * sparse_stride + (i - nd)];
    __builtin_amdgcn_wave_barrier();
    u64 key[R];
    u32 cnt = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const u32 e = (u32)lane * R + r;
        key[r] = 0ull;
        if (e < n) {
            const u32 id = ids[e];
            bool first = true;
            for (u32 j = 0; j < e; j++) first &= ids[j] != id;
            if (first) { // the first occurrence owns the id
                float score = 0.0f;
                for (u32 j = 0; j < nd; j++)
                    if (ids[j] == id) score = __fdiv_rn(1.0f, __fadd_rn(__fadd_rn((float)j, kc), 1.1920929e-07f)); // insert() overwrites
                for (u32 j = nd; j < n; j++)
                    if (ids[j] == id) score = __fadd_rn(score, __fdiv_rn(1.0f, __fadd_rn(__fadd_rn((float)(j - nd), kc), 1.1920929e-07f)));
                key[r] = pack_key(simkey(score), id);
                cnt++;
            }
        }
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) cnt += (u32)__shfl_xor((int)cnt, m, 64);
    bitonic_sort_desc<R>(key, lane);
    const u32 nout = cnt < top_k ? cnt : top_k;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const u32 e = (u32)lane * R + r;
        if (e < nout) {
            out_ids[(u64)q * top_k + e] = (u32)key[r];
            out_scores[(u64)q * top_k + e] = simkey_inv((u32)(key[r] >> 32));
        }
    }
    if (lane == 0) out_counts[q] = nout;
}

} // namespace

struct cos_bm25 {
    int32_t device = 0;
    u32 n_terms = 0, documents_count = 0, max_doc = 0;
    std::vector<u32> term_hashes;
    std::vector<u64> offsets;
    u32 *d_docs = nullptr;
    float *d_tfs = nullptr;
    std::vector<u32> dir_row; // [n_terms] row in the tile directory or NO_DIR
    u32 *d_tile_dir = nullptr; // [rows][n_tiles + 1]
    // per-handle workspace of the search (grown on demand, reused across calls: no allocation on the query path)
    std::mutex mu;
    QueryTerms *d_qt = nullptr, *h_qt = nullptr; // device / pinned host; followed by the launch order u32[capB] in the same allocation
    u64 *d_buckets = nullptr;
    u32 *d_ids = nullptr, *d_cnt = nullptr;
    float *d_sc = nullptr;
    u32 capB = 0, cap_k = 0;
    hipStream_t stream = nullptr;
    // cos_hybrid_search_batch: dense half + fusion (second stream, buffers grown on demand)
    hipStream_t stream_dense = nullptr;
    hipEvent_t ev_sparse = nullptr;
    float *d_hq = nullptr, *d_dsc = nullptr;
    u32 *d_did = nullptr, *d_dcnt = nullptr;
    // what goes back to the caller, side by side for ONE copy: [fused ids B x k | fused scores B x k | counts B | dense status B]
    u32 *d_ret = nullptr;
    void *h_ret = nullptr; // its pinned landing area
    // one capacity per buffer, written back by grow_buf itself: a failed hipMalloc leaves that buffer null WITH capacity 0,
    // so a later, smaller batch grows it again instead of launching on a null pointer
    size_t cap_hq = 0, cap_did = 0, cap_dsc = 0, cap_dcnt = 0, cap_ret = 0, cap_hret = 0;
};

extern "C" int32_t cos_bm25_create(int32_t device, const uint32_t *term_hashes, const uint64_t *offsets, uint32_t n_terms, const uint32_t *doc_ids,
                                   const float *tfs, uint32_t documents_count, cos_bm25 **out) {
    if (!term_hashes || !offsets || !doc_ids || !tfs || !out || n_terms == 0) return cos_fail(COS_ERR_INVALID, "bad argument");
    *out = nullptr;
    for (u32 t = 1; t < n_terms; t++)
        if (term_hashes[t] <= term_hashes[t - 1]) return cos_fail(COS_ERR_INVALID, "term hashes must be strictly ascending");
    for (u64 i = 0; i < offsets[n_terms]; i++) // compute_bm25_term_frequency (indexes/tf_idf/mod.rs:362-371) of a count is always finite
        if (!std::isfinite(tfs[i])) return cos_fail(COS_ERR_INVALID, "stored term frequency %llu is not finite", (unsigned long long)i);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return cos_fail(COS_ERR_NO_DEVICE, "no HIP device visible; the GPU path has no CPU fallback");
    HIP_TRY(hipSetDevice(device));
    cos_bm25 *b = new cos_bm25();
    b->device = device;
    b->n_terms = n_terms;
    b->documents_count = documents_count;
    b->term_hashes.assign(term_hashes, term_hashes + n_terms);
    b->offsets.assign(offsets, offsets + n_terms + 1);
    const u64 nnz = offsets[n_terms];
    for (u32 t = 0; t < n_terms; t++)
        if (offsets[t + 1] > offsets[t]) b->max_doc = std::max(b->max_doc, doc_ids[offsets[t + 1] - 1]); // lists are doc-id ascending
    // tile directory of the long posting lists (one pass over their postings on the host)
    const u32 n_tiles = (b->max_doc + 1 + TILE - 1) / TILE;
    b->dir_row.assign(n_terms, NO_DIR);
    std::vector<u32> dir;
    u32 rows = 0;
    for (u32 t = 0; t < n_terms; t++) {
        const u64 lo = offsets[t], hi = offsets[t + 1];
        if (hi - lo <= DIR_MIN) continue;
        if (hi - lo > 0xFFFFFFFFull) { cos_bm25_destroy(b); return cos_fail(COS_ERR_UNIMPLEMENTED, "posting list of term %u too long", term_hashes[t]); }
        b->dir_row[t] = rows++;
        const size_t base = dir.size();
        dir.resize(base + n_tiles + 1);
        u64 p = lo;
        for (u32 tile = 0; tile <= n_tiles; tile++) {
            const u64 bound = (u64)tile * TILE;
            while (p < hi && doc_ids[p] < bound) p++;
            dir[base + tile] = (u32)(p - lo);
        }
        dir[base + n_tiles] = (u32)(hi - lo);
    }
    hipError_t e = hipMalloc(&b->d_docs, std::max<u64>(nnz, 1) * 4);
    if (e == hipSuccess) e = hipMalloc(&b->d_tfs, std::max<u64>(nnz, 1) * 4);
    if (e == hipSuccess) e = hipMalloc(&b->d_tile_dir, std::max<size_t>(dir.size(), 1) * 4);
    if (e == hipSuccess) e = hipMemcpy(b->d_docs, doc_ids, nnz * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(b->d_tfs, tfs, nnz * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && !dir.empty()) e = hipMemcpy(b->d_tile_dir, dir.data(), dir.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) { cos_bm25_destroy(b); HIP_TRY(e); }
    *out = b;
    return COS_OK;
}

extern "C" int32_t cos_bm25_destroy(cos_bm25 *b) {
    if (!b) return COS_OK;
    (void)hipSetDevice(b->device);
    if (b->stream) { (void)hipStreamSynchronize(b->stream); (void)hipStreamDestroy(b->stream); }
    if (b->stream_dense) { (void)hipStreamSynchronize(b->stream_dense); (void)hipStreamDestroy(b->stream_dense); }
    if (b->ev_sparse) (void)hipEventDestroy(b->ev_sparse);
    void *ptrs[] = {b->d_docs, b->d_tfs, b->d_qt, b->d_buckets, b->d_ids, b->d_cnt, b->d_sc, b->d_tile_dir,
                    b->d_hq, b->d_dsc, b->d_did, b->d_dcnt, b->d_ret};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (b->h_qt) (void)hipHostFree(b->h_qt);
    if (b->h_ret) (void)hipHostFree(b->h_ret);
    delete b;
    return COS_OK;
}

// host side of a batch: sort each query's terms by hash, look the posting lists up, idf via libm log1pf
// (sparse_ann_query.rs:298-302) -> QueryTerms in pinned memory
static int32_t bm25_prepare(cos_bm25 *b, const uint32_t *q_terms, const uint32_t *q_offsets, u32 B) {
    for (u32 q = 0; q < B; q++) {
        std::vector<u32> t(q_terms + q_offsets[q], q_terms + q_offsets[q + 1]);
        std::sort(t.begin(), t.end());
        QueryTerms &qt = b->h_qt[q];
        qt.n = 0;
        for (u32 h : t) {
            auto it = std::lower_bound(b->term_hashes.begin(), b->term_hashes.end(), h);
            if (it == b->term_hashes.end() || *it != h) continue; // no node / no term: skipped (:165-167)
            if (qt.n == MAX_QTERMS) return cos_fail(COS_ERR_UNIMPLEMENTED, "more than %u matching terms in query %u", MAX_QTERMS, q);
            const size_t ti = (size_t)(it - b->term_hashes.begin());
            const u32 len = (u32)(b->offsets[ti + 1] - b->offsets[ti]);
            qt.begin[qt.n] = b->offsets[ti];
            qt.end[qt.n] = b->offsets[ti + 1];
            qt.idf[qt.n] = log1pf(((float)(u32)(b->documents_count - len) + 0.5f) / ((float)len + 0.5f));
            qt.dir[qt.n] = b->dir_row[ti];
            qt.n++;
        }
    }
    // launch order: heaviest query (most postings) first; ties by index
    std::vector<std::pair<u64, u32>> w(B);
    for (u32 q = 0; q < B; q++) {
        u64 tot = 0;
        for (u32 t = 0; t < b->h_qt[q].n; t++) tot += b->h_qt[q].end[t] - b->h_qt[q].begin[t];
        w[q] = {~tot, q};
    }
    std::sort(w.begin(), w.end());
    u32 *order = (u32 *)(b->h_qt + b->capB);
    for (u32 q = 0; q < B; q++) order[q] = w[q].second;
    return COS_OK;
}

static int32_t bm25_workspace(cos_bm25 *b, u32 B, u32 top_k) {
    if (!b->stream) HIP_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    if (B > b->capB || top_k > b->cap_k) {
        HIP_TRY(hipStreamSynchronize(b->stream));
        const u32 nb = std::max(B, b->capB), nk = std::max(top_k, b->cap_k);
        void *ptrs[] = {b->d_qt, b->d_buckets, b->d_ids, b->d_cnt, b->d_sc};
        for (void *p : ptrs) if (p) (void)hipFree(p);
        if (b->h_qt) (void)hipHostFree(b->h_qt);
        b->d_qt = nullptr; b->h_qt = nullptr; b->d_buckets = nullptr; b->d_ids = nullptr; b->d_cnt = nullptr; b->d_sc = nullptr;
        b->capB = b->cap_k = 0;
        HIP_TRY(hipMalloc((void **)&b->d_qt, (size_t)nb * (sizeof(QueryTerms) + 4)));
        HIP_TRY(hipHostMalloc((void **)&b->h_qt, (size_t)nb * (sizeof(QueryTerms) + 4)));
        HIP_TRY(hipMalloc((void **)&b->d_buckets, (size_t)nb * BUCKETS * 8));
        HIP_TRY(hipMalloc((void **)&b->d_ids, (size_t)nb * nk * 4));
        HIP_TRY(hipMalloc((void **)&b->d_sc, (size_t)nb * nk * 4));
        HIP_TRY(hipMalloc((void **)&b->d_cnt, (size_t)nb * 4));
        b->capB = nb;
        b->cap_k = nk;
    }
    return COS_OK;
}

// scoring + bucket top-k enqueued on `st`; outputs are device pointers
static int32_t bm25_launch(cos_bm25 *b, u32 B, u32 top_k, u32 *d_out_ids, float *d_out_scores, u32 *d_out_counts, hipStream_t st) {
    HIP_TRY(hipMemcpyAsync(b->d_qt, b->h_qt, (size_t)B * sizeof(QueryTerms), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(b->d_qt + b->capB, b->h_qt + b->capB, (size_t)B * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(b->d_buckets, 0, (size_t)B * BUCKETS * 8, st));
    const u32 span = b->max_doc + 1; // doc ids are internal ids; the largest one bounds the tile count
    // launch shape: enough blocks that the heaviest query's share is small against the whole launch, few enough that a block's fixed
    // cost (512 buckets, 8192 accumulators to reset and fold per tile) stays small against its postings.  c5, 256 queries
    // (profiles/archive/r02_c5_bm25_*): 2048 blocks 0.72 ms, 4096 0.67, 8192 0.66, 16384 0.69, 32768 0.89.  COS_BM25_BLOCKS overrides (experiments).
    const u32 target_blocks = (u32)std::max<long long>(1, tune_or(TUNE_BM25_BLOCKS, 8192));
    const u32 n_tiles = (span + TILE - 1) / TILE;
    const u32 splits = std::max(1u, std::min(n_tiles, std::max(1u, target_blocks / B)));
    hipLaunchKernelGGL(bm25_score_kernel, dim3(B * splits), dim3(256), 0, st, b->d_docs, b->d_tfs, b->d_qt, span, b->d_tile_dir, b->d_buckets,
                       (const u32 *)(b->d_qt + b->capB), splits);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(bm25_topk_kernel, dim3(B), dim3(64), 0, st, b->d_buckets, B, top_k, d_out_ids, d_out_scores, d_out_counts);
    HIP_TRY(hipGetLastError());
    return COS_OK;
}

extern "C" int32_t cos_bm25_search_batch_device(cos_bm25 *b, const uint32_t *q_terms, const uint32_t *q_offsets, uint32_t B, uint32_t top_k,
                                                uint32_t *d_out_ids, float *d_out_scores, uint32_t *d_out_counts, void *stream) {
    if (!b || !q_terms || !q_offsets || !d_out_ids || !d_out_scores || !d_out_counts || B == 0 || top_k == 0) return cos_fail(COS_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(b->device));
    std::lock_guard<std::mutex> g(b->mu);
    int32_t rc = bm25_workspace(b, B, top_k);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(b->stream)); // the pinned term table of the previous batch must have been consumed
    rc = bm25_prepare(b, q_terms, q_offsets, B);
    if (rc) return rc;
    hipStream_t st = stream ? (hipStream_t)stream : b->stream;
    rc = bm25_launch(b, B, top_k, d_out_ids, d_out_scores, d_out_counts, st);
    if (rc) return rc;
    if (st != b->stream) HIP_TRY(hipStreamSynchronize(st)); // caller's stream: the pinned table may be reused as soon as we return
    return COS_OK;
}

extern "C" int32_t cos_bm25_search_batch(cos_bm25 *b, const uint32_t *q_terms, const uint32_t *q_offsets, uint32_t B, uint32_t top_k,
                                         uint32_t *out_ids, float *out_scores, uint32_t *out_counts) {
    if (!b || !q_terms || !q_offsets || !out_ids || !out_scores || !out_counts || B == 0 || top_k == 0) return cos_fail(COS_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(b->device));
    std::lock_guard<std::mutex> g(b->mu);
    int32_t rc = bm25_workspace(b, B, top_k);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(b->stream));
    rc = bm25_prepare(b, q_terms, q_offsets, B);
    if (rc) return rc;
    rc = bm25_launch(b, B, top_k, b->d_ids, b->d_sc, b->d_cnt, b->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out_ids, b->d_ids, (size_t)B * top_k * 4, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(out_scores, b->d_sc, (size_t)B * top_k * 4, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipMemcpyAsync(out_counts, b->d_cnt, (size_t)B * 4, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return COS_OK;
}

extern "C" int32_t cos_rrf_fuse_batch(const uint32_t *dense_ids, const uint32_t *dense_counts, uint32_t dense_stride, const uint32_t *sparse_ids,
                                      const uint32_t *sparse_counts, uint32_t sparse_stride, uint32_t B, float fusion_constant_k, uint32_t top_k,
                                      uint32_t *out_ids, float *out_scores, uint32_t *out_counts) {
    if (!dense_ids || !dense_counts || !sparse_ids || !sparse_counts || !out_ids || !out_scores || !out_counts || B == 0 || top_k == 0)
        return cos_fail(COS_ERR_INVALID, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return cos_fail(COS_ERR_NO_DEVICE, "no HIP device visible; the GPU path has no CPU fallback");
    u32 maxn = 0;
    for (u32 q = 0; q < B; q++) {
        if (dense_counts[q] > dense_stride || sparse_counts[q] > sparse_stride) return cos_fail(COS_ERR_INVALID, "count exceeds stride (query %u)", q);
        maxn = std::max(maxn, dense_counts[q] + sparse_counts[q]);
    }
    if (maxn > 1024) return cos_fail(COS_ERR_UNIMPLEMENTED, "RRF lists longer than 1024 entries");
    u32 *d_d = nullptr, *d_dc = nullptr, *d_s = nullptr, *d_sc = nullptr, *d_oi = nullptr, *d_oc = nullptr;
    float *d_os = nullptr;
    hipError_t e = hipMalloc(&d_d, (size_t)B * dense_stride * 4);
    if (e == hipSuccess) e = hipMalloc(&d_s, (size_t)B * sparse_stride * 4);
    if (e == hipSuccess) e = hipMalloc(&d_dc, (size_t)B * 4);
    if (e == hipSuccess) e = hipMalloc(&d_sc, (size_t)B * 4);
    if (e == hipSuccess) e = hipMalloc(&d_oi, (size_t)B * top_k * 4);
    if (e == hipSuccess) e = hipMalloc(&d_os, (size_t)B * top_k * 4);
    if (e == hipSuccess) e = hipMalloc(&d_oc, (size_t)B * 4);
    if (e == hipSuccess) e = hipMemcpy(d_d, dense_ids, (size_t)B * dense_stride * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_s, sparse_ids, (size_t)B * sparse_stride * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_dc, dense_counts, (size_t)B * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_sc, sparse_counts, (size_t)B * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const size_t smem = (size_t)std::max(maxn, 1u) * 4;
#define LAUNCH(R) hipLaunchKernelGGL(rrf_kernel<R>, dim3(B), dim3(64), smem, 0, d_d, d_dc, dense_stride, d_s, d_sc, sparse_stride, B, fusion_constant_k, top_k, d_oi, d_os, d_oc)
        if (maxn <= 64) LAUNCH(1);
        else if (maxn <= 128) LAUNCH(2);
        else if (maxn <= 256) LAUNCH(4);
        else if (maxn <= 512) LAUNCH(8);
        else LAUNCH(16);
#undef LAUNCH
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out_ids, d_oi, (size_t)B * top_k * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_scores, d_os, (size_t)B * top_k * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_counts, d_oc, (size_t)B * 4, hipMemcpyDeviceToHost);
    void *ptrs[] = {d_d, d_s, d_dc, d_sc, d_oi, d_os, d_oc};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    HIP_TRY(e);
    return COS_OK;
}


// ------------------------------------------------------------------------------------------------
// repo::hybrid_search (api/vectordb/search/repo.rs:168-341) in one call: the dense index and the BM25 index are each asked
// for top_k * 3 (:200, :240, :251) — here concurrently, on two streams of the same device — and the two lists are fused with
// RRF on the device; only the fused top_k crosses PCIe.
// ------------------------------------------------------------------------------------------------
template <typename T>
static hipError_t grow_buf(T *&p, size_t &cap, size_t need) {
    if (need <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc((void **)&p, need * sizeof(T));
    if (e == hipSuccess) cap = need;
    return e;
}

extern "C" int32_t cos_hybrid_search_batch(cos_index *ix, cos_bm25 *b, const float *queries, const uint32_t *q_terms, const uint32_t *q_offsets, uint32_t B,
                                           uint32_t top_k, float fusion_constant_k, uint32_t *out_ids, float *out_scores, uint32_t *out_counts) {
    if (!ix || !b || !queries || !q_terms || !q_offsets || !out_ids || !out_scores || !out_counts || B == 0 || top_k == 0) return cos_fail(COS_ERR_INVALID, "bad argument");
    if (ix->p.device != b->device) return cos_fail(COS_ERR_INVALID, "the dense index and the BM25 index live on different devices");
    const u32 k3 = 3 * top_k, maxn = 2 * k3;
    if (maxn > 1024) return cos_fail(COS_ERR_UNIMPLEMENTED, "top_k above 170: RRF lists longer than 1024 entries");
    HIP_TRY(hipSetDevice(b->device));
    std::lock_guard<std::mutex> g(b->mu);
    int32_t rc = bm25_workspace(b, B, k3);
    if (rc) return rc;
    if (!b->stream_dense) {
        int prio_low = 0, prio_high = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
        HIP_TRY(hipStreamCreateWithPriority(&b->stream_dense, hipStreamNonBlocking, prio_high));
    }
    if (!b->ev_sparse) HIP_TRY(hipEventCreateWithFlags(&b->ev_sparse, hipEventDisableTiming));
    HIP_TRY(hipStreamSynchronize(b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream_dense));
    const size_t dim = ix->p.dim;
    HIP_TRY(grow_buf(b->d_hq, b->cap_hq, (size_t)B * dim));
    HIP_TRY(grow_buf(b->d_did, b->cap_did, (size_t)B * k3));
    HIP_TRY(grow_buf(b->d_dsc, b->cap_dsc, (size_t)B * k3));
    HIP_TRY(grow_buf(b->d_dcnt, b->cap_dcnt, (size_t)B));
    const size_t nk = (size_t)B * top_k, ret_words = 2 * nk + 2 * (size_t)B;
    HIP_TRY(grow_buf(b->d_ret, b->cap_ret, ret_words));
    if (ret_words * 4 > b->cap_hret) {
        if (b->h_ret) (void)hipHostFree(b->h_ret);
        b->h_ret = nullptr;
        b->cap_hret = 0;
        HIP_TRY(hipHostMalloc(&b->h_ret, ret_words * 4));
        b->cap_hret = ret_words * 4;
    }
    u32 *d_fid = b->d_ret, *d_fcnt = b->d_ret + 2 * nk;
    float *d_fsc = (float *)(b->d_ret + nk);
    int32_t *d_dst = (int32_t *)(b->d_ret + 2 * nk + B);
    // The dense half goes FIRST, on a stream of the highest priority: its walk is a chain of dependent rounds on a quarter of the chip's
    // wave slots (one workgroup per query), the sparse half is a bandwidth kernel of 8192 workgroups that fills whatever the walk
    // leaves free.  Until round 6 the sparse half was enqueued first: its workgroups held every CU until they drained and the dense
    // half's first dispatch (the query copy) waited for them — the two halves ran one after the other (c5: 2.55 ms per batch,
    // profiles/r06_final_kernel_trace_c5_rocprofv3.txt: a 256-workgroup copy of 0.52 ms beside a 0.57 ms bm25_score_kernel).
    hipStream_t sd = b->stream_dense;
    HIP_TRY(hipMemcpyAsync(b->d_hq, queries, (size_t)B * dim * 4, hipMemcpyHostToDevice, sd));
    rc = cos_search_batch_device(ix, b->d_hq, B, k3, b->d_did, b->d_dsc, b->d_dcnt, d_dst, sd);
    if (rc) return rc;
    // sparse half on its stream, beside the walk; its host side (the term table of the batch) is prepared while the device already walks
    rc = bm25_prepare(b, q_terms, q_offsets, B);
    if (rc) return rc;
    rc = bm25_launch(b, B, k3, b->d_ids, b->d_sc, b->d_cnt, b->stream);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(b->ev_sparse, b->stream));
    // fusion once both lists are there
    HIP_TRY(hipStreamWaitEvent(sd, b->ev_sparse, 0));
    const size_t smem = (size_t)maxn * 4;
#define LAUNCH(R) hipLaunchKernelGGL(rrf_kernel<R>, dim3(B), dim3(64), smem, sd, b->d_did, b->d_dcnt, k3, b->d_ids, b->d_cnt, k3, B, fusion_constant_k, top_k, d_fid, d_fsc, d_fcnt)
    if (maxn <= 64) LAUNCH(1);
    else if (maxn <= 128) LAUNCH(2);
    else if (maxn <= 256) LAUNCH(4);
    else if (maxn <= 512) LAUNCH(8);
    else LAUNCH(16);
#undef LAUNCH
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(b->h_ret, b->d_ret, ret_words * 4, hipMemcpyDeviceToHost, sd)); // (until round 6: four pageable copies)
    HIP_TRY(hipStreamSynchronize(sd));
    const u32 *hr = (const u32 *)b->h_ret;
    memcpy(out_ids, hr, nk * 4);
    memcpy(out_scores, hr + nk, nk * 4);
    memcpy(out_counts, hr + 2 * nk, (size_t)B * 4);
    const int32_t *status = (const int32_t *)(hr + 2 * nk + B);
    for (u32 q = 0; q < B; q++)
        if (status[q] != COS_OK) return cos_fail(status[q], "dense half: query %u failed with status %d (zero-norm vector -> DistanceError::CalculationError)", q, status[q]);
    return COS_OK;
}
