// kernels_sparse.hip — learned-sparse (SPLADE-style) inverted index search.
//   SparseAnnQueryBasic::sequential_search          models/sparse_ann_query.rs:68-147
//   InvertedIndexNode::quantize                     models/inverted_index.rs:168-172
//   InvertedIndex::search_internal / finalize_sparse_ann_results (raw-value rerank)   indexes/inverted/mod.rs:278-381
// The reference keeps, per dimension, one list of vector ids per QUANTIZED value (key); a query dimension with quantized value qq
// adds qq * key to every vector of every key list it visits (all keys when qq is above the early-termination threshold, the
// upper keys otherwise).  Restated as CSR — dims[T] ascending, key_off[T][2^bits + 1], vec_ids — the lists a query term visits
// are one contiguous range, so the kernel streams it once, coalesced (HBM-bound, 4 B per posting; no MFMA: integer adds).
// Sums are exact u32 -> atomic adds in any order give the reference's value.  The reference returns the survivors of
// select_nth_unstable in hash-map order; here (as in the oracle) they are ordered by similarity descending, larger id first.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

#include "engine_internal.h"

using namespace cosdev;

namespace {

constexpr u32 SEL = 64; // candidates kept per query: top_k * reranking_factor <= 64

struct SparseDev {
    const u32 *dims;      // [T]
    const u64 *key_off;   // [T][Q + 1]
    const u32 *vec_ids;   // postings
    const u64 *row_off;   // raw vectors (rerank): [n + 1]
    const u32 *raw_dims;
    const float *raw_vals;
    u32 T, Q, n, bits;
    float upper;
};

// Rust `as u8` / `as u32` on f32 (saturating, NaN -> 0) and f32::clamp
__device__ __forceinline__ u32 f32_as_u8(float v) { return !(v == v) || v <= 0.0f ? 0u : (v >= 255.0f ? 255u : (u32)(int)v); }
__device__ __forceinline__ u32 f32_as_u32(float v) { return !(v == v) || v <= 0.0f ? 0u : (v >= 4294967296.0f ? 0xFFFFFFFFu : (u32)v); }
__device__ __forceinline__ u32 sparse_quantize(float value, float upper, u32 bits) { // inverted_index.rs:168-172
    const u32 quantization = (1u << bits) - 1u;
    const float max_val = (float)quantization;
    float t = __fmul_rn(__fdiv_rn(value, upper), max_val);
    t = t < 0.0f ? 0.0f : (t > max_val ? max_val : t); // clamp keeps NaN
    const u32 q = f32_as_u8(t);
    return q < quantization ? q : quantization;
}

// grid (B, terms_max): block (b, i) streams the posting range of query b's i-th term
__global__ __launch_bounds__(256) void sparse_accumulate_kernel(const SparseDev ix, const u32 *__restrict__ q_dims, const float *__restrict__ q_vals,
                                                                const u32 *__restrict__ q_off, float early_terminate_threshold, u32 *__restrict__ acc /*[B][n]*/,
                                                                uint8_t *__restrict__ touched /*[B][n]: reached with weight 0*/) {
    const u32 b = blockIdx.x, i = blockIdx.y;
    const u32 t0 = q_off[b], nt = q_off[b + 1] - t0;
    if (i >= nt) return;
    const u32 dim = q_dims[t0 + i];
    u32 lo = 0, hi = ix.T; // find_node
    while (lo < hi) { const u32 mid = lo + (hi - lo) / 2; if (ix.dims[mid] < dim) lo = mid + 1; else hi = mid; }
    if (lo == ix.T || ix.dims[lo] != dim) return;
    const float qf = (float)ix.Q;
    float etv = __fmul_rn(qf, early_terminate_threshold);
    etv = etv > 255.0f ? 255.0f : etv;
    const u32 early_terminate_value = f32_as_u8(etv), low_threshold = f32_as_u32(__fmul_rn(early_terminate_threshold, qf));
    const u32 qq = sparse_quantize(q_vals[t0 + i], ix.upper, ix.bits);
    const u32 k0 = qq > low_threshold ? 0u : early_terminate_value;
    if (k0 >= ix.Q) return;
    const u64 *ko = ix.key_off + (u64)lo * (ix.Q + 1);
    const u64 beg = ko[k0], end = ko[ix.Q];
    u32 *ab = acc + (u64)b * ix.n;
    uint8_t *tb = touched + (u64)b * ix.n;
    // key of posting p = the last key whose list starts at or before p.  A thread's postings ascend, so after one binary search the
    // key pointer only moves forward (the first version searched the 2^bits + 1 offsets again for every posting).
    // A vector is a result as soon as any visited list holds it, also with similarity 0 (key 0, or a query value that quantizes
    // to 0): weight-0 postings set a BYTE flag with a plain store — idempotent, no read-modify-write — and everything else is one
    // atomic add; the first version OR-ed a bit into a bitmap for every posting, and the ids of a list ascend: up to 32 lanes of a
    // wave hit the same word, which same-address atomics serialise (the BM25 kernel's lesson, kernels_hybrid.hip).
    u32 l2 = k0;
    bool first = true;
    for (u64 p = beg + threadIdx.x; p < end; p += blockDim.x) {
        if (first) {
            u32 h2 = ix.Q;
            while (l2 + 1 < h2) { const u32 mid = (l2 + h2) / 2; if (ko[mid] <= p) l2 = mid; else h2 = mid; }
            first = false;
        } else {
            while (l2 + 1 < ix.Q && ko[l2 + 1] <= p) l2++;
        }
        const u32 v = ix.vec_ids[p];
        const u32 w = qq * l2;
        if (w) atomicAdd(&ab[v], w);
        else tb[v] = 1;
    }
}

// one wave per (query, segment): top-SEL of the reached vectors (non-zero sum, or the weight-0 flag) by (similarity, id);
// key = (sim + 1) << 32 | id (0 = empty)
__global__ __launch_bounds__(64) void sparse_select_segments(const u32 *__restrict__ acc, const uint8_t *__restrict__ touched, u32 n, u32 seg_len,
                                                             u64 *__restrict__ part /*[B][S][64]*/) {
    const int lane = threadIdx.x;
    const u32 q = blockIdx.x, seg = blockIdx.y, S = gridDim.y;
    Pool<1> pool;
    pool.clear();
    u64 thr = 0ull;
    const u32 c0 = seg * seg_len, c1 = (u64)c0 + seg_len < n ? c0 + seg_len : n;
    for (u32 c = c0; c < c1; c += 64) {
        const u32 v = c + lane;
        u64 key = 0ull;
        if (v < c1) {
            const u32 a = acc[(u64)q * n + v];
            if (a != 0u || touched[(u64)q * n + v]) key = ((u64)a + 1ull) << 32 | v;
        }
        u64 m = __ballot(key > thr);
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const u64 kk = readlane_u64(key, l);
            if (kk > thr) {
                pool.insert_at(kk, pool.rank_of(kk), lane);
                thr = readlane_u64(pool.e[0], SEL - 1);
            }
        }
    }
    part[((u64)q * S + seg) * SEL + lane] = pool.e[0];
}

// one wave per query: merge the segment pools, optional raw-value rerank, write the top k
__global__ __launch_bounds__(64) void sparse_finish_kernel(const SparseDev ix, const u64 *__restrict__ part, u32 S, const u32 *__restrict__ q_dims,
                                                           const float *__restrict__ q_vals, const u32 *__restrict__ q_off, u32 top_k, u32 k_with_reranking,
                                                           int rerank, u32 *__restrict__ out_ids, float *__restrict__ out_scores, u32 *__restrict__ out_counts) {
    const int lane = threadIdx.x;
    const u32 q = blockIdx.x;
    Pool<1> pool;
    pool.clear();
    u64 thr = 0ull;
    for (u32 sgm = 0; sgm < S; sgm++) {
        const u64 key = part[((u64)q * S + sgm) * SEL + lane];
        u64 m = __ballot(key > thr);
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const u64 kk = readlane_u64(key, l);
            if (kk > thr) {
                pool.insert_at(kk, pool.rank_of(kk), lane);
                thr = readlane_u64(pool.e[0], SEL - 1);
            }
        }
    }
    const u32 have = (u32)__popcll(__ballot(pool.e[0] != 0ull));
    const u32 ncand = have < k_with_reranking ? have : k_with_reranking; // select_nth + truncate(k * reranking_factor)
    const u64 mine = pool.e[0];
    if (!rerank) {
        const u32 nout = ncand < top_k ? ncand : top_k;
        if ((u32)lane < nout) {
            out_ids[(u64)q * top_k + lane] = (u32)mine;
            out_scores[(u64)q * top_k + lane] = (float)((u32)(mine >> 32) - 1u); // `similarity as f32`
        }
        if (lane == 0) out_counts[q] = nout;
        return;
    }
    // finalize_sparse_ann_results: dp over the QUERY pairs in order, f32 multiply then add; sort by total_cmp descending
    u64 res[1] = {0ull};
    if ((u32)lane < ncand) {
        const u32 v = (u32)mine;
        const u64 b = ix.row_off[v], e = ix.row_off[v + 1];
        float dp = 0.0f;
        for (u32 i = q_off[q]; i < q_off[q + 1]; i++) {
            const u32 d = q_dims[i];
            u64 lo = b, hi = e;
            while (lo < hi) { const u64 mid = lo + (hi - lo) / 2; if (ix.raw_dims[mid] < d) lo = mid + 1; else hi = mid; }
            if (lo < e && ix.raw_dims[lo] == d) dp = __fadd_rn(dp, __fmul_rn(ix.raw_vals[lo], q_vals[i]));
        }
        res[0] = pack_key(simkey(dp), v);
    }
    bitonic_sort_desc<1>(res, lane);
    const u32 nout = ncand < top_k ? ncand : top_k;
    if ((u32)lane < nout) {
        out_ids[(u64)q * top_k + lane] = (u32)res[0];
        out_scores[(u64)q * top_k + lane] = simkey_inv((u32)(res[0] >> 32));
    }
    if (lane == 0) out_counts[q] = nout;
}

} // namespace

struct cos_sparse {
    int32_t device = 0;
    u32 bits = 0, T = 0, n = 0;
    float upper = 1.0f;
    bool have_raw = false;
    u32 *d_dims = nullptr, *d_vec_ids = nullptr, *d_raw_dims = nullptr;
    u64 *d_key_off = nullptr, *d_row_off = nullptr;
    float *d_raw_vals = nullptr;
    // grow-only workspace of cos_sparse_search_batch (no allocation on the query path once warm); `mu` serialises callers
    std::mutex mu;
    struct Buf { void *p = nullptr; size_t cap = 0; } w_qd, w_qv, w_qo, w_acc, w_touched, w_part, w_oi, w_os, w_oc;
};

struct SparseView { void *p; template <typename T> T *as() const { return (T *)p; } };

static hipError_t sparse_grow(cos_sparse::Buf &b, size_t need) {
    if (need <= b.cap) return hipSuccess;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    hipError_t e = hipMalloc(&b.p, need);
    if (e == hipSuccess) b.cap = need;
    return e;
}

extern "C" int32_t cos_sparse_destroy(cos_sparse *s) {
    if (!s) return COS_OK;
    (void)hipSetDevice(s->device);
    void *ptrs[] = {s->d_dims, s->d_vec_ids, s->d_raw_dims, s->d_key_off, s->d_row_off, s->d_raw_vals, s->w_qd.p, s->w_qv.p, s->w_qo.p,
                    s->w_acc.p, s->w_touched.p, s->w_part.p, s->w_oi.p, s->w_os.p, s->w_oc.p};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    delete s;
    return COS_OK;
}

extern "C" int32_t cos_sparse_create(int32_t device, uint32_t quantization_bits, float values_upper_bound, const uint32_t *dims, uint32_t n_dims,
                                     const uint64_t *key_offsets, const uint32_t *vec_ids, uint32_t n_vectors, const uint64_t *row_offsets,
                                     const uint32_t *raw_dims, const float *raw_vals, cos_sparse **out) {
    if (!dims || !key_offsets || !vec_ids || !out || n_dims == 0 || n_vectors == 0) return cos_fail(COS_ERR_INVALID, "bad argument");
    *out = nullptr;
    if (quantization_bits < 1 || quantization_bits > 8) return cos_fail(COS_ERR_INVALID, "quantization_bits must be in [1, 8] (keys are u8)");
    const u32 Q = 1u << quantization_bits;
    for (u32 t = 0; t < n_dims; t++) {
        if (t && dims[t] <= dims[t - 1]) return cos_fail(COS_ERR_INVALID, "dimension indices must be strictly ascending");
        for (u32 k = 0; k < Q; k++)
            if (key_offsets[(size_t)t * (Q + 1) + k] > key_offsets[(size_t)t * (Q + 1) + k + 1]) return cos_fail(COS_ERR_INVALID, "key offsets of dimension %u decrease", dims[t]);
        if (t && key_offsets[(size_t)t * (Q + 1)] != key_offsets[(size_t)(t - 1) * (Q + 1) + Q]) return cos_fail(COS_ERR_INVALID, "posting ranges of consecutive dimensions must be contiguous");
    }
    const u64 nnz = key_offsets[(size_t)(n_dims - 1) * (Q + 1) + Q];
    for (u64 p = 0; p < nnz; p++)
        if (vec_ids[p] >= n_vectors) return cos_fail(COS_ERR_INVALID, "posting %llu names vector %u of %u", (unsigned long long)p, vec_ids[p], n_vectors);
    if ((row_offsets != nullptr) != (raw_dims != nullptr) || (raw_dims != nullptr) != (raw_vals != nullptr)) return cos_fail(COS_ERR_INVALID, "raw CSR: all three arrays or none");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return cos_fail(COS_ERR_NO_DEVICE, "no HIP device visible; the GPU path has no CPU fallback");
    HIP_TRY(hipSetDevice(device));
    cos_sparse *s = new cos_sparse();
    s->device = device; s->bits = quantization_bits; s->T = n_dims; s->n = n_vectors; s->upper = values_upper_bound;
    auto up = [&](void **dst, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dst, bytes ? bytes : 1);
        return e == hipSuccess ? hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) : e;
    };
    hipError_t e = up((void **)&s->d_dims, dims, (size_t)n_dims * 4);
    if (e == hipSuccess) e = up((void **)&s->d_key_off, key_offsets, (size_t)n_dims * (Q + 1) * 8);
    if (e == hipSuccess) e = up((void **)&s->d_vec_ids, vec_ids, (size_t)nnz * 4);
    if (e == hipSuccess && row_offsets) {
        const u64 rnnz = row_offsets[n_vectors];
        e = up((void **)&s->d_row_off, row_offsets, ((size_t)n_vectors + 1) * 8);
        if (e == hipSuccess) e = up((void **)&s->d_raw_dims, raw_dims, (size_t)rnnz * 4);
        if (e == hipSuccess) e = up((void **)&s->d_raw_vals, raw_vals, (size_t)rnnz * 4);
        s->have_raw = true;
    }
    if (e != hipSuccess) { cos_sparse_destroy(s); HIP_TRY(e); }
    *out = s;
    return COS_OK;
}

// InvertedIndexNode::quantize on the host (the same f32 operations as sparse_quantize above; this file is built with -ffp-contract=off)
static inline uint32_t host_sparse_quantize(float value, float upper, uint32_t bits) {
    const uint32_t quantization = (1u << bits) - 1u;
    const float max_val = (float)quantization;
    float t = (value / upper) * max_val;
    t = t < 0.0f ? 0.0f : (t > max_val ? max_val : t); // f32::clamp keeps NaN
    const uint32_t q = !(t == t) || t <= 0.0f ? 0u : (t >= 255.0f ? 255u : (uint32_t)(int)t); // `as u8`
    return q < quantization ? q : quantization;
}

// InvertedIndex::insert for a whole collection (indexes/inverted/mod.rs + models/inverted_index.rs:176-200): the vectors are taken in
// id order and every (dimension, value) pair pushes the id to the END of the list of (dimension, quantize(value)) — so the CSR this
// produces is what the host's tree holds after inserting ids 0 .. n-1.  Host code, no device.
extern "C" int32_t cos_sparse_build_csr(uint32_t quantization_bits, float values_upper_bound, uint32_t n_vectors, const uint64_t *row_offsets,
                                        const uint32_t *raw_dims, const float *raw_vals, uint32_t *out_dims, uint64_t *out_key_offsets,
                                        uint32_t *out_vec_ids, uint32_t *n_dims) {
    if (!row_offsets || !raw_dims || !raw_vals || !n_dims || n_vectors == 0) return cos_fail(COS_ERR_INVALID, "bad argument");
    if (quantization_bits < 1 || quantization_bits > 8) return cos_fail(COS_ERR_INVALID, "quantization_bits must be in [1, 8] (keys are u8)");
    const u32 Q = 1u << quantization_bits;
    const u64 nnz = row_offsets[n_vectors];
    for (u32 v = 0; v < n_vectors; v++)
        if (row_offsets[v + 1] < row_offsets[v]) return cos_fail(COS_ERR_INVALID, "row offsets decrease at vector %u", v);
    std::vector<u32> dims(raw_dims, raw_dims + nnz);
    std::sort(dims.begin(), dims.end());
    dims.erase(std::unique(dims.begin(), dims.end()), dims.end());
    const u32 T = (u32)dims.size();
    const bool sizes_only = !out_dims && !out_key_offsets && !out_vec_ids;
    if (sizes_only) { *n_dims = T; return COS_OK; }
    if (!out_dims || !out_key_offsets || !out_vec_ids) return cos_fail(COS_ERR_INVALID, "all three output arrays or none");
    if (*n_dims < T) { *n_dims = T; return cos_fail(COS_ERR_INVALID, "%u distinct dimensions, room for fewer", T); }
    *n_dims = T;
    std::vector<u64> count((size_t)T * Q, 0);
    std::vector<u32> slot(nnz);
    for (u64 p = 0; p < nnz; p++) {
        const u32 t = (u32)(std::lower_bound(dims.begin(), dims.end(), raw_dims[p]) - dims.begin());
        slot[p] = t * Q + host_sparse_quantize(raw_vals[p], values_upper_bound, quantization_bits);
        count[slot[p]]++;
    }
    std::vector<u64> cursor((size_t)T * Q);
    u64 run = 0;
    for (u32 t = 0; t < T; t++) {
        for (u32 k = 0; k < Q; k++) {
            out_key_offsets[(size_t)t * (Q + 1) + k] = run;
            cursor[(size_t)t * Q + k] = run;
            run += count[(size_t)t * Q + k];
        }
        out_key_offsets[(size_t)t * (Q + 1) + Q] = run;
        out_dims[t] = dims[t];
    }
    for (u32 v = 0; v < n_vectors; v++) // id order = push order
        for (u64 p = row_offsets[v]; p < row_offsets[v + 1]; p++) out_vec_ids[cursor[slot[p]]++] = v;
    return COS_OK;
}

// cos_sparse_build_csr + cos_sparse_create in one call; keep_raw != 0 also uploads the raw vectors for the raw-value rerank
// (their dims must then ascend within a row, finalize_sparse_ann_results' lookup is a binary search)
extern "C" int32_t cos_sparse_create_from_vectors(int32_t device, uint32_t quantization_bits, float values_upper_bound, uint32_t n_vectors,
                                                  const uint64_t *row_offsets, const uint32_t *raw_dims, const float *raw_vals, int32_t keep_raw,
                                                  cos_sparse **out) {
    if (!out) return cos_fail(COS_ERR_INVALID, "null argument");
    *out = nullptr;
    u32 T = 0;
    int32_t rc = cos_sparse_build_csr(quantization_bits, values_upper_bound, n_vectors, row_offsets, raw_dims, raw_vals, nullptr, nullptr, nullptr, &T);
    if (rc) return rc;
    const u32 Q = 1u << quantization_bits;
    std::vector<u32> dims(std::max(T, 1u)), ids((size_t)std::max<u64>(row_offsets[n_vectors], 1));
    std::vector<uint64_t> ko((size_t)std::max(T, 1u) * (Q + 1));
    rc = cos_sparse_build_csr(quantization_bits, values_upper_bound, n_vectors, row_offsets, raw_dims, raw_vals, dims.data(), ko.data(), ids.data(), &T);
    if (rc) return rc;
    if (T == 0) return cos_fail(COS_ERR_INVALID, "no postings");
    return cos_sparse_create(device, quantization_bits, values_upper_bound, dims.data(), T, ko.data(), ids.data(), n_vectors,
                             keep_raw ? row_offsets : nullptr, keep_raw ? raw_dims : nullptr, keep_raw ? raw_vals : nullptr, out);
}

extern "C" int32_t cos_sparse_search_batch(cos_sparse *s, const uint32_t *q_dims, const float *q_vals, const uint32_t *q_offsets, uint32_t B, uint32_t top_k,
                                           float early_terminate_threshold, uint32_t reranking_factor, uint32_t *out_ids, float *out_scores,
                                           uint32_t *out_counts) {
    if (!s || !q_dims || !q_vals || !q_offsets || !out_ids || !out_scores || !out_counts || B == 0 || top_k == 0) return cos_fail(COS_ERR_INVALID, "bad argument");
    const bool rerank = reranking_factor != 0;
    if (rerank && !s->have_raw) return cos_fail(COS_ERR_NOT_READY, "raw-value rerank needs the raw sparse vectors (cos_sparse_create row_offsets / raw_dims / raw_vals)");
    const u32 kwr = top_k * (rerank ? reranking_factor : 1u);
    if (kwr > SEL) return cos_fail(COS_ERR_UNIMPLEMENTED, "top_k x reranking_factor must be <= %u", SEL);
    HIP_TRY(hipSetDevice(s->device));
    u32 max_terms = 0;
    for (u32 b = 0; b < B; b++) {
        if (q_offsets[b + 1] < q_offsets[b]) return cos_fail(COS_ERR_INVALID, "query offsets decrease");
        max_terms = std::max(max_terms, q_offsets[b + 1] - q_offsets[b]);
    }
    const u32 nq = q_offsets[B];
    SparseDev dev{s->d_dims, s->d_key_off, s->d_vec_ids, s->d_row_off, s->d_raw_dims, s->d_raw_vals, s->T, 1u << s->bits, s->n, s->bits, s->upper};
    // queries are processed in chunks so that the per-query accumulators stay below 2 GiB
    const u32 chunkB = (u32)std::max<u64>(1, std::min<u64>(B, (2ull << 30) / ((u64)s->n * 4)));
    u32 Sg = std::max<u32>(1u, std::min<u32>(64u, std::min<u32>((s->n + 4095) / 4096, (4096 + chunkB - 1) / chunkB)));
    const u32 seg_len = ((s->n + Sg - 1) / Sg + 63) / 64 * 64;
    std::lock_guard<std::mutex> guard(s->mu);
    HIP_TRY(sparse_grow(s->w_qd, (size_t)std::max(nq, 1u) * 4));
    HIP_TRY(sparse_grow(s->w_qv, (size_t)std::max(nq, 1u) * 4));
    HIP_TRY(sparse_grow(s->w_qo, ((size_t)B + 1) * 4));
    HIP_TRY(sparse_grow(s->w_acc, (size_t)chunkB * s->n * 4));
    HIP_TRY(sparse_grow(s->w_touched, (size_t)chunkB * s->n));
    HIP_TRY(sparse_grow(s->w_part, (size_t)chunkB * Sg * SEL * 8));
    HIP_TRY(sparse_grow(s->w_oi, (size_t)B * top_k * 4));
    HIP_TRY(sparse_grow(s->w_os, (size_t)B * top_k * 4));
    HIP_TRY(sparse_grow(s->w_oc, (size_t)B * 4));
    const SparseView d_qd{s->w_qd.p}, d_qv{s->w_qv.p}, d_qo{s->w_qo.p}, d_acc{s->w_acc.p}, d_touched{s->w_touched.p}, d_part{s->w_part.p}, d_oi{s->w_oi.p},
        d_os{s->w_os.p}, d_oc{s->w_oc.p};
    HIP_TRY(hipMemcpy(d_qd.p, q_dims, (size_t)nq * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_qv.p, q_vals, (size_t)nq * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_qo.p, q_offsets, ((size_t)B + 1) * 4, hipMemcpyHostToDevice));
    for (u32 b0 = 0; b0 < B; b0 += chunkB) {
        const u32 nb = std::min(chunkB, B - b0);
        HIP_TRY(hipMemsetAsync(d_acc.p, 0, (size_t)nb * s->n * 4, 0));
        HIP_TRY(hipMemsetAsync(d_touched.p, 0, (size_t)nb * s->n, 0));
        if (max_terms) {
            hipLaunchKernelGGL(sparse_accumulate_kernel, dim3(nb, max_terms), dim3(256), 0, 0, dev, d_qd.as<u32>(), d_qv.as<float>(), d_qo.as<u32>() + b0,
                               early_terminate_threshold, d_acc.as<u32>(), d_touched.as<uint8_t>());
            HIP_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(sparse_select_segments, dim3(nb, Sg), dim3(64), 0, 0, d_acc.as<u32>(), d_touched.as<uint8_t>(), s->n, seg_len, d_part.as<u64>());
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(sparse_finish_kernel, dim3(nb), dim3(64), 0, 0, dev, d_part.as<u64>(), Sg, d_qd.as<u32>(), d_qv.as<float>(), d_qo.as<u32>() + b0, top_k, kwr,
                           rerank ? 1 : 0, d_oi.as<u32>() + (size_t)b0 * top_k, d_os.as<float>() + (size_t)b0 * top_k, d_oc.as<u32>() + b0);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpy(out_ids, d_oi.p, (size_t)B * top_k * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_scores, d_os.p, (size_t)B * top_k * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_counts, d_oc.p, (size_t)B * 4, hipMemcpyDeviceToHost));
    return COS_OK;
}
