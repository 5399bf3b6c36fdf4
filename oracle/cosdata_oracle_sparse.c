/*
 * cosdata_oracle_sparse.c — learned-sparse inverted index search.  TEST INFRASTRUCTURE (cosdata_oracle.h).
 *   SparseAnnQueryBasic::sequential_search         models/sparse_ann_query.rs:68-147
 *   InvertedIndexNode::quantize                    models/inverted_index.rs:168-172
 *   InvertedIndex::search_internal + finalize_sparse_ann_results (raw-value rerank)   indexes/inverted/mod.rs:278-381
 * Index layout restated as CSR: dims[T] ascending; for dimension t and quantized key q in [0, 2^bits) the vector ids
 * vec_ids[key_off[t*(2^bits+1)+q] .. key_off[t*(2^bits+1)+q+1]) (the reference keeps one list per (dimension, key)).
 * Documented choice where the reference is unspecified: it returns the candidates in hash-map order (select_nth_unstable, no
 * sort); here they come out by similarity descending, larger id first; the reranked list by dot product descending
 * (total_cmp), larger id first.
 */
#include "cosdata_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* Rust `as u8` on f32: saturating, NaN -> 0 */
static uint8_t f32_as_u8(float v) {
    if (!(v == v) || v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)(int)v;
}
static uint32_t f32_as_u32(float v) {
    if (!(v == v) || v <= 0.0f) return 0;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}
static float clampf(float v, float lo, float hi) { /* f32::clamp: NaN stays NaN */
    if (v < lo) return lo;
    if (v > hi) return hi;
    return v;
}

uint8_t coso_sparse_quantize(float value, float values_upper_bound, int bits) {
    const uint8_t quantization = (uint8_t)((1u << bits) - 1u);
    const float max_val = (float)quantization;
    uint8_t q = f32_as_u8(clampf((value / values_upper_bound) * max_val, 0.0f, max_val));
    return q < quantization ? q : quantization;
}

typedef struct { uint32_t sim, id; } sres;
static int cmp_sres(const void *a, const void *b) {
    const sres *x = (const sres *)a, *y = (const sres *)b;
    if (x->sim != y->sim) return x->sim > y->sim ? -1 : 1;
    return x->id > y->id ? -1 : (x->id < y->id ? 1 : 0);
}

/* returns the number of candidates written (<= cap); k_with_reranking = 0 means "k = None": every touched vector */
int coso_sparse_search(const uint32_t *dims, uint32_t T, const uint64_t *key_off, const uint32_t *vec_ids, uint32_t n_vectors, int bits,
                       float values_upper_bound, float early_terminate_threshold, const uint32_t *q_dims, const float *q_vals, uint32_t nq,
                       uint32_t k_with_reranking, uint32_t *out_ids, uint32_t *out_sims, uint32_t cap) {
    if (!dims || !key_off || bits < 1 || bits > 8) return -COSO_ERR_INVALID;
    const uint32_t Q = 1u << bits, one_quantized = Q - 1u;
    const float qf = (float)Q;
    float etv = qf * early_terminate_threshold;
    if (etv > 255.0f) etv = 255.0f; /* .min(u8::MAX as f32) */
    const uint8_t early_terminate_value = f32_as_u8(etv);
    const uint32_t low_threshold = f32_as_u32(early_terminate_threshold * qf);
    uint32_t *acc = (uint32_t *)calloc(n_vectors ? n_vectors : 1, 4);
    uint8_t *touched = (uint8_t *)calloc(n_vectors ? n_vectors : 1, 1);
    for (uint32_t i = 0; i < nq; i++) {
        uint32_t lo = 0, hi = T; /* find_node */
        while (lo < hi) { uint32_t mid = lo + (hi - lo) / 2; if (dims[mid] < q_dims[i]) lo = mid + 1; else hi = mid; }
        if (lo == T || dims[lo] != q_dims[i]) continue;
        const uint32_t qq = coso_sparse_quantize(q_vals[i], values_upper_bound, bits);
        const uint32_t k0 = qq > low_threshold ? 0u : (uint32_t)early_terminate_value;
        for (uint32_t key = k0; key <= one_quantized; key++) {
            const uint64_t b = key_off[(size_t)lo * (Q + 1) + key], e = key_off[(size_t)lo * (Q + 1) + key + 1];
            for (uint64_t p = b; p < e; p++) {
                const uint32_t v = vec_ids[p];
                if (v >= n_vectors) { free(acc); free(touched); return -COSO_ERR_INVALID; }
                acc[v] += qq * key;
                touched[v] = 1;
            }
        }
    }
    uint32_t m = 0;
    for (uint32_t v = 0; v < n_vectors; v++) m += touched[v];
    sres *r = (sres *)malloc((size_t)(m ? m : 1) * sizeof(sres));
    m = 0;
    for (uint32_t v = 0; v < n_vectors; v++)
        if (touched[v]) { r[m].sim = acc[v]; r[m].id = v; m++; }
    qsort(r, m, sizeof(sres), cmp_sres);
    if (k_with_reranking && m > k_with_reranking) m = k_with_reranking;
    if (m > cap) m = cap;
    for (uint32_t i = 0; i < m; i++) { out_ids[i] = r[i].id; out_sims[i] = r[i].sim; }
    free(r); free(acc); free(touched);
    return (int)m;
}

typedef struct { float dp; uint32_t id; } rres;
static int32_t tkey(float v) { int32_t b; memcpy(&b, &v, 4); b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1); return b; }
static int cmp_rres(const void *a, const void *b) {
    const rres *x = (const rres *)a, *y = (const rres *)b;
    int32_t kx = tkey(x->dp), ky = tkey(y->dp);
    if (kx != ky) return kx > ky ? -1 : 1;
    return x->id > y->id ? -1 : (x->id < y->id ? 1 : 0);
}

/* finalize_sparse_ann_results: dp = sum over the QUERY pairs, in query order, of raw value * query value for the dimensions
 * the vector has (f32, mul then add); sort descending; truncate k (0 = None).  Raw vectors as CSR (dims ascending per row). */
int coso_sparse_rerank(const uint64_t *row_off, const uint32_t *raw_dims, const float *raw_vals, const uint32_t *cand_ids, uint32_t m,
                       const uint32_t *q_dims, const float *q_vals, uint32_t nq, uint32_t top_k, uint32_t *out_ids, float *out_scores) {
    rres *r = (rres *)malloc((size_t)(m ? m : 1) * sizeof(rres));
    for (uint32_t c = 0; c < m; c++) {
        const uint32_t v = cand_ids[c];
        const uint64_t b = row_off[v], e = row_off[v + 1];
        float dp = 0.0f;
        for (uint32_t i = 0; i < nq; i++) {
            uint64_t lo = b, hi = e;
            while (lo < hi) { uint64_t mid = lo + (hi - lo) / 2; if (raw_dims[mid] < q_dims[i]) lo = mid + 1; else hi = mid; }
            if (lo < e && raw_dims[lo] == q_dims[i]) dp += raw_vals[lo] * q_vals[i];
        }
        r[c].dp = dp;
        r[c].id = v;
    }
    qsort(r, m, sizeof(rres), cmp_rres);
    if (top_k && m > top_k) m = top_k;
    for (uint32_t i = 0; i < m; i++) { out_ids[i] = r[i].id; out_scores[i] = r[i].dp; }
    free(r);
    return (int)m;
}
