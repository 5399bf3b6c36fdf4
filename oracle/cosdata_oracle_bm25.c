/*
 * cosdata_oracle_bm25.c — BM25 scoring + reciprocal-rank fusion of the oracle.
 * TEST INFRASTRUCTURE (see cosdata_oracle.h).  Spec: SURVEY.md Appendix A.5.
 */
#include "cosdata_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define BUCKETS 512 /* sparse_ann_query.rs:154 */

/* sparse_ann_query.rs:298-302 */
float coso_bm25_idf(uint32_t documents_count, uint32_t containing) {
    float num = (float)(uint32_t)(documents_count - containing) + 0.5f;
    float den = (float)containing + 0.5f;
    return log1pf(num / den);
}
/* indexes/tf_idf/mod.rs:362-371 */
float coso_bm25_tf(uint32_t count, uint32_t doc_len, float avg_len, float k1, float b) {
    float c = (float)count;
    return c * (k1 + 1.0f) / (c + k1 * (1.0f - b + b * ((float)doc_len / avg_len)));
}

static inline int32_t total_key(float v) {
    int32_t b;
    memcpy(&b, &v, 4);
    b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
    return b;
}
typedef struct { float score; uint32_t id; } sres;
static int cmp_sres_desc(const void *a, const void *b) {
    const sres *x = (const sres *)a, *y = (const sres *)b;
    int32_t kx = total_key(x->score), ky = total_key(y->score);
    if (kx != ky) return kx > ky ? -1 : 1;
    return x->id > y->id ? -1 : (x->id < y->id ? 1 : 0); /* documented tie-break: larger id first */
}
static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
static int64_t find_term(const uint32_t *terms, uint32_t T, uint32_t h) {
    uint32_t lo = 0, hi = T;
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (terms[mid] < h) lo = mid + 1; else hi = mid;
    }
    return (lo < T && terms[lo] == h) ? (int64_t)lo : -1;
}

/* search_bm25 (sparse_ann_query.rs:149-233): document-at-a-time merge over the query terms'
 * posting lists; score(doc) = Σ tf·idf accumulated in ASCENDING TERM-HASH order (the reference's
 * heap order among equal doc ids is unspecified); 512 buckets keyed doc_id % 512 keep the strictly
 * greater score (first-seen wins ties; docs are visited in ascending id). */
int coso_bm25_search(const uint32_t *term_hashes, const uint64_t *offsets, uint32_t T, const uint32_t *doc_ids,
                     const float *tfs, uint32_t documents_count, const uint32_t *query_terms, uint32_t nq, uint32_t top_k,
                     uint32_t *out_ids, float *out_scores) {
    uint32_t *qt = (uint32_t *)malloc((size_t)(nq ? nq : 1) * 4);
    memcpy(qt, query_terms, (size_t)nq * 4);
    qsort(qt, nq, 4, cmp_u32);
    uint64_t *pos = (uint64_t *)malloc((size_t)(nq ? nq : 1) * 8), *end = (uint64_t *)malloc((size_t)(nq ? nq : 1) * 8);
    float *idf = (float *)malloc((size_t)(nq ? nq : 1) * 4);
    uint32_t nl = 0;
    for (uint32_t i = 0; i < nq; i++) {
        int64_t t = find_term(term_hashes, T, qt[i]);
        if (t < 0) continue;
        pos[nl] = offsets[t];
        end[nl] = offsets[t + 1];
        idf[nl] = coso_bm25_idf(documents_count, (uint32_t)(offsets[t + 1] - offsets[t]));
        nl++;
    }
    sres buckets[BUCKETS];
    for (int i = 0; i < BUCKETS; i++) { buckets[i].id = 0xFFFFFFFFu; buckets[i].score = -INFINITY; }
    for (;;) {
        uint32_t doc = 0xFFFFFFFFu;
        int any = 0;
        for (uint32_t l = 0; l < nl; l++)
            if (pos[l] < end[l] && (!any || doc_ids[pos[l]] < doc)) { doc = doc_ids[pos[l]]; any = 1; }
        if (!any) break;
        float score = 0.0f;
        int first = 1;
        for (uint32_t l = 0; l < nl; l++) { /* ascending term hash */
            if (pos[l] < end[l] && doc_ids[pos[l]] == doc) {
                float p = tfs[pos[l]] * idf[l];
                if (first) { score = p; first = 0; } else score = score + p;
                pos[l]++;
            }
        }
        uint32_t bi = doc % BUCKETS;
        if (score > buckets[bi].score) { buckets[bi].id = doc; buckets[bi].score = score; }
    }
    sres res[BUCKETS];
    int m = 0;
    for (int i = 0; i < BUCKETS; i++)
        if (buckets[i].id != 0xFFFFFFFFu) res[m++] = buckets[i];
    qsort(res, (size_t)m, sizeof(sres), cmp_sres_desc);
    if ((uint32_t)m > top_k) m = (int)top_k;
    for (int i = 0; i < m; i++) { out_ids[i] = res[i].id; out_scores[i] = res[i].score; }
    free(qt); free(pos); free(end); free(idf);
    return m;
}

/* hybrid_search fusion (api/vectordb/search/repo.rs:311-340):
 * score = 1/(rank + k + f32::EPSILON), dense list inserted first, sparse list added. */
int coso_rrf_fuse(const uint32_t *dense_ids, uint32_t nd, const uint32_t *sparse_ids, uint32_t ns, float k_rrf, uint32_t top_k,
                  uint32_t *out_ids, float *out_scores) {
    sres *acc = (sres *)malloc((size_t)(nd + ns + 1) * sizeof(sres));
    uint32_t m = 0;
    for (uint32_t r = 0; r < nd; r++) {
        float sc = 1.0f / ((float)r + k_rrf + 1.1920929e-07f);
        uint32_t j = 0;
        for (; j < m; j++)
            if (acc[j].id == dense_ids[r]) break;
        if (j == m) { acc[m].id = dense_ids[r]; m++; }
        acc[j].score = sc; /* HashMap::insert overwrites */
    }
    for (uint32_t r = 0; r < ns; r++) {
        float sc = 1.0f / ((float)r + k_rrf + 1.1920929e-07f);
        uint32_t j = 0;
        for (; j < m; j++)
            if (acc[j].id == sparse_ids[r]) break;
        if (j == m) { acc[m].id = sparse_ids[r]; acc[m].score = 0.0f; m++; }
        acc[j].score = acc[j].score + sc;
    }
    qsort(acc, m, sizeof(sres), cmp_sres_desc);
    if (m > top_k) m = top_k;
    for (uint32_t i = 0; i < m; i++) { out_ids[i] = acc[i].id; out_scores[i] = acc[i].score; }
    free(acc);
    return (int)m;
}
