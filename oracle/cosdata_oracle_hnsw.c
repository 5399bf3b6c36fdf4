/*
 * cosdata_oracle_hnsw.c — HNSW walk, finalisation and deterministic builder of the oracle.
 * TEST INFRASTRUCTURE (see cosdata_oracle.h).  Behavioural spec: SURVEY.md Appendix A.
 */
#include "cosdata_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define IDX_NONE 0xFFFFFFFFu
#define KEEP_SEARCH 100 /* vector_store.rs:1194 final_len (search) */
#define KEEP_INDEX 64   /* vector_store.rs:1194 final_len (indexing) */

typedef struct {
    uint32_t n, cap, M;
    uint32_t *node_id; /* internal id per node (COSO_ROOT_ID for the root) */
    uint32_t *nbr;     /* [n][M] level-local node index, IDX_NONE = null slot */
    float *nbr_sim;    /* [n][M] */
    uint32_t *child;   /* [n] node index one level down (levels >= 1) */
    uint8_t *low_idx;  /* prob_node.rs:108 lowest_index cache */
    float *low_sim;
    uint32_t root_idx;
    int sorted; /* node_id ascending (imported) -> binary search allowed */
} level_t;

struct coso_index {
    coso_params p;
    uint32_t n;
    const float *raw;
    size_t cb;
    uint8_t *codes; /* [n+1][cb] */
    float *mags;    /* [n+1] */
    float *root_raw;
    level_t *lv; /* [num_layers+1] */
    int has_root;
    /* coso_index_build_rounds leaves what a later coso_index_build_rounds_continue needs: the RNG stream after the last level draw
     * and how many vectors the graph holds (index_embeddings called again on a live index, vector_store.rs:714-780) */
    uint64_t rng_state;
    uint32_t n_built;
    int rounds_state_valid, rounds_greedy;
    /* corpora too large for a host copy of the raw f32 table (bench.py --workload c4shard): the rerank reads the raw rows of
     * the few candidates it needs from a caller-provided subset (ids ascending) instead of ix->raw */
    const uint32_t *sub_ids;
    const float *sub_rows;
    uint32_t sub_n;
    /* metadata-filtered search: the pseudo-root component — pseudo nodes and the Metadata replicas — is a second
     * graph with its own levels and entry point (the pseudo root); its nodes carry metadata dimensions (types.rs:106-147) */
    uint32_t mdim, replicas;  /* metadata dimensions; max_replica_per_node (ids of one embedding: base .. base + replicas - 1) */
    uint32_t n_meta;
    uint32_t *meta_id;        /* [n_meta] ascending replica ids (incl. the pseudo root u32::MAX - 257 and the pseudo nodes) */
    int32_t *meta_mbits;      /* [n_meta][mdim] */
    float *meta_mdims;        /* [n_meta][mdim] the same as f32 (cosine_similarity_mdims converts) */
    float *meta_mag;          /* [n_meta] */
    level_t *mlv;             /* [num_layers+1] */
};

typedef struct {
    int32_t key; /* metric-aware order key: larger = better */
    uint32_t id;
    uint32_t idx;
    float sim;
} hent;

typedef struct {
    hent *heap;
    size_t heap_cap;
    hent *res;
    size_t res_cap;
    uint64_t *visited;
    size_t visited_words;
    uint32_t *touched; /* EXACT mode: words dirtied by the previous walk (cleared lazily) */
    size_t ntouched, touched_cap;
    uint8_t *qcode;
} scratch_t;

static inline int32_t total_key(float v) {
    int32_t b;
    memcpy(&b, &v, 4);
    b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
    return b;
}
static inline int32_t order_key(int metric, float v) {
    int32_t k = total_key(v);
    return (metric == COSO_METRIC_EUCLIDEAN || metric == COSO_METRIC_HAMMING) ? ~k : k;
}
/* strict "a is greater than b": similarity first, then LARGER internal id (documented tie-break) */
static inline int hent_gt(const hent *a, const hent *b) {
    return a->key > b->key || (a->key == b->key && a->id > b->id);
}

/* vector row of an internal id: with a metadata schema every embedding reserves `replicas` ids (collection.rs:445-468) and the
 * Base node of vector row r has id r * replicas; without one (replicas == 0) id == row */
static inline uint32_t row_of(const coso_index *ix, uint32_t id) { return id == COSO_ROOT_ID ? ix->n : (ix->replicas > 1 ? id / ix->replicas : id); }
static inline uint32_t id_of_row(const coso_index *ix, uint32_t row) { return ix->replicas > 1 ? row * ix->replicas : row; }
static inline uint32_t level_M(const coso_index *ix, uint32_t level) {
    return level == 0 ? ix->p.level0_neighbors_count : ix->p.neighbors_count;
}
static inline float metric_min(int metric) { return metric == COSO_METRIC_COSINE ? -1.0f : -INFINITY; } /* types.rs:435-446 */
static inline float metric_max(int metric) { return metric == COSO_METRIC_COSINE ? 2.0f : INFINITY; }   /* types.rs:448-457 */

/* ------------------------------------------------------------------------------------------ */
coso_index *coso_index_create(const coso_params *p) {
    if (!p || p->dim == 0) return NULL;
    uint32_t M = p->neighbors_count, M0 = p->level0_neighbors_count;
    if (M == 0 || M0 == 0 || (M & (M - 1)) || (M0 & (M0 - 1))) return NULL; /* fixedset.rs needs a power of two */
    if (M > 256 || M0 > 256) return NULL;                                     /* lowest_index is a u8 */
    coso_index *ix = (coso_index *)calloc(1, sizeof(*ix));
    ix->p = *p;
    ix->cb = coso_code_bytes((int)p->storage, (int)p->resolution, (int)p->dim);
    if (ix->cb == 0) { free(ix); return NULL; }
    ix->lv = (level_t *)calloc(p->num_layers + 1, sizeof(level_t));
    for (uint32_t l = 0; l <= p->num_layers; l++) { ix->lv[l].M = level_M(ix, l); ix->lv[l].root_idx = IDX_NONE; }
    ix->root_raw = (float *)calloc(p->dim, sizeof(float));
    return ix;
}

static void level_free(level_t *L) {
    free(L->node_id); free(L->nbr); free(L->nbr_sim); free(L->child); free(L->low_idx); free(L->low_sim);
    uint32_t M = L->M;
    memset(L, 0, sizeof(*L));
    L->M = M;
    L->root_idx = IDX_NONE;
}
void coso_index_destroy(coso_index *ix) {
    if (!ix) return;
    for (uint32_t l = 0; l <= ix->p.num_layers; l++) level_free(&ix->lv[l]);
    if (ix->mlv) for (uint32_t l = 0; l <= ix->p.num_layers; l++) level_free(&ix->mlv[l]);
    free(ix->mlv); free(ix->meta_id); free(ix->meta_mbits); free(ix->meta_mdims); free(ix->meta_mag);
    free(ix->lv); free(ix->codes); free(ix->mags); free(ix->root_raw); free(ix);
}

int coso_index_set_vectors(coso_index *ix, const float *raw, uint32_t n) {
    if (!ix || (!raw && n)) return COSO_ERR_INVALID;
    ix->rounds_state_valid = 0;
    free(ix->codes); free(ix->mags);
    ix->n = n;
    ix->raw = raw;
    ix->codes = (uint8_t *)calloc((size_t)n + 2, ix->cb); /* row n = root, row n + 1 = the pseudo nodes' vector (f4a) */
    ix->mags = (float *)calloc((size_t)n + 2, sizeof(float));
    int rc = COSO_OK;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        int r = coso_quantize(raw + (size_t)i * ix->p.dim, (int)ix->p.dim, (int)ix->p.storage, (int)ix->p.resolution,
                              ix->p.range_lo, ix->p.range_hi, ix->codes + (size_t)i * ix->cb, &ix->mags[i]);
        if (r != COSO_OK) rc = r;
    }
    if (ix->has_root) /* re-quantize the root into its row */
        coso_quantize(ix->root_raw, (int)ix->p.dim, (int)ix->p.storage, (int)ix->p.resolution, ix->p.range_lo,
                      ix->p.range_hi, ix->codes + (size_t)n * ix->cb, &ix->mags[n]);
    return rc;
}

/* Chunked variant of coso_index_set_vectors for corpora whose raw f32 table does not fit in host memory: allocate the code
 * table, then quantize chunks of rows as they are streamed in; ix->raw stays NULL (see coso_index_set_raw_subset). */
int coso_index_alloc_vectors(coso_index *ix, uint32_t n) {
    if (!ix) return COSO_ERR_INVALID;
    free(ix->codes); free(ix->mags);
    ix->n = n;
    ix->raw = NULL;
    ix->codes = (uint8_t *)calloc((size_t)n + 2, ix->cb);
    ix->mags = (float *)calloc((size_t)n + 2, sizeof(float));
    return ix->codes && ix->mags ? COSO_OK : COSO_ERR_INVALID;
}
int coso_index_quantize_rows(coso_index *ix, uint32_t start, const float *raw_chunk, uint32_t m) {
    if (!ix || !ix->codes || (uint64_t)start + m > ix->n) return COSO_ERR_INVALID;
    int rc = COSO_OK;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)m; i++) {
        int r = coso_quantize(raw_chunk + (size_t)i * ix->p.dim, (int)ix->p.dim, (int)ix->p.storage, (int)ix->p.resolution,
                              ix->p.range_lo, ix->p.range_hi, ix->codes + ((size_t)start + (size_t)i) * ix->cb, &ix->mags[start + i]);
        if (r != COSO_OK) rc = r;
    }
    return rc;
}
int coso_index_set_raw_subset(coso_index *ix, const uint32_t *ids_sorted, const float *rows, uint32_t m) {
    if (!ix) return COSO_ERR_INVALID;
    for (uint32_t i = 1; i < m; i++)
        if (ids_sorted[i] <= ids_sorted[i - 1]) return COSO_ERR_INVALID;
    ix->sub_ids = ids_sorted;
    ix->sub_rows = rows;
    ix->sub_n = m;
    return COSO_OK;
}
/* raw f32 row of an internal id for the exact rerank: the full table, or the caller's subset (NULL if absent) */
static const float *raw_row(const coso_index *ix, uint32_t id) {
    if (ix->raw) return ix->raw + (size_t)id * ix->p.dim;
    uint32_t lo = 0, hi = ix->sub_n;
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (ix->sub_ids[mid] < id) lo = mid + 1; else hi = mid;
    }
    return (lo < ix->sub_n && ix->sub_ids[lo] == id) ? ix->sub_rows + (size_t)lo * ix->p.dim : NULL;
}

const float *coso_index_root_raw(const coso_index *ix) { return ix->root_raw; }
int coso_index_set_root_raw(coso_index *ix, const float *root) {
    if (!ix->codes) return COSO_ERR_INVALID;
    memcpy(ix->root_raw, root, (size_t)ix->p.dim * 4);
    ix->has_root = 1;
    return coso_quantize(ix->root_raw, (int)ix->p.dim, (int)ix->p.storage, (int)ix->p.resolution, ix->p.range_lo,
                         ix->p.range_hi, ix->codes + (size_t)ix->n * ix->cb, &ix->mags[ix->n]);
}
const void *coso_index_codes(const coso_index *ix) { return ix->codes; }
const float *coso_index_mags(const coso_index *ix) { return ix->mags; }
void coso_index_set_ef_search(coso_index *ix, uint32_t ef) { ix->p.ef_search = ef; }
void coso_index_set_visited_mode(coso_index *ix, uint32_t mode) { ix->p.visited_mode = mode; }
/* drop every level (the vectors stay): lets one quantized corpus take several imported graphs in turn */
void coso_index_clear_graph(coso_index *ix) {
    if (!ix) return;
    ix->rounds_state_valid = 0;
    for (uint32_t l = 0; l <= ix->p.num_layers; l++) level_free(&ix->lv[l]);
}
/* level_0_neighbors_count of the NEXT graph (indexes/hnsw/types.rs:10-17): also the size of the visited filter,
 * PerformantFixedSet::new(level_0_neighbors_count) (vector_store.rs:266-270).  Drops the current graph; the vectors stay. */
int coso_index_set_level0_neighbors(coso_index *ix, uint32_t m0) {
    if (!ix || m0 == 0 || (m0 & (m0 - 1)) || m0 > 256) return COSO_ERR_INVALID;
    coso_index_clear_graph(ix);
    ix->p.level0_neighbors_count = m0;
    ix->lv[0].M = m0;
    return COSO_OK;
}
/* neighbors_count (the upper levels' M and filter size) of the NEXT graph; drops the current graph, the vectors stay */
int coso_index_set_neighbors(coso_index *ix, uint32_t m) {
    if (!ix || m == 0 || (m & (m - 1)) || m > 256) return COSO_ERR_INVALID;
    coso_index_clear_graph(ix);
    ix->p.neighbors_count = m;
    for (uint32_t l = 1; l <= ix->p.num_layers; l++) ix->lv[l].M = m;
    return COSO_OK;
}
uint32_t coso_index_level_count(const coso_index *ix, uint32_t level) { return level <= ix->p.num_layers ? ix->lv[level].n : 0; }

/* ------------------------------------------------------------------------------------------
 * scratch
 * ---------------------------------------------------------------------------------------- */
static scratch_t *scratch_new(const coso_index *ix) {
    scratch_t *s = (scratch_t *)calloc(1, sizeof(*s));
    s->heap_cap = 8192;
    s->heap = (hent *)malloc(s->heap_cap * sizeof(hent));
    s->res_cap = 1024;
    s->res = (hent *)malloc(s->res_cap * sizeof(hent));
    uint32_t Mmax = ix->p.level0_neighbors_count > ix->p.neighbors_count ? ix->p.level0_neighbors_count : ix->p.neighbors_count;
    s->visited_words = ix->p.visited_mode == COSO_VISITED_EXACT ? ((size_t)ix->n + 64) / 64 + 1 : Mmax;
    if (s->visited_words < Mmax) s->visited_words = Mmax;
    s->visited = (uint64_t *)calloc(s->visited_words, 8);
    s->touched_cap = 4096;
    s->touched = (uint32_t *)malloc(s->touched_cap * 4);
    s->qcode = (uint8_t *)malloc(ix->cb);
    return s;
}
static void scratch_free(scratch_t *s) {
    if (!s) return;
    free(s->heap); free(s->res); free(s->visited); free(s->touched); free(s->qcode); free(s);
}

static inline void heap_push(scratch_t *s, size_t *n, hent e) {
    if (*n == s->heap_cap) { s->heap_cap *= 2; s->heap = (hent *)realloc(s->heap, s->heap_cap * sizeof(hent)); }
    size_t i = (*n)++;
    hent *h = s->heap;
    while (i > 0) {
        size_t p = (i - 1) / 2;
        if (!hent_gt(&e, &h[p])) break;
        h[i] = h[p];
        i = p;
    }
    h[i] = e;
}
static inline hent heap_pop(scratch_t *s, size_t *n) {
    hent *h = s->heap;
    hent top = h[0];
    hent last = h[--(*n)];
    size_t i = 0, cnt = *n;
    for (;;) {
        size_t c = 2 * i + 1;
        if (c >= cnt) break;
        if (c + 1 < cnt && hent_gt(&h[c + 1], &h[c])) c++;
        if (!hent_gt(&h[c], &last)) break;
        h[i] = h[c];
        i = c;
    }
    if (cnt) h[i] = last;
    return top;
}
static int cmp_hent_desc(const void *a, const void *b) {
    const hent *x = (const hent *)a, *y = (const hent *)b;
    return hent_gt(x, y) ? -1 : (hent_gt(y, x) ? 1 : 0);
}

/* visited filter: REF = PerformantFixedSet::new(M_level) (vector_store.rs:266-270), EXACT = bitset over rows */
static inline int visited_test(const coso_index *ix, const scratch_t *s, uint32_t M, uint32_t id) {
    if (ix->p.visited_mode == COSO_VISITED_EXACT) {
        if (id == COSO_QUERY_ID) return 0;
        uint32_t r = row_of(ix, id);
        return (int)((s->visited[r >> 6] >> (r & 63)) & 1ull);
    }
    return (int)((s->visited[(id >> 6) & (M - 1)] >> (id & 63)) & 1ull);
}
static inline void visited_set(const coso_index *ix, scratch_t *s, uint32_t M, uint32_t id) {
    if (ix->p.visited_mode == COSO_VISITED_EXACT) {
        if (id == COSO_QUERY_ID) return;
        uint32_t r = row_of(ix, id);
        if (s->visited[r >> 6] == 0) {
            if (s->ntouched == s->touched_cap) { s->touched_cap *= 2; s->touched = (uint32_t *)realloc(s->touched, s->touched_cap * 4); }
            s->touched[s->ntouched++] = r >> 6;
        }
        s->visited[r >> 6] |= 1ull << (r & 63);
        return;
    }
    s->visited[(id >> 6) & (M - 1)] |= 1ull << (id & 63);
}

static inline int node_distance(const coso_index *ix, const uint8_t *qcode, float qmag, uint32_t row, float *out) {
    return coso_distance((int)ix->p.metric, (int)ix->p.storage, (int)ix->p.resolution, (int)ix->p.dim, qcode, qmag,
                         ix->codes + (size_t)row * ix->cb, ix->mags[row], out);
}

/* traverse_find_nearest (vector_store.rs:1112-1204) on one level.
 * self_id: id pre-inserted in the visited filter (query id :271, or the new node's id :807).
 * Returns number of results in s->res (sorted desc), or a negative status. */
static int walk_level_ex(const coso_index *ix, uint32_t level, uint32_t entry_idx, const uint8_t *qcode, float qmag,
                         uint32_t self_id, int seed_self, uint32_t ef, uint32_t keep, scratch_t *s, coso_stats *st);
static int walk_level(const coso_index *ix, uint32_t level, uint32_t entry_idx, const uint8_t *qcode, float qmag,
                      uint32_t self_id, uint32_t ef, uint32_t keep, scratch_t *s, coso_stats *st) {
    return walk_level_ex(ix, level, entry_idx, qcode, qmag, self_id, 1, ef, keep, s, st);
}
/* seed_self = 0: delete_embedding's walks (vector_store.rs:1232-1248) start from a filter nothing was inserted into */
static int walk_level_ex(const coso_index *ix, uint32_t level, uint32_t entry_idx, const uint8_t *qcode, float qmag,
                         uint32_t self_id, int seed_self, uint32_t ef, uint32_t keep, scratch_t *s, coso_stats *st) {
    const level_t *L = &ix->lv[level];
    const uint32_t M = L->M;
    const int metric = (int)ix->p.metric;
    uint32_t slots = M < ix->p.shortlist_size ? M : ix->p.shortlist_size; /* .take(shortlist_size) :1164 */
    if (ix->p.visited_mode == COSO_VISITED_EXACT) {
        for (size_t t = 0; t < s->ntouched; t++) s->visited[s->touched[t]] = 0;
        s->ntouched = 0;
    } else memset(s->visited, 0, (size_t)M * 8);
    if (seed_self) visited_set(ix, s, M, self_id);

    size_t hn = 0, rn = 0;
    float d0;
    int rc = node_distance(ix, qcode, qmag, row_of(ix, L->node_id[entry_idx]), &d0);
    if (st) st->evals++;
    if (rc != COSO_OK) return -rc;
    visited_set(ix, s, M, L->node_id[entry_idx]);
    hent e0 = {order_key(metric, d0), L->node_id[entry_idx], entry_idx, d0};
    heap_push(s, &hn, e0);

    uint32_t nodes_visited = 0;
    while (hn > 0) {
        hent cur = heap_pop(s, &hn);
        if (nodes_visited >= ef) break; /* the popped element is discarded */
        nodes_visited++;
        if (rn == s->res_cap) { s->res_cap *= 2; s->res = (hent *)realloc(s->res, s->res_cap * sizeof(hent)); }
        s->res[rn++] = cur;
        if (st) { st->expansions++; st->adj_bytes += (uint64_t)M * 4; }
        const uint32_t *nb = L->nbr + (size_t)cur.idx * M;
        for (uint32_t j = 0; j < slots; j++) {
            uint32_t nidx = nb[j];
            if (nidx == IDX_NONE) continue;
            uint32_t nid = L->node_id[nidx];
            if (visited_test(ix, s, M, nid)) continue;
            float d;
            rc = node_distance(ix, qcode, qmag, row_of(ix, nid), &d);
            if (st) st->evals++;
            if (rc != COSO_OK) return -rc;
            visited_set(ix, s, M, nid);
            hent e = {order_key(metric, d), nid, nidx, d};
            heap_push(s, &hn, e);
        }
    }
    qsort(s->res, rn, sizeof(hent), cmp_hent_desc); /* select_nth + truncate + sort desc :1194-1201 */
    if (rn > keep) rn = keep;
    return (int)rn;
}

/* ann_search (vector_store.rs:256-402): every level runs the full ef-bounded walk; results of all
 * levels are concatenated top level first. out must hold (num_layers+1)*KEEP_SEARCH entries. */
static int ann_search_internal(const coso_index *ix, const uint8_t *qcode, float qmag, scratch_t *s, hent *out,
                               uint32_t *level_counts, coso_stats *st) {
    const uint32_t Ltop = ix->p.num_layers;
    uint32_t entry = ix->lv[Ltop].root_idx;
    if (entry == IDX_NONE) return -COSO_ERR_INVALID;
    int total = 0;
    for (int level = (int)Ltop; level >= 0; level--) {
        int cnt = walk_level(ix, (uint32_t)level, entry, qcode, qmag, COSO_QUERY_ID, ix->p.ef_search, KEEP_SEARCH, s, st);
        if (cnt < 0) return cnt;
        if (cnt == 0) { /* vector_store.rs:329-380 fallback: the entry node's own distance */
            const level_t *L = &ix->lv[level];
            float d;
            int rc = node_distance(ix, qcode, qmag, row_of(ix, L->node_id[entry]), &d);
            if (rc != COSO_OK) return -rc;
            hent e = {order_key((int)ix->p.metric, d), L->node_id[entry], entry, d};
            s->res[0] = e;
            cnt = 1;
        }
        memcpy(out + total, s->res, (size_t)cnt * sizeof(hent));
        if (level_counts) level_counts[Ltop - (uint32_t)level] = (uint32_t)cnt;
        total += cnt;
        if (level > 0) entry = ix->lv[level].child[s->res[0].idx];
    }
    return total;
}

int coso_ann_search(const coso_index *ix, const float *query, uint32_t *out_ids, float *out_sims, uint32_t *level_counts) {
    scratch_t *s = scratch_new(ix);
    float qmag;
    int rc = coso_quantize(query, (int)ix->p.dim, (int)ix->p.storage, (int)ix->p.resolution, ix->p.range_lo, ix->p.range_hi,
                           s->qcode, &qmag);
    if (rc != COSO_OK) { scratch_free(s); return -rc; }
    hent *all = (hent *)malloc((size_t)(ix->p.num_layers + 1) * KEEP_SEARCH * sizeof(hent));
    int total = ann_search_internal(ix, s->qcode, qmag, s, all, level_counts, NULL);
    for (int i = 0; i < total; i++) { out_ids[i] = all[i].id; out_sims[i] = all[i].sim; }
    free(all);
    scratch_free(s);
    return total;
}

typedef struct { float cs; uint32_t id; } fent;
static int cmp_fent_desc(const void *a, const void *b) {
    const fent *x = (const fent *)a, *y = (const fent *)b;
    int32_t kx = total_key(x->cs), ky = total_key(y->cs);
    if (kx != ky) return kx > ky ? -1 : 1;
    return x->id > y->id ? -1 : (x->id < y->id ? 1 : 0);
}

/* search_internal (indexes/hnsw/mod.rs:390-440) for one query */
/* candidates_only: stop after remove_duplicates_and_filter and return the <= 5k ids the rerank would read (out_ids [5k]) */
static int search_one(const coso_index *ix, const float *q, uint32_t top_k, scratch_t *s, hent *all, fent *f, uint32_t *out_ids,
                      float *out_scores, uint32_t *out_count, coso_stats *st, int candidates_only) {
    float qmag;
    int rc = coso_quantize(q, (int)ix->p.dim, (int)ix->p.storage, (int)ix->p.resolution, ix->p.range_lo, ix->p.range_hi,
                           s->qcode, &qmag);
    if (rc != COSO_OK) return rc;
    int total = ann_search_internal(ix, s->qcode, qmag, s, all, NULL, st);
    if (total < 0) return -total;
    /* remove_duplicates_and_filter (common.rs:381-412): first-seen dedup, drop root, sort desc, keep 5k */
    int m = 0;
    for (int i = 0; i < total; i++) {
        int dup = 0;
        for (int j = 0; j < m; j++)
            if (all[j].id == all[i].id) { dup = 1; break; }
        if (dup) continue;
        all[m++] = all[i];
    }
    int w = 0;
    for (int i = 0; i < m; i++)
        if (all[i].id != COSO_ROOT_ID) all[w++] = all[i];
    m = w;
    qsort(all, (size_t)m, sizeof(hent), cmp_hent_desc);
    if ((uint32_t)m > 5 * top_k) m = (int)(5 * top_k);
    if (candidates_only) {
        for (int i = 0; i < m; i++) out_ids[i] = all[i].id;
        *out_count = (uint32_t)m;
        return COSO_OK;
    }
    /* finalize_ann_results (vector_store.rs:404-445): exact cosine on RAW f32, norm recomputed per candidate */
    const int d = (int)ix->p.dim;
    float mag_query = coso_seq_norm_f32(q, d);
    for (int i = 0; i < m; i++) {
        uint32_t id = all[i].id;
        const float *rv = raw_row(ix, row_of(ix, id));
        if (!rv) return COSO_ERR_INVALID; /* raw subset does not cover this candidate */
        float dp = coso_dot_f32(q, rv, d);
        float mag_raw = coso_seq_norm_f32(rv, d);
        float cs = dp / (mag_query * mag_raw);
        fent t = {cs, id};
        f[i] = t;
    }
    qsort(f, (size_t)m, sizeof(fent), cmp_fent_desc);
    if ((uint32_t)m > top_k) m = (int)top_k;
    for (int i = 0; i < m; i++) { out_ids[i] = f[i].id; out_scores[i] = f[i].cs; }
    *out_count = (uint32_t)m;
    return COSO_OK;
}

int coso_search_batch(const coso_index *ix, const float *queries, uint32_t B, uint32_t top_k, uint32_t *out_ids,
                      float *out_scores, uint32_t *out_counts, int32_t *out_status, coso_stats *stats, int threads) {
    if (!ix || (!ix->raw && !ix->sub_rows) || top_k == 0) return COSO_ERR_INVALID;
    int first_err = COSO_OK;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        scratch_t *s = scratch_new(ix);
        hent *all = (hent *)malloc((size_t)(ix->p.num_layers + 1) * KEEP_SEARCH * sizeof(hent));
        fent *fbuf = (fent *)malloc((size_t)(ix->p.num_layers + 1) * KEEP_SEARCH * sizeof(fent));
#pragma omp for schedule(dynamic, 1)
        for (int64_t b = 0; b < (int64_t)B; b++) {
            coso_stats st = {0, 0, 0};
            uint32_t cnt = 0;
            int rc = search_one(ix, queries + (size_t)b * ix->p.dim, top_k, s, all, fbuf, out_ids + (size_t)b * top_k,
                                out_scores + (size_t)b * top_k, &cnt, &st, 0);
            out_counts[b] = rc == COSO_OK ? cnt : 0;
            if (out_status) out_status[b] = rc;
            if (stats) stats[b] = st;
            if (rc != COSO_OK) {
#pragma omp critical
                if (first_err == COSO_OK) first_err = rc;
            }
        }
        free(all);
        free(fbuf);
        scratch_free(s);
    }
    return first_err;
}

/* walk + remove_duplicates_and_filter only: the <= 5*top_k ids per query whose raw rows finalize_ann_results reads
 * (out_ids [B][5*top_k]).  Lets a caller that cannot hold the raw table on the host fetch exactly those rows. */
int coso_candidates_batch(const coso_index *ix, const float *queries, uint32_t B, uint32_t top_k, uint32_t *out_ids,
                          uint32_t *out_counts, int threads) {
    if (!ix || top_k == 0) return COSO_ERR_INVALID;
    int first_err = COSO_OK;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        scratch_t *s = scratch_new(ix);
        hent *all = (hent *)malloc((size_t)(ix->p.num_layers + 1) * KEEP_SEARCH * sizeof(hent));
#pragma omp for schedule(dynamic, 1)
        for (int64_t b = 0; b < (int64_t)B; b++) {
            uint32_t cnt = 0;
            int rc = search_one(ix, queries + (size_t)b * ix->p.dim, top_k, s, all, NULL, out_ids + (size_t)b * 5 * top_k, NULL, &cnt, NULL, 1);
            out_counts[b] = rc == COSO_OK ? cnt : 0;
            if (rc != COSO_OK) {
#pragma omp critical
                if (first_err == COSO_OK) first_err = rc;
            }
        }
        free(all);
        scratch_free(s);
    }
    return first_err;
}

/* ------------------------------------------------------------------------------------------
 * deterministic builder (vector_store.rs:714-1109, prob_node.rs:210-329)
 * ---------------------------------------------------------------------------------------- */
static uint64_t splitmix64(uint64_t *st) {
    uint64_t z = (*st += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static float rand_f32(uint64_t *st) { return (float)(splitmix64(st) >> 40) * (1.0f / 16777216.0f); } /* 24-bit U[0,1) like rand::random::<f32>() */

static uint32_t level_append(level_t *L, uint32_t id, int metric) {
    if (L->n == L->cap) {
        uint32_t nc = L->cap ? L->cap * 2 : 1024;
        L->node_id = (uint32_t *)realloc(L->node_id, (size_t)nc * 4);
        L->nbr = (uint32_t *)realloc(L->nbr, (size_t)nc * L->M * 4);
        L->nbr_sim = (float *)realloc(L->nbr_sim, (size_t)nc * L->M * 4);
        L->child = (uint32_t *)realloc(L->child, (size_t)nc * 4);
        L->low_idx = (uint8_t *)realloc(L->low_idx, nc);
        L->low_sim = (float *)realloc(L->low_sim, (size_t)nc * 4);
        L->cap = nc;
    }
    uint32_t i = L->n++;
    L->node_id[i] = id;
    for (uint32_t j = 0; j < L->M; j++) { L->nbr[(size_t)i * L->M + j] = IDX_NONE; L->nbr_sim[(size_t)i * L->M + j] = 0.0f; }
    L->child[i] = IDX_NONE;
    L->low_idx[i] = 0;                    /* prob_node.rs:140 */
    L->low_sim[i] = metric_min(metric);
    return i;
}

static void remove_neighbor_by_idx(level_t *L, uint32_t node, uint32_t target) { /* remove_neighbor_by_id :285-306 */
    uint32_t *nb = L->nbr + (size_t)node * L->M;
    for (uint32_t j = 0; j < L->M; j++)
        if (nb[j] == target) { nb[j] = IDX_NONE; return; }
}

/* ProbNode::add_neighbor (prob_node.rs:210-283). Returns slot index or -1. */
static int add_neighbor(level_t *L, int metric, uint32_t self, uint32_t nbr, float dist) {
    const uint32_t M = L->M;
    uint32_t lowest_idx = L->low_idx[self];
    float lowest_sim = L->low_sim[self];
    if (coso_metric_cmp(metric, dist, lowest_sim) <= 0) return -1;
    uint32_t *nb = L->nbr + (size_t)self * M;
    float *ns = L->nbr_sim + (size_t)self * M;
    int ok = (nb[lowest_idx] == IDX_NONE) || coso_metric_cmp(metric, dist, ns[lowest_idx]) > 0;
    uint32_t old = IDX_NONE;
    if (ok) { old = nb[lowest_idx]; nb[lowest_idx] = nbr; ns[lowest_idx] = dist; }
    uint32_t nl = 0;
    float nsim = metric_max(metric);
    for (uint32_t j = 0; j < M; j++) {
        if (nb[j] == IDX_NONE) { nsim = metric_min(metric); nl = j; break; }
        if (coso_metric_cmp(metric, ns[j], nsim) < 0) { nsim = ns[j]; nl = j; }
    }
    L->low_idx[self] = (uint8_t)nl;
    L->low_sim[self] = nsim;
    if (!ok) return -1;
    if (old != IDX_NONE) remove_neighbor_by_idx(L, old, self); /* evictee drops its back edge; its cache is NOT refreshed */
    return (int)lowest_idx;
}

typedef struct { uint32_t idx; float sim; } zent;

static void create_node_edges(coso_index *ix, uint32_t level, uint32_t node, const zent *z, int zn) { /* :976-1074 */
    level_t *L = &ix->lv[level];
    const int metric = (int)ix->p.metric;
    uint32_t succ = 0;
    for (int i = 0; i < zn; i++) {
        if (succ >= L->M) break;
        int r = add_neighbor(L, metric, node, z[i].idx, z[i].sim);
        if (r >= 0) {
            int r2 = add_neighbor(L, metric, z[i].idx, node, z[i].sim);
            if (r2 >= 0) succ++;
            else if (L->nbr[(size_t)node * L->M + (uint32_t)r] == z[i].idx) L->nbr[(size_t)node * L->M + (uint32_t)r] = IDX_NONE; /* remove_neighbor_by_index_and_id */
        }
    }
}

static int index_embedding(coso_index *ix, uint32_t id, uint32_t parent_idx, uint32_t entry_idx, int level, int max_level,
                           scratch_t *s) { /* :782-937 */
    const uint32_t row = row_of(ix, id);
    const uint8_t *code = ix->codes + (size_t)row * ix->cb;
    int zn = walk_level(ix, (uint32_t)level, entry_idx, code, ix->mags[row], id, ix->p.ef_construction, KEEP_INDEX, s, NULL);
    if (zn < 0) return -zn;
    zent z[KEEP_INDEX];
    if (zn == 0) {
        float d;
        int rc = node_distance(ix, code, ix->mags[row], row_of(ix, ix->lv[level].node_id[entry_idx]), &d);
        if (rc != COSO_OK) return rc;
        z[0].idx = entry_idx; z[0].sim = d; zn = 1;
    } else {
        for (int i = 0; i < zn; i++) { z[i].idx = s->res[i].idx; z[i].sim = s->res[i].sim; }
    }
    uint32_t child = level > 0 ? ix->lv[level].child[z[0].idx] : IDX_NONE;
    if (level > max_level) {
        if (level != 0) return index_embedding(ix, id, IDX_NONE, child, level - 1, max_level, s);
        return COSO_OK;
    }
    uint32_t me = level_append(&ix->lv[level], id, (int)ix->p.metric);
    if (parent_idx != IDX_NONE) ix->lv[level + 1].child[parent_idx] = me;
    if (level != 0) {
        int rc = index_embedding(ix, id, me, child, level - 1, max_level, s);
        if (rc != COSO_OK) return rc;
    }
    create_node_edges(ix, (uint32_t)level, me, z, zn);
    return COSO_OK;
}

int coso_index_build(coso_index *ix) {
    if (!ix || !ix->codes) return COSO_ERR_INVALID;
    const uint32_t Ltop = ix->p.num_layers;
    uint64_t rng = ix->p.seed ? ix->p.seed : 0x1234567ull;
    for (uint32_t l = 0; l <= Ltop; l++) level_free(&ix->lv[l]);
    /* create_root_node (vector_store.rs:44-150): random vector in values_range, id u32::MAX, one node per level */
    for (uint32_t i = 0; i < ix->p.dim; i++) ix->root_raw[i] = ix->p.range_lo + rand_f32(&rng) * (ix->p.range_hi - ix->p.range_lo);
    int rc = coso_index_set_root_raw(ix, ix->root_raw);
    if (rc != COSO_OK) return rc;
    for (uint32_t l = 0; l <= Ltop; l++) {
        uint32_t r = level_append(&ix->lv[l], COSO_ROOT_ID, (int)ix->p.metric);
        ix->lv[l].root_idx = r;
        if (l > 0) ix->lv[l].child[r] = ix->lv[l - 1].root_idx;
    }
    double *pv = (double *)malloc((Ltop + 1) * sizeof(double));
    uint8_t *pl = (uint8_t *)malloc(Ltop + 1);
    coso_level_probs(4.0, (int)Ltop, pv, pl); /* api_service.rs:109 */
    scratch_t *s = scratch_new(ix);
    for (uint32_t r = 0; r < ix->n && rc == COSO_OK; r++) {
        double x = (double)rand_f32(&rng);
        int max_level = coso_max_insert_level(x, pv, pl, (int)Ltop + 1);
        rc = index_embedding(ix, id_of_row(ix, r), IDX_NONE, ix->lv[Ltop].root_idx, (int)Ltop, max_level, s);
    }
    scratch_free(s);
    free(pv); free(pl);
    return rc;
}

/* Batch-synchronous variant of the builder — the CPU statement of cosdata_amd/csrc/builder.hip:
 * every id of a batch walks the graph snapshot that precedes the batch (all levels, ef_construction,
 * keep 64, visited pre-seeded with its id); then nodes are created and edges connected level by
 * level in id order with the same create_node_edges/add_neighbor semantics.  Batch sizes follow
 * min(Bmax, max(1, inserted/4)).  With Bmax = 1 this is NOT coso_index_build (there the lower
 * levels are linked before the upper level's walk result is used) but differs only in schedule. */
int coso_index_build_batched(coso_index *ix, uint32_t batch_size) {
    if (!ix || !ix->codes) return COSO_ERR_INVALID;
    const uint32_t Ltop = ix->p.num_layers, L1 = Ltop + 1;
    const uint32_t Bmax = batch_size ? batch_size : 4096u;
    uint64_t rng = ix->p.seed ? ix->p.seed : 0x1234567ull;
    for (uint32_t l = 0; l <= Ltop; l++) level_free(&ix->lv[l]);
    for (uint32_t i = 0; i < ix->p.dim; i++) ix->root_raw[i] = ix->p.range_lo + rand_f32(&rng) * (ix->p.range_hi - ix->p.range_lo);
    int rc = coso_index_set_root_raw(ix, ix->root_raw);
    if (rc != COSO_OK) return rc;
    for (uint32_t l = 0; l <= Ltop; l++) {
        uint32_t r = level_append(&ix->lv[l], COSO_ROOT_ID, (int)ix->p.metric);
        ix->lv[l].root_idx = r;
        if (l > 0) ix->lv[l].child[r] = ix->lv[l - 1].root_idx;
    }
    double *pv = (double *)malloc(L1 * sizeof(double));
    uint8_t *pl = (uint8_t *)malloc(L1);
    coso_level_probs(4.0, (int)Ltop, pv, pl);
    uint8_t *max_level = (uint8_t *)malloc(ix->n ? ix->n : 1);
    for (uint32_t id = 0; id < ix->n; id++) max_level[id] = (uint8_t)coso_max_insert_level((double)rand_f32(&rng), pv, pl, (int)L1);
    scratch_t *s = scratch_new(ix);
    zent *z = (zent *)malloc((size_t)Bmax * L1 * KEEP_INDEX * sizeof(zent));
    uint32_t *zn = (uint32_t *)malloc((size_t)Bmax * L1 * 4);
    uint32_t *me = (uint32_t *)malloc((size_t)Bmax * L1 * 4);
    uint32_t inserted = 0;
    while (inserted < ix->n && rc == COSO_OK) {
        uint32_t bs = inserted / 4u;
        if (bs < 1) bs = 1;
        if (bs > Bmax) bs = Bmax;
        if (bs > ix->n - inserted) bs = ix->n - inserted;
        /* 1. walks on the snapshot */
        for (uint32_t b = 0; b < bs && rc == COSO_OK; b++) {
            const uint32_t row = inserted + b, id = id_of_row(ix, row);
            const uint8_t *code = ix->codes + (size_t)row * ix->cb;
            uint32_t entry = ix->lv[Ltop].root_idx;
            for (int level = (int)Ltop; level >= 0; level--) {
                zent *zz = z + ((size_t)b * L1 + (uint32_t)level) * KEEP_INDEX;
                int cnt = walk_level(ix, (uint32_t)level, entry, code, ix->mags[row], id, ix->p.ef_construction, KEEP_INDEX, s, NULL);
                if (cnt < 0) { rc = -cnt; break; }
                if (cnt == 0) {
                    float d;
                    rc = node_distance(ix, code, ix->mags[row], row_of(ix, ix->lv[level].node_id[entry]), &d);
                    if (rc != COSO_OK) break;
                    zz[0].idx = entry; zz[0].sim = d; cnt = 1;
                } else
                    for (int i = 0; i < cnt; i++) { zz[i].idx = s->res[i].idx; zz[i].sim = s->res[i].sim; }
                zn[(size_t)b * L1 + (uint32_t)level] = (uint32_t)cnt;
                if (level > 0) entry = ix->lv[level].child[zz[0].idx];
            }
        }
        if (rc != COSO_OK) break;
        /* 2. create the nodes of the batch (child links top-down) */
        for (uint32_t b = 0; b < bs; b++) {
            const uint32_t id = id_of_row(ix, inserted + b);
            uint32_t parent = IDX_NONE;
            for (int level = (int)max_level[inserted + b]; level >= 0; level--) {
                uint32_t m = level_append(&ix->lv[level], id, (int)ix->p.metric);
                me[(size_t)b * L1 + (uint32_t)level] = m;
                if (parent != IDX_NONE) ix->lv[level + 1].child[parent] = m;
                parent = m;
            }
        }
        /* 3. connect edges level by level, new nodes in id order */
        for (uint32_t l = 0; l <= Ltop; l++)
            for (uint32_t b = 0; b < bs; b++) {
                if (max_level[inserted + b] < l) continue;
                create_node_edges(ix, l, me[(size_t)b * L1 + l], z + ((size_t)b * L1 + l) * KEEP_INDEX, (int)zn[(size_t)b * L1 + l]);
            }
        inserted += bs;
    }
    free(z); free(zn); free(me); free(max_level); free(pv); free(pl);
    scratch_free(s);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * PROTOTYPE (not yet mirrored by the device builder): round-synchronous link schedule, the plan of
 * DESIGN.md §10.1 for moving the link phase to the GPU.  Walks and node creation are those of
 * coso_index_build_batched.  Per level, the batch's nodes are linked in ROUNDS: pending nodes, in
 * id order, claim {self} + their candidates; a node runs in the current round iff it is the first
 * claimer of every row it claimed (a blocked node keeps its claims, so nodes that conflict keep
 * their id order).  Rows outside a node's own claim are only ever touched through evictions
 * (the evictee drops its back edge, prob_node.rs:271-279): those removals are queued and applied
 * in (evictor id, occurrence) order when the round ends.  Runnable nodes of a round touch disjoint
 * rows, so they can execute concurrently; the result does not depend on their interleaving.
 * stats[0] = rounds summed over (batch, level), stats[1] = (batch, level) pairs with >= 1 node,
 * stats[2] = nodes linked, stats[3] = nodes that ran in the first round of their (batch, level).
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint32_t old_idx, target; } evict_t;

static int add_neighbor_deferred(level_t *L, int metric, uint32_t self, uint32_t nbr, float dist, uint32_t *evicted) {
    const uint32_t M = L->M;
    uint32_t lowest_idx = L->low_idx[self];
    float lowest_sim = L->low_sim[self];
    *evicted = IDX_NONE;
    if (coso_metric_cmp(metric, dist, lowest_sim) <= 0) return -1;
    uint32_t *nb = L->nbr + (size_t)self * M;
    float *ns = L->nbr_sim + (size_t)self * M;
    int ok = (nb[lowest_idx] == IDX_NONE) || coso_metric_cmp(metric, dist, ns[lowest_idx]) > 0;
    uint32_t old = IDX_NONE;
    if (ok) { old = nb[lowest_idx]; nb[lowest_idx] = nbr; ns[lowest_idx] = dist; }
    uint32_t nl = 0;
    float nsim = metric_max(metric);
    for (uint32_t j = 0; j < M; j++) {
        if (nb[j] == IDX_NONE) { nsim = metric_min(metric); nl = j; break; }
        if (coso_metric_cmp(metric, ns[j], nsim) < 0) { nsim = ns[j]; nl = j; }
    }
    L->low_idx[self] = (uint8_t)nl;
    L->low_sim[self] = nsim;
    if (!ok) return -1;
    *evicted = old; /* the caller removes `self` from old's row now (own claim) or at the end of the round */
    return (int)lowest_idx;
}

/* the batch loop of coso_index_build_rounds for the vectors [first_row, ix->n): max_level_of[i] is the level of vector first_row + i */
static int build_rounds_range(coso_index *ix, uint32_t first_row, uint32_t Bmax, int greedy, const uint8_t *max_level_of, uint64_t *st) {
    const uint32_t Ltop = ix->p.num_layers, L1 = Ltop + 1;
    const int metric = (int)ix->p.metric;
    const uint8_t *max_level = max_level_of - first_row; /* indexed by vector row below */
    int rc = COSO_OK;
    scratch_t *s = scratch_new(ix);
    zent *z = (zent *)malloc((size_t)Bmax * L1 * KEEP_INDEX * sizeof(zent));
    uint32_t *zn = (uint32_t *)malloc((size_t)Bmax * L1 * 4);
    uint32_t *me = (uint32_t *)malloc((size_t)Bmax * L1 * 4);
    /* claim table per node index of a level (level 0 is the largest: n + 1 nodes) */
    uint32_t *claim_round = (uint32_t *)calloc((size_t)ix->n + 2, 4), *claim_owner = (uint32_t *)calloc((size_t)ix->n + 2, 4);
    uint32_t round_id = 0;
    uint32_t *pending = (uint32_t *)malloc((size_t)Bmax * 4), *next = (uint32_t *)malloc((size_t)Bmax * 4);
    evict_t *q = (evict_t *)malloc((size_t)Bmax * 2 * KEEP_INDEX * sizeof(evict_t));
    uint32_t inserted = first_row;
    while (inserted < ix->n && rc == COSO_OK) {
        uint32_t bs = inserted / 4u;
        if (bs < 1) bs = 1;
        if (bs > Bmax) bs = Bmax;
        if (bs > ix->n - inserted) bs = ix->n - inserted;
        for (uint32_t b = 0; b < bs && rc == COSO_OK; b++) { /* 1. walks on the snapshot (as coso_index_build_batched) */
            const uint32_t row = inserted + b, id = id_of_row(ix, row);
            const uint8_t *code = ix->codes + (size_t)row * ix->cb;
            uint32_t entry = ix->lv[Ltop].root_idx;
            for (int level = (int)Ltop; level >= 0; level--) {
                zent *zz = z + ((size_t)b * L1 + (uint32_t)level) * KEEP_INDEX;
                int cnt = walk_level(ix, (uint32_t)level, entry, code, ix->mags[row], id, ix->p.ef_construction, KEEP_INDEX, s, NULL);
                if (cnt < 0) { rc = -cnt; break; }
                if (cnt == 0) {
                    float d;
                    rc = node_distance(ix, code, ix->mags[row], row_of(ix, ix->lv[level].node_id[entry]), &d);
                    if (rc != COSO_OK) break;
                    zz[0].idx = entry; zz[0].sim = d; cnt = 1;
                } else
                    for (int i = 0; i < cnt; i++) { zz[i].idx = s->res[i].idx; zz[i].sim = s->res[i].sim; }
                zn[(size_t)b * L1 + (uint32_t)level] = (uint32_t)cnt;
                if (level > 0) entry = ix->lv[level].child[zz[0].idx];
            }
        }
        if (rc != COSO_OK) break;
        for (uint32_t b = 0; b < bs; b++) { /* 2. nodes of the batch */
            const uint32_t id = id_of_row(ix, inserted + b);
            uint32_t parent = IDX_NONE;
            for (int level = (int)max_level[inserted + b]; level >= 0; level--) {
                uint32_t m = level_append(&ix->lv[level], id, metric);
                me[(size_t)b * L1 + (uint32_t)level] = m;
                if (parent != IDX_NONE) ix->lv[level + 1].child[parent] = m;
                parent = m;
            }
        }
        for (uint32_t l = 0; l <= Ltop; l++) { /* 3. rounds */
            level_t *L = &ix->lv[l];
            uint32_t np = 0;
            for (uint32_t b = 0; b < bs; b++) if (max_level[inserted + b] >= l) pending[np++] = b;
            if (np == 0) continue;
            st[1]++;
            st[2] += np;
            int first = 1;
            while (np > 0) {
                round_id++;
                st[0]++;
                uint32_t nn = 0, qn = 0;
                for (uint32_t k = 0; k < np; k++) {
                    const uint32_t b = pending[k], node = me[(size_t)b * L1 + l];
                    const zent *zz = z + ((size_t)b * L1 + l) * KEEP_INDEX;
                    const int cnt = (int)zn[(size_t)b * L1 + l];
                    int runnable = 1;
                    if (!greedy) {
                        /* ordered: claim {self} + candidates; first claimer of a row in this round owns it, blocked nodes keep
                         * their claims so that conflicting nodes are linked in id order */
                        for (int i = -1; i < cnt; i++) {
                            const uint32_t r = i < 0 ? node : zz[i].idx;
                            if (claim_round[r] == round_id) { if (claim_owner[r] != b) runnable = 0; }
                            else { claim_round[r] = round_id; claim_owner[r] = b; }
                        }
                    } else {
                        /* greedy: only RUNNING nodes hold claims (maximal independent set in id order); conflicting nodes may
                         * be linked out of id order, still deterministic */
                        for (int i = -1; i < cnt && runnable; i++) {
                            const uint32_t r = i < 0 ? node : zz[i].idx;
                            if (claim_round[r] == round_id && claim_owner[r] != b) runnable = 0;
                        }
                        if (runnable)
                            for (int i = -1; i < cnt; i++) {
                                const uint32_t r = i < 0 ? node : zz[i].idx;
                                claim_round[r] = round_id; claim_owner[r] = b;
                            }
                    }
                    if (!runnable) { next[nn++] = b; continue; }
                    if (first) st[3]++;
                    /* create_node_edges (vector_store.rs:976-1074) with evictions outside the own claim queued */
                    uint32_t succ = 0;
                    for (int i = 0; i < cnt; i++) {
                        if (succ >= L->M) break;
                        uint32_t ev;
                        int r = add_neighbor_deferred(L, metric, node, zz[i].idx, zz[i].sim, &ev);
                        if (ev != IDX_NONE) {
                            if (claim_round[ev] == round_id && claim_owner[ev] == b) remove_neighbor_by_idx(L, ev, node);
                            else { q[qn].old_idx = ev; q[qn].target = node; qn++; }
                        }
                        if (r >= 0) {
                            int r2 = add_neighbor_deferred(L, metric, zz[i].idx, node, zz[i].sim, &ev);
                            if (ev != IDX_NONE) {
                                if (claim_round[ev] == round_id && claim_owner[ev] == b) remove_neighbor_by_idx(L, ev, zz[i].idx);
                                else { q[qn].old_idx = ev; q[qn].target = zz[i].idx; qn++; }
                            }
                            if (r2 >= 0) succ++;
                            else if (L->nbr[(size_t)node * L->M + (uint32_t)r] == zz[i].idx) L->nbr[(size_t)node * L->M + (uint32_t)r] = IDX_NONE;
                        }
                    }
                }
                for (uint32_t e = 0; e < qn; e++) remove_neighbor_by_idx(L, q[e].old_idx, q[e].target); /* end of round */
                uint32_t *t = pending; pending = next; next = t;
                np = nn;
                first = 0;
            }
        }
        inserted += bs;
    }
    free(z); free(zn); free(me); free(claim_round); free(claim_owner); free(pending); free(next); free(q);
    scratch_free(s);
    return rc;
}

int coso_index_build_rounds(coso_index *ix, uint32_t batch_size, int greedy, uint64_t *stats) {
    if (!ix || !ix->codes) return COSO_ERR_INVALID;
    const uint32_t Ltop = ix->p.num_layers, L1 = Ltop + 1;
    const uint32_t Bmax = batch_size ? batch_size : 4096u;
    const int metric = (int)ix->p.metric;
    uint64_t rng = ix->p.seed ? ix->p.seed : 0x1234567ull;
    uint64_t st[4] = {0, 0, 0, 0};
    ix->rounds_state_valid = 0;
    for (uint32_t l = 0; l <= Ltop; l++) level_free(&ix->lv[l]);
    for (uint32_t i = 0; i < ix->p.dim; i++) ix->root_raw[i] = ix->p.range_lo + rand_f32(&rng) * (ix->p.range_hi - ix->p.range_lo);
    int rc = coso_index_set_root_raw(ix, ix->root_raw);
    if (rc != COSO_OK) return rc;
    for (uint32_t l = 0; l <= Ltop; l++) {
        uint32_t r = level_append(&ix->lv[l], COSO_ROOT_ID, metric);
        ix->lv[l].root_idx = r;
        if (l > 0) ix->lv[l].child[r] = ix->lv[l - 1].root_idx;
    }
    double *pv = (double *)malloc(L1 * sizeof(double));
    uint8_t *pl = (uint8_t *)malloc(L1);
    coso_level_probs(4.0, (int)Ltop, pv, pl);
    uint8_t *max_level = (uint8_t *)malloc(ix->n ? ix->n : 1);
    for (uint32_t id = 0; id < ix->n; id++) max_level[id] = (uint8_t)coso_max_insert_level((double)rand_f32(&rng), pv, pl, (int)L1);
    rc = build_rounds_range(ix, 0, Bmax, greedy, max_level, st);
    if (rc == COSO_OK) { ix->rng_state = rng; ix->n_built = ix->n; ix->rounds_state_valid = 1; ix->rounds_greedy = greedy; }
    if (stats) memcpy(stats, st, sizeof(st));
    free(max_level); free(pv); free(pl);
    return rc;
}

/* index_embeddings on a LIVE index (vector_store.rs:714-780 called with a later batch of a transaction): m more vectors take the
 * internal ids [n, n + m) (collection.rs:451-468: sequential).  raw_all is the caller's WHOLE table [n + m][dim] (rows [0, n)
 * unchanged; borrowed like coso_index_set_vectors').  The root's code row moves behind the new rows (row_of(root) = n). */
int coso_index_append_vectors(coso_index *ix, const float *raw_all, uint32_t m) {
    if (!ix || !ix->codes || !raw_all || m == 0 || (uint64_t)ix->n + m >= 0xFFFFFFF0ull) return COSO_ERR_INVALID;
    const uint32_t n0 = ix->n, n1 = n0 + m;
    uint8_t *codes = (uint8_t *)calloc((size_t)n1 + 2, ix->cb);
    float *mags = (float *)calloc((size_t)n1 + 2, sizeof(float));
    if (!codes || !mags) { free(codes); free(mags); return COSO_ERR_INVALID; }
    memcpy(codes, ix->codes, (size_t)n0 * ix->cb);
    memcpy(mags, ix->mags, (size_t)n0 * 4);
    memcpy(codes + (size_t)n1 * ix->cb, ix->codes + (size_t)n0 * ix->cb, 2 * ix->cb); /* root + the pseudo nodes' vector */
    memcpy(mags + n1, ix->mags + n0, 8);
    free(ix->codes); free(ix->mags);
    ix->codes = codes; ix->mags = mags; ix->raw = raw_all; ix->n = n1;
    return coso_index_quantize_rows(ix, n0, raw_all + (size_t)n0 * ix->p.dim, m);
}

/* ... and their insertion: the schedule of coso_index_build_rounds CONTINUED at inserted = the vectors already in the graph — level
 * draws from the same RNG stream (the draws a full build would have given these ids), batches of min(batch_size, max(1,
 * inserted / 4)) walking the snapshot that precedes them, ordered claims.  The graph equals a full build's only if the earlier
 * build ended on one of its batch boundaries; it always equals the device's cos_index_append (builder.hip), which runs this. */
/* 1 if the graph has the state coso_index_build_rounds_continue / an append continue from (built by the rounds builder, or restored) */
int coso_index_can_continue(const coso_index *ix) { return ix && ix->codes && ix->rounds_state_valid && ix->n_built == ix->n; }

int coso_index_build_rounds_continue(coso_index *ix, uint32_t batch_size, uint64_t *stats) {
    if (!ix || !ix->codes || !ix->rounds_state_valid || ix->n_built > ix->n) return COSO_ERR_INVALID;
    const uint32_t Ltop = ix->p.num_layers, L1 = Ltop + 1, first = ix->n_built, m = ix->n - first;
    uint64_t st[4] = {0, 0, 0, 0};
    if (m == 0) { if (stats) memcpy(stats, st, sizeof(st)); return COSO_OK; }
    double *pv = (double *)malloc(L1 * sizeof(double));
    uint8_t *pl = (uint8_t *)malloc(L1);
    coso_level_probs(4.0, (int)Ltop, pv, pl);
    uint8_t *max_level = (uint8_t *)malloc(m);
    uint64_t rng = ix->rng_state;
    for (uint32_t i = 0; i < m; i++) max_level[i] = (uint8_t)coso_max_insert_level((double)rand_f32(&rng), pv, pl, (int)L1);
    ix->rounds_state_valid = 0;
    int rc = build_rounds_range(ix, first, batch_size ? batch_size : 4096u, ix->rounds_greedy, max_level, st);
    if (rc == COSO_OK) { ix->rng_state = rng; ix->n_built = ix->n; ix->rounds_state_valid = 1; }
    if (stats) memcpy(stats, st, sizeof(st));
    free(max_level); free(pv); free(pl);
    return rc;
}

/* What a RELOADED index continues from (an imported graph; the device's cos_index_restore_link_state): the reference persists every
 * neighbour slot's similarity (serializer/hnsw/neighbors.rs:22-61) and recomputes a node's cached (lowest index, lowest similarity) when
 * it deserializes it — ProbNode::new_with_neighbors_and_versions (prob_node.rs:145-181): the first empty slot with MetricResult::min,
 * else the first strictly smallest similarity.  Here the similarities are recomputed (the distance is symmetric in its bits: integer
 * dots, one commutative product of norms, one quotient), the caches follow that rule, and the level draws of later appends continue the
 * seed's stream as if the n resident vectors had been drawn from it. */
int coso_index_restore_link_state(coso_index *ix) {
    if (!ix || !ix->codes) return COSO_ERR_INVALID;
    const uint32_t Ltop = ix->p.num_layers;
    const int metric = (int)ix->p.metric;
    for (uint32_t l = 0; l <= Ltop; l++) if (ix->lv[l].n == 0 || ix->lv[l].root_idx == IDX_NONE) return COSO_ERR_INVALID;
    int bad = COSO_OK;
    for (uint32_t l = 0; l <= Ltop; l++) {
        level_t *L = &ix->lv[l];
        const uint32_t M = L->M;
#pragma omp parallel for schedule(dynamic, 256)
        for (int64_t i = 0; i < (int64_t)L->n; i++) {
            const uint32_t row = row_of(ix, L->node_id[i]);
            uint32_t nl = 0;
            float nsim = metric_max(metric);
            int found_empty = 0;
            for (uint32_t j = 0; j < M; j++) {
                const uint32_t x = L->nbr[(size_t)i * M + j];
                if (x == IDX_NONE) { L->nbr_sim[(size_t)i * M + j] = 0.0f; if (!found_empty) { found_empty = 1; nl = j; nsim = metric_min(metric); } continue; }
                float d;
                int rc = node_distance(ix, ix->codes + (size_t)row * ix->cb, ix->mags[row], row_of(ix, L->node_id[x]), &d);
                if (rc != COSO_OK) {
#pragma omp critical
                    bad = rc;
                    d = 0.0f;
                }
                L->nbr_sim[(size_t)i * M + j] = d;
                if (!found_empty && coso_metric_cmp(metric, d, nsim) < 0) { nsim = d; nl = j; }
            }
            L->low_idx[i] = (uint8_t)nl;
            L->low_sim[i] = nsim;
        }
        L->sorted = 0;
    }
    if (bad != COSO_OK) return bad;
    uint64_t rng = ix->p.seed ? ix->p.seed : 0x1234567ull;
    for (uint64_t i = 0; i < (uint64_t)ix->p.dim + ix->n; i++) (void)rand_f32(&rng); /* the root's components, then one level draw per resident vector */
    ix->rng_state = rng;
    ix->n_built = ix->n;
    ix->rounds_state_valid = 1;
    ix->rounds_greedy = 0;
    return COSO_OK;
}

/* delete_embedding (vector_store.rs:1206-1400) for one internal id.  Per level, top down: a walk for the vector's OWN code from the entry
 * node with ef = 512, keep 100, on a filter nothing was pre-inserted into (:1232-1248; nodes_visited is a fresh `&mut 0` per level);
 * descend through the best hit's child (:1250-1259); if the node is among the results (:1261-1275) it is taken out of them
 * (swap_remove), every neighbour drops its back edge (remove_neighbor_by_id :1303: the first slot holding the id; the neighbour's
 * lowest cache is NOT refreshed), and a neighbour left with no neighbour at all is linked again (:1305-1357): the remaining results
 * minus the neighbour itself, re-scored against the neighbour, sorted descending, create_node_edges.  The node itself is dropped
 * (:1366-1369): here its slots are emptied and it stays in the arrays, unreachable (edges are symmetric by construction).  Ties in
 * the re-sort: larger id first, like everywhere.  Returns COSO_OK also when the walk did not reach the node on some level. */
int coso_index_delete(coso_index *ix, uint32_t id) {
    if (!ix || !ix->codes || id == COSO_ROOT_ID || row_of(ix, id) >= ix->n) return COSO_ERR_INVALID;
    const uint32_t Ltop = ix->p.num_layers;
    const int metric = (int)ix->p.metric;
    if (ix->lv[Ltop].root_idx == IDX_NONE) return COSO_ERR_INVALID;
    const uint32_t row = row_of(ix, id);
    const uint8_t *code = ix->codes + (size_t)row * ix->cb;
    scratch_t *s = scratch_new(ix);
    zent *res = (zent *)malloc(KEEP_SEARCH * sizeof(zent)), *rel = (zent *)malloc(KEEP_SEARCH * sizeof(zent));
    hent *srt = (hent *)malloc(KEEP_SEARCH * sizeof(hent));
    uint32_t entry = ix->lv[Ltop].root_idx;
    int rc = COSO_OK;
    for (int level = (int)Ltop; level >= 0 && rc == COSO_OK; level--) {
        level_t *L = &ix->lv[level];
        int cnt = walk_level_ex(ix, (uint32_t)level, entry, code, ix->mags[row], id, 0, 512, KEEP_SEARCH, s, NULL);
        if (cnt < 0) { rc = -cnt; break; }
        if (cnt == 0) { if (level > 0) entry = L->child[entry]; continue; }
        for (int i = 0; i < cnt; i++) { res[i].idx = s->res[i].idx; res[i].sim = s->res[i].sim; }
        if (level > 0) entry = L->child[res[0].idx];
        int at = -1;
        for (int i = 0; i < cnt; i++) if (L->node_id[res[i].idx] == id) { at = i; break; }
        if (at < 0) continue;
        const uint32_t node = res[at].idx;
        res[at] = res[cnt - 1]; /* swap_remove */
        cnt--;
        uint32_t *nb = L->nbr + (size_t)node * L->M;
        for (uint32_t j = 0; j < L->M && rc == COSO_OK; j++) {
            const uint32_t x = nb[j];
            if (x == IDX_NONE) continue;
            uint32_t *xb = L->nbr + (size_t)x * L->M;
            int removed = 0, empty = 1;
            for (uint32_t k = 0; k < L->M; k++) if (xb[k] == node) { xb[k] = IDX_NONE; removed = 1; break; }
            for (uint32_t k = 0; k < L->M; k++) if (xb[k] != IDX_NONE) { empty = 0; break; }
            if (!(removed && empty)) continue;
            /* the orphan is linked again from the walk's results, re-scored against it */
            const uint32_t xrow = row_of(ix, L->node_id[x]);
            int rn = 0;
            for (int i = 0; i < cnt; i++) {
                if (res[i].idx == x) continue;
                float d;
                rc = node_distance(ix, ix->codes + (size_t)xrow * ix->cb, ix->mags[xrow], row_of(ix, L->node_id[res[i].idx]), &d);
                if (rc != COSO_OK) break;
                hent e = {order_key(metric, d), L->node_id[res[i].idx], res[i].idx, d};
                srt[rn++] = e;
            }
            if (rc != COSO_OK) break;
            qsort(srt, (size_t)rn, sizeof(hent), cmp_hent_desc);
            for (int i = 0; i < rn; i++) { rel[i].idx = srt[i].idx; rel[i].sim = srt[i].sim; }
            create_node_edges(ix, (uint32_t)level, x, rel, rn);
        }
        for (uint32_t j = 0; j < L->M; j++) nb[j] = IDX_NONE; /* the node is dropped */
    }
    free(res); free(rel); free(srt);
    scratch_free(s);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * flat export / import: node ids ascending with the root (u32::MAX) last; neighbour slots as
 * internal ids, COSO_SLOT_EMPTY for null pointers.  Slot order is preserved.
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint32_t id, idx; } idpair;
static int cmp_idpair(const void *a, const void *b) {
    uint32_t x = ((const idpair *)a)->id, y = ((const idpair *)b)->id;
    return x < y ? -1 : (x > y ? 1 : 0);
}
int coso_index_export_level(const coso_index *ix, uint32_t level, uint32_t *node_ids, uint32_t *nbr_ids, float *nbr_sims) {
    if (level > ix->p.num_layers) return COSO_ERR_INVALID;
    const level_t *L = &ix->lv[level];
    idpair *ord = (idpair *)malloc((size_t)L->n * sizeof(idpair));
    for (uint32_t i = 0; i < L->n; i++) { ord[i].id = L->node_id[i]; ord[i].idx = i; }
    qsort(ord, L->n, sizeof(idpair), cmp_idpair);
    for (uint32_t k = 0; k < L->n; k++) {
        uint32_t i = ord[k].idx;
        node_ids[k] = L->node_id[i];
        for (uint32_t j = 0; j < L->M; j++) {
            uint32_t nb = L->nbr[(size_t)i * L->M + j];
            nbr_ids[(size_t)k * L->M + j] = nb == IDX_NONE ? COSO_SLOT_EMPTY : L->node_id[nb];
            if (nbr_sims) nbr_sims[(size_t)k * L->M + j] = nb == IDX_NONE ? 0.0f : L->nbr_sim[(size_t)i * L->M + j];
        }
    }
    free(ord);
    return COSO_OK;
}

static uint32_t find_sorted(const uint32_t *ids, uint32_t n, uint32_t id) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (ids[mid] < id) lo = mid + 1; else hi = mid;
    }
    return (lo < n && ids[lo] == id) ? lo : IDX_NONE;
}
static int resolve_children(coso_index *ix, uint32_t level) {
    if (level == 0 || level > ix->p.num_layers) return COSO_OK;
    level_t *L = &ix->lv[level], *D = &ix->lv[level - 1];
    if (!L->n || !D->n || !L->sorted || !D->sorted) return COSO_OK;
    for (uint32_t i = 0; i < L->n; i++) {
        uint32_t c = find_sorted(D->node_id, D->n, L->node_id[i]);
        if (c == IDX_NONE) return COSO_ERR_INVALID; /* every node exists on all lower levels */
        L->child[i] = c;
    }
    return COSO_OK;
}
int coso_index_import_level(coso_index *ix, uint32_t level, uint32_t n_nodes, const uint32_t *node_ids, const uint32_t *nbr_ids) {
    if (ix) ix->rounds_state_valid = 0; /* an imported graph has no similarities / lowest caches to continue from */
    if (level > ix->p.num_layers || n_nodes == 0) return COSO_ERR_INVALID;
    level_t *L = &ix->lv[level];
    level_free(L);
    for (uint32_t i = 0; i < n_nodes; i++) {
        if (i && node_ids[i] <= node_ids[i - 1]) return COSO_ERR_INVALID;
        level_append(L, node_ids[i], (int)ix->p.metric);
    }
    L->sorted = 1;
    if (node_ids[n_nodes - 1] != COSO_ROOT_ID) return COSO_ERR_INVALID;
    L->root_idx = n_nodes - 1;
    int dense = 1; /* level 0 holds ids 0..n-2 then the root: id -> index without searching */
    for (uint32_t i = 0; i + 1 < n_nodes; i++)
        if (node_ids[i] != i) { dense = 0; break; }
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int64_t i = 0; i < (int64_t)n_nodes; i++)
        for (uint32_t j = 0; j < L->M; j++) {
            uint32_t nid = nbr_ids[(size_t)i * L->M + j];
            if (nid == COSO_SLOT_EMPTY) continue;
            uint32_t k = dense ? (nid == COSO_ROOT_ID ? n_nodes - 1 : (nid < n_nodes - 1 ? nid : IDX_NONE)) : find_sorted(L->node_id, L->n, nid);
            if (k == IDX_NONE) { bad = 1; continue; }
            L->nbr[(size_t)i * L->M + j] = k;
        }
    if (bad) return COSO_ERR_INVALID;
    int rc = resolve_children(ix, level);
    if (rc == COSO_OK) rc = resolve_children(ix, level + 1);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * exhaustive search over the QUANTIZED codes ("flat" mode of the device engine): every stored vector is
 * scored with the index metric exactly like one traverse_find_nearest evaluation, the list is sorted like
 * remove_duplicates_and_filter (desc, larger id first), cut to 5k and handed to finalize_ann_results' exact
 * f32 rerank.  It is what the walk would return if it visited every node.
 * ---------------------------------------------------------------------------------------- */
/* the best min(n, 5*top_k) stored vectors of one query by (order key desc, id desc): identical to sorting every
 * (similarity, id) pair like remove_duplicates_and_filter and truncating, without the n-element sort */
static int flat_candidates_one(const coso_index *ix, const float *q, uint32_t top_k, uint8_t *qcode, hent *best, uint32_t *out_m) {
    const int d = (int)ix->p.dim;
    const uint32_t cap = 5 * top_k;
    float qmag;
    int rc = coso_quantize(q, d, (int)ix->p.storage, (int)ix->p.resolution, ix->p.range_lo, ix->p.range_hi, qcode, &qmag);
    uint32_t m = 0;
    for (uint32_t i = 0; i < ix->n && rc == COSO_OK; i++) {
        float sv;
        rc = node_distance(ix, qcode, qmag, i, &sv);
        if (rc != COSO_OK) break;
        hent e = {order_key((int)ix->p.metric, sv), i, i, sv};
        if (m == cap && cmp_hent_desc(&e, &best[cap - 1]) >= 0) continue;
        uint32_t pos = m < cap ? m : cap - 1; /* insertion into a sorted (desc) list */
        while (pos > 0 && cmp_hent_desc(&e, &best[pos - 1]) < 0) { best[pos] = best[pos - 1]; pos--; }
        best[pos] = e;
        if (m < cap) m++;
    }
    *out_m = rc == COSO_OK ? m : 0;
    return rc;
}

/* ids [B][5*top_k] of the candidates whose raw rows the flat search's exact rerank reads (corpora streamed through
 * coso_index_quantize_rows have no raw table on the host: fetch these rows, coso_index_set_raw_subset, then search) */
int coso_flat_candidates_batch(const coso_index *ix, const float *queries, uint32_t B, uint32_t top_k, uint32_t *out_ids,
                               uint32_t *out_counts, int threads) {
    if (!ix || !ix->codes || top_k == 0) return COSO_ERR_INVALID;
    if (threads < 1) threads = 1;
    int first_err = COSO_OK;
#pragma omp parallel num_threads(threads)
    {
        hent *best = (hent *)malloc((size_t)(5 * top_k + 1) * sizeof(hent));
        uint8_t *qcode = (uint8_t *)malloc(ix->cb);
#pragma omp for schedule(dynamic, 1)
        for (int64_t b = 0; b < (int64_t)B; b++) {
            uint32_t m = 0;
            int rc = flat_candidates_one(ix, queries + (size_t)b * ix->p.dim, top_k, qcode, best, &m);
            for (uint32_t i = 0; i < 5 * top_k; i++) out_ids[(size_t)b * 5 * top_k + i] = i < m ? best[i].id : 0xFFFFFFFFu;
            out_counts[b] = m;
            if (rc != COSO_OK) {
#pragma omp critical
                if (first_err == COSO_OK) first_err = rc;
            }
        }
        free(best); free(qcode);
    }
    return first_err;
}

int coso_flat_search_batch(const coso_index *ix, const float *queries, uint32_t B, uint32_t top_k, uint32_t *out_ids,
                           float *out_scores, uint32_t *out_counts, int threads) {
    if (!ix || !ix->codes || (!ix->raw && !ix->sub_rows) || top_k == 0) return COSO_ERR_INVALID;
    if (threads < 1) threads = 1;
    int first_err = COSO_OK;
    const int d = (int)ix->p.dim;
#pragma omp parallel num_threads(threads)
    {
        hent *best = (hent *)malloc((size_t)(5 * top_k + 1) * sizeof(hent));
        fent *f = (fent *)malloc((size_t)(5 * top_k + 1) * sizeof(fent));
        uint8_t *qcode = (uint8_t *)malloc(ix->cb);
#pragma omp for schedule(dynamic, 1)
        for (int64_t b = 0; b < (int64_t)B; b++) {
            const float *q = queries + (size_t)b * d;
            uint32_t m = 0;
            int rc = flat_candidates_one(ix, q, top_k, qcode, best, &m);
            out_counts[b] = 0;
            float mag_query = coso_seq_norm_f32(q, d);
            for (uint32_t i = 0; i < m && rc == COSO_OK; i++) {
                const float *rv = raw_row(ix, best[i].id);
                if (!rv) { rc = COSO_ERR_INVALID; break; }   /* a streamed corpus without this candidate's raw row */
                fent t = {coso_dot_f32(q, rv, d) / (mag_query * coso_seq_norm_f32(rv, d)), best[i].id};
                f[i] = t;
            }
            if (rc != COSO_OK) {
#pragma omp critical
                if (first_err == COSO_OK) first_err = rc;
                continue;
            }
            qsort(f, m, sizeof(fent), cmp_fent_desc);
            if (m > top_k) m = top_k;
            for (uint32_t i = 0; i < m; i++) { out_ids[(size_t)b * top_k + i] = f[i].id; out_scores[(size_t)b * top_k + i] = f[i].cs; }
            out_counts[b] = m;
        }
        free(best); free(f); free(qcode);
    }
    return first_err;
}

/* ------------------------------------------------------------------------------------------
 * exact brute force (ground truth for recall): same formula and arithmetic order as the rerank
 * ---------------------------------------------------------------------------------------- */
int coso_bruteforce_topk(const float *raw, uint32_t n, uint32_t dim, const float *queries, uint32_t B, uint32_t k,
                         uint32_t *out_ids, float *out_scores, int threads) {
    if (k == 0 || k > n) return COSO_ERR_INVALID;
    if (threads < 1) threads = 1;
    float *mags = (float *)malloc((size_t)n * sizeof(float));
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t i = 0; i < (int64_t)n; i++) mags[i] = coso_seq_norm_f32(raw + (size_t)i * dim, (int)dim);
#pragma omp parallel num_threads(threads)
    {
        fent *top = (fent *)malloc((size_t)(k + 1) * sizeof(fent));
#pragma omp for schedule(dynamic, 1)
        for (int64_t b = 0; b < (int64_t)B; b++) {
            const float *q = queries + (size_t)b * dim;
            float mq = coso_seq_norm_f32(q, (int)dim);
            uint32_t cnt = 0;
            for (uint32_t i = 0; i < n; i++) {
                float dp = coso_dot_f32(q, raw + (size_t)i * dim, (int)dim);
                fent e = {dp / (mq * mags[i]), i};
                if (cnt == k && cmp_fent_desc(&e, &top[k - 1]) >= 0) continue;
                uint32_t pos = cnt < k ? cnt : k - 1; /* insertion into a sorted (desc) list */
                while (pos > 0 && cmp_fent_desc(&e, &top[pos - 1]) < 0) { top[pos] = top[pos - 1]; pos--; }
                top[pos] = e;
                if (cnt < k) cnt++;
            }
            for (uint32_t i = 0; i < k; i++) { out_ids[(size_t)b * k + i] = top[i].id; out_scores[(size_t)b * k + i] = top[i].cs; }
        }
        free(top);
    }
    free(mags);
    return COSO_OK;
}

/* ==========================================================================================
 * Metadata-filtered search — CPU restatement.
 *   ReplicaNodeKind / VectorData::replica_node_kind     models/types.rs:171-251
 *   Metadata (mag, mbits)                               models/types.rs:106-147
 *   CosineSimilarity::calculate, (node kind, query kind) arms      distance/cosine.rs:36-102
 *   cosine_similarity_mdims                             distance/cosine.rs:243-262
 *   ann_search with query_filter_dims                   vector_store.rs:256-402 (one walk per QueryFilterDimensions, ONE
 *                                                       visited filter per level, drop -1.0, sort desc, take 100)
 *   search_internal: filtered queries start at the pseudo root     indexes/hnsw/mod.rs:413-423
 *   remove_duplicates_and_filter drops pseudo nodes     models/common.rs:381-412
 *   raw embedding of a replica = base id = id - id % max_replica_per_node     models/collection.rs:368-384
 *   index side: replicas / pseudo nodes, edge refusal   vector_store.rs:484-640, 1017-1041; metadata/mod.rs:182-217
 * What stays on the host in the reference and here: schema -> dimensions (metadata/schema.rs, query_filtering.rs).  This
 * component works on the numeric form: per node a replica id and its metadata dimensions, per query a list of filter dimension
 * vectors in {-1, 0, 1}.  Vector rows: replica id / replicas for embeddings (ids are reserved `replicas` at a time,
 * collection.rs:445-468), row n + 1 for every pseudo node (they share the pseudo root's all-zero vector, api_service.rs:143-158).
 * ========================================================================================== */
#define PSEUDO_LO 0xFFFFFEFEu /* u32::MAX - 257: pseudo_root_id() (metadata/mod.rs:217-224) */
#define PSEUDO_HI 0xFFFFFFFDu /* u32::MAX - 2 */
enum { KIND_BASE = 0, KIND_PSEUDO = 1, KIND_METADATA = 2 };

static int meta_index(const coso_index *ix, uint32_t id) {
    uint32_t lo = 0, hi = ix->n_meta;
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (ix->meta_id[mid] < id) lo = mid + 1; else hi = mid;
    }
    return (lo < ix->n_meta && ix->meta_id[lo] == id) ? (int)lo : -1;
}
static inline uint32_t meta_row_of(const coso_index *ix, uint32_t id) {
    return (id >= PSEUDO_LO && id <= PSEUDO_HI) ? ix->n + 1 : id / ix->replicas;
}
/* VectorData::replica_node_kind (types.rs:219-241): `has_id` = the VectorData carries an id (stored nodes, new nodes being
 * indexed); a search query has none */
static int kind_of(float mmag, int has_id, uint32_t id) {
    if (mmag == 0.0f) return KIND_BASE;
    if (has_id && id >= PSEUDO_LO && id <= PSEUDO_HI) return KIND_PSEUDO;
    return KIND_METADATA;
}

int coso_meta_enable(coso_index *ix, uint32_t mdim, uint32_t max_replicas) {
    if (!ix || !ix->codes || mdim == 0 || max_replicas == 0) return COSO_ERR_INVALID;
    ix->mdim = mdim;
    ix->replicas = max_replicas;
    float *z = (float *)calloc(ix->p.dim, sizeof(float)); /* pseudo_node_vector: all zeros (metadata/mod.rs:211-214) */
    int rc = coso_quantize(z, (int)ix->p.dim, (int)ix->p.storage, (int)ix->p.resolution, ix->p.range_lo, ix->p.range_hi,
                           ix->codes + ((size_t)ix->n + 1) * ix->cb, &ix->mags[ix->n + 1]);
    free(z);
    if (!ix->mlv) {
        ix->mlv = (level_t *)calloc(ix->p.num_layers + 1, sizeof(level_t));
        for (uint32_t l = 0; l <= ix->p.num_layers; l++) { ix->mlv[l].M = level_M(ix, l); ix->mlv[l].root_idx = IDX_NONE; }
    }
    return rc;
}

/* node table of the component: ascending replica ids (embedding replicas, then the pseudo root u32::MAX - 257 and the pseudo
 * nodes after it), mbits [n][mdim].  mag = sqrt(sum of squares) like Metadata::from (types.rs:112-126). */
int coso_meta_set_nodes(coso_index *ix, uint32_t n_nodes, const uint32_t *ids_sorted, const int32_t *mbits) {
    if (!ix || !ix->mdim || !ids_sorted || !mbits) return COSO_ERR_INVALID;
    for (uint32_t i = 1; i < n_nodes; i++)
        if (ids_sorted[i] <= ids_sorted[i - 1]) return COSO_ERR_INVALID;
    free(ix->meta_id); free(ix->meta_mbits); free(ix->meta_mdims); free(ix->meta_mag);
    const uint32_t md = ix->mdim;
    ix->n_meta = n_nodes;
    ix->meta_id = (uint32_t *)malloc((size_t)n_nodes * 4 + 4);
    ix->meta_mbits = (int32_t *)malloc((size_t)n_nodes * md * 4 + 4);
    ix->meta_mdims = (float *)malloc((size_t)n_nodes * md * 4 + 4);
    ix->meta_mag = (float *)malloc((size_t)n_nodes * 4 + 4);
    memcpy(ix->meta_id, ids_sorted, (size_t)n_nodes * 4);
    memcpy(ix->meta_mbits, mbits, (size_t)n_nodes * md * 4);
    for (uint32_t i = 0; i < n_nodes; i++) {
        if (!(ids_sorted[i] >= PSEUDO_LO && ids_sorted[i] <= PSEUDO_HI) && ids_sorted[i] / ix->replicas >= ix->n) return COSO_ERR_INVALID;
        for (uint32_t j = 0; j < md; j++) ix->meta_mdims[(size_t)i * md + j] = (float)mbits[(size_t)i * md + j];
        ix->meta_mag[i] = coso_seq_norm_f32(ix->meta_mdims + (size_t)i * md, (int)md);
    }
    return COSO_OK;
}

/* the query side of a distance: the quantized vector + (optional) metadata dimensions */
typedef struct {
    const uint8_t *code;
    float mag;
    const int32_t *mbits; /* NULL = no metadata */
    const float *mdims;
    float mmag;
    int kind;             /* KIND_* of the query / new node */
} qdesc;

/* CosineSimilarity::calculate / DotProductDistance on (stored node y, query x) — distance/cosine.rs:36-102 */
static int meta_distance(const coso_index *ix, const qdesc *q, uint32_t node_id, float *out) {
    const int k = meta_index(ix, node_id);
    if (k < 0) return COSO_ERR_INVALID;
    const uint32_t md = ix->mdim;
    const int ykind = kind_of(ix->meta_mag[k], 1, node_id), xkind = q->kind;
    const uint32_t row = meta_row_of(ix, node_id);
    if (ix->p.metric != COSO_METRIC_COSINE) /* other metrics do not look at the node kinds at all */
        return coso_distance((int)ix->p.metric, (int)ix->p.storage, (int)ix->p.resolution, (int)ix->p.dim, q->code, q->mag,
                             ix->codes + (size_t)row * ix->cb, ix->mags[row], out);
    if (ykind == KIND_PSEUDO && xkind == KIND_PSEUDO) { /* cosine_similarity_mdims */
        float dp = coso_dot_f32(q->mdims, ix->meta_mdims + (size_t)k * md, (int)md);
        float den = q->mmag * ix->meta_mag[k];
        if (den == 0.0f) return COSO_ERR_CALCULATION;
        *out = dp / den;
        return COSO_OK;
    }
    if (ykind == KIND_PSEUDO && xkind == KIND_METADATA) { /* exact mbits match -> 1.0, anything else -> -1.0 */
        *out = memcmp(q->mbits, ix->meta_mbits + (size_t)k * md, (size_t)md * 4) == 0 ? 1.0f : -1.0f;
        return COSO_OK;
    }
    if (ykind == KIND_BASE && xkind == KIND_BASE)
        return coso_distance(COSO_METRIC_COSINE, (int)ix->p.storage, (int)ix->p.resolution, (int)ix->p.dim, q->code, q->mag,
                             ix->codes + (size_t)row * ix->cb, ix->mags[row], out);
    if (ykind == KIND_METADATA && xkind == KIND_METADATA) {
        float dp = coso_dot_f32(q->mdims, ix->meta_mdims + (size_t)k * md, (int)md);
        float den = q->mmag * ix->meta_mag[k];
        if (den == 0.0f) return COSO_ERR_CALCULATION;
        if (dp / den > 0.99f)
            return coso_distance(COSO_METRIC_COSINE, (int)ix->p.storage, (int)ix->p.resolution, (int)ix->p.dim, q->code, q->mag,
                                 ix->codes + (size_t)row * ix->cb, ix->mags[row], out);
        *out = -1.0f;
        return COSO_OK;
    }
    if (ykind == KIND_BASE && xkind == KIND_METADATA) { *out = 0.0f; return COSO_OK; }
    return COSO_ERR_UNIMPLEMENTED; /* the reference's `unreachable!()` arms */
}

/* traverse_find_nearest on one level of the component.  The caller owns the visited filter (it is shared by the walks of
 * every filter of a query on that level, vector_store.rs:266-313) */
static int walk_level_meta(const coso_index *ix, uint32_t level, uint32_t entry_idx, const qdesc *q, uint32_t ef, uint32_t keep, scratch_t *s) {
    const level_t *L = &ix->mlv[level];
    const uint32_t M = L->M;
    const int metric = (int)ix->p.metric;
    uint32_t slots = M < ix->p.shortlist_size ? M : ix->p.shortlist_size;
    size_t hn = 0, rn = 0;
    float d0;
    int rc = meta_distance(ix, q, L->node_id[entry_idx], &d0);
    if (rc != COSO_OK) return -rc;
    s->visited[(L->node_id[entry_idx] >> 6) & (M - 1)] |= 1ull << (L->node_id[entry_idx] & 63);
    hent e0 = {order_key(metric, d0), L->node_id[entry_idx], entry_idx, d0};
    heap_push(s, &hn, e0);
    uint32_t nodes_visited = 0;
    while (hn > 0) {
        hent cur = heap_pop(s, &hn);
        if (nodes_visited >= ef) break;
        nodes_visited++;
        if (rn == s->res_cap) { s->res_cap *= 2; s->res = (hent *)realloc(s->res, s->res_cap * sizeof(hent)); }
        s->res[rn++] = cur;
        const uint32_t *nb = L->nbr + (size_t)cur.idx * M;
        for (uint32_t j = 0; j < slots; j++) {
            uint32_t nidx = nb[j];
            if (nidx == IDX_NONE) continue;
            uint32_t nid = L->node_id[nidx];
            if ((s->visited[(nid >> 6) & (M - 1)] >> (nid & 63)) & 1ull) continue;
            float d;
            rc = meta_distance(ix, q, nid, &d);
            if (rc != COSO_OK) return -rc;
            s->visited[(nid >> 6) & (M - 1)] |= 1ull << (nid & 63);
            hent e = {order_key(metric, d), nid, nidx, d};
            heap_push(s, &hn, e);
        }
    }
    qsort(s->res, rn, sizeof(hent), cmp_hent_desc);
    if (rn > keep) rn = keep;
    return (int)rn;
}

/* ann_search with query_filter_dims (vector_store.rs:273-313): filters [nf][mdim] in {-1,0,1}.  out: concatenated per-level
 * lists (<= 100 each), top level first. */
static int ann_search_filtered(const coso_index *ix, const uint8_t *qcode, float qmag, const int32_t *filters, uint32_t nf, scratch_t *s, hent *out,
                               uint32_t *level_counts) {
    const uint32_t Ltop = ix->p.num_layers, md = ix->mdim;
    if (!ix->mlv || ix->mlv[Ltop].root_idx == IDX_NONE) return -COSO_ERR_INVALID;
    uint32_t entry = ix->mlv[Ltop].root_idx;
    float *fd = (float *)malloc((size_t)(nf ? nf : 1) * md * 4);
    qdesc *qd = (qdesc *)malloc((size_t)(nf ? nf : 1) * sizeof(qdesc));
    for (uint32_t f = 0; f < nf; f++) {
        for (uint32_t j = 0; j < md; j++) fd[(size_t)f * md + j] = (float)filters[(size_t)f * md + j];
        qdesc d = {qcode, qmag, filters + (size_t)f * md, fd + (size_t)f * md, coso_seq_norm_f32(fd + (size_t)f * md, (int)md), 0};
        d.kind = kind_of(d.mmag, 0, 0); /* a query has no id (types.rs:227-236) */
        qd[f] = d;
    }
    hent *cand = (hent *)malloc((size_t)(nf ? nf : 1) * KEEP_SEARCH * sizeof(hent));
    int total = 0, rc = 0;
    for (int level = (int)Ltop; level >= 0 && rc == 0; level--) {
        const level_t *L = &ix->mlv[level];
        memset(s->visited, 0, (size_t)L->M * 8);
        s->visited[(COSO_QUERY_ID >> 6) & (L->M - 1)] |= 1ull << (COSO_QUERY_ID & 63);
        int nc = 0;
        for (uint32_t f = 0; f < nf; f++) {
            int cnt = walk_level_meta(ix, (uint32_t)level, entry, &qd[f], ix->p.ef_search, KEEP_SEARCH, s);
            if (cnt < 0) { rc = cnt; break; }
            for (int i = 0; i < cnt; i++) {
                if (ix->p.metric == COSO_METRIC_COSINE && s->res[i].sim == -1.0f) continue; /* strong mismatch: dropped */
                cand[nc++] = s->res[i];
            }
        }
        if (rc) break;
        qsort(cand, (size_t)nc, sizeof(hent), cmp_hent_desc);
        if (nc > KEEP_SEARCH) nc = KEEP_SEARCH;
        if (nc == 0) { /* vector_store.rs:329-380: the entry node with its strongest match over the filters */
            int have = 0;
            hent best = {0, 0, 0, 0.0f};
            for (uint32_t f = 0; f < nf; f++) {
                float d;
                int r = meta_distance(ix, &qd[f], L->node_id[entry], &d);
                if (r != COSO_OK) { rc = -r; break; }
                hent e = {order_key((int)ix->p.metric, d), L->node_id[entry], entry, d};
                if (!have || e.key > best.key) { best = e; have = 1; }
            }
            if (rc) break;
            if (!have) { rc = -COSO_ERR_INVALID; break; }
            cand[0] = best;
            nc = 1;
        }
        memcpy(out + total, cand, (size_t)nc * sizeof(hent));
        if (level_counts) level_counts[Ltop - (uint32_t)level] = (uint32_t)nc;
        total += nc;
        if (level > 0) entry = L->child[cand[0].idx];
    }
    free(cand); free(qd); free(fd);
    return rc ? rc : total;
}

/* search_internal with a filter (indexes/hnsw/mod.rs:390-440) for B queries; filters of query b = filter_dims rows
 * [filter_off[b], filter_off[b+1]) */
int coso_search_filtered_batch(const coso_index *ix, const float *queries, uint32_t B, const uint32_t *filter_off, const int32_t *filter_dims,
                               uint32_t top_k, uint32_t *out_ids, float *out_scores, uint32_t *out_counts, int32_t *out_status, int threads) {
    if (!ix || !ix->raw || !ix->mdim || top_k == 0 || ix->p.visited_mode != COSO_VISITED_REF) return COSO_ERR_INVALID;
    int first_err = COSO_OK;
    if (threads < 1) threads = 1;
    const int d = (int)ix->p.dim;
#pragma omp parallel num_threads(threads)
    {
        scratch_t *s = scratch_new(ix);
        hent *all = (hent *)malloc((size_t)(ix->p.num_layers + 1) * KEEP_SEARCH * sizeof(hent));
        fent *f = (fent *)malloc((size_t)(ix->p.num_layers + 1) * KEEP_SEARCH * sizeof(fent));
#pragma omp for schedule(dynamic, 1)
        for (int64_t b = 0; b < (int64_t)B; b++) {
            const float *q = queries + (size_t)b * d;
            float qmag;
            int rc = coso_quantize(q, d, (int)ix->p.storage, (int)ix->p.resolution, ix->p.range_lo, ix->p.range_hi, s->qcode, &qmag);
            int total = 0, m = 0;
            if (rc == COSO_OK) {
                total = ann_search_filtered(ix, s->qcode, qmag, filter_dims + (size_t)filter_off[b] * ix->mdim, filter_off[b + 1] - filter_off[b], s, all, NULL);
                if (total < 0) { rc = -total; total = 0; }
            }
            if (rc == COSO_OK) {
                /* remove_duplicates_and_filter: first seen, drop the root and every pseudo node, sort desc, keep 5k */
                for (int i = 0; i < total; i++) {
                    int dup = 0;
                    for (int j = 0; j < m; j++)
                        if (all[j].id == all[i].id) { dup = 1; break; }
                    if (dup) continue;
                    all[m++] = all[i];
                }
                int w = 0;
                for (int i = 0; i < m; i++) {
                    const int k = meta_index(ix, all[i].id);
                    const int pseudo = k >= 0 && kind_of(ix->meta_mag[k], 1, all[i].id) == KIND_PSEUDO;
                    if (all[i].id != COSO_ROOT_ID && !pseudo) all[w++] = all[i];
                }
                m = w;
                qsort(all, (size_t)m, sizeof(hent), cmp_hent_desc);
                if ((uint32_t)m > 5 * top_k) m = (int)(5 * top_k);
                float mag_query = coso_seq_norm_f32(q, d);
                for (int i = 0; i < m; i++) { /* raw embedding of a replica = its base id (collection.rs:368-384) */
                    const float *rv = ix->raw + (size_t)(all[i].id / ix->replicas) * d;
                    float dp = coso_dot_f32(q, rv, d);
                    fent t = {dp / (mag_query * coso_seq_norm_f32(rv, d)), all[i].id};
                    f[i] = t;
                }
                qsort(f, (size_t)m, sizeof(fent), cmp_fent_desc);
                if ((uint32_t)m > top_k) m = (int)top_k;
                for (int i = 0; i < m; i++) { out_ids[(size_t)b * top_k + i] = f[i].id; out_scores[(size_t)b * top_k + i] = f[i].cs; }
            }
            out_counts[b] = rc == COSO_OK ? (uint32_t)m : 0;
            if (out_status) out_status[b] = rc;
            if (rc != COSO_OK) {
#pragma omp critical
                if (first_err == COSO_OK) first_err = rc;
            }
        }
        free(all); free(f);
        scratch_free(s);
    }
    return first_err;
}

/* raw per-level lists of one filtered query (ids, sims, counts top level first) — what the device walk is compared with */
int coso_ann_search_filtered(const coso_index *ix, const float *query, const int32_t *filter_dims, uint32_t nf, uint32_t *out_ids, float *out_sims,
                             uint32_t *level_counts) {
    if (!ix || !ix->mdim) return -COSO_ERR_INVALID;
    scratch_t *s = scratch_new(ix);
    float qmag;
    int rc = coso_quantize(query, (int)ix->p.dim, (int)ix->p.storage, (int)ix->p.resolution, ix->p.range_lo, ix->p.range_hi, s->qcode, &qmag);
    if (rc != COSO_OK) { scratch_free(s); return -rc; }
    hent *all = (hent *)malloc((size_t)(ix->p.num_layers + 1) * KEEP_SEARCH * sizeof(hent));
    int total = ann_search_filtered(ix, s->qcode, qmag, filter_dims, nf, s, all, level_counts);
    for (int i = 0; i < total; i++) { out_ids[i] = all[i].id; out_sims[i] = all[i].sim; }
    free(all);
    scratch_free(s);
    return total;
}

/* ---- builder of the component (index side; vector_store.rs:714-1074 with prop_metadata) ---------------------------
 * Insertion order = ascending position in `order` (node-table indices): the reference inserts the pseudo nodes when the index
 * is created (api_service.rs:190-210) and the replicas of an embedding as it arrives.  max_levels[i] = the level drawn for
 * node-table entry i (the host's get_max_insert_level over levels_prob / pseudo_level_probs); the pseudo root is on every level. */
static uint32_t mlevel_find(const level_t *L, uint32_t id) {
    for (uint32_t i = 0; i < L->n; i++) if (L->node_id[i] == id) return i;
    return IDX_NONE;
}

static void create_node_edges_meta(coso_index *ix, uint32_t level, uint32_t node, const zent *z, int zn) {
    level_t *L = &ix->mlv[level];
    const int metric = (int)ix->p.metric;
    const int kself = meta_index(ix, L->node_id[node]);
    const int self_kind = kind_of(ix->meta_mag[kself], 1, L->node_id[node]);
    uint32_t succ = 0;
    for (int i = 0; i < zn; i++) {
        if (succ >= L->M) break;
        const uint32_t nid = L->node_id[z[i].idx];
        const int kn = meta_index(ix, nid);
        const int nkind = kind_of(ix->meta_mag[kn], 1, nid);
        if (metric == COSO_METRIC_COSINE) { /* vector_store.rs:1017-1041 */
            if (nkind == KIND_PSEUDO && self_kind == KIND_METADATA && z[i].sim != 1.0f) continue;
            if (nkind == KIND_METADATA && self_kind == KIND_METADATA && z[i].sim == -1.0f) continue;
        }
        int r = add_neighbor(L, metric, node, z[i].idx, z[i].sim);
        if (r >= 0) {
            int r2 = add_neighbor(L, metric, z[i].idx, node, z[i].sim);
            if (r2 >= 0) succ++;
            else if (L->nbr[(size_t)node * L->M + (uint32_t)r] == z[i].idx) L->nbr[(size_t)node * L->M + (uint32_t)r] = IDX_NONE;
        }
    }
}

static int index_embedding_meta(coso_index *ix, uint32_t id, const qdesc *q, uint32_t parent_idx, uint32_t entry_idx, int level, int max_level,
                                scratch_t *s) {
    level_t *L = &ix->mlv[level];
    memset(s->visited, 0, (size_t)L->M * 8);
    s->visited[(id >> 6) & (L->M - 1)] |= 1ull << (id & 63); /* skipm.insert(new_node_id) :807 */
    int zn = walk_level_meta(ix, (uint32_t)level, entry_idx, q, ix->p.ef_construction, KEEP_INDEX, s);
    if (zn < 0) return -zn;
    zent z[KEEP_INDEX];
    if (zn == 0) { /* only when ef_construction == 0; the fallback's VectorData has id: None (vector_store.rs:833-838) */
        float d;
        qdesc q0 = *q;
        q0.kind = kind_of(q->mmag, 0, 0);
        int rc = meta_distance(ix, &q0, L->node_id[entry_idx], &d);
        if (rc != COSO_OK) return rc;
        z[0].idx = entry_idx; z[0].sim = d; zn = 1;
    } else
        for (int i = 0; i < zn; i++) { z[i].idx = s->res[i].idx; z[i].sim = s->res[i].sim; }
    uint32_t child = level > 0 ? L->child[z[0].idx] : IDX_NONE;
    if (level > max_level) {
        if (level != 0) return index_embedding_meta(ix, id, q, IDX_NONE, child, level - 1, max_level, s);
        return COSO_OK;
    }
    uint32_t me = level_append(L, id, (int)ix->p.metric);
    if (parent_idx != IDX_NONE) ix->mlv[level + 1].child[parent_idx] = me;
    if (level != 0) {
        int rc = index_embedding_meta(ix, id, q, me, child, level - 1, max_level, s);
        if (rc != COSO_OK) return rc;
    }
    create_node_edges_meta(ix, (uint32_t)level, me, z, zn);
    return COSO_OK;
}

int coso_meta_build(coso_index *ix, const uint8_t *max_levels /*[n_meta], node-table order*/) {
    if (!ix || !ix->mdim || !ix->n_meta || !max_levels || ix->p.visited_mode != COSO_VISITED_REF) return COSO_ERR_INVALID;
    const uint32_t Ltop = ix->p.num_layers, md = ix->mdim;
    const int kroot = meta_index(ix, PSEUDO_LO);
    if (kroot < 0) return COSO_ERR_INVALID; /* the node table must hold the pseudo root */
    for (uint32_t l = 0; l <= Ltop; l++) level_free(&ix->mlv[l]);
    for (uint32_t l = 0; l <= Ltop; l++) { /* create_pseudo_root_node: one node per level, linked parent/child */
        uint32_t r = level_append(&ix->mlv[l], PSEUDO_LO, (int)ix->p.metric);
        ix->mlv[l].root_idx = r;
        if (l > 0) ix->mlv[l].child[r] = ix->mlv[l - 1].root_idx;
    }
    scratch_t *s = scratch_new(ix);
    int rc = COSO_OK;
    /* pseudo nodes first (ids after the pseudo root, ascending), then the replicas in id order */
    for (int pass = 0; pass < 2 && rc == COSO_OK; pass++)
        for (uint32_t i = 0; i < ix->n_meta && rc == COSO_OK; i++) {
            const uint32_t id = ix->meta_id[i];
            const int pseudo = id >= PSEUDO_LO && id <= PSEUDO_HI;
            if (id == PSEUDO_LO || pseudo != (pass == 0)) continue;
            const uint32_t row = meta_row_of(ix, id);
            /* x side of the distance while indexing: VectorData{id: Some(prop_value.id), ..} — the embedding's BASE id, or the
             * pseudo root's id for pseudo nodes (vector_store.rs:820-832, 621-640) */
            const uint32_t xid = pseudo ? PSEUDO_LO : id - id % ix->replicas;
            qdesc q = {ix->codes + (size_t)row * ix->cb, ix->mags[row], ix->meta_mbits + (size_t)i * md, ix->meta_mdims + (size_t)i * md,
                       ix->meta_mag[i], kind_of(ix->meta_mag[i], 1, xid)};
            if (q.kind == KIND_BASE) continue; /* base replicas live under the MAIN root (types.rs:196-204): not in this component */
            rc = index_embedding_meta(ix, id, &q, IDX_NONE, ix->mlv[Ltop].root_idx, (int)Ltop, (int)max_levels[i], s);
        }
    scratch_free(s);
    (void)mlevel_find;
    return rc;
}

/* The same component built BATCH-SYNCHRONOUSLY — the schedule a device-side builder would run (cf. coso_index_build_rounds for the
 * base graph): the nodes of a batch (size min(batch_size, max(1, inserted / 4)), pseudo nodes first, then the replicas in id
 * order) walk the snapshot that precedes the batch on every level, then their nodes are appended and the levels are linked in
 * rounds with ordered claims ({self} + candidates; the first claimer of a row in batch order owns it, blocked nodes wait), the
 * edge-refusal rules of vector_store.rs:1017-1041 applied exactly where create_node_edges_meta applies them, evictions outside
 * the own claim applied at the end of the round.  batch_size = 1 is coso_meta_build.  stats (optional): rounds, level-batches,
 * node-levels linked. */
int coso_meta_build_rounds(coso_index *ix, const uint8_t *max_levels, uint32_t batch_size, uint64_t *stats) {
    if (!ix || !ix->mdim || !ix->n_meta || !max_levels || ix->p.visited_mode != COSO_VISITED_REF) return COSO_ERR_INVALID;
    const uint32_t Ltop = ix->p.num_layers, L1 = Ltop + 1, md = ix->mdim;
    const uint32_t Bmax = batch_size ? batch_size : 4096u;
    const int metric = (int)ix->p.metric;
    if (meta_index(ix, PSEUDO_LO) < 0) return COSO_ERR_INVALID;
    uint64_t st[3] = {0, 0, 0};
    for (uint32_t l = 0; l <= Ltop; l++) level_free(&ix->mlv[l]);
    for (uint32_t l = 0; l <= Ltop; l++) { /* create_pseudo_root_node */
        uint32_t r = level_append(&ix->mlv[l], PSEUDO_LO, metric);
        ix->mlv[l].root_idx = r;
        if (l > 0) ix->mlv[l].child[r] = ix->mlv[l - 1].root_idx;
    }
    /* insertion order: pseudo nodes (ascending id), then the Metadata replicas (ascending id); Base replicas are not in this component */
    uint32_t *ord = (uint32_t *)malloc((size_t)ix->n_meta * 4), total = 0;
    for (int pass = 0; pass < 2; pass++)
        for (uint32_t i = 0; i < ix->n_meta; i++) {
            const uint32_t id = ix->meta_id[i];
            const int pseudo = id >= PSEUDO_LO && id <= PSEUDO_HI;
            if (id == PSEUDO_LO || pseudo != (pass == 0)) continue;
            const uint32_t xid = pseudo ? PSEUDO_LO : id - id % ix->replicas;
            if (kind_of(ix->meta_mag[i], 1, xid) == KIND_BASE) continue;
            ord[total++] = i;
        }
    scratch_t *s = scratch_new(ix);
    zent *z = (zent *)malloc((size_t)Bmax * L1 * KEEP_INDEX * sizeof(zent));
    uint32_t *zn = (uint32_t *)malloc((size_t)Bmax * L1 * 4), *me = (uint32_t *)malloc((size_t)Bmax * L1 * 4);
    uint32_t *claim_round = (uint32_t *)calloc((size_t)ix->n_meta + 2, 4), *claim_owner = (uint32_t *)calloc((size_t)ix->n_meta + 2, 4);
    uint32_t *pending = (uint32_t *)malloc((size_t)Bmax * 4), *next = (uint32_t *)malloc((size_t)Bmax * 4);
    evict_t *q = (evict_t *)malloc((size_t)Bmax * 2 * KEEP_INDEX * sizeof(evict_t));
    uint32_t round_id = 0, inserted = 0;
    int rc = COSO_OK;
    while (inserted < total && rc == COSO_OK) {
        uint32_t bs = inserted / 4u;
        if (bs < 1) bs = 1;
        if (bs > Bmax) bs = Bmax;
        if (bs > total - inserted) bs = total - inserted;
        for (uint32_t b = 0; b < bs && rc == COSO_OK; b++) { /* 1. walks on the snapshot, every level, like index_embedding_meta */
            const uint32_t i = ord[inserted + b], id = ix->meta_id[i], row = meta_row_of(ix, id);
            const int pseudo = id >= PSEUDO_LO && id <= PSEUDO_HI;
            const uint32_t xid = pseudo ? PSEUDO_LO : id - id % ix->replicas;
            qdesc qd = {ix->codes + (size_t)row * ix->cb, ix->mags[row], ix->meta_mbits + (size_t)i * md, ix->meta_mdims + (size_t)i * md,
                        ix->meta_mag[i], kind_of(ix->meta_mag[i], 1, xid)};
            uint32_t entry = ix->mlv[Ltop].root_idx;
            for (int level = (int)Ltop; level >= 0; level--) {
                level_t *L = &ix->mlv[level];
                zent *zz = z + ((size_t)b * L1 + (uint32_t)level) * KEEP_INDEX;
                memset(s->visited, 0, (size_t)L->M * 8);
                s->visited[(id >> 6) & (L->M - 1)] |= 1ull << (id & 63);
                int cnt = walk_level_meta(ix, (uint32_t)level, entry, &qd, ix->p.ef_construction, KEEP_INDEX, s);
                if (cnt < 0) { rc = -cnt; break; }
                if (cnt == 0) {
                    float d;
                    qdesc q0 = qd;
                    q0.kind = kind_of(qd.mmag, 0, 0);
                    rc = meta_distance(ix, &q0, L->node_id[entry], &d);
                    if (rc != COSO_OK) break;
                    zz[0].idx = entry; zz[0].sim = d; cnt = 1;
                } else
                    for (int k = 0; k < cnt; k++) { zz[k].idx = s->res[k].idx; zz[k].sim = s->res[k].sim; }
                zn[(size_t)b * L1 + (uint32_t)level] = (uint32_t)cnt;
                if (level > 0) entry = L->child[zz[0].idx];
            }
        }
        if (rc != COSO_OK) break;
        for (uint32_t b = 0; b < bs; b++) { /* 2. nodes of the batch */
            const uint32_t i = ord[inserted + b], id = ix->meta_id[i];
            uint32_t parent = IDX_NONE;
            for (int level = (int)max_levels[i]; level >= 0; level--) {
                uint32_t m = level_append(&ix->mlv[level], id, metric);
                me[(size_t)b * L1 + (uint32_t)level] = m;
                if (parent != IDX_NONE) ix->mlv[level + 1].child[parent] = m;
                parent = m;
            }
        }
        for (uint32_t l = 0; l <= Ltop; l++) { /* 3. rounds */
            level_t *L = &ix->mlv[l];
            uint32_t np = 0;
            for (uint32_t b = 0; b < bs; b++) if (max_levels[ord[inserted + b]] >= l) pending[np++] = b;
            if (np == 0) continue;
            st[1]++;
            st[2] += np;
            while (np > 0) {
                round_id++;
                st[0]++;
                uint32_t nn = 0, qn = 0;
                for (uint32_t k = 0; k < np; k++) {
                    const uint32_t b = pending[k], node = me[(size_t)b * L1 + l];
                    const zent *zz = z + ((size_t)b * L1 + l) * KEEP_INDEX;
                    const int cnt = (int)zn[(size_t)b * L1 + l];
                    int runnable = 1;
                    for (int c = -1; c < cnt; c++) { /* ordered claims */
                        const uint32_t r = c < 0 ? node : zz[c].idx;
                        if (claim_round[r] == round_id) { if (claim_owner[r] != b) runnable = 0; }
                        else { claim_round[r] = round_id; claim_owner[r] = b; }
                    }
                    if (!runnable) { next[nn++] = b; continue; }
                    const int kself = meta_index(ix, L->node_id[node]);
                    const int self_kind = kind_of(ix->meta_mag[kself], 1, L->node_id[node]);
                    uint32_t succ = 0;
                    for (int c = 0; c < cnt; c++) { /* create_node_edges_meta with deferred evictions */
                        if (succ >= L->M) break;
                        const uint32_t nid = L->node_id[zz[c].idx];
                        const int nkind = kind_of(ix->meta_mag[meta_index(ix, nid)], 1, nid);
                        if (metric == COSO_METRIC_COSINE) { /* vector_store.rs:1017-1041 */
                            if (nkind == KIND_PSEUDO && self_kind == KIND_METADATA && zz[c].sim != 1.0f) continue;
                            if (nkind == KIND_METADATA && self_kind == KIND_METADATA && zz[c].sim == -1.0f) continue;
                        }
                        uint32_t ev;
                        int r = add_neighbor_deferred(L, metric, node, zz[c].idx, zz[c].sim, &ev);
                        if (ev != IDX_NONE) {
                            if (claim_round[ev] == round_id && claim_owner[ev] == b) remove_neighbor_by_idx(L, ev, node);
                            else { q[qn].old_idx = ev; q[qn].target = node; qn++; }
                        }
                        if (r >= 0) {
                            int r2 = add_neighbor_deferred(L, metric, zz[c].idx, node, zz[c].sim, &ev);
                            if (ev != IDX_NONE) {
                                if (claim_round[ev] == round_id && claim_owner[ev] == b) remove_neighbor_by_idx(L, ev, zz[c].idx);
                                else { q[qn].old_idx = ev; q[qn].target = zz[c].idx; qn++; }
                            }
                            if (r2 >= 0) succ++;
                            else if (L->nbr[(size_t)node * L->M + (uint32_t)r] == zz[c].idx) L->nbr[(size_t)node * L->M + (uint32_t)r] = IDX_NONE;
                        }
                    }
                }
                for (uint32_t e = 0; e < qn; e++) remove_neighbor_by_idx(L, q[e].old_idx, q[e].target); /* end of round */
                uint32_t *t = pending; pending = next; next = t;
                np = nn;
            }
        }
        inserted += bs;
    }
    if (stats) memcpy(stats, st, sizeof(st));
    free(ord); free(z); free(zn); free(me); free(claim_round); free(claim_owner); free(pending); free(next); free(q);
    scratch_free(s);
    return rc;
}

uint32_t coso_meta_level_count(const coso_index *ix, uint32_t level) { return (ix->mlv && level <= ix->p.num_layers) ? ix->mlv[level].n : 0; }

/* flat export / import of the component (same conventions as coso_index_export_level, the pseudo root is an ordinary id) */
int coso_meta_export_level(const coso_index *ix, uint32_t level, uint32_t *node_ids, uint32_t *nbr_ids) {
    if (!ix->mlv || level > ix->p.num_layers) return COSO_ERR_INVALID;
    const level_t *L = &ix->mlv[level];
    idpair *ord = (idpair *)malloc((size_t)L->n * sizeof(idpair));
    for (uint32_t i = 0; i < L->n; i++) { ord[i].id = L->node_id[i]; ord[i].idx = i; }
    qsort(ord, L->n, sizeof(idpair), cmp_idpair);
    for (uint32_t k = 0; k < L->n; k++) {
        node_ids[k] = ord[k].id;
        for (uint32_t j = 0; j < L->M; j++) {
            uint32_t t = L->nbr[(size_t)ord[k].idx * L->M + j];
            nbr_ids[(size_t)k * L->M + j] = t == IDX_NONE ? COSO_SLOT_EMPTY : L->node_id[t];
        }
    }
    free(ord);
    return COSO_OK;
}

/* pseudo_level_probs (metadata/mod.rs:182-209): values/levels get num_levels + 1 entries */
void coso_pseudo_level_probs(int num_levels, int num_pseudo_nodes, double *values, uint8_t *levels) {
    int ilog10 = 0;
    for (int v = num_pseudo_nodes; v >= 10; v /= 10) ilog10++;
    int higher = ilog10 + 1, lower;
    if (higher > num_levels) { higher = 0; lower = num_levels; } else lower = num_levels - higher;
    int k = 0;
    if (higher > 0) {
        double hv[64];
        uint8_t hl[64];
        coso_level_probs(10.0, higher, hv, hl);
        for (int i = 0; i <= higher; i++) {
            if (hl[i] == 0) continue;
            values[k] = hv[i];
            levels[k] = (uint8_t)(lower + hl[i]);
            k++;
        }
    }
    for (int i = lower; i >= 0; i--) { values[k] = 0.0; levels[k] = (uint8_t)i; k++; }
}
