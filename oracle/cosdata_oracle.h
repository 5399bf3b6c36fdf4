/*
 * cosdata_oracle.h — CPU restatement ("oracle") of the cosdata dense/hybrid search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product: only
 * tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this library,
 * and there only as the checker / reported CPU baseline, never as a fallback for the HIP path.
 *
 * What it restates (cosdata/cosdata @ 2025-09-19, paths relative to the reference root):
 *   src/quantization/scalar.rs:10-52        ScalarQuantization::quantize
 *   src/models/common.rs:226-275            to_float_flag / quantize_to_u8_bits
 *   src/models/dot_product.rs:9-157         scalar dot products + dispatch
 *   src/models/dot_product/x86_64.rs:22-66,103-211,418-444   AVX2 u8 / quaternary / f32 order
 *   src/distance/{cosine,dotproduct,euclidean,hamming}.rs    metric dispatch + error behaviour
 *   src/models/types.rs:382-457             MetricResult total order, min/max
 *   src/models/fixedset.rs:1-29             PerformantFixedSet (lossy visited filter)
 *   src/vector_store.rs:256-445,1112-1204   ann_search / traverse_find_nearest / finalize_ann_results
 *   src/models/common.rs:373-429            get_max_insert_level / remove_duplicates_and_filter /
 *                                           generate_level_probs
 *   src/vector_store.rs:714-1109 + src/models/prob_node.rs:210-329   deterministic builder
 *   src/models/sparse_ann_query.rs:149-302  BM25 (search_bm25, get_idf)
 *   src/api/vectordb/search/repo.rs:311-340 RRF fusion
 *
 * Parity pinning: the reference holds NO golden vectors for quantize / cosine / walk / rerank /
 * BM25 / RRF and cannot be compiled here (no rustc).  The only known-answer
 * material in the reference's own tests — popcount patterns (x86_64.rs:676-723), the
 * generate_level_probs(10,9) table (common.rs:741-756), MetricResult ordering
 * (types.rs:1611-1633) and the quaternary identity (x86_64.rs:455-505) — is checked in
 * tests/test_oracle_kat.py.  Everything else: "parity unpinned by the reference's own tests";
 * the fixtures under tests/golden/ generated from this oracle are the pin.
 *
 * Documented deviations where the reference is nondeterministic:
 *   - ties in every sort/heap are broken by internal id, larger id = "greater"
 *     (reference: raw pointer value / sort_unstable order);
 *   - BM25 per-document score is summed in ascending term-hash order
 *     (reference: BinaryHeap order among equal doc ids is unspecified);
 *   - RRF ties: smaller id first (reference: FxHashMap iteration order).
 */
#ifndef COSDATA_ORACLE_H
#define COSDATA_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums shared with include/cosdata_hip.h (same numeric values) ---- */
enum { COSO_METRIC_COSINE = 0, COSO_METRIC_EUCLIDEAN = 1, COSO_METRIC_HAMMING = 2, COSO_METRIC_DOT = 3 };
enum { COSO_STORAGE_U8 = 0, COSO_STORAGE_SUBBYTE = 1, COSO_STORAGE_F16 = 2, COSO_STORAGE_F32 = 3 };
enum {
    COSO_OK = 0,
    COSO_ERR_STORAGE_MISMATCH = 1, /* DistanceError::StorageMismatch */
    COSO_ERR_CALCULATION = 2,      /* DistanceError::CalculationError (zero norm, bad resolution) */
    COSO_ERR_INVALID = 3,
    COSO_ERR_UNIMPLEMENTED = 4     /* reference `unimplemented!()` arms */
};

#define COSO_ROOT_ID 0xFFFFFFFFu  /* vector_store.rs:47 InternalId::from(u32::MAX) */
#define COSO_QUERY_ID 0xFFFFFFFEu /* indexes/hnsw/mod.rs:398 */
#define COSO_SLOT_EMPTY 0xFFFFFFFDu /* null neighbour pointer in the flat export */
#define COSO_VISITED_REF 0   /* PerformantFixedSet replica (ID parity) */
#define COSO_VISITED_EXACT 1 /* exact visited set (recall mode) */

/* ---- numeric kernels ---- */
size_t coso_code_bytes(int storage, int resolution, int dim);
/* quantize one vector; `code` gets coso_code_bytes() bytes; SubByte codes are plane-major
 * (plane p at code + p*ceil(dim/8)), plane 0 = MSB exactly as quantize_to_u8_bits writes it. */
int coso_quantize(const float *x, int dim, int storage, int resolution, float lo, float hi,
                  void *code, float *mag);
uint64_t coso_dot_u8(const uint8_t *a, const uint8_t *b, int n);
uint64_t coso_dot_u8_scalar(const uint8_t *a, const uint8_t *b, int n);
float coso_dot_f32(const float *a, const float *b, int n);        /* AVX2+FMA order */
float coso_dot_f32_scalar_order(const float *a, const float *b, int n); /* same order, scalar fmaf */
float coso_dot_f16(const uint16_t *a, const uint16_t *b, int n);
float coso_dot_subbyte(const uint8_t *x, const uint8_t *y, int resolution, int plane_bytes, int *status);
float coso_dot_quaternary_scalar(const uint8_t *x, const uint8_t *y, int plane_bytes);
uint64_t coso_count_ones(const uint8_t *p, int n); /* nibble-LUT popcount, x86_64.rs:190-211 */
float coso_seq_norm_f32(const float *x, int n);    /* sqrt(sequential non-fused sum x*x) */
uint16_t coso_f32_to_f16(float x);
float coso_f16_to_f32(uint16_t h);
/* DistanceMetric::calculate on two stored vectors (Base,Base arm). */
int coso_distance(int metric, int storage, int resolution, int dim, const void *x, float x_mag,
                  const void *y, float y_mag, float *out);
/* MetricResult::cmp: returns -1/0/1 (types.rs:401-411) */
int coso_metric_cmp(int metric, float a, float b);
void coso_level_probs(double x, int num_levels, double *values, uint8_t *levels); /* num_levels+1 entries */
int coso_max_insert_level(double x, const double *values, const uint8_t *levels, int n);

/* "auto" quantization range sampling (indexes/hnsw/mod.rs:202-351) */
void coso_sample_values_range(const float *x, uint64_t total, float clamp_margin_percent, float *lo, float *hi);

/* PerformantFixedSet, exposed for unit tests */
typedef struct { uint64_t *buckets; uint32_t len; } coso_fixedset;
void coso_fixedset_insert(coso_fixedset *s, uint32_t v);
int coso_fixedset_is_member(const coso_fixedset *s, uint32_t v);

/* ---- HNSW index ---- */
typedef struct {
    uint32_t dim;
    uint32_t metric;
    uint32_t storage;
    uint32_t resolution;     /* SubByte bits (1,2,3) */
    float range_lo, range_hi; /* values_range */
    uint32_t num_layers;     /* HNSWHyperParams.num_layers (levels 0..num_layers) */
    uint32_t neighbors_count;         /* M   */
    uint32_t level0_neighbors_count;  /* M0  */
    uint32_t ef_construction;
    uint32_t ef_search;
    uint32_t shortlist_size; /* config.search.shortlist_size */
    uint32_t visited_mode;   /* COSO_VISITED_* */
    uint64_t seed;           /* builder RNG seed (levels + root vector) */
} coso_params;

typedef struct coso_index coso_index;

coso_index *coso_index_create(const coso_params *p);
void coso_index_destroy(coso_index *ix);
/* Give the index its corpus: raw f32 [n][dim] (borrowed, must outlive the index).
 * Quantizes every vector with the index storage (ScalarQuantization::quantize). */
int coso_index_set_vectors(coso_index *ix, const float *raw, uint32_t n);
/* Corpora whose raw f32 table does not fit in host memory: allocate the code table, quantize chunks of rows as they are
 * streamed in (same ScalarQuantization::quantize), and give the exact rerank a subset of raw rows (ids ascending, borrowed). */
int coso_index_alloc_vectors(coso_index *ix, uint32_t n);
int coso_index_quantize_rows(coso_index *ix, uint32_t start, const float *raw_chunk, uint32_t m);
int coso_index_set_raw_subset(coso_index *ix, const uint32_t *ids_sorted, const float *rows, uint32_t m);
/* Deterministic single-threaded builder with reference edge semantics (App. A.4).
 * Inserts ids [0, n) in order.  Creates the root first (random vector in values_range). */
int coso_index_build(coso_index *ix);
/* Batch-synchronous schedule (CPU statement of the device builder, cosdata_amd/csrc/builder.hip). */
int coso_index_build_batched(coso_index *ix, uint32_t batch_size);
/* prototype of the round-synchronous link schedule; stats[4] = rounds, (batch,level) pairs, nodes, first-round nodes */
int coso_index_build_rounds(coso_index *ix, uint32_t batch_size, int greedy, uint64_t *stats);
/* index_embeddings on a LIVE index (vector_store.rs:714-780 called again; index_embedding :782-975 per vector): m more vectors take
 * the ids [n, n + m); raw_all = the caller's whole [n + m][dim] table (borrowed).  Then coso_index_build_rounds_continue inserts them
 * with the schedule of coso_index_build_rounds continued (same RNG stream, same batch rule); needs a graph built by
 * coso_index_build_rounds on this handle.  CPU statement of cos_index_append (include/cosdata_hip.h). */
int coso_index_append_vectors(coso_index *ix, const float *raw_all, uint32_t m);
int coso_index_build_rounds_continue(coso_index *ix, uint32_t batch_size, uint64_t *stats);
int coso_index_can_continue(const coso_index *ix);
/* delete_embedding (vector_store.rs:1206-1400) for one internal id: per level a walk for the vector's own code (ef 512, keep 100, filter
 * not pre-seeded), the node's neighbours drop their back edges, a neighbour left without any neighbour is linked again from the walk's
 * results; the node's own slots are emptied.  CPU statement of cos_index_delete. */
int coso_index_delete(coso_index *ix, uint32_t id);
/* the state a RELOADED (imported) graph continues from: slot similarities recomputed, lowest caches by the deserializer's rule
 * (prob_node.rs:145-181), the seed's RNG stream advanced past the resident vectors; coso_index_append_vectors / _build_rounds_continue /
 * coso_index_delete then work on it.  CPU statement of cos_index_restore_link_state. */
int coso_index_restore_link_state(coso_index *ix);
/* Flat export/import (the same arrays include/cosdata_hip.h uploads). */
uint32_t coso_index_level_count(const coso_index *ix, uint32_t level);
int coso_index_export_level(const coso_index *ix, uint32_t level, uint32_t *node_ids /*[n_l]*/,
                            uint32_t *nbr_ids /*[n_l][M_l]*/, float *nbr_sims /*optional*/);
int coso_index_import_level(coso_index *ix, uint32_t level, uint32_t n_nodes, const uint32_t *node_ids,
                            const uint32_t *nbr_ids);
const float *coso_index_root_raw(const coso_index *ix); /* root's f32 vector [dim] */
int coso_index_set_root_raw(coso_index *ix, const float *root);
const void *coso_index_codes(const coso_index *ix);     /* [n+1][code_bytes], row n = root */
const float *coso_index_mags(const coso_index *ix);     /* [n+1] */
void coso_index_set_ef_search(coso_index *ix, uint32_t ef);
void coso_index_set_visited_mode(coso_index *ix, uint32_t mode);
void coso_index_clear_graph(coso_index *ix); /* drop all levels, keep the vectors */
int coso_index_set_neighbors(coso_index *ix, uint32_t m);          /* same for neighbors_count (levels >= 1) */
int coso_index_set_level0_neighbors(coso_index *ix, uint32_t m0); /* clears the graph; the next one has M0 = m0 (power of two <= 256) */

/* per-query device-comparable counters */
typedef struct {
    uint64_t evals;      /* distance evaluations */
    uint64_t expansions; /* popped + expanded nodes */
    uint64_t adj_bytes;  /* expansions_l * M_l * 4 summed over levels */
} coso_stats;

/* HNSWIndex::search_internal for B queries (data-parallel over queries with `threads`
 * OpenMP workers, like IndexOps::batch_search's rayon fan-out).  Outputs are [B][top_k];
 * out_counts[b] = number of valid entries.  Returns the first error status (the reference's
 * `collect::<Result<..>>` fails the whole batch), per-query statuses in out_status if non-NULL. */
int coso_search_batch(const coso_index *ix, const float *queries, uint32_t B, uint32_t top_k,
                      uint32_t *out_ids, float *out_scores, uint32_t *out_counts, int32_t *out_status,
                      coso_stats *stats /*[B] or NULL*/, int threads);
/* walk + remove_duplicates_and_filter only: out_ids [B][5*top_k] = the candidates whose raw rows the rerank reads */
int coso_candidates_batch(const coso_index *ix, const float *queries, uint32_t B, uint32_t top_k, uint32_t *out_ids,
                          uint32_t *out_counts, int threads);
/* Raw walk output before finalisation: ann_search's concatenated per-level lists.
 * out_ids/out_sims: [ (num_layers+1) * 100 ]; returns count or negative status. */
int coso_ann_search(const coso_index *ix, const float *query, uint32_t *out_ids, float *out_sims,
                    uint32_t *level_counts /*[num_layers+1], top level first*/);
/* ids [B][5*top_k] of the flat search's rerank candidates (streamed corpora: fetch those raw rows, set_raw_subset, search) */
int coso_flat_candidates_batch(const coso_index *ix, const float *queries, uint32_t B, uint32_t top_k, uint32_t *out_ids,
                               uint32_t *out_counts, int threads);
/* exhaustive search over the quantized codes + exact rerank of the best 5k (device "flat" mode) */
int coso_flat_search_batch(const coso_index *ix, const float *queries, uint32_t B, uint32_t top_k, uint32_t *out_ids,
                           float *out_scores, uint32_t *out_counts, int threads);
/* exact brute-force cosine top-k on raw f32 (reference-order dot, same formula as the rerank) */
int coso_bruteforce_topk(const float *raw, uint32_t n, uint32_t dim, const float *queries, uint32_t B,
                         uint32_t k, uint32_t *out_ids, float *out_scores, int threads);

/* ---- metadata-filtered search: the pseudo-root component ---- */
/* mdim metadata dimensions per node, `max_replicas` internal ids reserved per embedding (base id = vector row * max_replicas,
 * replica i = base id + i; pseudo nodes: u32::MAX - 257 (the pseudo root) .. u32::MAX - 2).  Call after set_vectors. */
int coso_meta_enable(coso_index *ix, uint32_t mdim, uint32_t max_replicas);
/* node table: ascending replica ids (pseudo root + pseudo nodes last), metadata dimensions mbits[n][mdim] */
int coso_meta_set_nodes(coso_index *ix, uint32_t n_nodes, const uint32_t *ids_sorted, const int32_t *mbits);
/* builds the component with the reference's index-side rules; max_levels[n_nodes] = the level drawn per node (table order) */
int coso_meta_build(coso_index *ix, const uint8_t *max_levels);
/* the same component in the batch-synchronous schedule of a device-side builder; batch_size = 1 is coso_meta_build; stats[3] optional */
int coso_meta_build_rounds(coso_index *ix, const uint8_t *max_levels, uint32_t batch_size, uint64_t *stats);
uint32_t coso_meta_level_count(const coso_index *ix, uint32_t level);
int coso_meta_export_level(const coso_index *ix, uint32_t level, uint32_t *node_ids, uint32_t *nbr_ids);
/* search_internal with a filter: filters of query b = rows [filter_off[b], filter_off[b+1]) of filter_dims[][mdim] (-1/0/1) */
int coso_search_filtered_batch(const coso_index *ix, const float *queries, uint32_t B, const uint32_t *filter_off, const int32_t *filter_dims,
                               uint32_t top_k, uint32_t *out_ids, float *out_scores, uint32_t *out_counts, int32_t *out_status, int threads);
int coso_ann_search_filtered(const coso_index *ix, const float *query, const int32_t *filter_dims, uint32_t nf, uint32_t *out_ids, float *out_sims,
                             uint32_t *level_counts);
void coso_pseudo_level_probs(int num_levels, int num_pseudo_nodes, double *values, uint8_t *levels); /* metadata/mod.rs:182-209 */

/* ---- BM25 + RRF (config c5) ---- */
float coso_bm25_idf(uint32_t documents_count, uint32_t containing);
float coso_bm25_tf(uint32_t count, uint32_t doc_len, float avg_len, float k1, float b);
/* CSR postings: term_hashes[T] ascending, offsets[T+1], doc_ids/tfs[nnz] doc-id ascending per term */
int coso_bm25_search(const uint32_t *term_hashes, const uint64_t *offsets, uint32_t T, const uint32_t *doc_ids,
                     const float *tfs, uint32_t documents_count, const uint32_t *query_terms, uint32_t nq,
                     uint32_t top_k, uint32_t *out_ids, float *out_scores);
int coso_rrf_fuse(const uint32_t *dense_ids, uint32_t nd, const uint32_t *sparse_ids, uint32_t ns, float k_rrf,
                  uint32_t top_k, uint32_t *out_ids, float *out_scores);

/* ---- learned-sparse inverted index ---- */
uint8_t coso_sparse_quantize(float value, float values_upper_bound, int bits); /* inverted_index.rs:168-172 */
/* CSR: dims[T] ascending; ids of (dimension t, key q) = vec_ids[key_off[t*(2^bits+1)+q] .. key_off[t*(2^bits+1)+q+1]) */
int coso_sparse_search(const uint32_t *dims, uint32_t T, const uint64_t *key_off, const uint32_t *vec_ids, uint32_t n_vectors, int bits,
                       float values_upper_bound, float early_terminate_threshold, const uint32_t *q_dims, const float *q_vals, uint32_t nq,
                       uint32_t k_with_reranking, uint32_t *out_ids, uint32_t *out_sims, uint32_t cap);
int coso_sparse_rerank(const uint64_t *row_off, const uint32_t *raw_dims, const float *raw_vals, const uint32_t *cand_ids, uint32_t m,
                       const uint32_t *q_dims, const float *q_vals, uint32_t nq, uint32_t top_k, uint32_t *out_ids, float *out_scores);

#ifdef __cplusplus
}
#endif
#endif
