"""Host-side mirror of the reference's operators around the dense walk:

    distance_batch      DistanceMetric::calculate            src/models/types.rs:469
    BM25Index           TFIDFIndex / search_bm25             src/indexes/tf_idf/mod.rs:243, src/models/sparse_ann_query.rs:149
    rrf_fuse_batch      RRF fusion of hybrid_search          src/api/vectordb/search/repo.rs:311-340

All numeric work runs in libcosdata_hip.so (gfx950 kernels); no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check
from .index import DistanceMetric, StorageType, _c, _p


def distance_batch(metric: DistanceMetric, storage_type: StorageType, dim: int, x_codes, x_mags, y_codes, y_mags, pair_x, pair_y):
    """out[p] = metric(x[pair_x[p]], y[pair_y[p]]) on stored vectors in the reference Storage layout.
    Returns (values f32[n_pairs], status i32[n_pairs]); status 2 = CalculationError (zero norm), 1 = StorageMismatch."""
    xc, yc = _c(x_codes, np.uint8), _c(y_codes, np.uint8)
    xm, ym = _c(x_mags, np.float32), _c(y_mags, np.float32)
    px, py = _c(pair_x, np.uint32), _c(pair_y, np.uint32)
    out = np.zeros(px.size, np.float32)
    status = np.zeros(px.size, np.int32)
    check(_lib.lib().cos_distance_batch(int(metric), int(storage_type.kind), storage_type.resolution, dim, _p(xc), _p(xm), xc.shape[0],
                                        _p(yc), _p(ym), yc.shape[0], _p(px), _p(py), px.size, _p(out), _p(status)))
    return out, status


class BM25Index:
    """Device-resident CSR postings of a TFIDFIndexRoot: term hashes ascending, offsets[T+1], (doc id, stored tf)."""

    def __init__(self, term_hashes, offsets, doc_ids, tfs, documents_count: int, device: int = 0):
        th, off = _c(term_hashes, np.uint32), _c(offsets, np.uint64)
        di, tf = _c(doc_ids, np.uint32), _c(tfs, np.float32)
        self._h = C.c_void_p()
        check(_lib.lib().cos_bm25_create(device, _p(th), _p(off), th.size, _p(di), _p(tf), documents_count, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            _lib.lib().cos_bm25_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def search_batch(self, q_terms, q_offsets, top_k: int):
        """search_bm25 for B queries given as pre-hashed terms (CSR). -> ids [B][k], scores [B][k], counts [B]."""
        qt, qo = _c(q_terms, np.uint32), _c(q_offsets, np.uint32)
        B = qo.size - 1
        ids = np.full((B, top_k), 0xFFFFFFFF, np.uint32)
        sc = np.zeros((B, top_k), np.float32)
        cnt = np.zeros(B, np.uint32)
        check(_lib.lib().cos_bm25_search_batch(self._h, _p(qt), _p(qo), B, top_k, _p(ids), _p(sc), _p(cnt)))
        return ids, sc, cnt


def _bm25_search_batch_device(self, q_terms, q_offsets, top_k: int, out_ids_ptr: int, out_scores_ptr: int, out_counts_ptr: int, stream: int = 0):
    """same scoring, results left in device memory ([B][k] ids, [B][k] scores, [B] counts); enqueued on `stream`"""
    qt, qo = _c(q_terms, np.uint32), _c(q_offsets, np.uint32)
    check(_lib.lib().cos_bm25_search_batch_device(self._h, _p(qt), _p(qo), qo.size - 1, top_k, C.c_void_p(out_ids_ptr), C.c_void_p(out_scores_ptr),
                                                  C.c_void_p(out_counts_ptr), C.c_void_p(stream)))


BM25Index.search_batch_device = _bm25_search_batch_device


def rrf_fuse_batch(dense_ids, dense_counts, sparse_ids, sparse_counts, fusion_constant_k: float, top_k: int):
    d, s = _c(dense_ids, np.uint32), _c(sparse_ids, np.uint32)
    dc, sc_ = _c(dense_counts, np.uint32), _c(sparse_counts, np.uint32)
    B = d.shape[0]
    ids = np.full((B, top_k), 0xFFFFFFFF, np.uint32)
    sc = np.zeros((B, top_k), np.float32)
    cnt = np.zeros(B, np.uint32)
    check(_lib.lib().cos_rrf_fuse_batch(_p(d), _p(dc), d.shape[1], _p(s), _p(sc_), s.shape[1], B, fusion_constant_k, top_k,
                                        _p(ids), _p(sc), _p(cnt)))
    return ids, sc, cnt


def hybrid_search_batch(index, bm25: "BM25Index", queries, q_terms, q_offsets, top_k: int, fusion_constant_k: float = 60.0):
    """repo::hybrid_search for B queries in one call: dense (top_k*3) + BM25 (top_k*3) concurrently on the device, RRF there"""
    q = index._queries(queries)
    qt, qo = _c(q_terms, np.uint32), _c(q_offsets, np.uint32)
    B = q.shape[0]
    ids = np.full((B, top_k), 0xFFFFFFFF, np.uint32)
    sc = np.zeros((B, top_k), np.float32)
    cnt = np.zeros(B, np.uint32)
    check(_lib.lib().cos_hybrid_search_batch(index._h, bm25._h, _p(q), _p(qt), _p(qo), B, top_k, fusion_constant_k, _p(ids), _p(sc), _p(cnt)))
    return ids, sc, cnt


class InvertedIndex:
    """Device-resident learned-sparse inverted index (src/indexes/inverted/mod.rs, src/models/inverted_index.rs) as CSR:
    dims ascending, key_offsets [T][2^bits + 1], vec_ids; optional raw sparse vectors (CSR) for the raw-value rerank."""

    def __init__(self, quantization_bits: int, values_upper_bound: float, dims, key_offsets, vec_ids, n_vectors: int,
                 raw_row_offsets=None, raw_dims=None, raw_vals=None, device: int = 0):
        d, ko, vi = _c(dims, np.uint32), _c(key_offsets, np.uint64), _c(vec_ids, np.uint32)
        raw = None
        if raw_row_offsets is not None:
            raw = (_c(raw_row_offsets, np.uint64), _c(raw_dims, np.uint32), _c(raw_vals, np.float32))
        self._h = C.c_void_p()
        check(_lib.lib().cos_sparse_create(device, quantization_bits, values_upper_bound, _p(d), d.size, _p(ko), _p(vi), n_vectors,
                                           _p(raw[0]) if raw else None, _p(raw[1]) if raw else None, _p(raw[2]) if raw else None,
                                           C.byref(self._h)))

    @classmethod
    def from_vectors(cls, quantization_bits: int, values_upper_bound: float, row_offsets, raw_dims, raw_vals, keep_raw: bool = True,
                     device: int = 0):
        """InvertedIndex::insert for ids 0 .. n-1 (cos_sparse_create_from_vectors): the CSR is built by the library on the host"""
        ro, rd, rv = _c(row_offsets, np.uint64), _c(raw_dims, np.uint32), _c(raw_vals, np.float32)
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        check(_lib.lib().cos_sparse_create_from_vectors(device, quantization_bits, values_upper_bound, ro.size - 1, _p(ro), _p(rd), _p(rv),
                                                        1 if keep_raw else 0, C.byref(self._h)))
        return self

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            _lib.lib().cos_sparse_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def search_batch(self, q_dims, q_vals, q_offsets, top_k: int, early_terminate_threshold: float = 0.0, reranking_factor: int = 0):
        """InvertedIndex::search_internal for B queries (CSR pairs): ids [B][k], scores [B][k], counts [B]"""
        qd, qv, qo = _c(q_dims, np.uint32), _c(q_vals, np.float32), _c(q_offsets, np.uint32)
        B = qo.size - 1
        ids = np.full((B, top_k), 0xFFFFFFFF, np.uint32)
        scores = np.zeros((B, top_k), np.float32)
        counts = np.zeros(B, np.uint32)
        check(_lib.lib().cos_sparse_search_batch(self._h, _p(qd), _p(qv), _p(qo), B, top_k, early_terminate_threshold, reranking_factor,
                                                 _p(ids), _p(scores), _p(counts)))
        return ids, scores, counts


def _sparse_last_stats(self):
    """kernel time + visited postings of the most recent search_batch on this handle"""
    st = _lib.CosSparseStats()
    check(_lib.lib().cos_sparse_last_stats(self._h, C.byref(st)))
    return st


InvertedIndex.last_stats = _sparse_last_stats


def _sparse_packed(self) -> bool:
    """True when the handle keeps one packed u32 per posting (cos_sparse_layout): the default since round 5 wherever vector ids fit 24 bits"""
    v = C.c_uint32(0)
    check(_lib.lib().cos_sparse_layout(self._h, C.byref(v)))
    return bool(v.value)


InvertedIndex.packed = property(_sparse_packed)


def sparse_build_csr(quantization_bits: int, values_upper_bound: float, row_offsets, raw_dims, raw_vals):
    """cos_sparse_build_csr (host code, no device): raw sparse vectors in id order -> (dims, key_offsets, vec_ids)"""
    ro, rd, rv = _c(row_offsets, np.uint64), _c(raw_dims, np.uint32), _c(raw_vals, np.float32)
    n, nd = ro.size - 1, C.c_uint32(0)
    L = _lib.lib()
    check(L.cos_sparse_build_csr(quantization_bits, values_upper_bound, n, _p(ro), _p(rd), _p(rv), None, None, None, C.byref(nd)))
    dims = np.zeros(max(nd.value, 1), np.uint32)
    ko = np.zeros(max(nd.value, 1) * ((1 << quantization_bits) + 1), np.uint64)
    ids = np.zeros(max(int(ro[-1]), 1), np.uint32)
    check(L.cos_sparse_build_csr(quantization_bits, values_upper_bound, n, _p(ro), _p(rd), _p(rv), _p(dims), _p(ko), _p(ids), C.byref(nd)))
    return dims[:nd.value], ko[:nd.value * ((1 << quantization_bits) + 1)], ids[:int(ro[-1])]


STEM_FN = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.POINTER(C.c_char), C.c_size_t, C.POINTER(C.c_char), C.c_size_t)


def stem_english(word: str) -> str:
    """the library's English Snowball stemmer (cos_stem_english) on one lowercased token"""
    raw = word.encode("utf-8")
    out = C.create_string_buffer(len(raw) + 8)
    n = _lib.lib().cos_stem_english(None, raw, len(raw), out, len(raw) + 8)
    return out.raw[:n].decode("utf-8")


def process_text(text: str, max_token_len: int = 40, average_document_length: float = 1.0, k1: float = 1.5, b: float = 0.75, stemmer="english"):
    """TFIDFIndex's process_text: text -> (term hashes ascending u32[], stored BM25 term frequencies f32[]).
    `stemmer`: "english" (default) = the library's English Snowball stemmer, like the reference's `Stemmer::create()`; None =
    hash the lowercased token unstemmed; or a callable str -> str (e.g. a shim over the host's own stemmer)."""
    raw = text.encode("utf-8")
    cap = max(16, len(raw) // 2 + 4)
    hashes = np.zeros(cap, np.uint32)
    tfs = np.zeros(cap, np.float32)
    n = C.c_uint32()
    cb = None
    if stemmer == "english":
        cb = C.cast(_lib.lib().cos_stem_english, C.c_void_p)
    elif stemmer is not None:
        def _shim(_ctx, tok, tok_len, out, out_cap):
            res = stemmer(C.string_at(tok, tok_len).decode("utf-8")).encode("utf-8")[:out_cap]
            C.memmove(out, res, len(res))
            return len(res)
        cb = STEM_FN(_shim)
    check(_lib.lib().cos_text_process(raw, len(raw), max_token_len, average_document_length, k1, b, C.cast(cb, C.c_void_p) if cb is not None else None, None,
                                      _p(hashes), _p(tfs), cap, C.byref(n)))
    return hashes[:n.value].copy(), tfs[:n.value].copy()


def count_tokens(text: str, max_token_len: int = 40) -> int:
    raw = text.encode("utf-8")
    return int(_lib.lib().cos_text_count_tokens(raw, len(raw), max_token_len))
