"""Host-side mirror of the reference's operator interface for the dense search path.

Names, argument meaning and error behaviour follow cosdata's Rust API (paths relative to the
reference root) so the parity tests read like the reference's own:

    HNSWHyperParams            src/indexes/hnsw/types.rs:10-17
    DistanceMetric             src/models/types.rs:462-468
    StorageType                src/quantization/mod.rs:20-25
    HNSWIndex.search_internal  src/indexes/hnsw/mod.rs:390-440
    HNSWIndex.batch_search     src/indexes/mod.rs:260-272
    ScalarQuantization.quantize   src/quantization/scalar.rs:10-52

Everything numeric happens in libcosdata_hip.so (hand-written gfx950 kernels) through the C ABI
of include/cosdata_hip.h.  There is no CPU fallback here.
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import CosdataError, CosFlatStats, CosParams, CosSearchStats, CosTimingSummary, CosWalkSplit, check

ROOT_ID, QUERY_ID, SLOT_EMPTY = 0xFFFFFFFF, 0xFFFFFFFE, 0xFFFFFFFD
VISITED_REF, VISITED_EXACT = 0, 1


class DistanceMetric(enum.IntEnum):
    Cosine = 0
    Euclidean = 1
    Hamming = 2
    DotProduct = 3


class StorageKind(enum.IntEnum):
    UnsignedByte = 0
    SubByte = 1
    HalfPrecisionFP = 2
    FullPrecisionFP = 3


@dataclass(frozen=True)
class StorageType:
    """quantization::StorageType; SubByte carries its resolution (bits)."""
    kind: StorageKind
    resolution: int = 0

    @staticmethod
    def UnsignedByte():
        return StorageType(StorageKind.UnsignedByte)

    @staticmethod
    def SubByte(resolution: int):
        return StorageType(StorageKind.SubByte, resolution)

    @staticmethod
    def FullPrecisionFP():
        return StorageType(StorageKind.FullPrecisionFP)

    @staticmethod
    def HalfPrecisionFP():
        return StorageType(StorageKind.HalfPrecisionFP)


@dataclass
class HNSWHyperParams:
    """indexes/hnsw/types.rs:10-17 with config.toml:20-24 defaults."""
    num_layers: int = 9
    ef_construction: int = 128
    ef_search: int = 256
    level_0_neighbors_count: int = 64
    neighbors_count: int = 32


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class ScalarQuantization:
    """QuantizationMetric::Scalar — quantize() runs the device kernel and returns codes in the
    reference's Storage layout."""

    @staticmethod
    def quantize(vectors, storage_type: StorageType, values_range: Tuple[float, float]):
        x = _c(np.atleast_2d(vectors), np.float32)
        n, d = x.shape
        cb = _lib.lib().cos_code_bytes(int(storage_type.kind), storage_type.resolution, d)
        codes = np.zeros((n, cb), np.uint8)
        mags = np.zeros(n, np.float32)
        check(_lib.lib().cos_quantize_batch(int(storage_type.kind), storage_type.resolution, d, values_range[0], values_range[1],
                                            _p(x), n, _p(codes), _p(mags)))
        return codes, mags


def sample_values_range(sample, clamp_margin_percent: float = 1.0):
    """"auto" quantization (indexes/hnsw/mod.rs:202-351): values_range from the first sample_threshold embeddings."""
    x = _c(np.atleast_2d(sample), np.float32)
    lo, hi = C.c_float(), C.c_float()
    check(_lib.lib().cos_sample_values_range(_p(x), x.shape[0], x.shape[1], clamp_margin_percent, C.byref(lo), C.byref(hi)))
    return (lo.value, hi.value)


class HNSWIndex:
    """Device-resident snapshot of one HNSW index (one shard)."""

    def __init__(self, dim: int, hnsw_params: Optional[HNSWHyperParams] = None,
                 distance_metric: DistanceMetric = DistanceMetric.Cosine,
                 storage_type: StorageType = StorageType.UnsignedByte(),
                 values_range: Tuple[float, float] = (-1.0, 1.0), shortlist_size: int = 64,
                 visited_mode: int = VISITED_REF, device: int = 0, id_base: int = 0, seed: int = 42):
        self.dim = dim
        self.hnsw_params = hnsw_params or HNSWHyperParams()
        self.distance_metric = distance_metric
        self.storage_type = storage_type
        self.values_range = values_range
        hp = self.hnsw_params
        self._params = CosParams(C.sizeof(CosParams), _lib.ABI_VERSION, dim, int(distance_metric), int(storage_type.kind),
                                 storage_type.resolution, values_range[0], values_range[1], hp.num_layers, hp.neighbors_count,
                                 hp.level_0_neighbors_count, hp.ef_construction, hp.ef_search, shortlist_size, visited_mode,
                                 device, id_base, 0, seed)
        self._h = C.c_void_p()
        check(_lib.lib().cos_index_create(C.byref(self._params), C.byref(self._h)))
        self.n = 0
        self._keepalive = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            _lib.lib().cos_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- uploads ----------------------------------------------------------------------------
    def upload_vectors(self, raw):
        raw = _c(raw, np.float32)
        if raw.ndim != 2 or raw.shape[1] != self.dim:
            raise CosdataError(_lib.ERR_INVALID, f"vectors must be [n][{self.dim}] f32, got shape {tuple(raw.shape)}")
        check(_lib.lib().cos_index_upload_vectors(self._h, _p(raw), raw.shape[0], 0))
        self.n = raw.shape[0]
        self.mdim = 0  # a re-upload drops the metadata schema with the graph (cosdata_hip.h): enable_metadata again before filtering
        return self

    def upload_vectors_device(self, dev_ptr: int, n: int, keepalive=None):
        """raw f32 [n][dim] already in HBM (e.g. a torch tensor's data_ptr()); borrowed, not copied."""
        check(_lib.lib().cos_index_upload_vectors(self._h, C.c_void_p(dev_ptr), n, 1))
        self.n = n
        self.mdim = 0  # see upload_vectors
        self._keepalive = keepalive
        return self

    def _queries(self, queries) -> np.ndarray:
        """[B][dim] contiguous f32; a wrong dimension is an error, never an out-of-bounds host read on the C side."""
        q = _c(np.atleast_2d(queries), np.float32)
        if q.ndim != 2 or q.shape[1] != self.dim:
            raise CosdataError(_lib.ERR_INVALID, f"queries must be [B][{self.dim}] f32, got shape {tuple(q.shape)}")
        return q

    def set_root(self, root_raw):
        r = _c(root_raw, np.float32)
        if r.size != self.dim:
            raise CosdataError(_lib.ERR_INVALID, f"root vector must hold {self.dim} values, got {r.size}")
        check(_lib.lib().cos_index_set_root(self._h, _p(r)))
        return self

    def level_M(self, level: int) -> int:
        return self.hnsw_params.level_0_neighbors_count if level == 0 else self.hnsw_params.neighbors_count

    def upload_graph_level(self, level: int, node_ids, nbr_ids):
        ids, nbr = _c(node_ids, np.uint32), _c(nbr_ids, np.uint32)
        assert nbr.shape == (ids.size, self.level_M(level))
        check(_lib.lib().cos_index_upload_graph_level(self._h, level, ids.size, _p(ids), _p(nbr)))
        return self

    def upload_graph(self, levels: Sequence[Tuple[np.ndarray, np.ndarray]], root_raw):
        self.set_root(root_raw)
        for l, (ids, nbr) in enumerate(levels):
            self.upload_graph_level(l, ids, nbr)
        return self

    def level_count(self, level: int) -> int:
        n = C.c_uint32()
        check(_lib.lib().cos_index_level_count(self._h, level, C.byref(n)))
        return n.value

    def download_graph(self):
        out = []
        for l in range(self.hnsw_params.num_layers + 1):
            n = self.level_count(l)
            ids = np.zeros(n, np.uint32)
            nbr = np.zeros((n, self.level_M(l)), np.uint32)
            check(_lib.lib().cos_index_download_graph_level(self._h, l, _p(ids), _p(nbr)))
            out.append((ids, nbr))
        return out

    def download_root(self):
        r = np.zeros(self.dim, np.float32)
        check(_lib.lib().cos_index_download_root(self._h, _p(r)))
        return r

    def download_codes(self):
        cb = _lib.lib().cos_code_bytes(int(self.storage_type.kind), self.storage_type.resolution, self.dim)
        codes = np.zeros((self.n + 1, cb), np.uint8)
        mags = np.zeros(self.n + 1, np.float32)
        check(_lib.lib().cos_index_download_codes(self._h, _p(codes), _p(mags)))
        return codes, mags

    def load_reference_dir(self, dense_hnsw_dir: str, root_ptr_offset: int, verify_codes: bool = False):
        """graph (+ root code) of an index persisted by the reference server -> this handle; vectors must be uploaded first"""
        check(_lib.lib().cos_index_load_reference_dir(self._h, dense_hnsw_dir.encode(), root_ptr_offset, 1 if verify_codes else 0))
        return self

    def build(self, batch_size: int = 0):
        """vector_store::index_embeddings on the device."""
        check(_lib.lib().cos_index_build(self._h, batch_size))
        return self

    def append(self, raw_new, batch_size: int = 0):
        """vector_store::index_embeddings on a LIVE index (cos_index_append): `raw_new` [m][dim] take the ids [n, n + m) and are inserted
        into the resident graph — the schedule of build() continued, no rebuild.  For an index that owns its raw rows (upload_vectors)."""
        x = _c(np.atleast_2d(raw_new), np.float32)
        if x.ndim != 2 or x.shape[1] != self.dim or x.shape[0] == 0:
            raise CosdataError(_lib.ERR_INVALID, f"new vectors must be [m][{self.dim}] f32, got shape {tuple(x.shape)}")
        check(_lib.lib().cos_index_append(self._h, _p(x), x.shape[0], 0, batch_size))
        self.n += x.shape[0]
        return self

    def append_device(self, dev_ptr_all: int, m: int, batch_size: int = 0, keepalive=None):
        """the same for an index that BORROWS its raw rows (upload_vectors_device): `dev_ptr_all` = the caller's whole grown table
        [n + m][dim] in HBM (rows [0, n) unchanged), borrowed from now on"""
        check(_lib.lib().cos_index_append(self._h, C.c_void_p(dev_ptr_all), m, 1, batch_size))
        self.n += m
        self._keepalive = keepalive
        return self

    def delete(self, ids):
        """vector_store::delete_embedding (cos_index_delete) for the internal ids of `ids`, in the given order"""
        a = _c(np.atleast_1d(ids), np.uint32)
        check(_lib.lib().cos_index_delete(self._h, _p(a), a.size))
        return self

    def restore_link_state(self):
        """an UPLOADED graph becomes appendable / deletable (cos_index_restore_link_state): similarities recomputed, lowest caches by the
        reference's reload rule"""
        check(_lib.lib().cos_index_restore_link_state(self._h))
        return self

    def release_link_state(self):
        """frees what build() keeps for append() (4 bytes per neighbour slot); append() then fails with NotReady"""
        check(_lib.lib().cos_index_release_link_state(self._h))
        return self

    # ---- search -----------------------------------------------------------------------------
    def set_ef_search(self, ef: int):
        check(_lib.lib().cos_index_set_ef_search(self._h, ef))
        self.hnsw_params.ef_search = ef

    def set_coalescing(self, max_queries: int, window_us: int = 200):
        """Fuse concurrent batch_search() calls of different threads into one launch (0 = off)."""
        check(_lib.lib().cos_index_set_coalescing(self._h, max_queries, window_us))

    def coalescing_stats(self) -> dict:
        """cos_index_coalescing_stats: launches / queries / requests and why the launches left, since the last set_coalescing()."""
        st = _lib.CosCoalescingStats()
        check(_lib.lib().cos_index_coalescing_stats(self._h, C.byref(st)))
        return {f: int(getattr(st, f)) for f, _ in st._fields_}

    def set_visited_mode(self, mode: int):
        check(_lib.lib().cos_index_set_visited_mode(self._h, mode))

    LATENCY_MODE_DEFAULT_MAX_B = 2048  # COS_LATENCY_MODE_DEFAULT_MAX_B (include/cosdata_hip.h)

    def set_latency_mode(self, max_queries: int):
        """Launches of at most `max_queries` queries take the latency variant of the walk (0 = never); same results."""
        check(_lib.lib().cos_index_set_latency_mode(self._h, max_queries))

    LATENCY_WAVES_DEFAULT_MAX_B = 512  # COS_LATENCY_WAVES_DEFAULT_MAX_B (include/cosdata_hip.h)

    def set_latency_waves(self, max_queries: int):
        """Launches of at most `max_queries` queries give every query four waves (kernels_walk_lat4.hip; 0 = never); same results."""
        check(_lib.lib().cos_index_set_latency_waves(self._h, max_queries))

    WALK_ORDER_DEFAULT_MIN_B = 8192  # COS_WALK_ORDER_DEFAULT_MIN_B (include/cosdata_hip.h)

    def set_walk_order(self, min_queries: int):
        """Launches of at least `min_queries` queries (0 = never) walk the levels down to the key level (walk_order_cuts) in arrival
        order, are then sorted by the depth-first position of the best node found there and walk every level below it with the
        sorted queries dealt to the XCDs in contiguous runs (kernels_order.hip); same results."""
        check(_lib.lib().cos_index_set_walk_order(self._h, min_queries))

    def walk_order_cuts(self):
        """Levels after which a locality-ordered launch is cut and re-sorted (descending); [] if the graph has none it can use."""
        lv = (C.c_uint32 * 16)()
        n = C.c_uint32()
        check(_lib.lib().cos_index_walk_order_cuts(self._h, lv, 16, C.byref(n)))
        return [int(lv[i]) for i in range(min(n.value, 16))]

    WALK_TABLE_DEFAULT_MIN_B = 1         # COS_WALK_TABLE_DEFAULT_MIN_B (include/cosdata_hip.h): every launch
    WALK_TABLE_DEFAULT_MAX_COLS = 8192   # COS_WALK_TABLE_DEFAULT_MAX_COLS
    WALK_TABLE_AUTO = 0xFFFFFFFF         # COS_WALK_TABLE_AUTO: levels of at most c x ef_search x neighbors_count nodes (c = 8 up to ef 64, 6 above)

    def set_walk_table(self, max_cols: int = WALK_TABLE_AUTO, min_queries: int = WALK_TABLE_DEFAULT_MIN_B):
        """Launches of at least `min_queries` queries over u8 codes precompute similarity(query, node) for every node of the top
        levels (one i8 MFMA GEMM) and walk those levels from the table; same results.  max_cols: WALK_TABLE_AUTO (levels of at most
        c x ef_search x neighbors_count nodes each, c = 8 up to ef_search 64 and 6 above: include/cosdata_hip.h) or a cap on the table levels' nodes together.  0 for either = never."""
        check(_lib.lib().cos_index_set_walk_table(self._h, max_cols, min_queries))

    def walk_table_info(self):
        """(lowest level, columns) of the level table the next big launch would use; (0, 0) = none."""
        lv, cols = C.c_uint32(), C.c_uint32()
        check(_lib.lib().cos_index_walk_table_info(self._h, C.byref(lv), C.byref(cols)))
        return int(lv.value), int(cols.value)

    def last_walk_split(self, stream: int = 0) -> CosWalkSplit:
        """The last batch on `stream` split by dispatch (roofline report): level-table GEMM, levels above / below the cut."""
        st = CosWalkSplit()
        st.struct_size = C.sizeof(CosWalkSplit)
        check(_lib.lib().cos_index_last_walk_split(self._h, C.c_void_p(stream), C.byref(st)))
        return st

    def batch_search(self, queries, top_k: int, return_status: bool = False):
        """IndexOps::batch_search: [B][dim] raw f32 -> (ids [B][k], scores [B][k], counts [B]).
        Raises CosdataError (status 2 = CalculationError) if any query fails, like the
        reference's collect::<Result<_>>; return_status=True returns per-query statuses instead."""
        q = self._queries(queries)
        B = q.shape[0]
        ids = np.full((B, top_k), 0xFFFFFFFF, np.uint32)
        scores = np.zeros((B, top_k), np.float32)
        counts = np.zeros(B, np.uint32)
        status = np.zeros(B, np.int32)
        rc = _lib.lib().cos_search_batch(self._h, _p(q), B, top_k, _p(ids), _p(scores), _p(counts), _p(status))
        if return_status:
            return ids, scores, counts, rc, status
        check(rc)
        return ids, scores, counts

    def search_internal(self, query, top_k: int):
        """HNSWIndex::search_internal for one query -> list of (internal_id, score)."""
        ids, scores, counts = self.batch_search(np.asarray(query, np.float32)[None, :], top_k)
        return [(int(ids[0, i]), float(scores[0, i])) for i in range(int(counts[0]))]

    def batch_search_device(self, q_ptr: int, B: int, top_k: int, out_ids_ptr: int, out_scores_ptr: int, out_counts_ptr: int,
                            out_status_ptr: int, stream: int = 0):
        """Device-pointer variant (inputs resident in HBM, enqueued on `stream`, no sync)."""
        check(_lib.lib().cos_search_batch_device(self._h, C.c_void_p(q_ptr), B, top_k, C.c_void_p(out_ids_ptr),
                                                 C.c_void_p(out_scores_ptr), C.c_void_p(out_counts_ptr),
                                                 C.c_void_p(out_status_ptr), C.c_void_p(stream)))

    def ann_search_batch(self, queries):
        """ann_search's per-level lists before finalisation: ids/sims [B][L+1][100], counts [B][L+1]."""
        q = self._queries(queries)
        B, L1 = q.shape[0], self.hnsw_params.num_layers + 1
        ids = np.zeros((B, L1, 100), np.uint32)
        sims = np.zeros((B, L1, 100), np.float32)
        counts = np.zeros((B, L1), np.uint32)
        status = np.zeros(B, np.int32)
        check(_lib.lib().cos_ann_search_batch(self._h, _p(q), B, _p(ids), _p(sims), _p(counts), _p(status)))
        return ids, sims, counts

    # ---- metadata-filtered search (pseudo-root component) ---------------------------------------
    def enable_metadata(self, mdim: int, max_replicas_per_node: int):
        check(_lib.lib().cos_index_enable_metadata(self._h, mdim, max_replicas_per_node))
        self.mdim = mdim
        return self

    def upload_meta_graph(self, node_ids, mbits, levels):
        """node table (ascending replica ids, mbits [n][mdim]) + the component's levels [(node_ids, nbr_ids), ...]"""
        ids, mb = _c(node_ids, np.uint32), _c(mbits, np.int32)
        assert mb.shape == (ids.size, self.mdim)
        check(_lib.lib().cos_index_upload_meta_nodes(self._h, ids.size, _p(ids), _p(mb)))
        for l, (lid, nbr) in enumerate(levels):
            lid, nbr = _c(lid, np.uint32), _c(nbr, np.uint32)
            assert nbr.shape == (lid.size, self.level_M(l))
            check(_lib.lib().cos_index_upload_meta_graph_level(self._h, l, lid.size, _p(lid), _p(nbr)))
        return self

    def build_meta(self, node_ids, mbits, max_levels, batch_size: int = 0):
        """the pseudo-root component built on the device (cos_index_build_meta): node table + the level every node was drawn"""
        ids, mb, ml = _c(node_ids, np.uint32), _c(mbits, np.int32), _c(max_levels, np.uint8)
        assert mb.shape == (ids.size, self.mdim) and ml.size == ids.size
        check(_lib.lib().cos_index_build_meta(self._h, ids.size, _p(ids), _p(mb), _p(ml), batch_size))
        return self

    def download_meta_graph(self):
        out = []
        for l in range(self.hnsw_params.num_layers + 1):
            n = C.c_uint32()
            check(_lib.lib().cos_index_meta_level_count(self._h, l, C.byref(n)))
            ids = np.zeros(n.value, np.uint32)
            nbr = np.zeros((n.value, self.level_M(l)), np.uint32)
            check(_lib.lib().cos_index_download_meta_graph_level(self._h, l, _p(ids), _p(nbr)))
            out.append((ids, nbr))
        return out

    def _filters(self, filter_offsets, filter_dims):
        off, fd = _c(filter_offsets, np.uint32), _c(np.atleast_2d(filter_dims), np.int8)
        if fd.shape[1] != self.mdim or off[-1] != fd.shape[0]:
            raise CosdataError(_lib.ERR_INVALID, f"filters must be [F][{self.mdim}] i8 with offsets ending at F")
        return off, fd

    def search_filtered(self, queries, filter_offsets, filter_dims, top_k: int, return_status: bool = False):
        """search_internal with a Filter: (replica ids [B][k], exact cosine scores, counts)"""
        q = self._queries(queries)
        off, fd = self._filters(filter_offsets, filter_dims)
        B = q.shape[0]
        ids = np.full((B, top_k), 0xFFFFFFFF, np.uint32)
        scores = np.zeros((B, top_k), np.float32)
        counts = np.zeros(B, np.uint32)
        status = np.zeros(B, np.int32)
        rc = _lib.lib().cos_search_filtered_batch(self._h, _p(q), B, _p(off), _p(fd), top_k, _p(ids), _p(scores), _p(counts), _p(status))
        if return_status:
            return ids, scores, counts, rc, status
        check(rc)
        return ids, scores, counts

    def ann_search_filtered(self, queries, filter_offsets, filter_dims):
        q = self._queries(queries)
        off, fd = self._filters(filter_offsets, filter_dims)
        B, L1 = q.shape[0], self.hnsw_params.num_layers + 1
        ids = np.zeros((B, L1, 100), np.uint32)
        sims = np.zeros((B, L1, 100), np.float32)
        counts = np.zeros((B, L1), np.uint32)
        status = np.zeros(B, np.int32)
        check(_lib.lib().cos_ann_search_filtered_batch(self._h, _p(q), B, _p(off), _p(fd), _p(ids), _p(sims), _p(counts), _p(status)))
        return ids, sims, counts

    def enable_timing(self, on: bool = True):
        check(_lib.lib().cos_index_enable_timing(self._h, 1 if on else 0))

    def last_stats(self, stream: int = 0) -> CosSearchStats:
        st = CosSearchStats()
        check(_lib.lib().cos_index_last_stats(self._h, C.c_void_p(stream), C.byref(st)))
        return st

    def timing_summary(self, stream: int) -> CosTimingSummary:
        """HIP-event times of the launches enqueued on `stream` since enable_timing(True)."""
        st = CosTimingSummary()
        check(_lib.lib().cos_index_timing_summary(self._h, C.c_void_p(stream), C.byref(st)))
        return st

    def flat_search(self, queries, top_k: int, with_stats: bool = False):
        """Exhaustive search over the quantized codes (i8 MFMA GEMM) + exact rerank of the best 5k."""
        q = self._queries(queries)
        B = q.shape[0]
        ids = np.full((B, top_k), 0xFFFFFFFF, np.uint32)
        scores = np.zeros((B, top_k), np.float32)
        counts = np.zeros(B, np.uint32)
        st = CosFlatStats()
        check(_lib.lib().cos_flat_search_batch(self._h, _p(q), B, top_k, _p(ids), _p(scores), _p(counts), C.byref(st)))
        return (ids, scores, counts, st) if with_stats else (ids, scores, counts)

    def bruteforce_topk(self, queries, k: int):
        q = self._queries(queries)
        ids = np.zeros((q.shape[0], k), np.uint32)
        scores = np.zeros((q.shape[0], k), np.float32)
        check(_lib.lib().cos_bruteforce_topk(self._h, _p(q), q.shape[0], k, _p(ids), _p(scores)))
        return ids, scores
