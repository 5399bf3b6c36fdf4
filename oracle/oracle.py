"""ctypes binding of the CPU oracle (oracle/libcosdata_oracle.so).

TEST INFRASTRUCTURE ONLY — see oracle/cosdata_oracle.h.  Importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never from cosdata_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcosdata_oracle.so")

METRIC_COSINE, METRIC_EUCLIDEAN, METRIC_HAMMING, METRIC_DOT = 0, 1, 2, 3
STORAGE_U8, STORAGE_SUBBYTE, STORAGE_F16, STORAGE_F32 = 0, 1, 2, 3
OK, ERR_STORAGE_MISMATCH, ERR_CALCULATION, ERR_INVALID, ERR_UNIMPLEMENTED = 0, 1, 2, 3, 4
ROOT_ID, QUERY_ID, SLOT_EMPTY = 0xFFFFFFFF, 0xFFFFFFFE, 0xFFFFFFFD
VISITED_REF, VISITED_EXACT = 0, 1


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class Params(C.Structure):
    _fields_ = [
        ("dim", C.c_uint32), ("metric", C.c_uint32), ("storage", C.c_uint32), ("resolution", C.c_uint32),
        ("range_lo", C.c_float), ("range_hi", C.c_float), ("num_layers", C.c_uint32),
        ("neighbors_count", C.c_uint32), ("level0_neighbors_count", C.c_uint32),
        ("ef_construction", C.c_uint32), ("ef_search", C.c_uint32), ("shortlist_size", C.c_uint32),
        ("visited_mode", C.c_uint32), ("seed", C.c_uint64),
    ]


class Stats(C.Structure):
    _fields_ = [("evals", C.c_uint64), ("expansions", C.c_uint64), ("adj_bytes", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    P = C.POINTER
    vp, f32p, u32p, u8p, u64p = C.c_void_p, P(C.c_float), P(C.c_uint32), P(C.c_uint8), P(C.c_uint64)
    sig = {
        "coso_code_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
        "coso_quantize": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, vp, f32p]),
        "coso_dot_u8": (C.c_uint64, [vp, vp, C.c_int]),
        "coso_dot_u8_scalar": (C.c_uint64, [vp, vp, C.c_int]),
        "coso_dot_f32": (C.c_float, [vp, vp, C.c_int]),
        "coso_dot_f32_scalar_order": (C.c_float, [vp, vp, C.c_int]),
        "coso_dot_f16": (C.c_float, [vp, vp, C.c_int]),
        "coso_dot_subbyte": (C.c_float, [vp, vp, C.c_int, C.c_int, P(C.c_int)]),
        "coso_dot_quaternary_scalar": (C.c_float, [vp, vp, C.c_int]),
        "coso_count_ones": (C.c_uint64, [vp, C.c_int]),
        "coso_seq_norm_f32": (C.c_float, [vp, C.c_int]),
        "coso_f32_to_f16": (C.c_uint16, [C.c_float]),
        "coso_f16_to_f32": (C.c_float, [C.c_uint16]),
        "coso_distance": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_float, vp, C.c_float, f32p]),
        "coso_metric_cmp": (C.c_int, [C.c_int, C.c_float, C.c_float]),
        "coso_level_probs": (None, [C.c_double, C.c_int, P(C.c_double), u8p]),
        "coso_max_insert_level": (C.c_int, [C.c_double, P(C.c_double), u8p, C.c_int]),
        "coso_sample_values_range": (None, [vp, C.c_uint64, C.c_float, f32p, f32p]),
        "coso_index_create": (vp, [P(Params)]),
        "coso_index_destroy": (None, [vp]),
        "coso_index_set_vectors": (C.c_int, [vp, vp, C.c_uint32]),
        "coso_index_alloc_vectors": (C.c_int, [vp, C.c_uint32]),
        "coso_index_quantize_rows": (C.c_int, [vp, C.c_uint32, vp, C.c_uint32]),
        "coso_index_set_raw_subset": (C.c_int, [vp, vp, vp, C.c_uint32]),
        "coso_candidates_batch": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32, vp, vp, C.c_int]),
        "coso_index_build": (C.c_int, [vp]),
        "coso_index_build_batched": (C.c_int, [vp, C.c_uint32]),
        "coso_index_build_rounds": (C.c_int, [vp, C.c_uint32, C.c_int, vp]),
        "coso_index_append_vectors": (C.c_int, [vp, vp, C.c_uint32]),
        "coso_index_build_rounds_continue": (C.c_int, [vp, C.c_uint32, vp]),
        "coso_index_delete": (C.c_int, [vp, C.c_uint32]),
        "coso_index_can_continue": (C.c_int, [vp]),
        "coso_index_restore_link_state": (C.c_int, [vp]),
        "coso_index_level_count": (C.c_uint32, [vp, C.c_uint32]),
        "coso_index_export_level": (C.c_int, [vp, C.c_uint32, vp, vp, vp]),
        "coso_index_import_level": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp, vp]),
        "coso_index_root_raw": (f32p, [vp]),
        "coso_index_set_root_raw": (C.c_int, [vp, vp]),
        "coso_index_codes": (vp, [vp]),
        "coso_index_mags": (f32p, [vp]),
        "coso_index_set_ef_search": (None, [vp, C.c_uint32]),
        "coso_index_set_visited_mode": (None, [vp, C.c_uint32]),
        "coso_search_batch": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp, vp, vp, C.c_int]),
        "coso_ann_search": (C.c_int, [vp, vp, vp, vp, vp]),
        "coso_flat_search_batch": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32, vp, vp, vp, C.c_int]),
        "coso_flat_candidates_batch": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32, vp, vp, C.c_int]),
        "coso_index_clear_graph": (None, [vp]),
        "coso_index_set_level0_neighbors": (C.c_int, [vp, C.c_uint32]),
        "coso_index_set_neighbors": (C.c_int, [vp, C.c_uint32]),
        "coso_bruteforce_topk": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp, C.c_uint32, C.c_uint32, vp, vp, C.c_int]),
        "coso_meta_enable": (C.c_int, [vp, C.c_uint32, C.c_uint32]),
        "coso_meta_set_nodes": (C.c_int, [vp, C.c_uint32, vp, vp]),
        "coso_meta_build": (C.c_int, [vp, vp]),
        "coso_meta_build_rounds": (C.c_int, [vp, vp, C.c_uint32, vp]),
        "coso_meta_level_count": (C.c_uint32, [vp, C.c_uint32]),
        "coso_meta_export_level": (C.c_int, [vp, C.c_uint32, vp, vp]),
        "coso_search_filtered_batch": (C.c_int, [vp, vp, C.c_uint32, vp, vp, C.c_uint32, vp, vp, vp, vp, C.c_int]),
        "coso_ann_search_filtered": (C.c_int, [vp, vp, vp, C.c_uint32, vp, vp, vp]),
        "coso_pseudo_level_probs": (None, [C.c_int, C.c_int, P(C.c_double), u8p]),
        "coso_sparse_quantize": (C.c_uint8, [C.c_float, C.c_float, C.c_int]),
        "coso_sparse_search": (C.c_int, [vp, C.c_uint32, vp, vp, C.c_uint32, C.c_int, C.c_float, C.c_float, vp, vp, C.c_uint32, C.c_uint32, vp, vp, C.c_uint32]),
        "coso_sparse_rerank": (C.c_int, [vp, vp, vp, vp, C.c_uint32, vp, vp, C.c_uint32, C.c_uint32, vp, vp]),
        "coso_bm25_idf": (C.c_float, [C.c_uint32, C.c_uint32]),
        "coso_bm25_tf": (C.c_float, [C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float]),
        "coso_bm25_search": (C.c_int, [vp, vp, C.c_uint32, vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, vp, vp]),
        "coso_rrf_fuse": (C.c_int, [vp, C.c_uint32, vp, C.c_uint32, C.c_float, C.c_uint32, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# ------------------------------------------------------------------------------------------------
# numeric kernels
# ------------------------------------------------------------------------------------------------
def code_bytes(storage, resolution, dim):
    return lib().coso_code_bytes(storage, resolution, dim)


def quantize(x, storage, resolution=0, lo=-1.0, hi=1.0):
    """ScalarQuantization::quantize for one vector -> (code bytes as uint8 array, mag)."""
    x = _c(x, np.float32)
    code = np.zeros(code_bytes(storage, resolution, x.size), np.uint8)
    mag = C.c_float()
    rc = lib().coso_quantize(_p(x), x.size, storage, resolution, lo, hi, _p(code), C.byref(mag))
    if rc != OK:
        raise ValueError(f"coso_quantize status {rc}")
    return code, np.float32(mag.value)


def quantize_batch(X, storage, resolution=0, lo=-1.0, hi=1.0):
    X = _c(X, np.float32)
    n, d = X.shape
    codes = np.zeros((n, code_bytes(storage, resolution, d)), np.uint8)
    mags = np.zeros(n, np.float32)
    for i in range(n):
        codes[i], mags[i] = quantize(X[i], storage, resolution, lo, hi)
    return codes, mags


def dot_u8(a, b, scalar=False):
    a, b = _c(a, np.uint8), _c(b, np.uint8)
    f = lib().coso_dot_u8_scalar if scalar else lib().coso_dot_u8
    return int(f(_p(a), _p(b), a.size))


def dot_f32(a, b, scalar_order=False):
    a, b = _c(a, np.float32), _c(b, np.float32)
    f = lib().coso_dot_f32_scalar_order if scalar_order else lib().coso_dot_f32
    return np.float32(f(_p(a), _p(b), a.size))


def dot_f16(a_bits, b_bits):
    a, b = _c(a_bits, np.uint16), _c(b_bits, np.uint16)
    return np.float32(lib().coso_dot_f16(_p(a), _p(b), a.size))


def dot_subbyte(x_planes, y_planes, resolution):
    """x_planes: uint8 [resolution, plane_bytes] plane-major (plane 0 first, as stored)."""
    x, y = _c(x_planes, np.uint8), _c(y_planes, np.uint8)
    st = C.c_int()
    v = lib().coso_dot_subbyte(_p(x), _p(y), resolution, x.size // max(resolution, 1), C.byref(st))
    return np.float32(v), st.value


def dot_quaternary_scalar(x_planes, y_planes):
    x, y = _c(x_planes, np.uint8), _c(y_planes, np.uint8)
    return np.float32(lib().coso_dot_quaternary_scalar(_p(x), _p(y), x.size // 2))


def count_ones(buf):
    b = _c(buf, np.uint8)
    return int(lib().coso_count_ones(_p(b), b.size))


def seq_norm(x):
    x = _c(x, np.float32)
    return np.float32(lib().coso_seq_norm_f32(_p(x), x.size))


def distance(metric, storage, resolution, dim, x_code, x_mag, y_code, y_mag):
    """DistanceMetric::calculate -> (status, value)."""
    x, y = _c(x_code, np.uint8), _c(y_code, np.uint8)
    out = C.c_float()
    rc = lib().coso_distance(metric, storage, resolution, dim, _p(x), float(x_mag), _p(y), float(y_mag), C.byref(out))
    return rc, np.float32(out.value)


def metric_cmp(metric, a, b):
    return lib().coso_metric_cmp(metric, float(a), float(b))


def sample_values_range(sample, clamp_margin_percent=1.0):
    x = _c(sample, np.float32)
    lo, hi = C.c_float(), C.c_float()
    lib().coso_sample_values_range(_p(x), x.size, clamp_margin_percent, C.byref(lo), C.byref(hi))
    return (lo.value, hi.value)


def level_probs(x, num_levels):
    v = (C.c_double * (num_levels + 1))()
    l = (C.c_uint8 * (num_levels + 1))()
    lib().coso_level_probs(x, num_levels, v, l)
    return [(v[i], l[i]) for i in range(num_levels + 1)]


def max_insert_level(x, probs):
    n = len(probs)
    v = (C.c_double * n)(*[p[0] for p in probs])
    l = (C.c_uint8 * n)(*[p[1] for p in probs])
    return lib().coso_max_insert_level(x, v, l, n)


# ------------------------------------------------------------------------------------------------
# HNSW index
# ------------------------------------------------------------------------------------------------
@dataclass
class HNSWParams:
    dim: int
    metric: int = METRIC_COSINE
    storage: int = STORAGE_U8
    resolution: int = 0
    range_lo: float = -1.0
    range_hi: float = 1.0
    num_layers: int = 9            # config.toml:24
    neighbors_count: int = 32      # config.toml:20
    level0_neighbors_count: int = 64
    ef_construction: int = 128
    ef_search: int = 256
    shortlist_size: int = 64       # config.toml:32
    visited_mode: int = VISITED_REF
    seed: int = 42

    def to_c(self) -> Params:
        return Params(self.dim, self.metric, self.storage, self.resolution, self.range_lo, self.range_hi, self.num_layers,
                      self.neighbors_count, self.level0_neighbors_count, self.ef_construction, self.ef_search,
                      self.shortlist_size, self.visited_mode, self.seed)


class OracleIndex:
    def __init__(self, params: HNSWParams):
        self.params = params
        cp = params.to_c()
        self._h = lib().coso_index_create(C.byref(cp))
        if not self._h:
            raise ValueError("coso_index_create failed (bad params)")
        self._raw = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().coso_index_destroy(self._h)
                self._h = None
        except Exception:  # interpreter shutdown
            pass

    def set_vectors(self, raw):
        self._raw = _c(raw, np.float32)
        rc = lib().coso_index_set_vectors(self._h, _p(self._raw), self._raw.shape[0])
        if rc != OK:
            raise ValueError(f"set_vectors status {rc}")
        return self

    # ---- corpora without a host copy of the raw table (bench.py --workload c4shard) ----
    def alloc_vectors(self, n):
        self._raw = None
        self._n = int(n)
        rc = lib().coso_index_alloc_vectors(self._h, self._n)
        if rc != OK:
            raise ValueError(f"alloc_vectors status {rc}")
        return self

    def quantize_rows(self, start, raw_chunk):
        x = _c(raw_chunk, np.float32)
        rc = lib().coso_index_quantize_rows(self._h, int(start), _p(x), x.shape[0])
        if rc != OK:
            raise ValueError(f"quantize_rows status {rc}")

    def set_raw_subset(self, ids_sorted, rows):
        self._sub = (_c(ids_sorted, np.uint32), _c(rows, np.float32))  # borrowed by the C side: keep alive
        rc = lib().coso_index_set_raw_subset(self._h, _p(self._sub[0]), _p(self._sub[1]), self._sub[0].size)
        if rc != OK:
            raise ValueError(f"set_raw_subset status {rc}")

    def candidates_batch(self, queries, top_k, threads=1):
        """ids [B][5k] (+ counts [B]) of the candidates whose raw rows the exact rerank reads"""
        q = _c(queries, np.float32)
        B = q.shape[0]
        ids = np.full((B, 5 * top_k), 0xFFFFFFFF, np.uint32)
        counts = np.zeros(B, np.uint32)
        rc = lib().coso_candidates_batch(self._h, _p(q), B, top_k, _p(ids), _p(counts), threads)
        if rc != OK:
            raise ValueError(f"candidates status {rc}")
        return ids, counts

    def build(self):
        rc = lib().coso_index_build(self._h)
        if rc != OK:
            raise ValueError(f"build status {rc}")
        return self

    def build_batched(self, batch_size=0):
        rc = lib().coso_index_build_batched(self._h, batch_size)
        if rc != OK:
            raise ValueError(f"build_batched status {rc}")
        return self

    def build_rounds(self, batch_size=0, greedy=False):
        """prototype: round-synchronous link schedule; returns (self, stats dict)"""
        st = np.zeros(4, np.uint64)
        rc = lib().coso_index_build_rounds(self._h, batch_size, 1 if greedy else 0, st.ctypes.data_as(C.c_void_p))
        if rc != OK:
            raise ValueError(f"build_rounds status {rc}")
        return self, {"rounds": int(st[0]), "batch_levels": int(st[1]), "nodes": int(st[2]), "first_round_nodes": int(st[3])}

    def restore_link_state(self):
        """an imported graph becomes appendable / deletable: similarities recomputed, lowest caches by the reference's reload rule"""
        rc = lib().coso_index_restore_link_state(self._h)
        if rc != OK:
            raise ValueError(f"restore_link_state status {rc}")
        return self

    def delete(self, ids):
        """delete_embedding for every id of `ids`, one after the other in the given order"""
        for i in np.atleast_1d(np.asarray(ids, np.uint32)):
            rc = lib().coso_index_delete(self._h, int(i))
            if rc != OK:
                raise ValueError(f"delete status {rc} (id {int(i)})")
        return self

    def append(self, raw_new, batch_size=0):
        """index_embeddings on a live index: `raw_new` [m][dim] take the ids [n, n + m) and are inserted by the rounds schedule continued
        (needs a graph built by build_rounds on this handle); returns (self, stats dict)"""
        x = _c(raw_new, np.float32)
        if self._raw is None:
            raise ValueError("append needs the raw table (set_vectors)")
        if not lib().coso_index_can_continue(self._h):     # (checked BEFORE the tables grow, like cos_index_append)
            raise ValueError("append needs a graph built by build_rounds on this handle, or restore_link_state")
        self._raw = np.ascontiguousarray(np.concatenate([self._raw, x.reshape(-1, self._raw.shape[1])]))   # the oracle borrows the WHOLE table
        rc = lib().coso_index_append_vectors(self._h, _p(self._raw), x.shape[0])
        if rc != OK:
            raise ValueError(f"append_vectors status {rc}")
        st = np.zeros(4, np.uint64)
        rc = lib().coso_index_build_rounds_continue(self._h, batch_size, st.ctypes.data_as(C.c_void_p))
        if rc != OK:
            raise ValueError(f"build_rounds_continue status {rc}")
        return self, {"rounds": int(st[0]), "batch_levels": int(st[1]), "nodes": int(st[2]), "first_round_nodes": int(st[3])}

    @property
    def n(self):
        return getattr(self, "_n", 0) if self._raw is None else self._raw.shape[0]

    def level_M(self, level):
        return self.params.level0_neighbors_count if level == 0 else self.params.neighbors_count

    def export_level(self, level, with_sims=False):
        n = lib().coso_index_level_count(self._h, level)
        M = self.level_M(level)
        ids = np.zeros(n, np.uint32)
        nbr = np.zeros((n, M), np.uint32)
        sims = np.zeros((n, M), np.float32) if with_sims else None
        rc = lib().coso_index_export_level(self._h, level, _p(ids), _p(nbr), _p(sims) if with_sims else None)
        if rc != OK:
            raise ValueError(f"export status {rc}")
        return (ids, nbr, sims) if with_sims else (ids, nbr)

    def export_graph(self):
        return [self.export_level(l) for l in range(self.params.num_layers + 1)]

    def import_level(self, level, node_ids, nbr_ids):
        ids, nbr = _c(node_ids, np.uint32), _c(nbr_ids, np.uint32)
        rc = lib().coso_index_import_level(self._h, level, ids.size, _p(ids), _p(nbr))
        if rc != OK:
            raise ValueError(f"import status {rc} (level {level})")

    def import_graph(self, levels, root_raw):
        lib().coso_index_clear_graph(self._h)          # a second graph over the same vectors replaces the first
        self.set_root_raw(root_raw)
        for l, (ids, nbr) in enumerate(levels):
            self.import_level(l, ids, nbr)
        return self

    def root_raw(self):
        p = lib().coso_index_root_raw(self._h)
        return np.ctypeslib.as_array(p, shape=(self.params.dim,)).copy()

    def set_root_raw(self, root):
        r = _c(root, np.float32)
        rc = lib().coso_index_set_root_raw(self._h, _p(r))
        if rc != OK:
            raise ValueError(f"set_root_raw status {rc}")

    def codes(self):
        cb = code_bytes(self.params.storage, self.params.resolution, self.params.dim)
        p = C.cast(lib().coso_index_codes(self._h), C.POINTER(C.c_uint8))
        return np.ctypeslib.as_array(p, shape=(self.n + 1, cb)).copy()

    def mags(self):
        return np.ctypeslib.as_array(lib().coso_index_mags(self._h), shape=(self.n + 1,)).copy()

    def set_level0_neighbors(self, m0):
        """level_0_neighbors_count of the NEXT imported / built graph (drops the current one, keeps the quantized vectors)"""
        rc = lib().coso_index_set_level0_neighbors(self._h, int(m0))
        if rc != OK:
            raise ValueError(f"set_level0_neighbors status {rc}")
        self.params.level0_neighbors_count = int(m0)
        return self

    def set_neighbors(self, m):
        """neighbors_count of the NEXT imported / built graph (drops the current one, keeps the quantized vectors)"""
        rc = lib().coso_index_set_neighbors(self._h, int(m))
        if rc != OK:
            raise ValueError(f"set_neighbors status {rc}")
        self.params.neighbors_count = int(m)
        return self

    def set_ef_search(self, ef):
        self.params.ef_search = ef
        lib().coso_index_set_ef_search(self._h, ef)

    def set_visited_mode(self, mode):
        self.params.visited_mode = mode
        lib().coso_index_set_visited_mode(self._h, mode)

    def search_batch(self, queries, top_k, threads=1, with_stats=False, raise_on_error=True):
        q = _c(queries, np.float32)
        B = q.shape[0]
        ids = np.full((B, top_k), 0xFFFFFFFF, np.uint32)
        scores = np.zeros((B, top_k), np.float32)
        counts = np.zeros(B, np.uint32)
        status = np.zeros(B, np.int32)
        stats = (Stats * B)()
        rc = lib().coso_search_batch(self._h, _p(q), B, top_k, _p(ids), _p(scores), _p(counts), _p(status),
                                     C.cast(stats, C.c_void_p), threads)
        if rc != OK and raise_on_error:
            raise ValueError(f"search status {rc}")
        out = [ids, scores, counts]
        if not raise_on_error:
            out += [rc, status]
        if with_stats:
            out.append(np.array([(s.evals, s.expansions, s.adj_bytes) for s in stats], dtype=np.uint64))
        return tuple(out)

    def flat_search_batch(self, queries, top_k, threads=1):
        q = _c(queries, np.float32)
        B = q.shape[0]
        ids = np.full((B, top_k), 0xFFFFFFFF, np.uint32)
        scores = np.zeros((B, top_k), np.float32)
        counts = np.zeros(B, np.uint32)
        rc = lib().coso_flat_search_batch(self._h, _p(q), B, top_k, _p(ids), _p(scores), _p(counts), threads)
        if rc != OK:
            raise ValueError(f"flat search status {rc}")
        return ids, scores, counts

    def flat_candidates_batch(self, queries, top_k, threads=1):
        """ids [B][5k] (+ counts [B]) of the flat search's rerank candidates (streamed corpora: fetch those raw rows first)"""
        q = _c(queries, np.float32)
        B = q.shape[0]
        ids = np.full((B, 5 * top_k), 0xFFFFFFFF, np.uint32)
        counts = np.zeros(B, np.uint32)
        rc = lib().coso_flat_candidates_batch(self._h, _p(q), B, top_k, _p(ids), _p(counts), threads)
        if rc != OK:
            raise ValueError(f"flat candidates status {rc}")
        return ids, counts

    # ---- metadata-filtered search (f4a) ----
    def meta_enable(self, mdim, max_replicas):
        self.mdim, self.replicas = int(mdim), int(max_replicas)
        rc = lib().coso_meta_enable(self._h, self.mdim, self.replicas)
        if rc != OK:
            raise ValueError(f"meta_enable status {rc}")
        return self

    def meta_set_nodes(self, ids_sorted, mbits):
        ids, mb = _c(ids_sorted, np.uint32), _c(mbits, np.int32)
        assert mb.shape == (ids.size, self.mdim)
        rc = lib().coso_meta_set_nodes(self._h, ids.size, _p(ids), _p(mb))
        if rc != OK:
            raise ValueError(f"meta_set_nodes status {rc}")
        return self

    def meta_build(self, max_levels):
        ml = _c(max_levels, np.uint8)
        rc = lib().coso_meta_build(self._h, _p(ml))
        if rc != OK:
            raise ValueError(f"meta_build status {rc}")
        return self

    def meta_build_rounds(self, max_levels, batch_size):
        """the pseudo-root component in the batch-synchronous schedule of a device-side builder (batch_size = 1 == meta_build);
        returns (self, {rounds, level_batches, node_levels})"""
        ml = _c(max_levels, np.uint8)
        st = np.zeros(3, np.uint64)
        rc = lib().coso_meta_build_rounds(self._h, _p(ml), batch_size, _p(st))
        if rc != OK:
            raise ValueError(f"meta_build_rounds status {rc}")
        return self, {"rounds": int(st[0]), "level_batches": int(st[1]), "node_levels": int(st[2])}

    def meta_export_graph(self):
        out = []
        for l in range(self.params.num_layers + 1):
            n = lib().coso_meta_level_count(self._h, l)
            ids = np.zeros(n, np.uint32)
            nbr = np.zeros((n, self.level_M(l)), np.uint32)
            rc = lib().coso_meta_export_level(self._h, l, _p(ids), _p(nbr))
            if rc != OK:
                raise ValueError(f"meta export status {rc}")
            out.append((ids, nbr))
        return out

    def search_filtered_batch(self, queries, filter_off, filter_dims, top_k, threads=1, raise_on_error=True):
        q = _c(queries, np.float32)
        B = q.shape[0]
        off, fd = _c(filter_off, np.uint32), _c(filter_dims, np.int32)
        ids = np.full((B, top_k), 0xFFFFFFFF, np.uint32)
        scores = np.zeros((B, top_k), np.float32)
        counts = np.zeros(B, np.uint32)
        status = np.zeros(B, np.int32)
        rc = lib().coso_search_filtered_batch(self._h, _p(q), B, _p(off), _p(fd), top_k, _p(ids), _p(scores), _p(counts), _p(status), threads)
        if rc != OK and raise_on_error:
            raise ValueError(f"filtered search status {rc}")
        return (ids, scores, counts) if raise_on_error else (ids, scores, counts, rc, status)

    def ann_search_filtered(self, query, filter_dims):
        q, fd = _c(query, np.float32), _c(np.atleast_2d(filter_dims), np.int32)
        cap = (self.params.num_layers + 1) * 100
        ids = np.zeros(cap, np.uint32)
        sims = np.zeros(cap, np.float32)
        lc = np.zeros(self.params.num_layers + 1, np.uint32)
        n = lib().coso_ann_search_filtered(self._h, _p(q), _p(fd), fd.shape[0], _p(ids), _p(sims), _p(lc))
        if n < 0:
            raise ValueError(f"ann_search_filtered status {-n}")
        return ids[:n], sims[:n], lc

    def ann_search(self, query):
        """ann_search output before finalisation: (ids, sims, per-level counts top level first)."""
        q = _c(query, np.float32)
        cap = (self.params.num_layers + 1) * 100
        ids = np.zeros(cap, np.uint32)
        sims = np.zeros(cap, np.float32)
        lc = np.zeros(self.params.num_layers + 1, np.uint32)
        n = lib().coso_ann_search(self._h, _p(q), _p(ids), _p(sims), _p(lc))
        if n < 0:
            raise ValueError(f"ann_search status {-n}")
        return ids[:n], sims[:n], lc


def bruteforce_topk(raw, queries, k, threads=1):
    raw, q = _c(raw, np.float32), _c(queries, np.float32)
    ids = np.zeros((q.shape[0], k), np.uint32)
    scores = np.zeros((q.shape[0], k), np.float32)
    rc = lib().coso_bruteforce_topk(_p(raw), raw.shape[0], raw.shape[1], _p(q), q.shape[0], k, _p(ids), _p(scores), threads)
    if rc != OK:
        raise ValueError(f"bruteforce status {rc}")
    return ids, scores


def pseudo_level_probs(num_levels, num_pseudo_nodes):
    v = (C.c_double * (num_levels + 1))()
    l = (C.c_uint8 * (num_levels + 1))()
    lib().coso_pseudo_level_probs(num_levels, num_pseudo_nodes, v, l)
    return [(v[i], l[i]) for i in range(num_levels + 1)]


def schema_level_probs(num_layers, num_pseudo_incl_root, factor=4.0):
    """levels_prob of an index whose collection has a metadata schema (api_service.rs:113-131): the layers pseudo_level_probs
    gives to the pseudo nodes get probability 1.0 (x >= 1.0 never holds: no replica lands there), the layers below use
    generate_level_probs(4.0, lower)"""
    plp = pseudo_level_probs(num_layers, num_pseudo_incl_root)
    lower = sum(1 for p, _ in plp if p == 0.0) - 1
    higher = num_layers - lower
    return [(1.0, num_layers - i) for i in range(higher)] + level_probs(factor, lower)


# ------------------------------------------------------------------------------------------------
# BM25 + RRF
# ------------------------------------------------------------------------------------------------
def bm25_idf(n_docs, containing):
    return np.float32(lib().coso_bm25_idf(n_docs, containing))


def bm25_tf(count, doc_len, avg_len, k1, b):
    return np.float32(lib().coso_bm25_tf(count, doc_len, avg_len, k1, b))


def bm25_search(term_hashes, offsets, doc_ids, tfs, n_docs, query_terms, top_k):
    th, off = _c(term_hashes, np.uint32), _c(offsets, np.uint64)
    di, tf, qt = _c(doc_ids, np.uint32), _c(tfs, np.float32), _c(query_terms, np.uint32)
    ids = np.zeros(max(top_k, 1), np.uint32)
    sc = np.zeros(max(top_k, 1), np.float32)
    m = lib().coso_bm25_search(_p(th), _p(off), th.size, _p(di), _p(tf), n_docs, _p(qt), qt.size, top_k, _p(ids), _p(sc))
    return ids[:m], sc[:m]


def rrf_fuse(dense_ids, sparse_ids, k_rrf, top_k):
    d, s = _c(dense_ids, np.uint32), _c(sparse_ids, np.uint32)
    ids = np.zeros(max(top_k, 1), np.uint32)
    sc = np.zeros(max(top_k, 1), np.float32)
    m = lib().coso_rrf_fuse(_p(d), d.size, _p(s), s.size, k_rrf, top_k, _p(ids), _p(sc))
    return ids[:m], sc[:m]


# ------------------------------------------------------------------------------------------------
# learned-sparse inverted index (f4b)
# ------------------------------------------------------------------------------------------------
def sparse_quantize(value, upper, bits):
    return int(lib().coso_sparse_quantize(float(value), float(upper), int(bits)))


def sparse_search(dims, key_off, vec_ids, n_vectors, bits, upper, early_terminate_threshold, q_dims, q_vals, k_with_reranking=0):
    """sequential_search: (ids, similarities) by similarity descending (larger id first); k_with_reranking = 0 -> all touched"""
    d, ko, vi = _c(dims, np.uint32), _c(key_off, np.uint64), _c(vec_ids, np.uint32)
    qd, qv = _c(q_dims, np.uint32), _c(q_vals, np.float32)
    cap = n_vectors if not k_with_reranking else k_with_reranking
    ids = np.zeros(max(cap, 1), np.uint32)
    sims = np.zeros(max(cap, 1), np.uint32)
    m = lib().coso_sparse_search(_p(d), d.size, _p(ko), _p(vi), n_vectors, bits, upper, early_terminate_threshold, _p(qd), _p(qv), qd.size,
                                 k_with_reranking, _p(ids), _p(sims), cap)
    if m < 0:
        raise ValueError(f"sparse_search status {-m}")
    return ids[:m], sims[:m]


def sparse_rerank(row_off, raw_dims, raw_vals, cand_ids, q_dims, q_vals, top_k=0):
    ro, rd, rv = _c(row_off, np.uint64), _c(raw_dims, np.uint32), _c(raw_vals, np.float32)
    ci, qd, qv = _c(cand_ids, np.uint32), _c(q_dims, np.uint32), _c(q_vals, np.float32)
    ids = np.zeros(max(ci.size, 1), np.uint32)
    sc = np.zeros(max(ci.size, 1), np.float32)
    m = lib().coso_sparse_rerank(_p(ro), _p(rd), _p(rv), _p(ci), ci.size, _p(qd), _p(qv), qd.size, top_k, _p(ids), _p(sc))
    return ids[:m], sc[:m]
