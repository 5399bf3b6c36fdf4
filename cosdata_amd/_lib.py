"""ctypes loader for libcosdata_hip.so (the C ABI of include/cosdata_hip.h).

The product path has no CPU fallback: if the shared library is missing, or no gfx950 device is
visible, calls fail loudly (CosdataError) instead of routing anywhere else.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_PKG, "libcosdata_hip.so")
CSRC = os.path.join(_PKG, "csrc")

OK, ERR_STORAGE_MISMATCH, ERR_CALCULATION, ERR_INVALID, ERR_UNIMPLEMENTED, ERR_HIP, ERR_NOT_READY, ERR_NO_DEVICE = range(8)
STATUS_NAMES = {
    0: "OK", 1: "StorageMismatch", 2: "CalculationError", 3: "Invalid", 4: "Unimplemented", 5: "HipError",
    6: "NotReady", 7: "NoDevice",
}
ABI_VERSION = 1


class CosdataError(RuntimeError):
    """Non-zero cos_status.  `.status` mirrors DistanceError / WaCustomError of the reference:
    1 = StorageMismatch (QuantizationMismatch), 2 = CalculationError."""

    def __init__(self, status: int, message: str):
        super().__init__(f"[{STATUS_NAMES.get(status, status)}] {message}")
        self.status = status


class CosParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("abi_version", C.c_uint32), ("dim", C.c_uint32), ("metric", C.c_uint32),
        ("storage", C.c_uint32), ("resolution", C.c_uint32), ("range_lo", C.c_float), ("range_hi", C.c_float),
        ("num_layers", C.c_uint32), ("neighbors_count", C.c_uint32), ("level0_neighbors_count", C.c_uint32),
        ("ef_construction", C.c_uint32), ("ef_search", C.c_uint32), ("shortlist_size", C.c_uint32),
        ("visited_mode", C.c_uint32), ("device", C.c_int32), ("id_base", C.c_uint32), ("reserved", C.c_uint32),
        ("seed", C.c_uint64),
    ]


class CosSearchStats(C.Structure):
    _fields_ = [
        ("evals", C.c_uint64), ("expansions", C.c_uint64), ("adj_bytes", C.c_uint64), ("rerank_rows", C.c_uint64),
        ("walk_ms", C.c_float), ("finalize_ms", C.c_float), ("prep_ms", C.c_float), ("reserved", C.c_uint32),
    ]


class CosWalkSplit(C.Structure):
    """cos_walk_split: the last batch of a stream split by dispatch (level-table GEMM | levels above the cut | levels below it)."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("queries", C.c_uint32), ("table_level_min", C.c_uint32), ("table_cols", C.c_uint32),
        ("cut_after_level", C.c_uint32), ("reserved", C.c_uint32),
        ("table_ms", C.c_float), ("upper_ms", C.c_float), ("sort_ms", C.c_float), ("lower_ms", C.c_float),
        ("table_int8_ops", C.c_double), ("table_evals", C.c_uint64),
        ("upper_evals", C.c_uint64), ("upper_expansions", C.c_uint64), ("upper_adj_bytes", C.c_uint64),
        ("lower_evals", C.c_uint64), ("lower_expansions", C.c_uint64), ("lower_adj_bytes", C.c_uint64),
    ]


class CosTimingSummary(C.Structure):
    _fields_ = [("launches", C.c_uint32), ("prep_ms_sum", C.c_float), ("walk_ms_sum", C.c_float), ("finalize_ms_sum", C.c_float),
                ("walk_ms_min", C.c_float), ("walk_ms_max", C.c_float)]


class CosCoalescingStats(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("queries", C.c_uint64), ("requests", C.c_uint64), ("closed_full", C.c_uint64),
                ("closed_quiet", C.c_uint64), ("closed_deadline", C.c_uint64), ("solo_calls", C.c_uint64)]


class CosSparseStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_float), ("blocks", C.c_uint32), ("postings_visited", C.c_uint64), ("posting_bytes", C.c_uint64)]


class CosFlatStats(C.Structure):
    _fields_ = [("gemm_ms", C.c_float), ("gemm_launches", C.c_uint32), ("int8_ops", C.c_double), ("code_bytes", C.c_double)]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into cosdata_amd/libcosdata_hip.so (in-tree)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_PKG, "..", "include", "cosdata_hip.h"))
    stale = force or not os.path.exists(SO_PATH) or any(os.path.getmtime(s) > os.path.getmtime(SO_PATH) for s in srcs)
    if force:  # every translation unit again (about a minute): the "does it build" check must not be satisfied by a shipped .so
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    if stale:
        cmd = ["make", "-C", CSRC, "-j", str(max(2, min(16, 2 * (os.cpu_count() or 1))))]
        if not verbose:
            cmd.append("-s")
        subprocess.check_call(cmd)
    return SO_PATH


_lib = None

# every symbol include/cosdata_hip.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "cos_index_create", "cos_index_destroy", "cos_last_error_string", "cos_device_count",
    "cos_index_upload_vectors", "cos_index_set_root", "cos_index_upload_graph_level", "cos_index_level_count",
    "cos_index_download_graph_level", "cos_index_download_codes", "cos_index_download_root", "cos_index_build", "cos_index_append", "cos_index_delete", "cos_index_restore_link_state", "cos_index_release_link_state", "cos_index_enable_metadata", "cos_index_upload_meta_nodes", "cos_index_upload_meta_graph_level", "cos_index_build_meta", "cos_index_meta_level_count", "cos_index_download_meta_graph_level", "cos_search_filtered_batch",
    "cos_ann_search_filtered_batch", "cos_index_load_reference_dir", "cos_reference_dir_level_counts", "cos_reference_dir_read_level",
    "cos_search_batch", "cos_search_batch_device", "cos_ann_search_batch", "cos_index_set_coalescing", "cos_index_coalescing_stats", "cos_index_set_ef_search",
    "cos_index_set_visited_mode", "cos_index_set_latency_mode", "cos_index_set_latency_waves", "cos_index_set_walk_order", "cos_index_walk_order_cuts", "cos_index_set_walk_table", "cos_index_walk_table_info", "cos_index_last_walk_split", "cos_index_enable_timing", "cos_index_last_stats", "cos_index_timing_summary", "cos_quantize_batch",
    "cos_code_bytes", "cos_sample_values_range", "cos_distance_batch", "cos_bruteforce_topk", "cos_flat_search_batch", "cos_bm25_create", "cos_bm25_destroy",
    "cos_bm25_search_batch", "cos_bm25_search_batch_device", "cos_rrf_fuse_batch", "cos_hybrid_search_batch", "cos_text_process", "cos_text_count_tokens", "cos_bm25_term_frequency", "cos_xxhash32", "cos_stem_english", "cos_sparse_create", "cos_sparse_build_csr", "cos_sparse_create_from_vectors", "cos_sparse_destroy", "cos_sparse_search_batch", "cos_sparse_last_stats", "cos_sparse_layout", "cos_merge_topk_device", "cos_merge_topk_packed_device", "cos_hbm_probe",
    "cos_shardset_unique_id", "cos_shardset_create", "cos_shardset_destroy", "cos_shardset_search_batch", "cos_shardset_exchange_device",
    "cos_tuning_set", "cos_tuning_clear", "cos_tuning_get",
]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise CosdataError(ERR_NO_DEVICE, f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                          "(the HIP extension is mandatory; there is no CPU path)")
    # The PyTorch wheel bundles its own copy of the HIP runtime.  Two runtimes can live in one process only when torch's is
    # initialised first (the other order leaves torch with "No HIP GPUs are available"), so a Python process that will also use
    # torch for device buffers / torch.distributed gets torch imported before this library.  C / C++ / Rust hosts are unaffected.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(SO_PATH)
    vp, i32, u32, f32 = C.c_void_p, C.c_int32, C.c_uint32, C.c_float
    sig = {
        "cos_index_create": [C.POINTER(CosParams), C.POINTER(vp)],
        "cos_index_destroy": [vp],
        "cos_device_count": [C.POINTER(i32)],
        "cos_index_upload_vectors": [vp, vp, u32, u32],
        "cos_index_set_root": [vp, vp],
        "cos_index_upload_graph_level": [vp, u32, u32, vp, vp],
        "cos_index_level_count": [vp, u32, C.POINTER(u32)],
        "cos_index_download_graph_level": [vp, u32, vp, vp],
        "cos_index_download_codes": [vp, vp, vp],
        "cos_index_download_root": [vp, vp],
        "cos_index_build": [vp, u32],
        "cos_index_append": [vp, vp, u32, u32, u32],
        "cos_index_release_link_state": [vp],
        "cos_index_restore_link_state": [vp],
        "cos_index_delete": [vp, vp, u32],
        "cos_index_load_reference_dir": [vp, C.c_char_p, u32, u32],
        "cos_index_enable_metadata": [vp, u32, u32],
        "cos_index_upload_meta_nodes": [vp, u32, vp, vp],
        "cos_index_upload_meta_graph_level": [vp, u32, u32, vp, vp],
        "cos_search_filtered_batch": [vp, vp, u32, vp, vp, u32, vp, vp, vp, vp],
        "cos_ann_search_filtered_batch": [vp, vp, u32, vp, vp, vp, vp, vp, vp],
        "cos_reference_dir_level_counts": [C.c_char_p, u32, u32, u32, vp],
        "cos_reference_dir_read_level": [C.c_char_p, u32, u32, u32, u32, vp, vp],
        "cos_search_batch": [vp, vp, u32, u32, vp, vp, vp, vp],
        "cos_search_batch_device": [vp, vp, u32, u32, vp, vp, vp, vp, vp],
        "cos_ann_search_batch": [vp, vp, u32, vp, vp, vp, vp],
        "cos_index_set_ef_search": [vp, u32],
        "cos_index_set_coalescing": [vp, u32, u32],
        "cos_index_coalescing_stats": [vp, C.POINTER(CosCoalescingStats)],
        "cos_index_set_visited_mode": [vp, u32],
        "cos_index_build_meta": [vp, u32, vp, vp, vp, u32],
        "cos_index_meta_level_count": [vp, u32, C.POINTER(u32)],
        "cos_index_download_meta_graph_level": [vp, u32, vp, vp],
        "cos_index_set_latency_mode": [vp, u32],
        "cos_index_set_latency_waves": [vp, u32],
        "cos_index_set_walk_order": [vp, u32],
        "cos_index_walk_order_cuts": [vp, C.POINTER(u32), u32, C.POINTER(u32)],
        "cos_index_enable_timing": [vp, i32],
        "cos_index_last_stats": [vp, vp, C.POINTER(CosSearchStats)],
        "cos_index_set_walk_table": [vp, u32, u32],
        "cos_index_walk_table_info": [vp, C.POINTER(u32), C.POINTER(u32)],
        "cos_index_last_walk_split": [vp, vp, C.POINTER(CosWalkSplit)],
        "cos_index_timing_summary": [vp, vp, C.POINTER(CosTimingSummary)],
        "cos_quantize_batch": [u32, u32, u32, f32, f32, vp, u32, vp, vp],
        "cos_sample_values_range": [vp, u32, u32, f32, C.POINTER(f32), C.POINTER(f32)],
        "cos_distance_batch": [u32, u32, u32, u32, vp, vp, u32, vp, vp, u32, vp, vp, u32, vp, vp],
        "cos_bruteforce_topk": [vp, vp, u32, u32, vp, vp],
        "cos_flat_search_batch": [vp, vp, u32, u32, vp, vp, vp, C.POINTER(CosFlatStats)],
        "cos_bm25_create": [i32, vp, vp, u32, vp, vp, u32, C.POINTER(vp)],
        "cos_bm25_destroy": [vp],
        "cos_bm25_search_batch": [vp, vp, vp, u32, u32, vp, vp, vp],
        "cos_bm25_search_batch_device": [vp, vp, vp, u32, u32, vp, vp, vp, vp],
        "cos_rrf_fuse_batch": [vp, vp, u32, vp, vp, u32, u32, f32, u32, vp, vp, vp],
        "cos_hybrid_search_batch": [vp, vp, vp, vp, vp, u32, u32, f32, vp, vp, vp],
        "cos_sparse_create": [i32, u32, f32, vp, u32, vp, vp, u32, vp, vp, vp, C.POINTER(vp)],
        "cos_sparse_build_csr": [u32, f32, u32, vp, vp, vp, vp, vp, vp, C.POINTER(u32)],
        "cos_sparse_create_from_vectors": [i32, u32, f32, u32, vp, vp, vp, i32, C.POINTER(vp)],
        "cos_sparse_destroy": [vp],
        "cos_sparse_search_batch": [vp, vp, vp, vp, u32, u32, f32, u32, vp, vp, vp],
        "cos_sparse_last_stats": [vp, vp],
        "cos_sparse_layout": [vp, C.POINTER(u32)],
        "cos_merge_topk_device": [vp, vp, vp, u32, u32, u32, vp, vp, vp, i32, vp],
        "cos_merge_topk_packed_device": [vp, u32, u32, u32, vp, vp, vp, i32, vp],
        "cos_hbm_probe": [i32, u32, C.c_uint64, u32, u32, C.POINTER(C.c_double)],
        "cos_shardset_unique_id": [vp],
        "cos_shardset_create": [vp, u32, u32, u32, vp, C.POINTER(vp)],
        "cos_shardset_destroy": [vp],
        "cos_shardset_search_batch": [vp, vp, u32, u32, vp, vp, vp],
        "cos_shardset_exchange_device": [vp, vp, u32, u32, vp, vp, vp, vp, vp],
        "cos_tuning_set": [C.c_char_p, C.c_int64],
        "cos_tuning_clear": [C.c_char_p],
        "cos_tuning_get": [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(i32)],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.restype = i32
        fn.argtypes = args
    L.cos_last_error_string.restype = C.c_char_p
    L.cos_last_error_string.argtypes = []
    L.cos_stem_english.restype = C.c_size_t
    L.cos_stem_english.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    L.cos_text_process.restype = i32
    L.cos_text_process.argtypes = [C.c_char_p, C.c_size_t, u32, f32, f32, f32, vp, vp, vp, vp, u32, C.POINTER(u32)]
    L.cos_text_count_tokens.restype = u32
    L.cos_text_count_tokens.argtypes = [C.c_char_p, C.c_size_t, u32]
    L.cos_bm25_term_frequency.restype = f32
    L.cos_bm25_term_frequency.argtypes = [u32, u32, f32, f32, f32]
    L.cos_xxhash32.restype = u32
    L.cos_xxhash32.argtypes = [C.c_char_p, C.c_size_t, u32]
    L.cos_code_bytes.restype = C.c_size_t
    L.cos_code_bytes.argtypes = [u32, u32, u32]
    _lib = L
    return L


def check(rc: int):
    if rc != OK:
        raise CosdataError(rc, lib().cos_last_error_string().decode("utf-8", "replace"))


def tuning_set(name: str, value: int):
    """cos_tuning_set: an experiment knob of the library (cosdata_amd/csrc/tuning.h); never changes a result."""
    check(lib().cos_tuning_set(name.encode(), int(value)))


def tuning_clear(name: str | None = None):
    """cos_tuning_clear: one knob (or, with None, every knob) back to its built-in default."""
    check(lib().cos_tuning_clear(name.encode() if name is not None else None))


def tuning_get(name: str):
    """-> the knob's value, or None when nobody set it."""
    v, isset = C.c_int64(), C.c_int32()
    check(lib().cos_tuning_get(name.encode(), C.byref(v), C.byref(isset)))
    return int(v.value) if isset.value else None


class tuning:
    """with tuning(flat_tile_kernel=1): ...  — knobs set for the block, restored afterwards."""

    def __init__(self, **knobs):
        self.knobs = knobs
        self.old = {}

    def __enter__(self):
        try:
            for k, v in self.knobs.items():
                old = tuning_get(k)          # (an unknown name raises here, before anything of it is recorded)
                tuning_set(k, v)
                self.old[k] = old
        except Exception:
            self.__exit__()                  # a rejected name must not leave the earlier knobs of the block set
            raise
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                tuning_clear(k)
            else:
                tuning_set(k, v)
        return False
