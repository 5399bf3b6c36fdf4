"""MFMA flat scan (cos_bruteforce_topk): the ids/scores must equal the oracle's exact brute force bit-for-bit
(the GEMM only generates candidates; the reference-order kernel produces the answer)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,dim,B,k", [(3000, 96, 37, 10), (20000, 768, 130, 10), (5001, 100, 5, 32), (700, 20, 300, 1), (70000, 96, 37, 10),
                                       (150001, 128, 300, 32)])
def test_bruteforce_matches_oracle(n, dim, B, k):
    import cosdata_amd as ca
    X = H.clustered_corpus(n, dim, n_centers=20, seed=3)
    Q = H.queries_from(X, B, noise=0.05, seed=8)
    ix = ca.HNSWIndex(dim, ca.HNSWHyperParams(num_layers=3), storage_type=ca.StorageType.UnsignedByte())
    ix.upload_vectors(X)
    ids, sc = ix.bruteforce_topk(Q, k)
    oids, osc = O.bruteforce_topk(X, Q, k, threads=4)
    assert np.array_equal(ids, oids)
    assert np.array_equal(sc.view(np.uint32), osc.view(np.uint32))


def test_bruteforce_transpose_detecting():
    """asymmetric inputs: a swapped MFMA C/D mapping or operand transpose cannot pass (cdna guide G9)."""
    import cosdata_amd as ca
    n, dim = 512, 64
    X = np.zeros((n, dim), np.float32)
    for i in range(n):
        X[i, i % dim] = 1.0
        X[i, (i * 7 + 3) % dim] += 0.25 + (i // dim) * 0.01
    Q = np.zeros((200, dim), np.float32)
    for b in range(200):
        Q[b, (b * 5) % dim] = 1.0
        Q[b, (b * 11 + 1) % dim] = 0.1 + 0.001 * b
    ix = ca.HNSWIndex(dim, ca.HNSWHyperParams(num_layers=3))
    ix.upload_vectors(X)
    ids, sc = ix.bruteforce_topk(Q, 8)
    oids, osc = O.bruteforce_topk(X, Q, 8, threads=2)
    assert np.array_equal(ids, oids) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))


@pytest.mark.parametrize("storage,res", [(O.STORAGE_U8, 0), (O.STORAGE_SUBBYTE, 2)])
@pytest.mark.parametrize("n,dim,B,k", [(4000, 96, 70, 10), (9000, 768, 300, 10), (3001, 100, 5, 12), (2500, 1000, 33, 3)])
def test_flat_code_scan_matches_oracle(storage, res, n, dim, B, k):
    """i8-MFMA exhaustive scan over the quantized codes + exact rerank == the oracle's flat search, bit for bit."""
    import cosdata_amd as ca
    X = H.uniform_corpus(n, dim, seed=13) * 0.9
    Q = np.concatenate([H.queries_from(X, B - 2, noise=0.05, seed=8), H.uniform_corpus(2, dim, seed=77)])
    ix = ca.HNSWIndex(dim, ca.HNSWHyperParams(num_layers=3), storage_type=ca.StorageType(ca.StorageKind(storage), res))
    ix.upload_vectors(X)
    ids, sc, cnt, st = ix.flat_search(Q, k, with_stats=True)
    oix = O.OracleIndex(O.HNSWParams(dim=dim, storage=storage, resolution=res, num_layers=3)).set_vectors(X)
    oids, osc, ocnt = oix.flat_search_batch(Q, k, threads=4)
    assert np.array_equal(cnt, ocnt)
    assert np.array_equal(ids, oids)
    assert np.array_equal(sc.view(np.uint32), osc.view(np.uint32))
    assert st.gemm_launches >= 1 and st.gemm_ms > 0


def test_flat_code_scan_zero_norm_is_calculation_error():
    import cosdata_amd as ca
    X = H.uniform_corpus(500, 64, seed=1)
    X[17] = -1.0  # all-zero u8 code -> zero norm
    ix = ca.HNSWIndex(64, ca.HNSWHyperParams(num_layers=3))
    ix.upload_vectors(X)
    with pytest.raises(ca.CosdataError) as ei:
        ix.flat_search(X[:4], 5)
    assert ei.value.status == 2


def test_flat_code_scan_zero_norm_query_and_the_cached_stored_flag():
    """round 6: the stored vectors' zero-norm flag is computed once per upload and the queries' flag comes back with the results (no
    device round trip before the first GEMM): a zero-norm QUERY is still a CalculationError for the call, a clean call on the same
    handle still answers, and a re-upload that introduces a zero-norm vector is seen"""
    import cosdata_amd as ca
    X = H.uniform_corpus(20000, 64, seed=3)
    ix = ca.HNSWIndex(64, ca.HNSWHyperParams(num_layers=3))
    ix.upload_vectors(X)
    Q = H.queries_from(X, 6, seed=1)
    ok = ix.flat_search(Q, 5)
    Qz = Q.copy()
    Qz[3] = -1.0                                               # all-zero u8 code -> |q| = 0
    with pytest.raises(ca.CosdataError) as ei:
        ix.flat_search(Qz, 5)
    assert ei.value.status == 2
    again = ix.flat_search(Q, 5)                               # the flag does not stick to the handle
    assert all(np.array_equal(a, b) for a, b in zip(ok, again))
    X2 = X.copy()
    X2[1234] = -1.0
    ix.upload_vectors(X2)                                      # a new upload: the cached stored flag must not survive it
    with pytest.raises(ca.CosdataError) as ei:
        ix.flat_search(Q, 5)
    assert ei.value.status == 2


@pytest.mark.parametrize("storage,res", [(O.STORAGE_U8, 0), (O.STORAGE_SUBBYTE, 2)])
@pytest.mark.parametrize("n,dim,B,k", [(70000, 96, 70, 10), (40000, 768, 260, 12), (30001, 384, 70, 10), (20003, 1024, 40, 10),
                                       (25000, 200, 300, 5), (21000, 960, 9, 10)])
def test_flat_code_scan_fused_epilogue_matches_oracle(storage, res, n, dim, B, k):
    """n above the 16384-candidate seed chunk: the remaining chunks run the threshold-filtered (fused) epilogue — survivors are
    appended per query instead of a [B][chunk] score matrix — and the answer must still be the oracle's, bit for bit."""
    import cosdata_amd as ca
    X = H.clustered_corpus(n, dim, n_centers=40, sigma=0.2, seed=23)
    Q = np.concatenate([H.queries_from(X, B - 2, noise=0.05, seed=8), H.uniform_corpus(2, dim, seed=77)])
    ix = ca.HNSWIndex(dim, ca.HNSWHyperParams(num_layers=3), storage_type=ca.StorageType(ca.StorageKind(storage), res))
    ix.upload_vectors(X)
    ids, sc, cnt, st = ix.flat_search(Q, k, with_stats=True)
    oix = O.OracleIndex(O.HNSWParams(dim=dim, storage=storage, resolution=res, num_layers=3)).set_vectors(X)
    oids, osc, ocnt = oix.flat_search_batch(Q, k, threads=8)
    assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))
    assert st.gemm_launches >= 2


@pytest.mark.parametrize("storage,res", [(O.STORAGE_U8, 0), (O.STORAGE_SUBBYTE, 2)])
def test_flat_code_scan_fused_overflow_falls_back_exactly(storage, res):
    """adversarial order: every later vector is closer to the query than all earlier ones, so everything beats the running
    threshold and the per-query append buffer overflows — the scan must notice and repeat on the unfused path (same answer)"""
    import cosdata_amd as ca
    n, dim = 60000, 64
    rng = np.random.default_rng(4)
    q = rng.standard_normal(dim).astype(np.float32)
    q /= np.linalg.norm(q)
    noise = rng.standard_normal((n, dim)).astype(np.float32)
    noise /= np.linalg.norm(noise, axis=1, keepdims=True)
    w = np.linspace(0.0, 1.0, n, dtype=np.float32)[:, None]            # later rows lean further towards q
    X = (w * q[None, :] + (1.0 - w) * noise).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    X *= 0.9
    Q = np.stack([q * 0.9, -q * 0.9, X[100], X[59000]]).astype(np.float32)
    ix = ca.HNSWIndex(dim, ca.HNSWHyperParams(num_layers=3), storage_type=ca.StorageType(ca.StorageKind(storage), res))
    ix.upload_vectors(X)
    ids, sc, cnt = ix.flat_search(Q, 10)
    oix = O.OracleIndex(O.HNSWParams(dim=dim, storage=storage, resolution=res, num_layers=3)).set_vectors(X)
    oids, osc, ocnt = oix.flat_search_batch(Q, 10, threads=8)
    assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))


def test_flat_code_scan_fp4_two_waves_per_simd_adversarial_order_and_ragged_batch():
    """flat_scan_q2_fp4_w8 (tuning knob flat_fp4_w8, 768-dim quaternary codes) where its staging overflows: vectors ordered so that every
    later one beats the running thresholds, a batch of 261 queries (a second, almost empty row block), n not a multiple of the 64-column
    tile — the same answer as the oracle and as the one-wave-per-SIMD kernel"""
    import cosdata_amd as ca
    from cosdata_amd import _lib
    n, dim = 40037, 768
    rng = np.random.default_rng(14)
    q = rng.standard_normal(dim).astype(np.float32)
    q /= np.linalg.norm(q)
    noise = rng.standard_normal((n, dim)).astype(np.float32)
    noise /= np.linalg.norm(noise, axis=1, keepdims=True)
    w = np.linspace(0.0, 1.0, n, dtype=np.float32)[:, None]
    X = (w * q[None, :] + (1.0 - w) * noise).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    X *= 0.9 * np.sqrt(dim) / 8.0                                       # components large enough to leave the zero digit
    X = np.clip(X, -0.999, 0.999).astype(np.float32)
    Q = np.concatenate([np.stack([q, -q]) * X[-1].max(), X[rng.integers(0, n, 259)]]).astype(np.float32)
    ix = ca.HNSWIndex(dim, ca.HNSWHyperParams(num_layers=3), storage_type=ca.StorageType(ca.StorageKind(O.STORAGE_SUBBYTE), 2))
    ix.upload_vectors(X)
    ref = ix.flat_search(Q, 10)
    for v in (1, 2, 3, 4):
        with _lib.tuning(flat_fp4_w8=v):
            got = ix.flat_search(Q, 10)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1].view(np.uint32), ref[1].view(np.uint32)) and np.array_equal(got[2], ref[2]), v
    oix = O.OracleIndex(O.HNSWParams(dim=dim, storage=O.STORAGE_SUBBYTE, resolution=2, num_layers=3)).set_vectors(X)
    oids, osc, ocnt = oix.flat_search_batch(Q, 10, threads=8)
    assert np.array_equal(got[2], ocnt) and np.array_equal(got[0], oids) and np.array_equal(got[1].view(np.uint32), osc.view(np.uint32))


def test_flat_code_scan_query_resident_kernel_equals_tile_kernel():
    """quaternary fused chunks run on the query-resident kernel (A fragments in registers, candidates streamed through LDS) — since
    round 5 with the digits as e2m1 nibbles on the scaled FP4 MFMA (flat_scan_q2_fp4); tuning knob flat_fp4 = 0 keeps the i8 digits
    (flat_scan_q2_areg), flat_tile_kernel = 1 forces the 256 x 128 tile kernel, flat_unfused = 1 the score-matrix path, flat_fp4_w8
    picks between the FP4 kernel's one-wave-per-SIMD and two-waves-per-SIMD forms (round 6): five implementations, one answer"""
    import cosdata_amd as ca
    n, dim, B, k = 50000, 768, 64, 10
    X = H.clustered_corpus(n, dim, n_centers=30, sigma=0.25, seed=31)
    Q = H.queries_from(X, B, noise=0.05, seed=9)
    ix = ca.HNSWIndex(dim, ca.HNSWHyperParams(num_layers=3), storage_type=ca.StorageType(ca.StorageKind(O.STORAGE_SUBBYTE), 2))
    ix.upload_vectors(X)
    ref = ix.flat_search(Q, k)
    from cosdata_amd import _lib
    for env, val in (("flat_fp4", 0), ("flat_tile_kernel", 1), ("flat_unfused", 1), ("flat_fp4_w8", 1), ("flat_fp4_w8", 2), ("flat_fp4_w8", 3), ("flat_fp4_w8", 4), ("flat_fp4_w8", 0)):
        with _lib.tuning(**{env: val}):
            got = ix.flat_search(Q, k)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1].view(np.uint32), ref[1].view(np.uint32)) and np.array_equal(got[2], ref[2]), env


@pytest.mark.parametrize("storage,res", [(O.STORAGE_U8, 0), (O.STORAGE_SUBBYTE, 2)])
def test_flat_code_scan_dot_product_metric_fused(storage, res):
    """DotProduct metric through the fused chunks (scores are the raw integer dots: no norms, no reciprocal estimate) — the
    quaternary case runs the query-resident kernel's `metric != cosine` arm"""
    import cosdata_amd as ca
    n, dim, B, k = 45000, 384, 70, 10
    X = H.clustered_corpus(n, dim, n_centers=25, sigma=0.25, seed=37) * 0.8
    Q = H.queries_from(X, B, noise=0.05, seed=5)
    ix = ca.HNSWIndex(dim, ca.HNSWHyperParams(num_layers=3), distance_metric=ca.DistanceMetric.DotProduct,
                      storage_type=ca.StorageType(ca.StorageKind(storage), res))
    ix.upload_vectors(X)
    ids, sc, cnt = ix.flat_search(Q, k)
    oix = O.OracleIndex(O.HNSWParams(dim=dim, metric=O.METRIC_DOT, storage=storage, resolution=res, num_layers=3)).set_vectors(X)
    oids, osc, ocnt = oix.flat_search_batch(Q, k, threads=8)
    assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))


@pytest.mark.parametrize("dim,metric", [(768, "cosine"), (512, "cosine"), (128, "dot"), (384, "cosine"), (768, "dot"), (1024, "cosine"), (1024, "dot")])
def test_flat_u8_scan_query_resident_kernel_equals_oracle_and_tile_kernel(dim, metric):
    """u8 fused chunks on flat_scan_u8_areg (round 6: rows of 128..1024 dims that are whole 64-byte chunks; 1024 dims: 32 query rows per wave); tuning knob flat_tile_kernel = 1
    keeps the 256 x 128 tile kernel, flat_unfused = 1 the score-matrix path: three implementations and the oracle, one answer.  Ragged
    shapes: n not a multiple of the 64-column tile, 261 queries (a second, almost empty row block)"""
    import cosdata_amd as ca
    from cosdata_amd import _lib
    n, B, k = 70013, 261, 10
    X = H.clustered_corpus(n, dim, n_centers=40, sigma=0.25, seed=51) * 0.9
    Q = H.queries_from(X, B, noise=0.05, seed=12)
    m = ca.DistanceMetric.Cosine if metric == "cosine" else ca.DistanceMetric.DotProduct
    ix = ca.HNSWIndex(dim, ca.HNSWHyperParams(num_layers=3), distance_metric=m, storage_type=ca.StorageType(ca.StorageKind(O.STORAGE_U8), 0))
    ix.upload_vectors(X)
    ref = ix.flat_search(Q, k)
    for env in ("flat_tile_kernel", "flat_unfused"):
        with _lib.tuning(**{env: 1}):
            got = ix.flat_search(Q, k)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1].view(np.uint32), ref[1].view(np.uint32)) and np.array_equal(got[2], ref[2]), env
    oix = O.OracleIndex(O.HNSWParams(dim=dim, metric=O.METRIC_COSINE if metric == "cosine" else O.METRIC_DOT, storage=O.STORAGE_U8, resolution=0,
                                     num_layers=3)).set_vectors(X)
    oids, osc, ocnt = oix.flat_search_batch(Q, k, threads=8)
    assert np.array_equal(ref[2], ocnt) and np.array_equal(ref[0], oids) and np.array_equal(ref[1].view(np.uint32), osc.view(np.uint32))


def test_flat_code_scan_fp4_two_waves_per_simd_dot_product_metric():
    """DotProduct metric on flat_scan_q2_fp4_w8 (768-dim quaternary codes: scores are the raw integer dots, the screen's reciprocal is the
    constant 4) against the oracle and the one-wave-per-SIMD kernel"""
    import cosdata_amd as ca
    from cosdata_amd import _lib
    n, dim, B, k = 45011, 768, 70, 10
    X = H.clustered_corpus(n, dim, n_centers=25, sigma=0.25, seed=37) * 0.8
    Q = H.queries_from(X, B, noise=0.05, seed=5)
    ix = ca.HNSWIndex(dim, ca.HNSWHyperParams(num_layers=3), distance_metric=ca.DistanceMetric.DotProduct,
                      storage_type=ca.StorageType(ca.StorageKind(O.STORAGE_SUBBYTE), 2))
    ix.upload_vectors(X)
    ids, sc, cnt = ix.flat_search(Q, k)
    with _lib.tuning(flat_fp4_w8=0):
        got = ix.flat_search(Q, k)
    assert np.array_equal(got[0], ids) and np.array_equal(got[1].view(np.uint32), sc.view(np.uint32)) and np.array_equal(got[2], cnt)
    oix = O.OracleIndex(O.HNSWParams(dim=dim, metric=O.METRIC_DOT, storage=O.STORAGE_SUBBYTE, resolution=2, num_layers=3)).set_vectors(X)
    oids, osc, ocnt = oix.flat_search_batch(Q, k, threads=8)
    assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))
