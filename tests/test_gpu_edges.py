"""GPU parity on the corners of the path: non-default graph shapes (M, M0, shortlist_size, num_layers), the
DotProduct metric, tiny and degenerate corpora, large top_k, single-query batches — each against the oracle on the
same graph (walk lists per level AND final results, bit-exact)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H
from tests.test_gpu_parity import _assert_same_search, _assert_same_walk

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,M0,shortlist", [(16, 32, 64), (8, 16, 64), (32, 64, 20), (32, 64, 40), (16, 64, 64), (64, 64, 64), (4, 8, 3)])
@pytest.mark.parametrize("storage,res", [(O.STORAGE_U8, 0), (O.STORAGE_SUBBYTE, 2), (O.STORAGE_F32, 0)])
def test_graph_shapes(M, M0, shortlist, storage, res):
    X = H.clustered_corpus(3000, 112, n_centers=12, seed=M * 7 + M0)
    oix = H.oracle_index(X, storage, res, num_layers=4, ef_construction=40, ef_search=40, neighbors_count=M,
                         level0_neighbors_count=M0, shortlist_size=shortlist)
    dix = H.device_index_from_oracle(oix, X)
    Q = H.queries_from(X, 20, seed=M0)
    _assert_same_walk(oix, dix, Q)
    _assert_same_search(oix, dix, Q, 10)


@pytest.mark.parametrize("storage,res", [(O.STORAGE_U8, 0), (O.STORAGE_SUBBYTE, 2), (O.STORAGE_F16, 0)])
def test_dot_product_metric(storage, res):
    X = H.uniform_corpus(2500, 96, seed=31)
    oix = H.oracle_index(X, storage, res, metric=O.METRIC_DOT, num_layers=4, ef_construction=48, ef_search=48)
    dix = H.device_index_from_oracle(oix, X)
    Q = H.queries_from(X, 24, seed=8)
    _assert_same_walk(oix, dix, Q)
    _assert_same_search(oix, dix, Q, 10)


def test_dot_product_f32_is_storage_mismatch():
    """DotProductDistance::calculate has no FullPrecisionFP arm (dotproduct.rs:20-64): both sides report StorageMismatch."""
    import cosdata_amd as ca
    X = H.uniform_corpus(100, 32, seed=1)
    with pytest.raises(ValueError):
        H.oracle_index(X, O.STORAGE_F32, 0, metric=O.METRIC_DOT, num_layers=2, ef_construction=16, ef_search=16)
    with pytest.raises(ca.CosdataError) as ei:
        ca.HNSWIndex(32, ca.HNSWHyperParams(num_layers=2), ca.DistanceMetric.DotProduct, ca.StorageType.FullPrecisionFP(), (-1.0, 1.0))
    assert ei.value.status == 1


@pytest.mark.parametrize("n", [1, 2, 5, 63, 64, 65])
def test_tiny_corpora(n):
    X = H.uniform_corpus(n, 48, seed=n)
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=3, ef_construction=16, ef_search=16)
    dix = H.device_index_from_oracle(oix, X)
    Q = H.uniform_corpus(6, 48, seed=100 + n)
    _assert_same_walk(oix, dix, Q)
    _assert_same_search(oix, dix, Q, 10)   # top_k > n: counts < top_k, tail untouched
    _assert_same_search(oix, dix, Q[:1], 1)


@pytest.mark.parametrize("top_k", [1, 3, 50, 100, 200])
def test_top_k_range(top_k):
    X = H.clustered_corpus(6000, 64, n_centers=8, seed=77)
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=5, ef_construction=64, ef_search=256)
    dix = H.device_index_from_oracle(oix, X)
    Q = H.queries_from(X, 12, seed=5)
    _assert_same_search(oix, dix, Q, top_k)


@pytest.mark.parametrize("ef", [1, 2, 17, 65, 100, 257, 400, 512, 513, 700, 1024])
def test_ef_range(ef):
    X = H.clustered_corpus(5000, 80, n_centers=10, seed=13)
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=4, ef_construction=48, ef_search=ef)
    dix = H.device_index_from_oracle(oix, X)
    Q = H.queries_from(X, 10, seed=ef)
    _assert_same_walk(oix, dix, Q)
    _assert_same_search(oix, dix, Q, 10)


@pytest.mark.parametrize("storage,res,dim", [(O.STORAGE_SUBBYTE, 2, 256), (O.STORAGE_F32, 0, 48), (O.STORAGE_U8, 0, 1024)])
def test_ef_above_512_other_storages_and_wide_rows(storage, res, dim):
    """round 5: ef_search up to 1024 (a pool of 64 x 16 keys per wave): quaternary and f32 storages, 1024-byte u8 rows; more pops
    than the level has nodes on the upper levels (the walk ends when the pool runs dry)"""
    X = H.clustered_corpus(3000, dim, n_centers=10, seed=21 + dim) * (0.9 if storage == O.STORAGE_SUBBYTE else 1.0)
    oix = H.oracle_index(X, storage, res, num_layers=4, ef_construction=40, ef_search=1000)
    dix = H.device_index_from_oracle(oix, X)
    Q = H.queries_from(X, 6, seed=2) * (0.9 if storage == O.STORAGE_SUBBYTE else 1.0)
    _assert_same_walk(oix, dix, Q)
    _assert_same_search(oix, dix, Q, 10)


def test_ef_above_the_general_kernels_lds_is_refused_loudly():
    import cosdata_amd as ca
    X = H.clustered_corpus(500, 32, n_centers=4, seed=1)
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=2, ef_construction=16, ef_search=16)
    dix = H.device_index_from_oracle(oix, X)
    with pytest.raises(ca.CosdataError) as ei:
        dix.set_ef_search(16385)
    assert ei.value.status == 4          # Unimplemented: never a silent difference
    dix.set_ef_search(16384)
    dix.set_ef_search(1024)


def test_single_layer_and_deep_hierarchy():
    X = H.uniform_corpus(2000, 64, seed=3)
    for L in (1, 2, 12):
        oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=L, ef_construction=32, ef_search=32)
        dix = H.device_index_from_oracle(oix, X)
        Q = H.queries_from(X, 8, seed=L)
        _assert_same_walk(oix, dix, Q)
        _assert_same_search(oix, dix, Q, 10)


def test_c1_config_build_and_search():
    """BASELINE configs[0] (tests/test.py:73-88 of the reference): 10k x 768 uniform(-1,1), num_layers 7,
    ef_construction 512, ef_search 256, M 32 / M0 64, queries = corpus vectors, top_k 5 and 10 — built on the
    device AND by the oracle's statement of the same schedule (build_rounds); graphs, walks and results must be identical."""
    import cosdata_amd as ca
    n, d = 10000, 768
    X = H.uniform_corpus(n, d, seed=42)
    p = O.HNSWParams(dim=d, num_layers=7, ef_construction=512, ef_search=256, seed=42)
    oix = O.OracleIndex(p).set_vectors(X)
    oix.build_rounds(256, greedy=False)
    hp = ca.HNSWHyperParams(num_layers=7, ef_construction=512, ef_search=256, level_0_neighbors_count=64, neighbors_count=32)
    dix = ca.HNSWIndex(d, hp, ca.DistanceMetric.Cosine, ca.StorageType.UnsignedByte(), (-1.0, 1.0), 64, seed=42)
    dix.upload_vectors(X)
    dix.build(256)
    og, dg = oix.export_graph(), dix.download_graph()
    assert len(og) == len(dg)
    for (oi, on), (di, dn) in zip(og, dg):
        assert np.array_equal(oi, di) and np.array_equal(on, dn)
    Q = X[:100].copy()
    _assert_same_walk(oix, dix, Q[:16])
    for k in (5, 10):
        _assert_same_search(oix, dix, Q, k)
    ids = dix.batch_search(Q, 5)[0]
    assert (ids[:, 0] == np.arange(100)).mean() >= 0.9   # a corpus vector (nearly always) finds itself; uniform high-d data is adversarial


def test_zero_raw_vectors_and_zero_query_follow_x86_nan_order():
    """finalize_ann_results divides by |q|*|raw| without a zero check (vector_store.rs:425-427): a zero raw vector (or query)
    gives 0/0.  On x86-64, the reference's platform, that is the NEGATIVE default NaN, which f32::total_cmp sorts below
    everything — the zero vector drops to the end of the rerank.  The device must reproduce ids AND the 0xFFC00000 bits."""
    X = H.clustered_corpus(2000, 64, n_centers=6, seed=4)
    X[[5, 700, 1500]] = 0.0          # u8 codes 127...: valid quantized norm, reachable by the walk, |raw| = 0
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=3, ef_construction=32, ef_search=64)
    dix = H.device_index_from_oracle(oix, X)
    Q = H.queries_from(X, 12, seed=9)
    Q[3] = 0.0                       # |q| = 0 in the rerank: every score is 0/0
    Q[4] = X[1] * 1e-3               # tiny but non-zero
    _assert_same_walk(oix, dix, Q)
    _assert_same_search(oix, dix, Q, 10)
    _assert_same_search(oix, dix, Q, 400)   # returns every candidate: the zero vectors must close each list with -NaN
    ids, sc, cnt = dix.batch_search(Q, 400)
    assert all(int(ids[b, int(cnt[b]) - 1]) in (5, 700, 1500) for b in range(12) if b != 3)
    assert (sc[3, :int(cnt[3])].view(np.uint32) == 0xFFC00000).all()
    keep = [i for i in range(12) if i != 3]
    bi, bs = dix.bruteforce_topk(Q[keep], 10)
    oi, osc = O.bruteforce_topk(X, Q[keep], 10, threads=4)
    assert np.array_equal(bi, oi) and np.array_equal(bs.view(np.uint32), osc.view(np.uint32))
