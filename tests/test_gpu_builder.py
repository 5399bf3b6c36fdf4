"""Device builder (cos_index_build: GPU walks + the round-synchronous link kernels) vs the oracle's statement of the same
schedule (coso_index_build_rounds, ordered variant): identical graphs, slot for slot."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _device_index(X, storage, res, metric=0, shortlist_size=64, **kw):
    import cosdata_amd as ca
    hp = ca.HNSWHyperParams(num_layers=kw.get("num_layers", 9), ef_construction=kw.get("ef_construction", 128),
                            ef_search=kw.get("ef_search", 256), neighbors_count=kw.get("neighbors_count", 32),
                            level_0_neighbors_count=kw.get("level0_neighbors_count", 64))
    dix = ca.HNSWIndex(X.shape[1], hp, ca.DistanceMetric(metric), ca.StorageType(ca.StorageKind(storage), res), shortlist_size=shortlist_size,
                       seed=kw.get("seed", 42))
    dix.upload_vectors(X)
    if "latency" in kw:
        dix.set_latency_mode(kw["latency"])
    dix.set_latency_waves(kw.get("waves", 0))   # four waves per new vector only where a test asks for it
    return dix


def _assert_same_graph(dix, oix):
    assert np.array_equal(dix.download_root(), oix.root_raw())
    dg, og = dix.download_graph(), oix.export_graph()
    assert len(dg) == len(og)
    for l, ((di, dn), (oi, on)) in enumerate(zip(dg, og)):
        assert np.array_equal(di, oi), f"level {l}: node sets differ"
        assert np.array_equal(dn, on), f"level {l}: adjacency differs in {np.count_nonzero((dn != on).any(axis=1))} of {len(on)} rows"


@pytest.mark.parametrize("storage,res", [(O.STORAGE_U8, 0), (O.STORAGE_SUBBYTE, 2), (O.STORAGE_F32, 0), (O.STORAGE_F16, 0), (O.STORAGE_SUBBYTE, 1), (O.STORAGE_SUBBYTE, 3)])
@pytest.mark.parametrize("n,dim,bs,latency,waves", [(1500, 96, 64, 0, 0), (4000, 128, 512, 0, 0), (4000, 128, 512, 2048, 0), (4000, 128, 512, 2048, 2048)])
def test_device_build_equals_oracle_rounds(storage, res, n, dim, bs, latency, waves):
    """latency = cos_index_set_latency_mode: the builder's walks run on walk_kernel (0) or, for u8 / quaternary codes, on
    walk_lat_kernel (batches of <= 2048 new vectors) — same graph either way"""
    if latency and storage not in (O.STORAGE_U8,) and not (storage == O.STORAGE_SUBBYTE and res == 2):
        pytest.skip("the latency kernel covers u8 and quaternary codes")
    X = H.clustered_corpus(n, dim, n_centers=16, seed=9)
    kw = dict(num_layers=5, ef_construction=64, ef_search=64, seed=77)
    dix = _device_index(X, storage, res, latency=latency, waves=waves, **kw).build(bs)
    oix = O.OracleIndex(O.HNSWParams(dim=dim, storage=storage, resolution=res, **kw)).set_vectors(X).build_rounds(bs, greedy=False)[0]
    _assert_same_graph(dix, oix)
    # and the freshly built device graph answers exactly like the oracle on it
    Q = H.queries_from(X, 32, seed=5)
    ids, sc, cnt = dix.batch_search(Q, 10)
    oids, osc, ocnt = oix.search_batch(Q, 10, threads=4)[:3]
    assert np.array_equal(ids, oids) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))
    gt, _ = O.bruteforce_topk(X, Q, 10, threads=4)
    recall = np.mean([len(set(ids[i]) & set(gt[i])) / 10 for i in range(len(Q))])
    if storage != O.STORAGE_SUBBYTE:  # the reference's 2-bit quantizer ignores values_range: normalised data collapses to ~1 bit
        assert recall >= 0.8, recall


@pytest.mark.parametrize("M,M0,shortlist,bs", [(16, 32, 64, 4096), (8, 16, 64, 300), (4, 8, 3, 128), (32, 128, 64, 512), (64, 256, 64, 256)])
def test_device_link_graph_shapes(M, M0, shortlist, bs):
    """slot counts below and above one wave (M0 = 128 / 256: several slots per lane), tiny M (heavy eviction traffic), and
    batches as large as a quarter of the graph (many conflicting claims -> many rounds)"""
    X = H.clustered_corpus(3000, 64, n_centers=6, seed=M + M0)
    kw = dict(num_layers=4, ef_construction=48, ef_search=48, seed=5, neighbors_count=M, level0_neighbors_count=M0)
    oix = O.OracleIndex(O.HNSWParams(dim=64, shortlist_size=shortlist, **kw)).set_vectors(X).build_rounds(bs, greedy=False)[0]
    for latency, waves in ((0, 0), (0xFFFFFFFF, 0), (0xFFFFFFFF, 0xFFFFFFFF)):
        dix = _device_index(X, O.STORAGE_U8, 0, shortlist_size=shortlist, latency=latency, waves=waves, **kw).build(bs)
        _assert_same_graph(dix, oix)


def test_device_link_dot_metric_and_duplicates():
    """DotProduct (MetricResult::min = -inf) and a corpus with exact duplicate vectors (equal similarities everywhere: every
    strict comparison of add_neighbor is exercised on ties)"""
    X = H.uniform_corpus(1200, 48, seed=3)
    X[600:900] = X[:300]
    kw = dict(num_layers=3, ef_construction=40, ef_search=40, seed=9)
    for metric in (O.METRIC_COSINE, O.METRIC_DOT):
        dix = _device_index(X, O.STORAGE_U8, 0, metric=metric, **kw).build(200)
        oix = O.OracleIndex(O.HNSWParams(dim=48, metric=metric, **kw)).set_vectors(X).build_rounds(200, greedy=False)[0]
        _assert_same_graph(dix, oix)


def test_failed_build_leaves_no_graph():
    """a zero-norm vector aborts index_embedding with CalculationError (cosine.rs:228-232): the handle must come out of the
    failed build without a graph (search -> NotReady), not half-linked"""
    import cosdata_amd as ca
    X = H.uniform_corpus(800, 32, seed=1)
    X[500] = -1.0   # quantizes to all-zero bytes -> |v| = 0
    dix = _device_index(X, O.STORAGE_U8, 0, num_layers=3, ef_construction=32, ef_search=32)
    with pytest.raises(ca.CosdataError) as ei:
        dix.build(64)
    assert ei.value.status == 2
    with pytest.raises(ca.CosdataError) as ei:
        dix.batch_search(X[:4], 5)
    assert ei.value.status == 6
