"""GPU: the walk outside the fast kernels' parameter domain (kernels_walk_general.hip): ef above 1024 (hnsw/types.rs:10-17 takes any
u32) and more than 64 scanned neighbour slots per node (min(neighbors_count, shortlist_size); the reference's own gRPC test config
sets shortlist_size = 100, grpc/vectors/tests.rs:47).  Rounds 1-5 refused both at cos_index_create.  Per-level lists, results,
counts and score bits must equal the oracle's (traverse_find_nearest, vector_store.rs:1112-1204) for every storage, both filters,
the device builder (ef_construction above 1024, wide shortlists) and delete_embedding's unseeded walks."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H
from tests.test_gpu_parity import _assert_same_search, _assert_same_walk

pytestmark = pytest.mark.gpu


STORAGES = [(O.STORAGE_U8, 0, 96), (O.STORAGE_SUBBYTE, 2, 256), (O.STORAGE_SUBBYTE, 1, 256), (O.STORAGE_SUBBYTE, 3, 128),
            (O.STORAGE_F32, 0, 40), (O.STORAGE_F16, 0, 72)]


@pytest.mark.parametrize("storage,res,dim", STORAGES)
@pytest.mark.parametrize("ef,M,M0,shortlist", [(1500, 16, 32, 64), (48, 128, 256, 128), (1100, 64, 128, 100)])
def test_general_walk_equals_oracle(storage, res, dim, ef, M, M0, shortlist):
    scale = 0.9 if storage == O.STORAGE_SUBBYTE else 1.0
    X = H.clustered_corpus(2500, dim, n_centers=10, seed=5 + dim + ef) * scale
    oix = H.oracle_index(X, storage, res, num_layers=3, ef_construction=40, ef_search=ef, neighbors_count=M, level0_neighbors_count=M0,
                         shortlist_size=shortlist)
    dix = H.device_index_from_oracle(oix, X)
    Q = H.queries_from(X, 7, seed=ef) * scale
    _assert_same_walk(oix, dix, Q)
    _assert_same_search(oix, dix, Q, 10)


def test_general_walk_exact_filter_dot_metric_and_wide_rows():
    # COS_VISITED_EXACT, DotProductDistance, u8 rows of more than 64 chunks' worth of lanes in one pass (1536 dims)
    X = H.clustered_corpus(2000, 1536, n_centers=8, seed=91)
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=3, ef_construction=32, ef_search=1300, metric=O.METRIC_DOT, neighbors_count=128,
                         level0_neighbors_count=128, shortlist_size=128)
    dix = H.device_index_from_oracle(oix, X, visited_mode=1)
    oix.set_visited_mode(O.VISITED_EXACT)
    Q = H.queries_from(X, 5, seed=3)
    _assert_same_walk(oix, dix, Q)
    _assert_same_search(oix, dix, Q, 25)


def test_general_walk_zero_norm_query_is_a_calculation_error():
    X = H.uniform_corpus(1500, 64, seed=17)
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=2, ef_construction=24, ef_search=1200)
    dix = H.device_index_from_oracle(oix, X)
    Q = H.queries_from(X, 4, seed=1)
    Q[2, :] = -1.0  # quantizes to all-zero bytes -> |q| = 0 -> DistanceError::CalculationError (cosine.rs:228-232)
    o = oix.search_batch(Q, 5, raise_on_error=False)
    ids, sc, cnt, rc, status = dix.batch_search(Q, 5, return_status=True)
    assert rc == 2 and o[3] == 2
    assert np.array_equal(status, o[4]) and list(status) == [0, 0, 2, 0]
    good = np.array([0, 1, 3])
    assert np.array_equal(ids[good], o[0][good]) and np.array_equal(sc[good].view(np.uint32), o[1][good].view(np.uint32))


def test_switching_between_the_domains_on_one_handle():
    # ef_search moves across 1024 on a live handle: the fast kernels (with their level table) and the general kernel take turns
    X = H.clustered_corpus(4000, 80, n_centers=10, seed=29)
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=4, ef_construction=48, ef_search=64)
    dix = H.device_index_from_oracle(oix, X)
    Q = H.queries_from(X, 6, seed=8)
    for ef in (64, 2048, 1024, 1025, 200):
        oix.set_ef_search(ef)
        dix.set_ef_search(ef)
        _assert_same_search(oix, dix, Q, 10)


@pytest.mark.parametrize("efc,M,M0,shortlist", [(1100, 8, 16, 64), (40, 128, 128, 128)])
def test_device_builder_in_the_general_domain(efc, M, M0, shortlist):
    import cosdata_amd as ca
    X = H.clustered_corpus(1800, 48, n_centers=8, seed=61)
    p = O.HNSWParams(dim=48, num_layers=3, ef_construction=efc, ef_search=32, neighbors_count=M, level0_neighbors_count=M0,
                     shortlist_size=shortlist, seed=7)
    oix = O.OracleIndex(p).set_vectors(X)
    oix.build_rounds(256, greedy=False)
    hp = ca.HNSWHyperParams(num_layers=3, ef_construction=efc, ef_search=32, level_0_neighbors_count=M0, neighbors_count=M)
    dix = ca.HNSWIndex(48, hp, ca.DistanceMetric.Cosine, ca.StorageType.UnsignedByte(), (p.range_lo, p.range_hi), shortlist, seed=7)
    dix.upload_vectors(X)
    dix.build(256)
    og, dg = oix.export_graph(), dix.download_graph()
    assert len(og) == len(dg)
    for (on, oa), (dn, da) in zip(og, dg):
        assert np.array_equal(on, dn) and np.array_equal(oa, da)
    Q = H.queries_from(X, 6, seed=2)
    _assert_same_search(oix, dix, Q, 10)
    if shortlist > 64:   # delete_embedding's walks (ef 512, unseeded filter) on an index of the general domain
        ids = np.array([5, 77, 1203], np.uint32)
        oix.delete(ids)
        dix.delete(ids)
        og, dg = oix.export_graph(), dix.download_graph()
        for (on, oa), (dn, da) in zip(og, dg):
            assert np.array_equal(on, dn) and np.array_equal(oa, da)


@pytest.mark.parametrize("storage,res,dim", [(O.STORAGE_U8, 0, 3072), (O.STORAGE_U8, 0, 2050), (O.STORAGE_U8, 0, 3090), (O.STORAGE_U8, 0, 4112),
                                             (O.STORAGE_SUBBYTE, 2, 4160), (O.STORAGE_SUBBYTE, 1, 8320), (O.STORAGE_SUBBYTE, 3, 2080)])
def test_rows_wider_than_the_fast_kernels_chunk_passes(storage, res, dim):
    """u8 above 2048 dimensions (text-embedding-3-large is 3072) and SubByte rows of more than 64 x 16 bytes: refused until round 6.
    u8 rows of 2049..4096 dimensions take walk_kernel with three / four chunk passes, wider ones (and the SubByte rows) walk_general_kernel.
    Device build (walks with ef_construction + link) and search against the oracle on the same schedule."""
    import cosdata_amd as ca
    scale = 0.9 if storage == O.STORAGE_SUBBYTE else 1.0
    X = H.clustered_corpus(1200, dim, n_centers=8, seed=dim) * scale
    p = O.HNSWParams(dim=dim, storage=storage, resolution=res, num_layers=3, ef_construction=32, ef_search=48, seed=3)
    oix = O.OracleIndex(p).set_vectors(X)
    oix.build_rounds(256, greedy=False)
    hp = ca.HNSWHyperParams(num_layers=3, ef_construction=32, ef_search=48, level_0_neighbors_count=p.level0_neighbors_count, neighbors_count=p.neighbors_count)
    dix = ca.HNSWIndex(dim, hp, ca.DistanceMetric.Cosine, ca.StorageType(ca.StorageKind(storage), res), (p.range_lo, p.range_hi), p.shortlist_size, seed=3)
    dix.upload_vectors(X)
    dix.build(256)
    og, dg = oix.export_graph(), dix.download_graph()
    assert len(og) == len(dg)
    for (on, oa), (dn, da) in zip(og, dg):
        assert np.array_equal(on, dn) and np.array_equal(oa, da)
    Q = H.queries_from(X, 6, seed=2) * scale
    _assert_same_walk(oix, dix, Q)
    _assert_same_search(oix, dix, Q, 10)


@pytest.mark.parametrize("storage,res,dim", [(O.STORAGE_U8, 0, 3072), (O.STORAGE_SUBBYTE, 2, 4160)])
def test_exhaustive_scan_over_wide_rows(storage, res, dim):
    """cos_flat_search_batch on rows the walk's fast kernels do not hold: the tile GEMM loops over any number of k panels"""
    import cosdata_amd as ca
    X = H.uniform_corpus(2200, dim, seed=13) * 0.9
    Q = H.queries_from(X, 9, noise=0.05, seed=8)
    ix = ca.HNSWIndex(dim, ca.HNSWHyperParams(num_layers=3), storage_type=ca.StorageType(ca.StorageKind(storage), res))
    ix.upload_vectors(X)
    ids, sc, cnt = ix.flat_search(Q, 10)
    oix = O.OracleIndex(O.HNSWParams(dim=dim, storage=storage, resolution=res, num_layers=3)).set_vectors(X)
    oids, osc, ocnt = oix.flat_search_batch(Q, 10, threads=4)
    assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))


@pytest.mark.parametrize("dim,ef,visited", [(3072, 48, 0), (2560, 100, 0), (3072, 200, 1), (2100, 400, 0), (3000, 1000, 0),
                                            (4096, 64, 0), (3500, 128, 1), (4000, 256, 0), (3100, 512, 0), (4096, 1024, 0)])
def test_three_and_four_chunk_passes_of_the_fast_kernel(dim, ef, visited):
    """2049..4096-dim u8 rows in walk_kernel (three / four chunk passes; every pool width, both filters, the level table and the
    locality-ordered split through the walk variants of _assert_same_walk)"""
    X = H.clustered_corpus(2500, dim, n_centers=10, seed=dim + ef)
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=3, ef_construction=40, ef_search=ef)
    dix = H.device_index_from_oracle(oix, X, visited_mode=visited)
    if visited:
        oix.set_visited_mode(O.VISITED_EXACT)
    Q = H.queries_from(X, 7, seed=ef)
    _assert_same_walk(oix, dix, Q)
    _assert_same_search(oix, dix, Q, 10)
