"""GPU: the locality order of big launches (cos_index_set_walk_order, kernels_order.hip) changes when and where a query's walk runs,
never what it returns.  Launches of COS_WALK_ORDER_DEFAULT_MIN_B queries and more take it by default."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _same(a, b):
    return all(np.array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32)) for x, y in zip(a, b))


def _check_against_oracle(oix, res, Q, top_k, sample):
    ids, sc, cnt = res
    oids, osc, ocnt = oix.search_batch(Q[sample], top_k, threads=4)[:3]
    assert np.array_equal(cnt[sample], ocnt)
    for j, b in enumerate(sample):
        c = int(ocnt[j])
        assert np.array_equal(ids[b, :c], oids[j, :c]), f"query {b}"
        assert np.array_equal(sc[b, :c].view(np.uint32), osc[j, :c].view(np.uint32)), f"query {b}"


@pytest.mark.parametrize("visited", [0, 1])
def test_big_launch_is_ordered_by_default_and_identical(visited):
    import cosdata_amd as ca
    X = H.clustered_corpus(6000, 96, n_centers=24, seed=31)
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=4, ef_construction=48, ef_search=32)
    dix = H.device_index_from_oracle(oix, X, visited_mode=visited)
    if visited:
        oix.set_visited_mode(O.VISITED_EXACT)
    B = ca.HNSWIndex.WALK_ORDER_DEFAULT_MIN_B + 808         # not a multiple of 8 XCDs x anything convenient: 9000
    Q = H.queries_from(X, B, noise=0.05, seed=5)
    cuts = dix.walk_order_cuts()
    assert cuts and all(1 <= l <= 4 for l in cuts) and cuts == sorted(cuts, reverse=True)
    ordered = dix.batch_search(Q, 10)                         # default: locality order from 8192 queries
    dix.set_walk_order(0)
    plain = dix.batch_search(Q, 10)
    dix.set_walk_order(ca.HNSWIndex.WALK_ORDER_DEFAULT_MIN_B)
    assert _same(ordered, plain)
    _check_against_oracle(oix, ordered, Q, 10, np.arange(0, B, 37))
    walk_o = dix.ann_search_batch(Q)                          # the walk's own per-level lists
    dix.set_walk_order(0)
    walk_p = dix.ann_search_batch(Q)
    assert _same(walk_o, walk_p)


def test_order_table_follows_the_graph():
    """a new graph on the same handle: the order key's table is rebuilt (a stale one would index another graph's nodes)"""
    import cosdata_amd as ca
    X = H.clustered_corpus(5000, 64, n_centers=16, seed=8)
    o1 = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=4, ef_construction=32, ef_search=32, seed=1)
    o2 = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=4, ef_construction=32, ef_search=32, seed=2)
    assert any(a[0].size != b[0].size or not np.array_equal(a[1], b[1]) for a, b in zip(o1.export_graph(), o2.export_graph()))
    dix = H.device_index_from_oracle(o1, X)
    B = ca.HNSWIndex.WALK_ORDER_DEFAULT_MIN_B
    Q = H.queries_from(X, B, noise=0.05, seed=6)
    r1 = dix.batch_search(Q, 5)
    _check_against_oracle(o1, r1, Q, 5, np.arange(0, B, 61))
    dix.upload_vectors(X)                                     # drops graph 1 (a level upload checks its nodes against the level below)
    dix.upload_graph(o2.export_graph(), o2.root_raw())
    r2 = dix.batch_search(Q, 5)
    _check_against_oracle(o2, r2, Q, 5, np.arange(0, B, 61))
    dix.set_walk_order(0)
    assert _same(r2, dix.batch_search(Q, 5))
