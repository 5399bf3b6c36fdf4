"""CPU: the oracle's append (coso_index_append_vectors + coso_index_build_rounds_continue: index_embeddings called again on a live
index, vector_store.rs:714-780).  Pins, without a device, what tests/test_gpu_append.py holds the device to."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H


def _graphs_equal(a, b):
    return len(a) == len(b) and all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(a, b))


def _boundaries(B, upto):
    out, ins = [], 0
    while ins < upto:
        ins += min(B, max(1, ins // 4))
        out.append(ins)
    return out


@pytest.mark.parametrize("storage,res,dim,B", [(O.STORAGE_U8, 0, 64, 200), (O.STORAGE_SUBBYTE, 2, 128, 64), (O.STORAGE_F32, 0, 32, 1)])
def test_append_on_a_batch_boundary_is_the_full_build(storage, res, dim, B):
    """the rounds schedule inserts batches of min(B, max(1, inserted / 4)): stopping on one of its boundaries and continuing gives the full
    build's graph (same RNG stream for the level draws, same batches, same snapshots) — and level draws are those of the full build
    wherever the first build stops"""
    bs = _boundaries(B, 2600)
    n0 = [b for b in bs if b >= 900][0]
    n1 = [b for b in bs if b >= 1800][0]
    X = H.clustered_corpus(n1, dim, n_centers=12, seed=dim)
    p = dict(dim=dim, storage=storage, resolution=res, num_layers=4, ef_construction=40, ef_search=32, seed=9)
    full = O.OracleIndex(O.HNSWParams(**p)).set_vectors(X)
    full.build_rounds(B)
    part = O.OracleIndex(O.HNSWParams(**p)).set_vectors(X[:n0])
    part.build_rounds(B)
    part.append(X[n0:], B)
    assert _graphs_equal(part.export_graph(), full.export_graph())
    off = O.OracleIndex(O.HNSWParams(**p)).set_vectors(X[:n0 - 7])          # NOT a boundary: other batches, other graph, same node sets
    off.build_rounds(B)
    off.append(X[n0 - 7:], B)
    g, f = off.export_graph(), full.export_graph()
    assert all(np.array_equal(x[0], y[0]) for x, y in zip(g, f))
    Q = H.queries_from(X, 64, noise=0.05, seed=1)
    assert np.array_equal(part.search_batch(Q, 10, threads=2)[0], full.search_batch(Q, 10, threads=2)[0])


def test_append_needs_a_rounds_built_graph_and_finds_the_new_vectors():
    X = H.clustered_corpus(1600, 48, n_centers=10, seed=4)
    p = O.HNSWParams(dim=48, num_layers=3, ef_construction=32, ef_search=32, seed=1)
    seq = O.OracleIndex(p).set_vectors(X[:1000]).build()                     # the sequential builder leaves no schedule to continue
    with pytest.raises(ValueError):
        seq.append(X[1000:], 64)
    a = O.OracleIndex(p).set_vectors(X[:1000])
    a.build_rounds(64)
    a.append(X[1000:1300], 64)
    a.append(X[1300:], 64)                                                    # twice in a row
    assert [len(l[0]) for l in a.export_graph()][0] == 1601
    Q = H.queries_from(X[1000:], 100, noise=0.03, seed=2)
    ids = a.search_batch(Q, 10, threads=2)[0]
    gt, _ = O.bruteforce_topk(X, Q, 10, threads=2)
    assert np.mean([len(set(ids[i]) & set(gt[i])) / 10 for i in range(len(Q))]) > 0.9 and (ids >= 1000).any()
    imp = O.OracleIndex(p).set_vectors(X).import_graph(a.export_graph(), a.root_raw())   # an imported graph cannot be continued either
    with pytest.raises(ValueError):
        imp.append(X[:5], 64)


def test_delete_removes_every_edge_to_the_node_and_the_id_from_every_answer():
    """coso_index_delete (delete_embedding, vector_store.rs:1206-1400): edges are symmetric by construction, so after the node's neighbours
    dropped their back edges no slot of any level names it; its own slots are empty; searches for the deleted vectors themselves return
    other ids; untouched queries keep their answers wherever no deleted id was among them"""
    X = H.clustered_corpus(2500, 64, n_centers=16, seed=8)
    p = O.HNSWParams(dim=64, num_layers=4, ef_construction=40, ef_search=48, seed=2)
    a = O.OracleIndex(p).set_vectors(X)
    a.build_rounds(128)
    Q = H.queries_from(X, 150, noise=0.05, seed=3)
    before = a.search_batch(Q, 10, threads=2)[0]
    dele = np.arange(0, 400, 3, dtype=np.uint32)
    a.delete(dele)
    for ids, nbr in a.export_graph():
        assert not np.isin(nbr, dele).any()
        assert (nbr[np.isin(ids, dele)] == O.SLOT_EMPTY).all()
    after = a.search_batch(Q, 10, threads=2)[0]
    assert not np.isin(after, dele).any()
    assert not np.isin(a.search_batch(X[dele[:50]], 5, threads=2)[0], dele).any()
    untouched = ~np.isin(before, dele).any(axis=1)
    assert untouched.sum() > 20 and np.mean((before[untouched] == after[untouched]).all(axis=1)) > 0.8
    with pytest.raises(ValueError):
        a.delete([2500])                                                       # not a resident vector


def test_restored_link_state_of_an_imported_graph():
    """coso_index_restore_link_state = what the reference has after a reload: the slot similarities it persisted (recomputed here: they are
    the ones the builder stored, bit for bit) and the lowest caches by ProbNode::new_with_neighbors_and_versions (prob_node.rs:145-181).  An
    imported graph then takes appends and deletes; with the same seed the appended ids get the levels a native build draws."""
    X = H.clustered_corpus(2600, 64, n_centers=14, seed=6)
    p = dict(dim=64, num_layers=4, ef_construction=40, ef_search=40, seed=13)
    a = O.OracleIndex(O.HNSWParams(**p)).set_vectors(X[:1800])
    a.build_rounds(128)
    b = O.OracleIndex(O.HNSWParams(**p)).set_vectors(X[:1800]).import_graph(a.export_graph(), a.root_raw())
    with pytest.raises(ValueError):
        b.append(X[1800:1810], 128)                                          # nothing to continue from yet
    b.restore_link_state()
    for l in range(5):                                                        # recomputed similarities == the ones the builder stored
        ia, na, sa = a.export_level(l, with_sims=True)
        ib, nb, sb = b.export_level(l, with_sims=True)
        assert np.array_equal(ia, ib) and np.array_equal(na, nb)
        live = na != O.SLOT_EMPTY
        assert np.array_equal(sa[live].view(np.uint32), sb[live].view(np.uint32))
    a.append(X[1800:], 128)
    b.append(X[1800:], 128)
    ga, gb = a.export_graph(), b.export_graph()
    assert all(np.array_equal(x[0], y[0]) for x, y in zip(ga, gb))           # the same level draws
    b.delete(np.arange(3, 300, 11, dtype=np.uint32))
    Q = H.queries_from(X, 120, noise=0.05, seed=8)
    ids = b.search_batch(Q, 10, threads=2)[0]
    gt, _ = O.bruteforce_topk(X, Q, 10, threads=2)
    keep = ~np.isin(gt, np.arange(3, 300, 11)).any(axis=1)
    assert np.mean([len(set(ids[i]) & set(gt[i])) / 10 for i in np.nonzero(keep)[0]]) > 0.9
