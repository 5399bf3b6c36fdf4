"""Metadata-filtered search in the oracle: the reference's own known answer for pseudo_level_probs, the
(node kind, query kind) arms of the cosine dispatch on hand-made cases, and the end-to-end meaning of a filter: only replicas
whose metadata matches come back, scored by the exact cosine of their embedding."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import meta_helpers as MH


def test_pseudo_level_probs_known_answer():
    """metadata/mod.rs:286-301 (the reference's own test)"""
    assert O.pseudo_level_probs(9, 128) == [(0.999, 9), (0.99, 8), (0.9, 7), (0.0, 6), (0.0, 5), (0.0, 4), (0.0, 3), (0.0, 2), (0.0, 1), (0.0, 0)]
    assert O.pseudo_level_probs(2, 1234) == [(0.0, 2), (0.0, 1), (0.0, 0)]          # more digits than levels: every level is "lower"
    assert O.pseudo_level_probs(3, 7) == [(0.9, 3), (0.0, 2), (0.0, 1), (0.0, 0)]


def test_filter_encoding_helper_matches_the_reference_examples():
    """query_filtering.rs:118-127: value 7 in 5 dims -> [0,0,1,1,1]; NotEqual -> [1,1,-1,-1,-1]"""
    enc = lambda v, size, neq: [(-1 if x else 1) for x in MH.bits(v, size)] if neq else MH.bits(v, size)
    assert enc(7, 5, False) == [0, 0, 1, 1, 1] and enc(7, 5, True) == [1, 1, -1, -1, -1]


@pytest.fixture(scope="module")
def world():
    sc = MH.Scenario(n=1200, dim=48, seed=2)
    return sc, sc.oracle()


def test_filtered_results_satisfy_the_filter_and_carry_exact_scores(world):
    sc, oix = world
    Q, off, rows, desc = sc.queries(nq=30, seed=5)
    ids, scores, counts = oix.search_filtered_batch(Q, off, rows, 10, threads=4)
    seen = 0
    for b, (kind, c, s) in enumerate(desc):
        for j in range(int(counts[b])):
            rid = int(ids[b, j])
            r, rep = rid // 4, rid % 4
            assert rid < MH.PSEUDO_ROOT                                       # pseudo nodes never come back (common.rs:400-402)
            ok_color, ok_size = sc.color[r] == c, sc.size[r] == s
            if kind == "is_color":
                assert rep == 1 and ok_color
            elif kind == "is_size":
                assert rep == 2 and ok_size
            elif kind == "and":
                assert rep == 3 and ok_color and ok_size
            elif kind == "or":
                assert (rep == 1 and ok_color) or (rep == 2 and ok_size)
            x, q = sc.X[r].astype(np.float64), Q[b].astype(np.float64)
            assert abs(float(scores[b, j]) - x @ q / np.linalg.norm(x) / np.linalg.norm(q)) < 1e-5
            seen += 1
        if kind in ("is_color", "is_size", "or"):
            assert counts[b] == 10
    assert seen > 150


def test_filtered_walk_lists_drop_strong_mismatches_and_start_at_the_pseudo_root(world):
    sc, oix = world
    Q, off, rows, desc = sc.queries(nq=6, seed=9)
    for b in range(6):
        ids, sims, lc = oix.ann_search_filtered(Q[b], rows[off[b]:off[b + 1]])
        assert lc.sum() == ids.size and (lc >= 1).all() and (lc <= 100).all()
        o = 0
        for c in lc:
            seg = sims[o:o + int(c)]
            assert (np.diff(seg) <= 0).all()                                   # sorted descending
            if int(c) > 1:
                assert not (seg == -1.0).any()                                 # -1.0 only survives through the empty-list fallback
            o += int(c)


def test_base_graph_of_a_metadata_collection_uses_base_ids(world):
    """unfiltered search on the same collection: ids are base ids (multiples of max_replicas), raw rows = id / 4"""
    sc, oix = world
    Q, *_ = sc.queries(nq=8, seed=1)
    ids, scores, counts = oix.search_batch(Q, 10, threads=2)
    assert (counts == 10).all() and (ids % 4 == 0).all() and (ids // 4 < sc.n).all()
    for lid, _ in oix.export_graph():
        assert ((lid[:-1] % 4) == 0).all() and lid[-1] == 0xFFFFFFFF


def _same_component(a, b):
    ga, gb = a.meta_export_graph(), b.meta_export_graph()
    assert len(ga) == len(gb)
    for l, ((ia, na), (ib, nb)) in enumerate(zip(ga, gb)):
        assert np.array_equal(ia, ib), f"level {l}: node sets differ"
        assert np.array_equal(na, nb), f"level {l}: adjacency differs in {np.count_nonzero((na != nb).any(axis=1))} rows"


def test_batch_synchronous_component_builder_with_batches_of_one_is_the_sequential_builder():
    """coso_meta_build_rounds is the schedule a device-side builder of the pseudo-root component would run; with batch_size = 1 every
    node walks the graph all its predecessors are linked into and is alone in its rounds, so it must reproduce
    index_embedding / create_node_edges of the sequential builder (vector_store.rs:714-1074) slot for slot — including the
    edge-refusal rules and the deferred back-edge removals"""
    for seed, kw in ((2, {}), (5, dict(neighbors_count=8, level0_neighbors_count=16)), (7, dict(num_layers=3, ef_construction=16))):
        sc = MH.Scenario(n=500, dim=32, seed=seed, **kw)
        a = sc.oracle()
        b = O.OracleIndex(sc.params).set_vectors(sc.X)
        b.meta_enable(MH.MDIM, MH.REPLICAS)
        b.build()
        b.meta_set_nodes(sc.node_ids, sc.mbits)
        _, st = b.meta_build_rounds(sc.max_levels, 1)
        _same_component(a, b)
        assert st["rounds"] == st["node_levels"]  # one node per round


def test_batch_synchronous_component_builder_is_deterministic_and_serves_filters():
    """bigger batches give a different (batch-synchronous) graph: deterministic, same node sets per level as the sequential one,
    and filtered search on it still returns only matching replicas with full lists for the broad filters"""
    sc = MH.Scenario(n=1500, dim=48, seed=4)
    seq = sc.oracle()
    graphs = []
    for _ in range(2):
        b = O.OracleIndex(sc.params).set_vectors(sc.X)
        b.meta_enable(MH.MDIM, MH.REPLICAS)
        b.build()
        b.meta_set_nodes(sc.node_ids, sc.mbits)
        _, st = b.meta_build_rounds(sc.max_levels, 256)
        graphs.append(b)
    _same_component(graphs[0], graphs[1])
    assert st["rounds"] < st["node_levels"]  # nodes do share rounds
    for (ia, _), (ib, _) in zip(seq.meta_export_graph(), graphs[0].meta_export_graph()):
        assert np.array_equal(ia, ib)          # who lives on which level does not depend on the schedule
    Q, off, rows, desc = sc.queries(nq=30, seed=6)
    ids, scores, counts = graphs[0].search_filtered_batch(Q, off, rows, 10, threads=4)
    ids_s, _, counts_s = seq.search_filtered_batch(Q, off, rows, 10, threads=4)
    overlap = []
    for b, (kind, c, s) in enumerate(desc):
        for j in range(int(counts[b])):
            rid = int(ids[b, j])
            r, rep = rid // 4, rid % 4
            assert rid < MH.PSEUDO_ROOT
            if kind == "is_color":
                assert rep == 1 and sc.color[r] == c
            elif kind == "is_size":
                assert rep == 2 and sc.size[r] == s
            elif kind == "and":
                assert rep == 3 and sc.color[r] == c and sc.size[r] == s
        if kind in ("is_color", "is_size", "or"):
            assert counts[b] == 10
            overlap.append(len(set(ids[b, :10].tolist()) & set(ids_s[b, :10].tolist())) / 10)
    assert np.mean(overlap) >= 0.7, np.mean(overlap)  # both graphs find mostly the same nearest matching replicas
