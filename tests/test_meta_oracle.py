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
