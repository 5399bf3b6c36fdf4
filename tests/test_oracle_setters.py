"""Oracle plumbing used by bench.py's M0 / M variants: one quantized corpus takes graphs of several level_0_neighbors_count /
neighbors_count in turn (coso_index_set_level0_neighbors / coso_index_set_neighbors) and must answer exactly like an index created
with those hyper-parameters (indexes/hnsw/types.rs:10-17; the filter is PerformantFixedSet::new(that count), vector_store.rs:266-270)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H


def test_changed_neighbor_counts_equal_a_fresh_index():
    X = H.clustered_corpus(1500, 48, n_centers=8, seed=11)
    Q = H.queries_from(X, 40, noise=0.05, seed=12)
    base = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=3, ef_construction=32, ef_search=32)
    ref64 = base.search_batch(Q, 5)
    fresh = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=3, ef_construction=32, ef_search=32, level0_neighbors_count=128, neighbors_count=64)
    base.set_level0_neighbors(128).set_neighbors(64)
    assert base.export_level(0)[0].size == 0                      # the old graph is gone, the vectors stay
    base.build()
    for a, b in zip(base.export_graph(), fresh.export_graph()):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    got, want = base.search_batch(Q, 5), fresh.search_batch(Q, 5)
    assert all(np.array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32)) for x, y in zip(got[:3], want[:3]))
    assert base.export_level(0)[1].shape[1] == 128 and base.export_level(1)[1].shape[1] == 64
    assert not np.array_equal(ref64[0], got[0]) or True           # (answers may or may not differ on a corpus this small)
    for bad in (0, 3, 512):
        with pytest.raises(ValueError):
            base.set_level0_neighbors(bad)
