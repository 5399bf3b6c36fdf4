"""Oracle modes bench.py uses at full size: a corpus quantized in streamed chunks (no raw table on the host) with the rerank rows
handed over per candidate, the bounded top-5k selection of the exhaustive scan, and a second graph imported over the same vectors.
Each must equal the plain in-memory oracle bit for bit."""
import numpy as np
import pytest

from oracle import oracle as O


def _data(n=900, d=48, nq=12, seed=3):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    Q = (X[rng.integers(0, n, nq)] + 0.05 * rng.standard_normal((nq, d))).astype(np.float32)
    return X, Q


@pytest.mark.parametrize("storage,res", [(O.STORAGE_SUBBYTE, 2), (O.STORAGE_U8, 0)])
def test_streamed_flat_search_equals_in_memory_flat_search(storage, res):
    X, Q = _data()
    k = 7
    p = O.HNSWParams(dim=X.shape[1], storage=storage, resolution=res)
    full = O.OracleIndex(p).set_vectors(X)
    ids, sc, cnt = full.flat_search_batch(Q, k, threads=2)
    st = O.OracleIndex(p).alloc_vectors(X.shape[0])
    for s0 in range(0, X.shape[0], 250):
        st.quantize_rows(s0, X[s0:s0 + 250])
    with pytest.raises(ValueError):
        st.flat_search_batch(Q, k)                       # no raw rows yet: refused, never a wrong answer
    cand, ccnt = st.flat_candidates_batch(Q, k, threads=2)
    assert (ccnt == 5 * k).all()
    u = np.unique(cand)
    st.set_raw_subset(u, X[u])
    i2, s2, c2 = st.flat_search_batch(Q, k, threads=2)
    assert np.array_equal(ids, i2) and np.array_equal(sc.view(np.uint32), s2.view(np.uint32)) and np.array_equal(cnt, c2)
    # the bounded selection == sort every (similarity, id) pair desc (larger id first on ties) and truncate
    codes, mags = full.codes(), full.mags()
    qc, qm = O.quantize(Q[0], storage, res)
    sims = np.array([O.distance(O.METRIC_COSINE, storage, res, X.shape[1], qc, qm, codes[i], mags[i])[1] for i in range(X.shape[0])], np.float32)
    order = sorted(range(X.shape[0]), key=lambda i: (float(sims[i]), i), reverse=True)[:5 * k]
    assert cand[0].tolist() == order


def test_second_graph_over_the_same_vectors_replaces_the_first():
    X, Q = _data(700, 32, 10, 5)
    a = O.OracleIndex(O.HNSWParams(dim=32, num_layers=3, ef_construction=32, ef_search=32, seed=1)).set_vectors(X).build()
    b = O.OracleIndex(O.HNSWParams(dim=32, num_layers=3, ef_construction=32, ef_search=32, seed=2)).set_vectors(X).build()
    ra, rb = a.search_batch(Q, 5), b.search_batch(Q, 5)
    host = O.OracleIndex(O.HNSWParams(dim=32, num_layers=3, ef_construction=32, ef_search=32, seed=9)).set_vectors(X)
    for src, ref in ((a, ra), (b, rb), (a, ra)):
        host.import_graph(src.export_graph(), src.root_raw())
        got = host.search_batch(Q, 5)
        assert all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(got[:3], ref[:3]))
