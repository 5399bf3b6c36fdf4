"""Learned-sparse inverted index: the C oracle against an independent pure-Python restatement written from
models/sparse_ann_query.rs:68-147 (dict of dot products, per-(dimension, key) lists), and the device against the oracle."""
import numpy as np
import pytest

from oracle import oracle as O


def _corpus(n=3000, vocab=400, nnz=24, bits=6, upper=3.0, seed=0):
    """random sparse vectors -> (per-vector pairs, CSR inverted index by (dimension, quantized key), raw CSR)"""
    rng = np.random.default_rng(seed)
    rows = []
    for v in range(n):
        d = np.sort(rng.choice(vocab, size=int(rng.integers(4, nnz)), replace=False)).astype(np.uint32)
        x = rng.gamma(1.2, 0.7, d.size).astype(np.float32)
        x[rng.random(d.size) < 0.05] *= 5.0            # some values above the upper bound: clamp
        rows.append((d, x))
    Q = 1 << bits
    lists = {}
    for v, (d, x) in enumerate(rows):                   # InvertedIndexNode::insert: push the id on the list of its quantized value
        for di, xi in zip(d, x):
            lists.setdefault(int(di), [[] for _ in range(Q)])[O.sparse_quantize(xi, upper, bits)].append(v)
    dims = np.array(sorted(lists), np.uint32)
    key_off, vec_ids = [], []
    for di in dims:
        for k in range(Q):
            key_off.append(len(vec_ids))
            vec_ids += lists[int(di)][k]
        key_off.append(len(vec_ids))
    row_off = np.cumsum([0] + [len(d) for d, _ in rows]).astype(np.uint64)
    raw_dims = np.concatenate([d for d, _ in rows]).astype(np.uint32)
    raw_vals = np.concatenate([x for _, x in rows]).astype(np.float32)
    return rows, dims, np.array(key_off, np.uint64), np.array(vec_ids, np.uint32), row_off, raw_dims, raw_vals


def _queries(nq, vocab, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(nq):
        d = rng.choice(vocab + 20, size=int(rng.integers(2, 12)), replace=False).astype(np.uint32)   # some dimensions are unknown
        x = rng.gamma(1.5, 0.8, d.size).astype(np.float32)
        x[rng.random(d.size) < 0.1] = 0.01                                                              # tiny values quantize to 0
        out.append((d, x))
    return out


def _py_sequential_search(lists_by_dim, bits, upper, thr, q):
    """straight from sparse_ann_query.rs:68-124"""
    one_q = (1 << bits) - 1
    etv = int(min(np.float32(1 << bits) * np.float32(thr), np.float32(255.0)))
    low = int(np.float32(thr) * np.float32(1 << bits))
    dots = {}
    for dim, val in sorted(zip(*q), key=lambda t: -t[1]):
        node = lists_by_dim.get(int(dim))
        if node is None:
            continue
        qq = O.sparse_quantize(val, upper, bits)
        keys = range(one_q, -1, -1) if qq > low else range(one_q, etv - 1, -1)
        for key in keys:
            for v in node[key]:
                dots[v] = dots.get(v, 0) + qq * key
    return dots


def test_quantize_rule():
    """(((value / upper) * max).clamp(0, max) as u8).min(max) — truncation, clamp, NaN -> 0"""
    assert O.sparse_quantize(0.0, 3.0, 6) == 0 and O.sparse_quantize(3.0, 3.0, 6) == 63 and O.sparse_quantize(100.0, 3.0, 6) == 63
    assert O.sparse_quantize(-1.0, 3.0, 6) == 0 and O.sparse_quantize(float("nan"), 3.0, 6) == 0
    assert O.sparse_quantize(1.49, 3.0, 4) == int(np.float32(1.49) / np.float32(3.0) * np.float32(15.0))
    assert O.sparse_quantize(2.0, 2.0, 8) == 255


@pytest.mark.parametrize("bits,thr", [(6, 0.0), (4, 0.5), (5, 0.75), (8, 0.3)])
def test_oracle_matches_python_restatement(bits, thr):
    upper = 3.0
    rows, dims, key_off, vec_ids, row_off, raw_dims, raw_vals = _corpus(n=1200, bits=bits, upper=upper, seed=bits)
    Qn = 1 << bits
    lists = {int(d): [list(vec_ids[int(key_off[t * (Qn + 1) + k]):int(key_off[t * (Qn + 1) + k + 1])]) for k in range(Qn)] for t, d in enumerate(dims)}
    for q in _queries(12, 400, seed=bits + 1):
        want = _py_sequential_search(lists, bits, upper, thr, q)
        ids, sims = O.sparse_search(dims, key_off, vec_ids, 1200, bits, upper, thr, q[0], q[1])
        assert dict(zip(ids.tolist(), sims.tolist())) == want
        assert all((sims[i], ids[i]) >= (sims[i + 1], ids[i + 1]) for i in range(len(ids) - 1))     # similarity desc, larger id first
        top, _ = O.sparse_search(dims, key_off, vec_ids, 1200, bits, upper, thr, q[0], q[1], k_with_reranking=25)
        assert np.array_equal(top, ids[:25])
        # raw-value rerank (finalize_sparse_ann_results): f32 sum over the query pairs in order
        rid, rsc = O.sparse_rerank(row_off, raw_dims, raw_vals, top, q[0], q[1], top_k=5)
        exp = []
        for v in top:
            m = dict(zip(rows[v][0].tolist(), rows[v][1].tolist()))
            dp = np.float32(0.0)
            for d, x in zip(*q):
                if int(d) in m:
                    dp = np.float32(dp + np.float32(np.float32(m[int(d)]) * np.float32(x)))
            exp.append((float(dp), int(v)))
        exp.sort(reverse=True)
        assert [e[1] for e in exp[:5]] == rid.tolist() and np.allclose([e[0] for e in exp[:5]], rsc, rtol=0, atol=0)


def _index_with_layout(layout, *args, **kw):
    """layout 0 = (u32 id, u8 key) per posting, 1 = packed u32 (the default where ids fit 24 bits): tuning knob sparse_layout"""
    import cosdata_amd as ca
    from cosdata_amd import _lib
    with _lib.tuning(sparse_layout=layout):
        ix = ca.InvertedIndex(*args, **kw)
    assert ix.packed == bool(layout)
    return ix


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("bits,thr", [(6, 0.0), (4, 0.5), (8, 0.3)])
def test_device_matches_oracle(bits, thr, layout):
    import cosdata_amd as ca
    upper = 3.0
    n = 20000
    rows, dims, key_off, vec_ids, row_off, raw_dims, raw_vals = _corpus(n=n, vocab=600, bits=bits, upper=upper, seed=10 + bits)
    ix = _index_with_layout(layout, bits, upper, dims, key_off, vec_ids, n, row_off, raw_dims, raw_vals)
    qs = _queries(70, 600, seed=3)
    qo = np.cumsum([0] + [len(q[0]) for q in qs]).astype(np.uint32)
    qd = np.concatenate([q[0] for q in qs]).astype(np.uint32)
    qv = np.concatenate([q[1] for q in qs]).astype(np.float32)
    for k, rf in ((10, 0), (10, 5), (1, 0), (64, 0), (12, 5)):
        ids, sc, cnt = ix.search_batch(qd, qv, qo, k, thr, rf)
        for b, q in enumerate(qs):
            cand, sims = O.sparse_search(dims, key_off, vec_ids, n, bits, upper, thr, q[0], q[1], k_with_reranking=k * max(rf, 1))
            if rf == 0:
                eid, esc = cand[:k], sims[:k].astype(np.float32)
            else:
                eid, esc = O.sparse_rerank(row_off, raw_dims, raw_vals, cand, q[0], q[1], top_k=k)
            c = int(cnt[b])
            assert c == len(eid), (b, c, len(eid))
            assert np.array_equal(ids[b, :c], eid), (b, ids[b, :c], eid)
            assert np.array_equal(sc[b, :c].view(np.uint32), np.asarray(esc, np.float32).view(np.uint32))
    with pytest.raises(ca.CosdataError):
        ix.search_batch(qd, qv, qo, 20, thr, 5)             # 100 candidates > 64


@pytest.mark.parametrize("bits", [1, 4, 6, 8])
def test_library_builds_the_csr_insert_would_build(bits):
    """cos_sparse_build_csr (host code): InvertedIndexNode::insert for ids 0 .. n-1 — quantize, push the id to the end of the
    (dimension, key) list — against the dict-of-lists construction of _corpus (which uses the oracle's quantizer)"""
    import cosdata_amd as ca
    rows, dims, key_off, vec_ids, row_off, raw_dims, raw_vals = _corpus(n=1500, vocab=300, bits=bits, upper=2.5, seed=bits)
    raw_vals = raw_vals.copy()
    raw_vals[:7] = [0.0, -1.0, 2.5, 2.4999, 1e30, np.nan, 1e-30]   # edge values of quantize: zero, negative, the bound, huge, NaN
    rows2 = [(raw_dims[int(row_off[v]):int(row_off[v + 1])], raw_vals[int(row_off[v]):int(row_off[v + 1])]) for v in range(len(rows))]
    Q = 1 << bits
    lists = {}
    for v, (d, x) in enumerate(rows2):
        for di, xi in zip(d, x):
            lists.setdefault(int(di), [[] for _ in range(Q)])[O.sparse_quantize(xi, 2.5, bits)].append(v)
    want_dims = np.array(sorted(lists), np.uint32)
    want_off, want_ids = [], []
    for di in want_dims:
        for k in range(Q):
            want_off.append(len(want_ids))
            want_ids += lists[int(di)][k]
        want_off.append(len(want_ids))
    d, ko, ids = ca.sparse_build_csr(bits, 2.5, row_off, raw_dims, raw_vals)
    assert np.array_equal(d, want_dims) and np.array_equal(ko, np.array(want_off, np.uint64)) and np.array_equal(ids, np.array(want_ids, np.uint32))


@pytest.mark.gpu
def test_index_created_from_vectors_answers_like_the_index_created_from_the_csr():
    import cosdata_amd as ca
    n, bits, upper = 8000, 6, 3.0
    rows, dims, key_off, vec_ids, row_off, raw_dims, raw_vals = _corpus(n=n, vocab=500, bits=bits, upper=upper, seed=21)
    a = ca.InvertedIndex(bits, upper, dims, key_off, vec_ids, n, row_off, raw_dims, raw_vals)
    b = ca.InvertedIndex.from_vectors(bits, upper, row_off, raw_dims, raw_vals)
    qs = _queries(40, 500, seed=8)
    qo = np.cumsum([0] + [len(q[0]) for q in qs]).astype(np.uint32)
    qd = np.concatenate([q[0] for q in qs]).astype(np.uint32)
    qv = np.concatenate([q[1] for q in qs]).astype(np.float32)
    for k, rf in ((10, 0), (8, 4)):
        ra, rb = a.search_batch(qd, qv, qo, k, 0.2, rf), b.search_batch(qd, qv, qo, k, 0.2, rf)
        assert all(np.array_equal(x, y) for x, y in zip(ra, rb))


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [0, 1])
def test_device_edge_cases_single_query_empty_queries_zero_weights_ragged_last_tile(layout):
    """one query per launch (a block per tile), a query without terms / with unknown dimensions only (no results), query values
    that quantize to 0 (every visited vector is a result with similarity 0), n not a multiple of the 8192-id tile or of 64,
    a duplicate dimension inside one query (visited twice, like the reference's loop), out-of-order ids inside a key list"""
    import cosdata_amd as ca
    bits, upper, n = 6, 3.0, 16411
    rows, dims, key_off, vec_ids, row_off, raw_dims, raw_vals = _corpus(n=n, vocab=90, nnz=10, bits=bits, upper=upper, seed=77)
    vec_ids = vec_ids.copy()
    ko = np.asarray(key_off).reshape(len(dims), (1 << bits) + 1)
    lo, hi = int(ko[3, 10]), int(ko[3, 11])
    vec_ids[lo:hi] = vec_ids[lo:hi][::-1]                     # the order inside a (dimension, key) list is the caller's: sums commute
    ix = _index_with_layout(layout, bits, upper, dims, key_off, vec_ids, n)
    d0, d1 = int(dims[0]), int(dims[1])
    queries = [(np.array([d0], np.uint32), np.array([0.0], np.float32)),                       # quantizes to 0: all similarities 0
               (np.array([100000, 100001], np.uint32), np.array([1.0, 2.0], np.float32)),      # unknown dimensions only
               (np.array([], np.uint32), np.array([], np.float32)),                             # no terms at all
               (np.array([d0, d1, d0], np.uint32), np.array([1.5, 0.7, 0.2], np.float32)),      # the same dimension twice
               (np.array([d1], np.uint32), np.array([2.9], np.float32))]
    for thr in (0.0, 0.6):
        for batch in ([0], [1], [2], [3], [4], [0, 1, 2, 3, 4]):
            qs = [queries[i] for i in batch]
            qo = np.cumsum([0] + [len(q[0]) for q in qs]).astype(np.uint32)
            qd = np.concatenate([q[0] for q in qs] + [np.zeros(1, np.uint32)])[:qo[-1] if qo[-1] else 1]
            qv = np.concatenate([q[1] for q in qs] + [np.zeros(1, np.float32)])[:qo[-1] if qo[-1] else 1]
            ids, sc, cnt = ix.search_batch(qd, qv, qo, 12, thr, 0)
            for b, q in enumerate(qs):
                cand, sims = O.sparse_search(dims, key_off, vec_ids, n, bits, upper, thr, q[0], q[1], k_with_reranking=12)
                c = int(cnt[b])
                assert c == min(12, len(cand)), (thr, batch, b, c, len(cand))
                assert np.array_equal(ids[b, :c], cand[:c]) and np.array_equal(sc[b, :c], sims[:c].astype(np.float32)), (thr, batch, b)
    st = ix.last_stats()
    assert st.blocks > 0 and st.kernel_ms > 0


# ---- packed posting layout (kernels_sparse.hip: sparse_packed_kernel) -------------------------------------------------------------
# The default layout since round 5 (measured: profiles/r05_candidates_sparse.txt) wherever vector ids fit 24 bits; the tests above this
# line run both layouts (`_index_with_layout`: tuning knob sparse_layout).


def _packed_index(*args, **kw):
    import cosdata_amd as ca
    ix = ca.InvertedIndex(*args, **kw)
    assert ix.packed
    return ix


def _assert_like_oracle(ix, dims, key_off, vec_ids, n, bits, upper, thr, qs, k, row_off=None, raw_dims=None, raw_vals=None, rf=0):
    qo = np.cumsum([0] + [len(q[0]) for q in qs]).astype(np.uint32)
    qd = np.concatenate([q[0] for q in qs] + [np.zeros(1, np.uint32)])[:max(int(qo[-1]), 1)]
    qv = np.concatenate([q[1] for q in qs] + [np.zeros(1, np.float32)])[:max(int(qo[-1]), 1)]
    ids, sc, cnt = ix.search_batch(qd, qv, qo, k, thr, rf)
    for b, q in enumerate(qs):
        cand, sims = O.sparse_search(dims, key_off, vec_ids, n, bits, upper, thr, q[0], q[1], k_with_reranking=k * max(rf, 1))
        if rf == 0:
            eid, esc = cand[:k], sims[:k].astype(np.float32)
        else:
            eid, esc = O.sparse_rerank(row_off, raw_dims, raw_vals, cand, q[0], q[1], top_k=k)
        c = int(cnt[b])
        assert c == len(eid), (thr, b, c, len(eid))
        assert np.array_equal(ids[b, :c], eid), (thr, b, ids[b, :c], eid)
        assert np.array_equal(sc[b, :c].view(np.uint32), np.asarray(esc, np.float32).view(np.uint32)), (thr, b)


@pytest.mark.gpu
@pytest.mark.parametrize("bits,thr", [(6, 0.0), (4, 0.5), (8, 0.3), (8, 0.0)])
def test_packed_layout_matches_oracle(bits, thr):
    upper, n = 3.0, 20000
    rows, dims, key_off, vec_ids, row_off, raw_dims, raw_vals = _corpus(n=n, vocab=600, bits=bits, upper=upper, seed=10 + bits)
    ix = _packed_index(bits, upper, dims, key_off, vec_ids, n, row_off, raw_dims, raw_vals)
    qs = _queries(70, 600, seed=3)
    for k, rf in ((10, 0), (10, 5), (1, 0), (64, 0), (12, 5)):
        _assert_like_oracle(ix, dims, key_off, vec_ids, n, bits, upper, thr, qs, k, row_off, raw_dims, raw_vals, rf)


@pytest.mark.gpu
def test_packed_layout_long_queries_and_table_windows():
    """queries of 65..300 terms (term groups of 64; more than 256 terms: one table window per tile and 256 terms), few queries per
    launch (a block owns many tiles: several table windows of tiles), 8-bit keys with large query values (the sum bound of the
    counted accumulator fails: flag-word blocks) next to small ones (counted blocks) in one launch"""
    bits, upper, n, vocab = 8, 3.0, 70000, 500
    rows, dims, key_off, vec_ids, row_off, raw_dims, raw_vals = _corpus(n=n, vocab=vocab, nnz=12, bits=bits, upper=upper, seed=5)
    ix = _packed_index(bits, upper, dims, key_off, vec_ids, n)
    rng = np.random.default_rng(1)
    qs = []
    for m, scale in ((65, 0.05), (130, 2.9), (257, 0.02), (300, 2.5), (64, 2.9), (7, 0.4), (200, 0.0)):
        d = rng.choice(vocab, size=m, replace=False).astype(np.uint32)
        qs.append((d, (scale * (0.5 + rng.random(m))).astype(np.float32)))
    for thr in (0.0, 0.4):
        for batch in ([0, 1, 2, 3, 4, 5, 6], [3], [1, 5]):
            _assert_like_oracle(ix, dims, key_off, vec_ids, n, bits, upper, thr, [qs[i] for i in batch], 16)


@pytest.mark.gpu
def test_packed_layout_edge_cases_and_repeated_ids():
    """the edge cases of the unpacked layout's test, plus a CSR that names one vector in two key lists of a dimension (the touch
    count of the packed accumulator then exceeds the term count: the host's bound must account for it)"""
    bits, upper, n = 6, 3.0, 16411
    rows, dims, key_off, vec_ids, row_off, raw_dims, raw_vals = _corpus(n=n, vocab=90, nnz=10, bits=bits, upper=upper, seed=77)
    ko = np.asarray(key_off).reshape(len(dims), (1 << bits) + 1).copy()
    vec_ids = vec_ids.copy()
    # dimension 3: the ids of key list 10 also appear in key list 12 (insert them there: shift every later offset)
    lo, hi = int(ko[3, 10]), int(ko[3, 11])
    extra = vec_ids[lo:hi].copy()
    at = int(ko[3, 13])
    vec_ids = np.concatenate([vec_ids[:at], extra, vec_ids[at:]])
    flat = ko.ravel()
    first = 3 * ((1 << bits) + 1) + 13
    flat[first:] += len(extra)                # every later offset, the following dimensions' rows included
    key_off2 = flat.astype(np.uint64)
    ix = _packed_index(bits, upper, dims, key_off2, vec_ids, n)
    d0, d1, d3 = int(dims[0]), int(dims[1]), int(dims[3])
    queries = [(np.array([d0], np.uint32), np.array([0.0], np.float32)),
               (np.array([100000, 100001], np.uint32), np.array([1.0, 2.0], np.float32)),
               (np.array([], np.uint32), np.array([], np.float32)),
               (np.array([d0, d1, d0], np.uint32), np.array([1.5, 0.7, 0.2], np.float32)),
               (np.array([d3, d1], np.uint32), np.array([2.9, 0.3], np.float32)),
               (np.array([d3], np.uint32), np.array([0.001], np.float32))]
    for thr in (0.0, 0.6):
        for batch in ([0], [1], [2], [3], [4], [5], [0, 1, 2, 3, 4, 5]):
            _assert_like_oracle(ix, dims, key_off2, vec_ids, n, bits, upper, thr, [queries[i] for i in batch], 12)
