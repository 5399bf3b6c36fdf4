"""BM25 at a size where the tile directory matters (several 8192-doc tiles, posting lists on both sides of the 256-posting
threshold, lists that end exactly on tile boundaries), the device-output entry point, and the one-call hybrid search
(dense HNSW + BM25 + RRF like repo::hybrid_search, api/vectordb/search/repo.rs:168-341) — all against the oracle composition."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _postings(n_docs, vocab, seed):
    rng = np.random.default_rng(seed)
    terms = np.sort(rng.choice(1 << 31, vocab, replace=False)).astype(np.uint32)
    # document frequencies: a few very long lists, a long tail of short ones, some exactly at the directory threshold
    df = np.clip((n_docs / (1.0 + np.arange(vocab)) ** 1.05).astype(np.int64), 1, n_docs)
    df[10:14] = [256, 257, 255, 8192]
    df = np.minimum(df, n_docs)
    rng.shuffle(df)
    docs, tfs, offsets = [], [], [0]
    for t in range(vocab):
        d = np.sort(rng.choice(n_docs, int(df[t]), replace=False)).astype(np.uint32)
        if t % 7 == 0 and d.size > 2:
            d[0], d[-1] = 0, n_docs - 1                      # first / last document
            d = np.unique(d)
        if t % 11 == 0 and d.size > 4:
            d[1] = 8192 if 8192 < n_docs else d[1]           # exactly on a tile boundary
            d = np.unique(d)
        docs.append(d)
        tfs.append(np.array([O.bm25_tf(int(c), int(dl), 100.0, 1.5, 0.75) for c, dl in zip(rng.integers(1, 6, d.size), rng.integers(20, 300, d.size))], np.float32))
        offsets.append(offsets[-1] + d.size)
    return terms, np.array(offsets, np.uint64), np.concatenate(docs), np.concatenate(tfs)


def _queries(terms, B, seed):
    rng = np.random.default_rng(seed)
    qt, qo = [], [0]
    for _ in range(B):
        m = int(rng.integers(1, 9))
        t = rng.choice(terms, m, replace=False)
        if rng.random() < 0.3:
            t = np.concatenate([t, np.array([12345], np.uint32)])   # a term without a posting list
        qt.append(t.astype(np.uint32))
        qo.append(qo[-1] + t.size)
    return np.concatenate(qt), np.array(qo, np.uint32)


def test_bm25_tile_directory_matches_oracle():
    import torch
    import cosdata_amd as ca
    n_docs, vocab, B, k = 40000, 600, 96, 30
    terms, offsets, docs, tfs = _postings(n_docs, vocab, 3)
    bm = ca.BM25Index(terms, offsets, docs, tfs, n_docs)
    q_terms, q_off = _queries(terms, B, 4)
    ids, sc, cnt = bm.search_batch(q_terms, q_off, k)
    for i in range(B):
        oi, osc = O.bm25_search(terms, offsets, docs, tfs, n_docs, q_terms[q_off[i]:q_off[i + 1]], k)
        c = int(cnt[i])
        assert c == oi.size and np.array_equal(ids[i, :c], oi), i
        assert np.array_equal(sc[i, :c].view(np.uint32), osc.view(np.uint32)), i
    dev = torch.device("cuda:0")
    o_i = torch.zeros(B, k, dtype=torch.int32, device=dev); o_s = torch.zeros(B, k, device=dev); o_c = torch.zeros(B, dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(device=dev)
    bm.search_batch_device(q_terms, q_off, k, o_i.data_ptr(), o_s.data_ptr(), o_c.data_ptr(), st.cuda_stream)
    st.synchronize()
    assert np.array_equal(o_c.cpu().numpy().view(np.uint32), cnt)
    for i in range(B):
        c = int(cnt[i])
        assert np.array_equal(o_i.cpu().numpy().view(np.uint32)[i, :c], ids[i, :c]) and np.array_equal(o_s.cpu().numpy()[i, :c], sc[i, :c])


def test_hybrid_search_one_call_matches_oracle_composition():
    """dense (top_k * 3) and BM25 (top_k * 3) run concurrently on the device, RRF fuses them there: ids / scores equal the
    oracle's dense search + bm25 + rrf_fuse"""
    import cosdata_amd as ca
    n, d, B, k = 6000, 96, 40, 10
    X = H.clustered_corpus(n, d, n_centers=12, seed=6)
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=4, ef_construction=48, ef_search=96)
    dix = H.device_index_from_oracle(oix, X)
    terms, offsets, docs, tfs = _postings(n, 300, 9)
    bm = ca.BM25Index(terms, offsets, docs, tfs, n)
    Q = H.queries_from(X, B, seed=2)
    q_terms, q_off = _queries(terms, B, 5)
    ids, sc, cnt = ca.hybrid_search_batch(dix, bm, Q, q_terms, q_off, k, 60.0)
    od = oix.search_batch(Q, 3 * k, threads=4)
    for i in range(B):
        oi, _ = O.bm25_search(terms, offsets, docs, tfs, n, q_terms[q_off[i]:q_off[i + 1]], 3 * k)
        fi, fs = O.rrf_fuse(od[0][i, :od[2][i]], oi, 60.0, k)
        c = int(cnt[i])
        assert c == fi.size and np.array_equal(ids[i, :c], fi), (i, ids[i, :c], fi)
        assert np.array_equal(sc[i, :c].view(np.uint32), fs.view(np.uint32)), i


def test_hybrid_search_dense_failure_is_reported_and_the_handles_stay_usable():
    """a zero-norm query fails the dense half (CalculationError): the one call reports it (round 6: the dense status comes back in the same
    pinned copy as the fused lists), and the next call on the same handles answers"""
    import cosdata_amd as ca
    n, d, B, k = 3000, 64, 12, 5
    X = H.uniform_corpus(n, d, seed=8)
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=3, ef_construction=32, ef_search=48)
    dix = H.device_index_from_oracle(oix, X)
    terms, offsets, docs, tfs = _postings(n, 200, 4)
    bm = ca.BM25Index(terms, offsets, docs, tfs, n)
    Q = H.queries_from(X, B, seed=2)
    q_terms, q_off = _queries(terms, B, 4)
    good = ca.hybrid_search_batch(dix, bm, Q, q_terms, q_off, k, 60.0)
    Qz = Q.copy()
    Qz[5] = -1.0   # quantizes to all-zero bytes -> |q| = 0
    with pytest.raises(ca.CosdataError) as ei:
        ca.hybrid_search_batch(dix, bm, Qz, q_terms, q_off, k, 60.0)
    assert ei.value.status == 2
    again = ca.hybrid_search_batch(dix, bm, Q, q_terms, q_off, k, 60.0)
    assert all(np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32)) for a, b in zip(good, again))


def test_bm25_many_queries_several_tiles_per_block():
    """a launch with more queries than blocks-per-launch / tiles: a block walks several tiles (reset of the accumulators between
    tiles, the chunk pipeline across a tile boundary), heaviest-first launch order with many ties"""
    import cosdata_amd as ca
    n_docs, vocab, B, k = 40000, 400, 2100, 10
    terms, offsets, docs, tfs = _postings(n_docs, vocab, 13)
    bm = ca.BM25Index(terms, offsets, docs, tfs, n_docs)
    q_terms, q_off = _queries(terms, B, 14)
    ids, sc, cnt = bm.search_batch(q_terms, q_off, k)
    for i in list(range(0, B, 17)) + [B - 1]:
        oi, osc = O.bm25_search(terms, offsets, docs, tfs, n_docs, q_terms[q_off[i]:q_off[i + 1]], k)
        c = int(cnt[i])
        assert c == oi.size and np.array_equal(ids[i, :c], oi), i
        assert np.array_equal(sc[i, :c].view(np.uint32), osc.view(np.uint32)), i
    # the same queries one small batch at a time (one tile per block) give the same answers
    for s0 in (0, 700, 2000):
        i2, s2, c2 = bm.search_batch(q_terms[q_off[s0]:q_off[s0 + 64]], q_off[s0:s0 + 65] - q_off[s0], k)
        assert np.array_equal(c2, cnt[s0:s0 + 64])
        for j in range(64):
            c = int(c2[j])
            assert np.array_equal(i2[j, :c], ids[s0 + j, :c]) and np.array_equal(s2[j, :c].view(np.uint32), sc[s0 + j, :c].view(np.uint32))


def test_bm25_rejects_non_finite_term_frequencies():
    """the accumulators mark 'no term reached this document yet' with a NaN pattern, so stored tfs must be finite — which
    compute_bm25_term_frequency (indexes/tf_idf/mod.rs:362-371) of a count always is"""
    import cosdata_amd as ca
    terms, offsets, docs, tfs = _postings(5000, 50, 2)
    bad = tfs.copy()
    bad[7] = np.nan
    with pytest.raises(ca.CosdataError) as ei:
        ca.BM25Index(terms, offsets, docs, bad, 5000)
    assert ei.value.status == 3  # COS_ERR_INVALID
