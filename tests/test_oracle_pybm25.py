"""Independent pure-Python restatement of SparseAnnQueryBasic::search_bm25 (models/sparse_ann_query.rs:149-233), written
from the reference source: a heap of posting-list heads ordered by their next doc id, document-at-a-time accumulation of
tf*idf, 512 buckets keyed doc_id % 512 that keep the strictly greater score, sort by total_cmp desc, truncate k — against
the C oracle (which the device is held to).  The two documented choices where the reference is unordered are applied here
too: heads with the same doc id are drained in ascending term-hash order, and equal final scores go to the larger doc id."""
import heapq

import numpy as np
import pytest

from oracle import oracle as O


def _py_search_bm25(terms, offsets, docs, tfs, n_docs, q_terms, k):
    tmap = {int(t): i for i, t in enumerate(terms.tolist())}
    heads = []                                   # (next doc id, term hash, entry order, cursor, end, idf)
    for order, th in enumerate(q_terms.tolist()):
        ti = tmap.get(int(th))
        if ti is None:
            continue                             # find_node / lookup miss: the term contributes nothing
        lo, hi = int(offsets[ti]), int(offsets[ti + 1])
        if lo == hi:
            continue
        idf = O.bm25_idf(n_docs, hi - lo)        # get_idf(documents_count, documents.len())  (:298-302)
        heapq.heappush(heads, (int(docs[lo]), int(th), order, lo, hi, idf))
    buckets = [(0xFFFFFFFF, np.float32(-np.inf))] * 512
    while heads:
        doc, th, order, cur, end, idf = heapq.heappop(heads)
        score = np.float32(tfs[cur]) * idf
        if cur + 1 < end:
            heapq.heappush(heads, (int(docs[cur + 1]), th, order, cur + 1, end, idf))
        while heads and heads[0][0] == doc:
            _, th2, o2, c2, e2, idf2 = heapq.heappop(heads)
            score = np.float32(score + np.float32(tfs[c2]) * idf2)
            if c2 + 1 < e2:
                heapq.heappush(heads, (int(docs[c2 + 1]), th2, o2, c2 + 1, e2, idf2))
        b = doc % 512
        if score > buckets[b][1]:
            buckets[b] = (doc, score)
    res = [(s, d) for d, s in buckets if d != 0xFFFFFFFF]
    res.sort(key=lambda t: (float(t[0]), t[1]), reverse=True)
    res = res[:k]
    return np.array([d for _, d in res], np.uint32), np.array([s for s, _ in res], np.float32)


@pytest.mark.parametrize("seed", range(6))
def test_python_bm25_equals_c_oracle(seed):
    rng = np.random.default_rng(seed)
    n_docs, T = int(rng.integers(600, 3000)), 120
    terms = np.sort(rng.choice(1 << 31, T, replace=False).astype(np.uint32))
    lens = np.minimum(rng.zipf(1.4, T) * 2, n_docs // 2).astype(np.int64)
    offsets = np.zeros(T + 1, np.uint64)
    offsets[1:] = np.cumsum(lens)
    docs = np.concatenate([np.sort(rng.choice(n_docs, int(l), replace=False)) for l in lens]).astype(np.uint32)
    tfs = np.array([O.bm25_tf(int(c), int(dl), 120.0, 1.5, 0.75) for c, dl in
                    zip(rng.integers(1, 6, docs.size), rng.integers(40, 260, docs.size))], np.float32)
    for _ in range(12):
        m = int(rng.integers(1, 9))
        q = rng.choice(terms, m, replace=True).astype(np.uint32)      # duplicates allowed: each entry is its own head
        if rng.random() < 0.3:
            q[0] = 7                                                   # a term without a posting list
        for k in (1, 10, 30):
            pi, ps = _py_search_bm25(terms, offsets, docs, tfs, n_docs, q, k)
            ci, cs = O.bm25_search(terms, offsets, docs, tfs, n_docs, q, k)
            assert np.array_equal(pi, ci), (q, k)
            assert np.array_equal(ps.view(np.uint32), cs.view(np.uint32)), (q, k)
