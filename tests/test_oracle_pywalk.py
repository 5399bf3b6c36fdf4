"""Independent restatement of the walk's CONTROL FLOW in pure Python (small cases only), run against the C oracle on
random graphs: traverse_find_nearest (vector_store.rs:1112-1204) and ann_search (vector_store.rs:256-402) written straight
from the reference — BinaryHeap pops, the ef cut-off that discards the (ef+1)-th pop, slot-order scan limited by
shortlist_size, the lossy PerformantFixedSet (ids alias mod 64*M, query id pre-seeded), keep-100 per level, descent through
the best hit's child, results of every level concatenated.  Distances come from the oracle's operator (its arithmetic is
pinned separately in test_oracle_kat.py); everything else here shares no code with cosdata_oracle_hnsw.c."""
import heapq

import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H


def _py_ann_search(levels, codes, mags, qcode, qmag, p, exact=False):
    """levels[l] = (node_ids ascending root last, nbr_ids [n][M_l]); returns (ids, sims, per-level counts), top level first."""
    n_vec = codes.shape[0] - 1                      # row n = root
    row = lambda nid: n_vec if nid == O.ROOT_ID else nid

    def dist(nid):
        rc, v = O.distance(p.metric, p.storage, p.resolution, p.dim, qcode, qmag, codes[row(nid)], mags[row(nid)])
        assert rc == O.OK
        return np.float32(v)

    out_ids, out_sims, counts = [], [], []
    entry = O.ROOT_ID
    for level in range(p.num_layers, -1, -1):
        ids, nbr = levels[level]
        pos = {int(v): i for i, v in enumerate(ids.tolist())}
        M = p.level0_neighbors_count if level == 0 else p.neighbors_count
        bits = 64 * M
        key = (lambda v: v) if exact else (lambda v: v & (bits - 1))      # fixedset.rs: plain id bits, no hashing
        visited = set() if exact else {key(O.QUERY_ID)}                   # vector_store.rs:266-271
        heap = []                                                         # max-heap on (sim, id): larger id wins ties
        s0 = dist(entry)
        visited.add(key(entry))
        heapq.heappush(heap, (-float(s0), -entry, entry, s0))
        popped = []
        while heap:
            _, _, node, s = heapq.heappop(heap)
            if len(popped) >= p.ef_search:                                # :1151-1153 — the popped element is discarded
                break
            popped.append((s, node))
            for j in range(min(M, p.shortlist_size)):                     # :1161-1165 slot order
                nid = int(nbr[pos[node], j])
                if nid == O.SLOT_EMPTY or key(nid) in visited:
                    continue
                d = dist(nid)
                visited.add(key(nid))
                heapq.heappush(heap, (-float(d), -nid, nid, d))
        popped.sort(key=lambda t: (float(t[0]), t[1]), reverse=True)      # :1194-1201 keep the best 100, sorted
        popped = popped[:100]
        if not popped:                                                    # :329-380 (ef == 0)
            popped = [(dist(entry), entry)]
        out_ids += [t[1] for t in popped]
        out_sims += [t[0] for t in popped]
        counts.append(len(popped))
        entry = popped[0][1]                                              # child link = same id one level down
    return np.array(out_ids, np.uint32), np.array(out_sims, np.float32), np.array(counts, np.uint32)


@pytest.mark.parametrize("storage,res", [(O.STORAGE_U8, 0), (O.STORAGE_SUBBYTE, 2), (O.STORAGE_F32, 0)])
@pytest.mark.parametrize("M,M0,ef,shortlist,exact", [(8, 16, 12, 64, False), (4, 8, 40, 6, False), (16, 32, 7, 20, False),
                                                     (8, 16, 25, 64, True), (2, 4, 100, 64, False)])
def test_python_walk_equals_c_oracle(storage, res, M, M0, ef, shortlist, exact):
    rng = np.random.default_rng(M * 131 + ef)
    X = H.clustered_corpus(500, 24, n_centers=7, seed=M0 + ef)
    p = O.HNSWParams(dim=24, storage=storage, resolution=res, num_layers=3, neighbors_count=M, level0_neighbors_count=M0,
                     ef_construction=24, ef_search=ef, shortlist_size=shortlist,
                     visited_mode=O.VISITED_EXACT if exact else O.VISITED_REF, seed=5)
    ix = O.OracleIndex(p).set_vectors(X).build()
    levels = ix.export_graph()
    codes, mags = O.quantize_batch(np.vstack([X, ix.root_raw()[None, :]]), storage, res, -1.0, 1.0)
    Q = np.concatenate([H.queries_from(X, 6, seed=ef), rng.uniform(-1, 1, (2, 24)).astype(np.float32)])
    for q in Q:
        qcode, qmag = O.quantize(q, storage, res, -1.0, 1.0)
        pi, ps, pc = _py_ann_search(levels, codes, mags, qcode, qmag, p, exact)
        ci, cs, cc = ix.ann_search(q)
        assert np.array_equal(pc, cc), (pc, cc)
        assert np.array_equal(pi, ci)
        assert np.array_equal(ps.view(np.uint32), cs.view(np.uint32))
