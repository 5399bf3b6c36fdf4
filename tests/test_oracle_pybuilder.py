"""Independent restatement of graph CONSTRUCTION in pure Python (small cases), run against the C oracle's sequential
builder: index_embedding (vector_store.rs:782-937: walk with ef_construction and the new id pre-seeded in the visited filter,
keep 64, lower levels linked before this level's edges), create_node_edges (:976-1074: both directions, at most M successful
edges, roll the forward edge back if the backward one is refused) and ProbNode::add_neighbor (prob_node.rs:210-283: the
cached (lowest_idx, lowest_sim), replace-if-better, rescan starting from MetricResult::max = 2.0, the evictee drops its back
edge and its cache is NOT refreshed) — written from the reference source, sharing no code with cosdata_oracle_hnsw.c.
Level draws and the root vector are taken from the oracle's build (the reference uses thread_rng there)."""
import heapq

import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H

MIN_COS, MAX_COS = np.float32(-1.0), np.float32(2.0)   # MetricResult::min / ::max (types.rs:435-457)


class Node:
    __slots__ = ("id", "nbr", "low")

    def __init__(self, nid, M):
        self.id = nid
        self.nbr = [None] * M                 # slot -> (neighbour id, sim) | None        (prob_node.rs:97)
        self.low = (0, MIN_COS)               # lowest_index cache                        (prob_node.rs:140)


def add_neighbor(levelmap, node, nbr_id, dist):
    lowest_idx, lowest_sim = node.low
    if dist <= lowest_sim:                                                   # :224
        return None
    cur = node.nbr[lowest_idx]
    ok = cur is None or dist > cur[1]                                        # fetch_update closure :233-241
    old = None
    if ok:
        old, node.nbr[lowest_idx] = cur, (nbr_id, dist)
    new_idx, new_sim = 0, MAX_COS                                            # :243-257
    for idx, e in enumerate(node.nbr):
        if e is None:
            new_idx, new_sim = idx, MIN_COS
            break
        if e[1] < new_sim:
            new_idx, new_sim = idx, e[1]
    node.low = (new_idx, new_sim)
    if not ok:
        return None
    if old is not None:                                                      # :263-271 remove_neighbor_by_id(self.id) on the evictee
        ev = levelmap[old[0]]
        for j, e in enumerate(ev.nbr):
            if e is not None and e[0] == node.id:
                ev.nbr[j] = None
                break
    return lowest_idx


def create_node_edges(levelmap, node, z, max_edges):
    succ = 0
    for nbr_id, dist in z:
        if succ >= max_edges:
            break
        nb = levelmap[nbr_id]
        idx = add_neighbor(levelmap, node, nbr_id, dist)
        if idx is not None:
            if add_neighbor(levelmap, nb, node.id, dist) is not None:
                succ += 1
            else:                                                            # remove_neighbor_by_index_and_id
                e = node.nbr[idx]
                if e is not None and e[0] == nbr_id:
                    node.nbr[idx] = None


def build_py(X, root_raw, max_level, p):
    n = X.shape[0]
    codes, mags = O.quantize_batch(np.vstack([X, root_raw[None, :]]), p.storage, p.resolution, p.range_lo, p.range_hi)
    row = lambda nid: n if nid == O.ROOT_ID else nid
    L = p.num_layers
    Ml = lambda l: p.level0_neighbors_count if l == 0 else p.neighbors_count
    levels = [{O.ROOT_ID: Node(O.ROOT_ID, Ml(l))} for l in range(L + 1)]

    def dist(a, b):
        rc, v = O.distance(p.metric, p.storage, p.resolution, p.dim, codes[row(a)], mags[row(a)], codes[row(b)], mags[row(b)])
        assert rc == O.OK
        return np.float32(v)

    def traverse(level, new_id, entry):                                      # traverse_find_nearest, is_indexing = true
        M = Ml(level)
        mask = 64 * M - 1
        visited = {new_id & mask}                                            # skipm.insert(new_node_id) :803-807
        heap, results = [], []
        s0 = dist(new_id, entry)
        visited.add(entry & mask)
        heapq.heappush(heap, (-float(s0), -entry, entry, s0))
        while heap:
            _, _, node, s = heapq.heappop(heap)
            if len(results) >= p.ef_construction:
                break
            results.append((s, node))
            for e in levels[level][node].nbr[:min(M, p.shortlist_size)]:
                if e is None or (e[0] & mask) in visited:
                    continue
                d = dist(new_id, e[0])
                visited.add(e[0] & mask)
                heapq.heappush(heap, (-float(d), -e[0], e[0], d))
        results.sort(key=lambda t: (float(t[0]), t[1]), reverse=True)
        return [(nid, s) for s, nid in results[:64]]

    def index_embedding(new_id, level, entry, mlev):
        z = traverse(level, new_id, entry)
        if not z:
            z = [(entry, dist(new_id, entry))]
        child = z[0][0]                                                      # same id one level down
        if level > mlev:
            if level != 0:
                index_embedding(new_id, level - 1, child, mlev)
            return
        node = Node(new_id, Ml(level))
        levels[level][new_id] = node
        if level != 0:
            index_embedding(new_id, level - 1, child, mlev)                  # lower levels are linked first
        create_node_edges(levels[level], node, z, Ml(level))

    for nid in range(n):
        index_embedding(nid, L, O.ROOT_ID, int(max_level[nid]))
    return levels


@pytest.mark.parametrize("storage,res", [(O.STORAGE_U8, 0), (O.STORAGE_SUBBYTE, 2)])
@pytest.mark.parametrize("M,M0,efc,shortlist", [(8, 16, 24, 64), (4, 8, 12, 64), (16, 32, 40, 10)])
def test_python_builder_equals_c_oracle(storage, res, M, M0, efc, shortlist):
    X = H.clustered_corpus(260, 20, n_centers=5, seed=M + efc)
    p = O.HNSWParams(dim=20, storage=storage, resolution=res, num_layers=3, neighbors_count=M, level0_neighbors_count=M0,
                     ef_construction=efc, ef_search=16, shortlist_size=shortlist, seed=11)
    ix = O.OracleIndex(p).set_vectors(X).build()
    graph = ix.export_graph()
    max_level = np.zeros(260, np.int64)
    for l, (ids, _) in enumerate(graph):
        for v in ids.tolist():
            if v != O.ROOT_ID:
                max_level[v] = max(max_level[v], l)
    py = build_py(X, ix.root_raw(), max_level, p)
    for l, (ids, nbr) in enumerate(graph):
        assert sorted(py[l].keys()) == ids.tolist()
        for i, nid in enumerate(ids.tolist()):
            want = [O.SLOT_EMPTY if e is None else e[0] for e in py[l][nid].nbr]
            assert nbr[i].tolist() == want, (l, nid)
