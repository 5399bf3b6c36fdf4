"""A small collection with a metadata schema, in the numeric form the engine takes.

Schema (what metadata/schema.rs would encode): field `color` with values 1..3 (2 binary dims), field `size` with values 1..5
(3 dims), supported conditions AND(color, size) -> 5 metadata dimensions, max_replicas_per_node = 4:
    base id 4r        Base replica (all-zero dims: lives under the MAIN root, not in the pseudo-root component)
    4r + 1            color only      [c1 c0 | 0 0 0]
    4r + 2            size only       [0 0 | s2 s1 s0]
    4r + 3            color AND size  [c1 c0 | s2 s1 s0]
Pseudo nodes (api_service.rs:135-210): the pseudo root u32::MAX - 257 with all dims = HIGH_WEIGHT (1), then one node per
value combination (3 + 5 + 15), all sharing the all-zero vector."""
from __future__ import annotations

import numpy as np

from oracle import oracle as O
from tests import helpers as H

PSEUDO_ROOT = 0xFFFFFFFF - 257
MDIM, REPLICAS = 5, 4


def bits(v: int, size: int):
    return [(v >> (size - 1 - i)) & 1 for i in range(size)]


def dims(color=None, size=None, neq_color=False, neq_size=False):
    """metadata / filter dimensions: Equal -> the binary digits, NotEqual -> 1 -> -1, 0 -> 1 (query_filtering.rs:29-45)"""
    enc = lambda b, neq: [(-1 if x else 1) for x in b] if neq else b
    c = enc(bits(color, 2), neq_color) if color is not None else [0, 0]
    s = enc(bits(size, 3), neq_size) if size is not None else [0, 0, 0]
    return c + s


class Scenario:
    def __init__(self, n=1500, dim=64, seed=0, storage=O.STORAGE_U8, res=0, **hp):
        rng = np.random.default_rng(seed)
        self.n, self.dim = n, dim
        self.X = H.clustered_corpus(n, dim, n_centers=10, sigma=0.25, seed=seed + 1)
        self.color = rng.integers(1, 4, n)
        self.size = rng.integers(1, 6, n)
        ids, mb = [], []
        for r in range(n):
            c, s = int(self.color[r]), int(self.size[r])
            ids += [4 * r + 1, 4 * r + 2, 4 * r + 3]
            mb += [dims(color=c), dims(size=s), dims(color=c, size=s)]
        pseudo = [[1] * MDIM] + [dims(color=c) for c in range(1, 4)] + [dims(size=s) for s in range(1, 6)] + \
                 [dims(color=c, size=s) for c in range(1, 4) for s in range(1, 6)]
        ids += [PSEUDO_ROOT + i for i in range(len(pseudo))]
        mb += pseudo
        self.node_ids = np.array(ids, np.uint32)
        self.mbits = np.array(mb, np.int32)
        # enough slots for the root to keep an edge to every pseudo node and enough layers for the top mixed level to hold only
        # a handful of replicas: a matching pseudo node whose slots fill up with replicas EVICTS its edge to the root
        # (prob_node.rs:271-279), after which a filtered walk can only enter it from above
        self.hp = dict(num_layers=6, ef_construction=48, ef_search=64, neighbors_count=32, level0_neighbors_count=64)
        self.hp.update(hp)
        L = self.hp["num_layers"]
        # level draws as the reference makes them: replicas from the schema-adjusted levels_prob (api_service.rs:113-131: the
        # upper layers belong to the pseudo nodes), pseudo nodes from pseudo_level_probs over the non-root pseudo nodes
        lp, plp = O.schema_level_probs(L, len(pseudo)), O.pseudo_level_probs(L, len(pseudo) - 1)
        self.max_levels = np.array([O.max_insert_level(float(np.float32(rng.random())), plp if i >= PSEUDO_ROOT else lp)
                                    for i in self.node_ids], np.uint8)
        self.params = O.HNSWParams(dim=dim, storage=storage, resolution=res, seed=seed + 5, **self.hp)

    def oracle(self):
        oix = O.OracleIndex(self.params).set_vectors(self.X)
        oix.meta_enable(MDIM, REPLICAS)          # before any graph: base ids become 4r
        oix.build()
        oix.meta_set_nodes(self.node_ids, self.mbits).meta_build(self.max_levels)
        return oix

    def device(self, oix, id_base=0):
        import cosdata_amd as ca
        p = self.params
        hp = ca.HNSWHyperParams(num_layers=p.num_layers, ef_construction=p.ef_construction, ef_search=p.ef_search,
                                level_0_neighbors_count=p.level0_neighbors_count, neighbors_count=p.neighbors_count)
        dix = ca.HNSWIndex(self.dim, hp, ca.DistanceMetric(p.metric), ca.StorageType(ca.StorageKind(p.storage), p.resolution),
                           (p.range_lo, p.range_hi), p.shortlist_size, id_base=id_base)
        dix.upload_vectors(self.X).enable_metadata(MDIM, REPLICAS)
        dix.upload_graph(oix.export_graph(), oix.root_raw())
        dix.upload_meta_graph(self.node_ids, self.mbits, oix.meta_export_graph())
        return dix

    def queries(self, nq=24, seed=3):
        """(Q, offsets, filter rows, description per query)"""
        rng = np.random.default_rng(seed)
        Q = H.queries_from(self.X, nq, noise=0.05, seed=seed)
        off, rows, desc = [0], [], []
        for b in range(nq):
            c, s = int(rng.integers(1, 4)), int(rng.integers(1, 6))
            kind = b % 6
            if kind == 0:
                f, d = [dims(color=c)], ("is_color", c, None)
            elif kind == 1:
                f, d = [dims(size=s)], ("is_size", None, s)
            elif kind == 2:
                f, d = [dims(color=c, size=s)], ("and", c, s)
            elif kind == 3:
                f, d = [dims(color=c), dims(size=s)], ("or", c, s)
            elif kind == 4:
                f, d = [dims(color=c, neq_color=True)], ("neq_color", c, None)
            else:
                f, d = [dims(color=c, size=s, neq_size=True), dims(size=s), dims(color=(c % 3) + 1)], ("mixed", c, s)
            rows += f
            off.append(len(rows))
            desc.append(d)
        return Q, np.array(off, np.uint32), np.array(rows, np.int32), desc
