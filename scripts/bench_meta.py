#!/usr/bin/env python3
"""Metadata-filtered search at a size the tests do not reach: a collection of N vectors with the two-field schema
of tests/meta_helpers.py (3 colours x 5 sizes, AND supported -> 5 metadata dimensions, 4 replicas per embedding, 24 pseudo nodes).
Times the device builders (base graph, pseudo-root component) and cos_search_filtered_batch; checks a reduced-size twin of the same
collection against the oracle (component built by coso_meta_build_rounds, filtered search).  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cosdata_amd as ca
from oracle import oracle as O
from tests import meta_helpers as MH

N = int(os.environ.get("META_N", 200_000)); DIM = 768; B = 256
out = {"config": {"workload": f"metadata-filtered search: {N} x {DIM} u8, 4 replicas per embedding, 5 metadata dimensions, batch {B}"}}


def make(n, dim, seed, **hp):
    sc = MH.Scenario(n=n, dim=dim, seed=seed, **hp)
    p = sc.params
    h = ca.HNSWHyperParams(num_layers=p.num_layers, ef_construction=p.ef_construction, ef_search=p.ef_search,
                           level_0_neighbors_count=p.level0_neighbors_count, neighbors_count=p.neighbors_count)
    dix = ca.HNSWIndex(dim, h, ca.DistanceMetric(p.metric), ca.StorageType(ca.StorageKind(p.storage), p.resolution), (p.range_lo, p.range_hi), p.shortlist_size,
                       seed=p.seed)
    dix.upload_vectors(sc.X).enable_metadata(MH.MDIM, MH.REPLICAS)
    return sc, dix


# ---- parity twin: small enough for the oracle's builders --------------------------------------------------------------------------
sc, dix = make(6000, 96, 21)
oix = O.OracleIndex(sc.params).set_vectors(sc.X)
oix.meta_enable(MH.MDIM, MH.REPLICAS)
oix.build_rounds(512)
oix.meta_set_nodes(sc.node_ids, sc.mbits).meta_build_rounds(sc.max_levels, 512)
dix.build(512)
dix.build_meta(sc.node_ids, sc.mbits, sc.max_levels, 512)
same_graph = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(dix.download_meta_graph(), oix.meta_export_graph()))
Q, off, rows, _ = sc.queries(nq=256, seed=9)
gi, gs, gc = dix.search_filtered(Q, off, rows.astype(np.int8), 10)
ei, es, ec = oix.search_filtered_batch(Q, off, rows, 10, threads=8)
bad = sum(not (gc[b] == ec[b] and np.array_equal(gi[b, :gc[b]], ei[b, :ec[b]]) and np.array_equal(gs[b, :gc[b]].view(np.uint32), es[b, :ec[b]].view(np.uint32)))
          for b in range(256))
out["parity_vs_oracle"] = {"collection": "6000 x 96 twin, both builders on the device vs coso_index_build_rounds + coso_meta_build_rounds (batch 512)",
                           "component_graph_identical": bool(same_graph), "queries": 256, "mismatching_queries": int(bad)}
del dix, oix

# ---- timing at N ---------------------------------------------------------------------------------------------------------------------
sc, dix = make(N, DIM, 5, ef_search=64)
t = time.perf_counter(); dix.build(4096); t_base = time.perf_counter() - t
t = time.perf_counter(); dix.build_meta(sc.node_ids, sc.mbits, sc.max_levels, int(os.environ.get("META_BATCH", 0))); t_meta = time.perf_counter() - t
out["build"] = {"base_graph_s": t_base, "pseudo_root_component_s": t_meta, "component_nodes": int(len(sc.node_ids))}
for nq in (256, 4096):
    Q, off, rows, _ = sc.queries(nq=nq, seed=3)
    fd = rows.astype(np.int8)
    dix.search_filtered(Q, off, fd, 10)
    reps = 5
    t = time.perf_counter()
    for _ in range(reps):
        ids, scs, cnt = dix.search_filtered(Q, off, fd, 10)
    el = (time.perf_counter() - t) / reps
    out[f"filtered_search_B{nq}"] = {"ms_per_batch_host_api": el * 1e3, "qps": nq / el, "filters_per_query": float(off[-1]) / nq,
                                     "queries_with_results": int((cnt > 0).sum())}
print(json.dumps(out))
