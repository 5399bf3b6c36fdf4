#!/usr/bin/env python3
"""Randomised soak of the query-resident quaternary scan: many (n, dim, B, metric, corpus) draws, each compared bit for bit with
the tile kernel (COS_FLAT_TILE_KERNEL=1) and, for the small ones, with the oracle.  Prints one line per draw and a summary."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import cosdata_amd as ca
import helpers as H
from oracle import oracle as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
draws = int(sys.argv[2]) if len(sys.argv) > 2 else 24
bad = 0
for t in range(draws):
    dim = int(rng.choice([65, 128, 200, 256, 300, 384, 500, 512, 700, 768, 1000, 1024]))
    n = int(rng.integers(17000, 140000))
    B = int(rng.choice([1, 3, 17, 64, 100, 256, 257, 300, 513]))
    metric = ca.DistanceMetric.Cosine if rng.random() < 0.7 else ca.DistanceMetric.DotProduct
    kind = rng.integers(0, 3)
    if kind == 0:
        X = H.clustered_corpus(n, dim, n_centers=int(rng.integers(2, 60)), sigma=float(rng.uniform(0.05, 0.4)), seed=int(rng.integers(1 << 30)))
    elif kind == 1:
        X = H.uniform_corpus(n, dim, seed=int(rng.integers(1 << 30))) * 0.9
    else:   # sorted by similarity to one direction: thresholds keep rising through the scan (many survivors per chunk)
        X = H.uniform_corpus(n, dim, seed=int(rng.integers(1 << 30))) * 0.9
        v = X[0] / np.linalg.norm(X[0])
        X = X[np.argsort(X @ v)]
    Q = H.queries_from(X, B, noise=float(rng.uniform(0.0, 0.2)), seed=int(rng.integers(1 << 30)))
    ix = ca.HNSWIndex(dim, ca.HNSWHyperParams(num_layers=3), distance_metric=metric, storage_type=ca.StorageType(ca.StorageKind(O.STORAGE_SUBBYTE), 2))
    ix.upload_vectors(np.ascontiguousarray(X, dtype=np.float32))
    k = int(rng.choice([1, 5, 10, 12]))
    a = ix.flat_search(Q, k)
    os.environ["COS_FLAT_TILE_KERNEL"] = "1"
    b = ix.flat_search(Q, k)
    del os.environ["COS_FLAT_TILE_KERNEL"]
    same = np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2], b[2])
    orc = ""
    if n * dim * B < 4e9:
        om = O.METRIC_COSINE if metric == ca.DistanceMetric.Cosine else O.METRIC_DOT
        oix = O.OracleIndex(O.HNSWParams(dim=dim, metric=om, storage=O.STORAGE_SUBBYTE, resolution=2, num_layers=3)).set_vectors(X)
        o = oix.flat_search_batch(Q, k, threads=16)
        ok = np.array_equal(a[0], o[0]) and np.array_equal(a[1].view(np.uint32), o[1].view(np.uint32)) and np.array_equal(a[2], o[2])
        orc = f" oracle={'ok' if ok else 'MISMATCH'}"
        same = same and ok
    bad += not same
    print(f"draw {t}: n={n} dim={dim} B={B} k={k} metric={metric.name} corpus={kind} same_as_tile={'ok' if same else 'MISMATCH'}{orc}", flush=True)
print(f"soak: {draws - bad}/{draws} draws identical")
sys.exit(1 if bad else 0)
