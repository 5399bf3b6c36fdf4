"""Shared helpers for the parity tests: seeded corpora, oracle<->device index pairs."""
from __future__ import annotations

import numpy as np

from oracle import oracle as O


def uniform_corpus(n, d, seed=42):
    return np.random.default_rng(seed).uniform(-1.0, 1.0, (n, d)).astype(np.float32)


def clustered_corpus(n, d, n_centers=64, sigma=0.15, seed=42):
    """Gaussian mixture, L2-normalised (mirrors tests/test-dataset.py:414-430 normalisation)."""
    rng = np.random.default_rng(seed)
    centers = rng.standard_normal((n_centers, d)).astype(np.float32)
    centers /= np.linalg.norm(centers, axis=1, keepdims=True)
    x = centers[rng.integers(0, n_centers, n)] + sigma * rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)


def queries_from(corpus, nq, noise=0.02, seed=43):
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, corpus.shape[0], nq)
    q = corpus[idx] + noise * rng.standard_normal((nq, corpus.shape[1])).astype(np.float32)
    return q.astype(np.float32)


def oracle_index(X, storage=O.STORAGE_U8, resolution=0, **kw):
    p = O.HNSWParams(dim=X.shape[1], storage=storage, resolution=resolution, **kw)
    return O.OracleIndex(p).set_vectors(X).build()


def device_index_from_oracle(oix: "O.OracleIndex", X, visited_mode=0, id_base=0):
    import cosdata_amd as ca
    p = oix.params
    st = ca.StorageType(ca.StorageKind(p.storage), p.resolution)
    hp = ca.HNSWHyperParams(num_layers=p.num_layers, ef_construction=p.ef_construction, ef_search=p.ef_search,
                            level_0_neighbors_count=p.level0_neighbors_count, neighbors_count=p.neighbors_count)
    dix = ca.HNSWIndex(p.dim, hp, ca.DistanceMetric(p.metric), st, (p.range_lo, p.range_hi), p.shortlist_size,
                       visited_mode=visited_mode, id_base=id_base)
    dix.upload_vectors(X)
    dix.upload_graph(oix.export_graph(), oix.root_raw())
    return dix
