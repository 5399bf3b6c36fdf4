#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the oracle (run once; fixtures are committed).

The reference holds no golden vectors for quantize / cosine / walk / rerank / BM25 / RRF
and cannot be compiled in this image, so these oracle-generated fixtures are the pin: the oracle must keep
reproducing them (tests/test_golden.py, CPU) and the HIP path must reproduce them bit-for-bit without the
oracle in the loop (tests/test_golden.py -m gpu)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

STORAGES = {"u8": (O.STORAGE_U8, 0), "q2": (O.STORAGE_SUBBYTE, 2), "f32": (O.STORAGE_F32, 0), "f16": (O.STORAGE_F16, 0)}


def corpus(n, d, seed):
    rng = np.random.default_rng(seed)
    c = rng.standard_normal((12, d)).astype(np.float32)
    x = c[rng.integers(0, 12, n)] * 0.35 + 0.25 * rng.standard_normal((n, d)).astype(np.float32)
    return np.clip(x, -1.2, 1.2).astype(np.float32)


def main():
    rng = np.random.default_rng(20250919)
    # ---- quantize + distance known answers ------------------------------------------------------
    xq = rng.uniform(-1.3, 1.3, (24, 100)).astype(np.float32)
    xq[0, :6] = [1.0, -1.0, np.nan, 5e30, -5e30, 0.999999]
    xq[1] = 0.0   # |x| = 0: zero norm for SubByte / f32 storage
    xq[2] = -1.0  # all bytes 0: zero norm for u8 storage
    out = {"x": xq}
    for name, (st, res) in STORAGES.items():
        codes, mags = O.quantize_batch(xq, st, res, -1.0, 1.0)
        out[f"{name}_codes"], out[f"{name}_mags"] = codes, mags
        pairs = rng.integers(0, 24, (64, 2)).astype(np.uint32)
        pairs[0], pairs[1] = [1, 3], [2, 3]  # zero-norm rows -> DistanceError::CalculationError for cosine
        vals = np.zeros((2, 64), np.float32)
        stat = np.zeros((2, 64), np.int32)
        for mi, metric in enumerate((O.METRIC_COSINE, O.METRIC_DOT)):
            for p, (a, b) in enumerate(pairs):
                rc, v = O.distance(metric, st, res, 100, codes[a], mags[a], codes[b], mags[b])
                stat[mi, p], vals[mi, p] = rc, (v if rc == O.OK else 0.0)
        out[f"{name}_pairs"], out[f"{name}_dist"], out[f"{name}_dist_status"] = pairs, vals, stat
    np.savez_compressed(os.path.join(HERE, "quantize_distance.npz"), **out)

    # ---- graph + walk + search --------------------------------------------------------------------
    n, d = 600, 48
    X = corpus(n, d, 11)
    Q = np.concatenate([X[rng.integers(0, n, 12)] + 0.03 * rng.standard_normal((12, d)).astype(np.float32),
                        rng.uniform(-1, 1, (4, d)).astype(np.float32)]).astype(np.float32)
    for name, (st, res) in STORAGES.items():
        p = O.HNSWParams(dim=d, storage=st, resolution=res, num_layers=3, ef_construction=32, ef_search=24, seed=5)
        ix = O.OracleIndex(p).set_vectors(X).build()
        g = ix.export_graph()
        sav = {"X": X, "Q": Q, "root": ix.root_raw(), "num_layers": 3, "ef_construction": 32, "ef_search": 24}
        for l, (ids, nbr) in enumerate(g):
            sav[f"ids{l}"], sav[f"nbr{l}"] = ids, nbr
        walk_ids = np.full((Q.shape[0], 4, 100), 0, np.uint32)
        walk_sims = np.zeros((Q.shape[0], 4, 100), np.float32)
        walk_cnt = np.zeros((Q.shape[0], 4), np.uint32)
        for b in range(Q.shape[0]):
            wi, ws, lc = ix.ann_search(Q[b])
            off = 0
            for s, c in enumerate(lc):
                walk_ids[b, s, :c], walk_sims[b, s, :c] = wi[off:off + c], ws[off:off + c]
                off += int(c)
            walk_cnt[b] = lc
        sav["walk_ids"], sav["walk_sims"], sav["walk_counts"] = walk_ids, walk_sims, walk_cnt
        for k in (1, 5, 10):
            ids, sc, cnt = ix.search_batch(Q, k)[:3]
            sav[f"top{k}_ids"], sav[f"top{k}_scores"], sav[f"top{k}_counts"] = ids, sc, cnt
        np.savez_compressed(os.path.join(HERE, f"hnsw_{name}.npz"), **sav)

    # ---- BM25 + RRF ----------------------------------------------------------------------------------
    n_docs, T = 3000, 400
    terms = np.sort(rng.choice(1 << 31, T, replace=False).astype(np.uint32))
    lens = np.minimum(rng.zipf(1.3, T) * 3, 900).astype(np.int64)
    offsets = np.zeros(T + 1, np.uint64)
    offsets[1:] = np.cumsum(lens)
    docs = np.concatenate([np.sort(rng.choice(n_docs, int(l), replace=False)) for l in lens]).astype(np.uint32)
    tfs = np.array([O.bm25_tf(int(c), int(dl), 120.0, 1.5, 0.75) for c, dl in
                    zip(rng.integers(1, 6, docs.size), rng.integers(40, 260, docs.size))], np.float32)
    nq = 24
    q_off = np.zeros(nq + 1, np.uint32)
    q_terms = []
    for i in range(nq):
        m = int(rng.integers(2, 9))
        t = rng.choice(terms, m, replace=False).astype(np.uint32)
        if i % 5 == 0:
            t[0] = 12345  # a term with no posting list
        q_terms.append(t)
        q_off[i + 1] = q_off[i] + m
    q_terms = np.concatenate(q_terms)
    k = 10
    b_ids = np.full((nq, 3 * k), 0xFFFFFFFF, np.uint32)
    b_sc = np.zeros((nq, 3 * k), np.float32)
    b_cnt = np.zeros(nq, np.uint32)
    for i in range(nq):
        ids, sc = O.bm25_search(terms, offsets, docs, tfs, n_docs, q_terms[q_off[i]:q_off[i + 1]], 3 * k)
        b_ids[i, :ids.size], b_sc[i, :ids.size], b_cnt[i] = ids, sc, ids.size
    dense = rng.integers(0, n_docs, (nq, 3 * k)).astype(np.uint32)
    dense[:, 0] = b_ids[:, min(1, 3 * k - 1)]  # make the two lists overlap
    d_cnt = np.full(nq, 3 * k, np.uint32)
    f_ids = np.full((nq, k), 0xFFFFFFFF, np.uint32)
    f_sc = np.zeros((nq, k), np.float32)
    f_cnt = np.zeros(nq, np.uint32)
    for i in range(nq):
        ids, sc = O.rrf_fuse(dense[i, :d_cnt[i]], b_ids[i, :b_cnt[i]], 60.0, k)
        f_ids[i, :ids.size], f_sc[i, :ids.size], f_cnt[i] = ids, sc, ids.size
    np.savez_compressed(os.path.join(HERE, "bm25_rrf.npz"), terms=terms, offsets=offsets, docs=docs, tfs=tfs, n_docs=n_docs,
                        q_terms=q_terms, q_offsets=q_off, top_k=3 * k, bm25_ids=b_ids, bm25_scores=b_sc, bm25_counts=b_cnt,
                        dense_ids=dense, dense_counts=d_cnt, fusion_k=60.0, rrf_top_k=k, rrf_ids=f_ids, rrf_scores=f_sc,
                        rrf_counts=f_cnt)
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
