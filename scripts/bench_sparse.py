#!/usr/bin/env python3
"""Learned-sparse inverted index (cos_sparse_search_batch) on a SPLADE-shaped synthetic corpus: n vectors with
~nnz non-zero dimensions each (Zipf over a vocabulary), values log-normal, 6-bit keys; queries of ~24 terms.  Reports wall time per
256-query batch through the host API (it allocates its accumulators per call), the postings the batch visits and the rate, and
checks a sample of queries against the oracle.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cosdata_amd as ca
from oracle import oracle as O

n = int(os.environ.get("SPARSE_N", 400_000)); vocab = 30_000; nnz = 48; bits = 6; upper = 3.0; B = 256; k = 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
pz = 1.0 / torch.arange(1, vocab + 1, device=dev, dtype=torch.float64) ** 0.9
cdf = torch.cumsum(pz / pz.sum(), 0)
tot = n * nnz
dim = torch.searchsorted(cdf, torch.rand(tot, generator=g, device=dev, dtype=torch.float64)).clamp_(max=vocab - 1)
vid = torch.arange(n, device=dev).repeat_interleave(nnz)
key_pair = torch.unique(dim * n + vid)                                   # one posting per (dim, vector)
dim, vid = key_pair // n, key_pair % n
val = torch.exp(0.6 * torch.randn(dim.numel(), generator=g, device=dev)).clamp_(max=upper * 1.2).float()
Q = 1 << bits
qk = torch.clamp((val / upper * (Q - 1)).clamp(0, Q - 1).to(torch.int64), max=Q - 1)   # InvertedIndexNode::quantize (values >= 0)
order = torch.argsort((dim * Q + qk) * n + vid)
dim_s, qk_s, vid_s = dim[order], qk[order], vid[order]
dims_present = torch.unique(dim_s)
T = dims_present.numel()
slot = torch.searchsorted(dims_present, dim_s)
cnt = torch.bincount(slot * Q + qk_s, minlength=T * Q).view(T, Q)
key_off = torch.zeros(T, Q + 1, dtype=torch.int64, device=dev)
key_off[:, 1:] = torch.cumsum(cnt, 1)
base = torch.zeros(T, dtype=torch.int64, device=dev); base[1:] = torch.cumsum(cnt.sum(1), 0)[:-1]
key_off += base[:, None]
dims_h = dims_present.cpu().numpy().astype(np.uint32); ko_h = key_off.cpu().numpy().astype(np.uint64).ravel(); vid_h = vid_s.cpu().numpy().astype(np.uint32)
ix = ca.InvertedIndex(bits, upper, dims_h, ko_h, vid_h, n)
rng = np.random.default_rng(9)
pz_h = (pz / pz.sum()).cpu().numpy()
qd, qv, qo = [], [], [0]
for _ in range(B):
    m = int(rng.integers(16, 33))
    d = np.sort(rng.choice(vocab, m, replace=False, p=pz_h)).astype(np.uint32)
    qd.append(d); qv.append(np.exp(0.6 * rng.standard_normal(m)).astype(np.float32)); qo.append(qo[-1] + m)
qd, qv, qo = np.concatenate(qd), np.concatenate(qv), np.array(qo, np.uint32)
thr = 0.0
ix.search_batch(qd, qv, qo, k, thr, 0)
t = time.perf_counter(); reps = 3
for _ in range(reps): ids, sc, cnt_o = ix.search_batch(qd, qv, qo, k, thr, 0)
ms = (time.perf_counter() - t) / reps * 1e3
pos = {int(d): i for i, d in enumerate(dims_h)}
ko2 = ko_h.reshape(T, Q + 1)
visited = sum(int(ko2[pos[int(d)], Q] - ko2[pos[int(d)], 0]) for d in qd if int(d) in pos)
bad = 0
for b in range(0, B, 16):
    cand, sims = O.sparse_search(dims_h, ko_h, vid_h, n, bits, upper, thr, qd[qo[b]:qo[b + 1]], qv[qo[b]:qo[b + 1]], k_with_reranking=k)
    c = int(cnt_o[b])
    bad += not (c == min(k, len(cand)) and np.array_equal(ids[b, :c], cand[:c]) and np.array_equal(sc[b, :c], sims[:c].astype(np.float32)))
print(json.dumps({"config": f"learned-sparse inverted index: {n} vectors, vocab {vocab}, {int(dim.numel())} postings, {bits}-bit keys, batch {B}, top_k {k}",
                  "ms_per_batch_host_api": ms, "postings_visited_per_batch": visited, "posting_GBps_incl_setup": visited * 4 / (ms * 1e-3) / 1e9,
                  "accumulator_bytes_per_batch": int(B) * n * 5, "parity_vs_oracle": {"queries": len(range(0, B, 16)), "mismatching_queries": int(bad)}}))
