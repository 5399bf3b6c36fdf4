#!/usr/bin/env python3
"""Config c5 (BASELINE.json configs[4]): hybrid dense(768) + BM25 inverted-index fused scoring, 1M docs, one MI355X.
dense = HNSW walk + rerank (top_k*3), sparse = BM25 over CSR postings (top_k*3), fusion = RRF (k=60) -> top_k,
exactly the composition of repo::hybrid_search (api/vectordb/search/repo.rs:168-341).  Synthetic text side:
Zipf(1.1) vocabulary of 200k term hashes, document length ~Poisson(120), k1=1.5, b=0.75 (tests/test_hybrid.py:216-217).
Checks a sample against the oracle and prints one JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cosdata_amd as ca
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--vocab", type=int, default=200_000)
ap.add_argument("--doc-len", type=float, default=120.0)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--top-k", type=int, default=10)
ap.add_argument("--check", type=int, default=256)
a = ap.parse_args()
dev = torch.device("cuda:0")
n, d, V, B, k = a.n, a.dim, a.vocab, a.batch, a.top_k
g = torch.Generator(device=dev); g.manual_seed(11)
# ---- dense side (same generator family as bench.py) ----
nc = max(64, n // 1000)
centers = torch.randn(nc, d, generator=g, device=dev); centers /= centers.norm(dim=1, keepdim=True)
def mix(m, seed):
    gg = torch.Generator(device=dev); gg.manual_seed(seed)
    out = torch.empty(m, d, device=dev)
    for s in range(0, m, 1 << 18):
        kk = min(1 << 18, m - s)
        x = centers[torch.randint(0, nc, (kk,), generator=gg, device=dev)] + (0.8 / d ** 0.5) * torch.randn(kk, d, generator=gg, device=dev)
        out[s:s + kk] = x / x.norm(dim=1, keepdim=True)
    return out
X = mix(n, 42); Q = mix(B, 43)
ix = ca.HNSWIndex(d, ca.HNSWHyperParams(), ca.DistanceMetric.Cosine, ca.StorageType.UnsignedByte(), (-1.0, 1.0))
ix.upload_vectors_device(X.data_ptr(), n, keepalive=X)
t = time.time(); ix.build(4096); t_build = time.time() - t
# ---- text side: Zipf tokens -> (term, doc, count) -> CSR postings with stored BM25 tf ----
t = time.time()
ranks = torch.arange(1, V + 1, device=dev, dtype=torch.float64)
pz = (1.0 / ranks ** 1.1); pz /= pz.sum()
lens = torch.poisson(torch.full((n,), a.doc_len, device=dev), generator=g).clamp_(min=1).to(torch.int64)
tot = int(lens.sum().item())
doc_of_tok = torch.repeat_interleave(torch.arange(n, device=dev), lens)
cdf = torch.cumsum(pz, 0)
term_rank = torch.searchsorted(cdf, torch.rand(tot, generator=g, device=dev, dtype=torch.float64)).clamp_(max=V - 1)
hashes = torch.unique(torch.randint(0, 1 << 31, (V * 2,), generator=g, device=dev, dtype=torch.int64))[:V]  # ascending distinct term hashes
assert hashes.numel() == V
key = term_rank * n + doc_of_tok
ukey, counts = torch.unique(key, return_counts=True)            # sorted by (term, doc)
p_term = ukey // n; p_doc = (ukey % n).to(torch.int32)
avg_len = float(lens.double().mean().item())
c = counts.to(torch.float32); dl = lens[p_doc.long()].to(torch.float32)
k1, b = 1.5, 0.75
tf = c * (k1 + 1.0) / (c + k1 * (1.0 - b + b * (dl / avg_len)))   # compute_bm25_term_frequency (f32)
df = torch.bincount(p_term, minlength=V)
offsets = torch.zeros(V + 1, dtype=torch.int64, device=dev); offsets[1:] = torch.cumsum(df, 0)
keep = df > 0
th_h = hashes.cpu().numpy().astype(np.uint32); off_h = offsets.cpu().numpy().astype(np.uint64)
docs_h = p_doc.cpu().numpy().astype(np.uint32); tf_h = tf.cpu().numpy().astype(np.float32)
t_text = time.time() - t
bm = ca.BM25Index(th_h, off_h, docs_h, tf_h, n)
# queries: 2-8 terms, Zipf-distributed
rng = np.random.default_rng(5)
q_terms, q_off = [], [0]
pz_h = pz.cpu().numpy()
for i in range(B):
    m = int(rng.integers(2, 9))
    q_terms.append(th_h[rng.choice(V, m, replace=False, p=pz_h)])
    q_off.append(q_off[-1] + m)
q_terms = np.concatenate(q_terms).astype(np.uint32); q_off = np.array(q_off, np.uint32)
Qh = Q.cpu().numpy()
# ---- hybrid search on the GPU ----
def hybrid():
    d_ids, d_sc, d_cnt = ix.batch_search(Qh, 3 * k)
    s_ids, s_sc, s_cnt = bm.search_batch(q_terms, q_off, 3 * k)
    f_ids, f_sc, f_cnt = ca.rrf_fuse_batch(d_ids, d_cnt, s_ids, s_cnt, 60.0, k)
    return (d_ids, d_cnt), (s_ids, s_sc, s_cnt), (f_ids, f_sc, f_cnt)
hybrid()
t = time.time(); reps = 5
for _ in range(reps): dres, sres, fres = hybrid()
el = (time.time() - t) / reps
t = time.time()
for _ in range(reps): bm.search_batch(q_terms, q_off, 3 * k)
el_bm = (time.time() - t) / reps
# device-output entry point: no allocation / D2H on the query path; kernel time from HIP events on the caller's stream
o_i = torch.zeros(B, 3 * k, dtype=torch.int32, device=dev); o_s = torch.zeros(B, 3 * k, device=dev); o_c = torch.zeros(B, dtype=torch.int32, device=dev)
st_bm = torch.cuda.Stream(device=dev)
bm.search_batch_device(q_terms, q_off, 3 * k, o_i.data_ptr(), o_s.data_ptr(), o_c.data_ptr(), st_bm.cuda_stream)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t = time.time()
with torch.cuda.stream(st_bm):
    ev0.record(st_bm)
    for _ in range(reps): bm.search_batch_device(q_terms, q_off, 3 * k, o_i.data_ptr(), o_s.data_ptr(), o_c.data_ptr(), st_bm.cuda_stream)
    ev1.record(st_bm)
st_bm.synchronize()
el_bm_dev = (time.time() - t) / reps
bm_kernel_ms = ev0.elapsed_time(ev1) / reps
dev_equal_host = bool(np.array_equal(o_i.cpu().numpy().view(np.uint32), bm.search_batch(q_terms, q_off, 3 * k)[0]))
# one-call hybrid: dense and BM25 concurrently on the device, RRF there
ca.hybrid_search_batch(ix, bm, Qh, q_terms, q_off, k, 60.0)
t = time.time()
for _ in range(reps): h_ids, h_sc, h_cnt = ca.hybrid_search_batch(ix, bm, Qh, q_terms, q_off, k, 60.0)
el_h1 = (time.time() - t) / reps
one_call_equal = bool(np.array_equal(h_ids, fres[0]) and np.array_equal(h_sc.view(np.uint32), fres[1].view(np.uint32)) and np.array_equal(h_cnt, fres[2]))
post_bytes = 0
pos = {int(h): i for i, h in enumerate(th_h)}
for h in q_terms: post_bytes += int(off_h[pos[int(h)] + 1] - off_h[pos[int(h)]]) * 8
# ---- parity on a sample vs the oracle composition ----
m = min(a.check, B)
bad = 0
oix = O.OracleIndex(O.HNSWParams(dim=d, seed=42)).set_vectors(X.cpu().numpy()).import_graph(ix.download_graph(), ix.download_root())
od = oix.search_batch(Qh[:m], 3 * k, threads=os.cpu_count() or 1)
for i in range(m):
    oi, osc = O.bm25_search(th_h, off_h, docs_h, tf_h, n, q_terms[q_off[i]:q_off[i + 1]], 3 * k)
    fi, fs = O.rrf_fuse(od[0][i, :od[2][i]], oi, 60.0, k)
    ok = (np.array_equal(sres[0][i, :sres[2][i]], oi) and np.array_equal(sres[1][i, :sres[2][i]].view(np.uint32), osc.view(np.uint32))
          and np.array_equal(fres[0][i, :fres[2][i]], fi) and np.array_equal(fres[1][i, :fres[2][i]].view(np.uint32), fs.view(np.uint32))
          and np.array_equal(dres[0][i, :dres[1][i]], od[0][i, :od[2][i]]))
    bad += (not ok)
print(json.dumps({"config": f"c5: hybrid dense({d}) HNSW + BM25 + RRF, {n} docs, batch {B}, top_k {k}",
                  "postings": int(ukey.numel()), "vocab": V, "avg_doc_len": avg_len, "build_s": t_build, "text_gen_s": t_text,
                  "hybrid_qps_host_api": B / el, "hybrid_ms_per_batch": el * 1e3,
                  "bm25_ms_per_batch_host_api": el_bm * 1e3, "bm25_posting_bytes_per_batch": post_bytes,
                  "bm25_GBps_host_api_incl_setup": post_bytes / el_bm / 1e9,
                  "bm25_ms_per_batch_device_api": el_bm_dev * 1e3, "bm25_stream_ms_per_batch_hip_events": bm_kernel_ms,
                  "bm25_GBps_device_api": post_bytes / (bm_kernel_ms * 1e-3) / 1e9, "bm25_frac_of_hbm_8TBps": post_bytes / (bm_kernel_ms * 1e-3) / 8e12,
                  "bm25_device_api_equals_host_api": dev_equal_host,
                  "hybrid_one_call_ms_per_batch": el_h1 * 1e3, "hybrid_one_call_qps": B / el_h1, "hybrid_one_call_equals_three_calls": one_call_equal,
                  "parity_vs_oracle": {"queries": m, "mismatching_queries": int(bad)}}))
