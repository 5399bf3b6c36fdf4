#!/usr/bin/env python3
"""Config c5 (BASELINE.json configs[4]): hybrid dense(768) + BM25 inverted-index fused scoring, 1M docs, one MI355X.
dense = HNSW walk + rerank (top_k*3), sparse = BM25 over CSR postings (top_k*3), fusion = RRF (k=60) -> top_k,
exactly the composition of repo::hybrid_search (api/vectordb/search/repo.rs:168-341).  Synthetic text side:
Zipf(1.1) vocabulary of 200k term hashes, document length ~Poisson(120), k1=1.5, b=0.75 (tests/test_hybrid.py:216-217).
`run()` returns the record bench.py appends under configs.c5 (roofline of the BM25 posting scan and of the dense half, the
oracle composition on the host cores as the CPU baseline, bit-for-bit parity on every query of the batch); run as a script
it prints that record as one JSON line."""
import argparse, json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

HBM_PEAK_GBPS = 8000.0


def run(n=1_000_000, dim=768, vocab=200_000, doc_len=120.0, batch=256, top_k=10, check=256, cpu_seconds=5.0, device=0):
    import torch
    import cosdata_amd as ca
    import bench
    t_all = time.time()
    dev = torch.device(f"cuda:{device}")
    d, V, B, k = dim, vocab, batch, top_k
    g = torch.Generator(device=dev); g.manual_seed(11)
    # ---- dense side (same generator family as bench.py) ----
    nc = max(64, n // 1000)
    centers = torch.randn(nc, d, generator=g, device=dev); centers /= centers.norm(dim=1, keepdim=True)
    X = bench.mixture(torch, n, d, 42, dev, centers); Q = bench.mixture(torch, B, d, 43, dev, centers)
    ix = ca.HNSWIndex(d, ca.HNSWHyperParams(), ca.DistanceMetric.Cosine, ca.StorageType.UnsignedByte(), (-1.0, 1.0), device=device)
    ix.upload_vectors_device(X.data_ptr(), n, keepalive=X)
    t = time.time(); ix.build(4096); t_build = time.time() - t
    # ---- text side: Zipf tokens -> (term, doc, count) -> CSR postings with stored BM25 tf ----
    t = time.time()
    ranks = torch.arange(1, V + 1, device=dev, dtype=torch.float64)
    pz = (1.0 / ranks ** 1.1); pz /= pz.sum()
    lens = torch.poisson(torch.full((n,), doc_len, device=dev), generator=g).clamp_(min=1).to(torch.int64)
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
    th_h = hashes.cpu().numpy().astype(np.uint32); off_h = offsets.cpu().numpy().astype(np.uint64)
    docs_h = p_doc.cpu().numpy().astype(np.uint32); tf_h = tf.cpu().numpy().astype(np.float32)
    n_postings = int(ukey.numel())
    del ukey, counts, key, term_rank, doc_of_tok, p_term, c, dl
    t_text = time.time() - t
    bm = ca.BM25Index(th_h, off_h, docs_h, tf_h, n, device=device)
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
    h_i, h_s, h_c = bm.search_batch(q_terms, q_off, 3 * k)
    d_i, d_s, d_c = o_i.cpu().numpy().view(np.uint32), o_s.cpu().numpy(), o_c.cpu().numpy().view(np.uint32)
    live = np.arange(3 * k)[None, :] < h_c[:, None]      # entries past a query's count are not part of the answer
    dev_equal_host = bool(np.array_equal(d_c, h_c) and np.array_equal(d_i[live], h_i[live]) and np.array_equal(d_s[live].view(np.uint32), h_s[live].view(np.uint32)))
    # dense half alone through the device entry point: walk kernel time + counters (what bounds the one-call hybrid)
    dd_i = torch.zeros(B, 3 * k, dtype=torch.int32, device=dev); dd_s = torch.zeros(B, 3 * k, device=dev)
    dd_c = torch.zeros(B, dtype=torch.int32, device=dev); dd_t = torch.zeros(B, dtype=torch.int32, device=dev)
    ix.enable_timing(True)
    for _ in range(3):
        ix.batch_search_device(Q.data_ptr(), B, 3 * k, dd_i.data_ptr(), dd_s.data_ptr(), dd_c.data_ptr(), dd_t.data_ptr(), st_bm.cuda_stream)
    st_bm.synchronize()
    stt = ix.last_stats(st_bm.cuda_stream)
    ix.enable_timing(False)
    dense_bytes = stt.evals * (d + 4) + stt.adj_bytes
    # one-call hybrid: dense and BM25 concurrently on the device, RRF there
    ca.hybrid_search_batch(ix, bm, Qh, q_terms, q_off, k, 60.0)
    t = time.time()
    for _ in range(reps): h_ids, h_sc, h_cnt = ca.hybrid_search_batch(ix, bm, Qh, q_terms, q_off, k, 60.0)
    el_h1 = (time.time() - t) / reps
    one_call_equal = bool(np.array_equal(h_ids, fres[0]) and np.array_equal(h_sc.view(np.uint32), fres[1].view(np.uint32)) and np.array_equal(h_cnt, fres[2]))
    post_bytes = 0
    pos = {int(h): i for i, h in enumerate(th_h)}
    for h in q_terms: post_bytes += int(off_h[pos[int(h)] + 1] - off_h[pos[int(h)]]) * 8
    bm_gbps = post_bytes / (bm_kernel_ms * 1e-3) / 1e9
    out = {"config": {"workload": f"c5: BASELINE configs[4]: hybrid dense({d}) HNSW + BM25 + RRF, {n} docs, batch {B}, top_k {k}",
                      "standard_size": n == 1_000_000 and d == 768 and B == 256, "docs": n, "dim": d, "query_batch": B, "top_k": k,
                      "postings": n_postings, "vocab": V, "avg_doc_len": avg_len, "ef_search": 256, "fusion_constant_k": 60,
                      "step": "one cos_hybrid_search_batch call = (quantize -> walk -> rerank, top_k x 3) || (BM25 posting scan, top_k x 3) -> RRF -> top_k"},
           "qps": B / el_h1, "unit": "queries/s", "ms_per_step": el_h1 * 1e3, "steps": reps, "warmup": 1, "dtype": "u8 walk + f32 BM25",
           # the step's DOMINANT kernel is the dense half's walk (one 256-query batch cannot fill the chip): it is the headline block;
           # the BM25 posting scan, which runs next to it on its own stream, is under `parts`
           "roofline": {"bound": "hbm", "achieved": dense_bytes / (stt.walk_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": dense_bytes / (stt.walk_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": bench.committed_kernel_traffic("c5_walk"),
                        "step_frac": (dense_bytes + post_bytes) / el_h1 / 1e9 / HBM_PEAK_GBPS,
                        "kernel": "walk_lat4_kernel (one 256-query batch, ef 256: four waves per query)",
                        "per_launch": {"algorithmic_bytes": float(dense_bytes), "avg_ms": stt.walk_ms, "evals": float(stt.evals),
                                       "expansions": float(stt.expansions)},
                        "parts": {"bm25_score": {"bound": "hbm", "kernel": "bm25_score_kernel (+ bm25_topk_kernel on the same stream)", "bytes": float(post_bytes),
                                                 "ms": bm_kernel_ms, "achieved": bm_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": bm_gbps / HBM_PEAK_GBPS,
                                                 "note": "8 B x the postings of every query term of the batch / the HIP-event time of one "
                                                         "cos_bm25_search_batch_device call on its stream (score + top-k kernels)"}},
                        "note": "achieved = algorithmic bytes of the dense half's walk (evaluations x (dim + 4) + adjacency) / the walk kernel's HIP-event "
                                "time; the one-call hybrid is bounded by this launch — a chain of dependent rounds per query with a quarter of the "
                                "chip's wave slots occupied; step_frac = (walk + posting bytes) / ms_per_step"},
           "build_s": t_build, "text_gen_s": t_text,
           "hybrid_three_calls_ms_per_batch": el * 1e3, "bm25_ms_per_batch_host_api": el_bm * 1e3, "bm25_ms_per_batch_device_api": el_bm_dev * 1e3,
           "bm25_device_api_equals_host_api": dev_equal_host, "hybrid_one_call_equals_three_calls": one_call_equal,
           "cpu_baseline": None, "parity_vs_oracle": None}
    # ---- CPU baseline + parity vs the oracle composition (dense oracle search, BM25 heap merge, RRF), every query of the batch ----
    if cpu_seconds > 0:
        from oracle import oracle as O
        cores = bench.effective_cores()
        m = min(check, B)
        oix = O.OracleIndex(O.HNSWParams(dim=d, seed=42)).set_vectors(X.cpu().numpy()).import_graph(ix.download_graph(), ix.download_root())

        def text_side(i):
            oi, osc = O.bm25_search(th_h, off_h, docs_h, tf_h, n, q_terms[q_off[i]:q_off[i + 1]], 3 * k)
            return oi, osc

        def compose():
            od = oix.search_batch(Qh[:m], 3 * k, threads=cores)
            with ThreadPoolExecutor(cores) as ex:       # ctypes releases the GIL: one query per core like rayon's fan-out
                sp = list(ex.map(text_side, range(m)))
            fused = [O.rrf_fuse(od[0][i, :od[2][i]], sp[i][0], 60.0, k) for i in range(m)]
            return od, sp, fused
        t = time.perf_counter()
        od, sp, fused = compose()
        first = time.perf_counter() - t
        reps_c = 1 + int(max(0, min(7, round(cpu_seconds / first) - 1)))
        for _ in range(reps_c - 1):
            compose()
        cpu_s = time.perf_counter() - t
        bad = 0
        for i in range(m):
            oi, osc = sp[i]
            fi, fs = fused[i]
            ok = (np.array_equal(sres[0][i, :sres[2][i]], oi) and np.array_equal(sres[1][i, :sres[2][i]].view(np.uint32), osc.view(np.uint32))
                  and np.array_equal(h_ids[i, :h_cnt[i]], fi) and np.array_equal(h_sc[i, :h_cnt[i]].view(np.uint32), fs.view(np.uint32))
                  and np.array_equal(dres[0][i, :dres[1][i]], od[0][i, :od[2][i]]))
            bad += (not ok)
        out["cpu_baseline"] = {"value": reps_c * m / cpu_s, "unit": "queries/s", "cores": cores, "kind": "port",
                               "sample": f"{reps_c} passes over the {m} hybrid queries ({cpu_s:.1f} s wall on {cores} threads): oracle dense search "
                                         "(ef 256, top_k x 3) + search_bm25 heap merge + RRF, one query per core"}
        out["parity_vs_oracle"] = {"queries": m, "mismatching_queries": int(bad),
                                   "checked": "dense ids, BM25 ids + score bits, fused (one-call) ids + score bits, per query"}
        del oix
    del ix, bm, X
    torch.cuda.empty_cache()
    out["seconds"] = time.time() - t_all
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--vocab", type=int, default=200_000)
    ap.add_argument("--doc-len", type=float, default=120.0)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--check", type=int, default=256)
    ap.add_argument("--cpu-seconds", type=float, default=5.0)
    a = ap.parse_args()
    print(json.dumps(run(a.n, a.dim, a.vocab, a.doc_len, a.batch, a.top_k, a.check, a.cpu_seconds)))
