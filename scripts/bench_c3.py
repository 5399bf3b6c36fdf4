#!/usr/bin/env python3
"""Config c3 (BASELINE.json configs[2]): 10M x 768 quaternary-quantized distance on one MI355X.
(i) exhaustive scan of the 2-bit codes (i8 MFMA GEMM, cos_flat_search_batch) and, with --walk-n, (ii) HNSW walk with
quaternary distance + f32 rerank on a subset; recall@10 vs exact f32 brute force for both.  `run()` returns the record
bench.py appends under configs.c3 (roofline of the scan kernel, the oracle's exhaustive scan on the host cores as the CPU
baseline, and bit-for-bit parity of the GPU answers with it on the same sample at the FULL corpus size); run as a script
it prints that record as one JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

I8_PEAK_TOPS = 5000.0
FP4_PEAK_TOPS = 10000.0   # dense FP4 (e2m1) MFMA peak, MI355X_MICROARCH.md: ~10 PF dense; measured 9099 TF (32x32x64)


def _cores():
    import bench
    return bench.effective_cores()


def run(n=10_000_000, dim=768, batch=256, walk_n=0, reps=3, cpu_seconds=5.0, device=0):
    import torch
    import bench
    import cosdata_amd as ca
    t_all = time.time()
    dev = torch.device(f"cuda:{device}")
    g = torch.Generator(device=dev); g.manual_seed(7)
    d, B = dim, batch
    nc = max(64, n // 1000)
    centers = torch.rand(nc, d, generator=g, device=dev) * 1.6 - 0.8

    def draw(m, seed):
        gg = torch.Generator(device=dev); gg.manual_seed(seed)
        out = torch.empty(m, d, device=dev)
        for s in range(0, m, 1 << 18):
            k = min(1 << 18, m - s)
            idx = torch.randint(0, nc, (k,), generator=gg, device=dev)
            out[s:s + k] = (centers[idx] + 0.2 * torch.randn(k, d, generator=gg, device=dev)).clamp_(-0.999, 0.999)
        return out
    X = draw(n, 42)
    Q = draw(B, 43)
    torch.cuda.synchronize()
    st_q2 = ca.StorageType.SubByte(2)
    ix = ca.HNSWIndex(d, ca.HNSWHyperParams(), ca.DistanceMetric.Cosine, st_q2, (-1.0, 1.0), device=device)
    t = time.time(); ix.upload_vectors_device(X.data_ptr(), n, keepalive=X); t_up = time.time() - t
    Qh = Q.cpu().numpy()
    gt, _ = ix.bruteforce_topk(Qh, 10)
    ix.flat_search(Qh, 10)                       # first call sizes the per-index workspace: not timed
    runs = []
    for _ in range(reps):
        t = time.time()
        ids, sc, cnt, st = ix.flat_search(Qh, 10, with_stats=True)
        runs.append((st.gemm_ms, time.time() - t, st))
    gemm_ms = float(np.median([r[0] for r in runs])); wall = float(np.median([r[1] for r in runs])); st = runs[-1][2]
    # full-size parity property: three implementations of the scan (query-resident kernel, 256x128 tile kernel with the fused
    # epilogue, unfused score-matrix path) must return the same ids / score bits / counts
    same = {}
    from cosdata_amd import _lib
    for env, val in (("flat_fp4", 0), ("flat_tile_kernel", 1), ("flat_unfused", 1), ("flat_fp4_w8", 0)):
        with _lib.tuning(**{env: val}):
            i2, s2, c2 = ix.flat_search(Qh, 10)
        same[env] = bool(np.array_equal(i2, ids) and np.array_equal(s2.view(np.uint32), sc.view(np.uint32)) and np.array_equal(c2, cnt))
    rec_flat = float(np.mean([len(set(ids[i].tolist()) & set(gt[i].tolist())) / 10 for i in range(B)]))
    tops = st.int8_ops / gemm_ms / 1e9
    # the same scan with the digits as i8 (round 4's kernel, tuning knob flat_fp4 = 0), timed the same way: what the FP4 operands buy
    fp4_on = _lib.tuning_get("flat_fp4") in (None, 1) and d % 64 == 0 and (d // 64) in (2, 4, 6, 8, 12, 16)
    with _lib.tuning(flat_fp4=0):
        ix.flat_search(Qh, 10)
        i8_runs = []
        for _ in range(reps):
            t = time.time()
            _, _, _, st8 = ix.flat_search(Qh, 10, with_stats=True)
            i8_runs.append((st8.gemm_ms, time.time() - t))
    i8_gemm_ms = float(np.median([r[0] for r in i8_runs])); i8_wall = float(np.median([r[1] for r in i8_runs]))
    PEAK = FP4_PEAK_TOPS if fp4_on else I8_PEAK_TOPS
    out = {"config": {"workload": f"c3: BASELINE configs[2]: {n} x {d} quaternary (SubByte 2, values_range (-1,1)), exhaustive scan of the codes, "
                                  f"query batch {B}", "standard_size": n == 10_000_000 and d == 768 and B == 256, "vectors": n, "dim": d, "query_batch": B,
                      "step": "one cos_flat_search_batch call = i8-MFMA scan of every code row + top-5k selection + exact f32 rerank + top-k"},
           "qps": B / wall, "unit": "queries/s", "ms_per_step": wall * 1e3, "steps": reps, "warmup": 1, "dtype": "fp4 e2m1 (2-bit digits, exact)" if fp4_on else "i8 (2-bit digits)",
           "recall_at_10": rec_flat, "recall_note": "vs exact f32 brute force; the reference's fixed-range 2-bit quantizer with MSB-first planes "
                                                    "multiplied LSB-first bounds it — the GPU reproduces the oracle exactly, the recall is the reference's",
           "roofline": {"bound": "mfma", "achieved": tops, "peak": PEAK, "unit": "TOP/s", "frac": tops / PEAK,
                        # fabric-side bytes of one step's scan kernels from the committed FETCH / WRITE pass of this script (scripts/final_profile.sh)
                        "traffic": (lambda t: t * st.gemm_launches if t else None)(bench.committed_kernel_traffic("c3_scan" if fp4_on else "c3_scan_i8")),
                        "kernel": (("flat_scan_q2_fp4_w8<12, 3> (query-resident scan, two waves per SIMD" if d == 768 and _lib.tuning_get("flat_fp4_w8") in (None, 4) else
                                   "flat_scan_q2_fp4<%d> (query-resident scan" % (d // 64)) + ", digits as e2m1 on v_mfma_scale_f32_32x32x64_f8f6f4; peak = dense FP4)") if fp4_on
                                  else "flat_scan_q2_areg<KC> (query-resident i8 MFMA scan)",
                        "same_scan_with_i8_digits": {"kernel": "flat_scan_q2_areg<KC>", "gemm_ms_all_launches": i8_gemm_ms, "ms_per_step": i8_wall * 1e3,
                                                     "achieved": st.int8_ops / i8_gemm_ms / 1e9, "peak": I8_PEAK_TOPS, "frac": st.int8_ops / i8_gemm_ms / 1e9 / I8_PEAK_TOPS},
                        "per_launch": {"gemm_ms_all_launches": gemm_ms, "gemm_launches": st.gemm_launches, "int8_ops": float(st.int8_ops),
                                       "code_bytes": float(st.code_bytes), "code_GBps": st.code_bytes / gemm_ms / 1e6},
                        "note": "achieved = 2 x B x N x dim integer ops / the scan kernels' HIP-event time inside the call (median of the timed calls)"},
           "flat": {"gemm_ms": gemm_ms, "gemm_launches": st.gemm_launches, "int8_tops": tops, "int8_peak_tops_dense": I8_PEAK_TOPS,
                    "code_GBps": st.code_bytes / gemm_ms / 1e6, "wall_s_incl_select_rerank_copies": wall,
                    "timing": f"median of {reps} calls after one untimed call", "qps_end_to_end": B / wall, "upload_quantize_s": t_up,
                    "same_answer_as_i8_digit_kernel": same["flat_fp4"], "same_answer_as_one_wave_per_simd_fp4_kernel": same["flat_fp4_w8"], "same_answer_as_tile_kernel": same["flat_tile_kernel"],
                    "same_answer_as_unfused_path": same["flat_unfused"]},
           "cpu_baseline": None, "parity_vs_oracle": None}

    # ---- CPU baseline + parity at the full corpus size: the oracle quantizes the corpus itself (streamed: no 30 GB host table),
    # scans every code row per query and reranks the best 5k from the raw rows of exactly those candidates
    if cpu_seconds > 0:
        from oracle import oracle as O
        cores = _cores()
        op = O.HNSWParams(dim=d, storage=O.STORAGE_SUBBYTE, resolution=2, range_lo=-1.0, range_hi=1.0)
        oix = O.OracleIndex(op).alloc_vectors(n)
        for s0 in range(0, n, 1 << 18):
            oix.quantize_rows(s0, X[s0:s0 + (1 << 18)].cpu().numpy())
        pm = min(B, cores)
        t = time.perf_counter()
        oix.flat_candidates_batch(Qh[:pm], 10, threads=cores)
        rate = pm / (time.perf_counter() - t)
        m = int(min(B, max(pm, rate * cpu_seconds)))
        cand, _ = oix.flat_candidates_batch(Qh[:m], 10, threads=cores)
        u = np.unique(cand[cand != 0xFFFFFFFF])
        oix.set_raw_subset(u, X[torch.from_numpy(u.astype(np.int64)).to(dev)].cpu().numpy())
        t = time.perf_counter()
        oi, osc, ocnt = oix.flat_search_batch(Qh[:m], 10, threads=cores)
        cpu_s = time.perf_counter() - t
        out["cpu_baseline"] = {"value": m / cpu_s, "unit": "queries/s", "cores": cores, "kind": "port",
                               "sample": f"{m} of the {B} queries against all {n} code rows ({cpu_s:.1f} s wall on {cores} threads): the oracle's "
                                         "exhaustive scan (AVX2 nibble-LUT popcount dot_product_quaternary per row, bounded top-5k, exact rerank); "
                                         "corpus quantized by the oracle in streamed chunks"}
        out["parity_vs_oracle"] = {"queries": m, "id_mismatch_queries": int((ids[:m] != oi).any(axis=1).sum()),
                                   "score_bit_mismatches": int((sc[:m].view(np.uint32) != osc.view(np.uint32)).sum()),
                                   "count_mismatches": int((cnt[:m] != ocnt).sum())}
        del oix
    del ix
    if walk_n > 0:  # (ii) HNSW walk with quaternary distance + f32 rerank (dot_product_quaternary inside traverse_find_nearest,
        # x86_64.rs:103-160 in vector_store.rs:1112-1204), graph built on the device over the first walk_n vectors
        from oracle import oracle as O
        m = min(walk_n, n)
        Xs = X[:m].contiguous()
        ix2 = ca.HNSWIndex(d, ca.HNSWHyperParams(), ca.DistanceMetric.Cosine, st_q2, (-1.0, 1.0), device=device)
        ix2.upload_vectors_device(Xs.data_ptr(), m, keepalive=Xs)
        t = time.time(); ix2.build(4096); t_build = time.time() - t
        gt2, _ = ix2.bruteforce_topk(Qh, 10)
        Bq = 8192
        Qb = draw(Bq, 44)
        outs = [(torch.zeros(Bq, 10, dtype=torch.int32, device=dev), torch.zeros(Bq, 10, device=dev), torch.zeros(Bq, dtype=torch.int32, device=dev),
                 torch.zeros(Bq, dtype=torch.int32, device=dev)) for _ in range(2)]
        ss = [torch.cuda.Stream(), torch.cuda.Stream()]

        def launch(i):
            o = outs[i % 2]
            ix2.batch_search_device(Qb.data_ptr(), Bq, 10, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), ss[i % 2].cuda_stream)
        nrep = 12

        def timed_launches():
            for i in range(4):
                launch(i)
            torch.cuda.synchronize()
            t = time.time()
            for i in range(nrep):
                launch(i)
            torch.cuda.synchronize()
            return time.time() - t
        # round 5: the level table for quaternary codes (the small top levels from one i8 MFMA GEMM over the digits; same walk, same
        # bits).  Timed without it first (round 4's path), then with the automatic rule, the launches' answers compared.
        ix2.enable_timing(True)
        ix2.set_walk_table(0, 0)
        el_plain = timed_launches()
        plain_ids = outs[0][0].clone()
        ix2.set_walk_table()
        tab_info = ix2.walk_table_info()
        el = timed_launches()
        table_same = bool(torch.equal(plain_ids, outs[0][0]))
        stt = ix2.last_stats(ss[0].cuda_stream)
        ids2, sc2, cnt2 = ix2.batch_search(Qh, 10)
        rec_walk = float(np.mean([len(set(ids2[i].tolist()) & set(gt2[i].tolist())) / 10 for i in range(B)]))
        walk_dispatches = 1 + (len(ix2.walk_order_cuts()) if Bq >= ca.HNSWIndex.WALK_ORDER_DEFAULT_MIN_B else 0)   # level ranges of one launch
        row_b = 2 * ((d + 63) // 64) * 8 + 4
        alg = float(stt.evals * row_b + stt.adj_bytes)
        gbps = alg / (stt.walk_ms * 1e-3) / 1e9 if stt.walk_ms > 0 else 0.0
        walk = {"n": m, "build_s": t_build, "queries_per_launch": Bq, "launches_in_flight": 2, "qps": nrep * Bq / el, "ms_per_launch": el / nrep * 1e3,
                "level_table": {"level_min": tab_info[0], "columns": tab_info[1], "qps_without_table": nrep * Bq / el_plain,
                                "ms_per_launch_without_table": el_plain / nrep * 1e3, "same_ids_with_and_without": table_same},
                "evals_per_query": stt.evals / Bq, "recall_at_10_vs_f32_bruteforce": rec_walk,
                "recall_note": "quaternary planes are stored MSB first and multiplied LSB first (SURVEY App. C #4): the walk ranks by the reference's "
                               "own quaternary dot, low recall is the reference's behaviour, reproduced bit for bit",
                "roofline": {"bound": "hbm", "kernel": "walk_kernel<ENG_Q2> (196 B per evaluation: latency / issue, not bandwidth, bounds rows this short)",
                             "achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0,
                             "traffic": (lambda t: t * walk_dispatches if t else None)(bench.committed_kernel_traffic("c3_walk")),
                             "per_launch": {"algorithmic_bytes": alg, "avg_ms": stt.walk_ms, "evals": int(stt.evals), "expansions": int(stt.expansions)}}}
        if cpu_seconds > 0:   # the oracle on the same graph: CPU baseline + parity of the 256 answers, bit for bit
            cores = _cores()
            op = O.HNSWParams(dim=d, storage=O.STORAGE_SUBBYTE, resolution=2, range_lo=-1.0, range_hi=1.0, ef_search=ix2.hnsw_params.ef_search,
                              ef_construction=ix2.hnsw_params.ef_construction, num_layers=ix2.hnsw_params.num_layers)
            oix = O.OracleIndex(op).set_vectors(Xs.cpu().numpy())
            oix.import_graph(ix2.download_graph(), ix2.download_root())
            t = time.time()
            oi, osc, ocnt = oix.search_batch(Qh, 10, threads=cores)[:3]
            cpu_s = time.time() - t
            walk["cpu_baseline"] = {"value": B / cpu_s, "unit": "queries/s", "cores": cores, "kind": "port",
                                    "sample": f"the {B} queries on the same {m}-vector graph ({cpu_s:.2f} s wall on {cores} threads), oracle/ walk with "
                                              "dot_product_quaternary + exact rerank"}
            walk["parity_vs_oracle"] = {"queries": B, "id_mismatch_queries": int((ids2 != oi).any(axis=1).sum()),
                                        "score_bit_mismatches": int((sc2.view(np.uint32) != osc.view(np.uint32)).sum()),
                                        "count_mismatches": int((cnt2 != ocnt).sum())}
            del oix
        out["hnsw_walk_quaternary"] = walk
        del ix2
    del X
    torch.cuda.empty_cache()
    out["seconds"] = time.time() - t_all
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--walk-n", type=int, default=1_000_000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--cpu-seconds", type=float, default=5.0)
    a = ap.parse_args()
    print(json.dumps(run(a.n, a.dim, a.batch, a.walk_n, a.reps, a.cpu_seconds)))
