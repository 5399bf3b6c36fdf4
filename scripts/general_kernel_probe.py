#!/usr/bin/env python3
"""What the walk outside the fast kernels' domain costs (kernels_walk_general.hip): (a) 3072-dim u8 rows, ef 64; (b) 768-dim rows with
ef 2048; (c) 768-dim rows, M0 128 with shortlist_size 128 — walk kernel ms per launch, evaluations, the bytes they stand for."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import cosdata_amd as ca  # noqa: E402

dev = torch.device("cuda:0")
CASES = (("u8_3072_dims_ef64_three_chunk_passes_of_walk_kernel", 200_000, 3072, 64, 64, 32, 64, 8192), ("u8_4096_dims_ef64_four_chunk_passes_of_walk_kernel", 150_000, 4096, 64, 64, 32, 64, 8192), ("u8_4608_dims_ef64", 130_000, 4608, 64, 64, 32, 64, 8192), ("u8_768_dims_ef2048", 200_000, 768, 2048, 64, 32, 64, 2048),
         ("u8_768_dims_shortlist128", 200_000, 768, 64, 128, 64, 128, 8192), ("u8_768_dims_fast_kernels_ef64", 200_000, 768, 64, 64, 32, 64, 8192),
         ("u8_768_dims_fast_kernels_ef1024", 200_000, 768, 1024, 64, 32, 64, 2048))
for name, N, D, ef, M0, M, shortlist, B in CASES:
    gc = torch.Generator(device=dev)
    gc.manual_seed(4242)
    centers = torch.randn(max(64, N // 1000), D, generator=gc, device=dev)
    centers /= centers.norm(dim=1, keepdim=True)
    X = bench.mixture(torch, N, D, 42, dev, centers)
    Q = bench.mixture(torch, B, D, 43, dev, centers)
    vr = ca.sample_values_range(X[:1000].cpu().numpy(), 1.0)
    hp = ca.HNSWHyperParams(num_layers=7, ef_construction=128, ef_search=ef, level_0_neighbors_count=M0, neighbors_count=M)
    ix = ca.HNSWIndex(D, hp, ca.DistanceMetric.Cosine, ca.StorageType.UnsignedByte(), vr, shortlist_size=shortlist, device=0, seed=42)
    ix.upload_vectors_device(X.data_ptr(), N, keepalive=X)
    import time
    t = time.time()
    ix.build(4096)
    build_s = time.time() - t
    ix.enable_timing(True)
    o = (torch.zeros(B, 10, dtype=torch.int32, device=dev), torch.zeros(B, 10, dtype=torch.float32, device=dev),
         torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev))
    st = torch.cuda.Stream(device=dev)
    ws = []
    for _ in range(6):
        ix.batch_search_device(Q.data_ptr(), B, 10, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st.cuda_stream)
        st.synchronize()
        s = ix.last_stats(st.cuda_stream)
        ws.append(s.walk_ms)
    walk_ms = sorted(ws[1:])[len(ws[1:]) // 2]
    gt = ix.bruteforce_topk(Q[:512].cpu().numpy(), 10)[0]
    ids = o[0][:512].cpu().numpy()
    rec = sum(len(set(ids[i].tolist()) & set(gt[i].tolist())) for i in range(512)) / 5120.0
    byts = s.evals * (D + 4) + s.adj_bytes
    print(json.dumps({"case": name, "rows": N, "dim": D, "ef": ef, "M0": M0, "M": M, "shortlist": shortlist, "queries": B, "build_s": round(build_s, 2),
                      "walk_ms": round(walk_ms, 3), "qps_walk_only": round(B / walk_ms * 1e3), "evals_per_query": round(s.evals / B, 1),
                      "expansions_per_query": round(s.expansions / B, 1), "algorithmic_GBps": round(byts / walk_ms / 1e6, 1), "recall_at_10": round(rec, 4)}), flush=True)
    del ix, X, Q
    torch.cuda.empty_cache()
