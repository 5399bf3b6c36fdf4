#!/usr/bin/env python3
"""One 256-query batch on c2's graph: walk kernel time, expansions and ROUNDS per query (lookahead windows issued), per kernel variant.
PROBE_N=30000 gives the control whose corpus sits in L2: what a round costs without HBM."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import cosdata_amd as ca  # noqa: E402

N, D = int(os.environ.get("PROBE_N", 1_000_000)), 768
EFS = [int(x) for x in os.environ.get("PROBE_EFS", "64,256").split(",")]
B, K = int(os.environ.get("PROBE_B", 256)), 10
dev = torch.device("cuda:0")
gc = torch.Generator(device=dev)
gc.manual_seed(4242)
centers = torch.randn(max(64, N // 1000), D, generator=gc, device=dev)
centers /= centers.norm(dim=1, keepdim=True)
X = bench.mixture(torch, N, D, 42, dev, centers)
Q = bench.mixture(torch, B, D, 43, dev, centers)
vr = ca.sample_values_range(X[:1000].cpu().numpy(), 1.0)
hp = ca.HNSWHyperParams(num_layers=9, ef_construction=128, ef_search=EFS[0], level_0_neighbors_count=64, neighbors_count=32)
ix = ca.HNSWIndex(D, hp, ca.DistanceMetric.Cosine, ca.StorageType.UnsignedByte(), vr, shortlist_size=64, device=0, seed=42)
ix.upload_vectors_device(X.data_ptr(), N, keepalive=X)
ix.build(4096)
ix.enable_timing(True)
o = (torch.zeros(B, K, dtype=torch.int32, device=dev), torch.zeros(B, K, dtype=torch.float32, device=dev),
     torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev))
st = torch.cuda.Stream(device=dev)
VARIANTS = (("throughput_auto_table", 0, 0, 2, {}), ("four_waves_auto_table", 2048, 512, 2, {"walk_small_table_tk": 0}), ("four_waves", 2048, 512, 0, {}))
for ef in EFS:
    ix.set_ef_search(ef)
    ref = None
    for name, lat, lat4, tab, knobs in VARIANTS:
        ix.set_latency_mode(lat)
        ix.set_latency_waves(lat4)
        ix.set_walk_table(ca.HNSWIndex.WALK_TABLE_AUTO if tab else 0, 1 if tab else 0)
        ix.set_walk_order(1 if tab == 3 else 0)
        ca._lib.tuning_clear(None)
        for kk, vv in knobs.items():
            ca._lib.tuning_set(kk, vv)
        ws = []
        for _ in range(12):
            ix.batch_search_device(Q.data_ptr(), B, K, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st.cuda_stream)
            st.synchronize()
            s = ix.last_stats(st.cuda_stream)
            ws.append((s.walk_ms, s.prep_ms, s.finalize_ms))
        ws = sorted(ws[2:])
        med = ws[len(ws) // 2]
        res = torch.cat([o[0].flatten(), o[1].view(torch.int32).flatten()]).clone()
        same = True if ref is None else bool(torch.equal(res, ref))
        ref = res if ref is None else ref
        sp = ix.last_walk_split(st.cuda_stream)
        print(json.dumps({"ef": ef, "variant": name, "walk_ms": round(med[0], 4), "prep_ms": round(med[1], 4), "finalize_ms": round(med[2], 4),
                          "expansions_per_query": round(s.expansions / B, 1), "rounds_per_query": round(s.reserved / B, 1),
                          "evals_per_query": round(s.evals / B, 1), "us_per_round": round(med[0] * 1e3 / max(1.0, s.reserved / B), 3),
                          "table_ms": round(sp.table_ms, 4), "upper_ms": round(sp.upper_ms,4), "lower_ms": round(sp.lower_ms,4), "sort_ms": round(sp.sort_ms,4), "cut": sp.cut_after_level, "upper_exp": sp.upper_expansions/B, "lower_exp": sp.lower_expansions/B, "table_level_min": sp.table_level_min, "identical": same}), flush=True)
