#!/usr/bin/env python3
"""Launch-shape sweep on ONE resident index (build once): queries per launch x launches in flight x ef -> end-to-end QPS and the
walk kernel's HIP-event duration / algorithmic GB/s.  Diagnostic for DESIGN.md §4.1 (how much of the headline comes from
overlapping launches vs from one launch filling the chip)."""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cosdata_amd as ca
from bench import mixture

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--queries", default="256,1024,4096,8192,16384,32768,65536")
ap.add_argument("--inflight", default="1,2")
ap.add_argument("--ef", default="64,256")
ap.add_argument("--visited", default="ref")
ap.add_argument("--launches", type=int, default=12)
ap.add_argument("--ef-construction", type=int, default=128)
args = ap.parse_args()
dev = torch.device("cuda:0")
n, d, k = args.n, args.dim, 10
gc = torch.Generator(device=dev); gc.manual_seed(4242)
centers = torch.randn(max(64, n // 1000), d, generator=gc, device=dev); centers /= centers.norm(dim=1, keepdim=True)
X = mixture(n, d, 42, dev, centers)
qs = [int(v) for v in args.queries.split(",")]
Q = mixture(max(qs) * 2, d, 43, dev, centers)
vr = ca.sample_values_range(X[:1000].cpu().numpy(), 1.0)
hp = ca.HNSWHyperParams(num_layers=9, ef_construction=args.ef_construction, ef_search=64)
ix = ca.HNSWIndex(d, hp, ca.DistanceMetric.Cosine, ca.StorageType.UnsignedByte(), vr, seed=42)
ix.upload_vectors_device(X.data_ptr(), n, keepalive=X)
t0 = time.time(); ix.build(4096); build_s = time.time() - t0
ix.set_visited_mode(1 if args.visited == "exact" else 0)
print(json.dumps({"build_seconds": build_s, "n": n, "dim": d}), flush=True)
for ef in [int(v) for v in args.ef.split(",")]:
    ix.set_ef_search(ef)
    for B in qs:
        for S in [int(v) for v in args.inflight.split(",")]:
            streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
            oi = torch.zeros(S, B, k, dtype=torch.int32, device=dev); osc = torch.zeros(S, B, k, device=dev)
            oc = torch.zeros(S, B, dtype=torch.int32, device=dev); ost = torch.zeros(S, B, dtype=torch.int32, device=dev)
            nl = max(args.launches, 2 * S) if B >= 4096 else args.launches * 4
            def go(i):
                s = i % S
                q = Q[(i % 2) * B:(i % 2 + 1) * B]
                ix.batch_search_device(q.data_ptr(), B, k, oi[s].data_ptr(), osc[s].data_ptr(), oc[s].data_ptr(), ost[s].data_ptr(), streams[s].cuda_stream)
            for i in range(2 * S): go(i)
            torch.cuda.synchronize()
            ix.enable_timing(True)
            t = time.perf_counter()
            for i in range(nl): go(i)
            torch.cuda.synchronize()
            el = time.perf_counter() - t
            ws = 0.0; cnt = 0; byts = []
            for s in range(S):
                ts = ix.timing_summary(streams[s].cuda_stream); ws += ts.walk_ms_sum; cnt += ts.launches
                st = ix.last_stats(streams[s].cuda_stream); byts.append(st.evals * (d + 4) + st.adj_bytes)
            ix.enable_timing(False)
            wms = ws / cnt; b = float(np.mean(byts))
            print(json.dumps({"ef": ef, "queries_per_launch": B, "inflight": S, "qps": nl * B / el, "walk_ms": wms,
                              "kernel_GBps": b / wms / 1e6, "aggregate_GBps": b * nl / el / 1e9, "ms_per_launch_wall": el / nl * 1e3}), flush=True)
