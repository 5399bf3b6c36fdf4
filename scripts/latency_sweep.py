#!/usr/bin/env python3
"""Latency variants of the walk (one wave per query: kernels_walk_lat.hip; four waves per query: kernels_walk_lat4.hip) against the
throughput kernel on the c2 index (1M x 768 u8, ef 64 and 256): ms per launch for launches of 64 .. 16384 queries, results compared
bit for bit at every size.
Writes one JSON line per (ef, B, variant)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cosdata_amd as ca
from bench import mixture

dev = torch.device("cuda:0")
n, d, k = int(os.environ.get("SWEEP_N", 1_000_000)), 768, 10
g = torch.Generator(device=dev); g.manual_seed(41)
c = torch.randn(1000, d, generator=g, device=dev); c = c / c.norm(dim=1, keepdim=True)
X = mixture(torch, n, d, 42, dev, c)
BMAX = 16384
Q = mixture(torch, BMAX, d, 43, dev, c)
vr = ca.sample_values_range(X[:1000].cpu().numpy(), 1.0)
ix = ca.HNSWIndex(d, ca.HNSWHyperParams(), ca.DistanceMetric.Cosine, ca.StorageType.UnsignedByte(), vr, 64, device=0, seed=42)
ix.upload_vectors_device(X.data_ptr(), n, keepalive=X)
t0 = time.time(); ix.build(4096); build_s = time.time() - t0
s = torch.cuda.Stream(device=dev)
o_i = torch.zeros(BMAX, k, dtype=torch.int32, device=dev); o_s = torch.zeros(BMAX, k, device=dev)
o_c = torch.zeros(BMAX, dtype=torch.int32, device=dev); o_t = torch.zeros(BMAX, dtype=torch.int32, device=dev)
print(json.dumps({"build_seconds": build_s, "n": n, "dim": d}), flush=True)


def run(B, reps):
    for _ in range(2):
        ix.batch_search_device(Q.data_ptr(), B, k, o_i.data_ptr(), o_s.data_ptr(), o_c.data_ptr(), o_t.data_ptr(), s.cuda_stream)
    s.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        ix.batch_search_device(Q.data_ptr(), B, k, o_i.data_ptr(), o_s.data_ptr(), o_c.data_ptr(), o_t.data_ptr(), s.cuda_stream)
    s.synchronize()
    ms = (time.perf_counter() - t) / reps * 1e3
    st = ix.last_stats(s.cuda_stream)
    return ms, (o_i[:B].clone(), o_s[:B].clone().view(torch.int32), o_c[:B].clone(), o_t[:B].clone()), st


for ef in (64, 256):
    ix.set_ef_search(ef)
    for B in [int(x) for x in os.environ.get("SWEEP_BS", "64,256,1024,2048,4096,8192,16384").split(",")]:
        reps = max(3, min(40, 16384 // B))
        ix.set_latency_waves(0)
        ix.set_latency_mode(0)
        ms0, ref, st0 = run(B, reps)
        print(json.dumps({"ef": ef, "B": B, "variant": "throughput", "ms": ms0, "qps": B / ms0 * 1e3, "evals_per_q": st0.evals / B,
                          "pops_per_q": st0.expansions / B, "rounds_per_q": st0.reserved / B}), flush=True)
        for la in [int(x) for x in os.environ.get("SWEEP_LA", "4").split(",")]:
            os.environ["COS_WALK_LAT_LA"] = str(la)
            ix.set_latency_mode(0xFFFFFFFF)
            ms1, out, st1 = run(B, reps)
            same = all(bool(torch.equal(a, b)) for a, b in zip(ref, out))
            print(json.dumps({"ef": ef, "B": B, "variant": f"latency la={la}", "ms": ms1, "qps": B / ms1 * 1e3, "speedup": ms0 / ms1,
                              "identical_to_throughput_kernel": same, "evals_per_q": st1.evals / B, "pops_per_q": st1.expansions / B,
                              "rounds_per_q": st1.reserved / B}), flush=True)
            assert same, (ef, B, la)
        if B <= int(os.environ.get("SWEEP_LAT4_MAX_B", 4096)):
            ix.set_latency_mode(0xFFFFFFFF)
            ix.set_latency_waves(0xFFFFFFFF)
            ms4, out4, st4 = run(B, reps)
            same = all(bool(torch.equal(a, b)) for a, b in zip(ref, out4))
            print(json.dumps({"ef": ef, "B": B, "variant": "four waves per query", "ms": ms4, "qps": B / ms4 * 1e3, "speedup": ms0 / ms4,
                              "identical_to_throughput_kernel": same, "evals_per_q": st4.evals / B, "pops_per_q": st4.expansions / B,
                              "rounds_per_q": st4.reserved / B}), flush=True)
            assert same, (ef, B, "lat4")
            ix.set_latency_waves(0)
