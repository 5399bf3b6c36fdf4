#!/usr/bin/env python3
"""One 256-query batch at a time on a small c2-like index through the four-wave latency walk (and the one-wave one): the workload for
`rocprofv3 --kernel-trace` / `--pmc SQ_*` passes while tuning kernels_walk_lat4.hip."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cosdata_amd as ca
from bench import mixture
dev = torch.device("cuda:0")
n, d, B, k = int(os.environ.get("N", 300_000)), 768, 256, 10
g = torch.Generator(device=dev); g.manual_seed(41)
c = torch.randn(max(64, n // 1000), d, generator=g, device=dev); c = c / c.norm(dim=1, keepdim=True)
X = mixture(torch, n, d, 42, dev, c); Q = mixture(torch, B, d, 43, dev, c)
vr = ca.sample_values_range(X[:1000].cpu().numpy(), 1.0)
ix = ca.HNSWIndex(d, ca.HNSWHyperParams(ef_search=int(os.environ.get("EF", 64))), ca.DistanceMetric.Cosine, ca.StorageType.UnsignedByte(), vr, 64, device=0, seed=42)
ix.upload_vectors_device(X.data_ptr(), n, keepalive=X)
ix.set_latency_waves(0)       # the build's walks stay out of the picture
ix.set_latency_mode(0)
ix.build(4096)
s = torch.cuda.Stream(device=dev)
o_i = torch.zeros(B, k, dtype=torch.int32, device=dev); o_s = torch.zeros(B, k, device=dev)
o_c = torch.zeros(B, dtype=torch.int32, device=dev); o_t = torch.zeros(B, dtype=torch.int32, device=dev)
for name, lat, waves in (("one wave", 0xFFFFFFFF, 0), ("four waves", 0xFFFFFFFF, 0xFFFFFFFF)):
    ix.set_latency_mode(lat); ix.set_latency_waves(waves)
    for _ in range(3):
        ix.batch_search_device(Q.data_ptr(), B, k, o_i.data_ptr(), o_s.data_ptr(), o_c.data_ptr(), o_t.data_ptr(), s.cuda_stream)
    s.synchronize()
    t = time.perf_counter()
    for _ in range(30):
        ix.batch_search_device(Q.data_ptr(), B, k, o_i.data_ptr(), o_s.data_ptr(), o_c.data_ptr(), o_t.data_ptr(), s.cuda_stream)
    s.synchronize()
    print(name, "single batch ms", (time.perf_counter() - t) / 30 * 1e3)
