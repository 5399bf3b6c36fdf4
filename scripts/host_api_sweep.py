#!/usr/bin/env python3
"""cos_search_batch (host buffers in / out, PCIe-inclusive) on the c2 index: concurrent synchronous callers x pageable / pinned
host memory, against the resident-query rate of the same launches.  One JSON line."""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import cosdata_amd as ca

n, d, B, k = int(os.environ.get("N", 1_000_000)), 768, 32768, 10
dev = torch.device("cuda:0")
gc = torch.Generator(device=dev); gc.manual_seed(4242)
centers = torch.randn(max(64, n // 1000), d, generator=gc, device=dev); centers /= centers.norm(dim=1, keepdim=True)
X = bench.mixture(torch, n, d, 42, dev, centers)
Q = bench.mixture(torch, 4 * B, d, 43, dev, centers)
vr = ca.sample_values_range(X[:1000].cpu().numpy(), 1.0)
ix = ca.HNSWIndex(d, ca.HNSWHyperParams(ef_search=64), ca.DistanceMetric.Cosine, ca.StorageType.UnsignedByte(), vr)
ix.upload_vectors_device(X.data_ptr(), n, keepalive=X)
ix.build(4096)
# resident reference: 2 streams, queries in HBM
S = 2
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
o = [(torch.zeros(B, k, dtype=torch.int32, device=dev), torch.zeros(B, k, device=dev), torch.zeros(B, dtype=torch.int32, device=dev),
      torch.zeros(B, dtype=torch.int32, device=dev)) for _ in range(S)]
def resident(reps):
    for i in range(reps):
        s = i % S
        ix.batch_search_device(Q[(i % 4) * B:(i % 4 + 1) * B].data_ptr(), B, k, o[s][0].data_ptr(), o[s][1].data_ptr(), o[s][2].data_ptr(), o[s][3].data_ptr(),
                               streams[s].cuda_stream)
    torch.cuda.synchronize()
resident(4)
t = time.perf_counter(); resident(16); res_qps = 16 * B / (time.perf_counter() - t)
out = {"resident_qps": res_qps, "host": []}
qp = [Q[j * B:(j + 1) * B].cpu().numpy() for j in range(4)]
pin = [torch.empty(B, d, pin_memory=True) for _ in range(4)]
for j in range(4):
    pin[j].copy_(Q[j * B:(j + 1) * B])
qpin = [p.numpy() for p in pin]
for name, bufs in (("pageable", qp), ("pinned", qpin)):
    for callers in (1, 2, 3, 4):
        reps = 6
        def call(j):
            for _ in range(reps):
                ix.batch_search(bufs[j], k)
        for j in range(callers):
            ix.batch_search(bufs[j], k)
        th = [threading.Thread(target=call, args=(j,)) for j in range(callers)]
        t = time.perf_counter()
        [x.start() for x in th]; [x.join() for x in th]
        el = time.perf_counter() - t
        out["host"].append({"memory": name, "callers": callers, "qps": callers * reps * B / el, "frac_of_resident": callers * reps * B / el / res_qps})
print(json.dumps(out))
