#!/usr/bin/env python3
"""One un-coalesced 256-query batch at a time (the reference's literal unit of work, indexes/mod.rs:260-272) on c2: ms per batch for
the four-wave latency kernel, the one-wave latency kernel, the throughput kernel, and the throughput kernel with the level table
(cos_index_set_walk_table with min_queries = 1), at PROBE_EFS.  Every variant must return the same bits."""
import json
import os
import sys
import time

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
o = (torch.zeros(B, K, dtype=torch.int32, device=dev), torch.zeros(B, K, dtype=torch.float32, device=dev),
     torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev))
st = torch.cuda.Stream(device=dev)


def run(reps=20):
    for _ in range(3):
        ix.batch_search_device(Q.data_ptr(), B, K, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st.cuda_stream)
    st.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        ix.batch_search_device(Q.data_ptr(), B, K, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st.cuda_stream)
    st.synchronize()
    return (time.perf_counter() - t) / reps * 1e3, torch.cat([o[0].flatten(), o[1].view(torch.int32).flatten()]).clone()


for ef in EFS:
    ix.set_ef_search(ef)
    ref = None
    for name, lat, lat4, tab in (("four_waves_per_query", 2048, 512, 0), ("four_waves_per_query_with_level_table", 2048, 512, 1),
                                 ("one_wave_latency_kernel", 2048, 0, 0), ("throughput_kernel", 0, 0, 0), ("throughput_kernel_with_level_table", 0, 0, 1),
                                 ("four_waves_per_query_with_the_automatic_table", 2048, 512, 2), ("throughput_kernel_with_the_automatic_table", 0, 0, 2)):
        ix.set_latency_mode(lat)
        ix.set_latency_waves(lat4)
        ix.set_walk_table((ca.HNSWIndex.WALK_TABLE_AUTO if tab == 2 else 8192) if tab else 0, 1 if tab else 0)
        ms, res = run()
        same = True if ref is None else bool(torch.equal(res, ref))
        ref = res if ref is None else ref
        print(json.dumps({"ef": ef, "queries": B, "variant": name, "ms_per_batch": round(ms, 4), "qps": round(B / ms * 1e3), "identical": same}), flush=True)
