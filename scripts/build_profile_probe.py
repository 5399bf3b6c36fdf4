#!/usr/bin/env python3
"""cos_index_build with the build_profile knob on two shapes (1M x 768, 64 / 32 and 2M x 1024, 256 / 64): batches, link rounds, walk and link
seconds as the library prints them to stderr."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench, cosdata_amd as ca
from cosdata_amd import _lib
dev = torch.device("cuda:0")
for N, D, M0, M in ((1_000_000, 768, 64, 32), (2_000_000, 1024, 256, 64)):
    gc = torch.Generator(device=dev); gc.manual_seed(4242)
    centers = torch.randn(max(64, N // 1000), D, generator=gc, device=dev); centers /= centers.norm(dim=1, keepdim=True)
    X = bench.mixture(torch, N, D, 42, dev, centers)
    vr = ca.sample_values_range(X[:1000].cpu().numpy(), 1.0)
    hp = ca.HNSWHyperParams(num_layers=9, ef_construction=128 if M0 == 64 else 256, ef_search=64, level_0_neighbors_count=M0, neighbors_count=M)
    ix = ca.HNSWIndex(D, hp, ca.DistanceMetric.Cosine, ca.StorageType.UnsignedByte(), vr, shortlist_size=64, device=0, seed=42)
    ix.upload_vectors_device(X.data_ptr(), N, keepalive=X)
    _lib.tuning_set("build_profile", 1)
    t = time.time(); ix.build(4096); print(N, D, M0, M, "build_s", round(time.time() - t, 2), flush=True)
    del ix, X; torch.cuda.empty_cache()
