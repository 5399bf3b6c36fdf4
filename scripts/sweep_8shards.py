#!/usr/bin/env python3
"""Merged recall@10 of the 8-shard proxy (scripts/bench_c4_8shards.py) over hyper-parameter sets on ONE corpus + ground truth.
SWEEP="m0:m:efc:ef;..." — one JSON line per set."""
import json
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import cosdata_amd as ca  # noqa: E402
from scripts import bench_c4_8shards  # noqa: E402

args = types.SimpleNamespace(top_k=10, batch=256, coalesce=128, inflight=2, recall_queries=4096, steps=2, warmup=0)
env = types.SimpleNamespace(args=args, torch=torch, dev=torch.device("cuda:0"), rank=0, world=1, local_rank=0, dist_on=False)
c4 = bench.DenseWorkload(env, "c4shard", n_override=int(os.environ.get("SWEEP_N", 0)))
hp0 = ca.HNSWHyperParams(num_layers=9, ef_construction=128, ef_search=64)
ix0 = ca.HNSWIndex(c4.d, hp0, ca.DistanceMetric.Cosine, ca.StorageType.UnsignedByte(), c4.values_range, device=0, seed=42)
ix0.upload_vectors_device(c4.X.data_ptr(), c4.n, keepalive=c4.X)
c4._ground_truth(ix0)
del ix0
for spec in os.environ.get("SWEEP", "128:64:256:96;256:128:256:96;128:32:128:96").split(";"):
    m0, m, efc, ef = (int(x) for x in spec.split(":"))
    print(json.dumps(bench_c4_8shards.run(c4, ef=ef, m0=m0, m=m, ef_construction=efc, recall_only=True)), flush=True)
