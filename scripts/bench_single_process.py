#!/usr/bin/env python3
"""`bench.py --gpus N --single-process`: the reference's own deployment shape — ONE host process holding every shard
(`IndexOps::batch_search`, indexes/mod.rs:260-272, is called by one process) — through `cos_shardset_search_batch`:
host query buffer in, merged global top-k out; per shard quantize -> walk -> rerank on its own device, ONE grouped
ncclAllGather of the packed per-shard records (communicators from ncclCommInitAll), merge kernel.  Everything below the
call is C++/HIP; this script only makes data, builds the shards and times the call.

The entry point takes HOST buffers, so the rate printed here is PCIe-inclusive by construction (`value_is_pcie_inclusive`);
the one-process-per-GPU run of bench.py (queries resident in HBM) stays the contract's `value`."""
import json
import os
import sys
import time

import numpy as np


def run(args, metric):
    import torch
    import bench
    import cosdata_amd as ca
    from cosdata_amd.shardset import ShardSet
    N = args.gpus
    have = torch.cuda.device_count()
    if have < N and not args.allow_shared_devices:
        raise SystemExit(f"bench.py --single-process: --gpus {N} but only {have} GPU(s) are visible; refusing to run fewer devices than asked "
                         "(--allow-shared-devices places several shards on one device for a code-path check)")
    n, d, corpus, efc, desc = bench.WORKLOADS[args.workload]
    n = args.n or n
    k, B = args.top_k, args.batch * max(1, args.coalesce)
    ef = (112 if args.workload == bench.MAIN_WORKLOAD else 64) if args.ef == "auto" else int(args.ef)
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    t_setup = time.time()
    shards, Xs = [], []
    values_range = None
    centers = {}
    for s in range(N):
        di = s % max(1, have)
        dev = torch.device(f"cuda:{di}")
        torch.cuda.set_device(di)
        if di not in centers:
            gc = torch.Generator(device=dev); gc.manual_seed(4242)
            c = torch.randn(max(64, (n * N) // 1000), d, generator=gc, device=dev)
            centers[di] = c / c.norm(dim=1, keepdim=True)
        X = bench.mixture(torch, n, d, 42 + 1000 * s, dev, centers[di])
        if values_range is None:
            values_range = ca.sample_values_range(X[:1000].cpu().numpy(), 1.0) if args.quantization == "auto" else (-1.0, 1.0)
        is_shard = args.workload == bench.MAIN_WORKLOAD   # the metric's shard takes the hyper-parameters that meet the target in reference semantics
        hp = ca.HNSWHyperParams(num_layers=9, ef_construction=args.ef_construction or efc, ef_search=ef,
                                level_0_neighbors_count=args.m0 or (bench.MAIN_M0 if is_shard else 64), neighbors_count=args.m or (bench.MAIN_M if is_shard else 32))
        ix = ca.HNSWIndex(d, hp, ca.DistanceMetric.Cosine, ca.StorageType.UnsignedByte(), values_range, device=di, id_base=s * n, seed=42 + s)
        ix.upload_vectors_device(X.data_ptr(), n, keepalive=X)
        ix.build(args.build_batch)
        shards.append(ix); Xs.append(X)
    ss = ShardSet(shards)
    dev0 = torch.device("cuda:0")
    torch.cuda.set_device(0)
    Qd = bench.mixture(torch, 2 * B, d, 43, dev0, centers[0])
    Qh = Qd.cpu().numpy()
    # recall of the merged answer vs the exact global top-k (per-shard exhaustive scans merged by exact score)
    nrq = min(args.recall_queries, 2048)
    Qr = bench.mixture(torch, nrq, d, 45, dev0, centers[0]).cpu().numpy()
    gi = np.concatenate([sh.bruteforce_topk(Qr, k)[0] for sh in shards], axis=1)
    gs = np.concatenate([sh.bruteforce_topk(Qr, k)[1] for sh in shards], axis=1)
    order = np.argsort(-gs, axis=1, kind="stable")[:, :k]
    gt = np.take_along_axis(gi, order, axis=1)
    ids, sc, cnt = ss.batch_search(Qr, k)
    recall = float(np.mean([len(set(ids[i, :cnt[i]].tolist()) & set(gt[i].tolist())) / k for i in range(nrq)]))
    owners = sorted(set((ids[cnt[:, None] > np.arange(k)[None, :]] // n).tolist()))
    for i in range(max(0, args.warmup)):
        ss.batch_search(Qh[(i % 2) * B:(i % 2 + 1) * B], k)
    t = time.perf_counter()
    for i in range(max(1, args.steps)):
        ss.batch_search(Qh[(i % 2) * B:(i % 2 + 1) * B], k)
    el = time.perf_counter() - t
    steps = max(1, args.steps)
    out = {"metric": metric, "value": steps * B / el, "unit": "queries/s", "n_gpus": N, "steps": steps, "warmup": max(0, args.warmup),
           "ms_per_step": el / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": args.workload + ": " + desc, "mode": "single process, cos_shardset_search_batch (host buffers in / merged top-k out)",
                      "vectors_per_gpu": n, "dim": d, "queries_per_step": B, "top_k": k, "ef_search": ef, "shards": N, "devices_visible": have,
                      "shards_share_devices": have < N, "parallelism": f"id-range shards x{N}, one host process, grouped ncclAllGather + merge kernel"},
           "value_is_pcie_inclusive": True, "recall_at_10": recall, "recall_queries": nrq, "shards_in_merged_answers": owners,
           "global_corpus_vectors": n * N, "setup_seconds": time.time() - t_setup}
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    ss.close()
