#!/usr/bin/env python3
"""bench.py — QPS of the dense HNSW search path on MI355X (driver contract: see the task prompt).

A "step" is one pass of the hot path over one query batch: cos_search_batch_device() =
quantize -> HNSW walk (every level, ef_search) -> dedup/top-5k -> exact f32 rerank -> top-k,
with the index and the queries resident in HBM.  Like the reference's own RPS harness
(tests/rps-test.py: 32 client threads x batches of 200) several batches are kept in flight on
the device: `--coalesce` client batches of 256 are fused into one launch (server-side dynamic
batching: the walk kernel gives every query one wavefront, so a 256-CU chip needs thousands of
queries per launch) and `--inflight` launches overlap on separate HIP streams.  The rate of one
un-coalesced 256-query batch at a time is reported alongside (`single_batch_qps`).

Default workload = BASELINE.json configs[1]: 1M x 768 dense cosine HNSW, query batch 256, one GPU
(`--workload c4shard` = one 12.5M x 1024 shard of configs[3]).  N > 1: one process per GPU, every
rank owns an independent shard (ID-range partition, weak scaling), queries are replicated, and each
step ends with the RCCL all-gather of the per-shard top-k + the S-way merge kernel.  The job's corpus grows with N
(N x vectors_per_gpu) while every rank does the same work per step, so `value` counts the units ALL ranks processed:
N x (queries searched over one shard) per second — the weak-scaling aggregate; the rate at which merged answers over
the N-times-larger corpus come out is reported next to it as `merged_qps` (= value / N).

Synthetic data (no network): a seeded Gaussian-mixture corpus, L2-normalised, generated on the
device; queries are fresh draws from the same mixture.  recall@10 is measured against exact
brute-force cosine on the same corpus, outside the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL / hipIpcGetMemHandle fail in legacy mode); the variable is
# normally exported already — keep it if the launcher dropped it.  Must be set before the HIP runtime initialises.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import cosdata_amd as ca  # noqa: E402
from cosdata_amd.sharding import (allgather_packed, global_topk_by_score, merge_topk_packed_device, packed_views,  # noqa: E402
                                  packed_words)

METRIC = "QPS at recall@10≥0.95, 1024-dim dense cosine, 1/2/4/8 MI355X"
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (n per GPU, dim, description)
    "c2": (1_000_000, 768, "BASELINE configs[1]: 1M x 768 dense cosine HNSW, query-batch 256, u8 auto-quantized storage"),
    "c4shard": (12_500_000, 1024, "one 12.5M x 1024 shard of BASELINE configs[3] (100M x 1024 over 8 GPUs)"),
    "smoke": (50_000, 768, "reduced-size plumbing run (NOT a benchmark number)"),
}


def mixture(n, d, seed, device, centers, sigma=0.8):
    """Gaussian mixture around shared unit-norm centres, L2-normalised (cf. tests/test-dataset.py:414-430)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty(n, d, device=device, dtype=torch.float32)
    step = 1 << 18
    for s in range(0, n, step):
        m = min(step, n - s)
        a = torch.randint(0, centers.shape[0], (m,), generator=g, device=device)
        x = centers[a] + (sigma / d ** 0.5) * torch.randn(m, d, generator=g, device=device)
        out[s:s + m] = x / x.norm(dim=1, keepdim=True)
    return out


def bruteforce_top10(X, Q, k=10):
    """exact cosine top-k by torch matmul (independent cross-check of the engine's own ground truth).
    Column chunks of 1M: torch.topk over multi-million-wide rows was observed to be unreliable."""
    ids = []
    for s in range(0, Q.shape[0], 256):
        ci, cs = [], []
        for c in range(0, X.shape[0], 1 << 20):
            t = (Q[s:s + 256] @ X[c:c + (1 << 20)].T).topk(k, dim=1)
            ci.append(t.indices + c)
            cs.append(t.values)
        ci, cs = torch.cat(ci, 1), torch.cat(cs, 1)
        ids.append(torch.gather(ci, 1, cs.topk(k, dim=1).indices))
    return torch.cat(ids)


def launch_shape(steps: int, warmup: int, coalesce: int):
    """(client batches per launch, timed launches, warm-up launches).  To time EXACTLY `steps` client batches the launch size
    is the largest divisor of `steps` not above `coalesce` — unless that would shrink launches below a quarter of `coalesce`
    (then the step count is rounded up to whole launches and the JSON line reports the number actually timed)."""
    C = max(1, coalesce)
    div = max(c for c in range(1, C + 1) if steps % c == 0)
    if div * 4 >= C:
        C = div
    return C, (steps + C - 1) // C, (warmup + C - 1) // C


def effective_cores():
    """CPU cores this process may actually use: scheduler affinity capped by the cgroup CPU quota (os.cpu_count() reports
    the host's logical CPUs even inside a container limited to a few of them)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                       # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = fh.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:   # cgroup v1
                q, per = float(fq.read()), float(fp.read())
                if q > 0 and per > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--n", type=int, default=0, help="override vectors per GPU (marks the run as non-standard)")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--coalesce", type=int, default=32, help="client batches fused per launch (dynamic batching)")
    ap.add_argument("--inflight", type=int, default=2, help="launches kept in flight (HIP streams)")
    ap.add_argument("--ef", default="auto", help="ef_search: an integer, or 'auto' = smallest of 32,48,64,96,128,192,256 whose "
                    "measured recall@10 is >= --recall-target (the metric is QPS AT recall@10 >= 0.95); config.toml default is 256")
    ap.add_argument("--recall-target", type=float, default=0.95)
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--visited", default="ref", choices=["ref", "exact"],
                    help="search-time visited filter: ref = PerformantFixedSet replica (ID parity with the reference), "
                         "exact = exact visited set (recall mode, not ID-identical); the graph is always built with ref")
    ap.add_argument("--ef-construction", type=int, default=128, help="config.toml default 128")
    ap.add_argument("--build-visited", default="ref", choices=["ref", "exact"], help="visited filter used by the builder's walks")
    ap.add_argument("--quantization", default="auto", choices=["auto", "range11"], help="auto = sampled values_range; range11 = (-1,1)")
    ap.add_argument("--ef-sweep", default="256,exact:32,exact:64",
                    help="comma list of extra settings timed after the main run on the same graph: '256' = ef_search 256 (config.toml "
                         "default) with the main visited filter; 'exact:32' / 'ref:128' = that ef with the named visited filter")
    ap.add_argument("--build-batch", type=int, default=4096)
    ap.add_argument("--recall-queries", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-probe", action="store_true", help="skip the empirical HBM ceiling probes (cos_hbm_probe)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    # COS_FORCE_DIST=1 exercises the N>1 code path (process group, all-gather, merge kernel) with a single rank
    force_dist = os.environ.get("COS_FORCE_DIST", "0") == "1"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or force_dist
    # stdout carries exactly ONE JSON line: libraries that print banners through C stdio (RCCL prints its version block on
    # fd 1 at exit) are pointed at stderr, the JSON line is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")

    n, d, desc = WORKLOADS[args.workload]
    if args.n:
        n = args.n
    Bc, k = args.batch, args.top_k                # Bc = client batch (a "step"); B = queries per launch
    ef = 256 if args.ef == "auto" else int(args.ef)
    C, n_launch, n_warm = launch_shape(args.steps, args.warmup, args.coalesce)
    B = Bc * C
    t_setup = time.time()

    # ---- synthetic shard + queries (resident in HBM) -------------------------------------------
    gc = torch.Generator(device=dev)
    gc.manual_seed(4242)
    n_centers = max(64, (n * world) // 1000)
    centers = torch.randn(n_centers, d, generator=gc, device=dev)
    centers /= centers.norm(dim=1, keepdim=True)
    X = mixture(n, d, 42 + 1000 * rank, dev, centers)            # this rank's shard: global ids [rank*n, (rank+1)*n)
    n_qsets = max(args.inflight, 2)
    Q = mixture(B * n_qsets, d, 43, dev, centers)                 # identical on every rank
    torch.cuda.synchronize()

    # ---- index: reference defaults (config.toml:20-24,32); "auto" quantization = u8 + values_range sampled from the
    # first sample_threshold embeddings (indexes/hnsw/mod.rs:202-351; tests/rps-test.py:73 uses the same mode) ---
    sample_threshold = 1000
    values_range = ca.sample_values_range(X[:sample_threshold].cpu().numpy(), 1.0) if args.quantization == "auto" else (-1.0, 1.0)
    if dist_on and world > 1:  # every shard must quantize with the same range: take rank 0's sample
        import torch.distributed as dist
        vr = torch.tensor(values_range, device=dev, dtype=torch.float64)
        dist.broadcast(vr, 0)
        values_range = (float(vr[0].item()), float(vr[1].item()))
    hp = ca.HNSWHyperParams(num_layers=9, ef_construction=args.ef_construction, ef_search=ef, level_0_neighbors_count=64, neighbors_count=32)
    ix = ca.HNSWIndex(d, hp, ca.DistanceMetric.Cosine, ca.StorageType.UnsignedByte(), values_range, shortlist_size=64,
                      device=local_rank, id_base=rank * n, seed=42 + rank,
                      visited_mode=ca.VISITED_EXACT if args.build_visited == "exact" else ca.VISITED_REF)
    ix.upload_vectors_device(X.data_ptr(), n, keepalive=X)
    t0 = time.time()
    ix.build(args.build_batch)
    build_s = time.time() - t0
    ix.set_visited_mode(ca.VISITED_EXACT if args.visited == "exact" else ca.VISITED_REF)

    # ---- buffers + streams ------------------------------------------------------------------------
    S = args.inflight
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    # the shard's result lives in ONE packed record [ids | scores | counts] so the sharded path exchanges it
    # with a single all-gather per launch (sharding.py)
    o_pack = torch.zeros(S, packed_words(B, k), dtype=torch.int32, device=dev)
    o_views = [packed_views(o_pack[s], B, k) for s in range(S)]
    o_ids = [v[0] for v in o_views]
    o_sc = [v[1] for v in o_views]
    o_cnt = [v[2] for v in o_views]
    o_st = torch.zeros(S, B, dtype=torch.int32, device=dev)
    if dist_on:
        g_pack = torch.zeros(S, world, packed_words(B, k), dtype=torch.int32, device=dev)
        m_ids = torch.zeros(S, B, k, dtype=torch.int32, device=dev)
        m_sc = torch.zeros(S, B, k, dtype=torch.float32, device=dev)
        m_cnt = torch.zeros(S, B, dtype=torch.int32, device=dev)
    lib = ca._lib.lib()

    def step(i):
        s = i % S
        st = streams[s]
        q = Q[(i % n_qsets) * B:(i % n_qsets + 1) * B]
        ix.batch_search_device(q.data_ptr(), B, k, o_ids[s].data_ptr(), o_sc[s].data_ptr(), o_cnt[s].data_ptr(), o_st[s].data_ptr(),
                               st.cuda_stream)
        if dist_on:  # per-shard top-k -> RCCL all-gather over xGMI -> S-way merge
            with torch.cuda.stream(st):
                allgather_packed(o_pack[s], g_pack[s])
            merge_topk_packed_device(g_pack[s], B, k, m_ids[s], m_sc[s], m_cnt[s], local_rank, st.cuda_stream)

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist_on:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize(dev)

    # ---- recall@10 vs exact brute force (outside the timed region) ---------------------------------
    nrq = min(args.recall_queries, B * n_qsets)
    # ground truth: the engine's own exhaustive scan (f32-MFMA GEMM -> top-64 -> reference-order re-score),
    # cross-checked against an independent torch matmul + topk
    Qh_gt = Q[:nrq].cpu().numpy()
    torch.cuda.synchronize(dev)
    t_bf = time.perf_counter()
    bf_ids, _bf_sc = ix.bruteforce_topk(Qh_gt, k)
    bf_seconds = time.perf_counter() - t_bf
    gt_local = torch.from_numpy((bf_ids.astype(np.int64) - rank * n)).to(dev)
    gt_torch = bruteforce_top10(X, Q[:nrq], k)
    gt_agree = float((gt_local.unsqueeze(2) == gt_torch.unsqueeze(1)).any(dim=2).float().mean().item())
    flat = {"queries": nrq, "seconds": bf_seconds, "tflops_end_to_end": 2.0 * nrq * n * d / bf_seconds / 1e12,
            "agreement_with_torch_topk": gt_agree,
            "note": "cos_bruteforce_topk incl. H2D/D2H, selection and exact re-score; kernel-level MFMA rate: profiles/"}
    ann = torch.zeros(nrq, k, dtype=torch.int64, device=dev)

    def merge_global(local_ids_global):
        """global answer = merge of the per-shard lists by exact cosine (recall bookkeeping only)"""
        import torch.distributed as dist
        loc = (local_ids_global - rank * n).clamp_(0, n - 1)
        sims = torch.einsum("qd,qkd->qk", Q[:nrq], X[loc])   # exact cosine of the k returned rows only (unit-norm corpus)
        return global_topk_by_score(local_ids_global, sims, k)

    gt = merge_global(gt_local + rank * n) if dist_on else gt_local

    def measure_recall(ef_value):
        ix.set_ef_search(ef_value)
        for s0 in range(0, nrq, B):
            m = min(B, nrq - s0)
            ix.batch_search_device(Q[s0:s0 + m].data_ptr(), m, k, o_ids[0].data_ptr(), o_sc[0].data_ptr(), o_cnt[0].data_ptr(),
                                   o_st[0].data_ptr(), streams[0].cuda_stream)
            streams[0].synchronize()
            ann[s0:s0 + m] = o_ids[0][:m].to(torch.int64) & 0xFFFFFFFF
        ann_g = merge_global(ann) if dist_on else ann
        hits = (ann_g.unsqueeze(2) == gt.unsqueeze(1)).any(dim=2).float().sum(dim=1)
        r = hits.mean() / k
        if dist_on:
            import torch.distributed as dist
            dist.broadcast(r, 0)  # every rank must take the same decision
        return float(r.item())

    ef_table = []
    if args.ef == "auto":
        for cand in (32, 48, 64, 96, 128, 192, 256):
            r = measure_recall(cand)
            ef_table.append({"ef_search": cand, "recall_at_10": r})
            ef = cand
            if r >= args.recall_target:
                break
    recall = measure_recall(ef)
    status_bad = int((o_st != 0).sum().item())

    # ---- size-independent properties of the returned lists, checked at the full workload size (no oracle needed): ids in
    # this shard's range and unique per query, scores non-increasing, every score = the exact f32 cosine of the row it names
    props = None
    if True:
        m = min(B, nrq)
        ix.batch_search_device(Q[:m].data_ptr(), m, k, o_ids[0].data_ptr(), o_sc[0].data_ptr(), o_cnt[0].data_ptr(), o_st[0].data_ptr(),
                               streams[0].cuda_stream)
        streams[0].synchronize()
        cnt = o_cnt[0][:m].to(torch.int64)
        valid = torch.arange(k, device=dev)[None, :] < cnt[:, None]
        gid = o_ids[0][:m].to(torch.int64) & 0xFFFFFFFF
        loc = gid - rank * n
        in_range = bool(((loc >= 0) & (loc < n))[valid].all().item())
        srt = torch.where(valid, gid, -1 - torch.arange(k, device=dev)[None, :].expand(m, k)).sort(dim=1).values
        unique = bool((srt[:, 1:] != srt[:, :-1]).all().item())
        sc = o_sc[0][:m]
        both = valid[:, 1:] & valid[:, :-1]
        sorted_desc = bool((sc[:, :-1] >= sc[:, 1:])[both].all().item())
        rows = X[loc.clamp(0, n - 1)].double()
        qd = Q[:m].double()
        exact = torch.einsum("qd,qkd->qk", qd, rows) / (qd.norm(dim=1, keepdim=True) * rows.norm(dim=2))
        max_err = float((sc.double() - exact).abs()[valid].max().item())
        props = {"queries": m, "ids_in_shard_range": in_range, "ids_unique": unique, "scores_sorted_desc": sorted_desc,
                 "max_abs_err_vs_f64_cosine": max_err, "full_lists": bool((cnt == k).all().item())}

    # ---- warmup + timed region ----------------------------------------------------------------------
    for i in range(n_warm):
        step(i)
    sync_all()
    ix.enable_timing(True)
    t0 = time.perf_counter()
    for i in range(n_launch):   # n_launch launches = n_launch * C client batches ("steps")
        step(i)
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist_on:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    steps_done = n_launch * C
    merged_qps = steps_done * Bc / elapsed   # answers over the global (world x n) corpus per second
    qps = merged_qps * world                  # units all ranks processed: every rank searched its shard for every query

    # per-launch walk-kernel figures: HIP events recorded by the library on the launch stream
    row_bytes = d + 4  # u8 code row + f32 norm per distance evaluation
    per = []
    for s in range(min(S, n_launch)):
        stt = ix.last_stats(streams[s].cuda_stream)
        per.append((stt.walk_ms, stt.evals * row_bytes + stt.adj_bytes, stt.evals, stt.expansions, stt.finalize_ms, stt.prep_ms, stt.reserved))
    ix.enable_timing(False)
    avg_ms = float(np.mean([p[0] for p in per]))
    avg_bytes = float(np.mean([p[1] for p in per]))
    # launches overlap on the chip: in-flight concurrency = sum of launch durations / wall time
    overlap = max(1.0, min(float(S), avg_ms * 1e-3 * n_launch / elapsed))
    achieved = avg_bytes * n_launch / elapsed / 1e9  # aggregate algorithmic GB/s of the walk kernel

    # single-stream (one batch at a time) rate
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for i in range(8):
        ix.batch_search_device(Q[:Bc].data_ptr(), Bc, k, o_ids[0].data_ptr(), o_sc[0].data_ptr(), o_cnt[0].data_ptr(), o_st[0].data_ptr(),
                               streams[0].cuda_stream)
    streams[0].synchronize()
    serial_qps = 8 * Bc / (time.perf_counter() - t1)

    # ---- optional ef_search sweep (same index, same launch shape): QPS and recall per ef ----------------------
    sweep = []
    main_mode = ca.VISITED_EXACT if args.visited == "exact" else ca.VISITED_REF
    for tok in [v for v in args.ef_sweep.split(",") if v]:   # "256" or "ref:256" / "exact:64" (other visited filter, same graph)
        mode_name, _, efs = tok.rpartition(":")
        ef2 = int(efs)
        mode2 = {"": main_mode, "ref": ca.VISITED_REF, "exact": ca.VISITED_EXACT}[mode_name]
        ix.set_visited_mode(mode2)
        rec2 = measure_recall(ef2)
        for i in range(n_warm):
            step(i)
        sync_all()
        t2 = time.perf_counter()
        for i in range(n_launch):
            step(i)
        sync_all()
        sweep.append({"ef_search": ef2, "visited": "exact" if mode2 == ca.VISITED_EXACT else "ref",
                      "qps": n_launch * C * Bc / (time.perf_counter() - t2), "recall_at_10": rec2})
    ix.set_ef_search(ef)
    ix.set_visited_mode(main_mode)

    # ---- CPU baseline: the oracle (C restatement of the Rust path) on this box's host cores ----------
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        cores = effective_cores()
        Xh = X.cpu().numpy()
        op = O.HNSWParams(dim=d, storage=O.STORAGE_U8, num_layers=9, ef_construction=args.ef_construction, ef_search=ef, seed=42,
                          range_lo=values_range[0], range_hi=values_range[1])
        op.visited_mode = O.VISITED_EXACT if args.visited == "exact" else O.VISITED_REF
        oix = O.OracleIndex(op).set_vectors(Xh)
        oix.import_graph(ix.download_graph(), ix.download_root())
        Qh = Q.cpu().numpy()
        t2 = time.perf_counter()
        probe = oix.search_batch(Qh[:max(64, cores)], k, threads=cores)
        rate = max(64, cores) / (time.perf_counter() - t2)
        nq = int(min(Qh.shape[0], max(256, rate * args.cpu_seconds)))
        # bounded sample: the distinct queries of the workload, repeated until about --cpu-seconds of wall time are spent
        t2 = time.perf_counter()
        oids, osc, ocnt = oix.search_batch(Qh[:nq], k, threads=cores)[:3]
        first = time.perf_counter() - t2
        reps = 1 + int(max(0, min(15, round(args.cpu_seconds / first) - 1)))
        for _ in range(reps - 1):
            oix.search_batch(Qh[:nq], k, threads=cores)
        cpu_s = time.perf_counter() - t2
        cpu = {"value": reps * nq / cpu_s, "unit": "queries/s", "cores": cores, "kind": "port",
               "sample": f"{reps} passes over {nq} queries of the same workload ({cpu_s:.1f} s wall on {cores} threads = usable "
                         f"cores: affinity/cgroup quota; host reports {os.cpu_count()} logical CPUs), C/AVX2 restatement of the Rust "
                         f"path (oracle/), one OpenMP thread per core like batch_search's rayon fan-out"}
        # free parity check on the same sample: GPU ids/scores vs oracle
        gi = np.zeros((nq, k), np.uint32)
        gs = np.zeros((nq, k), np.float32)
        for s0 in range(0, nq, B):
            m = min(B, nq - s0)
            ix.batch_search_device(Q[s0:s0 + m].data_ptr(), m, k, o_ids[0].data_ptr(), o_sc[0].data_ptr(), o_cnt[0].data_ptr(),
                                   o_st[0].data_ptr(), streams[0].cuda_stream)
            streams[0].synchronize()
            gi[s0:s0 + m] = o_ids[0][:m].cpu().numpy().view(np.uint32)
            gs[s0:s0 + m] = o_sc[0][:m].cpu().numpy()
        parity = {"queries": nq, "id_mismatch_queries": int((gi != oids).any(axis=1).sum()),
                  "score_bit_mismatches": int((gs.view(np.uint32) != osc.view(np.uint32)).sum())}

    # HBM traffic of the walk kernel: PMC FETCH_SIZE cannot be read from inside the process; it is taken from the
    # committed rocprofv3 --pmc pass of this same command (profiles/pmc_traffic.json, written by
    # scripts/pmc_traffic.py) when the workload / ef / launch shape match, else null.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            for ent in json.load(fh):
                if (ent["workload"], ent["ef_search"], ent["queries_per_launch"]) == (args.workload, ef, B) and not args.n:
                    traffic = ent["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass

    # empirical HBM ceilings on THIS part: streaming read and the walk's own access pattern — random
    # gathers of d-byte rows over a buffer the size of the code array — measured after the timed region
    empirical = None
    if rank == 0 and not args.no_hbm_probe:
        import ctypes as C_
        g = C_.c_double(0.0)
        empirical = {}
        for name, kind, nbytes, rb in (("stream_read", 0, 4 << 30, 0), ("stream_copy", 1, 2 << 30, 0),
                                       ("row_gather", 2, max(n * d, 1 << 20), d)):
            ca._lib.check(lib.cos_hbm_probe(local_rank, kind, nbytes, rb, 3, C_.byref(g)))
            empirical[name + "_GBps"] = g.value
        empirical["row_gather_row_bytes"] = d
        empirical["row_gather_buffer_bytes"] = n * d
        empirical["frac_of_row_gather"] = achieved / empirical["row_gather_GBps"]
        empirical["frac_of_stream_read"] = achieved / empirical["stream_read_GBps"]

    if rank == 0:
        out = {
            "metric": METRIC, "value": qps, "unit": "queries/s", "n_gpus": world, "steps": steps_done, "warmup": n_warm * C,
            "ms_per_step": elapsed / steps_done * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": args.workload + ": " + desc, "standard_size": not bool(args.n), "vectors_per_gpu": n, "dim": d,
                       "query_batch": Bc, "batches_per_launch": C, "launches_in_flight": S, "top_k": k, "ef_search": ef, "ef_policy": ("smallest ef with recall@10 >= %.2f" % args.recall_target) if args.ef == "auto" else "fixed", "M": 32, "M0": 64, "num_layers": 9, "ef_construction": args.ef_construction, "build_visited": args.build_visited,
                       "storage": f"u8 (quantization {args.quantization}, values_range {values_range})", "visited": "reference PerformantFixedSet (ID parity mode)" if args.visited == "ref" else "exact visited set (recall mode)",
                       "parallelism": f"id-range shards x{world}" + (" + RCCL all-gather top-k merge" if world > 1 else ""),
                       "corpus": f"Gaussian mixture, {n_centers} centres, sigma 0.8/sqrt(d), L2-normalised, seed 42"},
            "recall_at_10": recall, "recall_queries": nrq, "failed_queries": status_bad,
            "merged_qps": merged_qps, "global_corpus_vectors": n * world,
            "value_note": ("n_gpus == 1: value = queries/s over the whole corpus" if world == 1 else
                           "weak scaling: value = shard-level queries/s summed over ranks (every query is searched on every shard); "
                           "merged_qps = answers/s over the global corpus = value / n_gpus"),
            "single_batch_qps": serial_qps, "ef_selection": ef_table, "ef_sweep": sweep, "build_seconds": build_s, "setup_seconds": time.time() - t_setup,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "empirical": empirical, "kernel": "walk_kernel<ENG_U8, CH=1, R=%d, G64>" % (1 if ef <= 64 else (4 if ef <= 256 else 8)),
                         "per_launch": {"algorithmic_bytes": avg_bytes, "avg_ms": avg_ms, "in_flight": overlap,
                                        "evals": float(np.mean([p[2] for p in per])), "expansions": float(np.mean([p[3] for p in per])),
                                        "finalize_ms": float(np.mean([p[4] for p in per])), "prep_ms": float(np.mean([p[5] for p in per])),
                                        "adjacency_rounds": float(np.mean([p[6] for p in per]))},
                         "note": "one walk launch = query_batch x batches_per_launch queries; achieved = algorithmic bytes per launch x "
                                 "launches / timed wall time = bytes/avg_ms x in_flight (launches on different streams overlap)"},
            "flat_scan_ground_truth": flat, "result_properties": props, "cpu_baseline": cpu, "parity_vs_oracle": parity,
        }
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist_on:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
