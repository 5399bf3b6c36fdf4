#!/usr/bin/env python3
"""bench.py — QPS of the dense HNSW search path on MI355X (driver contract: see the task prompt).

A **step** is one pass of the hot path over one batch of synthetic input = ONE coalesced launch:
cos_search_batch_device() = quantize -> HNSW walk (every level, ef_search) -> dedup/top-5k -> exact f32 rerank -> top-k
over `--coalesce` client batches of `--batch` (256) queries fused server-side (dynamic batching, like the reference's own
RPS harness keeps 32 client threads x batches of 200 in flight, tests/rps-test.py: the walk kernel gives every query one
wavefront, so a 256-CU chip needs thousands of queries per launch), index and queries resident in HBM.  `--steps K
--warmup W` time exactly K launches after W untimed ones; `--inflight` launches overlap on separate HIP streams.  The
rate of one un-coalesced 256-query batch at a time is reported alongside (`single_batch_qps`).

Default workload = BASELINE.json configs[1]: 1M x 768 dense cosine HNSW, query batch 256, one GPU
(`--workload c4shard` = one 12.5M x 1024 shard of configs[3]).  N > 1: one process per GPU, every rank owns an
independent shard (ID-range partition, weak scaling: the corpus grows with N), queries are replicated, and each step ends
with the all-gather of the per-shard top-k (RCCL over xGMI) + the S-way merge kernel.  `value` is the rate at which
MERGED answers over the global corpus come out (queries/s); the shard-level work all ranks did is reported next to it as
`shard_searches_per_s` (= value x N).

Synthetic data (no network): a seeded Gaussian-mixture corpus, L2-normalised, generated on the device; queries are fresh
draws from the same mixture.  ef_search is SELECTED on one query set and recall@10 is REPORTED on a disjoint one, both
against exact brute-force cosine on the same corpus, outside the timed region.
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


def recall_stats(hits_per_query, k):
    """mean recall@k, its standard error and the one-sided 95 % lower confidence bound over the query sample"""
    r = np.asarray(hits_per_query, dtype=np.float64) / k
    mean = float(r.mean())
    se = float(r.std(ddof=1) / np.sqrt(r.size)) if r.size > 1 else 0.0
    return mean, se, mean - 1.645 * se


def select_ef(candidates, measure, target):
    """smallest ef whose recall on the SELECTION set clears the target with 95 % confidence (lower bound >= target);
    `measure(ef) -> (mean, se, lower)`.  Falls back to the largest candidate.  Returns (ef, table)."""
    table, ef = [], candidates[-1]
    for cand in candidates:
        mean, se, lower = measure(cand)
        table.append({"ef_search": cand, "recall_at_10": mean, "stderr": se, "lower95": lower})
        if lower >= target:
            ef = cand
            break
    return ef, table


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
    ap.add_argument("--steps", type=int, default=24, help="timed steps; one step = one coalesced launch of --coalesce x --batch queries")
    ap.add_argument("--warmup", type=int, default=4, help="untimed warm-up steps (launches)")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--n", type=int, default=0, help="override vectors per GPU (marks the run as non-standard)")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--coalesce", type=int, default=128, help="client batches fused per launch (dynamic batching); 128 x 256 = 32768 queries "
                    "refill the chip's 7168 wave slots several times over, so one launch runs the walk kernel at 0.9 of the HBM roof "
                    "(8192-query launches: 0.71 — every wave is resident at once and the chip drains as they finish)")
    ap.add_argument("--inflight", type=int, default=2, help="launches kept in flight (HIP streams)")
    ap.add_argument("--ef", default="auto", help="ef_search: an integer, or 'auto' = smallest of 32,48,64,96,128,192,256,384,512 whose recall@10 "
                    "on the SELECTION query set clears --recall-target with 95 %% confidence (the metric is QPS AT recall@10 >= 0.95); "
                    "config.toml default is 256")
    ap.add_argument("--recall-target", type=float, default=0.95)
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--visited", default="ref", choices=["ref", "exact"],
                    help="search-time visited filter: ref = PerformantFixedSet replica (ID parity with the reference), "
                         "exact = exact visited set (recall mode, not ID-identical)")
    ap.add_argument("--ef-construction", type=int, default=128, help="config.toml default 128")
    ap.add_argument("--build-visited", default="ref", choices=["ref", "exact"], help="visited filter used by the builder's walks")
    ap.add_argument("--quantization", default="auto", choices=["auto", "range11"], help="auto = sampled values_range; range11 = (-1,1)")
    ap.add_argument("--ef-sweep", default="256,exact:32,exact:64",
                    help="comma list of extra settings timed after the main run on the same graph: '256' = ef_search 256 (config.toml "
                         "default) with the main visited filter; 'exact:32' / 'ref:128' = that ef with the named visited filter")
    ap.add_argument("--build-batch", type=int, default=4096)
    ap.add_argument("--recall-queries", type=int, default=8192, help="size of EACH of the two disjoint recall query sets (selection / report)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-probe", action="store_true", help="skip the empirical HBM ceiling probes (cos_hbm_probe)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--exchange", default="auto", choices=["auto", "shardset", "torch"],
                    help="N > 1 exchange step: shardset = cos_shardset_* (RCCL all-gather + merge inside the C ABI); torch = "
                         "torch.distributed all_gather + cos_merge_topk_packed_device; auto = shardset, falling back loudly to torch")
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
    dist = None
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
    Bc, k, C = args.batch, args.top_k, max(1, args.coalesce)   # Bc = client batch; C client batches per launch
    B = Bc * C                                                  # queries per launch = one step
    n_launch, n_warm = max(1, args.steps), max(0, args.warmup)
    ef = 256 if args.ef == "auto" else int(args.ef)
    t_setup = time.time()

    # ---- synthetic shard + queries (resident in HBM) -------------------------------------------
    gc = torch.Generator(device=dev)
    gc.manual_seed(4242)
    n_centers = max(64, (n * world) // 1000)
    centers = torch.randn(n_centers, d, generator=gc, device=dev)
    centers /= centers.norm(dim=1, keepdim=True)
    X = mixture(n, d, 42 + 1000 * rank, dev, centers)            # this rank's shard: global ids [rank*n, (rank+1)*n)
    n_qsets = max(args.inflight, 2)
    Q = mixture(B * n_qsets, d, 43, dev, centers)                 # timed queries; identical on every rank
    nrq = args.recall_queries
    Q_sel = mixture(nrq, d, 44, dev, centers)                     # ef selection set
    Q_rep = mixture(nrq, d, 45, dev, centers)                     # disjoint hold-out: the recall that is REPORTED
    torch.cuda.synchronize()

    # ---- index: reference defaults (config.toml:20-24,32); "auto" quantization = u8 + values_range sampled from the
    # first sample_threshold embeddings (indexes/hnsw/mod.rs:202-351; tests/rps-test.py:73 uses the same mode) ---
    sample_threshold = 1000
    values_range = ca.sample_values_range(X[:sample_threshold].cpu().numpy(), 1.0) if args.quantization == "auto" else (-1.0, 1.0)
    if dist_on and world > 1:  # every shard must quantize with the same range: take rank 0's sample
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
    S = max(1, args.inflight)
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    # the shard's result lives in ONE packed record [ids | scores | counts] so the sharded path exchanges it
    # with a single all-gather per launch (sharding.py / cos_shardset_*)
    o_pack = torch.zeros(S, packed_words(B, k), dtype=torch.int32, device=dev)
    o_views = [packed_views(o_pack[s], B, k) for s in range(S)]
    o_ids = [v[0] for v in o_views]
    o_sc = [v[1] for v in o_views]
    o_cnt = [v[2] for v in o_views]
    o_st = torch.zeros(S, B, dtype=torch.int32, device=dev)
    exchange_kind = None
    shardset = None
    if dist_on:
        g_pack = torch.zeros(S, world, packed_words(B, k), dtype=torch.int32, device=dev)
        m_ids = torch.zeros(S, B, k, dtype=torch.int32, device=dev)
        m_sc = torch.zeros(S, B, k, dtype=torch.float32, device=dev)
        m_cnt = torch.zeros(S, B, dtype=torch.int32, device=dev)
        exchange_kind = "torch.distributed all_gather_into_tensor (RCCL) + cos_merge_topk_packed_device"
        if args.exchange in ("auto", "shardset"):
            try:
                from cosdata_amd.shardset import ProcessShardSet
                shardset = ProcessShardSet(ix, rank, world, local_rank, dist)
                exchange_kind = "cos_shardset_exchange_device: ncclAllGather (RCCL) + merge kernel inside the C ABI"
            except Exception as exc:  # noqa: BLE001 — reported in the JSON line, never silent
                if args.exchange == "shardset":
                    raise
                exchange_kind += f" [shardset unavailable: {type(exc).__name__}: {exc}]"
    lib = ca._lib.lib()

    def step(i):
        s = i % S
        st = streams[s]
        q = Q[(i % n_qsets) * B:(i % n_qsets + 1) * B]
        ix.batch_search_device(q.data_ptr(), B, k, o_ids[s].data_ptr(), o_sc[s].data_ptr(), o_cnt[s].data_ptr(), o_st[s].data_ptr(),
                               st.cuda_stream)
        if dist_on:  # per-shard top-k -> RCCL all-gather over xGMI -> S-way merge
            if shardset is not None:
                shardset.exchange_device(o_pack[s].data_ptr(), B, k, g_pack[s].data_ptr(), m_ids[s].data_ptr(), m_sc[s].data_ptr(),
                                         m_cnt[s].data_ptr(), st.cuda_stream)
            else:
                with torch.cuda.stream(st):
                    allgather_packed(o_pack[s], g_pack[s])
                merge_topk_packed_device(g_pack[s], B, k, m_ids[s], m_sc[s], m_cnt[s], local_rank, st.cuda_stream)

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # ---- recall@10 vs exact brute force (outside the timed region) ---------------------------------
    # ground truth: the engine's own exhaustive scan (f32-MFMA GEMM -> top-64 -> reference-order re-score),
    # cross-checked against an independent torch matmul + topk on a slice
    def ground_truth(Qset):
        qh = Qset.cpu().numpy()
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        ids, _ = ix.bruteforce_topk(qh, k)
        return torch.from_numpy(ids.astype(np.int64) - rank * n).to(dev), time.perf_counter() - t

    def merge_global(Qset, local_ids_global):
        """global answer = merge of the per-shard lists by exact cosine (recall bookkeeping only)"""
        loc = (local_ids_global - rank * n).clamp_(0, n - 1)
        sims = torch.einsum("qd,qkd->qk", Qset, X[loc])   # exact cosine of the k returned rows only (unit-norm corpus)
        return global_topk_by_score(local_ids_global, sims, k)

    gt_sel_local, bf_seconds = ground_truth(Q_sel)
    gt_rep_local, _ = ground_truth(Q_rep)
    nx = min(nrq, 1024)
    gt_torch = bruteforce_top10(X, Q_sel[:nx], k)
    gt_agree = float((gt_sel_local[:nx].unsqueeze(2) == gt_torch.unsqueeze(1)).any(dim=2).float().mean().item())
    flat = {"queries": nrq, "seconds": bf_seconds, "tflops_end_to_end": 2.0 * nrq * n * d / bf_seconds / 1e12,
            "agreement_with_torch_topk": gt_agree, "torch_checked_queries": nx,
            "note": "cos_bruteforce_topk incl. H2D/D2H, selection and exact re-score; kernel-level MFMA rate: profiles/"}
    gt_sel = merge_global(Q_sel, gt_sel_local + rank * n) if dist_on else gt_sel_local
    gt_rep = merge_global(Q_rep, gt_rep_local + rank * n) if dist_on else gt_rep_local
    ann = torch.zeros(nrq, k, dtype=torch.int64, device=dev)

    def measure_recall(ef_value, Qset, gt):
        """(mean, stderr, lower 95 % bound) of recall@k over Qset at ef_value; identical on every rank"""
        ix.set_ef_search(ef_value)
        for s0 in range(0, nrq, B):
            m = min(B, nrq - s0)
            ix.batch_search_device(Qset[s0:s0 + m].data_ptr(), m, k, o_ids[0].data_ptr(), o_sc[0].data_ptr(), o_cnt[0].data_ptr(),
                                   o_st[0].data_ptr(), streams[0].cuda_stream)
            streams[0].synchronize()
            ann[s0:s0 + m] = o_ids[0][:m].to(torch.int64) & 0xFFFFFFFF
        ann_g = merge_global(Qset, ann) if dist_on else ann
        hits = (ann_g.unsqueeze(2) == gt.unsqueeze(1)).any(dim=2).float().sum(dim=1)
        if dist_on:
            dist.broadcast(hits, 0)  # every rank must take the same decision
        return recall_stats(hits.cpu().numpy(), k)

    ef_table = []
    if args.ef == "auto":
        ef, ef_table = select_ef([32, 48, 64, 96, 128, 192, 256, 384, 512], lambda e: measure_recall(e, Q_sel, gt_sel), args.recall_target)
    recall, recall_se, recall_lo = measure_recall(ef, Q_rep, gt_rep)     # the reported figure: hold-out set
    status_bad = int((o_st != 0).sum().item())

    # ---- size-independent properties of the returned lists, checked at the full workload size (no oracle needed): ids in
    # this shard's range and unique per query, scores non-increasing, every score = the exact f32 cosine of the row it names
    m = min(B, nrq, 2048)
    ix.batch_search_device(Q_rep[:m].data_ptr(), m, k, o_ids[0].data_ptr(), o_sc[0].data_ptr(), o_cnt[0].data_ptr(), o_st[0].data_ptr(),
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
    qd = Q_rep[:m].double()
    exact = torch.einsum("qd,qkd->qk", qd, rows) / (qd.norm(dim=1, keepdim=True) * rows.norm(dim=2))
    max_err = float((sc.double() - exact).abs()[valid].max().item())
    props = {"queries": m, "ids_in_shard_range": in_range, "ids_unique": unique, "scores_sorted_desc": sorted_desc,
             "max_abs_err_vs_f64_cosine": max_err, "full_lists": bool((cnt == k).all().item())}
    del rows, qd, exact

    # ---- warmup + timed region: EXACTLY n_launch steps (launches) ---------------------------------------
    def timed_run():
        for i in range(n_warm):
            step(i)
        sync_all()
        ix.enable_timing(True)
        t = time.perf_counter()
        for i in range(n_launch):
            step(i)
        sync_all()
        el = time.perf_counter() - t
        if dist_on:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        # per-launch walk-kernel figures: HIP events recorded by the library on each launch stream (a ring of the last 128
        # launches per stream) + the counters of the last launch of every stream
        row_bytes = d + 4  # u8 code row + f32 norm per distance evaluation
        walk_sum = prep_sum = fin_sum = 0.0
        walk_min, walk_max, nl = 1e30, 0.0, 0
        cnts = []
        for s in range(min(S, n_launch)):
            ts = ix.timing_summary(streams[s].cuda_stream)
            walk_sum += ts.walk_ms_sum; prep_sum += ts.prep_ms_sum; fin_sum += ts.finalize_ms_sum; nl += ts.launches
            walk_min, walk_max = min(walk_min, ts.walk_ms_min), max(walk_max, ts.walk_ms_max)
            stt = ix.last_stats(streams[s].cuda_stream)
            cnts.append((stt.evals * row_bytes + stt.adj_bytes, stt.evals, stt.expansions, stt.reserved))
        ix.enable_timing(False)
        return {"elapsed": el, "walk_ms": walk_sum / nl, "prep_ms": prep_sum / nl, "finalize_ms": fin_sum / nl, "walk_ms_min": walk_min,
                "walk_ms_max": walk_max, "timed_launches_sampled": nl, "bytes": float(np.mean([c[0] for c in cnts])),
                "evals": float(np.mean([c[1] for c in cnts])), "expansions": float(np.mean([c[2] for c in cnts])),
                "rounds": float(np.mean([c[3] for c in cnts]))}

    tr = timed_run()
    elapsed = tr["elapsed"]
    merged_qps = n_launch * B / elapsed        # answers over the global (world x n) corpus per second
    shard_searches = merged_qps * world        # shard-level searches all ranks completed per second
    avg_ms, avg_bytes = tr["walk_ms"], tr["bytes"]
    kernel_gbps = avg_bytes / (avg_ms * 1e-3) / 1e9              # ONE walk launch: algorithmic bytes / its HIP-event duration
    aggregate_gbps = avg_bytes * n_launch / elapsed / 1e9         # all launches / wall time (launches on different streams overlap)
    overlap = max(1.0, min(float(S), avg_ms * 1e-3 * n_launch / elapsed))

    # single-stream (one un-coalesced client batch at a time) rate: launches this small take the latency variant of the walk
    # (cos_index_set_latency_mode); the throughput kernel is timed on the same batch and must return the same bits
    def serial_rate():
        for _ in range(2):
            ix.batch_search_device(Q[:Bc].data_ptr(), Bc, k, o_ids[0].data_ptr(), o_sc[0].data_ptr(), o_cnt[0].data_ptr(), o_st[0].data_ptr(),
                                   streams[0].cuda_stream)
        streams[0].synchronize()
        t1 = time.perf_counter()
        for _ in range(8):
            ix.batch_search_device(Q[:Bc].data_ptr(), Bc, k, o_ids[0].data_ptr(), o_sc[0].data_ptr(), o_cnt[0].data_ptr(), o_st[0].data_ptr(),
                                   streams[0].cuda_stream)
        streams[0].synchronize()
        rate = 8 * Bc / (time.perf_counter() - t1)
        return rate, (o_ids[0][:Bc].clone(), o_sc[0][:Bc].clone().view(torch.int32), o_cnt[0][:Bc].clone(), o_st[0][:Bc].clone())
    torch.cuda.synchronize(dev)
    serial_qps, serial_out = serial_rate()
    ix.set_latency_mode(0)
    serial_qps_throughput_kernel, serial_out_tk = serial_rate()
    ix.set_latency_mode(ca.HNSWIndex.LATENCY_MODE_DEFAULT_MAX_B)
    serial_identical = all(bool(torch.equal(a, b_)) for a, b_ in zip(serial_out, serial_out_tk))

    # ---- optional ef_search sweep (same index, same launch shape): QPS and hold-out recall per setting ----------------------
    sweep = []
    main_mode = ca.VISITED_EXACT if args.visited == "exact" else ca.VISITED_REF
    for tok in [v for v in args.ef_sweep.split(",") if v]:   # "256" or "ref:256" / "exact:64" (other visited filter, same graph)
        mode_name, _, efs = tok.rpartition(":")
        ef2 = int(efs)
        mode2 = {"": main_mode, "ref": ca.VISITED_REF, "exact": ca.VISITED_EXACT}[mode_name]
        ix.set_visited_mode(mode2)
        rec2 = measure_recall(ef2, Q_rep, gt_rep)
        tr2 = timed_run()
        sweep.append({"ef_search": ef2, "visited": "exact" if mode2 == ca.VISITED_EXACT else "ref", "qps": n_launch * B / tr2["elapsed"],
                      "recall_at_10": rec2[0], "recall_stderr": rec2[1], "walk_ms": tr2["walk_ms"],
                      "kernel_GBps": tr2["bytes"] / (tr2["walk_ms"] * 1e-3) / 1e9})
    ix.set_ef_search(ef)
    ix.set_visited_mode(main_mode)

    # ---- the boundary as the Rust host would call it: host buffers in, host buffers out (cos_search_batch: H2D of the queries, the
    # same kernels, D2H of ids / scores / counts, one synchronisation per call).  PCIe-inclusive, reported next to `value`, never it.
    host_api = None
    if rank == 0 and world == 1:
        qh = Q[:B].cpu().numpy()
        ix.batch_search(qh, k)                                  # warm-up: the calling thread's stream + workspace
        t1 = time.perf_counter()
        reps_h = 4
        for _ in range(reps_h):
            h_ids, h_sc, h_cnt = ix.batch_search(qh, k)
        el_h = (time.perf_counter() - t1) / reps_h
        host_api = {"queries_per_call": B, "ms_per_call": el_h * 1e3, "qps": B / el_h,
                    "h2d_bytes_per_call": int(B) * d * 4, "d2h_bytes_per_call": int(B) * (k * 8 + 4),
                    "note": "cos_search_batch on pageable host memory, one call at a time (no overlap of copies with the previous call's kernels)"}
        del qh

    # ---- CPU baseline + parity: the oracle (C restatement of the Rust path) on this box's host cores, same graph ----------
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        cores = effective_cores()
        op = O.HNSWParams(dim=d, storage=O.STORAGE_U8, num_layers=9, ef_construction=args.ef_construction, ef_search=ef, seed=42,
                          range_lo=values_range[0], range_hi=values_range[1])
        op.visited_mode = O.VISITED_EXACT if args.visited == "exact" else O.VISITED_REF
        full_raw = (n * d * 4) <= (8 << 30)      # up to 8 GB the oracle gets the raw table; beyond, rows are streamed through
        if full_raw:
            oix = O.OracleIndex(op).set_vectors(X.cpu().numpy())
        else:   # the oracle quantizes the corpus itself, chunk by chunk; the exact rerank later gets the raw rows it needs
            oix = O.OracleIndex(op).alloc_vectors(n)
            for s0 in range(0, n, 1 << 18):
                oix.quantize_rows(s0, X[s0:s0 + (1 << 18)].cpu().numpy())
        oix.import_graph(ix.download_graph(), ix.download_root())
        Qh = Q.cpu().numpy()
        t2 = time.perf_counter()
        pm = max(64, cores)
        if full_raw:
            oix.search_batch(Qh[:pm], k, threads=cores)
        else:
            oix.candidates_batch(Qh[:pm], k, threads=cores)
        rate = pm / (time.perf_counter() - t2)
        nq = int(min(Qh.shape[0], max(256, rate * args.cpu_seconds)))
        if not full_raw:   # raw rows of exactly the candidates finalize_ann_results will read for these nq queries
            cand, _ = oix.candidates_batch(Qh[:nq], k, threads=cores)
            u = np.unique(cand[cand != 0xFFFFFFFF])
            oix.set_raw_subset(u, X[torch.from_numpy(u.astype(np.int64)).to(dev)].cpu().numpy())
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
                         f"path (oracle/), one OpenMP thread per core like batch_search's rayon fan-out"
                         + ("" if full_raw else "; corpus quantized by the oracle in streamed chunks, rerank rows fetched per candidate")}
        # parity check on the same sample: GPU ids/scores vs oracle, bit for bit
        gi = np.zeros((nq, k), np.uint32)
        gs = np.zeros((nq, k), np.float32)
        gc_ = np.zeros(nq, np.uint32)
        for s0 in range(0, nq, B):
            mm = min(B, nq - s0)
            ix.batch_search_device(Q[s0:s0 + mm].data_ptr(), mm, k, o_ids[0].data_ptr(), o_sc[0].data_ptr(), o_cnt[0].data_ptr(),
                                   o_st[0].data_ptr(), streams[0].cuda_stream)
            streams[0].synchronize()
            gi[s0:s0 + mm] = o_ids[0][:mm].cpu().numpy().view(np.uint32)
            gs[s0:s0 + mm] = o_sc[0][:mm].cpu().numpy()
            gc_[s0:s0 + mm] = o_cnt[0][:mm].cpu().numpy().view(np.uint32)
        parity = {"queries": nq, "id_mismatch_queries": int((gi != oids).any(axis=1).sum()),
                  "score_bit_mismatches": int((gs.view(np.uint32) != osc.view(np.uint32)).sum()),
                  "count_mismatches": int((gc_ != ocnt).sum())}
        del oix

    # HBM traffic of the walk kernel: PMC FETCH_SIZE cannot be read from inside the process; it is taken from the
    # committed rocprofv3 --pmc pass of this same command (profiles/pmc_traffic.json, written by
    # scripts/pmc_traffic.py) when the workload / ef / launch shape match, else null.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            for ent in json.load(fh):
                if (ent["workload"], ent["ef_search"], ent["queries_per_launch"]) == (args.workload, ef, B) and not args.n \
                        and ent.get("visited", "ref") == args.visited:
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
        empirical["kernel_frac_of_row_gather"] = kernel_gbps / empirical["row_gather_GBps"]
        empirical["kernel_frac_of_stream_read"] = kernel_gbps / empirical["stream_read_GBps"]

    if rank == 0:
        out = {
            "metric": METRIC, "value": merged_qps, "unit": "queries/s", "n_gpus": world, "steps": n_launch, "warmup": n_warm,
            "ms_per_step": elapsed / n_launch * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": args.workload + ": " + desc, "standard_size": not bool(args.n), "vectors_per_gpu": n, "dim": d,
                       "step": f"one coalesced launch = {C} client batches x {Bc} queries = {B} queries through quantize -> walk -> rerank -> top-k",
                       "query_batch": Bc, "batches_per_launch": C, "queries_per_step": B, "launches_in_flight": S, "top_k": k, "ef_search": ef,
                       "ef_policy": ("smallest ef whose recall@10 on the selection query set is >= %.2f with 95%% confidence; recall_at_10 is "
                                     "measured on a disjoint hold-out set" % args.recall_target) if args.ef == "auto" else "fixed",
                       "M": 32, "M0": 64, "num_layers": 9, "ef_construction": args.ef_construction, "build_visited": args.build_visited,
                       "storage": f"u8 (quantization {args.quantization}, values_range {values_range})",
                       "visited": "reference PerformantFixedSet (ID parity mode)" if args.visited == "ref" else "exact visited set (recall mode)",
                       "parallelism": f"id-range shards x{world}" + (" + RCCL all-gather top-k merge" if world > 1 else ""),
                       "exchange": exchange_kind,
                       "corpus": f"Gaussian mixture, {n_centers} centres, sigma 0.8/sqrt(d), L2-normalised, seed 42"},
            "recall_at_10": recall, "recall_stderr": recall_se, "recall_lower95": recall_lo, "recall_queries": nrq,
            "recall_sets": "ef selected on query seed 44, recall reported on query seed 45 (disjoint draws of the same mixture)",
            "failed_queries": status_bad,
            "merged_qps": merged_qps, "shard_searches_per_s": shard_searches, "global_corpus_vectors": n * world,
            "value_note": ("n_gpus == 1: value = queries/s over the whole corpus" if world == 1 else
                           "value = merged answers/s over the GLOBAL corpus (n_gpus x vectors_per_gpu; every query is searched on every "
                           "shard, then all-gathered and merged); shard_searches_per_s = value x n_gpus is the shard-level work"),
            "single_batch_qps": serial_qps, "single_batch_qps_throughput_kernel": serial_qps_throughput_kernel,
            "single_batch_latency_walk_identical_to_throughput_walk": serial_identical,
            "ef_selection": ef_table, "ef_sweep": sweep, "build_seconds": build_s, "setup_seconds": time.time() - t_setup,
            "roofline": {"bound": "hbm", "achieved": kernel_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": kernel_gbps / HBM_PEAK_GBPS,
                         "traffic": traffic, "empirical": empirical,
                         "kernel": "walk_kernel<ENG_U8, CH=1, R=%d, G64, %s>" % (1 if ef <= 64 else (4 if ef <= 256 else 8), args.visited),
                         "aggregate": {"achieved": aggregate_gbps, "frac": aggregate_gbps / HBM_PEAK_GBPS, "in_flight": overlap,
                                       "note": "algorithmic bytes of ALL timed launches / timed wall time; exceeds the kernel-level figure "
                                               "only through launches overlapping on different streams"},
                         "per_launch": {"algorithmic_bytes": avg_bytes, "avg_ms": avg_ms, "min_ms": tr["walk_ms_min"], "max_ms": tr["walk_ms_max"],
                                        "launches_sampled": tr["timed_launches_sampled"], "evals": tr["evals"], "expansions": tr["expansions"],
                                        "finalize_ms": tr["finalize_ms"], "prep_ms": tr["prep_ms"], "adjacency_rounds": tr["rounds"]},
                         "note": "achieved = algorithmic bytes of one walk launch (evals x (dim+4) + expansions x M x 4, counted by the kernel) / "
                                 "that launch's average HIP-event duration on its own stream over the timed region"},
            "flat_scan_ground_truth": flat, "result_properties": props, "cpu_baseline": cpu, "parity_vs_oracle": parity,
            "host_api_pcie_inclusive": host_api,
        }
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if shardset is not None:
        shardset.close()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
