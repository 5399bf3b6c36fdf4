#!/usr/bin/env python3
"""bench.py — QPS of the dense HNSW search path on MI355X (driver contract: see the task prompt).

A **step** is one pass of the hot path over one batch of synthetic input = ONE coalesced launch:
cos_search_batch_device() = quantize -> HNSW walk (every level, ef_search) -> dedup/top-5k -> exact f32 rerank -> top-k
over `--coalesce` client batches of `--batch` (256) queries fused server-side (dynamic batching, like the reference's own
RPS harness keeps 32 client threads x batches of 200 in flight, tests/rps-test.py: the walk kernel gives every query one
wavefront, so a 256-CU chip needs thousands of queries per launch), index and queries resident in HBM.  `--steps K
--warmup W` time exactly K launches after W untimed ones; `--inflight` launches overlap on separate HIP streams.  The
rate of one un-coalesced 256-query batch at a time is reported alongside (`single_batch_qps`).

`value` = BASELINE.json configs[3], the configuration the metric is quoted on ("1024-dim dense cosine, 1/2/4/8 MI355X"): at `--gpus N`
the corpus is N id-range shards of 12.5M x 1024 (one per GPU; N = 8 is the 100M x 1024 of configs[3], N = 1 is its per-GPU slice and
the point SCALE's curve starts from), walked with the reference's own visited filter and the hyper-parameters that meet the recall
target in the Rust path's semantics (level_0_neighbors_count 256, neighbors_count 64: `c4shard` below).  Round 6 moved `value` here
from c2 (configs[1], 768-dim), which now rides along under `configs` with every block it had.  The SAME JSON line carries one
record per remaining BASELINE config under `configs` (each with its own roofline / cpu_baseline / parity_vs_oracle block):
`c4_8shards_one_device` (the same corpus as eight id-range shards behind cos_shardset_search_batch on one device), `c2` (1M x 768,
query batch 256: single-batch rate, host API and concurrent 256-query callers included), `c2_uniform` (the uniform(-1,1) corpus of
tests/test.py:88), `c5` (hybrid dense + BM25 + RRF over 1M documents), `c3` (10M x 768 quaternary codes, exhaustive scan + HNSW
walk with the quaternary distance).  `--configs` selects them (default: all at N = 1, none at N > 1); other c4shard variants
(`c4shard_ref` = default hyper-parameters, `c4shard_exact`, other M0 / M) on request.

N > 1: one process per GPU.  `--gpus N` without WORLD_SIZE in the environment starts the N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`); under an external
torchrun the flag must agree with WORLD_SIZE.  Every rank owns an independent shard (ID-range partition, weak scaling:
the corpus grows with N), queries are replicated, and each step ends with the all-gather of the per-shard top-k (RCCL
over xGMI) + the S-way merge kernel.  `value` is the rate at which MERGED answers over the global corpus come out
(queries/s); the shard-level work all ranks did is reported next to it as `shard_searches_per_s` (= value x N).
`--single-process` runs the reference's own deployment instead (one host process, N devices,
cos_shardset_search_batch: indexes/mod.rs:260-272).

Synthetic data (no network): a seeded Gaussian-mixture corpus, L2-normalised, generated on the device; queries are fresh
draws from the same mixture.  ef_search is SELECTED on one query set and recall@10 is REPORTED on a disjoint one, both
against exact brute-force cosine on the same corpus, outside the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL / hipIpcGetMemHandle fail in legacy mode); the variable is
# normally exported already — keep it if the launcher dropped it.  Must be set before the HIP runtime initialises.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "QPS at recall@10≥0.95, 1024-dim dense cosine, 1/2/4/8 MI355X"
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
I8_PEAK_TOPS = 5000.0    # dense i8 MFMA peak (= the dense fp8 figure); the guide's measured kernel rate is >= 3944 TOPS

WORKLOADS = {
    # name: (n per GPU, dim, corpus, default ef_construction, description)
    "c2": (1_000_000, 768, "mixture", 128, "BASELINE configs[1]: 1M x 768 dense cosine HNSW, query-batch 256, u8 auto-quantized storage"),
    "c2_uniform": (1_000_000, 768, "uniform", 128, "BASELINE configs[1] on the uniform(-1,1) corpus of tests/test.py:88 (adversarial for ANN)"),
    "c2_sigma01": (1_000_000, 768, "mixture_sigma01", 128, "BASELINE configs[1] on SURVEY 8(d)'s clustered set: 1 000 Gaussian centres, sigma 0.1 per component, L2-normalised"),
    "c4shard": (12_500_000, 1024, "mixture", 256, "one 12.5M x 1024 shard of BASELINE configs[3] (100M x 1024 over 8 GPUs)"),
    "smoke": (50_000, 768, "mixture", 128, "reduced-size plumbing run (NOT a benchmark number)"),
}
# c4shard_ref = the reference's default hyper-parameters (M0 64, M 32: its 4096- / 2048-bit visited filters saturate at recall 0.85 on
# this shard); c4shard_ref_m0_256_m_64 = the same reference semantics with level_0_neighbors_count 256 and neighbors_count 64 (both
# user hyper-parameters, indexes/hnsw/types.rs:10-17; the filter is PerformantFixedSet::new(that count), vector_store.rs:266-270):
# the configuration that meets the recall target on the metric's own shard in the Rust path's own semantics.
# (c4shard_ref, the default-hyper-parameter record — recall saturating at 0.846, profiles/r04_final_bench_default_full_record.json —, moved to the
# optional list in round 5: its minute of the default run went to c4_8shards_one_device.)
# Round 6: `value` IS the target-meeting shard (main workload c4shard, M0 256 / M 64); c2 is a record under `configs`.
ALL_CONFIGS = ["c4_8shards_one_device", "c2", "c2_sigma01", "c2_uniform", "c5", "c3"]
MAIN_WORKLOAD = "c4shard"
EF_CANDIDATES = [32, 48, 64, 80, 96, 112, 128, 160, 192, 256, 384, 512]   # ef_search is a user parameter (indexes/hnsw/types.rs:14): the grid it is selected from
MAIN_M0, MAIN_M = 256, 64
OPTIONAL_CONFIGS = ["c4shard_ref", "c4shard_exact", "c4shard_ref_m0_128", "c4shard_ref_m0_256", "c4shard_ref_m0_256_m_64", "c4shard_ref_m0_256_m_128"]   # --configs only


def value_workload(world):
    """what `value` is measured on at `--gpus world` (the same string in the real line's config.workload and in the launcher self-test)"""
    n, d = WORKLOADS[MAIN_WORKLOAD][0], WORKLOADS[MAIN_WORKLOAD][1]
    return (f"BASELINE configs[3]: {world} id-range shard(s) of {n} x {d} dense cosine (100M x {d} over 8 GPUs; one shard per GPU), "
            f"level_0_neighbors_count {MAIN_M0}, neighbors_count {MAIN_M}, reference visited filter"
            + (", RCCL all-gather of the per-shard top-k + merge inside the timed step" if world > 1 else ""))


# ------------------------------------------------------------------------------------------------------------------
# host-side helpers (no GPU needed; tests/test_bench_host_logic.py)
# ------------------------------------------------------------------------------------------------------------------
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


def resolve_configs(spec, world, workload):
    """`--configs` -> ordered list of extra config records to run.  auto: every other BASELINE config at N = 1 (when the main
    workload is the standard one), none at N > 1 (the main workload IS configs[3] there) or for non-standard main workloads."""
    if spec in ("none", ""):
        return []
    if spec == "auto":
        if workload != MAIN_WORKLOAD:
            return []
        return list(ALL_CONFIGS) if world == 1 else []
    names = list(ALL_CONFIGS) if spec == "all" else [v for v in spec.split(",") if v]
    bad = [v for v in names if v not in ALL_CONFIGS + OPTIONAL_CONFIGS]
    if bad:
        raise SystemExit(f"bench.py: unknown --configs entries {bad}; known: {ALL_CONFIGS}")
    if world > 1:
        dropped = [v for v in names if not v.startswith("c4shard")]
        if dropped:
            print(f"bench.py: configs {dropped} are single-GPU configurations, skipped at N = {world}", file=sys.stderr)
        names = [v for v in names if v.startswith("c4shard")]
    return names


def value_at_batch_256(host):
    """the concurrent-256-query-callers leg of host_api_pcie_inclusive as a top-level block: per caller count the best max_queries"""
    small = (host or {}).get("concurrent_256_query_callers") or []
    best = {}
    for e in small:
        if e.get("failed_calls", 0) == 0 and e.get("identical_to_uncoalesced_call") and e["qps"] > best.get(e["callers"], {"qps": 0.0})["qps"]:
            best[e["callers"]] = e
    if not best:
        return None
    return {"unit": "queries/s", "pcie_inclusive": True, "by_callers": {str(c): {"qps": best[c]["qps"], "coalescing_max_queries": best[c]["coalescing_max_queries"]}
                                                                           for c in sorted(best)},
            "min_over_caller_counts": min(v["qps"] for v in best.values())}


def committed_kernel_traffic(key):
    """fabric-side bytes PER DISPATCH of a config's dominant kernel from the committed PMC pass of the config's own script
    (scripts/pmc_traffic_kernel.py -> profiles/pmc_traffic.json), or None"""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            for ent in json.load(fh):
                if ent.get("workload") == key and "bytes_per_dispatch" in ent and ent.get("dispatches", 0) > 0:
                    return float(ent["bytes_per_dispatch"])
    except (OSError, ValueError, KeyError):
        pass
    return None


def ca_mod():
    import cosdata_amd
    return cosdata_amd


def native_callers_harness():
    """scripts/libcallers_bench.so (built from callers_bench.cpp with g++ on first use): N native threads calling cos_search_batch"""
    import ctypes as C_
    import subprocess
    src = os.path.join(ROOT, "scripts", "callers_bench.cpp")
    so = os.path.join(ROOT, "scripts", "libcallers_bench.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-o", so, src])
    h = C_.CDLL(so)
    vp, u32 = C_.c_void_p, C_.c_uint32
    h.run_callers.argtypes = [vp, vp, vp, u32, u32, u32, u32, u32, u32, C_.POINTER(C_.c_double), vp, vp, vp]
    h.run_callers.restype = C_.c_int32
    return h


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launcher_command(n, port, argv):
    """what `--gpus N` runs when no launcher started this process: one rank per GPU of this node (the driver's own command)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.join(ROOT, "bench.py")] + list(argv)


def launch_ranks(args, argv):
    """`--gpus N` (N > 1) with no WORLD_SIZE: start the N ranks ourselves and pass their single JSON line through."""
    if not args.launcher_selftest:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible on this node; refusing to run fewer ranks "
                             f"than asked (a line with n_gpus < {args.gpus} would misreport the run)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, effective_cores() // args.gpus)))
    cmd = launcher_command(args.gpus, free_port(), argv)
    print("bench.py: launching " + " ".join(cmd), file=sys.stderr)
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------------------------
# --launcher-selftest: the launch + rank plumbing on CPU (gloo), no search (tests/test_bench_launcher.py)
# ------------------------------------------------------------------------------------------------------------------
def launcher_selftest(args):
    """Every rank makes a small synthetic per-shard top-k record, the records go through the packed all-gather
    (cosdata_amd/sharding.py — the torch exchange path of the bench) and a host merge; timing is bracketed by a barrier and
    reduced with MAX over ranks like the real run.  Rank 0 prints ONE JSON line with n_gpus = the ranks that took part."""
    import torch
    import torch.distributed as dist
    from cosdata_amd.sharding import allgather_packed, packed_views, packed_words
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("gloo")
    B, k = 64, 10
    g = torch.Generator().manual_seed(1234 + rank)
    rec = torch.zeros(packed_words(B, k), dtype=torch.int32)
    ids, sc, cnt = packed_views(rec, B, k)
    sc.copy_(torch.rand(B, k, generator=g).sort(dim=1, descending=True).values)
    ids.copy_((torch.arange(B * k, dtype=torch.int32).view(B, k) % 1000) + rank * 1000)       # shard `rank` owns ids [1000 r, 1000 (r+1))
    cnt.fill_(k)
    pids = [None] * world
    dist.all_gather_object(pids, os.getpid())
    dist.barrier()
    t = time.perf_counter()
    for _ in range(max(1, args.steps)):
        gathered = allgather_packed(rec)
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    views = [packed_views(gathered[s].contiguous(), B, k) for s in range(world)]
    all_sc = torch.cat([v[1] for v in views], 1)
    all_id = torch.cat([v[0] for v in views], 1)
    top = all_sc.topk(k, dim=1)
    merged_ids = torch.gather(all_id, 1, top.indices)
    owners = sorted(set((merged_ids // 1000).flatten().tolist()))
    if rank == 0:
        print(json.dumps({"metric": "launcher self-test (gloo, CPU): rank plumbing + packed all-gather; NOT a benchmark", "value": 0.0,
                          "unit": "queries/s", "n_gpus": world, "steps": max(1, args.steps), "warmup": 0,
                          "ms_per_step": float(el.item()) / max(1, args.steps) * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "u8", "data": "synthetic (no search)",
                          "config": {"workload": "launcher-selftest", "value_workload_at_this_n": value_workload(world), "n_gpus": world,
                                     "dim": WORKLOADS[MAIN_WORKLOAD][1]},
                          "ranks": world, "rank_pids": pids, "distinct_processes": len(set(pids)), "shards_in_merged_answer": owners,
                          "asked_gpus": args.gpus}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------
# the dense HNSW workloads (c2, c2_uniform, c4shard): data + index + measurement, one record per (build, search) mode
# ------------------------------------------------------------------------------------------------------------------
class Env:
    """process-wide run state: rank/world, device, streams' owner"""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.args = args
        # COS_FORCE_DIST=1 exercises the N>1 code path (process group, all-gather, merge kernel) with a single rank
        self.force_dist = os.environ.get("COS_FORCE_DIST", "0") == "1"
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist_on = self.world > 1 or self.force_dist
        self.dist = None
        if self.dist_on:
            import torch.distributed as dist
            self.dist = dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{self.local_rank}"))
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device(f"cuda:{self.local_rank}")

    def sync_all(self):
        self.torch.cuda.synchronize(self.dev)
        if self.dist_on:
            self.dist.barrier()
            self.torch.cuda.synchronize(self.dev)


def mixture(torch, n, d, seed, device, centers, sigma=0.8):
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


def uniform_corpus(torch, n, d, seed, device):
    """values uniform(-1, 1) per component, NOT normalised: tests/test.py:88 (`np.random.uniform(-1, 1, dim)`)"""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty(n, d, device=device, dtype=torch.float32)
    step = 1 << 18
    for s in range(0, n, step):
        m = min(step, n - s)
        out[s:s + m] = torch.rand(m, d, generator=g, device=device) * 2.0 - 1.0
    return out


def bruteforce_top10(torch, X, Q, k=10):
    """exact cosine top-k by torch matmul (independent cross-check of the engine's own ground truth).
    Column chunks of 1M: torch.topk over multi-million-wide rows was observed to be unreliable."""
    ids = []
    Qn = Q / Q.norm(dim=1, keepdim=True)
    for s in range(0, Q.shape[0], 256):
        ci, cs = [], []
        for c in range(0, X.shape[0], 1 << 20):
            xc = X[c:c + (1 << 20)]
            t = ((Qn[s:s + 256] @ xc.T) / xc.norm(dim=1)[None, :]).topk(k, dim=1)
            ci.append(t.indices + c)
            cs.append(t.values)
        ci, cs = torch.cat(ci, 1), torch.cat(cs, 1)
        ids.append(torch.gather(ci, 1, cs.topk(k, dim=1).indices))
    return torch.cat(ids)


class DenseWorkload:
    """One dense corpus resident in HBM + its query sets + ground truth; `run_mode()` builds a graph with the given
    visited filters and measures it.  Several modes share the corpus, the ground truth and the oracle's quantized copy."""

    def __init__(self, env, name, n_override=0, ef_construction=0, quantization="auto", build_batch=4096, append_rows=0):
        import cosdata_amd as ca
        self.ca = ca
        self.env, self.name = env, name
        args, torch = env.args, env.torch
        n, d, corpus, efc, desc = WORKLOADS[name]
        self.n = n_override or n
        self.standard_size = not bool(n_override)
        self.d, self.corpus_kind, self.desc = d, corpus, desc
        self.ef_construction = ef_construction or efc
        self.quantization, self.build_batch = quantization, build_batch
        self.k, self.Bc, self.C = args.top_k, args.batch, max(1, args.coalesce)
        self.B = self.Bc * self.C
        self.S = args.inflight if args.inflight > 0 else (3 if env.dist_on else 2)
        self.nrq = args.recall_queries
        dev, rank, world = env.dev, env.rank, env.world
        n = self.n
        if corpus in ("mixture", "mixture_sigma01"):
            gc = torch.Generator(device=dev)
            gc.manual_seed(4242)
            # mixture_sigma01 = SURVEY 8(d) literally: 1 000 centres, sigma 0.1 PER COMPONENT (noise norm 0.1 sqrt(d) = 2.8 against unit
            # centres: the cluster structure is a small perturbation of an isotropic Gaussian — much harder than sigma 0.8 / sqrt(d))
            hard = corpus == "mixture_sigma01"
            self.n_centers = 1000 if hard else max(64, (n * world) // 1000)
            centers = torch.randn(self.n_centers, d, generator=gc, device=dev)
            centers /= centers.norm(dim=1, keepdim=True)
            sig = 0.1 * d ** 0.5 if hard else 0.8
            draw = lambda m, seed: mixture(torch, m, d, seed, dev, centers, sigma=sig)
            self.corpus_desc = (f"Gaussian mixture, {self.n_centers} centres, sigma " + ("0.1 per component (SURVEY 8d)" if hard else "0.8/sqrt(d)")
                                + ", L2-normalised, seed 42")
        else:
            draw = lambda m, seed: uniform_corpus(torch, m, d, seed, dev)
            self.corpus_desc = ("uniform(-1,1) per component, not normalised (tests/test.py:88), seed 42; queries are stored vectors, like "
                                "tests/test.py:120-139 searches the vectors it inserted (independent uniform draws have no neighbours to find "
                                "in 768 dimensions)")
        # append_rows more vectors of the same distribution behind the shard's n rows: what run_mode's append probe inserts into the RESIDENT
        # graph (cos_index_append with the caller's grown table, borrowed like the first n rows)
        self.append_rows = int(append_rows)
        self.X_all = draw(n + self.append_rows, 42 + 1000 * rank)   # this rank's shard: global ids [rank*n, (rank+1)*n)
        self.X = self.X_all[:n]
        draw_q = draw
        if corpus == "uniform" and world == 1:
            def draw_q(m, seed):
                gq = torch.Generator(device=dev)
                gq.manual_seed(seed)
                return self.X[torch.randint(0, n, (m,), generator=gq, device=dev)].contiguous()
        # one fresh query set per step of the timed region (warm-up included) up to 32 sets: no launch re-walks a set an earlier one
        # left in the caches (round 3 alternated over two sets)
        self.n_qsets = max(self.S, 2, min(32, max(1, args.steps) + max(0, args.warmup)))
        self.Q = draw_q(self.B * self.n_qsets, 43)               # timed queries; identical on every rank
        self.Q_sel = draw_q(self.nrq, 44)                         # ef selection set
        self.Q_rep = draw_q(self.nrq, 45)                         # disjoint hold-out: the recall that is REPORTED
        torch.cuda.synchronize()
        # "auto" quantization = u8 + values_range sampled from the first sample_threshold embeddings
        # (indexes/hnsw/mod.rs:202-351; tests/rps-test.py:73 uses the same mode)
        self.values_range = ca.sample_values_range(self.X[:1000].cpu().numpy(), 1.0) if quantization == "auto" else (-1.0, 1.0)
        if env.dist_on and world > 1:  # every shard must quantize with the same range: take rank 0's sample
            vr = torch.tensor(self.values_range, device=dev, dtype=torch.float64)
            env.dist.broadcast(vr, 0)
            self.values_range = (float(vr[0].item()), float(vr[1].item()))
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(self.S)]
        from cosdata_amd.sharding import packed_views, packed_words
        B, k, S = self.B, self.k, self.S
        # the shard's result lives in ONE packed record [ids | scores | counts] so the sharded path exchanges it
        # with a single all-gather per launch (sharding.py / cos_shardset_*)
        self.o_pack = torch.zeros(S, packed_words(B, k), dtype=torch.int32, device=dev)
        views = [packed_views(self.o_pack[s], B, k) for s in range(S)]
        self.o_ids, self.o_sc, self.o_cnt = [v[0] for v in views], [v[1] for v in views], [v[2] for v in views]
        self.o_st = torch.zeros(S, B, dtype=torch.int32, device=dev)
        if env.dist_on:
            self.g_pack = torch.zeros(S, world, packed_words(B, k), dtype=torch.int32, device=dev)
            self.m_ids = torch.zeros(S, B, k, dtype=torch.int32, device=dev)
            self.m_sc = torch.zeros(S, B, k, dtype=torch.float32, device=dev)
            self.m_cnt = torch.zeros(S, B, dtype=torch.int32, device=dev)
        self.gt_sel = self.gt_rep = None
        self.flat = None
        self.oracle = None          # (OracleIndex, full_raw flag, host queries)
        self.Qh = None

    def close(self):
        self.oracle = None
        for a in ("X", "X_all", "Q", "Q_sel", "Q_rep", "o_pack", "o_st", "g_pack", "m_ids", "m_sc", "m_cnt", "gt_sel", "gt_rep", "o_ids", "o_sc", "o_cnt"):
            if hasattr(self, a):
                setattr(self, a, None)
        self.env.torch.cuda.empty_cache()

    # ---- ground truth -----------------------------------------------------------------------------------------
    def _merge_global(self, Qset, local_ids_global):
        """global answer = merge of the per-shard lists by exact cosine (recall bookkeeping only)"""
        from cosdata_amd.sharding import global_topk_by_score
        torch, n, rank = self.env.torch, self.n, self.env.rank
        loc = (local_ids_global - rank * n).clamp_(0, n - 1)
        rows = self.X[loc]
        sims = torch.einsum("qd,qkd->qk", Qset, rows) / (Qset.norm(dim=1, keepdim=True) * rows.norm(dim=2))
        return global_topk_by_score(local_ids_global, sims, self.k)

    def _ground_truth(self, ix):
        """the engine's own exhaustive scan (f32-MFMA GEMM -> top-64 -> reference-order re-score), cross-checked against an
        independent torch matmul + topk on a slice"""
        env, torch, n, d, k, rank = self.env, self.env.torch, self.n, self.d, self.k, self.env.rank

        def gt(Qset):
            qh = Qset.cpu().numpy()
            torch.cuda.synchronize(env.dev)
            t = time.perf_counter()
            ids, _ = ix.bruteforce_topk(qh, k)
            return torch.from_numpy(ids.astype(np.int64) - rank * n).to(env.dev), time.perf_counter() - t
        gt_sel_local, bf_seconds = gt(self.Q_sel)
        gt_rep_local, _ = gt(self.Q_rep)
        nx = min(self.nrq, 1024)
        gt_torch = bruteforce_top10(torch, self.X, self.Q_sel[:nx], k)
        agree = float((gt_sel_local[:nx].unsqueeze(2) == gt_torch.unsqueeze(1)).any(dim=2).float().mean().item())
        self.flat = {"queries": self.nrq, "seconds": bf_seconds, "tflops_end_to_end": 2.0 * self.nrq * n * d / bf_seconds / 1e12,
                     "agreement_with_torch_topk": agree, "torch_checked_queries": nx,
                     "note": "cos_bruteforce_topk incl. H2D/D2H, selection and exact re-score; kernel-level MFMA rate: profiles/"}
        self.gt_sel = self._merge_global(self.Q_sel, gt_sel_local + rank * n) if env.dist_on else gt_sel_local
        self.gt_rep = self._merge_global(self.Q_rep, gt_rep_local + rank * n) if env.dist_on else gt_rep_local

    # ---- one (build filter, search filter) mode -------------------------------------------------------------------
    def run_mode(self, build_visited, visited, ef_arg="auto", ef_sweep="", cpu_seconds=12.0, single_batch=False, host_api=False,
                 hbm_probe=False, exchange="auto", m0=64, m_upper=32, append_probe=False):
        env, ca, torch = self.env, self.ca, self.env.torch
        args = env.args
        dev, rank, world, local_rank, dist, dist_on = env.dev, env.rank, env.world, env.local_rank, env.dist, env.dist_on
        n, d, k, B, Bc, S, nrq = self.n, self.d, self.k, self.B, self.Bc, self.S, self.nrq
        X, Q, streams = self.X, self.Q, self.streams
        o_ids, o_sc, o_cnt, o_st, o_pack = self.o_ids, self.o_sc, self.o_cnt, self.o_st, self.o_pack
        n_launch, n_warm = max(1, args.steps), max(0, args.warmup)
        ef = 256 if ef_arg == "auto" else int(ef_arg)
        t_setup = time.time()
        mode_of = lambda v: ca.VISITED_EXACT if v == "exact" else ca.VISITED_REF

        # index: reference defaults (config.toml:20-24,32)
        # level_0_neighbors_count is a user hyper-parameter of the reference (indexes/hnsw/types.rs:10-17) and also the size of its
        # visited filter, PerformantFixedSet::new(level_0_neighbors_count) (vector_store.rs:266-270): 64 = config.toml's default
        hp = ca.HNSWHyperParams(num_layers=9, ef_construction=self.ef_construction, ef_search=ef, level_0_neighbors_count=m0, neighbors_count=m_upper)
        ix = ca.HNSWIndex(d, hp, ca.DistanceMetric.Cosine, ca.StorageType.UnsignedByte(), self.values_range, shortlist_size=64,
                          device=local_rank, id_base=rank * n, seed=42 + rank, visited_mode=mode_of(build_visited))
        ix.upload_vectors_device(X.data_ptr(), n, keepalive=X)
        t0 = time.time()
        ix.build(self.build_batch)
        build_s = time.time() - t0
        ix.set_visited_mode(mode_of(visited))
        if self.gt_sel is None:
            self._ground_truth(ix)
        gt_sel, gt_rep = self.gt_sel, self.gt_rep

        exchange_kind = None
        shardset = None
        if dist_on:
            from cosdata_amd.sharding import allgather_packed, merge_topk_packed_device
            g_pack, m_ids, m_sc, m_cnt = self.g_pack, self.m_ids, self.m_sc, self.m_cnt
            exchange_kind = "torch.distributed all_gather_into_tensor (RCCL) + cos_merge_topk_packed_device"
            if exchange in ("auto", "shardset"):
                try:
                    from cosdata_amd.shardset import ProcessShardSet
                    shardset = ProcessShardSet(ix, rank, world, local_rank, dist)
                    exchange_kind = "cos_shardset_exchange_device: ncclAllGather (RCCL) + merge kernel inside the C ABI"
                except Exception as exc:  # noqa: BLE001 — reported in the JSON line, never silent
                    if exchange == "shardset":
                        raise
                    exchange_kind += f" [shardset unavailable: {type(exc).__name__}: {exc}]"
        lib = ca._lib.lib()

        def step(i):
            s = i % S
            st = streams[s]
            q = Q[(i % self.n_qsets) * B:(i % self.n_qsets + 1) * B]
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
            ann_g = self._merge_global(Qset, ann) if dist_on else ann
            hits = (ann_g.unsqueeze(2) == gt.unsqueeze(1)).any(dim=2).float().sum(dim=1)
            if dist_on:
                dist.broadcast(hits, 0)  # every rank must take the same decision
            return recall_stats(hits.cpu().numpy(), k)

        ef_table = []
        if ef_arg == "auto":
            ef, ef_table = select_ef(EF_CANDIDATES, lambda e: measure_recall(e, self.Q_sel, gt_sel), args.recall_target)
        recall, recall_se, recall_lo = measure_recall(ef, self.Q_rep, gt_rep)     # the reported figure: hold-out set
        status_bad = int((o_st != 0).sum().item())

        # ---- size-independent properties of the returned lists, checked at the full workload size (no oracle needed): ids in
        # this shard's range and unique per query, scores non-increasing, every score = the exact f32 cosine of the row it names
        m = min(B, nrq, 2048)
        ix.batch_search_device(self.Q_rep[:m].data_ptr(), m, k, o_ids[0].data_ptr(), o_sc[0].data_ptr(), o_cnt[0].data_ptr(), o_st[0].data_ptr(),
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
        qd = self.Q_rep[:m].double()
        exact = torch.einsum("qd,qkd->qk", qd, rows) / (qd.norm(dim=1, keepdim=True) * rows.norm(dim=2))
        max_err = float((sc.double() - exact).abs()[valid].max().item())
        props = {"queries": m, "ids_in_shard_range": in_range, "ids_unique": unique, "scores_sorted_desc": sorted_desc,
                 "max_abs_err_vs_f64_cosine": max_err, "full_lists": bool((cnt == k).all().item())}
        del rows, qd, exact

        # ---- warmup + timed region: EXACTLY n_launch steps (launches) ---------------------------------------
        def timed_run():
            for i in range(n_warm):
                step(i)
            env.sync_all()
            ix.enable_timing(True)
            t = time.perf_counter()
            for i in range(n_warm, n_warm + n_launch):          # query set i % n_qsets: every step of a default run has its own
                step(i)
            env.sync_all()
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
            cnts, splits = [], []
            for s in range(min(S, n_launch)):
                ts = ix.timing_summary(streams[s].cuda_stream)
                walk_sum += ts.walk_ms_sum; prep_sum += ts.prep_ms_sum; fin_sum += ts.finalize_ms_sum; nl += ts.launches
                walk_min, walk_max = min(walk_min, ts.walk_ms_min), max(walk_max, ts.walk_ms_max)
                stt = ix.last_stats(streams[s].cuda_stream)
                cnts.append((stt.evals * row_bytes + stt.adj_bytes, stt.evals, stt.expansions, stt.reserved))
                splits.append(ix.last_walk_split(streams[s].cuda_stream))   # the last launch of the stream, dispatch by dispatch
            ix.enable_timing(False)
            mean = lambda f: float(np.mean([f(x) for x in splits]))
            sp = {"table_level_min": int(splits[0].table_level_min), "table_cols": int(splits[0].table_cols), "cut_after_level": int(splits[0].cut_after_level),
                  "table_ms": mean(lambda x: x.table_ms), "upper_ms": mean(lambda x: x.upper_ms), "sort_ms": mean(lambda x: x.sort_ms),
                  "lower_ms": mean(lambda x: x.lower_ms), "table_int8_ops": mean(lambda x: x.table_int8_ops), "table_evals": mean(lambda x: x.table_evals),
                  "upper_evals": mean(lambda x: x.upper_evals), "upper_expansions": mean(lambda x: x.upper_expansions),
                  "upper_adj_bytes": mean(lambda x: x.upper_adj_bytes), "lower_evals": mean(lambda x: x.lower_evals),
                  "lower_expansions": mean(lambda x: x.lower_expansions), "lower_adj_bytes": mean(lambda x: x.lower_adj_bytes),
                  "launches_sampled": len(splits)}
            return {"elapsed": el, "walk_ms": walk_sum / nl, "prep_ms": prep_sum / nl, "finalize_ms": fin_sum / nl, "walk_ms_min": walk_min,
                    "walk_ms_max": walk_max, "timed_launches_sampled": nl, "bytes": float(np.mean([c[0] for c in cnts])),
                    "evals": float(np.mean([c[1] for c in cnts])), "expansions": float(np.mean([c[2] for c in cnts])),
                    "rounds": float(np.mean([c[3] for c in cnts])), "split": sp}

        tr = timed_run()
        # the same step three more times, ONE launch at a time: the durations of its dispatches without a neighbouring launch's
        # kernels on the chip (what a serialized rocprofv3 trace shows); the timed region's own events stretch a short kernel that
        # shares the chip with another stream's walk (the level-table GEMM: 0.4 ms alone, ~3 ms under a walk)
        ix.enable_timing(True)
        alone = []
        for i in range(3):
            step(n_warm + n_launch + i)
            env.sync_all()
            alone.append(ix.last_walk_split(streams[(n_warm + n_launch + i) % S].cuda_stream))
        ix.enable_timing(False)
        sp_alone = {k_: float(np.mean([getattr(x, k_) for x in alone])) for k_ in ("table_ms", "upper_ms", "sort_ms", "lower_ms")}
        # the locality order of big launches (cos_index_set_walk_order: from WALK_ORDER_DEFAULT_MIN_B queries, ef <= 256)
        cuts = ix.walk_order_cuts() if B >= ca.HNSWIndex.WALK_ORDER_DEFAULT_MIN_B and ef <= 256 else []
        elapsed = tr["elapsed"]
        merged_qps = n_launch * B / elapsed        # answers over the global (world x n) corpus per second
        # Algorithmic bytes of one step's walk AS THIS DESIGN EVALUATES IT (SURVEY.md 8d's per-unit figures): an evaluation that gathers
        # and dots a code row reads bytes_vec + 4; an evaluation served by the level table (cos_index_set_walk_table) reads the 4-byte
        # similarity the table GEMM left — that GEMM is a dispatch of its own with its own (MFMA) roofline below; an expansion reads
        # M_level x 4 of adjacency.  `reference_algorithmic_bytes` keeps SURVEY's figure for the same walk without a table (every
        # evaluation a row): the number round 3 divided by the walk time and that came out above the HBM peak, because it charges to HBM
        # rows that never leave the L2.
        sp = tr["split"]
        row_b = d + 4
        ref_alg_bytes = tr["bytes"]
        tab_upper = min(sp["table_evals"], sp["upper_evals"]) if sp["cut_after_level"] else 0.0   # the table's levels lie above the cut
        tab_lower = sp["table_evals"] - tab_upper
        upper_bytes = (sp["upper_evals"] - tab_upper) * row_b + tab_upper * 4.0 + sp["upper_adj_bytes"]
        lower_bytes = (sp["lower_evals"] - tab_lower) * row_b + tab_lower * 4.0 + sp["lower_adj_bytes"]
        avg_bytes = upper_bytes + lower_bytes
        avg_ms = tr["walk_ms"]                                           # ring average over every timed launch: upper range | sort | lower range
        kern_ms = avg_ms - (sp_alone["sort_ms"] if sp["cut_after_level"] else 0.0)   # the walk kernel's own dispatches (sp[...]: the last launch of each stream only)
        kernel_gbps = avg_bytes / (kern_ms * 1e-3) / 1e9              # ONE step's walk: algorithmic bytes / HIP-event time of its dispatches
        aggregate_gbps = avg_bytes * n_launch / elapsed / 1e9         # all launches / wall time (launches on different streams overlap)
        overlap = max(1.0, min(float(S), avg_ms * 1e-3 * n_launch / elapsed))

        # single-stream (one un-coalesced client batch at a time) rate: launches this small take the latency variant of the walk
        # (cos_index_set_latency_mode); the throughput kernel is timed on the same batch and must return the same bits
        serial = None
        if single_batch:
            def serial_rate():
                for _ in range(2):
                    ix.batch_search_device(Q[:Bc].data_ptr(), Bc, k, o_ids[0].data_ptr(), o_sc[0].data_ptr(), o_cnt[0].data_ptr(),
                                           o_st[0].data_ptr(), streams[0].cuda_stream)
                streams[0].synchronize()
                t1 = time.perf_counter()
                for _ in range(8):
                    ix.batch_search_device(Q[:Bc].data_ptr(), Bc, k, o_ids[0].data_ptr(), o_sc[0].data_ptr(), o_cnt[0].data_ptr(),
                                           o_st[0].data_ptr(), streams[0].cuda_stream)
                streams[0].synchronize()
                rate = 8 * Bc / (time.perf_counter() - t1)
                return rate, (o_ids[0][:Bc].clone(), o_sc[0][:Bc].clone().view(torch.int32), o_cnt[0][:Bc].clone(), o_st[0][:Bc].clone())
            torch.cuda.synchronize(dev)
            serial_qps, serial_out = serial_rate()              # default policy: throughput kernel + level table up to ef 64, four waves + table above
            ix.set_walk_table(0, 0)                             # without a table the named kernels are selectable
            ix.set_latency_waves(0)
            serial_qps_1w, serial_out_1w = serial_rate()        # one-wave latency kernel
            ix.set_latency_mode(0)
            serial_qps_tk, serial_out_tk = serial_rate()        # throughput kernel, no table
            ix.set_latency_mode(ca.HNSWIndex.LATENCY_MODE_DEFAULT_MAX_B)
            ix.set_latency_waves(ca.HNSWIndex.LATENCY_WAVES_DEFAULT_MAX_B)
            ix.set_walk_table(ca.HNSWIndex.WALK_TABLE_AUTO, ca.HNSWIndex.WALK_TABLE_DEFAULT_MIN_B)
            serial = {"qps": serial_qps, "qps_one_wave_latency_kernel": serial_qps_1w, "qps_throughput_kernel": serial_qps_tk,
                      "identical": all(bool(torch.equal(a, b_)) and bool(torch.equal(a, c_)) for a, b_, c_ in zip(serial_out, serial_out_tk, serial_out_1w))}

        # ---- optional ef_search sweep (same index, same launch shape): QPS and hold-out recall per setting ----------------------
        sweep = []
        main_mode = mode_of(visited)
        for tok in [v for v in ef_sweep.split(",") if v]:   # "256" or "ref:256" / "exact:64" (other visited filter, same graph)
            mode_name, _, efs = tok.rpartition(":")
            ef2 = int(efs)
            mode2 = {"": main_mode, "ref": ca.VISITED_REF, "exact": ca.VISITED_EXACT}[mode_name]
            ix.set_visited_mode(mode2)
            rec2 = measure_recall(ef2, self.Q_rep, gt_rep)
            tr2 = timed_run()
            sweep.append({"ef_search": ef2, "visited": "exact" if mode2 == ca.VISITED_EXACT else "ref", "qps": n_launch * B / tr2["elapsed"],
                          "recall_at_10": rec2[0], "recall_stderr": rec2[1], "walk_ms": tr2["walk_ms"],
                          "kernel_GBps": tr2["bytes"] / (tr2["walk_ms"] * 1e-3) / 1e9})
        ix.set_ef_search(ef)
        ix.set_visited_mode(main_mode)

        # ---- launch size: the same index at half and twice the step's queries per launch (dynamic batching depth).  A bigger launch
        # puts more queries into every region of the graph, so neighbouring queries find more of each other's rows in the XCD's L2: the
        # rate keeps rising with the launch size (and so does the time a query waits for its launch) — `value` stays at the step this
        # bench has used since round 1 (128 client batches).
        size_sweep = []
        if single_batch and not dist_on:
            for mult_num, mult_den in ((1, 2), (2, 1)):
                Bx = B * mult_num // mult_den
                Qx = Q[:Bx] if Bx <= Q.shape[0] else None
                if Qx is None:
                    continue
                outs = [(torch.zeros(Bx, k, dtype=torch.int32, device=dev), torch.zeros(Bx, k, dtype=torch.float32, device=dev),
                         torch.zeros(Bx, dtype=torch.int32, device=dev), torch.zeros(Bx, dtype=torch.int32, device=dev)) for _ in range(2)]
                def lx(i):
                    q = Q[((i % max(1, Q.shape[0] // Bx)) * Bx):((i % max(1, Q.shape[0] // Bx)) + 1) * Bx]
                    o = outs[i % 2]
                    ix.batch_search_device(q.data_ptr(), Bx, k, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), streams[i % 2].cuda_stream)
                for i in range(4):
                    lx(i)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                nx = 10
                for i in range(nx):
                    lx(i)
                torch.cuda.synchronize(dev)
                size_sweep.append({"queries_per_launch": Bx, "launches_in_flight": 2, "qps": nx * Bx / (time.perf_counter() - t1)})
                del outs

        # ---- the boundary as the Rust host would call it: host buffers in, host buffers out (cos_search_batch: H2D of the queries,
        # the same kernels, D2H of ids / scores / counts).  PCIe-inclusive, reported next to `value`, never it.
        host = None
        if host_api and rank == 0 and world == 1:
            import threading
            qh = [Q[j * B:(j + 1) * B].cpu().numpy() for j in range(min(self.n_qsets, S))]
            ix.batch_search(qh[0], k)                               # warm-up: pipe + workspaces
            t1 = time.perf_counter()
            reps_h = 8
            for _ in range(reps_h):
                ix.batch_search(qh[0], k)
            el_h = (time.perf_counter() - t1) / reps_h
            # the same S launches in flight as `value`: S host threads, each one synchronous call at a time (the reference's callers
            # are concurrent rayon / actix workers, indexes/mod.rs:268-271); a call's copies hide under its neighbour's walk
            def caller(j):
                for _ in range(reps_h):
                    ix.batch_search(qh[j % len(qh)], k)
            def concurrent(nc):
                for j in range(nc):
                    ix.batch_search(qh[j % len(qh)], k)
                th = [threading.Thread(target=caller, args=(j,)) for j in range(nc)]
                t1 = time.perf_counter()
                [t.start() for t in th]
                [t.join() for t in th]
                return time.perf_counter() - t1
            el_c = concurrent(S)
            el_c1 = concurrent(S + 1)   # one caller more than launches in flight: a call's finalize + copies never delay the next walk
            host = {"queries_per_call": B, "callers": S, "qps": S * reps_h * B / el_c, "ms_per_call_per_caller": el_c / reps_h * 1e3,
                    "qps_with_one_more_caller": (S + 1) * reps_h * B / el_c1,
                    "single_caller_qps": B / el_h, "single_caller_ms_per_call": el_h * 1e3,
                    "h2d_bytes_per_call": int(B) * d * 4, "d2h_bytes_per_call": int(B) * (k * 8 + 4),
                    "note": "cos_search_batch on pageable host memory, PCIe-inclusive.  qps: %d concurrent synchronous callers (the launches in "
                            "flight `value` is measured with); single_caller: one call at a time, where the call itself runs as a chunk pipeline "
                            "(H2D of chunk i+1 and finalize of chunk i-1 under the walk of chunk i, engine.hip search_host_pipelined)" % S}
            # ---- the reference's literal calling pattern (indexes/mod.rs:268-271, tests/rps-test.py:426-455): many concurrent callers of
            # ONE 256-query batch each, fused by the library's dynamic batching (cos_index_set_coalescing) into launches of up to B
            # queries; PCIe-inclusive.  Every caller's answer must be the answer of an un-coalesced call.
            small = []
            qsmall = np.ascontiguousarray(Q[:B].cpu().numpy().reshape(self.C, Bc, d))
            direct = ix.batch_search(qsmall[0], k)
            import ctypes as C_
            harness = native_callers_harness()       # scripts/callers_bench.cpp: native threads (128 Python threads measure the GIL)
            fn = C_.cast(lib.cos_search_batch, C_.c_void_p)
            # max_queries = half of what the callers offer at once, so that two launches alternate (one on the device, one being copied in
            # and out): 64 x 256 -> 8 192, 128 -> 16 384, 256 -> 32 768; plus the round-4 pairs that put every caller in ONE launch
            for nc, maxq in ((self.C // 2, B // 4), (self.C, B // 2), (2 * self.C, B), (self.C // 2, B // 2), (self.C, B)):
                ix.set_coalescing(maxq, 300)
                reps_s = 16
                secs = C_.c_double(0.0)
                i0 = np.zeros((Bc, k), np.uint32); s0 = np.zeros((Bc, k), np.float32); c0 = np.zeros(Bc, np.uint32)
                nfail, best = 0, 1e30
                for _ in range(3):      # best of three: with more native threads than host cores a run's rate follows the scheduler's mood
                    nfail += harness.run_callers(fn, ix._h, qsmall.ctypes.data_as(C_.c_void_p), self.C, Bc, d, k, nc, reps_s, C_.byref(secs),
                                                 i0.ctypes.data_as(C_.c_void_p), s0.ctypes.data_as(C_.c_void_p), c0.ctypes.data_as(C_.c_void_p))
                    best = min(best, secs.value)
                secs.value = best
                same = all(np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b_).view(np.uint32)) for a, b_ in zip((i0, s0, c0), direct))
                cst = ix.coalescing_stats()     # over the three runs: how many launches the callers' requests became, and why they left
                small.append({"callers": nc, "queries_per_call": Bc, "coalescing_max_queries": maxq, "coalescing_window_us": 300, "calls_per_caller": reps_s, "runs": 3,
                              "qps": nc * reps_s * Bc / secs.value, "failed_calls": int(nfail), "identical_to_uncoalesced_call": bool(same),
                              "launches": cst["launches"], "queries_per_launch": round(cst["queries"] / max(1, cst["launches"]), 1),
                              "launches_left_full_quiet_deadline": [cst["closed_full"], cst["closed_quiet"], cst["closed_deadline"]],
                              "solo_calls": cst["solo_calls"]})
            ix.set_coalescing(0, 0)
            host["concurrent_256_query_callers"] = small
            del qh, qsmall

        # ---- CPU baseline + parity: the oracle (C restatement of the Rust path) on this box's host cores, same graph ----------
        cpu = parity = None
        if rank == 0 and world == 1 and not args.no_cpu_baseline and cpu_seconds > 0:
            cpu, parity = self._cpu_baseline_and_parity(ix, ef, visited, cpu_seconds, m0, m_upper)

        # HBM traffic of the walk kernel: PMC FETCH_SIZE cannot be read from inside the process; it is taken from the
        # committed rocprofv3 --pmc pass of this same command (profiles/pmc_traffic.json, written by
        # scripts/pmc_traffic.py) when the workload / ef / launch shape match, else null.
        traffic, traffic_parts, traffic_k = None, {}, None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
                for ent in json.load(fh):
                    if (ent["workload"], ent["ef_search"], ent["queries_per_launch"]) == (self.name, ef, B) and self.standard_size \
                            and ent.get("visited", "ref") == visited:
                        traffic = ent["hbm_bytes_per_launch"]
                        traffic_parts = {p: v["bytes"] for p, v in (ent.get("parts") or {}).items()}
                        traffic_k = ent.get("fetch_factor_k")
        except (OSError, ValueError, KeyError):
            pass
        issue = None
        try:   # VALU / SALU issue of the walk kernel from the committed SQ pass of the same command (scripts/final_profile.sh)
            with open(os.path.join(ROOT, "profiles", "pmc_issue.json")) as fh:
                for ent in json.load(fh):
                    if (ent["workload"], ent["ef_search"], ent["queries_per_launch"]) == (self.name, ef, B) and ent.get("visited", "ref") == visited:
                        issue = {k: ent[k] for k in ("valu_per_step", "salu_per_step", "valu_per_eval", "salu_per_eval", "valu_busy", "salu_busy", "source") if k in ent}
        except (OSError, ValueError, KeyError):
            pass

        def part(bound, alg, ms, traffic_b=None, **extra):
            """one dispatch of the walk: `achieved` is the MEASURED fabric-side rate when the committed PMC pass has this dispatch (it cannot
            exceed the roof), else the algorithmic one; both are given"""
            alg_gbps = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            r = {"bound": bound, "algorithmic_bytes": alg, "ms": ms, "algorithmic_GBps": alg_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s"}
            if traffic_b is not None and ms > 0:
                r["traffic"] = traffic_b
                r["achieved"] = traffic_b / (ms * 1e-3) / 1e9
                r["achieved_basis"] = "fabric-side traffic (PMC FETCH_SIZE x k + WRITE_SIZE of the committed pass) / HIP-event time"
            else:
                r["achieved"] = alg_gbps
                r["achieved_basis"] = "algorithmic bytes / HIP-event time (no PMC pass for this launch shape)"
            r["frac"] = r["achieved"] / HBM_PEAK_GBPS
            r.update(extra)
            return r
        parts = None
        if sp["cut_after_level"] or sp["table_level_min"]:
            parts = {}
            if sp["table_level_min"]:
                tops = sp["table_int8_ops"] / (sp_alone["table_ms"] * 1e-3) / 1e12 if sp_alone["table_ms"] > 0 else 0.0
                tab_bytes = float(B) * sp["table_cols"] * 4.0
                wr_gbps = tab_bytes / (sp_alone["table_ms"] * 1e-3) / 1e9 if sp_alone["table_ms"] > 0 else 0.0
                areg = (d % 64 == 0) and (d // 64) in (2, 4, 6, 8, 12, 16)
                parts["level_table_gemm"] = {"bound": "hbm", "kernel": ("level_table_areg<%d>" % (d // 64) if areg else "flat_codes_gemm_i8<ENG_U8>") + " (+ code sums)",
                                             "levels": f">= {sp['table_level_min']}", "columns": sp["table_cols"], "ms": sp_alone["table_ms"],
                                             "ms_next_to_a_walk": sp["table_ms"], "table_bytes_written": tab_bytes,
                                             "achieved": wr_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": wr_gbps / HBM_PEAK_GBPS,
                                             "traffic": traffic_parts.get("table_gemm"),
                                             "mfma": {"int8_ops": sp["table_int8_ops"], "achieved": tops, "peak": I8_PEAK_TOPS, "unit": "TOP/s", "frac": tops / I8_PEAK_TOPS},
                                             "note": "short-K GEMM whose output is the table: queries x columns x 4 bytes written once against 2 x K int8 ops "
                                                     "per output = 13 TB/s of output at the i8 peak (K = 768), so the bound is the HBM WRITE stream, the "
                                                     "matrix cores idle most of the time by construction; the epilogue is add + convert + store (the walk "
                                                     "forms the quotient for the entries it reads).  It runs on the caller's stream under the previous "
                                                     "step's walk"}
            if sp["cut_after_level"]:
                parts["walk_upper"] = part("hbm", upper_bytes, sp["upper_ms"], traffic_parts.get("upper"), ms_alone=sp_alone["upper_ms"],
                                           levels=f"{9}..{sp['cut_after_level']} in arrival order", evals=sp["upper_evals"], table_evals=sp["table_evals"],
                                           expansions=sp["upper_expansions"])
                parts["walk_lower"] = part("hbm", lower_bytes, sp["lower_ms"], traffic_parts.get("lower"), ms_alone=sp_alone["lower_ms"],
                                           levels=f"{sp['cut_after_level'] - 1}..0 in locality order", evals=sp["lower_evals"], expansions=sp["lower_expansions"],
                                           note="neighbouring queries of the sorted launch share rows in the XCD's L2: algorithmic_GBps of this dispatch exceeds "
                                                "the HBM peak, its measured traffic (achieved) cannot")
                parts["order_sort_ms"] = sp_alone["sort_ms"]

        # empirical HBM ceilings on THIS part: streaming read and the walk's own access pattern — random
        # gathers of d-byte rows over a buffer the size of the code array — measured after the timed region
        empirical = None
        if hbm_probe and rank == 0:
            import ctypes as C_
            g = C_.c_double(0.0)
            empirical = {}
            for pname, kind, nbytes, rb in (("stream_read", 0, 4 << 30, 0), ("stream_copy", 1, 2 << 30, 0),
                                            ("row_gather", 2, max(n * d, 1 << 20), d), ("row_gather_64MB_table", 2, 64 << 20, d)):
                ca._lib.check(lib.cos_hbm_probe(local_rank, kind, nbytes, rb, 3, C_.byref(g)))
                empirical[pname + "_GBps"] = g.value
            empirical["row_gather_row_bytes"] = d
            empirical["row_gather_buffer_bytes"] = n * d
            empirical["kernel_frac_of_row_gather"] = kernel_gbps / empirical["row_gather_GBps"]
            empirical["kernel_frac_of_stream_read"] = kernel_gbps / empirical["stream_read_GBps"]

        # ---- round 6: insert into / delete from the RESIDENT graph (cos_index_append / cos_index_delete; index_embeddings on a live index,
        # vector_store.rs:714-780, and delete_embedding, :1206-1400) instead of the rebuild every commit used to be.  LAST: the index grows.
        append = None
        if append_probe and self.append_rows and rank == 0 and world == 1:
            m = self.append_rows
            torch.cuda.synchronize(dev)
            t_a = time.perf_counter()
            ix.append_device(self.X_all.data_ptr(), m, self.build_batch, keepalive=self.X_all)
            torch.cuda.synchronize(dev)
            append_s = time.perf_counter() - t_a
            # queries near the NEW vectors: found?  (recall against brute force over the grown corpus, the engine's own exhaustive scan)
            nq = 1024
            gq = torch.Generator(device=dev)
            gq.manual_seed(4711)
            pick = torch.randint(n, n + m, (nq,), generator=gq, device=dev)
            Qn = self.X_all[pick] + (0.05 / d ** 0.5) * torch.randn(nq, d, generator=gq, device=dev)
            Qn = (Qn / Qn.norm(dim=1, keepdim=True)).contiguous()
            ix.batch_search_device(Qn.data_ptr(), nq, k, o_ids[0].data_ptr(), o_sc[0].data_ptr(), o_cnt[0].data_ptr(), o_st[0].data_ptr(), streams[0].cuda_stream)
            streams[0].synchronize()
            got = o_ids[0][:nq].to(torch.int64) & 0xFFFFFFFF
            gt_n, _ = ix.bruteforce_topk(Qn.cpu().numpy(), k)
            gt_n = torch.from_numpy(gt_n.astype(np.int64)).to(dev)
            rec_new = float((got.unsqueeze(2) == gt_n.unsqueeze(1)).any(dim=2).float().mean().item())
            self_found = float((got == pick[:, None]).any(dim=1).float().mean().item())
            # delete a handful of the appended ids (a transaction's deletes), then the same queries: none of them comes back
            dele = pick[:32].unique().cpu().numpy().astype(np.uint32)
            t_d = time.perf_counter()
            ix.delete(dele)
            torch.cuda.synchronize(dev)
            delete_s = time.perf_counter() - t_d
            ix.batch_search_device(Qn.data_ptr(), nq, k, o_ids[0].data_ptr(), o_sc[0].data_ptr(), o_cnt[0].data_ptr(), o_st[0].data_ptr(), streams[0].cuda_stream)
            streams[0].synchronize()
            got2 = (o_ids[0][:nq].to(torch.int64) & 0xFFFFFFFF).cpu().numpy()
            append = {"appended_vectors": m, "onto_resident_vectors": n, "append_seconds": append_s, "vectors_per_second": m / append_s,
                      "full_rebuild_seconds_for_comparison": build_s, "level0_nodes_after": ix.level_count(0),
                      "queries_near_new_vectors": nq, "recall_at_10_vs_bruteforce_over_grown_corpus": rec_new, "query_finds_its_own_new_vector": self_found,
                      "deleted_ids": int(dele.size), "delete_seconds": delete_s, "ms_per_delete": delete_s / max(1, dele.size) * 1e3,
                      "deleted_ids_returned_afterwards": int(np.isin(got2, dele).sum()),
                      "note": "cos_index_append continues the build's schedule on the link state the build left (graph parity with the oracle doing the same: "
                              "tests/test_gpu_append.py); cos_index_delete = one walk + one unlink kernel per id"}
        if shardset is not None:
            shardset.close()
        rec = {
            "value": merged_qps, "elapsed": elapsed, "steps": n_launch, "warmup": n_warm, "ef": ef, "ef_table": ef_table,
            "recall": (recall, recall_se, recall_lo), "status_bad": status_bad, "props": props, "sweep": sweep, "size_sweep": size_sweep, "serial": serial,
            "host_api": host, "cpu": cpu, "parity": parity, "append": append, "build_s": build_s, "seconds": time.time() - t_setup, "exchange_kind": exchange_kind,
            "config": {"workload": (value_workload(world) if (self.name, m0, m_upper, visited) == (MAIN_WORKLOAD, MAIN_M0, MAIN_M, "ref")
                                    else self.name + ": " + self.desc),
                       "n_gpus": world, "standard_size": self.standard_size, "vectors_per_gpu": n, "dim": d,
                       "step": f"one coalesced launch = {self.C} client batches x {Bc} queries = {B} queries through quantize -> walk -> rerank -> top-k",
                       "query_batch": Bc, "batches_per_launch": self.C, "queries_per_step": B, "launches_in_flight": S, "top_k": k, "ef_search": ef,
                       "ef_policy": ("smallest ef whose recall@10 on the selection query set is >= %.2f with 95%% confidence; recall_at_10 is "
                                     "measured on a disjoint hold-out set" % args.recall_target) if ef_arg == "auto" else "fixed",
                       "M": m_upper, "M0": m0, "num_layers": 9, "ef_construction": self.ef_construction, "build_visited": build_visited,
                       "storage": f"u8 (quantization {self.quantization}, values_range {self.values_range})",
                       "visited": "reference PerformantFixedSet (ID parity mode)" if visited == "ref" else "exact visited set (recall mode)",
                       "parallelism": f"id-range shards x{world}" + (" + RCCL all-gather top-k merge" if world > 1 else ""),
                       "exchange": exchange_kind, "corpus": self.corpus_desc},
            "roofline": {"bound": "hbm", "achieved": kernel_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": kernel_gbps / HBM_PEAK_GBPS,
                         "step_frac": avg_bytes / (elapsed / n_launch) / 1e9 / HBM_PEAK_GBPS,   # the same bytes over the whole step (GEMM, sort, rerank and what does not overlap included)
                         "traffic": traffic, "traffic_GBps": (traffic / (kern_ms * 1e-3) / 1e9) if traffic else None,
                         "traffic_frac_of_peak": (traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                         "traffic_fetch_factor_k": traffic_k, "parts": parts, "issue": issue, "reference_algorithmic_bytes": ref_alg_bytes,
                         "empirical": empirical,
                         "kernel": "walk_kernel<ENG_U8, CH=1, R=%d, G64, %s>" % (1 if ef <= 64 else (2 if ef <= 128 else (4 if ef <= 256 else 8)), visited),
                         "kernel_launches_per_step": 1 + len(cuts),
                         "walk_order": ({"cut_after_levels": cuts,
                                         "meaning": "one walk = %d launches of the kernel over consecutive level ranges; after each cut level the %d queries "
                                                    "of the step are sorted by the depth-first position of the best node they found on it and dealt to the "
                                                    "XCDs in contiguous runs (cos_index_set_walk_order; same walks, same results); avg_ms spans all of it, "
                                                    "the sort included" % (1 + len(cuts), B)} if cuts else None),
                         "aggregate": {"achieved": aggregate_gbps, "frac": aggregate_gbps / HBM_PEAK_GBPS, "in_flight": overlap,
                                       "note": "algorithmic bytes of ALL timed launches / timed wall time; exceeds the kernel-level figure "
                                               "only through launches overlapping on different streams"},
                         "per_launch": {"algorithmic_bytes": avg_bytes, "kernel_ms": kern_ms, "avg_ms": avg_ms, "min_ms": tr["walk_ms_min"], "max_ms": tr["walk_ms_max"],
                                        "launches_sampled": tr["timed_launches_sampled"], "evals": tr["evals"], "expansions": tr["expansions"],
                                        "finalize_ms": tr["finalize_ms"], "prep_ms": tr["prep_ms"], "adjacency_rounds": tr["rounds"]},
                         "note": "achieved = algorithmic bytes of one step's walk (row evaluations x (dim+4) + table evaluations x 4 + expansions x M x 4, "
                                 "counted by the kernel) / the summed HIP-event durations of the walk kernel's dispatches of a step on their own stream "
                                 "(kernel_ms; kernel_launches_per_step dispatches: a rocprofv3 trace shows them, their durations add up to kernel_ms); "
                                 "traffic = fabric-side bytes of the same dispatches from the committed PMC pass (FETCH_SIZE x 1024 x k + WRITE_SIZE x 1024, k "
                                 "calibrated on a gather with a known byte count: profiles/pmc_calibration.json); parts = the same per dispatch"},
        }
        del ix
        return rec

    def _cpu_baseline_and_parity(self, ix, ef, visited, cpu_seconds, m0=64, m=32):
        """the oracle on all usable host cores over a bounded sample of the SAME queries on the SAME graph; the same sample is the
        parity check (ids, score bits, counts).  The oracle's quantized corpus is made once per workload and takes each mode's graph."""
        from oracle import oracle as O
        env, torch = self.env, self.env.torch
        n, d, k, B = self.n, self.d, self.k, self.B
        X, Q = self.X, self.Q
        cores = effective_cores()
        if self.oracle is None:
            op = O.HNSWParams(dim=d, storage=O.STORAGE_U8, num_layers=9, ef_construction=self.ef_construction, ef_search=ef, seed=42,
                              range_lo=self.values_range[0], range_hi=self.values_range[1])
            full_raw = (n * d * 4) <= (8 << 30)      # up to 8 GB the oracle gets the raw table; beyond, rows are streamed through
            if full_raw:
                oix = O.OracleIndex(op).set_vectors(X.cpu().numpy())
            else:   # the oracle quantizes the corpus itself, chunk by chunk; the exact rerank later gets the raw rows it needs
                oix = O.OracleIndex(op).alloc_vectors(n)
                for s0 in range(0, n, 1 << 18):
                    oix.quantize_rows(s0, X[s0:s0 + (1 << 18)].cpu().numpy())
            self.oracle = (oix, full_raw)
            self.Qh = Q.cpu().numpy()
        oix, full_raw = self.oracle
        Qh = self.Qh
        oix.set_visited_mode(O.VISITED_EXACT if visited == "exact" else O.VISITED_REF)
        oix.set_ef_search(ef)
        if oix.params.level0_neighbors_count != m0:
            oix.set_level0_neighbors(m0)
        if oix.params.neighbors_count != m:
            oix.set_neighbors(m)
        oix.import_graph(ix.download_graph(), ix.download_root())
        t2 = time.perf_counter()
        pm = max(64, cores)
        if full_raw:
            oix.search_batch(Qh[:pm], k, threads=cores)
        else:
            oix.candidates_batch(Qh[:pm], k, threads=cores)
        rate = pm / (time.perf_counter() - t2)
        nq = int(min(Qh.shape[0], max(256, rate * cpu_seconds)))
        if not full_raw:   # raw rows of exactly the candidates finalize_ann_results will read for these nq queries
            cand, _ = oix.candidates_batch(Qh[:nq], k, threads=cores)
            u = np.unique(cand[cand != 0xFFFFFFFF])
            oix.set_raw_subset(u, X[torch.from_numpy(u.astype(np.int64)).to(env.dev)].cpu().numpy())
        # bounded sample: the distinct queries of the workload, repeated until about cpu_seconds of wall time are spent
        t2 = time.perf_counter()
        oids, osc, ocnt = oix.search_batch(Qh[:nq], k, threads=cores)[:3]
        first = time.perf_counter() - t2
        reps = 1 + int(max(0, min(15, round(cpu_seconds / first) - 1)))
        for _ in range(reps - 1):
            oix.search_batch(Qh[:nq], k, threads=cores)
        cpu_s = time.perf_counter() - t2
        cpu = {"value": reps * nq / cpu_s, "unit": "queries/s", "cores": cores, "kind": "port",
               "sample": f"{reps} passes over {nq} queries of the same workload ({cpu_s:.1f} s wall on {cores} usable cores of {os.cpu_count()} logical "
                         f"CPUs), oracle/ (C/AVX2 restatement of the Rust path), one OpenMP thread per core like rayon's fan-out"
                         + ("" if full_raw else "; corpus quantized by the oracle in streamed chunks")}
        # parity check on the same sample: GPU ids/scores vs oracle, bit for bit
        o_ids, o_sc, o_cnt, o_st, streams = self.o_ids, self.o_sc, self.o_cnt, self.o_st, self.streams
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
                  "count_mismatches": int((gc_ != ocnt).sum()),
                  "oracle_mode": "exact visited set (identical to the oracle in that mode, not to the Rust path)" if visited == "exact"
                                 else "reference PerformantFixedSet (the Rust path's own filter)"}
        return cpu, parity


def compact_dense_record(rec, world):
    """a dense workload's record as it appears under `configs`: the main line's fields, trimmed to what differs (the line is parsed by
    the driver from the process's stdout: the five records together stay around 10 KB)"""
    r, se, lo = rec["recall"]
    c = rec["config"]
    roof = {k: v for k, v in rec["roofline"].items() if k not in ("note", "aggregate", "empirical")}
    return {"config": {k: c[k] for k in ("workload", "n_gpus", "standard_size", "vectors_per_gpu", "dim", "queries_per_step", "launches_in_flight", "top_k", "ef_search",
                                          "ef_construction", "build_visited", "visited", "storage", "corpus", "parallelism")},
            "qps": rec["value"], "unit": "queries/s", "ms_per_step": rec["elapsed"] / rec["steps"] * 1e3,
            "steps": rec["steps"], "warmup": rec["warmup"], "dtype": "u8",
            "recall_at_10": r, "recall_stderr": se, "recall_lower95": lo, "meets_recall_target": bool(lo >= 0.95),
            "ef_selection": [[e["ef_search"], round(e["recall_at_10"], 4)] for e in rec["ef_table"]],
            "failed_queries": rec["status_bad"], "build_seconds": rec["build_s"], "seconds": rec["seconds"],
            "shard_searches_per_s": rec["value"] * world,
            "roofline": roof, "cpu_baseline": rec["cpu"], "parity_vs_oracle": rec["parity"], "result_properties": rec["props"]}


def _slim_roofline(r):
    if not r:
        return r
    keep = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "step_frac", "traffic", "traffic_GBps", "traffic_frac_of_peak", "traffic_fetch_factor_k",
                                  "kernel", "kernel_launches_per_step", "reference_algorithmic_bytes", "issue") if k in r}
    if r.get("parts"):
        keep["parts"] = {pn: ({k: v for k, v in pv.items() if k not in ("note", "kernel")} if isinstance(pv, dict) else pv) for pn, pv in r["parts"].items()}
    if r.get("walk_order"):
        keep["walk_order_cut_after_levels"] = r["walk_order"]["cut_after_levels"]
    pl = r.get("per_launch") or {}
    keep["per_launch"] = {k: pl[k] for k in ("algorithmic_bytes", "kernel_ms", "avg_ms", "gemm_ms_all_launches", "int8_ops", "launches_sampled", "finalize_ms") if k in pl}
    if r.get("empirical"):
        keep["empirical"] = {k: r["empirical"][k] for k in ("stream_read_GBps", "row_gather_GBps", "row_gather_64MB_table_GBps") if k in r["empirical"]}
    return keep


def _slim_cpu(c):
    return {k: (v if k != "sample" else v[:120]) for k, v in c.items()} if c else c


def slim_line(out):
    """the stdout line: every result of the full record, none of its prose"""
    o = dict(out)
    o["config"] = {k: (v[:200] if k == "workload" else v) for k, v in out["config"].items() if k not in ("step", "ef_policy", "exchange", "corpus", "storage", "parallelism")}
    o["config"]["storage"] = out["config"]["storage"][:48]
    o["config"]["parallelism"] = out["config"]["parallelism"]
    for k in ("value_note", "recall_sets"):
        o.pop(k, None)
    o["roofline"] = _slim_roofline(out.get("roofline"))
    o["cpu_baseline"] = _slim_cpu(out.get("cpu_baseline"))
    o["ef_selection"] = [[e["ef_search"], round(e["recall_at_10"], 4)] for e in out.get("ef_selection", [])]
    o["ef_sweep"] = [{k: e[k] for k in ("ef_search", "visited", "qps", "recall_at_10", "walk_ms")} for e in out.get("ef_sweep", [])]
    if out.get("host_api_pcie_inclusive"):
        o["host_api_pcie_inclusive"] = {k: v for k, v in out["host_api_pcie_inclusive"].items() if k != "note"}
    if out.get("flat_scan_ground_truth"):
        o["flat_scan_ground_truth"] = {k: v for k, v in out["flat_scan_ground_truth"].items() if k != "note"}
    cfgs = {}
    for name, c in (out.get("configs") or {}).items():
        if "error" in c:
            cfgs[name] = c
            continue
        cc = c.get("config", {})
        e = {"workload": cc.get("workload", "")[:110], "standard_size": c.get("standard_size", cc.get("standard_size")),
             "qps": c.get("qps"), "ms_per_step": c.get("ms_per_step"), "seconds": c.get("seconds")}
        for k in ("recall_at_10", "meets_recall_target", "build_seconds", "merged_recall_at_10", "merged_equals_merge_of_shard_answers", "shards_in_merged_answers",
                  "ms_per_step_host_api", "qps_host_api_pcie_inclusive", "ms_eight_shard_searches", "ms_exchange_plus_merge", "packed_record_bytes_per_shard"):
            if k in c:
                e[k] = c[k]
        for k in ("shards", "vectors_per_shard", "M0", "M"):
            if k in cc:
                e[k] = cc[k]
        for k in ("ef_search", "visited", "build_visited"):
            if k in cc:
                e[k] = cc[k][:40] if isinstance(cc[k], str) else cc[k]
        e["roofline"] = _slim_roofline(c.get("roofline"))
        e["cpu_baseline"] = _slim_cpu(c.get("cpu_baseline"))
        e["parity_vs_oracle"] = {k: v for k, v in (c.get("parity_vs_oracle") or {}).items() if k not in ("oracle_mode", "checked", "collection")} or None
        if c.get("hnsw_walk_quaternary"):
            w = c["hnsw_walk_quaternary"]
            e["hnsw_walk_quaternary"] = {k: w[k] for k in ("n", "build_s", "qps", "ms_per_launch", "recall_at_10_vs_f32_bruteforce", "parity_vs_oracle") if k in w}
            e["hnsw_walk_quaternary"]["roofline"] = _slim_roofline(w.get("roofline"))
            e["hnsw_walk_quaternary"]["cpu_baseline"] = _slim_cpu(w.get("cpu_baseline"))
        for k in ("single_batch_qps", "single_batch_qps_one_wave_latency_kernel", "single_batch_qps_throughput_kernel",
                  "single_batch_latency_walk_identical_to_throughput_walk", "value_at_query_batch_256", "launch_size_sweep", "recall_lower95"):
            if c.get(k) is not None:
                e[k] = c[k]
        if c.get("ef_sweep"):
            e["ef_sweep"] = [{k: x[k] for k in ("ef_search", "visited", "qps", "recall_at_10", "walk_ms")} for x in c["ef_sweep"]]
        if c.get("host_api_pcie_inclusive"):
            e["host_api_pcie_inclusive"] = {k: v for k, v in c["host_api_pcie_inclusive"].items() if k not in ("note", "concurrent_256_query_callers")}
        if c.get("same_graph_exact_visited_set"):
            e["same_graph_exact_visited_set"] = [{k: x[k] for k in ("ef_search", "qps", "recall_at_10")} for x in c["same_graph_exact_visited_set"]]
        cfgs[name] = e
    o["configs"] = cfgs
    return o


def rounded(obj, digits=9):
    """floats of the output line at `digits` significant digits (json.dumps prints 17): a third of the line's bytes"""
    if isinstance(obj, float):
        return float(f"{obj:.{digits}g}") if obj == obj and abs(obj) != float("inf") else obj
    if isinstance(obj, dict):
        return {k: rounded(v, digits) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [rounded(v, digits) for v in obj]
    return obj


# ------------------------------------------------------------------------------------------------------------------
# watchdog: the main line must come out even if an extra config hangs
# ------------------------------------------------------------------------------------------------------------------
class Emitter:
    """owns the saved stdout descriptor; prints the ONE JSON line exactly once — at the normal end, or from the deadline timer with
    what has been measured so far (then the process exits hard: a hung HIP / RCCL call cannot be interrupted from Python)"""

    def __init__(self, json_fd, rank, deadline_s):
        self.fd, self.rank, self.lock, self.done, self.out = json_fd, rank, threading.Lock(), False, None
        self.timer = None
        if deadline_s > 0:
            self.timer = threading.Timer(deadline_s, self._deadline)
            self.timer.daemon = True
            self.timer.start()

    def _write(self):
        if self.rank == 0 and self.out is not None:
            full = rounded(self.out)
            # the complete record (every per-launch figure, ef tables, notes) goes to a file next to the other run artefacts; the ONE
            # line on stdout carries the same results without the prose, so that it stays a few KB whatever reads it
            path = None
            name = os.environ.get("COS_BENCH_FULL_RECORD", "bench_full_record.json")   # a script that runs bench.py several times names each record
            for d in (os.path.join(ROOT, "gpurun_out"), ROOT):
                try:
                    os.makedirs(d, exist_ok=True)
                    path = os.path.join(d, os.path.basename(name))
                    with open(path, "w") as fh:
                        json.dump(full, fh)
                    break
                except OSError:
                    path = None
            line = slim_line(full)
            line["full_record_file"] = os.path.relpath(path, ROOT) if path else None
            os.write(self.fd, (json.dumps(line) + "\n").encode())

    def _deadline(self):
        with self.lock:
            if self.done:
                return
            self.done = True
            if self.out is not None:
                self.out["configs_incomplete"] = "deadline reached before every extra config finished; records present are complete"
            self._write()
        os._exit(0 if self.out is not None else 3)

    def finish(self):
        with self.lock:
            if self.done:
                return
            self.done = True
            if self.timer is not None:
                self.timer.cancel()
            self._write()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks = GPUs of this node; > 1 without WORLD_SIZE in the environment starts them (torchrun)")
    ap.add_argument("--steps", type=int, default=24, help="timed steps; one step = one coalesced launch of --coalesce x --batch queries")
    ap.add_argument("--warmup", type=int, default=4, help="untimed warm-up steps (launches)")
    ap.add_argument("--workload", default=MAIN_WORKLOAD, choices=sorted(WORKLOADS),
                    help="the workload `value` is measured on; default = one 12.5M x 1024 shard of BASELINE configs[3] per GPU (the metric's own configuration)")
    ap.add_argument("--configs", default="auto", help="extra BASELINE configs measured after the main workload and appended to the same JSON "
                    "line: auto | all | none | comma list of " + ",".join(ALL_CONFIGS))
    ap.add_argument("--n", type=int, default=0, help="override vectors per GPU of the main workload (marks the run as non-standard)")
    ap.add_argument("--config-scale", type=float, default=1.0, help="shrink every EXTRA config's corpus by this factor (plumbing runs; marked non-standard)")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--coalesce", type=int, default=128, help="client batches fused per launch (dynamic batching); 128 x 256 = 32768 queries "
                    "refill the chip's 7168 wave slots several times over, so one launch runs the walk kernel at 0.9 of the HBM roof "
                    "(8192-query launches: 0.71 — every wave is resident at once and the chip drains as they finish)")
    ap.add_argument("--inflight", type=int, default=0, help="launches kept in flight (HIP streams); 0 = auto: 2, or 3 when every launch is followed by "
                    "the exchange step (N > 1): the all-gather waits for a finalize that shares the HBM with the next walk, a third launch keeps "
                    "the walk chain fed meanwhile (forced-dist run: 0.96 of the non-dist rate against 0.91 with two)")
    ap.add_argument("--ef", default="auto", help="ef_search: an integer, or 'auto' = smallest of 32,48,64,80,96,112,128,160,192,256,384,512 whose recall@10 "
                    "on the SELECTION query set clears --recall-target with 95 %% confidence (the metric is QPS AT recall@10 >= 0.95); "
                    "config.toml default is 256")
    ap.add_argument("--recall-target", type=float, default=0.95)
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--visited", default="ref", choices=["ref", "exact"],
                    help="search-time visited filter: ref = PerformantFixedSet replica (ID parity with the reference), "
                         "exact = exact visited set (recall mode, not ID-identical)")
    ap.add_argument("--ef-construction", type=int, default=0, help="0 = the workload's default (config.toml 128; c4shard 256)")
    ap.add_argument("--build-visited", default="ref", choices=["ref", "exact"], help="visited filter used by the builder's walks")
    ap.add_argument("--quantization", default="auto", choices=["auto", "range11"], help="auto = sampled values_range; range11 = (-1,1)")
    ap.add_argument("--ef-sweep", default="auto",
                    help="comma list of extra settings timed after the main run on the same graph: '256' = ef_search 256 (config.toml "
                         "default) with the main visited filter; 'exact:32' / 'ref:128' = that ef with the named visited filter; auto = none for "
                         "the 51 GB shard, '256,exact:32,exact:64' for c2 (as a main workload and as the record under `configs`)")
    ap.add_argument("--build-batch", type=int, default=4096)
    ap.add_argument("--m0", type=int, default=0, help="level_0_neighbors_count of the MAIN workload's graph; 0 = the workload's own: 256 for the "
                    "12.5M x 1024 shard (the setting that meets the recall target with the reference's visited filter), 64 = config.toml otherwise")
    ap.add_argument("--m", type=int, default=0, help="neighbors_count of the main workload's graph; 0 = the workload's own (64 for the shard, 32 otherwise)")
    ap.add_argument("--recall-queries", type=int, default=8192, help="size of EACH of the two disjoint recall query sets (selection / report)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--append-rows", type=int, default=100_000, help="vectors the append probe inserts into the main workload's resident graph after "
                    "everything else was measured (cos_index_append; N = 1 only)")
    ap.add_argument("--no-append-probe", action="store_true")
    ap.add_argument("--no-host-api", action="store_true", help="skip the host-buffer API legs (cos_search_batch from 1-256 host threads); the profiler "
                                                                "passes of scripts/final_profile*.sh use it: rocprofv3 does not survive a few thousand short-lived native threads")
    ap.add_argument("--no-hbm-probe", action="store_true", help="skip the empirical HBM ceiling probes (cos_hbm_probe)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--config-cpu-seconds", type=float, default=5.0, help="CPU-baseline sample of each extra config")
    ap.add_argument("--exchange", default="auto", choices=["auto", "shardset", "torch"],
                    help="N > 1 exchange step: shardset = cos_shardset_* (RCCL all-gather + merge inside the C ABI); torch = "
                         "torch.distributed all_gather + cos_merge_topk_packed_device; auto = shardset, falling back loudly to torch")
    ap.add_argument("--deadline-seconds", type=float, default=1500.0, help="print what has been measured and exit if the run is still "
                    "going after this long (the main line is never lost to a hung extra config); 0 = off")
    ap.add_argument("--single-process", action="store_true", help="N shards in ONE host process through cos_shardset_search_batch "
                    "(the reference's deployment: one process, rayon fan-out; indexes/mod.rs:260-272) instead of one process per GPU")
    ap.add_argument("--allow-shared-devices", action="store_true", help="--single-process: place several shards on one device if fewer "
                    "than --gpus are visible (code-path check on a 1-GPU box; NOT a scaling number)")
    ap.add_argument("--launcher-selftest", action="store_true", help="CPU/gloo check of the launch + rank plumbing, no search")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1 and not args.single_process:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus and not args.single_process:
        raise SystemExit(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={env_world} set by the launcher; refusing to report either")
    if args.launcher_selftest:
        return launcher_selftest(args)
    if args.single_process:
        from scripts import bench_single_process
        return bench_single_process.run(args, METRIC)

    import torch  # noqa: F401  (imported before the HIP library: one runtime)
    env = Env(args)
    rank, world = env.rank, env.world
    # stdout carries exactly ONE JSON line: libraries that print banners through C stdio (RCCL prints its version block on
    # fd 1 at exit) are pointed at stderr, the JSON line is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    emitter = Emitter(json_fd, rank, args.deadline_seconds)
    t_setup = time.time()

    # ---- the main workload: `value` -------------------------------------------------------------------------------
    C2_SWEEP = "256,exact:32,exact:64"
    is_shard = args.workload == "c4shard"
    m0 = args.m0 or (MAIN_M0 if is_shard else 64)
    mm = args.m or (MAIN_M if is_shard else 32)
    main_sweep = ("" if is_shard else C2_SWEEP) if args.ef_sweep == "auto" else args.ef_sweep
    wl = DenseWorkload(env, args.workload, n_override=args.n, ef_construction=args.ef_construction, quantization=args.quantization,
                       build_batch=args.build_batch, append_rows=0 if (args.no_append_probe or env.world > 1) else args.append_rows)
    # the host-buffer legs (cos_search_batch from 1-256 host threads: BASELINE configs[1]'s "query-batch=256") belong to c2, as a main
    # workload or as the record under `configs`
    rec = wl.run_mode(args.build_visited, args.visited, ef_arg=args.ef, ef_sweep=main_sweep, cpu_seconds=args.cpu_seconds,
                      single_batch=True, host_api=(not args.no_host_api) and not is_shard, hbm_probe=not args.no_hbm_probe, exchange=args.exchange,
                      m0=m0, m_upper=mm, append_probe=not args.no_append_probe)
    flat = wl.flat
    n = wl.n
    c4 = wl if is_shard else None      # the 51 GB shard, its ground truth and the oracle's quantized copy serve the 8-shard record too
    if c4 is None:
        wl.close()
    del wl
    recall, recall_se, recall_lo = rec["recall"]
    merged_qps = rec["value"]

    def serial_block(r):
        return {"single_batch_qps": r["serial"]["qps"], "single_batch_qps_one_wave_latency_kernel": r["serial"]["qps_one_wave_latency_kernel"],
                "single_batch_qps_throughput_kernel": r["serial"]["qps_throughput_kernel"],
                "single_batch_latency_walk_identical_to_throughput_walk": r["serial"]["identical"]}
    out = {
        "metric": METRIC, "value": merged_qps, "unit": "queries/s", "n_gpus": world, "steps": rec["steps"], "warmup": rec["warmup"],
        "ms_per_step": rec["elapsed"] / rec["steps"] * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": rec["config"],
        "recall_at_10": recall, "recall_stderr": recall_se, "recall_lower95": recall_lo, "recall_queries": args.recall_queries,
        "meets_recall_target": bool(recall_lo >= args.recall_target),
        "recall_sets": "ef selected on query seed 44, recall reported on query seed 45 (disjoint draws of the same corpus distribution); "
                       "ground truth = exact brute-force cosine top-10 over the GLOBAL corpus (every shard's scan, merged)",
        "failed_queries": rec["status_bad"],
        "merged_qps": merged_qps, "shard_searches_per_s": merged_qps * world, "global_corpus_vectors": n * world,
        "value_note": ("n_gpus == 1: value = queries/s over the whole corpus" if world == 1 else
                       "value = merged answers/s over the GLOBAL corpus (n_gpus x vectors_per_gpu; every query is searched on every "
                       "shard, then all-gathered and merged); shard_searches_per_s = value x n_gpus is the shard-level work"),
        **serial_block(rec),
        "ef_selection": rec["ef_table"], "ef_sweep": rec["sweep"], "launch_size_sweep": rec["size_sweep"], "build_seconds": rec["build_s"],
        "setup_seconds": time.time() - t_setup,
        "roofline": rec["roofline"],
        "flat_scan_ground_truth": flat, "result_properties": rec["props"], "cpu_baseline": rec["cpu"], "parity_vs_oracle": rec["parity"],
        "host_api_pcie_inclusive": rec["host_api"],
        "append_and_delete_on_the_resident_graph": rec["append"],
        # BASELINE configs[1] says "query-batch=256": the rate with the reference's own calling pattern — concurrent synchronous callers of
        # ONE 256-query batch each through cos_search_batch on host buffers (PCIe-inclusive), fused by the library's dynamic batching;
        # measured on c2 (here when c2 is the main workload, else inside configs.c2)
        "value_at_query_batch_256": value_at_batch_256(rec["host_api"]),
        # every COS_* variable of the process: the library's experiment switches must be visible in the record they shaped
        "env_overrides": {k: v for k, v in sorted(os.environ.items()) if k.startswith("COS_") and k != "COS_BENCH_FULL_RECORD"},
        "configs": {},
    }
    emitter.out = out

    # ---- the remaining BASELINE configs, each to the same parity + roofline + cpu_baseline bar ------------------------
    todo = resolve_configs(args.configs, world, args.workload)
    scale = args.config_scale
    for name in todo:
        t_c = time.time()
        try:
            if not (name.startswith("c4shard_") or name == "c4_8shards_one_device") and c4 is not None:
                c4.close()       # the single-GPU configs below do not share the shard: its 51 GB go back first
                c4 = None
                env.torch.cuda.empty_cache()
            if name == "c2":
                w2 = DenseWorkload(env, "c2", n_override=0 if scale == 1.0 else int(1_000_000 * scale), quantization="auto")
                r2 = w2.run_mode("ref", "ref", ef_sweep=C2_SWEEP, cpu_seconds=args.config_cpu_seconds, single_batch=True,
                                 host_api=not args.no_host_api, exchange=args.exchange)
                c2r = compact_dense_record(r2, world)
                c2r.update(serial_block(r2))
                c2r["ef_sweep"] = r2["sweep"]
                c2r["launch_size_sweep"] = r2["size_sweep"]
                c2r["host_api_pcie_inclusive"] = r2["host_api"]
                c2r["value_at_query_batch_256"] = value_at_batch_256(r2["host_api"])
                out["configs"][name] = c2r
                if out["value_at_query_batch_256"] is None:
                    out["value_at_query_batch_256"] = dict(c2r["value_at_query_batch_256"] or {}, measured_on="configs.c2 (BASELINE configs[1]: 1M x 768, query-batch 256)") or None
                w2.close()
                del w2
            elif name == "c2_sigma01":
                w2 = DenseWorkload(env, "c2_sigma01", n_override=0 if scale == 1.0 else int(1_000_000 * scale), quantization="auto")
                r2 = w2.run_mode("ref", "ref", ef_sweep="exact:256", cpu_seconds=args.config_cpu_seconds, exchange=args.exchange)
                out["configs"][name] = compact_dense_record(r2, world)
                out["configs"][name]["same_graph_exact_visited_set"] = r2["sweep"]
                out["configs"][name]["note"] = ("SURVEY 8(d)'s suggested clustered set (sigma 0.1 per component): beside `value`'s and c2's sigma 0.8/sqrt(d) mixture; "
                                                "what the reference's algorithm reaches on it with its default hyper-parameters, bit-identical to the oracle")
                w2.close()
                del w2
            elif name == "c2_uniform":
                w2 = DenseWorkload(env, "c2_uniform", n_override=0 if scale == 1.0 else int(1_000_000 * scale), quantization="auto")
                r2 = w2.run_mode("ref", "ref", ef_sweep="exact:512", cpu_seconds=args.config_cpu_seconds, exchange=args.exchange)
                out["configs"][name] = compact_dense_record(r2, world)
                out["configs"][name]["same_graph_exact_visited_set"] = r2["sweep"]
                out["configs"][name]["note"] = ("uniform(-1,1) components in 768 dimensions have no neighbourhood structure (every cosine is 0 +- 0.036): "
                                                "no ef reaches the recall target; the record shows what the reference's algorithm returns on such data "
                                                "(bit-identical to the oracle) and what the exact visited set changes")
                w2.close()
                del w2
            elif name.startswith("c4shard_"):
                if c4 is None:   # every mode shares the 51 GB shard, its ground truth and the oracle's quantized copy
                    c4 = DenseWorkload(env, "c4shard", n_override=0 if scale == 1.0 else int(12_500_000 * scale))
                v = "exact" if name == "c4shard_exact" else "ref"
                toks = name.split("_")
                cm0 = int(toks[toks.index("m0") + 1]) if "m0" in toks else 64
                cmm = int(toks[toks.index("m") + 1]) if "m" in toks else 32
                r4 = c4.run_mode(v, v, cpu_seconds=args.config_cpu_seconds, exchange=args.exchange, m0=cm0, m_upper=cmm)
                out["configs"][name] = compact_dense_record(r4, world)
            elif name == "c4_8shards_one_device":
                # the same 12.5M x 1024 corpus as eight id-range shards behind cos_shardset_search_batch: merged recall, exchange + merge cost
                from scripts import bench_c4_8shards
                if c4 is None:
                    c4 = DenseWorkload(env, "c4shard", n_override=0 if scale == 1.0 else int(12_500_000 * scale))
                if c4.gt_rep is None:   # (the main workload ran first in the default order: its ground truth is reused)
                    hp0 = ca_mod().HNSWHyperParams(num_layers=9, ef_construction=c4.ef_construction, ef_search=64)
                    ix0 = ca_mod().HNSWIndex(c4.d, hp0, ca_mod().DistanceMetric.Cosine, ca_mod().StorageType.UnsignedByte(), c4.values_range,
                                             device=env.local_rank, seed=42)
                    ix0.upload_vectors_device(c4.X.data_ptr(), c4.n, keepalive=c4.X)
                    c4._ground_truth(ix0)
                    del ix0
                out["configs"][name] = bench_c4_8shards.run(c4)
            elif name == "c3":
                from scripts import bench_c3
                out["configs"][name] = bench_c3.run(n=int(10_000_000 * scale), cpu_seconds=0.0 if args.no_cpu_baseline else args.config_cpu_seconds,
                                                    device=env.local_rank, walk_n=int(1_000_000 * scale))
            elif name == "c5":
                from scripts import bench_c5
                out["configs"][name] = bench_c5.run(n=int(1_000_000 * scale), device=env.local_rank,
                                                    cpu_seconds=0.0 if args.no_cpu_baseline else args.config_cpu_seconds)
        except Exception as exc:  # noqa: BLE001 — an extra config must never cost the main line; the failure is in the record
            import traceback
            traceback.print_exc()
            out["configs"][name] = {"error": f"{type(exc).__name__}: {exc}"}
            if world > 1:        # a rank-local failure would leave the others inside a collective: stop the extras on every rank
                break
        if isinstance(out["configs"].get(name), dict):
            out["configs"][name].setdefault("seconds", time.time() - t_c)
            if scale != 1.0:
                out["configs"][name]["standard_size"] = False
        env.torch.cuda.empty_cache()
    if c4 is not None:
        c4.close()
    out["total_seconds"] = time.time() - t_setup
    emitter.finish()
    if env.dist_on:
        env.dist.destroy_process_group()


if __name__ == "__main__":
    main()
