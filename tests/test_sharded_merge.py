"""N>1 path on CPU: world_size-2 gloo processes, each owning an ID-range shard (oracle search stands in for the
per-shard GPU search), the all-gather exchange of cosdata_amd/sharding.py, and the merge rule.  The merged result
must equal the exact top-k over the union of the shards' candidates, and — because every shard reranks with exact
f32 cosine — must be reproducible by a single process running both shards."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _merge_numpy(g_ids, g_sc, g_cnt, k):
    """checker for the merge rule: total_cmp desc, larger id first."""
    S, B, _ = g_ids.shape
    out_ids = np.full((B, k), 0xFFFFFFFF, np.uint32)
    out_sc = np.zeros((B, k), np.float32)
    out_cnt = np.zeros(B, np.uint32)
    for b in range(B):
        ent = [(float(g_sc[s, b, j]), int(g_ids[s, b, j])) for s in range(S) for j in range(int(g_cnt[s, b]))]
        ent.sort(key=lambda t: (t[0], t[1]), reverse=True)
        ent = ent[:k]
        out_cnt[b] = len(ent)
        for j, (sc, i) in enumerate(ent):
            out_ids[b, j], out_sc[b, j] = i, sc
    return out_ids, out_sc, out_cnt


def _shard_search(X, Q, lo, hi, k):
    from oracle import oracle as O
    p = O.HNSWParams(dim=X.shape[1], num_layers=4, ef_construction=48, ef_search=48, seed=11 + lo)
    ix = O.OracleIndex(p).set_vectors(np.ascontiguousarray(X[lo:hi])).build()
    ids, sc, cnt = ix.search_batch(Q, k)[:3]
    gids = np.where(np.arange(k)[None, :] < cnt[:, None], ids + lo, 0xFFFFFFFF).astype(np.uint32)  # id_base = lo
    return gids, sc, cnt


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cosdata_amd.sharding import allgather_packed, allgather_topk, packed_views, packed_words, shard_range
    rng = np.random.default_rng(123)
    X = rng.uniform(-1, 1, (1200, 40)).astype(np.float32)
    Q = (X[rng.integers(0, 1200, 16)] + 0.02 * rng.standard_normal((16, 40))).astype(np.float32)
    k = 6
    lo, hi = shard_range(1200, world, rank)
    gids, sc, cnt = _shard_search(X, Q, lo, hi, k)
    g_ids, g_sc, g_cnt = allgather_topk(torch.from_numpy(gids.view(np.int32)), torch.from_numpy(sc), torch.from_numpy(cnt.view(np.int32)))
    assert g_ids.shape == (world, 16, k)
    # packed exchange (what bench.py uses): ONE collective must carry exactly the same data
    rec = torch.zeros(packed_words(16, k), dtype=torch.int32)
    p_ids, p_sc, p_cnt = packed_views(rec, 16, k)
    p_ids.copy_(torch.from_numpy(gids.view(np.int32))); p_sc.copy_(torch.from_numpy(sc)); p_cnt.copy_(torch.from_numpy(cnt.view(np.int32)))
    g_rec = allgather_packed(rec)
    assert g_rec.shape == (world, packed_words(16, k))
    for s_ in range(world):
        u_ids, u_sc, u_cnt = packed_views(g_rec[s_].contiguous(), 16, k)
        assert torch.equal(u_ids, g_ids[s_]) and torch.equal(u_sc.view(torch.int32), g_sc[s_].view(torch.int32)) and torch.equal(u_cnt, g_cnt[s_])
    np.save(os.path.join(tmp, f"gathered_{rank}.npy"), g_ids.numpy())
    m_ids, m_sc, m_cnt = _merge_numpy(g_ids.numpy().view(np.uint32), g_sc.numpy(), g_cnt.numpy().view(np.uint32), k)
    np.savez(os.path.join(tmp, f"merged_{rank}.npz"), ids=m_ids, sc=m_sc, cnt=m_cnt)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_allgather_merge_gloo(tmp_path, world):
    """world 2, and world 8 = the shape of BASELINE configs[3]: eight ranks, each an id-range shard, ONE packed all-gather, the
    merge rule on every rank — against the single-process restatement of the same scheme"""
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "merged_0.npz"), np.load(tmp_path / "merged_1.npz")
    assert np.array_equal(a["ids"], b["ids"]) and np.array_equal(a["sc"], b["sc"])   # every rank ends with the same answer
    assert np.array_equal(np.load(tmp_path / "gathered_0.npy"), np.load(tmp_path / "gathered_1.npy"))
    for r in range(2, world):
        c = np.load(tmp_path / f"merged_{r}.npz")
        assert np.array_equal(a["ids"], c["ids"]) and np.array_equal(a["sc"], c["sc"])
    # single-process restatement of the same scheme
    sys.path.insert(0, ROOT)
    from cosdata_amd.sharding import shard_range
    rng = np.random.default_rng(123)
    X = rng.uniform(-1, 1, (1200, 40)).astype(np.float32)
    Q = (X[rng.integers(0, 1200, 16)] + 0.02 * rng.standard_normal((16, 40))).astype(np.float32)
    k = 6
    parts = [_shard_search(X, Q, *shard_range(1200, world, r), k) for r in range(world)]
    g_ids = np.stack([p[0] for p in parts]); g_sc = np.stack([p[1] for p in parts]); g_cnt = np.stack([p[2] for p in parts])
    m_ids, m_sc, m_cnt = _merge_numpy(g_ids, g_sc, g_cnt, k)
    assert np.array_equal(m_ids, a["ids"]) and np.array_equal(m_sc.view(np.uint32), a["sc"].view(np.uint32))
    # ids are global and each comes from the shard that owns it
    lo1, _ = shard_range(1200, world, 1)
    assert ((a["ids"] < 1200) | (a["ids"] == 0xFFFFFFFF)).all() and (a["ids"] >= lo1).any() and (a["ids"] < lo1).any()
    owners = {int(np.searchsorted([shard_range(1200, world, r)[1] for r in range(world)], i, side="right")) for i in a["ids"].ravel() if i != 0xFFFFFFFF}
    assert owners == set(range(world))                   # every shard contributes to somebody's merged list
    # quality: merged result vs exact brute force over the whole corpus
    from oracle import oracle as O
    gt, _ = O.bruteforce_topk(X, Q, k)
    recall = np.mean([len(set(a["ids"][i]) & set(gt[i])) / k for i in range(16)])
    assert recall >= 0.85


def test_shard_range_partition():
    from cosdata_amd.sharding import shard_range
    for n, w in [(100, 8), (12_500_000 * 8, 8), (7, 2)]:
        r = [shard_range(n, w, i) for i in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))


@pytest.mark.gpu
def test_merge_kernel_matches_rule():
    import cosdata_amd  # noqa: F401
    from cosdata_amd.sharding import merge_topk_device, merge_topk_packed_device
    rng = np.random.default_rng(5)
    for S, B, k in [(2, 33, 10), (8, 256, 10), (4, 7, 50), (8, 5, 100)]:
        sc = rng.choice(np.linspace(0.1, 0.9, 23).astype(np.float32), (S, B, k))  # many ties
        sc = -np.sort(-sc, axis=2)
        ids = rng.permutation(S * B * k).astype(np.uint32).reshape(S, B, k)
        cnt = rng.integers(0, k + 1, (S, B)).astype(np.uint32)
        exp = _merge_numpy(ids, sc, cnt, k)
        dev = torch.device("cuda:0")
        t = lambda a, dt: torch.from_numpy(a.view(dt) if a.dtype != np.float32 else a).to(dev)
        o_i = torch.zeros(B, k, dtype=torch.int32, device=dev); o_s = torch.zeros(B, k, dtype=torch.float32, device=dev)
        o_c = torch.zeros(B, dtype=torch.int32, device=dev)
        merge_topk_device(t(ids, np.int32), t(sc, np.float32), t(cnt, np.int32), o_i, o_s, o_c, 0, 0)
        torch.cuda.synchronize()
        # packed per-shard records [ids | scores | counts] through cos_merge_topk_packed_device
        packed = np.concatenate([ids.reshape(S, -1).view(np.int32), sc.reshape(S, -1).view(np.int32), cnt.view(np.int32)], axis=1)
        p_i = torch.zeros_like(o_i); p_s = torch.zeros_like(o_s); p_c = torch.zeros_like(o_c)
        merge_topk_packed_device(torch.from_numpy(np.ascontiguousarray(packed)).to(dev), B, k, p_i, p_s, p_c, 0, 0)
        torch.cuda.synchronize()
        for (r_i, r_s, r_c) in [(o_i, o_s, o_c), (p_i, p_s, p_c)]:
            gc = r_c.cpu().numpy().view(np.uint32)
            assert np.array_equal(gc, exp[2])
            for b in range(B):
                c = int(gc[b])
                assert np.array_equal(r_i.cpu().numpy().view(np.uint32)[b, :c], exp[0][b, :c])
                assert np.array_equal(r_s.cpu().numpy()[b, :c], exp[1][b, :c])


def _recall_worker(rank, world, port, tmp):
    """bench.py's recall bookkeeping for N > 1, on CPU: per-shard exact top-k (global ids) -> global_topk_by_score ->
    every rank holds the same global ground truth, equal to the single-process brute force over the whole corpus."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cosdata_amd.sharding import global_topk_by_score, shard_range
    from oracle import oracle as O
    rng = np.random.default_rng(77)
    n, d, nq, k = 900, 24, 10, 5
    X = rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    lo, hi = shard_range(n, world, rank)
    lid, _ = O.bruteforce_topk(np.ascontiguousarray(X[lo:hi]), Q, k)
    gids = torch.from_numpy(lid.astype(np.int64) + lo)
    Xt, Qt = torch.from_numpy(X), torch.from_numpy(Q)
    sims = torch.einsum("qd,qkd->qk", Qt, Xt[gids])           # what bench.py::merge_global computes
    merged = global_topk_by_score(gids, sims, k)
    np.save(os.path.join(tmp, f"gt_{rank}.npy"), merged.numpy())
    r = torch.tensor([0.5 + rank])                            # the ef decision is broadcast from rank 0
    dist.broadcast(r, 0)
    assert float(r.item()) == 0.5
    t = torch.tensor([1.0 + rank], dtype=torch.float64)       # timed region: MAX over ranks
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == float(world)
    dist.barrier()
    dist.destroy_process_group()


def test_global_ground_truth_merge_gloo(tmp_path):
    world = 2
    mp.spawn(_recall_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = np.load(tmp_path / "gt_0.npy"), np.load(tmp_path / "gt_1.npy")
    assert np.array_equal(a, b)
    from oracle import oracle as O
    rng = np.random.default_rng(77)
    X = rng.standard_normal((900, 24)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    Q = rng.standard_normal((10, 24)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    gt, _ = O.bruteforce_topk(X, Q, 5)
    assert all(set(a[i].tolist()) == set(gt[i].tolist()) for i in range(10))
