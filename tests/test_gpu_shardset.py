"""Sharded serving behind the C ABI (cos_shardset_*): S = 2 shards on one device must answer, bit for bit, like the oracle
running the same 2-shard scheme followed by the merge rule; the RCCL call path (dlopen, communicator, ncclAllGather) is
exercised with a world of one; the process-per-GPU exchange entry point is checked against the host-API result."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H
from tests.test_sharded_merge import _merge_numpy

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _two_shards(dim=1024, n1=2500, n2=2100, storage=O.STORAGE_U8, res=0):
    X = H.clustered_corpus(n1 + n2, dim, n_centers=12, sigma=0.03, seed=61)
    hp = dict(num_layers=4, ef_construction=48, ef_search=64)
    shards, oracles = [], []
    for base, Xs in ((0, X[:n1]), (n1, X[n1:])):
        Xs = np.ascontiguousarray(Xs)
        oix = H.oracle_index(Xs, storage, res, seed=3 + base, **hp)
        shards.append(H.device_index_from_oracle(oix, Xs, id_base=base))
        oracles.append((base, oix))
    return X, shards, oracles


def _oracle_two_shard(oracles, Q, k):
    parts = []
    for base, oix in oracles:
        oi, osc, oc = oix.search_batch(Q, k, threads=4)[:3]
        parts.append((np.where(np.arange(k)[None, :] < oc[:, None], oi + base, oi).astype(np.uint32), osc, oc))
    return _merge_numpy(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]), np.stack([p[2] for p in parts]), k)


@pytest.mark.parametrize("storage,res", [(O.STORAGE_U8, 0), (O.STORAGE_F32, 0)])
def test_shardset_two_shards_one_device_match_oracle(storage, res):
    from cosdata_amd.shardset import ShardSet
    X, shards, oracles = _two_shards(storage=storage, res=res)
    ss = ShardSet(shards)
    for B, k in ((40, 10), (3, 25), (1, 1)):
        Q = H.queries_from(X, B, noise=0.004, seed=B)
        ids, sc, cnt = ss.batch_search(Q, k)
        e_i, e_s, e_c = _oracle_two_shard(oracles, Q, k)
        assert np.array_equal(cnt, e_c)
        for b in range(B):
            c = int(cnt[b])
            assert np.array_equal(ids[b, :c], e_i[b, :c]), (b, ids[b, :c], e_i[b, :c])
            assert np.array_equal(sc[b, :c].view(np.uint32), e_s[b, :c].view(np.uint32))
    ss.close()


def test_shardset_eight_shards_one_device_match_oracle():
    """S = 8 id-range shards of one corpus (ragged sizes, sequential ids like collection.rs:451-468) on one device: the merged answer
    of cos_shardset_search_batch is, bit for bit, the oracle's per-shard searches followed by the merge rule — the shape of
    BASELINE configs[3] (8 shards, one exchange, one merge) at a size the oracle finishes in seconds"""
    from cosdata_amd.shardset import ShardSet
    dim, sizes = 256, [700, 650, 720, 600, 690, 710, 640, 705]
    X = H.clustered_corpus(sum(sizes), dim, n_centers=20, sigma=0.03, seed=88)
    hp = dict(num_layers=3, ef_construction=40, ef_search=48)
    shards, oracles, base = [], [], 0
    for s, n_s in enumerate(sizes):
        Xs = np.ascontiguousarray(X[base:base + n_s])
        oix = H.oracle_index(Xs, O.STORAGE_U8, 0, seed=11 + s, **hp)
        shards.append(H.device_index_from_oracle(oix, Xs, id_base=base))
        oracles.append((base, oix))
        base += n_s
    ss = ShardSet(shards)
    for B, k in ((300, 10), (5, 20)):
        Q = H.queries_from(X, B, noise=0.004, seed=100 + B)
        ids, sc, cnt = ss.batch_search(Q, k)
        e_i, e_s, e_c = _oracle_two_shard(oracles, Q, k)
        assert np.array_equal(cnt, e_c)
        owners = set()
        for b in range(B):
            c = int(cnt[b])
            assert np.array_equal(ids[b, :c], e_i[b, :c]), (b, ids[b, :c], e_i[b, :c])
            assert np.array_equal(sc[b, :c].view(np.uint32), e_s[b, :c].view(np.uint32))
            owners.update(int(np.searchsorted(np.cumsum(sizes), i, side="right")) for i in ids[b, :c])
        if B >= 100:
            assert len(owners) == 8          # every shard contributes to somebody's merged list
    ss.close()


def test_shardset_error_propagates():
    import cosdata_amd as ca
    from cosdata_amd.shardset import ShardSet
    X, shards, _ = _two_shards(dim=96, n1=500, n2=400)
    ss = ShardSet(shards)
    Q = H.queries_from(X, 4)
    Q[1] = -1.0   # zero quantized norm -> CalculationError on every shard
    with pytest.raises(ca.CosdataError) as ei:
        ss.batch_search(Q, 5)
    assert ei.value.status == 2
    with pytest.raises(ca.CosdataError) as ei:
        ss.batch_search(np.zeros((2, 95), np.float32), 5)
    assert ei.value.status == 3


_RCCL_CHILD = r"""
import os, sys
sys.path.insert(0, %(root)r)
os.environ["COS_TUNING"] = "shardset_force_rccl=1"
import numpy as np, torch
from oracle import oracle as O
from tests import helpers as H
from cosdata_amd.shardset import ShardSet, ProcessShardSet
from cosdata_amd.sharding import packed_views, packed_words
X = H.clustered_corpus(1500, 128, n_centers=8, seed=2)
oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=3, ef_construction=32, ef_search=48)
dix = H.device_index_from_oracle(oix, X)
Q = H.queries_from(X, 32, seed=9)
want = oix.search_batch(Q, 10, threads=2)[:3]
ss = ShardSet([dix])                      # world of one, RCCL forced: ncclCommInitAll + grouped ncclAllGather
got = ss.batch_search(Q, 10)
assert all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(got, want)), "single-process RCCL path differs"
ss.close()
ps = ProcessShardSet(dix, 0, 1, 0)        # process-per-GPU entry point, same world of one
dev = torch.device("cuda:0")
B, k = 32, 10
rec = torch.zeros(packed_words(B, k), dtype=torch.int32, device=dev)
ids, sc, cnt = packed_views(rec, B, k)
st = torch.zeros(B, dtype=torch.int32, device=dev)
g = torch.zeros(packed_words(B, k), dtype=torch.int32, device=dev)
mi = torch.zeros(B, k, dtype=torch.int32, device=dev); ms = torch.zeros(B, k, device=dev); mc = torch.zeros(B, dtype=torch.int32, device=dev)
s = torch.cuda.Stream(device=dev)
dix.batch_search_device(torch.from_numpy(Q).to(dev).data_ptr(), B, k, ids.data_ptr(), sc.data_ptr(), cnt.data_ptr(), st.data_ptr(), s.cuda_stream)
ps.exchange_device(rec.data_ptr(), B, k, g.data_ptr(), mi.data_ptr(), ms.data_ptr(), mc.data_ptr(), s.cuda_stream)
s.synchronize()
assert np.array_equal(mi.cpu().numpy().view(np.uint32), want[0]) and np.array_equal(ms.cpu().numpy().view(np.uint32), want[1].view(np.uint32))
assert np.array_equal(mc.cpu().numpy().view(np.uint32), want[2])
ps.close()
print("RCCL_PATH_OK")
"""


def test_rccl_call_path_world_of_one():
    """ncclCommInitAll / ncclAllGather through the dlopen'ed librccl with a single rank (the only RCCL run a one-GPU box can
    make); in a child process so the forced-RCCL environment does not leak into the other tests"""
    out = subprocess.run([sys.executable, "-c", _RCCL_CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=600,
                         env={**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert out.returncode == 0 and "RCCL_PATH_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
