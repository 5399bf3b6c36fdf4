"""GPU parity at the metric's own dimension (BASELINE.json: "1024-dim dense cosine"): walk lists per level and final
results bit-exact vs the oracle for u8 / quaternary / f32 storage at dim 1024, a >= 200k x 1024 clustered corpus with the
reference's default hyper-parameters (config.toml:20-24) in both visited modes, and the 2-shard scheme (id_base != 0 ->
packed per-shard record -> cos_merge_topk_packed_device) against the oracle running the same 2-shard scheme."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H
from tests.test_gpu_parity import _assert_same_search, _assert_same_walk

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,storage,res", [("u8", O.STORAGE_U8, 0), ("q2", O.STORAGE_SUBBYTE, 2), ("f32", O.STORAGE_F32, 0),
                                              ("f16", O.STORAGE_F16, 0)])
@pytest.mark.parametrize("corpus", ["uniform", "clustered"])
def test_dim1024_walk_and_search_match_oracle(name, storage, res, corpus):
    n, dim = 4000, 1024
    X = H.uniform_corpus(n, dim, seed=29) if corpus == "uniform" else H.clustered_corpus(n, dim, n_centers=24, sigma=0.03, seed=29)
    oix = H.oracle_index(X, storage, res, num_layers=5, ef_construction=64, ef_search=128)
    dix = H.device_index_from_oracle(oix, X)
    Q = np.concatenate([H.queries_from(X, 40, noise=0.005, seed=3), H.uniform_corpus(8, dim, seed=99)])
    _assert_same_walk(oix, dix, Q)
    _assert_same_search(oix, dix, Q, 10)
    for ef in (32, 64, 256, 512):
        oix.set_ef_search(ef)
        dix.set_ef_search(ef)
        _assert_same_search(oix, dix, Q[:16], 10)


def _device_built_pair(X, values_range, visited_mode=0, **hp):
    """index built by the DEVICE builder + the oracle loaded with the same graph (the oracle quantizes the corpus itself)"""
    import cosdata_amd as ca
    d = X.shape[1]
    h = ca.HNSWHyperParams(**hp)
    dix = ca.HNSWIndex(d, h, ca.DistanceMetric.Cosine, ca.StorageType.UnsignedByte(), values_range, shortlist_size=64, seed=5,
                       visited_mode=visited_mode)
    dix.upload_vectors(X).build(4096)
    op = O.HNSWParams(dim=d, storage=O.STORAGE_U8, range_lo=values_range[0], range_hi=values_range[1], num_layers=h.num_layers,
                      neighbors_count=h.neighbors_count, level0_neighbors_count=h.level_0_neighbors_count,
                      ef_construction=h.ef_construction, ef_search=h.ef_search, seed=5)
    op.visited_mode = visited_mode
    oix = O.OracleIndex(op).set_vectors(X).import_graph(dix.download_graph(), dix.download_root())
    return dix, oix


def test_200k_x_1024_clustered_default_params_both_visited_modes():
    """>= 200k x 1024, Gaussian mixture, auto-sampled values_range, reference defaults (num_layers 9, M 32 / M0 64,
    ef_construction 128, ef_search 256): REF filter (ID parity mode) and EXACT filter on a graph with >= 100k nodes."""
    import cosdata_amd as ca
    n, d = 200_000, 1024
    rng = np.random.default_rng(12)
    centers = rng.standard_normal((200, d)).astype(np.float32)
    centers /= np.linalg.norm(centers, axis=1, keepdims=True)
    X = np.empty((n, d), np.float32)
    for s in range(0, n, 20000):
        x = centers[rng.integers(0, 200, 20000)] + (0.8 / np.sqrt(d)) * rng.standard_normal((20000, d)).astype(np.float32)
        X[s:s + 20000] = x / np.linalg.norm(x, axis=1, keepdims=True)
    q = centers[rng.integers(0, 200, 256)] + (0.8 / np.sqrt(d)) * rng.standard_normal((256, d)).astype(np.float32)
    Q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    vr = ca.sample_values_range(X[:1000], 1.0)
    assert vr == O.sample_values_range(X[:1000], 1.0)
    dix, oix = _device_built_pair(X, vr)                           # defaults: 9 layers, ef_construction 128, ef_search 256
    assert dix.level_count(0) == n + 1
    _assert_same_walk(oix, dix, Q[:64])
    _assert_same_search(oix, dix, Q, 10)
    for ef in (64, 512):
        oix.set_ef_search(ef)
        dix.set_ef_search(ef)
        _assert_same_search(oix, dix, Q[:96], 10)
    ref_ids = dix.batch_search(Q, 10)[0]
    # EXACT visited filter on the same >= 100k-node graph
    oix.set_visited_mode(O.VISITED_EXACT)
    dix.set_visited_mode(1)
    for ef in (64, 256):
        oix.set_ef_search(ef)
        dix.set_ef_search(ef)
        _assert_same_walk(oix, dix, Q[:32])
        _assert_same_search(oix, dix, Q, 10)
    exact_ids = dix.batch_search(Q, 10)[0]
    gt, _ = dix.bruteforce_topk(Q, 10)
    rec = lambda a: np.mean([len(set(a[i]) & set(gt[i])) / 10 for i in range(len(Q))])
    assert rec(exact_ids) >= rec(ref_ids) - 0.02 and rec(exact_ids) > 0.9


def test_two_shards_on_one_gpu_match_oracle_two_shard_scheme():
    """SURVEY 8(e) end to end on one device: two cos_index shards (id_base 0 and n1) write their packed records
    [ids | scores | counts] with cos_search_batch_device, cos_merge_topk_packed_device merges them; the result must equal,
    bit for bit, the oracle searching the same two shards followed by the merge rule (score desc by total_cmp, larger id first)."""
    import torch
    import cosdata_amd as ca
    from cosdata_amd.sharding import merge_topk_packed_device, packed_views, packed_words
    from tests.test_sharded_merge import _merge_numpy
    dev = torch.device("cuda:0")
    n1, n2, d, B, k = 3000, 2600, 1024, 48, 10
    X = H.clustered_corpus(n1 + n2, d, n_centers=16, sigma=0.03, seed=91)
    Q = H.queries_from(X, B, noise=0.004, seed=17)
    shards = [(0, X[:n1]), (n1, X[n1:])]
    hp = dict(num_layers=4, ef_construction=48, ef_search=96)
    Qd = torch.from_numpy(Q).to(dev)
    recs, oracle_parts, keep = [], [], []
    for base, Xs in shards:
        oix = H.oracle_index(np.ascontiguousarray(Xs), O.STORAGE_U8, 0, seed=3 + base, **hp)
        dix = H.device_index_from_oracle(oix, np.ascontiguousarray(Xs), id_base=base)
        rec = torch.zeros(packed_words(B, k), dtype=torch.int32, device=dev)
        ids, sc, cnt = packed_views(rec, B, k)
        st = torch.zeros(B, dtype=torch.int32, device=dev)
        dix.batch_search_device(Qd.data_ptr(), B, k, ids.data_ptr(), sc.data_ptr(), cnt.data_ptr(), st.data_ptr(), 0)
        torch.cuda.synchronize()
        assert int(st.abs().sum().item()) == 0
        oi, osc, oc = oix.search_batch(Q, k, threads=4)[:3]
        gids = np.where(np.arange(k)[None, :] < oc[:, None], oi + base, oi).astype(np.uint32)
        # the per-shard record itself: global ids = id_base + local id
        assert np.array_equal(ids.cpu().numpy().view(np.uint32), gids)
        assert np.array_equal(sc.cpu().numpy().view(np.uint32), osc.view(np.uint32))
        assert np.array_equal(cnt.cpu().numpy().view(np.uint32), oc)
        recs.append(rec)
        oracle_parts.append((gids, osc, oc))
        keep.append(dix)
    g = torch.stack(recs).contiguous()
    m_i = torch.zeros(B, k, dtype=torch.int32, device=dev)
    m_s = torch.zeros(B, k, dtype=torch.float32, device=dev)
    m_c = torch.zeros(B, dtype=torch.int32, device=dev)
    merge_topk_packed_device(g, B, k, m_i, m_s, m_c, 0, 0)
    torch.cuda.synchronize()
    e_i, e_s, e_c = _merge_numpy(np.stack([p[0] for p in oracle_parts]), np.stack([p[1] for p in oracle_parts]),
                                 np.stack([p[2] for p in oracle_parts]), k)
    assert np.array_equal(m_c.cpu().numpy().view(np.uint32), e_c)
    assert np.array_equal(m_i.cpu().numpy().view(np.uint32), e_i)
    assert np.array_equal(m_s.cpu().numpy().view(np.uint32), e_s.view(np.uint32))
    got = m_i.cpu().numpy().view(np.uint32)
    assert (got >= n1).any() and (got < n1).any()          # both shards contribute
