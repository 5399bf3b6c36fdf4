"""GPU parity: the HIP path (through the C ABI) must match the oracle bit-exactly on identical inputs."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu

STORAGES = [
    ("u8", O.STORAGE_U8, 0),
    ("q2", O.STORAGE_SUBBYTE, 2),
    ("f32", O.STORAGE_F32, 0),
    ("f16", O.STORAGE_F16, 0),
    ("bin", O.STORAGE_SUBBYTE, 1),
    ("oct", O.STORAGE_SUBBYTE, 3),
]


@pytest.mark.parametrize("name,storage,res", STORAGES)
@pytest.mark.parametrize("dim", [96, 100, 768])
def test_quantize_matches_oracle(name, storage, res, dim):
    import cosdata_amd as ca
    rng = np.random.default_rng(1)
    x = rng.uniform(-1.3, 1.3, (37, dim)).astype(np.float32)
    x[0, :4] = [1.0, -1.0, np.nan, 5e30]  # wrap / saturate / NaN / huge quirks (SURVEY App. C 1,4)
    x[1, :] = 0.0
    codes, mags = ca.ScalarQuantization.quantize(x, ca.StorageType(ca.StorageKind(storage), res), (-1.0, 1.0))
    ocodes, omags = O.quantize_batch(x, storage, res, -1.0, 1.0)
    assert np.array_equal(codes, ocodes)
    assert np.array_equal(mags.view(np.uint32), omags.view(np.uint32))


@pytest.mark.parametrize("name,storage,res", STORAGES)
@pytest.mark.parametrize("dim", [100, 768])
def test_resident_codes_match_oracle(name, storage, res, dim):
    """cos_index_download_codes: the codes the index holds in HBM (device layout -> reference layout), root row last."""
    X = H.uniform_corpus(300, dim, seed=17) * 1.1
    oix = H.oracle_index(X, storage, res, num_layers=2, ef_construction=16, ef_search=16)
    dix = H.device_index_from_oracle(oix, X)
    codes, mags = dix.download_codes()
    ocodes, omags = O.quantize_batch(np.vstack([X, oix.root_raw()[None, :]]), storage, res, -1.0, 1.0)
    assert np.array_equal(np.asarray(codes).reshape(301, -1), np.ascontiguousarray(ocodes).view(np.uint8).reshape(301, -1))
    assert np.array_equal(np.asarray(mags).view(np.uint32), omags.view(np.uint32))


# (name, cos_index_set_latency_mode, cos_index_set_latency_waves, cos_index_set_walk_order, level table on).  With a level table
# (u8 codes) the library picks the kernel itself (walk_kernel_kind), so the variants that name a kernel switch the table off.
WALK_VARIANTS = [("throughput kernel", 0, 0, 0, False), ("one-wave latency kernel where it applies", 0xFFFFFFFF, 0, 0, False),
                 ("four-wave latency kernel where it applies", 0xFFFFFFFF, 0xFFFFFFFF, 0, False),
                 ("throughput kernel, split walk in locality order", 0, 0, 1, False),
                 ("default policy with the level table (throughput kernel, or four waves above ef 64)", 0xFFFFFFFF, 0xFFFFFFFF, 0, True),
                 ("level table + split walk in locality order", 0, 0, 1, True)]


def _set_variant(dix, max_b, max_b4, order_min=0, table=False):
    import cosdata_amd as ca
    dix.set_latency_mode(max_b)
    dix.set_latency_waves(max_b4)
    dix.set_walk_order(order_min)
    dix.set_walk_table(ca.HNSWIndex.WALK_TABLE_AUTO if table else 0, 1 if table else 0)


def _reset_variant(dix):
    import cosdata_amd as ca
    dix.set_latency_mode(ca.HNSWIndex.LATENCY_MODE_DEFAULT_MAX_B)
    dix.set_latency_waves(ca.HNSWIndex.LATENCY_WAVES_DEFAULT_MAX_B)
    dix.set_walk_order(ca.HNSWIndex.WALK_ORDER_DEFAULT_MIN_B)
    dix.set_walk_table(ca.HNSWIndex.WALK_TABLE_AUTO, ca.HNSWIndex.WALK_TABLE_DEFAULT_MIN_B)


def _assert_same_search(oix, dix, Q, top_k):
    """every variant of the walk (cos_index_set_latency_mode / _waves: walk_kernel / walk_lat_kernel / walk_lat4_kernel) must give the oracle's answer"""
    oids, osc, ocnt = oix.search_batch(Q, top_k, threads=4)[:3]
    for vname, max_b, max_b4, order_min, table in WALK_VARIANTS:
        _set_variant(dix, max_b, max_b4, order_min, table)
        ids, sc, cnt = dix.batch_search(Q, top_k)
        assert np.array_equal(cnt, ocnt), vname
        for b in range(Q.shape[0]):
            c = int(cnt[b])
            assert np.array_equal(ids[b, :c], oids[b, :c]), f"{vname}: query {b}: ids differ\n{ids[b,:c]}\n{oids[b,:c]}"
            assert np.array_equal(sc[b, :c].view(np.uint32), osc[b, :c].view(np.uint32)), f"{vname}: query {b}: scores differ"
    _reset_variant(dix)


def ca_default_latency():
    import cosdata_amd as ca
    return ca.HNSWIndex.LATENCY_MODE_DEFAULT_MAX_B


def _assert_same_walk(oix, dix, Q):
    ow = [oix.ann_search(Q[b]) for b in range(Q.shape[0])]
    for vname, max_b, max_b4, order_min, table in WALK_VARIANTS:
        _set_variant(dix, max_b, max_b4, order_min, table)
        ids, sims, counts = dix.ann_search_batch(Q)
        for b in range(Q.shape[0]):
            oi, osim, olc = ow[b]
            assert np.array_equal(counts[b], olc), f"{vname}: query {b}: level counts {counts[b]} vs {olc}"
            off = 0
            for s, c in enumerate(olc):
                c = int(c)
                assert np.array_equal(ids[b, s, :c], oi[off:off + c]), f"{vname}: query {b} slot {s}: walk ids differ"
                assert np.array_equal(sims[b, s, :c].view(np.uint32), osim[off:off + c].view(np.uint32)), f"{vname}: query {b} slot {s}: sims differ"
                off += c
    _reset_variant(dix)


@pytest.mark.parametrize("name,storage,res", STORAGES)
@pytest.mark.parametrize("n,dim,ef", [(2000, 96, 64), (3000, 100, 32), (6000, 768, 256), (1000, 48, 32), (1500, 400, 40), (800, 16, 16),
                                      (1200, 1536, 64)])
def test_search_matches_oracle(name, storage, res, n, dim, ef):
    X = H.uniform_corpus(n, dim, seed=7)
    oix = H.oracle_index(X, storage, res, num_layers=5, ef_construction=64, ef_search=ef)
    dix = H.device_index_from_oracle(oix, X)
    Q = np.concatenate([H.queries_from(X, 24, seed=3), H.uniform_corpus(8, dim, seed=99)])
    _assert_same_walk(oix, dix, Q)
    _assert_same_search(oix, dix, Q, 10)
    _assert_same_search(oix, dix, Q, 5)


def test_search_clustered_default_params():
    X = H.clustered_corpus(20000, 128, seed=5)
    oix = H.oracle_index(X, O.STORAGE_U8, 0, ef_construction=64)
    dix = H.device_index_from_oracle(oix, X)
    Q = H.queries_from(X, 64, seed=11)
    _assert_same_walk(oix, dix, Q)
    _assert_same_search(oix, dix, Q, 10)


def test_zero_norm_query_is_calculation_error():
    import cosdata_amd as ca
    X = H.uniform_corpus(500, 96, seed=2)
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=3, ef_construction=32, ef_search=32)
    dix = H.device_index_from_oracle(oix, X)
    Q = H.queries_from(X, 4)
    Q[2, :] = -1.0  # quantizes to all-zero bytes -> |q| = 0 -> DistanceError::CalculationError (cosine.rs:228-232)
    o = oix.search_batch(Q, 5, raise_on_error=False)
    for _, max_b, max_b4, order_min, table in WALK_VARIANTS:
        _set_variant(dix, max_b, max_b4, order_min, table)
        with pytest.raises(ca.CosdataError) as ei:
            dix.batch_search(Q, 5)
        assert ei.value.status == 2
        ids, sc, cnt, rc, status = dix.batch_search(Q, 5, return_status=True)
        assert rc == 2 and o[3] == 2
        assert np.array_equal(status, o[4])


@pytest.mark.parametrize("name,storage,res", STORAGES[:2])
def test_exact_visited_mode_matches_oracle(name, storage, res):
    """COS_VISITED_EXACT (recall mode): exact per-query visited bitset instead of the lossy PerformantFixedSet."""
    X = H.clustered_corpus(5000, 96, n_centers=16, seed=21)
    oix = H.oracle_index(X, storage, res, num_layers=4, ef_construction=48, ef_search=48)
    dix = H.device_index_from_oracle(oix, X)
    Q = H.queries_from(X, 40, seed=2)
    ref_ids = dix.batch_search(Q, 10)[0]
    oix.set_visited_mode(O.VISITED_EXACT)
    dix.set_visited_mode(1)
    _assert_same_walk(oix, dix, Q)
    _assert_same_search(oix, dix, Q, 10)
    exact_ids = dix.batch_search(Q, 10)[0]
    assert exact_ids.shape == ref_ids.shape


def test_auto_quantization_sampler_matches_oracle():
    import cosdata_amd as ca
    r = np.random.default_rng(4)
    for scale, n, d in [(0.036, 1000, 768), (0.2, 100, 96), (1.0, 64, 100), (0.01, 7, 33)]:
        x = (r.standard_normal((n, d)) * scale).astype(np.float32)
        assert ca.sample_values_range(x, 1.0) == O.sample_values_range(x, 1.0)
        assert ca.sample_values_range(x, 5.0) == O.sample_values_range(x, 5.0)


ALL_STORAGES = [("u8", O.STORAGE_U8, 0), ("bin", O.STORAGE_SUBBYTE, 1), ("q2", O.STORAGE_SUBBYTE, 2), ("oct", O.STORAGE_SUBBYTE, 3),
                ("f16", O.STORAGE_F16, 0), ("f32", O.STORAGE_F32, 0)]


@pytest.mark.parametrize("name,storage,res", ALL_STORAGES)
@pytest.mark.parametrize("dim", [100, 768])
def test_operators_every_storage_and_metric(name, storage, res, dim):
    """QuantizationMetric::quantize + DistanceMetric::calculate for every Storage kind x metric, incl. the error arms
    (StorageMismatch / CalculationError / unimplemented!) — values bit-exact vs the oracle."""
    import cosdata_amd as ca
    r = np.random.default_rng(17)
    x = r.uniform(-1.2, 1.2, (40, dim)).astype(np.float32)
    x[3] = 0.0
    x[4] = -1.0
    stype = ca.StorageType(ca.StorageKind(storage), res)
    codes, mags = ca.ScalarQuantization.quantize(x, stype, (-1.0, 1.0))
    ocodes, omags = O.quantize_batch(x, storage, res, -1.0, 1.0)
    assert np.array_equal(codes, ocodes)
    assert np.array_equal(mags.view(np.uint32), omags.view(np.uint32))
    px = r.integers(0, 40, 96).astype(np.uint32)
    py = r.integers(0, 40, 96).astype(np.uint32)
    px[:2], py[:2] = [3, 4], [5, 6]
    for metric in (O.METRIC_COSINE, O.METRIC_EUCLIDEAN, O.METRIC_HAMMING, O.METRIC_DOT):
        vals, status = ca.distance_batch(ca.DistanceMetric(metric), stype, dim, codes, mags, codes, mags, px, py)
        for p in range(px.size):
            rc, v = O.distance(metric, storage, res, dim, ocodes[px[p]], omags[px[p]], ocodes[py[p]], omags[py[p]])
            assert status[p] == rc, (name, metric, p, status[p], rc)
            if rc == 0:
                assert np.float32(vals[p]).tobytes() == np.float32(v).tobytes() or (np.isnan(vals[p]) and np.isnan(v)), (name, metric, p, vals[p], v)
