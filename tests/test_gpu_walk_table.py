"""GPU: the level table of big launches (cos_index_set_walk_table; WalkArgs::tab, kernels_flat.hip launch_level_table).  The similarity
of every (query, node) pair of the graph's small upper levels is computed ahead of the walk by one exact-integer i8 MFMA GEMM and the
walk of those levels reads it instead of gathering and dotting code rows: every per-level list and every result must keep its bits."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _same(a, b):
    return all(np.array_equal(np.asarray(x).view(np.uint32), np.asarray(y).view(np.uint32)) for x, y in zip(a, b))


def _check_against_oracle(oix, res, Q, top_k, sample):
    ids, sc, cnt = res
    oids, osc, ocnt = oix.search_batch(Q[sample], top_k, threads=4)[:3]
    assert np.array_equal(cnt[sample], ocnt)
    for j, b in enumerate(sample):
        c = int(ocnt[j])
        assert np.array_equal(ids[b, :c], oids[j, :c]), f"query {b}"
        assert np.array_equal(sc[b, :c].view(np.uint32), osc[j, :c].view(np.uint32)), f"query {b}"


@pytest.mark.parametrize("dim,visited,metric", [(96, 0, O.METRIC_COSINE), (96, 1, O.METRIC_COSINE), (768, 0, O.METRIC_COSINE),
                                                (1024, 0, O.METRIC_DOT), (80, 0, O.METRIC_DOT)])
def test_table_levels_keep_every_bit(dim, visited, metric):
    import cosdata_amd as ca
    n = 6000 if dim < 512 else 3000
    X = H.clustered_corpus(n, dim, n_centers=24, seed=31 + dim)
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=4, ef_construction=48, ef_search=32, metric=metric)
    dix = H.device_index_from_oracle(oix, X, visited_mode=visited)
    if visited:
        oix.set_visited_mode(O.VISITED_EXACT)
    B = ca.HNSWIndex.WALK_TABLE_DEFAULT_MIN_B + 37
    Q = H.queries_from(X, B, noise=0.05, seed=5)
    lmin, cols = dix.walk_table_info()
    assert lmin == 1 and cols == sum(dix.level_count(l) for l in range(1, 5))   # every upper level fits the default 8192 columns
    dix.enable_timing(True)
    with_tab = dix.batch_search(Q, 10)
    sp = dix.last_walk_split()
    st = dix.last_stats()
    assert sp.table_level_min == 1 and sp.table_cols == cols and sp.table_evals > 0 and sp.table_ms > 0
    assert sp.upper_evals + sp.lower_evals == st.evals and sp.upper_expansions + sp.lower_expansions == st.expansions
    walk_tab = dix.ann_search_batch(Q)
    dix.set_walk_table(0, 0)
    assert dix.walk_table_info() == (0, 0)
    plain = dix.batch_search(Q, 10)
    sp0 = dix.last_walk_split()
    assert sp0.table_level_min == 0 and sp0.table_evals == 0
    assert sp0.lower_evals + sp0.upper_evals == st.evals                       # the same walk, evaluation for evaluation
    walk_plain = dix.ann_search_batch(Q)
    assert _same(with_tab, plain)
    assert _same(walk_tab, walk_plain)
    _check_against_oracle(oix, with_tab, Q, 10, np.arange(0, B, 41))
    # a table that stops above level 1: levels below it take the row path inside the same launch
    top = sum(dix.level_count(l) for l in range(3, 5))
    dix.set_walk_table(top + 1, 1)
    assert dix.walk_table_info() == (3, top)
    assert _same(dix.ann_search_batch(Q[:300]), tuple(np.asarray(a)[:300] for a in walk_plain))


def test_table_with_locality_order_and_zero_query():
    """a launch big enough for both the table and the locality order; a zero query fails with CalculationError either way"""
    import cosdata_amd as ca
    X = H.clustered_corpus(6000, 96, n_centers=24, seed=77)
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=4, ef_construction=48, ef_search=32)
    dix = H.device_index_from_oracle(oix, X)
    B = ca.HNSWIndex.WALK_ORDER_DEFAULT_MIN_B + 100
    Q = H.queries_from(X, B, noise=0.05, seed=9)
    Q[17] = oix.params.range_lo          # quantizes to the all-zero code: |q| = 0 -> 0/0 at the first evaluation (cosine.rs:228-232)
    Q[B - 1] = oix.params.range_lo
    ids, sc, cnt, rc, status = dix.batch_search(Q, 10, return_status=True)
    dix.set_walk_table(0, 0)
    ids0, sc0, cnt0, rc0, status0 = dix.batch_search(Q, 10, return_status=True)
    assert rc == 2 and rc0 == 2 and status[17] == 2 and status[B - 1] == 2 and np.count_nonzero(status) == 2
    assert np.array_equal(status, status0) and _same((ids, sc, cnt), (ids0, sc0, cnt0))
    ok = np.array([b for b in range(0, B, 53) if status[b] == 0])
    _check_against_oracle(oix, (ids, sc, cnt), Q, 10, ok)


def test_table_follows_the_graph():
    """a new graph on the same handle: the table's operand (the nodes' code rows) is gathered again"""
    import cosdata_amd as ca
    X = H.clustered_corpus(5000, 64, n_centers=16, seed=8)
    o1 = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=4, ef_construction=32, ef_search=32, seed=1)
    o2 = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=4, ef_construction=32, ef_search=32, seed=2)
    dix = H.device_index_from_oracle(o1, X)
    B = ca.HNSWIndex.WALK_TABLE_DEFAULT_MIN_B
    Q = H.queries_from(X, B, noise=0.05, seed=6)
    _check_against_oracle(o1, dix.batch_search(Q, 5), Q, 5, np.arange(0, B, 61))
    dix.upload_vectors(X)
    dix.upload_graph(o2.export_graph(), o2.root_raw())
    r2 = dix.batch_search(Q, 5)
    assert dix.last_walk_split().table_evals > 0
    _check_against_oracle(o2, r2, Q, 5, np.arange(0, B, 61))


@pytest.mark.parametrize("dim,metric", [(768, O.METRIC_COSINE), (96, O.METRIC_DOT)])
def test_one_client_batch_reads_the_table_too(dim, metric):
    """a 256-query batch takes the four-wave latency kernel, which reads the level table as well (kernels_walk_lat4.hip): same bits
    as without the table and as the throughput kernel"""
    X = H.clustered_corpus(4000, dim, n_centers=16, seed=dim)
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=4, ef_construction=48, ef_search=48, metric=metric)
    dix = H.device_index_from_oracle(oix, X)
    Q = H.queries_from(X, 256, noise=0.05, seed=2)
    Q[5] = oix.params.range_lo                                # |q| = 0: CalculationError at the entry node's table entry (cosine only)
    assert dix.walk_table_info()[0] == 1
    res_tab = dix.batch_search(Q, 10, return_status=True)
    walk_tab = dix.ann_search_batch(np.delete(Q, 5, axis=0))
    dix.set_walk_table(0, 0)
    res_plain = dix.batch_search(Q, 10, return_status=True)
    walk_plain = dix.ann_search_batch(np.delete(Q, 5, axis=0))
    dix.set_latency_mode(0)                                   # throughput kernel, no table
    res_tk = dix.batch_search(Q, 10, return_status=True)
    for a, b in ((res_tab, res_plain), (res_tab, res_tk)):
        assert a[3] == b[3] and np.array_equal(a[4], b[4]) and _same(a[:3], b[:3])
    assert _same(walk_tab, walk_plain)
    assert (res_tab[4][5] == 2) == (metric == O.METRIC_COSINE)
    ok = np.array([b for b in range(0, 256, 7) if res_tab[4][b] == 0])
    _check_against_oracle(oix, res_tab[:3], Q, 10, ok)


@pytest.mark.parametrize("dim,metric", [(768, O.METRIC_COSINE), (1024, O.METRIC_COSINE), (128, O.METRIC_DOT), (384, O.METRIC_COSINE)])
def test_query_resident_table_gemm_equals_tile_kernel_and_rows(dim, metric):
    """round 5: the table is the integer dot `as f32` written by the query-resident GEMM (kernels_scan.hip level_table_areg: code rows of
    whole 64-byte chunks) or by the tile kernel (tuning knob walk_table_gemm = 0; other row lengths), the walk forms the quotient.
    Several query groups of 256 with a ragged last one, several column tiles of 64 with a ragged last one, tiles inside the matrix
    (unpredicated stores) and on its edges: both kernels, the table-less walk and the oracle agree bit for bit."""
    from cosdata_amd import _lib
    n = 2600
    X = H.clustered_corpus(n, dim, n_centers=20, seed=3 + dim)
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=4, ef_construction=40, ef_search=40, metric=metric)
    dix = H.device_index_from_oracle(oix, X)
    B = 256 * 2 + 64 + 37
    Q = H.queries_from(X, B, noise=0.05, seed=11)
    lmin, cols = dix.walk_table_info()
    assert lmin == 1 and cols > 128
    res = {}
    for gemm in (1, 0):
        with _lib.tuning(walk_table_gemm=gemm):
            res[gemm] = (dix.batch_search(Q, 10), dix.ann_search_batch(Q))
            assert dix.last_walk_split().table_evals > 0
    dix.set_walk_table(0, 0)
    plain = (dix.batch_search(Q, 10), dix.ann_search_batch(Q))
    for gemm in (1, 0):
        assert _same(res[gemm][0], plain[0]) and _same(res[gemm][1], plain[1]), gemm
    _check_against_oracle(oix, res[1][0], Q, 10, np.arange(0, B, 23))


def test_set_root_on_a_live_graph_regathers_the_table():
    """cos_index_set_root re-quantizes the root's code row; the level-table operand holds a gathered copy of it (the root is a node of
    every level), so the table must be invalidated: with and without a table the second root gives the same answers"""
    X = H.clustered_corpus(3000, 128, n_centers=12, seed=19)
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=4, ef_construction=32, ef_search=32)
    dix = H.device_index_from_oracle(oix, X)
    Q = H.queries_from(X, 64, noise=0.05, seed=4)
    first = dix.batch_search(Q, 5)
    assert dix.last_walk_split().table_evals > 0
    rng = np.random.default_rng(5)
    new_root = rng.uniform(-0.9, 0.9, size=128).astype(np.float32)
    dix.set_root(new_root)
    with_tab = dix.ann_search_batch(Q)
    assert dix.last_walk_split().table_evals > 0
    dix.set_walk_table(0, 0)
    plain = dix.ann_search_batch(Q)
    assert _same(with_tab, plain)


@pytest.mark.parametrize("dim,metric", [(768, O.METRIC_COSINE), (128, O.METRIC_COSINE), (384, O.METRIC_DOT)])
def test_quaternary_codes_walk_the_table_too(dim, metric):
    """round 5: the level table for SubByte(2) codes — the integer dot_product_quaternary of every (query, table node) pair from one
    i8 MFMA GEMM over the expanded digits (query-resident kernel; tuning knob walk_table_gemm = 0: the tile kernel), the walk forms
    the quotient with the nodes' raw-vector norms (scalar.rs:31-32).  Per-level lists and results keep every bit: both GEMM kernels,
    the table-less walk and the oracle agree."""
    from cosdata_amd import _lib
    n = 2600
    X = H.clustered_corpus(n, dim, n_centers=20, seed=5 + dim) * 0.9
    oix = H.oracle_index(X, O.STORAGE_SUBBYTE, 2, num_layers=4, ef_construction=40, ef_search=40, metric=metric)
    dix = H.device_index_from_oracle(oix, X)
    B = 256 + 64 + 21
    Q = H.queries_from(X, B, noise=0.05, seed=13) * 0.9
    dix.set_latency_mode(0)                 # small launches of quaternary codes would take the one-wave latency kernel, which reads no table
    dix.set_latency_waves(0)
    lmin, cols = dix.walk_table_info()
    assert lmin == 1 and cols > 128
    res = {}
    for gemm in (1, 0):
        with _lib.tuning(walk_table_gemm=gemm):
            res[gemm] = (dix.batch_search(Q, 10), dix.ann_search_batch(Q))
            assert dix.last_walk_split().table_evals > 0
    dix.set_walk_table(0, 0)
    plain = (dix.batch_search(Q, 10), dix.ann_search_batch(Q))
    assert dix.last_walk_split().table_evals == 0
    for gemm in (1, 0):
        assert _same(res[gemm][0], plain[0]) and _same(res[gemm][1], plain[1]), gemm
    _check_against_oracle(oix, res[1][0], Q, 10, np.arange(0, B, 23))


@pytest.mark.parametrize("ef,m0,m", [(64, 64, 32), (128, 64, 64), (100, 128, 64), (256, 64, 32)])
def test_ranked_merge_of_the_table_levels_keeps_every_bit(ef, m0, m):
    """round 6 (walk_kernel.inc commit_merge): on table levels an expansion's screened winners go into the pool by ONE ranked merge
    from `walk_merge_min` winners on.  Pools of one / two / four keys per lane (ef 64 / 65..128 / 129..256; `walk_r2` = 0 puts ef
    65..128 back on the four-key pool), merge thresholds 1 (always) / default / 0 (never): per-level lists and results identical to
    each other, to the walk without a table, and to the oracle."""
    import cosdata_amd as ca
    from cosdata_amd import _lib
    X = H.clustered_corpus(6000, 96, n_centers=24, seed=300 + ef)
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=4, ef_construction=64, ef_search=ef, level0_neighbors_count=m0, neighbors_count=m)
    dix = H.device_index_from_oracle(oix, X)
    B = ca.HNSWIndex.WALK_TABLE_DEFAULT_MIN_B + 19
    Q = H.queries_from(X, B, noise=0.05, seed=12)
    assert dix.walk_table_info()[0] == 1
    dix.set_latency_waves(0)      # above ef 64 one small launch would take the four-wave latency kernel: this test is about walk_kernel
    dix.enable_timing(True)
    ref_res = ref_walk = None
    for knobs in ({"walk_merge_min": 0}, {"walk_merge_min": 1}, {}, {"walk_merge_min": 2, "walk_r2": 0}, {"walk_merge_min": 64}):
        with _lib.tuning(**knobs):
            res = dix.batch_search(Q, 10)
            assert dix.last_walk_split().table_evals > 0
            walk = dix.ann_search_batch(Q)
        if ref_res is None:
            ref_res, ref_walk = res, walk
        assert _same(res, ref_res) and _same(walk, ref_walk), knobs
    dix.set_walk_table(0, 0)
    assert _same(dix.ann_search_batch(Q), ref_walk)
    _check_against_oracle(oix, ref_res, Q, 10, np.arange(0, B, 37))


@pytest.mark.parametrize("dim,visited,m0,m", [(768, 0, 64, 32), (1024, 0, 128, 64), (640, 1, 64, 32)])
def test_norms_beside_the_adjacency_keep_every_bit(dim, visited, m0, m):
    """round 6 (LevelDev::adj_mag): on row levels the winners' norms come with the adjacency row instead of one gather each; the same
    per-level lists and results with the knob off, after a root replaced on a live graph (the arrays are refilled), and as the oracle"""
    from cosdata_amd import _lib
    X = H.clustered_corpus(3000, dim, n_centers=16, seed=dim + 5)
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=4, ef_construction=48, ef_search=40, level0_neighbors_count=m0, neighbors_count=m)
    dix = H.device_index_from_oracle(oix, X, visited_mode=visited)
    if visited:
        oix.set_visited_mode(O.VISITED_EXACT)
    dix.set_latency_mode(0)
    dix.set_latency_waves(0)      # the throughput kernel at every launch size
    Q = H.queries_from(X, 1500, noise=0.05, seed=3)
    with _lib.tuning(walk_adj_mag=2):           # (2 = at every ef and launch size: by default a launch this small keeps the gathers)
        on = dix.batch_search(Q, 10)
        walk_on = dix.ann_search_batch(Q)
    with _lib.tuning(walk_adj_mag=0):
        off = dix.batch_search(Q, 10)
        walk_off = dix.ann_search_batch(Q)
    assert _same(on, off) and _same(walk_on, walk_off)
    _check_against_oracle(oix, on, Q, 10, np.arange(0, 1500, 23))
    root2 = np.ascontiguousarray(X[7] * 0.5 + X[11] * 0.5)
    dix.set_root(root2)
    oix.set_root_raw(root2)
    with _lib.tuning(walk_adj_mag=2):
        again = dix.batch_search(Q, 10)
    with _lib.tuning(walk_adj_mag=0):
        assert _same(again, dix.batch_search(Q, 10))
    Qbig = H.queries_from(X, 4200, noise=0.05, seed=13)   # a launch big enough for the default policy to read the norms beside the adjacency
    big = dix.batch_search(Qbig, 10)
    with _lib.tuning(walk_adj_mag=0):
        assert _same(big, dix.batch_search(Qbig, 10))
    _check_against_oracle(oix, again, Q, 10, np.arange(0, 1500, 23))
