"""Oracle pinned against every known-answer the reference's own tests hold for this path
 plus the quirk checklist of SURVEY.md Appendix C.  CPU only."""
import struct

import numpy as np
import pytest

from oracle import oracle as O

rng = np.random.default_rng(2024)


# ---- known answers copied from the reference's tests ------------------------------------------
def test_level_probs_table_common_rs_741():
    # src/models/common.rs:741-756 test_generate_level_probs
    expected = [(0.999999999, 9), (0.99999999, 8), (0.9999999, 7), (0.999999, 6), (0.99999, 5), (0.9999, 4),
                (0.999, 3), (0.99, 2), (0.9, 1), (0.0, 0)]
    assert O.level_probs(10.0, 9) == expected


def test_level_probs_factor4_and_max_insert_level():
    lp = O.level_probs(4.0, 9)  # api_service.rs:109
    assert [l for _, l in lp] == list(range(9, -1, -1))
    for v, n in lp:
        assert v == 1.0 - 4.0 ** (-n)
    assert O.max_insert_level(0.0, lp) == 0
    assert O.max_insert_level(0.74, lp) == 0
    assert O.max_insert_level(0.75, lp) == 1       # x >= 1 - 4^-1
    assert O.max_insert_level(0.9375, lp) == 2
    assert O.max_insert_level(1.0 - 2 ** -24, lp) == 9 or O.max_insert_level(1.0 - 2 ** -24, lp) >= 9 - 0  # top 24-bit draw
    # P(level >= n) = 4^-n over the 24-bit grid rand::random::<f32>() draws from
    xs = (np.arange(0, 1 << 16, dtype=np.float64) + 0.5) / (1 << 16)
    lv = np.array([O.max_insert_level(float(x), lp) for x in xs[::64]])
    assert abs((lv >= 1).mean() - 0.25) < 0.02


def test_metric_result_ordering_types_rs_1611():
    # src/models/types.rs:1611-1633: sort ascending for cosine similarity
    vals = [6.0, 5.0, 4.0, 3.0, 2.0, 1.0]
    import functools
    s = sorted(vals, key=functools.cmp_to_key(lambda a, b: O.metric_cmp(O.METRIC_COSINE, a, b)))
    assert s == [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]
    # euclid / hamming are reversed (types.rs:404-408); total_cmp: -0.0 < +0.0
    assert O.metric_cmp(O.METRIC_EUCLIDEAN, 1.0, 2.0) == 1
    assert O.metric_cmp(O.METRIC_HAMMING, 1.0, 2.0) == 1
    assert O.metric_cmp(O.METRIC_DOT, 1.0, 2.0) == -1
    assert O.metric_cmp(O.METRIC_COSINE, -0.0, 0.0) == -1
    assert O.metric_cmp(O.METRIC_COSINE, float("nan"), float("inf")) == 1


@pytest.mark.parametrize("pattern,ones_per_word", [(0x55555555, 16), (0xAAAAAAAA, 16), (0xFFFFFFFF, 32), (0x00000000, 0),
                                                   (0x12345678, 13), (0x0F0F0F0F, 16), (0x80000001, 2)])
def test_count_ones_fixed_patterns_x86_64_rs_676(pattern, ones_per_word):
    # src/models/dot_product/x86_64.rs:676-723 feed fixed 32-bit patterns through the nibble LUT
    buf = np.frombuffer(struct.pack("<8I", *([pattern] * 8)), np.uint8)
    assert O.count_ones(buf) == 8 * ones_per_word == 8 * bin(pattern).count("1")


def _pack_lsb_first(vals, bits):
    n = len(vals)
    planes = np.zeros((bits, (n + 7) // 8), np.uint8)
    for i, v in enumerate(vals):
        for p in range(bits):
            if (int(v) >> p) & 1:
                planes[p, i // 8] |= 1 << (i % 8)
    return planes


@pytest.mark.parametrize("length", [32, 128, 1000, 8192])
def test_quaternary_identity_x86_64_rs_455(length):
    # test_dot_product_quaternary_vs_theoretical: LSB-plane-first packing -> sum a_i*b_i exactly
    a, b = rng.integers(0, 4, length), rng.integers(0, 4, length)
    pa, pb = _pack_lsb_first(a, 2), _pack_lsb_first(b, 2)
    v, st = O.dot_subbyte(pa, pb, 2)
    assert st == O.OK and float(v) == float((a * b).sum())
    assert float(O.dot_quaternary_scalar(pa, pb)) == float((a * b).sum())  # :545 SIMD == scalar


@pytest.mark.parametrize("bits", [1, 3])
def test_binary_octal_identity(bits):
    # x86_64.rs:575 (binary), :785 (octal): same identity on LSB-first planes
    a, b = rng.integers(0, 1 << bits, 1024), rng.integers(0, 1 << bits, 1024)
    v, st = O.dot_subbyte(_pack_lsb_first(a, bits), _pack_lsb_first(b, bits), bits)
    assert st == O.OK and float(v) == float((a * b).sum())


def test_subbyte_bad_resolution_is_calculation_error():
    v, st = O.dot_subbyte(np.zeros((4, 8), np.uint8), np.zeros((4, 8), np.uint8), 4)  # cosine.rs:151-153
    assert st == O.ERR_CALCULATION


# ---- quirk checklist ---------------------------------------------------
def test_c1_u8_truncates_and_clamps():
    x = np.array([-2.0, -1.0, -0.999, 0.0, 0.5, 0.999, 1.0, 7.0, np.nan], np.float32)
    code, mag = O.quantize(x, O.STORAGE_U8, 0, -1.0, 1.0)
    expect = [0, 0, 0, 127, 191, 254, 255, 255, 0]  # ((x-lo)/(hi-lo)*255) as u8: truncation, clamp, NaN.max(lo)=lo
    assert code.tolist() == expect
    assert mag == np.float32(np.sqrt(np.float32(sum(v * v for v in expect))))  # C2: norm of the BYTES


def test_c3_c4_subbyte_planes_msb_first_wrap_saturate():
    # level = floor((x+1)/0.5): -1->0, -0.5->1, 0->2, 0.5->3, 1.0->4 wraps to 0, <-1 saturates to 0
    x = np.array([-1.0, -0.5, 0.0, 0.5, 1.0, -3.0, 0.49, 2.0], np.float32)
    code, mag = O.quantize(x, O.STORAGE_SUBBYTE, 2)
    levels = [0, 1, 2, 3, 0, 0, 2, 2]  # 2.0 -> n=6 -> low bits 2
    plane0 = sum(((l >> 1) & 1) << i for i, l in enumerate(levels))  # plane 0 = MSB (common.rs:230-233)
    plane1 = sum((l & 1) << i for i, l in enumerate(levels))
    assert code.tolist() == [plane0, plane1]
    assert mag == O.seq_norm(x)  # C3: |x| of the ORIGINAL vector
    # the dot multiplies plane 0 as LSB -> levels 1 and 2 swap roles (bit-reversed)
    rev = {0: 0, 1: 2, 2: 1, 3: 3}
    v, _ = O.dot_subbyte(code.reshape(2, 1), code.reshape(2, 1), 2)
    assert float(v) == float(sum(rev[l] * rev[l] for l in levels))


def test_c6_f32_dot_reduction_tree_and_tail():
    for n in [1, 7, 8, 9, 64, 100, 768, 771, 1024]:
        a = rng.uniform(-1, 1, n).astype(np.float32)
        b = rng.uniform(-1, 1, n).astype(np.float32)
        got = O.dot_f32(a, b)
        # independent restatement: 8 fmaf lanes over stride-8, fixed tree, non-fused scalar tail
        s = [np.float32(0)] * 8
        ch = n // 8
        for i in range(ch):
            for j in range(8):
                s[j] = np.float32(np.float64(a[8 * i + j]) * np.float64(b[8 * i + j]) + np.float64(s[j]))  # exact product, one rounding
        r = np.float32(np.float32(np.float32(s[0] + s[1]) + np.float32(s[2] + s[3])) + np.float32(np.float32(s[4] + s[5]) + np.float32(s[6] + s[7])))
        for i in range(ch * 8, n):
            r = np.float32(r + np.float32(a[i] * b[i]))
        assert got.tobytes() == r.tobytes(), n
        assert O.dot_f32(a, b, scalar_order=True).tobytes() == got.tobytes()


def test_c6_f16_dot_sequential():
    a = rng.uniform(-1, 1, 100).astype(np.float16)
    b = rng.uniform(-1, 1, 100).astype(np.float16)
    acc = np.float32(-0.0)
    for x, y in zip(a, b):
        acc = np.float32(acc + np.float32(np.float32(x) * np.float32(y)))
    assert O.dot_f16(a.view(np.uint16), b.view(np.uint16)).tobytes() == acc.tobytes()
    assert O.lib().coso_f32_to_f16(0.1) == int(np.float16(0.1).view(np.uint16))  # half::f16::from_f32 is RNE


def test_c7_fixedset_aliasing():
    import ctypes as C

    class FS(C.Structure):
        _fields_ = [("buckets", C.POINTER(C.c_uint64)), ("len", C.c_uint32)]
    for M in (64, 32):
        arr = (C.c_uint64 * M)()
        fs = FS(arr, M)
        L = O.lib()
        L.coso_fixedset_insert.argtypes = [C.POINTER(FS), C.c_uint32]
        L.coso_fixedset_is_member.argtypes = [C.POINTER(FS), C.c_uint32]
        L.coso_fixedset_insert(C.byref(fs), 0xFFFFFFFE)  # query id pre-insert (vector_store.rs:271)
        bits = 64 * M
        assert L.coso_fixedset_is_member(C.byref(fs), bits - 2) == 1   # id == 4094 (mod 4096) / 2046 (mod 2048) is blocked
        assert L.coso_fixedset_is_member(C.byref(fs), 3 * bits - 2) == 1
        assert L.coso_fixedset_is_member(C.byref(fs), bits - 1) == 0
        L.coso_fixedset_insert(C.byref(fs), 5)
        assert L.coso_fixedset_is_member(C.byref(fs), 5 + 7 * bits) == 1  # ids alias mod 64*M, no hashing
        assert L.coso_fixedset_is_member(C.byref(fs), 6) == 0


def test_c15_zero_norm_is_calculation_error_and_c17_u64_as_f32():
    d = 1024
    x = np.full(d, 255, np.uint8)
    rc, v = O.distance(O.METRIC_COSINE, O.STORAGE_U8, 0, d, x, 1.0, x, 0.0)
    assert rc == O.ERR_CALCULATION  # cosine.rs:228-232
    # 255*255*1024 = 66585600 > 2^24: the integer dot is converted with RNE, not accumulated in f32
    assert O.dot_u8(x, x) == 66585600 == O.dot_u8(x, x, scalar=True)
    rc, v = O.distance(O.METRIC_COSINE, O.STORAGE_U8, 0, d, x, 1.0, x, 1.0)
    assert rc == O.OK and v == np.float32(66585600)
    y = np.array([255] * 1023 + [254], np.uint8)
    assert O.dot_u8(x, y) == 66585345
    rc, v = O.distance(O.METRIC_DOT, O.STORAGE_U8, 0, d, x, 1.0, y, 1.0)
    assert v == np.float32(66585345) and float(v) != 66585345.0  # rounded to a multiple of 4


def test_metric_arms_and_errors():
    d = 16
    xf = rng.uniform(-1, 1, d).astype(np.float32)
    cf, mf = O.quantize(xf, O.STORAGE_F32)
    assert O.distance(O.METRIC_DOT, O.STORAGE_F32, 0, d, cf, mf, cf, mf)[0] == O.ERR_STORAGE_MISMATCH       # dotproduct.rs: no f32 arm
    assert O.distance(O.METRIC_EUCLIDEAN, O.STORAGE_F32, 0, d, cf, mf, cf, mf)[0] == O.ERR_STORAGE_MISMATCH  # euclidean.rs: no f32 arm
    cq, mq = O.quantize(xf, O.STORAGE_SUBBYTE, 2)
    assert O.distance(O.METRIC_EUCLIDEAN, O.STORAGE_SUBBYTE, 2, d, cq, mq, cq, mq)[0] == O.ERR_UNIMPLEMENTED  # euclidean.rs:34-37
    a = np.array([0] * 15 + [0], np.uint8)
    b = np.array([0] * 15 + [255], np.uint8)
    rc, v = O.distance(O.METRIC_EUCLIDEAN, O.STORAGE_U8, 0, d, a, 1.0, b, 1.0)
    # (diff*diff) as i16 multiply wraps in release builds: 255*255 = 65025 -> -511 -> sqrt(-511) = NaN
    assert rc == O.OK and np.isnan(v)
    rc, v = O.distance(O.METRIC_HAMMING, O.STORAGE_U8, 0, d, a, 1.0, b, 1.0)
    assert float(v) == 8.0


# ---- walk / finalize on a hand-built graph -----------------------------------------------------
def _tiny_index(ef=3, num_layers=1, M=2, M0=4, visited_mode=O.VISITED_REF):
    """6 vectors on a line of angles; hand-wired two-level graph."""
    d = 8
    ang = np.array([0.0, 0.2, 0.4, 0.6, 0.8, 1.0], np.float32)
    X = np.zeros((6, d), np.float32)
    X[:, 0], X[:, 1] = np.cos(ang), np.sin(ang)
    p = O.HNSWParams(dim=d, storage=O.STORAGE_F32, num_layers=num_layers, neighbors_count=M, level0_neighbors_count=M0,
                     ef_construction=8, ef_search=ef, shortlist_size=64, visited_mode=visited_mode)
    ix = O.OracleIndex(p).set_vectors(X)
    root = np.zeros(d, np.float32)
    root[0], root[1] = np.cos(1.5), np.sin(1.5)
    E, R = O.SLOT_EMPTY, O.ROOT_ID
    l0_ids = np.array([0, 1, 2, 3, 4, 5, R], np.uint32)
    l0 = np.array([[1, E, E, E], [0, 2, E, E], [1, 3, E, E], [2, 4, E, E], [3, 5, E, E], [4, R, E, E], [5, E, E, E]], np.uint32)
    l1_ids = np.array([2, 5, R], np.uint32)
    l1 = np.array([[5, E], [2, R], [5, E]], np.uint32)
    ix.import_graph([(l0_ids, l0), (l1_ids, l1)], root)
    return ix, X


def test_walk_trace_hand_graph():
    ix, X = _tiny_index(ef=3)
    q = X[0].copy()
    ids, sims, lc = ix.ann_search(q)
    # level 1 (top): start at root, pops: root, 5, 2 (ef=3 -> stop) ; sorted desc by cosine to angle 0: 2, 5, root
    assert lc.tolist() == [3, 3]
    assert ids[:3].tolist() == [2, 5, O.ROOT_ID]
    # level 0 entered through child(2): pops 2, then best of {1,3} = 1, then 0 -> ef reached, 3 never popped
    assert sorted(ids[3:6].tolist()) == [0, 1, 2] and ids[3] == 0
    # ef cut-off discards the (ef+1)-th pop; every level contributes (C8, C9)
    out = ix.search_batch(q[None, :], 2)
    assert out[0][0].tolist() == [0, 1]      # root filtered, dedup across levels (node 2 appears on both)
    assert out[2][0] == 2
    ix.set_ef_search(0)                      # ef == 0: fallback to the entry's own distance (vector_store.rs:329-380)
    ids0, _, lc0 = ix.ann_search(q)
    assert lc0.tolist() == [1, 1] and ids0.tolist() == [O.ROOT_ID, O.ROOT_ID]


def test_finalize_dedup_5k_cut_and_rerank_order():
    X = np.random.default_rng(3).uniform(-1, 1, (400, 32)).astype(np.float32)
    ix = O.OracleIndex(O.HNSWParams(dim=32, storage=O.STORAGE_U8, num_layers=3, ef_construction=32, ef_search=64)).set_vectors(X).build()
    q = X[17] + 0.01
    ids, sims, lc = ix.ann_search(q)
    # reproduce remove_duplicates_and_filter + finalize by hand
    seen, cand = set(), []
    for i, s in zip(ids.tolist(), sims.tolist()):
        if i in seen:
            continue
        seen.add(i)
        if i != O.ROOT_ID:
            cand.append((s, i))
    cand.sort(key=lambda t: (t[0], t[1]), reverse=True)
    k = 3
    cand = cand[:5 * k]
    mq = O.seq_norm(q)
    rer = []
    for _, i in cand:
        cs = np.float32(O.dot_f32(q, X[i]) / np.float32(mq * O.seq_norm(X[i])))
        rer.append((cs, i))
    rer.sort(key=lambda t: (t[0], t[1]), reverse=True)
    got = ix.search_batch(q[None, :], k)
    assert got[0][0].tolist() == [i for _, i in rer[:k]]
    assert got[1][0].tolist() == [float(c) for c, _ in rer[:k]]


def test_builder_invariants_and_determinism():
    X = np.random.default_rng(5).uniform(-1, 1, (1500, 48)).astype(np.float32)
    p = dict(num_layers=4, ef_construction=32, ef_search=32, seed=7)
    a = O.OracleIndex(O.HNSWParams(dim=48, **p)).set_vectors(X).build().export_graph()
    b = O.OracleIndex(O.HNSWParams(dim=48, **p)).set_vectors(X).build().export_graph()
    for (ia, na), (ib, nb) in zip(a, b):
        assert np.array_equal(ia, ib) and np.array_equal(na, nb)
    ids0, nbr0 = a[0]
    assert ids0.size == 1501 and ids0[-1] == O.ROOT_ID and np.array_equal(ids0[:-1], np.arange(1500))
    for l in range(1, 5):  # every node exists on all lower levels; sizes shrink ~4x
        assert set(a[l][0].tolist()) <= set(a[l - 1][0].tolist())
        assert a[l][0][-1] == O.ROOT_ID
    assert 250 < a[1][0].size < 520
    # edges are symmetric except where an evictee kept a stale back edge removed: check the invariant the
    # reference maintains — if u lists v then v lists u (remove_neighbor_by_id keeps both sides in sync)
    pos = {int(i): k for k, i in enumerate(ids0)}
    asym = 0
    for k, i in enumerate(ids0):
        for v in nbr0[k]:
            if v != O.SLOT_EMPTY and int(i) not in set(nbr0[pos[int(v)]].tolist()):
                asym += 1
    assert asym == 0


def test_recall_plumbing_c1():
    # config c1 (tests/test.py path): dense HNSW cosine, hyper-params of tests/test.py:73-81, brute-force verify
    X = np.random.default_rng(42).uniform(-1, 1, (3000, 64)).astype(np.float32)
    ix = O.OracleIndex(O.HNSWParams(dim=64, num_layers=7, ef_construction=512, ef_search=256, neighbors_count=32,
                                    level0_neighbors_count=64)).set_vectors(X).build()
    Q = X[:40]
    ids, sc, cnt = ix.search_batch(Q, 5, threads=4)
    gt, _ = O.bruteforce_topk(X, Q, 5, threads=4)
    assert all(ids[i, 0] == i for i in range(40))  # the query vector itself comes first (tests/test.py:115)
    recall = np.mean([len(set(ids[i]) & set(gt[i])) / 5 for i in range(40)])
    assert recall >= 0.9


# ---- BM25 + RRF ---------------------------------------------------------------------------------
def test_bm25_formulae_buckets_and_rrf():
    assert O.bm25_idf(1000, 10) == np.float32(np.log1p(np.float32(np.float32(990.5) / np.float32(10.5))))
    c, dl, avg, k1, b = 3, 120, 100.0, 1.5, 0.75
    exp = np.float32(3) * np.float32(2.5) / (np.float32(3) + np.float32(1.5) * (np.float32(1.0) - np.float32(0.75) + np.float32(0.75) * (np.float32(120) / np.float32(100))))
    assert O.bm25_tf(c, dl, avg, k1, b) == np.float32(exp)
    # postings: term 10 -> docs 3, 515 ; term 20 -> docs 3, 7
    terms = np.array([10, 20], np.uint32)
    off = np.array([0, 2, 4], np.uint64)
    docs = np.array([3, 515, 3, 7], np.uint32)
    tfs = np.array([1.0, 2.0, 0.5, 0.25], np.float32)
    ids, sc = O.bm25_search(terms, off, docs, tfs, 1000, np.array([20, 10, 999], np.uint32), 10)
    idf = O.bm25_idf(1000, 2)
    s3 = np.float32(np.float32(1.0) * idf + np.float32(0.5) * idf)       # ascending term hash: 10 then 20
    s515 = np.float32(2.0) * idf
    # doc 3 and doc 515 share bucket 3 (515 % 512): the strictly greater score stays -> 515 evicts 3 (C13)
    assert ids.tolist() == [515, 7] and sc.tolist() == [float(s515), float(np.float32(0.25) * idf)]
    assert s515 > s3
    fi, fs = O.rrf_fuse(np.array([5, 6, 7], np.uint32), np.array([7, 8], np.uint32), 60.0, 3)
    eps = np.float32(1.1920929e-07)
    r = lambda rank: np.float32(1.0) / np.float32(np.float32(rank) + np.float32(60.0) + eps)
    exp = {5: r(0), 6: r(1), 7: np.float32(r(2) + r(0)), 8: r(1)}
    order = sorted(exp.items(), key=lambda kv: (kv[1], kv[0]), reverse=True)[:3]
    assert fi.tolist() == [k for k, _ in order] and fs.tolist() == [float(v) for _, v in order]


def test_auto_quantization_sampler_thresholds():
    # indexes/hnsw/mod.rs:202-351: first threshold whose tail holds <= clamp_margin_percent of the sampled values
    r = np.random.default_rng(9)
    x = r.standard_normal((400, 96)).astype(np.float32) * 0.06
    lo, hi = O.sample_values_range(x, 1.0)
    T = [0.025, 0.05, 0.1, 0.2, 0.3, 0.4, 0.5]
    tot = np.float32(x.size)
    exp_hi = next((t for t in T if np.float32(np.float32((x > np.float32(t)).sum()) / tot) * np.float32(100) <= np.float32(1.0)), 1.0)
    exp_lo = next((-t for t in T if np.float32(np.float32((x < -np.float32(t)).sum()) / tot) * np.float32(100) <= np.float32(1.0)), -1.0)
    assert (lo, hi) == (np.float32(exp_lo), np.float32(exp_hi)) and hi in (np.float32(0.1), np.float32(0.2))
    assert O.sample_values_range(r.uniform(-1, 1, (100, 768)).astype(np.float32)) == (-1.0, 1.0)   # tests/test.py data
    assert O.sample_values_range(np.zeros((10, 8), np.float32)) == (np.float32(-0.025), np.float32(0.025))
    xs = np.zeros((1, 200), np.float32); xs[0, :2] = 0.7                                               # exactly 1 % above 0.5
    assert O.sample_values_range(xs, 1.0)[1] == np.float32(0.025)
    xs[0, 2] = 0.7                                                                                      # 1.5 % above every threshold
    assert O.sample_values_range(xs, 1.0)[1] == 1.0


def test_zero_raw_vector_gets_x86_negative_nan_and_ranks_last():
    """finalize_ann_results (vector_store.rs:425-427) divides by |q|*|raw| unchecked: 0/0 on x86-64 — the platform of the
    reference's AVX2 paths — is the negative default NaN, which f32::total_cmp orders below every number.  The oracle gets
    this from the host FPU; the device canonicalises to the same bits (x86_div in cosdata_amd/csrc/dot_engines.h)."""
    import platform
    if platform.machine() not in ("x86_64", "AMD64"):
        pytest.skip("the NaN sign of an invalid operation is the x86 convention")
    from tests import helpers as H
    X = H.clustered_corpus(1200, 48, n_centers=5, seed=4)
    X[[7, 500]] = 0.0
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=2, ef_construction=32, ef_search=64)
    Q = H.queries_from(X, 6, seed=9)
    ids, sc, cnt = oix.search_batch(Q, 300, threads=2)[:3]
    seen = 0
    for b in range(6):
        c = int(cnt[b])
        zpos = [j for j in range(c) if int(ids[b, j]) in (7, 500)]
        seen += len(zpos)
        assert all(j >= c - len(zpos) for j in zpos)                       # they close the list
        assert all(sc[b, j:j + 1].view(np.uint32)[0] == 0xFFC00000 for j in zpos)
        finite = sc[b, :c - len(zpos)]
        assert np.all(np.isfinite(finite)) and np.all(finite[:-1] >= finite[1:])
    assert seen > 0


def test_round_synchronous_link_prototype():
    """the round-synchronous link schedule (oracle prototype for the device link phase) is deterministic,
    links every node, and yields the same graph quality as the id-order replay of coso_index_build_batched."""
    from tests import helpers as H
    X = H.clustered_corpus(4000, 32, n_centers=40, seed=3)
    Q = H.queries_from(X, 64, seed=5)
    gt, _ = O.bruteforce_topk(X, Q, 10, threads=2)
    kw = dict(dim=32, num_layers=4, ef_construction=48, ef_search=48, seed=9)

    def recall(ix):
        ids = ix.search_batch(Q, 10, threads=2)[0]
        return float(np.mean([len(set(ids[i]) & set(gt[i])) / 10 for i in range(len(Q))]))

    base = recall(O.OracleIndex(O.HNSWParams(**kw)).set_vectors(X).build_batched(128))
    for greedy in (False, True):
        a, sa = O.OracleIndex(O.HNSWParams(**kw)).set_vectors(X).build_rounds(128, greedy=greedy)
        b, sb = O.OracleIndex(O.HNSWParams(**kw)).set_vectors(X).build_rounds(128, greedy=greedy)
        assert sa == sb
        for (ia, na), (ib, nb) in zip(a.export_graph(), b.export_graph()):
            assert np.array_equal(ia, ib) and np.array_equal(na, nb)              # deterministic
        assert sa["nodes"] >= 4000 and sa["rounds"] >= sa["batch_levels"] and sa["first_round_nodes"] <= sa["nodes"]
        assert abs(recall(a) - base) <= 0.02, (recall(a), base)
        deg = np.mean((a.export_graph()[0][1] != 0xFFFFFFFD).sum(axis=1))
        assert deg > 16


def test_streamed_quantize_and_raw_subset_equal_the_full_table_path():
    """bench.py --workload c4shard cannot hold the raw f32 corpus on the host: the oracle then quantizes streamed chunks
    (coso_index_quantize_rows) and reranks from a subset of raw rows (coso_index_set_raw_subset).  Same answers, bit for bit."""
    from tests import helpers as H
    X = H.clustered_corpus(3000, 64, seed=1)
    oix = H.oracle_index(X, num_layers=3, ef_construction=32, ef_search=32)
    Q = H.queries_from(X, 50)
    want = oix.search_batch(Q, 10, threads=2)
    o2 = O.OracleIndex(O.HNSWParams(dim=64, num_layers=3, ef_construction=32, ef_search=32)).alloc_vectors(3000)
    for s in range(0, 3000, 700):
        o2.quantize_rows(s, X[s:s + 700])
    o2.import_graph(oix.export_graph(), oix.root_raw())
    assert np.array_equal(o2.codes(), oix.codes()) and np.array_equal(o2.mags().view(np.uint32), oix.mags().view(np.uint32))
    ids, cnt = o2.candidates_batch(Q, 10, threads=2)
    assert (cnt <= 50).all() and (cnt > 0).all()
    u = np.unique(ids[ids != 0xFFFFFFFF])
    o2.set_raw_subset(u, X[u])
    got = o2.search_batch(Q, 10, threads=2)
    assert all(np.array_equal(a, b) for a, b in zip(want, got))
    o2.set_raw_subset(u[:-1], X[u[:-1]])                      # a candidate without its raw row is an error, never a guess
    with pytest.raises(ValueError):
        o2.search_batch(Q, 10, threads=2)
