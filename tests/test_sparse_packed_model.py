"""The packed posting layout of the learned-sparse index (kernels_sparse.hip: sparse_packed_kernel, the default layout
since round 5) restated in numpy with the kernel's own 32-bit arithmetic — the packed word, the wrap of postings that belong
to other tiles and of the zeros a step's buffer descriptor returns past its end, the dummy slots, the touch count next to the sum,
the host's overflow bound, table windows / term groups / 512-posting steps — and checked against the oracle.  This pins the ALGORITHM on the CPU (it was
written in round 4 without a device at hand); tests/test_sparse.py pins the code on the GPU, both layouts."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.test_sparse import _corpus, _queries

STILE, SDIR_MIN, SLICES, STEP, SEL = 8192, 256, 256, 512, 64
SPK_CNT, SPK_SUM = 1 << 22, (1 << 22) - 1
U32 = np.uint32


def _layout(dims, key_off, vec_ids, n, bits):
    """cos_sparse_create: per dimension ONE id-sorted list of packed words key << 24 | (id + 1), tile directory for long lists, multiplicity"""
    Q = 1 << bits
    ko = np.asarray(key_off, np.int64).reshape(len(dims), Q + 1)
    n_tiles = (n + STILE - 1) // STILE
    pk = np.zeros(int(ko[-1, Q]), U32)
    lists = []
    for t in range(len(dims)):
        b, e = int(ko[t, 0]), int(ko[t, Q])
        keys = np.repeat(np.arange(Q), np.diff(ko[t]))
        ids = np.asarray(vec_ids[b:e], np.int64)
        order = np.lexsort((keys, ids))                       # sort by (id, key): tmp = id << 8 | key
        ids, keys = ids[order], keys[order]
        pk[b:e] = (keys.astype(np.int64) << 24 | (ids + 1)).astype(U32)
        mult = 1
        if len(ids):
            _, counts = np.unique(ids, return_counts=True)
            mult = int(counts.max())
        tdir = None
        if e - b > SDIR_MIN:
            tdir = np.searchsorted(ids, np.arange(n_tiles + 1) * STILE, side="left")
        lists.append((b, e, tdir, mult))
    return pk, lists, n_tiles


def _terms(dims, lists, key_off, bits, upper, thr, qd, qv):
    """cos_sparse_search_batch's host part: find_node, quantize, the early-termination rule, the counted-accumulator bound"""
    Q = 1 << bits
    ko = np.asarray(key_off, np.int64).reshape(len(dims), Q + 1)
    etv = min(np.float32(Q) * np.float32(thr), np.float32(255.0))
    early = 0 if not (etv == etv) or etv <= 0 else int(etv)
    low_f = np.float32(thr) * np.float32(Q)
    low = 0 if not (low_f == low_f) or low_f <= 0 else int(low_f)
    terms, sum_bound, touch_bound = [], 0, 0
    pos = {int(d): i for i, d in enumerate(dims)}
    for d, v in zip(qd, qv):
        t = pos.get(int(d))
        if t is None:
            continue
        qq = O.sparse_quantize(v, upper, bits)
        k0 = 0 if qq > low else early
        if k0 >= Q or ko[t, Q] == ko[t, 0]:
            continue
        terms.append((t, qq, k0))
        sum_bound += lists[t][3] * qq * (Q - 1)
        touch_bound += lists[t][3]
    return terms, (sum_bound < SPK_CNT and touch_bound <= 1023)


def _block(pk, lists, terms, counted, n_tiles, split, splits):
    """one workgroup of sparse_packed_kernel: tiles split, split + splits, ...; returns its (similarity + 1) << 32 | id keys"""
    nt = len(terms)
    my_tiles = (n_tiles - split + splits - 1) // splits
    TW = min(nt, SLICES)
    TB = min(SLICES // nt, my_tiles) if nt < SLICES else 1
    lane = np.arange(64, dtype=np.int64)
    keys = []
    acc = np.zeros(STILE + 64, np.int64)                    # u32 in the kernel; the model asserts that nothing overflows
    zflag = np.zeros(STILE, bool)
    for ti0 in range(0, my_tiles, TB):
        tbn = min(TB, my_tiles - ti0)
        for tw0 in range(0, nt, TW):
            twn = min(TW, nt - tw0)
            table = {}
            for p in range(tbn * twn):                      # the slice table of the window
                tile, t = split + (ti0 + p // twn) * splits, tw0 + p % twn
                b, e, tdir, _ = lists[terms[t][0]]
                if tdir is not None:
                    b, e = b + int(tdir[tile]), b + int(tdir[tile + 1])
                table[p] = (b, e - b)
            for tj in range(tbn):
                tile = split + (ti0 + tj) * splits
                d0 = tile * STILE
                d1 = U32(d0 + 1)
                for g in range((twn + 63) // 64):
                    tb = g * 64
                    ng = min(64, twn - tb)
                    for tl in range(ng):                    # steps of the group, term by term (their order does not matter: adds commute)
                        b, ln = table[tj * twn + tb + tl]
                        _, qq, k0 = terms[tw0 + tb + tl]
                        for done in range(0, ln, STEP):
                            step_len = min(STEP, ln - done)
                            for u in range(8):
                                idx = lane + 64 * u
                                p = np.where(idx < step_len, pk[np.minimum(b + done + idx, len(pk) - 1)], U32(0)).astype(U32)   # past the step's end: 0
                                rel = (p - d1).astype(U32)                                       # wraps like the kernel's v_sub
                                key = (rel >> U32(24)).astype(np.int64)
                                slot24 = (rel & U32(0xFFFFFF)).astype(np.int64)
                                dummy = STILE + lane
                                if counted:
                                    at = np.where(key >= k0, slot24, dummy)
                                    at = np.minimum(at, dummy)
                                    np.add.at(acc, at, qq * key + SPK_CNT)
                                else:
                                    ok = (slot24 < STILE) & (key >= k0)
                                    w = qq * key
                                    np.add.at(acc, np.where(ok, slot24, dummy), np.where(ok, w, 0))
                                    zflag[slot24[ok & (w == 0)]] = True
                if tw0 + TW < nt:
                    continue
                a = acc[:STILE]
                assert a.max(initial=0) < (1 << 32)
                if counted:
                    reached = a != 0
                    sim = a & SPK_SUM
                    assert ((a >> 22) <= 1023).all()
                else:
                    reached = (a != 0) | zflag
                    sim = a
                ids = d0 + np.nonzero(reached)[0]
                keys += [((int(s) + 1) << 32) | int(i) for s, i in zip(sim[reached], ids)]
                acc[:STILE] = 0
                zflag[:] = False
    return keys


def _model_search(dims, key_off, vec_ids, n, bits, upper, thr, qd, qv, k, splits):
    pk, lists, n_tiles = _layout(dims, key_off, vec_ids, n, bits)
    terms, counted = _terms(dims, lists, key_off, bits, upper, thr, qd, qv)
    if not terms:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), counted
    keys = []
    for split in range(min(splits, n_tiles)):
        keys += sorted(_block(pk, lists, terms, counted, n_tiles, split, splits), reverse=True)[:SEL]   # a block's pool keeps its best 64
    keys = sorted(keys, reverse=True)[:k]
    return np.array([x & 0xFFFFFFFF for x in keys], np.int64), np.array([(x >> 32) - 1 for x in keys], np.int64), counted


@pytest.mark.parametrize("bits,thr,seed", [(6, 0.0, 1), (4, 0.5, 2), (8, 0.3, 3)])
def test_model_of_the_packed_kernel_matches_the_oracle(bits, thr, seed):
    upper, n = 3.0, 20000
    rows, dims, key_off, vec_ids, row_off, raw_dims, raw_vals = _corpus(n=n, vocab=300, nnz=14, bits=bits, upper=upper, seed=seed)
    for q in _queries(12, 300, seed=seed + 10):
        for splits in (1, 3):
            ids, sims, counted = _model_search(dims, key_off, vec_ids, n, bits, upper, thr, q[0], q[1], 20, splits)
            cand, osims = O.sparse_search(dims, key_off, vec_ids, n, bits, upper, thr, q[0], q[1], k_with_reranking=20)
            assert np.array_equal(ids, np.asarray(cand[:20], np.int64)), (bits, thr, splits)
            assert np.array_equal(sims, np.asarray(osims[:20], np.int64))


def test_model_long_queries_flag_blocks_and_repeated_ids():
    """more than 64 and more than 256 terms (groups, table windows), large 8-bit weights (the bound fails: flag-word blocks), a CSR
    that repeats ids inside a dimension, a last tile that is not full, ids next to the 24-bit wrap of other tiles"""
    bits, upper, n, vocab = 8, 3.0, 3 * STILE + 777, 400
    rows, dims, key_off, vec_ids, row_off, raw_dims, raw_vals = _corpus(n=n, vocab=vocab, nnz=30, bits=bits, upper=upper, seed=9)
    ko = np.asarray(key_off).reshape(len(dims), (1 << bits) + 1).copy()
    lo, hi, at = int(ko[5, 40]), int(ko[5, 41]), int(ko[5, 90])
    extra = np.asarray(vec_ids[lo:hi]).copy()
    vec_ids = np.concatenate([vec_ids[:at], extra, vec_ids[at:]])
    flat = ko.ravel()
    flat[5 * ((1 << bits) + 1) + 90:] += len(extra)
    key_off = flat.astype(np.uint64)
    rng = np.random.default_rng(4)
    seen_counted = set()
    for m, scale in ((65, 0.05), (130, 2.9), (300, 0.02), (270, 2.5), (3, 0.4)):
        d = rng.choice(vocab, size=m, replace=False).astype(np.uint32)
        if m == 3:
            d[0] = dims[5]
        v = (scale * (0.5 + rng.random(m))).astype(np.float32)
        for thr in (0.0, 0.4):
            ids, sims, counted = _model_search(dims, key_off, vec_ids, n, bits, upper, thr, d, v, 16, 2)
            seen_counted.add(counted)
            cand, osims = O.sparse_search(dims, key_off, vec_ids, n, bits, upper, thr, d, v, k_with_reranking=16)
            assert np.array_equal(ids, np.asarray(cand[:16], np.int64)), (m, scale, thr)
            assert np.array_equal(sims, np.asarray(osims[:16], np.int64))
    assert seen_counted == {True, False}
