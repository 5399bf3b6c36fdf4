"""Thread-safety of the shared handle (the reference calls search from rayon workers, indexes/mod.rs:268-271) and
host-API request coalescing (dynamic batching): concurrent callers get exactly the answers of serial calls."""
import threading

import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _index():
    X = H.clustered_corpus(8000, 96, n_centers=24, seed=31)
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=4, ef_construction=48, ef_search=64)
    return X, H.device_index_from_oracle(oix, X)


@pytest.mark.parametrize("coalesce", [0, 2048])
def test_concurrent_callers_match_serial(coalesce):
    import cosdata_amd as ca
    X, dix = _index()
    T, per = 12, 48
    Q = H.queries_from(X, T * per, seed=77)
    serial = dix.batch_search(Q, 10)
    dix.set_coalescing(coalesce, 500)
    out = [None] * T
    errs = []

    def worker(t):
        try:
            for _ in range(3):
                out[t] = dix.batch_search(Q[t * per:(t + 1) * per], 10)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for t in range(T):
        assert np.array_equal(out[t][0], serial[0][t * per:(t + 1) * per])
        assert np.array_equal(out[t][1].view(np.uint32), serial[1][t * per:(t + 1) * per].view(np.uint32))
        assert np.array_equal(out[t][2], serial[2][t * per:(t + 1) * per])


def test_coalesced_error_stays_with_its_request():
    import cosdata_amd as ca
    X, dix = _index()
    dix.set_coalescing(4096, 2000)
    Qa = H.queries_from(X, 32, seed=1)
    Qb = H.queries_from(X, 32, seed=2)
    Qb[5, :] = -1.0  # zero-norm u8 code -> CalculationError for THIS request only
    res = {}

    def call(name, q):
        try:
            res[name] = dix.batch_search(q, 5)
        except ca.CosdataError as e:
            res[name] = e

    th = [threading.Thread(target=call, args=("a", Qa)), threading.Thread(target=call, args=("b", Qb))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert isinstance(res["b"], ca.CosdataError) and res["b"].status == 2
    assert not isinstance(res["a"], Exception)
    dix.set_coalescing(0, 0)
    assert np.array_equal(res["a"][0], dix.batch_search(Qa, 5)[0])


def test_big_host_batch_runs_as_a_chunk_pipeline_with_the_answers_of_small_calls():
    """cos_search_batch with B >= 8192 and nobody else on the handle: <= 4 chunks, H2D / walk / finalize of neighbouring chunks
    overlapped on the call's private streams (engine.hip, search_host_pipelined).  Queries are independent: same bits as serial
    small calls, in both visited modes, with a ragged last chunk, and a failing query still fails the call with its status."""
    import cosdata_amd as ca
    X, dix = _index()
    Q = H.queries_from(X, 9001, seed=5)                     # chunks of 2304, 2304, 2304, 2089
    for mode in (ca.VISITED_REF, ca.VISITED_EXACT):
        dix.set_visited_mode(mode)
        small = [dix.batch_search(Q[s:s + 1000], 10) for s in range(0, 9001, 1000)]
        big = dix.batch_search(Q, 10)
        for j in range(3):
            assert np.array_equal(big[j].view(np.uint32), np.concatenate([p[j] for p in small]).view(np.uint32))
        again = dix.batch_search(Q[:8192], 7)                # workspaces re-used with another shape (top_k 7 reranks 35 candidates, not 50)
        assert np.array_equal(again[0], np.concatenate([dix.batch_search(Q[s:s + 1024], 7)[0] for s in range(0, 8192, 1024)]))
    dix.set_visited_mode(ca.VISITED_REF)
    big = dix.batch_search(Q, 10)
    Qbad = Q.copy()
    Qbad[8500, :] = -1.0                                     # zero-norm u8 code in the LAST chunk
    ids, sc, cnt, rc, status = dix.batch_search(Qbad, 10, return_status=True)
    assert rc == 2 and status[8500] == 2 and (np.delete(status, 8500) == 0).all()
    assert np.array_equal(np.delete(ids, 8500, axis=0), np.delete(big[0], 8500, axis=0))


def test_two_big_concurrent_callers_and_thread_churn():
    """two callers with big batches at once (one of them takes the whole-batch path, their copies overlap ACROSS calls), then 40
    short-lived threads: every call leases a pipe from the handle's bounded pool instead of owning a stream per thread id"""
    X, dix = _index()
    Q = H.queries_from(X, 2 * 8192, seed=9)
    serial = dix.batch_search(Q, 10)
    out, errs = [None, None], []

    def big(t):
        try:
            for _ in range(3):
                out[t] = dix.batch_search(Q[t * 8192:(t + 1) * 8192], 10)
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=big, args=(t,)) for t in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for t in range(2):
        for j in range(3):
            assert np.array_equal(out[t][j].view(np.uint32), serial[j][t * 8192:(t + 1) * 8192].view(np.uint32))
    res = {}

    def small(i):
        res[i] = dix.batch_search(Q[i * 16:(i + 1) * 16], 10)[0]
    for wave in range(4):                                    # 4 generations of 10 threads each: thread ids come and go
        th = [threading.Thread(target=small, args=(wave * 10 + i,)) for i in range(10)]
        [t.start() for t in th]
        [t.join() for t in th]
    for i in range(40):
        assert np.array_equal(res[i], serial[0][i * 16:(i + 1) * 16])
