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
