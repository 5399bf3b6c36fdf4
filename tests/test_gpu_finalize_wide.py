"""GPU: small launches finalize with eight waves per query (kernels_walk.hip finalize_one / rerank_wide: wave 0 selects the 5 x top_k
candidates, every wave dots its share of their raw rows with the eight-lane chains, wave 0 sorts).  The one-wave kernels stay for
big launches: both must return the same bits as each other and as the oracle (replace_with_exact_distances, vector_store.rs:404-445),
for lists that fit one register per lane and lists that do not, the screened kernel and the general one."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _bits(res):
    return [np.atleast_1d(np.asarray(x, dtype=np.int64 if np.isscalar(x) else None)).view(np.uint32).copy() for x in res]


@pytest.mark.parametrize("dim,metric,top_k", [(96, O.METRIC_COSINE, 10), (96, O.METRIC_COSINE, 1), (768, O.METRIC_COSINE, 30),
                                              (100, O.METRIC_DOT, 13), (64, O.METRIC_COSINE, 100), (40, O.METRIC_COSINE, 250)])
def test_wide_finalize_keeps_every_bit(dim, metric, top_k):
    from cosdata_amd import _lib
    n = 5000 if dim < 512 else 2500
    X = H.clustered_corpus(n, dim, n_centers=20, seed=77 + dim)
    oix = H.oracle_index(X, O.STORAGE_U8, 0, num_layers=4, ef_construction=48, ef_search=64, metric=metric)
    dix = H.device_index_from_oracle(oix, X)
    B = 150
    Q = H.queries_from(X, B, noise=0.05, seed=9)
    Q[7] = 0.0                                             # a zero query: CalculationError for cosine, an ordinary query for the dot product
    with _lib.tuning(finalize_wide_max_b=0):
        one_wave = dix.batch_search(Q, top_k, return_status=True)
    wide = dix.batch_search(Q, top_k, return_status=True)
    for a, b in zip(_bits(one_wave), _bits(wide)):
        assert np.array_equal(a, b)
    with _lib.tuning(finalize_fast=0):                     # the general kernel alone, wide
        wide_general = dix.batch_search(Q, top_k, return_status=True)
    for a, b in zip(_bits(one_wave), _bits(wide_general)):
        assert np.array_equal(a, b)
    ids, sc, cnt = wide[:3]
    sample = np.array([b for b in range(0, B, 11) if b != 7])
    oids, osc, ocnt = oix.search_batch(Q[sample], top_k, threads=4)[:3]
    assert np.array_equal(cnt[sample], ocnt)
    for j, b in enumerate(sample):
        c = int(ocnt[j])
        assert np.array_equal(ids[b, :c], oids[j, :c]), f"query {b}"
        assert np.array_equal(sc[b, :c].view(np.uint32), osc[j, :c].view(np.uint32)), f"query {b}"
