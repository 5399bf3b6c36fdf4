"""GPU: cos_index_append — vector_store::index_embeddings called again on a live index (vector_store.rs:714-780, index_embedding
:782-975): the new vectors are inserted into the RESIDENT graph by the schedule of cos_index_build continued.  The oracle's
coso_index_append_vectors + coso_index_build_rounds_continue is the CPU statement: per-level adjacency must be identical, slot for slot."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _same_graph(a, b):
    assert len(a) == len(b)
    for l, ((ia, na), (ib, nb)) in enumerate(zip(a, b)):
        assert np.array_equal(ia, ib), f"level {l}: node ids differ"
        assert np.array_equal(na, nb), f"level {l}: {int((na != nb).any(axis=1).sum())} rows differ"


def _device(X, p, visited=0, **kw):
    import cosdata_amd as ca
    st = ca.StorageType(ca.StorageKind(p.storage), p.resolution)
    hp = ca.HNSWHyperParams(num_layers=p.num_layers, ef_construction=p.ef_construction, ef_search=p.ef_search,
                            level_0_neighbors_count=p.level0_neighbors_count, neighbors_count=p.neighbors_count)
    dix = ca.HNSWIndex(p.dim, hp, ca.DistanceMetric(p.metric), st, (p.range_lo, p.range_hi), p.shortlist_size, seed=p.seed, visited_mode=visited)
    return dix.upload_vectors(X)


@pytest.mark.parametrize("storage,res,dim,n0,adds,batch,m0,m", [
    (O.STORAGE_U8, 0, 96, 3000, (1, 700, 1300), 256, 64, 32),
    (O.STORAGE_U8, 0, 768, 1500, (600,), 128, 64, 32),
    (O.STORAGE_SUBBYTE, 2, 128, 2000, (500, 500), 64, 32, 16),
    (O.STORAGE_F32, 0, 48, 1200, (400,), 1, 16, 8),          # batches of one = the sequential reference order
    (O.STORAGE_U8, 0, 64, 2500, (2500,), 512, 128, 64),
])
def test_append_equals_the_oracles_continued_build(storage, res, dim, n0, adds, batch, m0, m):
    total = n0 + sum(adds)
    X = H.clustered_corpus(total, dim, n_centers=20, seed=dim + n0)
    p = O.HNSWParams(dim=dim, storage=storage, resolution=res, num_layers=4, ef_construction=48, ef_search=40, seed=11,
                     level0_neighbors_count=m0, neighbors_count=m)
    oix = O.OracleIndex(p).set_vectors(X[:n0])
    oix.build_rounds(batch)
    dix = _device(X[:n0], p).build(batch)
    _same_graph(dix.download_graph(), oix.export_graph())
    at = n0
    for a in adds:
        oix.append(X[at:at + a], batch)
        dix.append(X[at:at + a], batch)
        at += a
        assert dix.n == at and dix.level_count(0) == at + 1
        _same_graph(dix.download_graph(), oix.export_graph())
    # the appended index searches like the oracle's (ids, score bits), incl. queries near the NEW vectors
    Q = H.queries_from(X[n0:], 300, noise=0.05, seed=3)
    ids, sc, cnt = dix.batch_search(Q, 10)
    oids, osc, ocnt = oix.search_batch(Q, 10, threads=4)[:3]
    assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))
    assert (ids >= n0).any()                                    # new vectors are found
    if storage != O.STORAGE_SUBBYTE:   # (the reference's quaternary walk ranks by its plane-order quirk, SURVEY App. C #4: ID parity is the bar there)
        gt, _ = O.bruteforce_topk(X[:at], Q, 10, threads=4)
        recall = np.mean([len(set(ids[i]) & set(gt[i])) / 10 for i in range(len(Q))])
        assert recall > 0.85, recall


def test_append_on_a_batch_boundary_equals_the_full_build():
    """the schedule's batches are min(B, max(1, inserted / 4)): a build that stops on one of those boundaries and is then continued IS
    the full build (same level draws, same batches, same snapshots)"""
    dim, B = 64, 200
    bounds, ins = [], 0
    while ins < 4000:
        ins += min(B, max(1, ins // 4))
        bounds.append(ins)
    n0 = [b for b in bounds if b >= 1500][0]
    n1 = [b for b in bounds if b >= 3000][0]
    X = H.clustered_corpus(n1, dim, n_centers=16, seed=5)
    p = O.HNSWParams(dim=dim, num_layers=4, ef_construction=40, ef_search=32, seed=3)
    full = _device(X, p).build(B)
    part = _device(X[:n0], p).build(B)
    part.append(X[n0:], B)
    _same_graph(part.download_graph(), full.download_graph())


def test_append_device_rows_and_refusals():
    import torch
    import cosdata_amd as ca
    dim, n0, m = 96, 2000, 800
    X = H.clustered_corpus(n0 + m, dim, n_centers=16, seed=9)
    p = O.HNSWParams(dim=dim, num_layers=4, ef_construction=40, ef_search=32, seed=2)
    oix = O.OracleIndex(p).set_vectors(X[:n0])
    oix.build_rounds(128)
    oix.append(X[n0:], 128)
    Xd = torch.from_numpy(X).cuda()
    hp = ca.HNSWHyperParams(num_layers=4, ef_construction=40, ef_search=32)
    dix = ca.HNSWIndex(dim, hp, ca.DistanceMetric.Cosine, ca.StorageType.UnsignedByte(), (p.range_lo, p.range_hi), seed=2)
    dix.upload_vectors_device(Xd.data_ptr(), n0, keepalive=Xd).build(128)
    with pytest.raises(ca.CosdataError):
        dix.append(X[n0:], 128)                                  # the index borrows its rows: host rows are refused
    dix.append_device(Xd.data_ptr(), m, 128, keepalive=Xd)
    _same_graph(dix.download_graph(), oix.export_graph())
    Q = H.queries_from(X, 200, noise=0.05, seed=4)
    ids, sc, cnt = dix.batch_search(Q, 10)
    oids, osc, _ = oix.search_batch(Q, 10, threads=4)[:3]
    assert np.array_equal(ids, oids) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))
    # no link state -> NotReady: an uploaded graph, a released state
    up = H.device_index_from_oracle(oix, X)
    with pytest.raises(ca.CosdataError) as e:
        up.append(X[:10])
    assert e.value.status == ca._lib.ERR_NOT_READY
    dix.release_link_state()
    with pytest.raises(ca.CosdataError) as e:
        dix.append_device(Xd.data_ptr(), 1)
    assert e.value.status == ca._lib.ERR_NOT_READY
    ids2, sc2, _ = dix.batch_search(Q, 10)                       # the graph is untouched by the refusals
    assert np.array_equal(ids2, ids) and np.array_equal(sc2.view(np.uint32), sc.view(np.uint32))


@pytest.mark.parametrize("storage,res,dim,n,m0,m,visited", [
    (O.STORAGE_U8, 0, 96, 3000, 64, 32, 0),
    (O.STORAGE_U8, 0, 768, 1500, 64, 32, 1),
    (O.STORAGE_SUBBYTE, 2, 128, 2000, 32, 16, 0),
    (O.STORAGE_F32, 0, 48, 1200, 16, 8, 0),
])
def test_delete_equals_the_oracles_delete_embedding(storage, res, dim, n, m0, m, visited):
    """cos_index_delete == coso_index_delete (delete_embedding, vector_store.rs:1206-1400), graph for graph after every chunk of ids;
    a deleted id is never returned again; append after delete continues from the same link state"""
    X = H.clustered_corpus(n + 300, dim, n_centers=20, seed=dim + n)
    p = O.HNSWParams(dim=dim, storage=storage, resolution=res, num_layers=4, ef_construction=48, ef_search=40, seed=21,
                     level0_neighbors_count=m0, neighbors_count=m, visited_mode=visited)
    oix = O.OracleIndex(p).set_vectors(X[:n])
    oix.build_rounds(128)
    dix = _device(X[:n], p, visited=visited).build(128)
    rng = np.random.default_rng(5)
    order = rng.permutation(n)[:240].astype(np.uint32)
    for chunk in (order[:1], order[1:40], order[40:]):
        oix.delete(chunk)
        dix.delete(chunk)
        _same_graph(dix.download_graph(), oix.export_graph())
    Q = X[order[:100]]                                           # the deleted vectors themselves as queries
    ids, sc, cnt = dix.batch_search(Q, 10)
    oids, osc, ocnt = oix.search_batch(Q, 10, threads=4)[:3]
    assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))
    # (a delete whose walk does not reach the node on some level — keep 100 of 512 pops: the quaternary ranking quirk, tiny neighbour lists —
    # leaves it linked there, in the reference too: vector_store.rs:1271-1275 `continue`; the bar is the oracle's answer, asserted above)
    if storage == O.STORAGE_U8:
        assert not np.isin(ids, order).any()
    oix.append(X[n:], 128)
    dix.append(X[n:], 128)
    _same_graph(dix.download_graph(), oix.export_graph())


@pytest.mark.parametrize("storage,res", [(O.STORAGE_U8, 0), (O.STORAGE_F16, 0), (O.STORAGE_SUBBYTE, 3)])
def test_delete_relinks_a_node_left_without_neighbours(storage, res):
    """tiny neighbour lists on a tiny corpus: deleting a node's ONLY neighbour makes delete_embedding link the node again from the walk's
    results (vector_store.rs:1305-1357) — the device runs such a level on the host in the reference's order (the pair distances it needs:
    the operator's device kernel, every storage since round 6)"""
    dim, n = 32, 400
    X = H.clustered_corpus(n, dim, n_centers=40, seed=77) * (0.9 if storage == O.STORAGE_SUBBYTE else 1.0)
    p = O.HNSWParams(dim=dim, storage=storage, resolution=res, num_layers=3, ef_construction=8, ef_search=16, seed=3, level0_neighbors_count=2, neighbors_count=2)
    oix = O.OracleIndex(p).set_vectors(X)
    oix.build_rounds(32)
    dix = _device(X, p).build(32)
    g = oix.export_graph()
    ids0, nbr0 = g[0]
    lonely = [int(nbr0[i][nbr0[i] != O.SLOT_EMPTY][0]) for i in range(n) if (nbr0[i] != O.SLOT_EMPTY).sum() == 1 and nbr0[i][nbr0[i] != O.SLOT_EMPTY][0] < n]
    assert lonely, "the corpus has no node with a single neighbour: pick other hyper-parameters"
    victims = np.array(sorted(set(lonely))[:25], np.uint32)
    before = oix.export_graph()
    oix.delete(victims)
    dix.delete(victims)
    _same_graph(dix.download_graph(), oix.export_graph())
    after = oix.export_graph()[0][1]
    E = O.SLOT_EMPTY
    relinked = [i for i in range(n) if i not in set(victims.tolist()) and (before[0][1][i] != E).sum() == 1
                and before[0][1][i][before[0][1][i] != E][0] in set(victims.tolist()) and (after[i] != E).any()]
    assert relinked, "no orphan was linked again: the slow path was not exercised"


@pytest.mark.parametrize("seed", [1, 2])
def test_random_interleaving_of_appends_and_deletes_follows_the_oracle(seed):
    """a transaction history: appends of random sizes (1 .. 400) and deletes of random live ids (single ids, small groups, ids appended a step
    earlier) in random order — after EVERY operation the device's graph is the oracle's, slot for slot, and at the end so are the searches"""
    rng = np.random.default_rng(1000 + seed)
    dim, n0, total = 64, 800, 3200
    X = H.clustered_corpus(total, dim, n_centers=24, seed=40 + seed)
    p = O.HNSWParams(dim=dim, num_layers=4, ef_construction=40, ef_search=40, seed=seed, level0_neighbors_count=32, neighbors_count=16)
    oix = O.OracleIndex(p).set_vectors(X[:n0])
    oix.build_rounds(96)
    dix = _device(X[:n0], p).build(96)
    at, dead = n0, set()
    for step in range(14):
        if at < total and (rng.random() < 0.6 or at - len(dead) < 50):
            m = int(min(total - at, rng.choice([1, 7, 60, 400])))
            oix.append(X[at:at + m], 96)
            dix.append(X[at:at + m], 96)
            at += m
        else:
            live = np.array(sorted(set(range(at)) - dead), np.uint32)
            k = int(rng.choice([1, 3, 25]))
            ids = rng.choice(live, size=min(k, live.size), replace=False).astype(np.uint32)
            oix.delete(ids)
            dix.delete(ids)
            dead |= set(ids.tolist())
        _same_graph(dix.download_graph(), oix.export_graph())
    Q = H.queries_from(X[:at], 200, noise=0.05, seed=seed)
    ids, sc, cnt = dix.batch_search(Q, 10)
    oids, osc, ocnt = oix.search_batch(Q, 10, threads=4)[:3]
    assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))


@pytest.mark.parametrize("storage,res,dim", [(O.STORAGE_U8, 0, 96), (O.STORAGE_SUBBYTE, 2, 128), (O.STORAGE_F32, 0, 48),
                                             (O.STORAGE_SUBBYTE, 1, 256), (O.STORAGE_SUBBYTE, 3, 96), (O.STORAGE_F16, 0, 64)])
def test_restored_link_state_of_an_uploaded_graph_takes_appends_and_deletes(storage, res, dim):
    """cos_index_restore_link_state == coso_index_restore_link_state: an UPLOADED graph (the reader path, the snapshot path) gets the link
    state the reference has after a reload; appends and deletes on it follow the oracle slot for slot"""
    import cosdata_amd as ca
    n0, m = 1800, 700
    X = H.clustered_corpus(n0 + m, dim, n_centers=16, seed=dim)
    p = O.HNSWParams(dim=dim, storage=storage, resolution=res, num_layers=4, ef_construction=40, ef_search=40, seed=17)
    src = O.OracleIndex(p).set_vectors(X[:n0])
    src.build_rounds(128)
    G, root = src.export_graph(), src.root_raw()
    oix = O.OracleIndex(p).set_vectors(X[:n0]).import_graph(G, root).restore_link_state()
    dix = _device(X[:n0], p)
    dix.upload_graph(G, root)
    with pytest.raises(ca.CosdataError):
        dix.append(X[n0:n0 + 5], 128)                           # an uploaded graph has no link state until it is restored
    dix.restore_link_state()
    oix.append(X[n0:], 128)
    dix.append(X[n0:], 128)
    _same_graph(dix.download_graph(), oix.export_graph())
    dele = np.arange(7, n0 + m, 53, dtype=np.uint32)
    oix.delete(dele)
    dix.delete(dele)
    _same_graph(dix.download_graph(), oix.export_graph())
    Q = H.queries_from(X, 200, noise=0.05, seed=5)
    ids, sc, cnt = dix.batch_search(Q, 10)
    oids, osc, ocnt = oix.search_batch(Q, 10, threads=4)[:3]
    assert np.array_equal(cnt, ocnt) and np.array_equal(ids, oids) and np.array_equal(sc.view(np.uint32), osc.view(np.uint32))
