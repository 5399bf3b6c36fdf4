"""Model check (CPU, pure Python) of the claims DESIGN.md makes about the walk kernels' commit step:

(a) the bounded sorted pool with the `rank < limit` insertion rule pops exactly what the reference's unbounded BinaryHeap pops
    (vector_store.rs:1125-1192) — the kernels' basic premise;
(b) the winner pre-screen (one compare against the pool entry at position limit - 1) never drops a key the loop would insert;
(b') the latency kernel's speculation: every winner of a window entry at commit time was unvisited under the filter at the start
    of the round, so its similarity has been computed;
(c) the data-parallel formulation of a window commit that DESIGN.md §10 proposes for a several-waves-per-query kernel — first
    occurrence per filter bit in (entry, slot) order, commit horizon = first entry whose best winner beats the last waiting window
    entry, pool := top-C of (pool minus the popped heads) ∪ (winners of the committed entries) — leaves the walk in a state that is
    indistinguishable from the sequential commit the shipped kernels run: same pops, same filter, same pool wherever it can still
    be popped, same window size next round.

The model is the kernels' round structure, not a full index: random graphs, random distinct similarities, the lossy PerformantFixedSet
replica (bit = id & (64 M - 1)) so that aliasing between slots and between window entries happens all the time."""
import heapq
import random

import pytest

EMPTY = None


def _level(rng, n, M, p_empty=0.1):
    adj = [[(rng.randrange(n) if rng.random() > p_empty else EMPTY) for _ in range(M)] for _ in range(n)]
    keys = list(range(1, n + 1))
    rng.shuffle(keys)  # distinct similarities, larger = better
    return adj, keys


def reference_pops(adj, keys, entry, ef, M):
    """traverse_find_nearest's loop: unbounded max-heap, ef pops, lossy visited filter, slot-order scan"""
    mask = 64 * M - 1
    vis = {entry & mask}
    heap = [(-keys[entry], entry)]
    out = []
    while heap and len(out) < ef:
        _, node = heapq.heappop(heap)
        out.append(node)
        for nb in adj[node]:
            if nb is EMPTY or (nb & mask) in vis:
                continue
            vis.add(nb & mask)
            heapq.heappush(heap, (-keys[nb], nb))
    return out, vis


class Walk:
    """the kernels' state: sorted bounded pool + window rounds"""

    def __init__(self, adj, keys, entry, ef, M, cap, la, prescreen, merge_min=0):
        self.adj, self.keys, self.ef, self.mask, self.cap, self.la, self.prescreen = adj, keys, ef, 64 * M - 1, cap, la, prescreen
        self.vis = {entry & self.mask}
        self.pool = [(keys[entry], entry)]  # descending by key
        self.pops = []
        self.merge_min = merge_min

    def _insert(self, item, limit):
        pos = sum(1 for e in self.pool if e[0] > item[0])
        if pos < limit:
            self.pool.insert(pos, item)
            del self.pool[self.cap:]
            return pos
        return None

    def _winners(self, node, vis):
        """slot-order scan against `vis` (mutated): the lower slot wins an alias"""
        win = []
        for nb in self.adj[node]:
            if nb is EMPTY or (nb & self.mask) in vis:
                continue
            vis.add(nb & self.mask)
            win.append((self.keys[nb], nb))
        return win

    def round_sequential(self):
        kwin = min(len(self.pool), self.la, self.ef - len(self.pops))
        window = self.pool[:kwin]
        # the latency kernel's speculation: similarities are computed up front for every neighbour that is unvisited under the
        # filter as it stands at the START of the round; whoever wins at commit time must be among them
        speculated = {(wi, nb) for wi, (_, node) in enumerate(window) for nb in self.adj[node]
                      if nb is not EMPTY and (nb & self.mask) not in self.vis}
        for wi, (_, node) in enumerate(window):
            assert self.pool[0][1] == node
            self.pool.pop(0)
            self.pops.append(node)
            limit = self.ef - len(self.pops)
            ahead = kwin - 1 - wi
            stale = False
            win = self._winners(node, self.vis)
            assert all((wi, nb) in speculated for _, nb in win)
            if self.prescreen and limit > 0:
                bar = self.pool[limit - 1][0] if limit - 1 < len(self.pool) else 0
                dropped = [w for w in win if not w[0] > bar]
                win = [w for w in win if w[0] > bar]
                for w in dropped:  # (b): the loop would have rejected every one of them
                    assert sum(1 for e in self.pool if e[0] > w[0]) >= limit
            elif self.prescreen:
                win = []
            if self.merge_min and len(win) >= self.merge_min:
                stale = self._merge(win, ahead)
            else:
                for w in win:
                    pos = self._insert(w, limit)
                    if pos is not None and pos < ahead:
                        stale = True
            if stale:
                break

    def _merge(self, win, ahead):
        """walk_kernel.inc commit_merge (round 6): every screened winner is ranked against the pool and against the other winners, each
        pool entry counts the winners below it, then ONE permutation: pool entry j -> j + (winners above it), winner -> (pool entries
        above it) + (winners above it); places past the pool's end are dropped.  Stale window = some winner has fewer than `ahead`
        POOL entries above it.  Empty pool slots are key 0 (below every winner)."""
        W = len(win)
        pool = self.pool + [(0, None)] * (self.cap - len(self.pool))
        below = [0] * self.cap
        rank, above = [], []
        for k, _ in win:
            rank.append(sum(1 for e in pool if e[0] > k))
            above.append(sum(1 for k2, _ in win if k2 > k))
            for j, e in enumerate(pool):
                below[j] += 1 if e[0] > k else 0
        out = [None] * self.cap
        for j, e in enumerate(pool):
            np_ = j + (W - below[j])
            if np_ < self.cap:
                assert out[np_] is None
                out[np_] = e
        for i, w in enumerate(win):
            np_ = rank[i] + above[i]
            if np_ < self.cap:
                assert out[np_] is None
                out[np_] = w
        assert all(o is not None for o in out)                       # every place below the pool's end is written exactly once
        self.pool = [e for e in out if e[1] is not None]
        assert all(o[1] is None for o in out[len(self.pool):])       # empties stay at the end
        return min(rank) < ahead

    def round_parallel(self):
        kwin = min(len(self.pool), self.la, self.ef - len(self.pops))
        window = self.pool[:kwin]
        # first occurrence per filter bit in (entry, slot) order, all entries at once, against the filter as it stands
        first = {}
        for wi, (_, node) in enumerate(window):
            for slot, nb in enumerate(self.adj[node]):
                if nb is EMPTY or (nb & self.mask) in self.vis:
                    continue
                first.setdefault(nb & self.mask, (wi, slot))
        winners = [[] for _ in range(kwin)]
        for wi, (_, node) in enumerate(window):
            for slot, nb in enumerate(self.adj[node]):
                if nb is not EMPTY and first.get(nb & self.mask) == (wi, slot):
                    winners[wi].append((self.keys[nb], nb))
        # commit horizon: the first entry whose best winner beats the last waiting window entry
        last_key = window[-1][0]
        j = kwin - 1
        for wi in range(kwin - 1):
            if winners[wi] and max(w[0] for w in winners[wi]) > last_key:
                j = wi
                break
        for wi in range(j + 1):
            self.pops.append(window[wi][1])
            for _, nb in winners[wi]:
                self.vis.add(nb & self.mask)
        merged = self.pool[j + 1:] + [w for wi in range(j + 1) for w in winners[wi]]
        merged.sort(key=lambda e: -e[0])
        self.pool = merged[:self.cap]

    def run(self, parallel):
        while self.pool and len(self.pops) < self.ef:
            (self.round_parallel if parallel else self.round_sequential)()
        return self.pops


CASES = [(300, 8, 64, 64, 4), (500, 16, 40, 64, 4), (200, 4, 64, 64, 4), (800, 32, 256, 256, 4), (150, 8, 17, 64, 1), (400, 16, 64, 64, 8),
         (60, 8, 64, 64, 4), (1000, 64, 100, 256, 4)]


@pytest.mark.parametrize("n,M,ef,cap,la", CASES)
def test_bounded_pool_with_windows_pops_what_the_reference_heap_pops(n, M, ef, cap, la):
    for seed in range(12):
        rng = random.Random(1000 * n + seed)
        adj, keys = _level(rng, n, M)
        entry = rng.randrange(n)
        want, want_vis = reference_pops(adj, keys, entry, ef, M)
        for prescreen in (False, True):
            w = Walk(adj, keys, entry, ef, M, cap, la, prescreen)
            assert w.run(parallel=False) == want, (seed, prescreen)
            # the filter may differ only by what the reference claimed while expanding its LAST pop's successors that the kernels
            # also claim: both expand exactly the same pops, so the sets are equal
            assert w.vis == want_vis


@pytest.mark.parametrize("n,M,ef,cap,la", CASES)
def test_data_parallel_window_commit_is_indistinguishable_from_the_sequential_one(n, M, ef, cap, la):
    for seed in range(12):
        rng = random.Random(77 * n + seed)
        adj, keys = _level(rng, n, M)
        entry = rng.randrange(n)
        a = Walk(adj, keys, entry, ef, M, cap, la, prescreen=False)
        b = Walk(adj, keys, entry, ef, M, cap, la, prescreen=False)
        while a.pool and len(a.pops) < ef:
            a.round_sequential()
            b.round_parallel()
            assert a.pops == b.pops and a.vis == b.vis, seed
            limit = ef - len(a.pops)
            assert a.pool[:limit] == b.pool[:limit], seed                          # everything that can still be popped
            assert min(len(a.pool), la, limit) == min(len(b.pool), la, limit), seed  # next round's window size
        assert not (b.pool and len(b.pops) < ef)


@pytest.mark.parametrize("n,M,ef,cap,la", CASES + [(600, 64, 128, 128, 4), (900, 64, 100, 128, 4), (500, 32, 200, 256, 4)])
def test_ranked_merge_commit_is_indistinguishable_from_the_serial_insert(n, M, ef, cap, la):
    """round 6: the table levels insert an expansion's screened winners by ONE ranked merge (walk_kernel.inc commit_merge) from
    `merge_min` winners on; the serial loop stays below that.  Same pops, same filter, same pool wherever it can still be popped,
    same stale-window decisions (= the same rounds), same window size next round — and the reference heap's pops at the end."""
    for seed in range(10):
        rng = random.Random(9100 * n + seed)
        adj, keys = _level(rng, n, M, p_empty=0.05)
        entry = rng.randrange(n)
        want, want_vis = reference_pops(adj, keys, entry, ef, M)
        for merge_min in (1, 3):
            a = Walk(adj, keys, entry, ef, M, cap, la, prescreen=True)
            b = Walk(adj, keys, entry, ef, M, cap, la, prescreen=True, merge_min=merge_min)
            rounds = 0
            while a.pool and len(a.pops) < ef:
                a.round_sequential()
                b.round_sequential()
                rounds += 1
                assert a.pops == b.pops and a.vis == b.vis, (seed, merge_min, rounds)   # equal pops after every round = equal stale decisions
                limit = ef - len(a.pops)
                assert a.pool[:limit] == b.pool[:limit], (seed, merge_min)
                assert min(len(a.pool), la, limit) == min(len(b.pool), la, limit), (seed, merge_min)
            assert not (b.pool and len(b.pops) < ef)
            assert b.pops == want and b.vis == want_vis
