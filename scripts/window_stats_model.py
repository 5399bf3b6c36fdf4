#!/usr/bin/env python3
"""How many entries of its lookahead window does the walk consume per round, and what would gathering table values AHEAD cost?
A CPU model (no device): the reference's traverse_find_nearest restated in Python (as tests/test_oracle_pywalk.py does), run the way
walk_kernel.inc runs it — per round, the adjacency rows of the next LA pool entries are fetched together and entry i + 1 is consumed
only while no candidate has been inserted ahead of it — on a graph built by the oracle.  Per level class (upper levels = what the
level table covers, level 0 / 1 = rows) it prints: expansions per round, the histogram of entries consumed per round, and for the
gather-ahead candidate (kernels_walk_spec.hip) with N entries ahead: dependent memory round trips per round before / after, and the
table gathers issued per gather used.  It also asserts the candidate's premise: the winners of an entry at the time it is consumed are
a subset of the neighbours the filter did not hold when the window was fetched.
usage: window_stats_model.py [n_vectors=20000] [dim=32] [queries=64] [ef=64] [LA=4]"""
import heapq
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from oracle import oracle as O  # noqa: E402
from tests import helpers as H  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 32
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 64
ef = int(sys.argv[4]) if len(sys.argv) > 4 else 64
LA = int(sys.argv[5]) if len(sys.argv) > 5 else 4
M, M0, L = 32, 64, 6

X = H.clustered_corpus(n, dim, n_centers=max(8, n // 1000), seed=3)
p = O.HNSWParams(dim=dim, storage=O.STORAGE_U8, resolution=0, num_layers=L, neighbors_count=M, level0_neighbors_count=M0, ef_construction=64, ef_search=ef,
                 shortlist_size=64, visited_mode=O.VISITED_REF, seed=5)
ix = O.OracleIndex(p).set_vectors(X).build()
levels = ix.export_graph()
codes, mags = O.quantize_batch(np.vstack([X, ix.root_raw()[None, :]]), O.STORAGE_U8, 0, -1.0, 1.0)
cf = codes.astype(np.float32)
Q = H.queries_from(X, nq, seed=11)
pos = [{int(v): i for i, v in enumerate(ids.tolist())} for ids, _ in levels]

stats = {}   # level class -> counters


def bump(cls, key, v=1):
    stats.setdefault(cls, {}).setdefault(key, 0)
    stats[cls][key] += v


for q in Q:
    qcode, qmag = O.quantize(q, O.STORAGE_U8, 0, -1.0, 1.0)
    qf = qcode.astype(np.float32)
    sims_all = (cf @ qf) / (np.float32(qmag) * mags + 1e-30)     # a stand-in for the exact u8 cosine: the ORDER of candidates is what matters here
    row = lambda nid: n if nid == O.ROOT_ID else nid
    entry = O.ROOT_ID
    for level in range(L, -1, -1):
        ids, nbr = levels[level]
        Ml = M0 if level == 0 else M
        bits = 64 * Ml
        key = lambda v: v & (bits - 1)
        visited = {key(O.QUERY_ID), key(entry)}
        heap = [(-float(sims_all[row(entry)]), -entry, entry)]
        npop = 0
        best = None
        cls = "level 0" if level == 0 else ("level 1" if level == 1 else "upper levels")
        while heap and npop < ef:
            # the window: the next LA pool entries in pop order, and what the filter holds NOW (the gather-ahead premise)
            window = heapq.nsmallest(min(LA, len(heap), ef - npop), heap)
            filter_at_fetch = set(visited)
            consumed = 0
            bump(cls, "rounds")
            for wi, ent in enumerate(window):
                top = heapq.heappop(heap)
                assert top == ent                                  # still provably the next pop
                node = ent[2]
                npop += 1
                consumed += 1
                if best is None or ent < best:
                    best = ent
                bump(cls, "expansions")
                ahead = window[wi + 1:]
                window_ok = True
                tentative = winners = 0
                for j in range(min(Ml, 64)):
                    nid = int(nbr[pos[level][node], j])
                    if nid == O.SLOT_EMPTY:
                        continue
                    if key(nid) not in filter_at_fetch:
                        tentative += 1
                    if key(nid) in visited:
                        continue
                    assert key(nid) not in filter_at_fetch         # winner => was unvisited when the window was fetched
                    visited.add(key(nid))
                    winners += 1
                    cand = (-float(sims_all[row(nid)]), -nid, nid)
                    heapq.heappush(heap, cand)
                    if ahead and cand < ahead[-1]:
                        window_ok = False                          # inserted among the window entries still waiting (pos < ahead in the kernel)
                bump(cls, "winners", winners)
                bump(cls, f"tentative_entry{wi}", tentative)
                bump(cls, f"consumed_entry{wi}")
                if not window_ok:
                    break
            bump(cls, f"k={consumed}")
            for wi in range(consumed, len(window)):                # fetched with the window, not consumed: their gathers are the waste
                node = window[wi][2]
                t = sum(1 for j in range(min(Ml, 64)) if int(nbr[pos[level][node], j]) != O.SLOT_EMPTY and key(int(nbr[pos[level][node], j])) not in filter_at_fetch)
                bump(cls, f"wasted_tentative_entry{wi}", t)
        entry = best[2] if best else entry

out = {"config": {"vectors": n, "dim": dim, "queries": nq, "ef": ef, "LA": LA, "M": M, "M0": M0, "levels": L + 1}}
for cls, s in stats.items():
    r, e = s["rounds"], s["expansions"]
    rec = {"rounds": r, "expansions": e, "expansions_per_round": e / r, "winners_per_expansion": s["winners"] / e,
           "consumed_per_round_histogram": {k_: s[k_] / r for k_ in sorted(s) if k_.startswith("k=")}}
    for N in (2, 4, 8):
        if N > LA:
            continue
        # dependent round trips per round on a table level: now 1 (adjacency) + one per consumed entry; with N ahead 1 + 1 + one per consumed entry beyond N
        beyond = sum(s.get(f"consumed_entry{wi}", 0) for wi in range(N, LA))
        used = sum(s.get(f"tentative_entry{wi}", 0) for wi in range(N))
        wasted = sum(s.get(f"wasted_tentative_entry{wi}", 0) for wi in range(N))
        rec[f"gather_ahead_{N}"] = {"round_trips_per_round_now": 1 + e / r, "round_trips_per_round_then": 2 + beyond / r,
                                    "gathers_issued_per_winner_gather_now": (used + wasted + sum(s.get(f"tentative_entry{wi}", 0) for wi in range(N, LA))) / max(1, s["winners"]),
                                    "share_of_ahead_gathers_for_unconsumed_entries": wasted / max(1, used + wasted)}
    out[cls] = rec
print(json.dumps(out, indent=1))
