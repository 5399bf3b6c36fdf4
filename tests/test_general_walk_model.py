"""The data structure of kernels_walk_general.hip, model-checked on the CPU: ONE array of ef keys holds a level's popped list (positions
[0, popped)) and its candidates (sorted descending behind them, truncated at limit = ef - popped).  It must hand out the same pop
sequence as the reference's unbounded BinaryHeap that is popped while fewer than ef results exist (vector_store.rs:1125-1190), never
write outside its ef slots, and the selection of the best `keep` popped keys must equal a sort of the popped list."""
import heapq
import random

import pytest


def _reference(ef, script):
    """script: per pop, the keys discovered by that expansion (in slot order).  Returns the popped keys in pop order."""
    heap, popped = [], []
    heapq.heappush(heap, -script[0][0])
    step = 1
    while heap and len(popped) < ef:
        popped.append(-heapq.heappop(heap))
        for k in (script[step] if step < len(script) else []):
            heapq.heappush(heap, -k)
        step += 1
    return popped


def _device_model(ef, script):
    arr = [None] * ef                                   # the LDS array: never indexed outside [0, ef)
    arr[0] = script[0][0]
    npool, npop, step = 1, 0, 1
    while npool > 0 and npop < ef:
        npop += 1                                       # the head stays where it is: position npop - 1 of the popped list
        npool -= 1
        limit = ef - npop
        for key in (script[step] if step < len(script) else []):
            if limit == 0:
                continue
            live = arr[npop:npop + npool]
            pos = sum(1 for v in live if v > key)       # (the kernel: a binary search over the descending candidates)
            if pos >= limit:
                continue
            newn = min(npool + 1, limit)
            end = newn - 1                              # candidates [pos, newn - 1) move up by one, the highest 64 first
            while end > pos:
                cnt = min(64, end - pos)
                start = end - cnt
                chunk = arr[npop + start:npop + end]
                arr[npop + start + 1:npop + end + 1] = chunk
                end = start
            arr[npop + pos] = key
            npool = newn
            assert npop + npool <= ef
        step += 1
    return arr[:npop]


@pytest.mark.parametrize("ef", [1, 2, 7, 64, 65, 300, 1500])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_one_array_reproduces_the_heap(ef, seed):
    rng = random.Random(seed * 1000 + ef)
    keys = rng.sample(range(1, 10 ** 7), 40 * ef + 200)  # distinct keys: (similarity, node) pairs of distinct nodes
    it = iter(keys)
    script = [[next(it)]]
    for _ in range(ef + 5):
        n_new = rng.choice([0, 0, 1, 3, 7, 20, 37])      # most pops discover little; some discover a whole row
        script.append([next(it) for _ in range(n_new)])
    want = _reference(ef, script)
    got = _device_model(ef, script)
    assert got == want
    # the level's result: the best `keep` popped keys by repeated selection == a descending sort of the popped list
    keep = 100
    lst = list(got)
    sel = []
    for _ in range(min(keep, len(lst))):
        m = max(lst)
        sel.append(m)
        lst[lst.index(m)] = 0
    assert sel == sorted(got, reverse=True)[:keep]


def test_walk_ends_when_the_candidates_run_dry():
    script = [[50], [40, 30], [], [], [99]]              # the fourth pop finds the array empty: three results, the 99 is never seen
    assert _device_model(10, script) == _reference(10, script) == [50, 40, 30]
