#!/usr/bin/env python3
"""The reference's calling pattern on the c2 index: N concurrent native callers of ONE 256-query batch each through cos_search_batch
with dynamic batching (cos_index_set_coalescing), PCIe-inclusive.  One JSON line per (callers, max_queries, window_us) with the rate
(best of PROBE_RUNS), what the batching did (cos_index_coalescing_stats) and whether caller 0's answer is the un-coalesced answer.
PROBE_CASES="64:16384:300;128:16384:300;128:32768:300;256:32768:300"."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import cosdata_amd as ca  # noqa: E402
from cosdata_amd import _lib  # noqa: E402

N, D, K, Bc = int(os.environ.get("PROBE_N", 1_000_000)), 768, 10, 256
CASES = [tuple(int(x) for x in c.split(":")) for c in os.environ.get("PROBE_CASES", "64:16384:300;128:16384:300;128:32768:300;256:32768:300").split(";")]
RUNS, REPS = int(os.environ.get("PROBE_RUNS", 3)), int(os.environ.get("PROBE_REPS", 16))
dev = torch.device("cuda:0")
gc = torch.Generator(device=dev)
gc.manual_seed(4242)
centers = torch.randn(max(64, N // 1000), D, generator=gc, device=dev)
centers /= centers.norm(dim=1, keepdim=True)
X = bench.mixture(torch, N, D, 42, dev, centers)
Q = bench.mixture(torch, 128 * Bc, D, 43, dev, centers)
vr = ca.sample_values_range(X[:1000].cpu().numpy(), 1.0)
hp = ca.HNSWHyperParams(num_layers=9, ef_construction=128, ef_search=64, level_0_neighbors_count=64, neighbors_count=32)
ix = ca.HNSWIndex(D, hp, ca.DistanceMetric.Cosine, ca.StorageType.UnsignedByte(), vr, shortlist_size=64, device=0, seed=42, visited_mode=ca.VISITED_REF)
ix.upload_vectors_device(X.data_ptr(), N, keepalive=X)
ix.build(4096)
qs = np.ascontiguousarray(Q.cpu().numpy().reshape(128, Bc, D))
direct = ix.batch_search(qs[0], K)
harness = bench.native_callers_harness()
fn = C.cast(_lib.lib().cos_search_batch, C.c_void_p)
print(json.dumps({"n": N, "dim": D, "host_cores": bench.effective_cores(), "queries_per_call": Bc, "calls_per_caller": REPS, "runs": RUNS}), flush=True)
for nc, maxq, win in CASES:
    ix.set_coalescing(maxq, win)
    secs = C.c_double(0.0)
    i0 = np.zeros((Bc, K), np.uint32); s0 = np.zeros((Bc, K), np.float32); c0 = np.zeros(Bc, np.uint32)
    nfail, times = 0, []
    for _ in range(RUNS):
        nfail += harness.run_callers(fn, ix._h, qs.ctypes.data_as(C.c_void_p), 128, Bc, D, K, nc, REPS, C.byref(secs),
                                     i0.ctypes.data_as(C.c_void_p), s0.ctypes.data_as(C.c_void_p), c0.ctypes.data_as(C.c_void_p))
        times.append(secs.value)
    st = ix.coalescing_stats()
    same = all(np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32)) for a, b in zip((i0, s0, c0), direct))
    print(json.dumps({"callers": nc, "max_queries": maxq, "window_us": win, "qps_best": round(nc * REPS * Bc / min(times)), "qps_runs": [round(nc * REPS * Bc / t) for t in times],
                      "failed_calls": int(nfail), "identical_to_uncoalesced_call": bool(same), "launches": st["launches"],
                      "queries_per_launch": round(st["queries"] / max(1, st["launches"]), 1), "left_full_quiet_deadline": [st["closed_full"], st["closed_quiet"], st["closed_deadline"]],
                      "solo_calls": st["solo_calls"]}), flush=True)
ix.set_coalescing(0, 0)
