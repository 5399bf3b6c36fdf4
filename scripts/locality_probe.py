#!/usr/bin/env python3
"""Does the ORDER in which a launch's queries meet the chip matter?  c2 corpus (1M x 768 mixture), one 32768-query launch at ef 64,
timed with the queries (a) in their natural (random) order, (b) sorted by nearest mixture centre, (c) sorted and dealt to the XCDs in
contiguous runs (workgroup b runs on XCD b % 8: sorted position (b % 8) * B/8 + b / 8).  Results are per-query independent, so only
the time may change.  One JSON line per order."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import cosdata_amd as ca  # noqa: E402

N = int(os.environ.get("PROBE_N", 1_000_000))
D, B, K, EF = 768, 32768, 10, int(os.environ.get("PROBE_EF", 64))
dev = torch.device("cuda:0")
gc = torch.Generator(device=dev)
gc.manual_seed(4242)
centers = torch.randn(max(64, N // 1000), D, generator=gc, device=dev)
centers /= centers.norm(dim=1, keepdim=True)
X = bench.mixture(torch, N, D, 42, dev, centers)
Q = bench.mixture(torch, B, D, 43, dev, centers)
vr = ca.sample_values_range(X[:1000].cpu().numpy(), 1.0)
hp = ca.HNSWHyperParams(num_layers=9, ef_construction=128, ef_search=EF, level_0_neighbors_count=64, neighbors_count=32)
def make_index():
    ix = ca.HNSWIndex(D, hp, ca.DistanceMetric.Cosine, ca.StorageType.UnsignedByte(), vr, shortlist_size=64, device=0, seed=42)
    ix.upload_vectors_device(X.data_ptr(), N, keepalive=X)
    t0 = time.time()
    ix.build(4096)
    ix.set_ef_search(EF)
    return ix, time.time() - t0


# the library's own locality order (COS_WALK_ORDER_MIN_B, read when an index is created): off for the first index, default for the second
os.environ["COS_WALK_ORDER_MIN_B"] = "0"
ix, build_s = make_index()

o_ids = torch.zeros(B, K, dtype=torch.int32, device=dev)
o_sc = torch.zeros(B, K, dtype=torch.float32, device=dev)
o_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
o_st = torch.zeros(B, dtype=torch.int32, device=dev)
st = torch.cuda.Stream(device=dev)


def run(q, reps=6):
    for _ in range(2):
        ix.batch_search_device(q.data_ptr(), B, K, o_ids.data_ptr(), o_sc.data_ptr(), o_cnt.data_ptr(), o_st.data_ptr(), st.cuda_stream)
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        ix.batch_search_device(q.data_ptr(), B, K, o_ids.data_ptr(), o_sc.data_ptr(), o_cnt.data_ptr(), o_st.data_ptr(), st.cuda_stream)
    e1.record(st)
    st.synchronize()
    return e0.elapsed_time(e1) / reps, o_ids.clone()


key = (Q @ centers.T).argmax(dim=1)
order = torch.argsort(key, stable=True)
b = torch.arange(B, device=dev)
swz = (b % 8) * (B // 8) + b // 8                      # workgroup b -> sorted position
orders = {
    "natural": b,
    "sorted_by_centre": order,
    "sorted_dealt_to_xcds": order[swz],
    "natural_dealt_to_xcds": swz,
}
base_ids = None
for name, o in orders.items():
    q = Q[o].contiguous()
    ms, ids = run(q)
    inv = torch.empty_like(o)
    inv[o] = b
    ids_nat = ids[inv]
    if base_ids is None:
        base_ids = ids_nat
    print(json.dumps({"order": name, "library_split": "off", "ms_per_launch": round(ms, 4), "qps": round(B / ms * 1e3), "ef": EF, "n": N,
                      "build_s": round(build_s, 2), "results_identical_to_natural_order": bool(torch.equal(ids_nat, base_ids))}), flush=True)

# the library's own order: queries in arrival (random) order, the walk cut after the given key levels
g_un = ix.download_graph()
del os.environ["COS_WALK_ORDER_MIN_B"]
for split in os.environ.get("PROBE_SPLITS", "default;1;4;4,1;3;3,1;5,1;5,3,1;2;4,2,1").split(";"):
    if split == "default":
        os.environ.pop("COS_WALK_SPLIT", None)
    else:
        os.environ["COS_WALK_SPLIT"] = split
    ix, _ = make_index()
    g_or = ix.download_graph()
    assert all((a[0] == c[0]).all() and (a[1] == c[1]).all() for a, c in zip(g_un, g_or)), "the two builds differ"
    ms, ids = run(Q)
    print(json.dumps({"order": "natural", "library_split": split, "ms_per_launch": round(ms, 4), "qps": round(B / ms * 1e3), "ef": EF,
                      "results_identical_to_natural_order": bool(torch.equal(ids, base_ids))}), flush=True)
