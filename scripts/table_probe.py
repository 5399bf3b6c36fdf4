#!/usr/bin/env python3
"""Level table on / off (cos_index_set_walk_table) on ONE index: ms per 32768-query launch with two launches in flight (as bench.py
runs them) and the split of a launch by dispatch (table GEMM | levels above the cut | levels below it).  PROBE_N / PROBE_D /
PROBE_EFS / PROBE_VISITED pick the corpus (mixture, as bench.py draws it), the beam widths and the visited filter;
PROBE_COLS = comma-separated max_cols values (0 = no table).  One JSON line per (ef, cols); ids must not change."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import cosdata_amd as ca  # noqa: E402

N = int(os.environ.get("PROBE_N", 1_000_000))
D = int(os.environ.get("PROBE_D", 768))
EFS = [int(x) for x in os.environ.get("PROBE_EFS", "64").split(",")]
COLS = [int(x) for x in os.environ.get("PROBE_COLS", "0,8192,24576").split(",")]
EXACT = os.environ.get("PROBE_VISITED", "ref") == "exact"
EFC = int(os.environ.get("PROBE_EFC", 128))
B, K = int(os.environ.get("PROBE_B", 32768)), 10
dev = torch.device("cuda:0")
gc = torch.Generator(device=dev)
gc.manual_seed(4242)
centers = torch.randn(max(64, N // 1000), D, generator=gc, device=dev)
centers /= centers.norm(dim=1, keepdim=True)
X = bench.mixture(torch, N, D, 42, dev, centers)
Q = bench.mixture(torch, 2 * B, D, 43, dev, centers)
vr = ca.sample_values_range(X[:1000].cpu().numpy(), 1.0)
mode = ca.VISITED_EXACT if EXACT else ca.VISITED_REF
hp = ca.HNSWHyperParams(num_layers=9, ef_construction=EFC, ef_search=EFS[0], level_0_neighbors_count=int(os.environ.get("PROBE_M0", 64)),
                        neighbors_count=int(os.environ.get("PROBE_M", 32)))
ix = ca.HNSWIndex(D, hp, ca.DistanceMetric.Cosine, ca.StorageType.UnsignedByte(), vr, shortlist_size=64, device=0, seed=42, visited_mode=mode)
ix.upload_vectors_device(X.data_ptr(), N, keepalive=X)
t0 = time.time()
ix.build(4096)
build_s = time.time() - t0
out = [(torch.zeros(B, K, dtype=torch.int32, device=dev), torch.zeros(B, K, dtype=torch.float32, device=dev),
        torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)) for _ in range(3)]
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
ix.enable_timing(True)


def launch(i, s):
    q = Q[(i % 2) * B:(i % 2 + 1) * B]
    o = out[s]
    ix.batch_search_device(q.data_ptr(), B, K, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), streams[s].cuda_stream)


def run(in_flight, reps=int(os.environ.get("PROBE_REPS", 12))):
    for i in range(6):
        launch(i, i % in_flight)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(reps):
        launch(i, i % in_flight)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


print(json.dumps({"n": N, "dim": D, "visited": "exact" if EXACT else "ref", "build_s": round(build_s, 2), "cut_after_levels": ix.walk_order_cuts(),
                  "level_counts": [ix.level_count(l) for l in range(10)]}), flush=True)
# PROBE_VARIANTS="base;walk_chain_phases=1;walk_chain_phases=1,walk_pb_upper=4": tuning-knob settings (cosdata_amd/csrc/tuning.h) to time
# one after the other on the same index; every variant starts from the built-in defaults
from cosdata_amd import _lib
VARIANTS = [v for v in os.environ.get("PROBE_VARIANTS", "base").split(";") if v]
ref = None
for variant, ef in [(v, e) for e in EFS for v in VARIANTS]:
    _lib.tuning_clear(None)
    for kv in ([] if variant == "base" else variant.split(",")):
        k, v = kv.split("=")
        _lib.tuning_set(k.strip(), int(v))
    ix.set_ef_search(ef)
    if variant == VARIANTS[0]:
        ref = None
    for cols in COLS:
        ix.set_walk_table(cols, 4096 if cols else 0)
        info = ix.walk_table_info()
        ms1 = run(1)
        sp = ix.last_walk_split(streams[0].cuda_stream)
        ms2 = run(2)
        ms3 = run(3) if os.environ.get("PROBE_3", "0") == "1" else 0.0
        ids = torch.cat([out[0][0], out[1][0]]).clone()
        same = True if ref is None else bool(torch.equal(ids, ref))
        ref = ids if ref is None else ref
        print(json.dumps({"variant": variant, "ef": ef, "max_cols": cols, "table_level_min": info[0], "table_cols": info[1],
                          "ms_per_launch_1_in_flight": round(ms1, 4), "ms_per_launch_2_in_flight": round(ms2, 4), "qps_2_in_flight": round(B / ms2 * 1e3), "ms_per_launch_3_in_flight": round(ms3, 4),
                          "alone": {"table_ms": round(sp.table_ms, 4), "upper_ms": round(sp.upper_ms, 4), "sort_ms": round(sp.sort_ms, 4),
                                    "lower_ms": round(sp.lower_ms, 4), "table_evals": sp.table_evals, "upper_evals": sp.upper_evals,
                                    "lower_evals": sp.lower_evals, "upper_exp": sp.upper_expansions, "lower_exp": sp.lower_expansions,
                                    "table_tops": round(sp.table_int8_ops / max(sp.table_ms, 1e-6) / 1e9, 1)},
                          "ids_identical_to_first_config": same}), flush=True)
