#!/usr/bin/env python3
"""Fabric-side traffic PER DISPATCH of one kernel from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of a config's own script (bench_c3.py,
bench_c5.py): bytes = FETCH_SIZE x 1024 x k + WRITE_SIZE x 1024 (k: profiles/pmc_calibration.json, like scripts/pmc_traffic.py).  The entry
goes into profiles/pmc_traffic.json under `workload` = <key> with ef_search 0 / queries_per_launch 0; the config's record multiplies it by
the dispatches of one step it counted itself (bench.committed_kernel_traffic).
usage: pmc_traffic_kernel.py <fetch.db> <write.db|-> <key> <kernel name pattern> [<grid workgroups>]"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
db_f, db_w, key, pattern = sys.argv[1], (sys.argv[2] if sys.argv[2] != "-" else None), sys.argv[3], sys.argv[4]
grid = int(sys.argv[5]) if len(sys.argv) > 5 else 0
try:
    cal = json.load(open(os.path.join(ROOT, "profiles", "pmc_calibration.json")))
    ks = [p["known_over_raw_bytes"] for p in cal["per_probe"] if p["probe"] == "row_gather" and p["buffer_bytes"] >= (512 << 20)]
    k_fetch = sum(ks) / len(ks)
except (OSError, ValueError, KeyError, ZeroDivisionError):
    k_fetch = 2.0


def avg(db, counter):
    if not db:
        return 0, 0.0
    q = "select count(*), avg(value) from counters_collection where kernel_name like ? and counter_name = ?" + (" and grid_size/workgroup_size = ?" if grid else "")
    r = sqlite3.connect(db).cursor().execute(q, ("%" + pattern + "%", counter) + ((grid,) if grid else ())).fetchone()
    return r[0] or 0, r[1] or 0.0


nf, f = avg(db_f, "FETCH_SIZE")
nw, w = avg(db_w, "WRITE_SIZE")
ent = {"workload": key, "kernel": pattern, "ef_search": 0, "queries_per_launch": 0, "visited": "ref", "grid": grid, "dispatches": nf, "fetch_factor_k": k_fetch,
       "fetch_size_kb_raw": f, "write_size_kb_raw": w, "bytes_per_dispatch": f * 1024.0 * k_fetch + w * 1024.0,
       "note": "fabric-side bytes per dispatch of this kernel (average over the pass); the config's record multiplies by its dispatches per step"}
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
try:
    allv = json.load(open(path))
except (OSError, ValueError):
    allv = []
allv = [e for e in allv if not (e["workload"] == key and e.get("kernel") == pattern)] + [ent]
json.dump(allv, open(path, "w"), indent=1)
print(json.dumps(ent))
