#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python bench.py` (or of a probe script that launches the same steps) into
profiles/pmc_traffic.json: fabric-side bytes of one step's walk, SPLIT by dispatch — the level range above the locality cut, the level
range below it, and the level-table GEMM.  bytes = FETCH_SIZE x 1024 x k + WRITE_SIZE x 1024, k from profiles/pmc_calibration.json
(scripts/pmc_calibrate.py: the factor measured on this part's own row-gather pattern with a known byte count; 2.0 on gfx950, which is
also what MI355X_MICROARCH.md prescribes for wide coalesced reads).

usage: pmc_traffic.py <fetch.db> <queries_per_launch> <workload> <ef> <walk kernel pattern> [<write.db>|-] [<visited>] [<GEMM workgroups>]
Under PMC the device runs one dispatch at a time and the walk chain orders the big walks of different streams, so the big dispatches
of the walk kernel alternate upper range, lower range, upper range, ... (an unsplit walk: every dispatch is "lower")."""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
db_fetch, grid, workload, ef, pattern = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
db_write = sys.argv[6] if len(sys.argv) > 6 and sys.argv[6] != "-" else None
visited = sys.argv[7] if len(sys.argv) > 7 else "ref"
gemm_grid = int(sys.argv[8]) if len(sys.argv) > 8 else 0   # workgroups of this launch shape's level-table GEMM (column tiles x row tiles); 0 = the largest seen

try:
    cal = json.load(open(os.path.join(ROOT, "profiles", "pmc_calibration.json")))
    ks = [p["known_over_raw_bytes"] for p in cal["per_probe"] if p["probe"] == "row_gather" and p["buffer_bytes"] >= (512 << 20)]
    k_fetch = sum(ks) / len(ks)
    k_src = "profiles/pmc_calibration.json: known bytes / raw FETCH_SIZE of cos_hbm_probe's 768-byte row gather over HBM-sized tables"
except (OSError, ValueError, KeyError, ZeroDivisionError):
    k_fetch, k_src = 2.0, "MI355X_MICROARCH.md (gfx950 wide-load correction); no calibration file"


def per_part(db, counter, like, split):
    """average counter value per dispatch, by position (even / odd big dispatch of the kernel) when `split`"""
    if not db:
        return {}
    cur = sqlite3.connect(db).cursor()
    cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
    ocol = next((c for c in ("dispatch_id", "start", "id") if c in cols), "rowid")
    rows = cur.execute(f"select value from counters_collection where kernel_name like ? and grid_size/workgroup_size = ? and counter_name = ? "
                       f"order by {ocol}", ("%" + like + "%", grid, counter)).fetchall()
    vals = [r[0] for r in rows]
    if not vals:
        return {}
    if not split:
        return {"all": (len(vals), sum(vals) / len(vals))}
    up, lo = vals[0::2], vals[1::2]
    return {"upper": (len(up), sum(up) / max(1, len(up))), "lower": (len(lo), sum(lo) / max(1, len(lo)))}


def deal_count(db):
    cur = sqlite3.connect(db).cursor()
    return cur.execute("select count(*) from kernels where name like '%deal_to_xcds%'").fetchone()[0]


split = deal_count(db_fetch) > 0
f = per_part(db_fetch, "FETCH_SIZE", pattern, split)
w = per_part(db_write, "WRITE_SIZE", pattern, split)
parts = {}
for part, (nd, kb) in f.items():
    wkb = w.get(part, (0, 0.0))[1]
    parts[part] = {"dispatches": nd, "fetch_size_kb_raw": kb, "write_size_kb_raw": wkb, "bytes": kb * 1024.0 * k_fetch + wkb * 1024.0}
# the level-table GEMM of the same steps (grid = column tiles x row tiles: match by name only)
cur = sqlite3.connect(db_fetch).cursor()
GEMM_NAME = "(kernel_name like '%level_table_areg%' or kernel_name like '%flat_codes_gemm_i8%')"   # query-resident kernel (round 5) or the tile kernel
GEMM_Q = (f"select count(*), avg(value) from counters_collection where {GEMM_NAME} and counter_name = ? and " +
          (f"grid_size/workgroup_size = {gemm_grid}" if gemm_grid else
           f"grid_size = (select max(grid_size) from counters_collection where {GEMM_NAME})"))   # this launch shape's GEMM only
g = cur.execute(GEMM_Q, ("FETCH_SIZE",)).fetchone()
gw = (0, 0.0)
if db_write:
    gw = sqlite3.connect(db_write).cursor().execute(GEMM_Q, ("WRITE_SIZE",)).fetchone()
if g and g[0]:
    parts["table_gemm"] = {"dispatches": g[0], "fetch_size_kb_raw": g[1], "write_size_kb_raw": gw[1] or 0.0,
                           "bytes": g[1] * 1024.0 * k_fetch + (gw[1] or 0.0) * 1024.0}
walk_bytes = sum(v["bytes"] for p, v in parts.items() if p != "table_gemm")
ent = {"kernel": pattern, "workload": workload, "ef_search": ef, "queries_per_launch": grid, "visited": visited, "fetch_factor_k": k_fetch,
       "fetch_factor_source": k_src, "split_walk": split, "parts": parts, "hbm_bytes_per_launch": walk_bytes,
       "note": "fabric-side bytes (L2 misses: served by the memory-side cache or HBM) per step, per dispatch of the walk; hbm_bytes_per_launch = "
               "the walk's dispatches of one step together"}
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
try:
    allv = json.load(open(path))
except (OSError, ValueError):
    allv = []
allv = [e for e in allv if (e["workload"], e["ef_search"], e["queries_per_launch"], e.get("visited", "ref")) != (workload, ef, grid, visited)] + [ent]
json.dump(allv, open(path, "w"), indent=1)
print(json.dumps(ent))
