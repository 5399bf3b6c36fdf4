#!/usr/bin/env python3
"""Turn a rocprofv3 --pmc FETCH_SIZE (and optionally WRITE_SIZE) pass of `python bench.py` into
profiles/pmc_traffic.json: HBM bytes per walk launch, with the gfx950 correction MI355X_MICROARCH.md prescribes
(FETCH_SIZE reports 1/2 of a wide 16 B/lane coalesced read; KB units)."""
import json, os, sqlite3, sys

db_fetch, grid, workload, ef, pattern = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
db_write = sys.argv[6] if len(sys.argv) > 6 and sys.argv[6] != "-" else None
per_launch = int(sys.argv[7]) if len(sys.argv) > 7 else 1  # dispatches of the kernel per step (a locality-ordered walk: 1 + number of cuts)
def avg(db, counter):
    cur = sqlite3.connect(db).cursor()
    r = cur.execute("select count(*), avg(value) from counters_collection where kernel_name like ? and "
                    "grid_size/workgroup_size = ? and counter_name = ?", ("%" + pattern + "%", grid, counter)).fetchone()
    return r
n, fetch_kb = avg(db_fetch, "FETCH_SIZE")
wn, write_kb = avg(db_write, "WRITE_SIZE") if db_write else (0, 0.0)
ent = {"kernel": pattern, "workload": workload, "ef_search": ef, "queries_per_launch": grid, "dispatches": n,
       "fetch_size_kb_raw": fetch_kb, "write_size_kb_raw": write_kb or 0.0,
       "dispatches_per_launch": per_launch,
       "hbm_bytes_per_launch": (fetch_kb * 2.0 + (write_kb or 0.0)) * 1024.0 * per_launch,
       "note": "(FETCH_SIZE x2 (gfx950 wide-load correction, MI355X_MICROARCH.md §HBM) + WRITE_SIZE, KB -> bytes), averaged over the kernel's "
               "dispatches, x dispatches_per_launch (the walk of one step)"}
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
try:
    allv = json.load(open(path))
except (OSError, ValueError):
    allv = []
allv = [e for e in allv if (e["workload"], e["ef_search"], e["queries_per_launch"]) != (workload, ef, grid)] + [ent]
json.dump(allv, open(path, "w"), indent=1)
print(json.dumps(ent))
