#!/usr/bin/env python3
"""Dispatches of the last hybrid call in a rocprofv3 kernel trace of scripts/bench_c5.py (start / end relative to the call's first dispatch)."""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
lane = next((c for c in ("stream_id", "stream", "queue_id", "queue") if c in cols), "0")
rows = cur.execute(f"select name, grid_x/workgroup_x, start, end, {lane} from kernels order by start").fetchall()
rrf = [i for i, r in enumerate(rows) if "rrf_kernel" in r[0]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
i1 = rrf[-back]
i0 = rrf[-back - 1] + 1
t0 = rows[i0][2]
for name, g, s, e, ln in rows[i0:i1 + 1]:
    print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  q{ln}  grid {g:6d}  {name[:60]}")
