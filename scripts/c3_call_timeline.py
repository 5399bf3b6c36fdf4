#!/usr/bin/env python3
"""Dispatches of the last exhaustive-scan calls in a rocprofv3 kernel trace of scripts/bench_c3.py (start / end relative to the first
dispatch of the window): from the quantize of a call's queries to the finalize of its results."""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
lane = next((c for c in ("stream_id", "stream", "queue_id", "queue") if c in cols), "0")
rows = cur.execute(f"select name, grid_x/workgroup_x, start, end, {lane} from kernels order by start").fetchall()
scans = [i for i, r in enumerate(rows) if "flat_scan_q2_fp4" in r[0]]
if not scans:
    sys.exit("no flat_scan_q2_fp4 dispatch in the trace")
i1 = min(len(rows) - 1, scans[-1] + 12)
i0 = max(0, scans[-1] - int(sys.argv[2]) if len(sys.argv) > 2 else scans[-1] - 40)
t0 = rows[i0][2]
for name, g, s, e, ln in rows[i0:i1 + 1]:
    print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  q{ln}  grid {g:6d}  {name[:70]}")
