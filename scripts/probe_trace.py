#!/usr/bin/env python3
"""Per-call kernel durations of the last 40 dispatches of a rocprofv3 kernel trace (.db): which of a launch's kernels costs what."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, grid_x/workgroup_x, start, end from kernels order by start").fetchall()
tail = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -60:]
t0 = tail[0][2]
for name, grid, s, e in tail:
    short = name.split("(")[0][-70:]
    print(f"{(s - t0) / 1e3:10.1f} us +{(e - s) / 1e3:9.1f} us  grid {grid:6d}  {short}")
