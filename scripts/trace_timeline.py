#!/usr/bin/env python3
"""Timeline of the LAST few steps of a rocprofv3 kernel trace of bench.py: every dispatch with its queue, start and end relative to the
window's first dispatch, and the gaps of the walk chain (time during which no walk kernel of the main launch size runs).
usage: trace_timeline.py <results.db> [<steps back from the end>=4] [<main grid>=32768]"""
import sqlite3
import sys

from rocprof_summary import short

db = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 4
grid = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
cur = sqlite3.connect(db).cursor()
cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
lane = next((c for c in ("stream_id", "stream", "queue_id", "queue") if c in cols), "0")
rows = cur.execute(f"select name, grid_x/workgroup_x, start, end, {lane} from kernels order by start").fetchall()
fin = [i for i, r in enumerate(rows) if "finalize_fast_kernel" in r[0] and r[1] == grid]
if len(fin) < back + 2:
    sys.exit("not enough steps in the trace")
# the timed region's steps: the longest run of finalize dispatches at the main grid; take the window of the last `back` of the first 3/4 of them
cut = fin[: max(back + 1, (len(fin) * 3) // 4)]
i0, i1 = cut[-back - 1], cut[-1]
t0 = rows[i0][2]
print(f"# {db}: dispatches between the finalize of step -{back} and the finalize of the last timed step ({lane})")
walk_busy = []
for name, g, s, e, ln in rows[i0:i1 + 1]:
    nm = short(name)
    print(f"{(s - t0) / 1e6:9.3f} .. {(e - t0) / 1e6:9.3f} ms  ({(e - s) / 1e6:7.3f})  q{ln}  grid {g:6d}  {nm[:70]}")
    if "walk_kernel" in nm and g == grid:
        walk_busy.append((s, e))
walk_busy.sort()
gaps, last = [], None
for s, e in walk_busy:
    if last is not None and s > last:
        gaps.append((s - last) / 1e6)
    last = e if last is None else max(last, e)
span = (walk_busy[-1][1] - walk_busy[0][0]) / 1e6 if walk_busy else 0.0
print(f"# walk kernel busy {sum(e - s for s, e in walk_busy) / 1e6:.3f} ms of a {span:.3f} ms span; gaps between consecutive walk dispatches (ms): "
      + ", ".join(f"{g:.3f}" for g in gaps))
