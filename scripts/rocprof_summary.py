#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd .db (kernel trace, optional PMC) for the kernels of this repo."""
import sqlite3
import sys


def short(name):
    for key in ("walk_lat4_kernel", "walk_lat_kernel", "walk_meta_kernel", "walk_spec_kernel", "walk_kernel", "finalize_kernel", "quantize_rows_kernel", "scatter_rows_kernel",
                "merge_topk_kernel", "deal_to_xcds_kernel", "level_table_areg", "flat_", "bm25_", "rrf_kernel", "sparse_tile_kernel", "sparse_finish_kernel", "link_kernel", "claim_kernel", "evict_kernel"):
        if key in name:
            i = name.index(key)
            if key == "walk_spec_kernel" and name[max(0, i - 7):i].startswith("spec") and name[i - 2:i] == "::":
                i -= 7  # keep the variant's namespace (spec2:: .. spec8::)
            j = name.find("(", i)
            return name[i:j if j > 0 else None]
    return name[:60]


def compact_rows(rows, max_grids=12):
    """a kernel launched with more than `max_grids` different grids (the builder's claim / link / evict kernels: one grid per insertion
    round, thousands of them) becomes ONE row with grid '*' — calls and total summed, avg = total / calls, min / max over all"""
    by_name = {}
    for r in rows:
        by_name.setdefault(r[0], []).append(r)
    out = []
    for name, rs in by_name.items():
        if len(rs) <= max_grids:
            out.extend(rs)
            continue
        calls = sum(r[2] for r in rs)
        tot = sum(r[6] for r in rs)
        out.append((name, "*", calls, tot / max(1, calls), min(r[4] for r in rs), max(r[5] for r in rs), tot))
    out.sort(key=lambda r: -r[6])
    return out


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    print(f"# {path}")
    print("## kernel trace: name | grid (workgroups) | calls | avg_us | min_us | max_us | total_ms | calls per step")
    rows = cur.execute("select name, grid_x/workgroup_x, count(*), avg(duration), min(duration), max(duration), sum(duration) "
                       "from kernels group by name, grid_x/workgroup_x order by sum(duration) desc").fetchall()
    # every kernel of this library is printed (a roofline block of bench.py must be recomputable from the file whatever the kernel's
    # share of the trace: round 4's `limit 14` dropped flat_scan_q2_areg and bm25_topk_kernel); foreign kernels (torch): the top 8
    rows = compact_rows(rows)
    # calls per step: a step of bench.py ends with ONE finalize_fast_kernel dispatch over the step's queries — the (kernel, grid) row of that
    # kernel with the largest total is the step count every other row of the same launch shape is divided by (recall passes, parity samples
    # and single-batch legs use other grids and show up as fractions)
    fin = [r for r in rows if "finalize_fast_kernel" in r[0] and isinstance(r[1], int)]
    n_steps = max(fin, key=lambda r: r[6])[2] if fin else 0
    print(f"## steps in this trace (dispatches of finalize_fast_kernel at its dominant grid): {n_steps}")
    foreign = 0
    for name, grid, calls, avg, mn, mx, tot in rows:
        if "cosdev" not in name and "anonymous namespace" not in name:
            foreign += 1
            if foreign > 8:
                continue
        g = f"{grid:8d}" if isinstance(grid, int) else f"{grid:>8s}"
        per = f"{calls / n_steps:8.2f}" if n_steps else "       -"
        print(f"{short(name):45s} | {g} | {calls:6d} | {avg/1e3:10.1f} | {mn/1e3:10.1f} | {mx/1e3:10.1f} | {tot/1e6:10.2f} | {per}")
    # A locality-ordered walk is several dispatches of one kernel per step (WalkArgs::phase): split them by what preceded them ON THE
    # SAME STREAM / QUEUE (the sort's deal_to_xcds_kernel precedes every level range but the first); with several steps in flight the
    # global start order interleaves streams, so the trace's stream or queue column is used when it has one.
    try:
        cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
    except sqlite3.Error:
        cols = []
    lane_col = next((c for c in ("stream_id", "stream", "queue_id", "queue") if c in cols), None)
    try:
        seq = cur.execute(f"select name, grid_x/workgroup_x, duration, {lane_col or '0'} from kernels order by start").fetchall()
    except sqlite3.Error:
        seq = []
    parts = {}
    prev = {}
    for name, grid, dur, lane in seq:
        if ("walk_kernel" in name or "walk_spec_kernel" in name) and grid >= 4096:
            after_deal = "deal_to_xcds" in prev.get(lane, "")
            part = "lower levels, after deal_to_xcds (locality order)" if after_deal else "upper levels (arrival order) or unsplit walk"
            parts.setdefault((short(name), grid, part), []).append(dur)
        prev[lane] = name
    if any("after deal_to_xcds" in k[2] for k in parts):
        how = f"previous dispatch with the same {lane_col}" if lane_col else "previous dispatch in global start order: approximate when steps overlap"
        print(f"## walk_kernel dispatches by position in the step ({how}): name | grid | part | calls | avg_us | total_ms")
        for (nm, grid, part), ds in sorted(parts.items()):
            print(f"{nm:45s} | {grid:8d} | {part} | {len(ds)} | {sum(ds)/len(ds)/1e3:.1f} | {sum(ds)/1e6:.2f}")
    try:
        pm = cur.execute("select kernel_name, grid_size/workgroup_size, counter_name, count(*), avg(value), sum(value) from counters_collection "
                         "where kernel_name like '%cosdev%' or kernel_name like '%anonymous namespace%' "
                         "group by kernel_name, grid_size/workgroup_size, counter_name having count(*) > 0 order by kernel_name, grid_size/workgroup_size, counter_name").fetchall()
    except sqlite3.Error:
        pm = []
    if pm:
        print("## PMC: kernel | grid | counter | dispatches | avg per dispatch | total")
        for name, grid, cn, calls, avg, tot in pm:
            print(f"{short(name):45s} | {grid:8d} | {cn} | {calls} | {avg:.1f} | {tot:.1f}")
        # counters of a split walk by level range: under PMC the dispatches of the device run one at a time and the walk chain orders
        # the big walks of different streams, so the big walk_kernel dispatches alternate upper range, lower range, upper, ...
        try:
            ccols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
        except sqlite3.Error:
            ccols = []
        ocol = next((c for c in ("dispatch_id", "start", "id") if c in ccols), None)
        if ocol and any("after deal_to_xcds" in k[2] for k in parts):
            rows = cur.execute(f"select kernel_name, grid_size/workgroup_size, counter_name, value, {ocol} from counters_collection "
                               f"where (kernel_name like '%walk_kernel%' or kernel_name like '%walk_spec_kernel%') and grid_size/workgroup_size >= 4096 order by {ocol}").fetchall()
            seen, acc = {}, {}
            for name, grid, cn, val, oid in rows:
                # (above ef 64 the two level ranges are different instantiations — four / eight row buffers —: the alternation is over the
                # walk kernel's big dispatches whatever their template arguments)
                key = (short(name).split("<")[0] + "<...> big dispatches", grid, cn)
                idx = seen.setdefault(key, {})
                if oid not in idx:
                    idx[oid] = len(idx)
                part = "upper range (even dispatches)" if idx[oid] % 2 == 0 else "lower range (odd dispatches)"
                a = acc.setdefault(key + (part,), [0, 0.0])
                a[0] += 1
                a[1] += val
            print(f"## PMC of the split walk by level range (dispatch order = {ocol}): kernel | grid | counter | part | dispatches | avg per dispatch")
            for (nm, grid, cn, part), (c, tot) in sorted(acc.items()):
                print(f"{nm:45s} | {grid:8d} | {cn} | {part} | {c} | {tot / c:.1f}")


if __name__ == "__main__":
    for p in sys.argv[1:]:
        main(p)
