"""scripts/rocprof_summary.py on a synthetic rocpd-like database: the per-kernel table, and the split of a locality-ordered walk's
dispatches by what preceded them on their own stream (two steps in flight interleave in the global start order)."""
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WALK = "void cosdev::(anonymous namespace)::walk_kernel<0, 1, 1, true, false, 8>(cosdev::IndexDev, cosdev::WalkArgs)"
DEAL = "cosdev::deal_to_xcds_kernel(unsigned int const*, unsigned int, unsigned int*)"
QUANT = "void cosdev::(anonymous namespace)::quantize_rows_kernel<0>(float const*)"


def _db(path, with_stream):
    con = sqlite3.connect(path)
    cols = "name text, grid_x int, workgroup_x int, start int, end int, duration int" + (", stream_id int" if with_stream else "")
    con.execute(f"create table kernels ({cols})")
    t = 0
    rows = []
    # two streams, each: quantize, walk (upper, 7000), deal, walk (lower, 2500); stream 1 starts while stream 0's upper walk runs
    plan = [(0, QUANT, 8192 * 64, 100), (0, WALK, 32768 * 64, 7000), (1, QUANT, 8192 * 64, 100), (0, DEAL, 128 * 256, 5), (1, WALK, 32768 * 64, 7000),
            (0, WALK, 32768 * 64, 2500), (1, DEAL, 128 * 256, 5), (1, WALK, 32768 * 64, 2500)]
    for stream, name, threads, dur in plan:
        wg = 256 if name == DEAL else 64
        rows.append((name, threads, wg, t, t + dur * 1000, dur * 1000) + ((stream,) if with_stream else ()))
        t += 10
    con.executemany(f"insert into kernels values ({','.join('?' * len(rows[0]))})", rows)
    con.commit()
    con.close()


def _run(path):
    return subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "rocprof_summary.py"), str(path)], capture_output=True, text=True, check=True).stdout


def test_split_by_stream(tmp_path):
    p = tmp_path / "s.db"
    _db(p, with_stream=True)
    out = _run(p)
    assert "walk_kernel<0, 1, 1, true, false, 8>" in out and "same stream_id" in out
    lower = [l for l in out.splitlines() if "after deal_to_xcds" in l and "|" in l and "walk_kernel<" in l]
    upper = [l for l in out.splitlines() if "upper levels" in l and "walk_kernel<" in l]
    assert len(lower) == 1 and len(upper) == 1
    assert [c.strip() for c in lower[0].split("|")][3:5] == ["2", "2500.0"]
    assert [c.strip() for c in upper[0].split("|")][3:5] == ["2", "7000.0"]


def test_without_a_stream_column_says_approximate(tmp_path):
    p = tmp_path / "g.db"
    _db(p, with_stream=False)
    out = _run(p)
    assert "approximate when steps overlap" in out


def test_every_library_kernel_is_printed_and_per_round_builder_kernels_are_one_row(tmp_path):
    """round 5: no `limit 14` (a roofline block must be recomputable from the file whatever the kernel's share of the trace); a
    kernel launched with thousands of different grids (the builder's link kernel: one grid per insertion round) is ONE row"""
    p = tmp_path / "many.db"
    con = sqlite3.connect(p)
    con.execute("create table kernels (name text, grid_x int, workgroup_x int, start int, end int, duration int, stream_id int)")
    link = "cosdev::(anonymous namespace)::link_kernel<1>(cosdev::LinkArgs)"
    rows, t = [], 0
    for g in range(1, 41):                      # 40 different grids of one kernel
        rows.append((link, g * 64, 64, t, t + 50_000, 50_000, 0)); t += 10
    for i in range(30):                         # 30 different small kernels of the library: all must appear
        rows.append((f"cosdev::(anonymous namespace)::tiny_kernel_{i}(int)", 64, 64, t, t + 1000, 1000, 0)); t += 10
    for i in range(20):                         # 20 foreign kernels: only the top 8 are kept
        rows.append((f"void at::native::foreign_{i}(float*)", 64, 64, t, t + 2000 + i, 2000 + i, 0)); t += 10
    con.executemany("insert into kernels values (?,?,?,?,?,?,?)", rows)
    con.commit()
    con.close()
    out = _run(p)
    table = out.split("## kernel trace")[1]
    link_rows = [l for l in table.splitlines() if l.startswith("link_kernel<1>")]
    assert len(link_rows) == 1 and "|        * |     40 |" in link_rows[0]            # one row, grid '*', 40 calls
    assert sum(1 for i in range(30) if f"tiny_kernel_{i}(" in table) == 30
    assert sum(1 for i in range(20) if f"foreign_{i}(" in table) == 8
