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
