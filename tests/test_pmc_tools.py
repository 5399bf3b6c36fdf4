"""scripts/pmc_traffic.py, pmc_issue.py and pmc_calibrate.py on synthetic rocpd-like databases: the per-dispatch split of a walk's
traffic (even big dispatches = upper range, odd = lower), the calibrated FETCH_SIZE factor, the level-table GEMM picked by its grid,
instructions per step / per evaluation.  The real inputs are produced on the GPU box by scripts/final_profile.sh."""
import json
import os
import shutil
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WALK = "void cosdev::(anonymous namespace)::walk_kernel<0, 1, 1, true, false, 8>(cosdev::IndexDev, cosdev::WalkArgs)"
GEMM = "void cosdev::(anonymous namespace)::flat_codes_gemm_i8<0, false, 2>(unsigned char const*)"
DEAL = "cosdev::deal_to_xcds_kernel(unsigned int const*, unsigned int, unsigned int, unsigned int*)"
GATHER = "cosdev::(anonymous namespace)::row_gather_kernel(unsigned char const*, unsigned int)"
STREAM = "cosdev::(anonymous namespace)::stream_read_kernel(uint4 const*)"


def _db(path, counter_rows, kernels=()):
    con = sqlite3.connect(path)
    con.execute("create table counters_collection (dispatch_id int, kernel_name text, grid_size int, workgroup_size int, counter_name text, value real)")
    con.execute("create table kernels (name text, grid_x int, workgroup_x int, start int, end int, duration int)")
    con.executemany("insert into counters_collection values (?,?,?,?,?,?)", counter_rows)
    con.executemany("insert into kernels values (?,?,?,?,?,?)", kernels)
    con.commit()
    con.close()


def _sandbox(tmp_path):
    """a copy of the scripts with their own profiles/ directory (the tools write profiles/*.json next to themselves)"""
    root = tmp_path / "repo"
    (root / "scripts").mkdir(parents=True)
    (root / "profiles").mkdir()
    for f in ("pmc_traffic.py", "pmc_issue.py", "pmc_calibrate.py"):
        shutil.copy(os.path.join(ROOT, "scripts", f), root / "scripts" / f)
    return root


def test_traffic_is_split_by_dispatch_and_uses_the_calibrated_factor(tmp_path):
    root = _sandbox(tmp_path)
    json.dump({"per_probe": [{"probe": "row_gather", "buffer_bytes": 4 << 20, "known_over_raw_bytes": 6000.0},
                             {"probe": "row_gather", "buffer_bytes": 768 << 20, "known_over_raw_bytes": 2.02},
                             {"probe": "row_gather", "buffer_bytes": 3 << 30, "known_over_raw_bytes": 1.98}]}, open(root / "profiles" / "pmc_calibration.json", "w"))
    rows, kern, d = [], [], 0
    for step in range(3):                                       # per step: GEMM (big grid), a small GEMM, upper walk, deal, lower walk
        for name, grid, wg, val in ((GEMM, 5376 * 512, 512, 1000.0), (GEMM, 42 * 512, 512, 10.0), (WALK, 32768 * 64, 64, 12000.0 + step),
                                    (DEAL, 128 * 256, 256, 1.0), (WALK, 32768 * 64, 64, 8000.0 + step)):
            rows.append((d, name, grid, wg, "FETCH_SIZE", val))
            kern.append((name, grid, wg, d * 10, d * 10 + 5, 5))
            d += 1
    fdb, wdb = tmp_path / "f.db", tmp_path / "w.db"
    _db(fdb, rows, kern)
    _db(wdb, [(i, n, g, w, "WRITE_SIZE", 100.0 if "walk" in n else 700.0) for i, n, g, w, _, _ in rows], kern)
    out = subprocess.run([sys.executable, str(root / "scripts" / "pmc_traffic.py"), str(fdb), "32768", "c2", "64", "walk_kernel<0, 1, 1, true, false, 8>", str(wdb),
                          "ref", "5376"], capture_output=True, text=True, check=True).stdout
    ent = json.loads(out)
    k = 2.0                                                      # mean of the two HBM-sized probes; the L2-resident one is ignored
    assert abs(ent["fetch_factor_k"] - k) < 1e-9 and ent["split_walk"] is True
    assert ent["parts"]["upper"]["dispatches"] == 3 and ent["parts"]["lower"]["dispatches"] == 3
    assert abs(ent["parts"]["upper"]["bytes"] - (12001.0 * 1024 * k + 100.0 * 1024)) < 1.0
    assert abs(ent["parts"]["lower"]["bytes"] - (8001.0 * 1024 * k + 100.0 * 1024)) < 1.0
    assert ent["parts"]["table_gemm"]["dispatches"] == 3 and abs(ent["parts"]["table_gemm"]["bytes"] - (1000.0 * 1024 * k + 700.0 * 1024)) < 1.0
    assert abs(ent["hbm_bytes_per_launch"] - (ent["parts"]["upper"]["bytes"] + ent["parts"]["lower"]["bytes"])) < 1.0
    saved = json.load(open(root / "profiles" / "pmc_traffic.json"))
    assert len(saved) == 1 and saved[0]["workload"] == "c2" and saved[0]["visited"] == "ref"


def test_issue_counts_per_step_and_per_evaluation(tmp_path):
    root = _sandbox(tmp_path)
    rows, kern, d = [], [], 0
    for step in range(2):
        for name, grid, wg, valu, salu, dur in ((WALK, 32768 * 64, 64, 1.0e9, 1.2e9, 4_000_000), (DEAL, 128 * 256, 256, 1.0, 1.0, 5_000), (WALK, 32768 * 64, 64, 0.9e9, 1.3e9, 2_500_000)):
            rows += [(d, name, grid, wg, "SQ_INSTS_VALU", valu), (d, name, grid, wg, "SQ_INSTS_SALU", salu)]
            kern.append((name, grid, wg, d * 10, d * 10 + dur, dur))
            d += 1
    db = tmp_path / "s.db"
    _db(db, rows, kern)
    out = subprocess.run([sys.executable, str(root / "scripts" / "pmc_issue.py"), str(db), "32768", "c2", "64", "walk_kernel<0, 1, 1, true, false, 8>", "125000000"],
                         capture_output=True, text=True, check=True).stdout
    ent = json.loads(out)
    assert ent["dispatches_per_step"] == 2 and abs(ent["valu_per_step"] - 1.9e9) < 1 and abs(ent["salu_per_step"] - 2.5e9) < 1
    assert abs(ent["valu_per_eval"] - 15.2) < 1e-6 and abs(ent["salu_per_eval"] - 20.0) < 1e-6
    kern_s = 6.5e-3                                              # 4.0 + 2.5 ms per step
    assert abs(ent["valu_busy"] - 1.9e9 * 4 / (1024 * 2.4e9 * kern_s)) < 1e-9


def test_calibration_reads_known_bytes_over_the_counter(tmp_path):
    root = _sandbox(tmp_path)
    rows, d = [], 0
    known = 32768 * 4096 * 768
    for size_i, raw in enumerate((16000.0, 46.0e6, 47.0e6, 50.0e6, 50.2e6)):          # five row-gather tables, three launches each
        for launch in range(3):
            rows.append((d, GATHER, 8192 * 256, 256, "FETCH_SIZE", raw * (1.5 if launch == 0 else 1.0)))   # the warm-up launch is ignored
            d += 1
    for launch in range(3):
        rows.append((d, STREAM, 8192 * 256, 256, "FETCH_SIZE", 2097166.0))
        d += 1
    db = tmp_path / "c.db"
    _db(db, rows)
    outp = tmp_path / "cal.json"
    subprocess.run([sys.executable, str(root / "scripts" / "pmc_calibrate.py"), "read", str(db), str(outp)], capture_output=True, text=True, check=True)
    cal = json.load(open(outp))
    ks = [p["known_over_raw_bytes"] for p in cal["per_probe"]]
    assert len(ks) == 6 and ks[0] > 1000 and abs(ks[3] - known / (50.0e6 * 1024)) < 1e-9 and abs(ks[5] - (4 << 30) / (2097166.0 * 1024)) < 1e-9
