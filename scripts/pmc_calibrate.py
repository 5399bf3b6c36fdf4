#!/usr/bin/env python3
"""Calibration of the PMC traffic figure on access patterns whose byte count is known (cos_hbm_probe): the walk's own pattern — random
gathers of 768-byte rows, 8 rows in flight per wave — over tables from L2-resident to HBM-sized, and a streaming read.

  python scripts/pmc_calibrate.py run            -> runs the probes (one JSON line each: GB/s by HIP events, bytes per launch); run it
                                                    under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` to collect the counter
  python scripts/pmc_calibrate.py read <db> ...  -> per probe launch: known bytes / (FETCH_SIZE x 1024) = the factor k that turns the
                                                    raw counter into bytes for that pattern; writes profiles/pmc_calibration.json
"""
import ctypes as C
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ROW = 768
GATHER_BYTES_PER_LAUNCH = 32768 * 4096 * ROW          # kernels_probe.hip: 32768 waves x 4096 rows
CONFIGS = [("row_gather", 2, 4 << 20), ("row_gather", 2, 48 << 20), ("row_gather", 2, 64 << 20), ("row_gather", 2, 768 << 20),
           ("row_gather", 2, 3 << 30), ("stream_read", 0, 4 << 30)]
ITERS = 2


def run():
    import cosdata_amd as ca
    lib = ca._lib.lib()
    g = C.c_double(0.0)
    for name, kind, nbytes in CONFIGS:
        ca._lib.check(lib.cos_hbm_probe(0, kind, nbytes, ROW if kind == 2 else 0, ITERS, C.byref(g)))
        per = GATHER_BYTES_PER_LAUNCH if kind == 2 else nbytes
        print(json.dumps({"probe": name, "buffer_bytes": nbytes, "row_bytes": ROW if kind == 2 else 0, "bytes_per_launch": per,
                          "launches": ITERS + 1, "GBps_hip_events": round(g.value, 1)}), flush=True)


def read(db, out_path):
    cur = sqlite3.connect(db).cursor()
    cols = [d[0] for d in cur.execute("select * from counters_collection limit 1").description]
    ocol = next((c for c in ("dispatch_id", "start", "id") if c in cols), None)
    rows = cur.execute(f"select kernel_name, value from counters_collection where counter_name = 'FETCH_SIZE' and "
                       f"(kernel_name like '%row_gather_kernel%' or kernel_name like '%stream_read_kernel%') order by {ocol or 'rowid'}").fetchall()
    per_cfg, i = [], 0
    for name, kind, nbytes in CONFIGS:
        vals = [v for _, v in rows[i:i + ITERS + 1]]
        i += ITERS + 1
        known = GATHER_BYTES_PER_LAUNCH if kind == 2 else nbytes
        timed = vals[1:] or vals                       # the first launch of a probe warms the TLB / the caches
        raw = sum(timed) / max(1, len(timed))
        per_cfg.append({"probe": name, "buffer_bytes": nbytes, "known_bytes_per_launch": known, "fetch_size_kb_raw": raw,
                        "known_over_raw_bytes": (known / (raw * 1024.0)) if raw > 0 else None})
    doc = {"counter": "FETCH_SIZE (rocprofv3 --pmc, KB)", "row_bytes": ROW, "per_probe": per_cfg,
           "note": "known_over_raw_bytes on the HBM-sized tables is the factor k for that access pattern (a table that fits the L2 fetches "
                   "almost nothing; one that fits the 256 MB memory-side cache is still counted: the counter sits between L2 and the fabric)"}
    json.dump(doc, open(out_path, "w"), indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        read(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "pmc_calibration.json"))
