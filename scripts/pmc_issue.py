#!/usr/bin/env python3
"""Instruction issue of the walk kernel from a rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU ... pass of the same steps bench.py times:
wave-instructions per step and per distance evaluation, and the share of the SIMDs' issue cycles they take (a wave64 VALU
instruction occupies its SIMD for 4 cycles; the scalar unit issues one instruction per SIMD turn: the same 4-cycle cadence).
Writes profiles/pmc_issue.json (read by bench.py into roofline.issue).

usage: pmc_issue.py <sq.db> <queries_per_launch> <workload> <ef> <walk kernel pattern> <evals_per_step> [<visited>]"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
db, grid, workload, ef, pattern, evals = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5], float(sys.argv[6])
visited = sys.argv[7] if len(sys.argv) > 7 else "ref"
SIMDS, CLOCK_HZ = 256 * 4, 2.4e9
cur = sqlite3.connect(db).cursor()
split = cur.execute("select count(*) from kernels where name like '%deal_to_xcds%'").fetchone()[0] > 0
per_step = 2 if split else 1


def avg(counter):
    r = cur.execute("select count(*), avg(value) from counters_collection where kernel_name like ? and grid_size/workgroup_size = ? and counter_name = ?",
                    ("%" + pattern + "%", grid, counter)).fetchone()
    return (r[1] or 0.0) * per_step, r[0]


valu, nd = avg("SQ_INSTS_VALU")
salu, _ = avg("SQ_INSTS_SALU")
vmem, _ = avg("SQ_INSTS_VMEM_RD")
lds, _ = avg("SQ_INSTS_LDS")
dur = cur.execute("select avg(duration) from kernels where name like ? and grid_x/workgroup_x = ?", ("%" + pattern + "%", grid)).fetchone()[0] or 0.0
kern_s = dur * 1e-9 * per_step
ent = {"kernel": pattern, "workload": workload, "ef_search": ef, "queries_per_launch": grid, "visited": visited, "dispatches": nd,
       "dispatches_per_step": per_step, "valu_per_step": valu, "salu_per_step": salu, "vmem_rd_per_step": vmem, "lds_per_step": lds,
       "valu_per_eval": valu / evals, "salu_per_eval": salu / evals, "kernel_ms_per_step_under_pmc": kern_s * 1e3,
       "valu_busy": valu * 4.0 / (SIMDS * CLOCK_HZ * kern_s) if kern_s else None, "salu_busy": salu * 4.0 / (SIMDS * CLOCK_HZ * kern_s) if kern_s else None,
       "source": os.path.basename(db) + ": rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS, wave-instructions summed over the walk "
                 "kernel's dispatches of a step; busy = instructions x 4 cycles / (1024 SIMDs x 2.4 GHz x kernel time under the counters)"}
path = os.path.join(ROOT, "profiles", "pmc_issue.json")
try:
    allv = json.load(open(path))
except (OSError, ValueError):
    allv = []
allv = [e for e in allv if (e["workload"], e["ef_search"], e["queries_per_launch"], e.get("visited", "ref")) != (workload, ef, grid, visited)] + [ent]
json.dump(allv, open(path, "w"), indent=1)
print(json.dumps(ent))
