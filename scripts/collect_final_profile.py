#!/usr/bin/env python3
"""After `gpurun -- bash scripts/final_profile.sh`: copy what the run left under gpurun_out/ into profiles/ under the round's names
(usage: collect_final_profile.py r06).  Kernel-trace / PMC summaries written by an older scripts/rocprof_summary.py are compacted with
the rule the current one applies (a kernel launched with more than 12 different grids becomes one row)."""
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "rNN"


def compact_text(src, dst):
    lines = open(src).read().split("\n")
    hdr = [k for k, l in enumerate(lines) if l.startswith("## ")]
    if not hdr:
        shutil.copy(src, dst)
        return
    first_end = hdr[1] if len(hdr) > 1 else len(lines)
    head, rows, rest = lines[:hdr[0] + 1], lines[hdr[0] + 1:first_end], lines[first_end:]
    by = {}
    for l in rows:
        f = [x.strip() for x in l.split("|")]
        if len(f) not in (7, 8):
            continue
        try:
            by.setdefault(f[0], []).append((f[0], f[1], int(f[2]), float(f[3]), float(f[4]), float(f[5]), float(f[6])))
        except ValueError:
            continue
    comp = []
    for name, rs in by.items():
        if len(rs) <= 12:
            comp += rs
        else:
            calls, tot = sum(r[2] for r in rs), sum(r[6] for r in rs)
            comp.append((name, "*", calls, tot * 1e3 / max(1, calls), min(r[4] for r in rs), max(r[5] for r in rs), tot))
    comp.sort(key=lambda r: -r[6])
    body = [f"{n:45s} | {g:>8s} | {c:6d} | {a:10.1f} | {mn:10.1f} | {mx:10.1f} | {t:10.2f}" for n, g, c, a, mn, mx, t in comp]
    rest = [l for l in rest if not re.match(r"^(link_kernel|claim_kernel|evict_kernel)", l)]
    open(dst, "w").write("\n".join(head + body + rest))


texts = {f"final_kernel_trace_{w}.txt": f"final_kernel_trace_{w}_rocprofv3.txt" for w in ("main", "c2", "c3", "c5")}
texts.update({f"final_pmc_{c}_{w}.txt": f"final_pmc_{c}_{w}_rocprofv3.txt" for c in ("fetch_size", "write_size", "sq_instruction_mix") for w in ("main", "c2", "c3", "c5")})
copies = {"final_bench_all_configs.json": "final_bench_default_stdout_line.json", "final_bench_all_configs_full_record.json": "final_bench_default_full_record.json",
          "final_bench_main_under_rocprofv3.json": "final_bench_main_under_rocprofv3.json", "final_bench_c2_under_rocprofv3.json": "final_bench_c2_under_rocprofv3.json",
          "final_c3.json": "final_c3.json", "final_c5.json": "final_c5.json", "final_pytest.log": "final_pytest.log", "final_profile.log": "final_profile_script.log",
          "final_step_timeline_main.txt": "final_step_timeline_main.txt"}
for a, b in texts.items():
    if os.path.exists(os.path.join(G, a)):
        compact_text(os.path.join(G, a), os.path.join(P, f"{tag}_{b}"))
for a, b in copies.items():
    if os.path.exists(os.path.join(G, a)):
        shutil.copy(os.path.join(G, a), os.path.join(P, f"{tag}_{b}"))
for a in ("pmc_traffic.json", "pmc_issue.json"):
    if os.path.exists(os.path.join(G, a)):
        shutil.copy(os.path.join(G, a), os.path.join(P, a))
print("collected into", P)
