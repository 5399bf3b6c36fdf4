#!/usr/bin/env python3
"""SQ counters of the exhaustive-scan kernels from a rocprofv3 --pmc pass of scripts/fp4_w8_probe.py: for every flat_scan kernel name, the
counters of its LARGEST dispatch (the last of the four chunk launches of a call: 8.5M of the 10M rows), as the maximum over dispatches.
usage: pmc_scan_counters.py <db> [<db> ...]   -> one JSON line per kernel"""
import json, re, sqlite3, sys


def short_name(n):
    m = re.search(r'flat_scan\w*(<[^>]*>)?', n)
    return m.group(0) if m else n

out = {}
for db in sys.argv[1:]:
    cur = sqlite3.connect(db).cursor()
    for name, counter, n, avg in cur.execute("select kernel_name, counter_name, count(*), max(value) from counters_collection where kernel_name like '%flat_scan%' "
                                             "group by kernel_name, counter_name"):
        short = short_name(name)
        out.setdefault(short, {})[counter] = avg
        out[short]["dispatches"] = n
    for name, n, avg in cur.execute("select name, count(*), max(duration) from kernels where name like '%flat_scan%' group by name"):
        short = short_name(name)
        out.setdefault(short, {})["max_duration_us_under_pmc"] = avg / 1e3
for k, v in out.items():
    print(json.dumps({"kernel": k, **v}))
