#!/usr/bin/env python3
"""Aggregate a rocprofv3 `--pmc ... --output-format csv` counter_collection.csv into one row per kernel.
usage: summarize_pmc.py <counter_collection.csv> [<counter_collection.csv> ...] > summary.csv"""
import collections
import csv
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
counters = sorted({c for v in rows.values() for c in v})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "dispatches"] + [f"{c}_sum" for c in counters] + [f"{c}_per_dispatch" for c in counters])
for k in sorted(rows, key=lambda k: -max(rows[k].values())):
    n = max(calls[k].values())
    w.writerow([k[:110], n] + [f"{rows[k].get(c, 0):.6g}" for c in counters] + [f"{rows[k].get(c, 0) / max(1, calls[k].get(c, 1)):.6g}" for c in counters])
