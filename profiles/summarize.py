#!/usr/bin/env python3
"""Condense rocprofv3 csv output (kernel trace stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name[:70]


stats = glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True)
for f in stats:
    print("== kernel stats:", os.path.relpath(f, root))
    rows = list(csv.DictReader(open(f)))
    print(f"{'kernel':72s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
    for r in rows[:16]:
        print(f"{short(r['Name']):72s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e3:12.1f} "
              f"{float(r['AverageNs'])/1e3:10.2f} {float(r['MinNs'])/1e3:9.2f} {float(r['MaxNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}")

for sub, title in (("pmc_fetch", "FETCH_SIZE (KiB units; x2 for wide coalesced reads on gfx950)"),
                   ("pmc_write", "WRITE_SIZE (KiB)"), ("pmc_l2", "TCC_HIT_sum / TCC_MISS_sum")):
    files = glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        print("==", title, os.path.relpath(f, root))
        agg = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in sorted(agg.items()):
            parts = [f"{c}: mean {sum(v)/len(v):.1f} (n={len(v)})" for c, v in sorted(cs.items())]
            print(f"  {k:72s} " + " | ".join(parts))
