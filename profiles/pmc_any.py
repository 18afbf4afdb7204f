#!/usr/bin/env python3
"""Mean per-launch value of every counter in every `pmc_*` directory under <root>, per kernel whose name matches <regex>.
   python3 profiles/pmc_any.py <root> [regex]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else "k_spmm")
for sub in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    for f in glob.glob(os.path.join(sub, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            if pat.search(r["Kernel_Name"]):
                agg[re.sub(r"\(.*", "", r["Kernel_Name"])[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in sorted(agg.items()):
            print(os.path.basename(sub), k, {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())})
