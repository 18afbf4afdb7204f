#!/usr/bin/env python3
"""Condense the PMC passes of `bench.py --spmm_only` (separate rocprofv3 --pmc runs, as the MI355X guide
prescribes: FETCH_SIZE / WRITE_SIZE / TCC_HIT+MISS each in its own pass) into per-launch memory-side
bytes of the dominant kernel:  traffic = 2 * FETCH_SIZE (gfx950 half-count of wide reads) + WRITE_SIZE,
KiB units, cross-checked with TCC_MISS * 128 B.

    python profiles/pmc_traffic.py <dir with pmc_fetch_<dt>/ pmc_write_<dt>/ pmc_l2_<dt>/> [--write workload [--out file.json]]

--write stores the result in profiles/hbm_traffic.json keyed by workload:dtype:k_spmm together with the
source hash of the library it was measured on; bench.py reports it only while that hash matches."""
import csv
import glob
import importlib
import json
import os
import sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def counters(d):
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    root = sys.argv[1]
    write = sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--write" else None
    out_path = sys.argv[5] if len(sys.argv) > 5 and sys.argv[4] == "--out" else os.path.join(REPO, "profiles", "hbm_traffic.json")
    out = {}
    for dt in ("fp32", "bf16", "fp8"):
        vals = {}
        for sub in ("pmc_fetch", "pmc_write", "pmc_l2"):
            for k, cs in counters(os.path.join(root, f"{sub}_{dt}")).items():
                if "k_spmm" not in k:
                    continue
                for c, v in cs.items():
                    v = sorted(v)[len(v) // 5:]          # drop the cold first launches
                    vals[c] = sum(v) / len(v)
        if "FETCH_SIZE" not in vals:
            continue
        traffic = (2 * vals["FETCH_SIZE"] + vals.get("WRITE_SIZE", 0.0)) * 1024
        hit, miss = vals.get("TCC_HIT_sum", 0.0), vals.get("TCC_MISS_sum", 0.0)
        print(f"{dt}: FETCH_SIZE {vals['FETCH_SIZE']:.0f} KiB (x2), WRITE_SIZE {vals.get('WRITE_SIZE', 0):.0f} KiB -> "
              f"{traffic / 1e6:.1f} MB per launch; TCC hit {hit:.0f} miss {miss:.0f} (hit rate "
              f"{hit / max(1.0, hit + miss):.3f}; miss x 128 B = {miss * 128 / 1e6:.1f} MB)")
        out[dt] = traffic
    if write:
        pkg = importlib.import_module("graph-and-sequential-recommendation-systems_amd")
        path = out_path
        try:
            db = json.load(open(path))
        except Exception:
            db = {}
        for dt, t in out.items():
            db[f"{write}:{dt}:k_spmm"] = {"bytes": t, "lib_hash": pkg.build.kernel_hash(),
                                          "source": f"rocprofv3 --pmc passes under {os.path.relpath(root, REPO)} (2*FETCH_SIZE + WRITE_SIZE)"}
        json.dump(db, open(path, "w"), indent=1)
        print("wrote", path)


if __name__ == "__main__":
    main()
