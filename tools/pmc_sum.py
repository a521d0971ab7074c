#!/usr/bin/env python3
"""Sum one rocprofv3 --pmc counter per kernel name: python tools/pmc_sum.py <dir> <COUNTER> [name filter]  (development tool).
FETCH_SIZE / WRITE_SIZE are KiB per dispatch; FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md)."""
import csv
import glob
import sys
from collections import defaultdict

d, counter = sys.argv[1], sys.argv[2]
flt = sys.argv[3] if len(sys.argv) > 3 else ""
acc = defaultdict(lambda: defaultdict(float))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and flt in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:100]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items()):
    n = len(v)
    tot = sum(v.values())
    scale = 2 * 1024 if counter == "FETCH_SIZE" else 1024 if counter == "WRITE_SIZE" else 1
    print(f"{k}: {n} dispatches, {counter} per dispatch = {tot / n * scale / 1e9 if scale > 1 else tot / n:.4f}{' GB' if scale > 1 else ''}")
