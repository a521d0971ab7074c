#!/usr/bin/env python3
"""Aggregate rocprofv3 CSV output (kernel trace / counter collection) per kernel name.

  python tools/pmc_summary.py <dir-with-csv> [out.md]
Counter rows: sums per kernel and per-dispatch averages.  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for
gfx950 wide streaming reads (the counter tallies 128-B requests at 64 B); FETCH_SIZE/WRITE_SIZE are in KiB."""
import collections
import csv
import glob
import os
import sys


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:60]


def main(d, out=None):
    lines = []
    for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(set)
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"])
        lines.append(f"## {os.path.relpath(path, d)}\n")
        names = sorted({c for v in agg.values() for c in v})
        lines.append("| kernel | dispatches | " + " | ".join(f"{c} (sum)" for c in names) + " | " + " | ".join(f"{c} / dispatch" for c in names) + " |")
        lines.append("|---|---|" + "---|" * (2 * len(names)))
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
            n = len(cnt[k])
            lines.append(f"| {k} | {n} | " + " | ".join(f"{v.get(c, 0):.4g}" for c in names) + " | " +
                         " | ".join(f"{v.get(c, 0) / n:.4g}" for c in names) + " |")
        lines.append("")
    for path in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        lines.append(f"## {os.path.relpath(path, d)}\n")
        lines.append("| kernel | calls | total ms | avg us | % |")
        lines.append("|---|---|---|---|---|")
        for r in list(csv.DictReader(open(path)))[:25]:
            lines.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | "
                         f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |")
        lines.append("")
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
