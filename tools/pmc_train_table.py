#!/usr/bin/env python3
"""Join separate rocprofv3 --pmc passes of the same command into one per-kernel table (markdown on stdout):

  python tools/pmc_train_table.py <dir with p1 (FETCH_SIZE), p2 (WRITE_SIZE), p3 (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE)>

FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md: the counter tallies 128-B requests at 64 B); FETCH/WRITE_SIZE
are in KiB; MFMA busy = SUM(MFMA_BUSY_CYCLES) / (SUM(GUI_ACTIVE) * 128); durations from the kernel trace of pass 3."""
import collections
import csv
import glob
import os
import sys


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void at::native::", "at::native::")
    return n.split("(")[0][:64]


def counters(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in n.items()}


def durations(d):
    tot = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            tot[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            cnt[k] += 1
    return tot, cnt


def main(root):
    rd, n1 = counters(os.path.join(root, "p1"))
    wr, _ = counters(os.path.join(root, "p2"))
    mf, _ = counters(os.path.join(root, "p3"))
    tot, cnt = durations(os.path.join(root, "p3"))
    print("| kernel | launches | avg us | read MB/launch | write MB/launch | HBM TB/s | MFMA busy |")
    print("|---|---|---|---|---|---|---|")
    for k in sorted(tot, key=lambda k: -tot[k])[:40]:
        n = cnt[k]
        us = tot[k] / n
        r = 2.0 * rd[k].get("FETCH_SIZE", 0.0) * 1024 / max(n1.get(k, n), 1) / 1e6
        w = wr[k].get("WRITE_SIZE", 0.0) * 1024 / max(n1.get(k, n), 1) / 1e6
        gui = mf[k].get("GRBM_GUI_ACTIVE", 0.0)
        busy = mf[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 128) if gui else 0.0
        print(f"| {k} | {n} | {us:.0f} | {r:.0f} | {w:.0f} | {(r + w) / us:.2f} | {busy:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1])
