#!/usr/bin/env python3
"""Join the passes of tools/pmc_issue_table.sh into one markdown table per kernel."""
import collections
import csv
import glob
import os
import sys


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:70]


def counters(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    return agg


def durations(d):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            tot[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            cnt[k] += 1
    return tot, cnt


def main(root, cmd="tools/profile_fwd.py bf16 3"):
    c1, c2, c3 = (counters(os.path.join(root, p)) for p in ("p1", "p2", "p3"))
    tot, cnt = durations(os.path.join(root, "p3"))
    print(f"Counters of `{cmd}` (three separate `rocprofv3 --pmc` passes, tools/pmc_issue_table.sh).  Wave-cycle split: "
          "issuing = SQ_ACTIVE_INST_ANY, issue-stalled = SQ_WAIT_INST_ANY, parked (s_waitcnt / barrier) = SQ_WAIT_ANY, each over SQ_WAVE_CYCLES; "
          "LDS busy = SQ_LDS_IDX_ACTIVE (LDS-array cycles, summed over the CUs) / (GRBM_GUI_ACTIVE x 32: CU-cycles), conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / "
          "(GRBM_GUI_ACTIVE x 128); effective clock = GRBM_GUI_ACTIVE / 8 / duration.\n")
    print("| kernel | launches | avg us | eff. clock GHz | MFMA busy | issuing | issue-stalled | parked | LDS busy (per CU) | LDS bank-conflict share | VALU instr per wave-cycle |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for k in sorted(tot, key=lambda k: -tot[k])[:24 if "train" in cmd else 14]:
        wc = c1[k].get("SQ_WAVE_CYCLES", 0.0) or 1.0
        gui = c3[k].get("GRBM_GUI_ACTIVE", 0.0)
        lds = c2[k].get("SQ_LDS_IDX_ACTIVE", 0.0)
        print(f"| {k} | {cnt[k]} | {tot[k] / cnt[k]:.0f} | {gui / 8.0 / (tot[k] * 1e3) if tot[k] else 0:.2f} | "
              f"{c3[k].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (gui * 128) if gui else 0:.2f} | {c1[k].get('SQ_ACTIVE_INST_ANY', 0.0) / wc:.2f} | "
              f"{c1[k].get('SQ_WAIT_INST_ANY', 0.0) / wc:.2f} | {c1[k].get('SQ_WAIT_ANY', 0.0) / wc:.2f} | "
              f"{lds / (gui * 32) if gui else 0:.2f} | {c2[k].get('SQ_LDS_BANK_CONFLICT', 0.0) / lds if lds else 0:.3f} | {c2[k].get('SQ_INSTS_VALU', 0.0) / (wc * 4.0) if wc else 0:.3f} |")


if __name__ == "__main__":
    main(*sys.argv[1:3])
