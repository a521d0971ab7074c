#!/usr/bin/env python3
"""Markdown table of the rocprofv3 passes of tools/pmc_walk_ab.sh: per kernel instantiation, counters per dispatch (mean over the
dispatches of that instantiation) and the kernel-trace duration of the same passes."""
import csv, glob, re, sys
from collections import defaultdict
root = sys.argv[1]
FILTER = sys.argv[2] if len(sys.argv) > 2 else "window_attn_qkv"      # substring of the kernel names to tabulate
cnt = defaultdict(lambda: defaultdict(list))     # kernel -> counter -> per-dispatch values
dur = defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    per = defaultdict(lambda: defaultdict(float))
    names = {}
    for r in csv.DictReader(open(f)):
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for d, cs in per.items():
        for c, v in cs.items():
            cnt[names[d]][c].append(v)
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
def short(k):
    m = re.search(r"(\w+)<([^>]*)>", k.replace("(anonymous namespace)::", ""))
    return f"{m.group(1)}<{m.group(2)}>" if m else k[:60]
keys = sorted(k for k in cnt if any(f in k for f in FILTER.split(",")))
cols = ["TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
        "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE", "FETCH_SIZE", "SQ_BUSY_CYCLES"]
cols = [c for c in cols if any(c in cnt[k] for k in keys)]
print("| kernel | us (profiled passes) | " + " | ".join(cols) + " | parked | stalled | issuing | MFMA busy |")
print("|---|---|" + "---|" * (len(cols) + 4))
mean = lambda v: sum(v) / len(v) if v else float("nan")
for k in keys:
    c = cnt[k]
    wc = mean(c.get("SQ_WAVE_CYCLES", []))
    row = [short(k), f"{mean(dur.get(k, [])):.1f}"] + [f"{mean(c[x]):.4g}" if x in c else "-" for x in cols]
    row += [f"{mean(c.get('SQ_WAIT_ANY', [])) / wc:.2f}" if wc == wc and "SQ_WAIT_ANY" in c else "-",
            f"{mean(c.get('SQ_WAIT_INST_ANY', [])) / wc:.2f}" if wc == wc and "SQ_WAIT_INST_ANY" in c else "-",
            f"{mean(c.get('SQ_ACTIVE_INST_ANY', [])) / wc:.2f}" if wc == wc and "SQ_ACTIVE_INST_ANY" in c else "-"]
    gui = mean(c.get("GRBM_GUI_ACTIVE", []))
    mb = mean(c.get("SQ_VALU_MFMA_BUSY_CYCLES", []))
    row.append(f"{mb / (gui / 8 * 1024):.2f}" if gui == gui and mb == mb else "-")      # busy cycles / (cycles x 4 SIMDs x 256 CUs)
    print("| " + " | ".join(row) + " |")
