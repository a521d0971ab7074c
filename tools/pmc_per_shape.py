#!/usr/bin/env python3
"""Per (kernel, grid size) effective clock and MFMA-busy fraction from ONE rocprofv3 --pmc pass that collected
SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE (the `mfma` pass of tools/pmc_traffic.sh):

  python tools/pmc_per_shape.py gpurun_out/pmc_<tag>_f32/mfma > profiles/<tag>_fwd_f32_per_shape.md

effective clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration (MI355X_MICROARCH.md, DVFS give-back); MFMA busy =
SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128).  Grid size separates the stage-0 launches (521 280 tokens, K = 192) from
the stage-1/2 launches (131 040 tokens, K = 384) of one kernel."""
import collections
import csv
import glob
import sys


def main(d):
    disp = collections.defaultdict(dict)
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            e = disp[(path, r["Dispatch_Id"])]
            e["name"] = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
            e["grid"] = int(r["Grid_Size"])
            e[r["Counter_Name"]] = float(r["Counter_Value"])
            e["dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for e in disp.values():
        if "GRBM_GUI_ACTIVE" not in e:
            continue
        a = agg[(e["name"][:60], e["grid"])]
        a[0] += 1
        a[1] += e["dur"]
        a[2] += e["GRBM_GUI_ACTIVE"]
        a[3] += e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    print("| kernel | grid (threads) | launches | avg us | eff. clock GHz | MFMA busy |")
    print("|---|---|---|---|---|---|")
    for k, a in sorted(agg.items(), key=lambda x: -x[1][1])[:24]:
        print(f"| {k[0]} | {k[1]} | {a[0]} | {a[1] / a[0]:.1f} | {a[2] / 8 / (a[1] * 1e3):.2f} | {a[3] / (a[2] * 128) if a[2] else 0:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1])
