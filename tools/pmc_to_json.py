#!/usr/bin/env python3
"""Per-launch HBM traffic and MFMA utilisation of a kernel family from separate rocprofv3 --pmc passes.

  python tools/pmc_to_json.py <dir with the counter_collection CSVs> > profiles/pmc_traffic_f32.json
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md).
mfma_busy = sum SQ_VALU_MFMA_BUSY_CYCLES / (sum GRBM_GUI_ACTIVE * 128)   (GUI_ACTIVE is summed over the 8 XCDs, the MFMA counter
over the 1024 SIMDs)."""
import collections, csv, glob, json, os, sys

FAMILIES = {"gemm": ("gemm_tn_f32_dma_kernel", "gemm_tn_f32_kernel"), "gemm_ln": ("gemm_ln_residual_f32",),
            "attn": ("window_attn_f32_kernel",)}


def main(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(lambda: collections.defaultdict(set))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            for fam, names in FAMILIES.items():
                if any(n in r["Kernel_Name"] for n in names):
                    agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
                    disp[fam][r["Counter_Name"]].add(r["Dispatch_Id"])
    out = {}
    for fam, c in agg.items():
        n = len(disp[fam]["FETCH_SIZE"]) or 1
        rd = c["FETCH_SIZE"] * 2 * 1024 / n
        wr = c["WRITE_SIZE"] * 1024 / max(len(disp[fam]["WRITE_SIZE"]), 1)
        out[fam] = {"launches": n, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                    "hbm_bytes_per_launch": rd + wr,
                    "mfma_busy_frac": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 128) if c["GRBM_GUI_ACTIVE"] else None,
                    "note": "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KiB->bytes; separate rocprofv3 --pmc passes over "
                            "`bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-bf16`"}
    out["source"] = "profiles/r01_fwd_f32_rocprof_summary.md (tools/pmc_to_json.py over the --pmc passes FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE)"
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
