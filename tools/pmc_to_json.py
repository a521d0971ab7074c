#!/usr/bin/env python3
"""Per-launch HBM traffic and MFMA utilisation of the forward's kernel families from separate rocprofv3 --pmc passes
(driven by tools/pmc_traffic.sh, which also takes a kernel trace of the same command).

  python tools/pmc_to_json.py <dir with trace/ fetch/ write/ mfma/> <f32|bf16>  > pmc_traffic_<dtype>.json
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md).
mfma_busy = sum SQ_VALU_MFMA_BUSY_CYCLES / (sum GRBM_GUI_ACTIVE * 128)   (GUI_ACTIVE is summed over the 8 XCDs, the MFMA
counter over the 1024 SIMDs).  The JSON records the git commit and the sha256 of every kernel source a family comes from:
bench.py recomputes those hashes and reports `traffic: null, traffic_stale: true` when a source changed since the
counters were taken.  Exits non-zero if a family priced here does not appear in the kernel trace of the same command."""
import collections
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pangu-pytorch_amd", "csrc")
# family -> (kernel-name substrings, source files)
FAMILIES = {
    "f32": {"gemm": (("gemm_tn_f32_dma_kernel", "gemm_tn_f32_kernel"), ("gemm_f32_dma.hip", "gemm_f32.hip")),
            "gemm_ln": (("gemm_ln_residual_f32",), ("gemm_ln_f32_dma.hip",)),
            "attn": (("window_attn_f32_kernel",), ("attn_f32.hip",))},
    "bf16": {"mlp_fused": (("mlp_ln_residual_bf16_kernel",), ("mlp_fused_bf16.hip",)),
             "attn_qkv": (("window_attn_qkv_bf16_kernel",), ("attn_bf16.hip", "attn_bf16_tile.h")),
             "gemm": (("gemm_tn_bf16", "gemm_ws_bf16"), ("gemm_bf16.hip", "gemm_ws_bf16.hip")),
             "gemm_ln": (("gemm_ln_residual_bf16",), ("gemm_ln_bf16.hip",))},
}


SHARED_SOURCES = ["common.h", os.path.join("..", "..", "include", "pangu_hip.h")]      # as bench.py: hashed with every family


def source_sha(files):
    h = hashlib.sha256()
    for f in list(files) + SHARED_SOURCES:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def main(d, dtype):
    fams = FAMILIES[dtype]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(lambda: collections.defaultdict(set))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            for fam, (names, _) in fams.items():
                if any(n in r["Kernel_Name"] for n in names):
                    agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
                    disp[fam][r["Counter_Name"]].add(r["Dispatch_Id"])
    traced, traced_ns = collections.Counter(), collections.Counter()
    for path in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            traced[r["Name"]] += int(r["Calls"])
            traced_ns[r["Name"]] += float(r.get("TotalDurationNs") or 0.0)
    out, missing = {}, []
    for fam, (names, files) in fams.items():
        launched = sum(c for k, c in traced.items() if any(n in k for n in names))
        c = agg.get(fam)
        if not c or not launched:
            missing.append(fam)
            continue
        n = len(disp[fam]["FETCH_SIZE"]) or 1
        rd = c["FETCH_SIZE"] * 2 * 1024 / n
        wr = c["WRITE_SIZE"] * 1024 / max(len(disp[fam]["WRITE_SIZE"]), 1)
        out[fam] = {"kernels": list(names), "launches_in_trace": launched, "launches": n, "hbm_read_bytes_per_launch": rd,
                    "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                    "mfma_busy_frac": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 128) if c["GRBM_GUI_ACTIVE"] else None,
                    "source_sha": source_sha(files), "sources": list(files)}
        # issue-side counters (pass `valu`): wave-level vector-ALU instructions (MFMAs included) and waves per launch; the effective
        # clock of the family = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the trace's duration of the same launches
        ns = sum(t for k, t in traced_ns.items() if any(x in k for x in names))
        if c.get("SQ_INSTS_VALU") and disp[fam]["SQ_INSTS_VALU"]:
            nv = len(disp[fam]["SQ_INSTS_VALU"])
            out[fam]["valu_insts_per_launch"] = c["SQ_INSTS_VALU"] / nv
            out[fam]["waves_per_launch"] = c["SQ_WAVES"] / max(len(disp[fam]["SQ_WAVES"]), 1)
        if ns > 0 and c["GRBM_GUI_ACTIVE"]:
            out[fam]["avg_launch_us_in_trace"] = ns / launched / 1e3
            out[fam]["eff_clock_ghz"] = (c["GRBM_GUI_ACTIVE"] / 8 / max(len(disp[fam]["GRBM_GUI_ACTIVE"]), 1)) / (ns / launched)
    try:
        commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        commit = ""
    out["commit"] = commit or os.environ.get("PANGU_COMMIT", "unknown (the GPU box has no .git; see the profile's file name / git log)")
    out["command"] = f"python3 tools/profile_fwd.py {dtype} 3   (rocprofv3 passes: FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE | SQ_INSTS_VALU SQ_WAVES; FETCH_SIZE x2)"
    print(json.dumps(out, indent=1))
    if missing:
        sys.stderr.write(f"kernel families without counters or not launched by the traced command: {missing}\n")
        sys.exit(1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "f32")
