#!/usr/bin/env python3
"""Join separate rocprofv3 --pmc passes of the same command into one per-kernel table (markdown on stdout):

  python tools/pmc_train_table.py <dir with p1 (FETCH_SIZE), p2 (WRITE_SIZE), p3 (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE)>

FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md: the counter tallies 128-B requests at 64 B); FETCH/WRITE_SIZE
are in KiB; MFMA busy = SUM(MFMA_BUSY_CYCLES) / (SUM(GUI_ACTIVE) * 128); durations from the kernel trace of pass 3; effective
clock = SUM(GUI_ACTIVE) / 8 XCDs / SUM(duration) of the same pass (the clock the chip holds under that kernel)."""
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


def _totals(root):
    rd, _ = counters(os.path.join(root, "p1"))
    wr, _ = counters(os.path.join(root, "p2"))
    return (sum(2.0 * v.get("FETCH_SIZE", 0.0) * 1024 for v in rd.values()),
            sum(v.get("WRITE_SIZE", 0.0) * 1024 for v in wr.values()))


def step_totals(root, steps):
    """Whole-step HBM bytes (FETCH_SIZE x2 + WRITE_SIZE over every kernel).  `root` holds the passes of `profile_train.py <dt>
    <steps> 1`; when `root`_w0 exists (the passes of `profile_train.py <dt> 0 1`: model construction, weight init, the warm-up
    step with Adam's state allocation) it is SUBTRACTED, so the figure is the steady-state step alone; without it the warm-up
    step counts as one more step (init kernels included: a few GB too many per step)."""
    r, w = _totals(root)
    if os.path.isdir(root + "_w0"):
        r0, w0 = _totals(root + "_w0")
        r, w, n, how = r - r0, w - w0, steps, "steady-state steps only (warm-up + init trace subtracted)"
    else:
        n, how = steps + 1, "warm-up step and init kernels included"
    return {"hbm_read_bytes_per_step": r / n, "hbm_write_bytes_per_step": w / n, "hbm_bytes_per_step": (r + w) / n,
            "steps_in_trace": n, "accounting": how}


def write_json(path, roots):
    """roots: {"f32": (dir, steps), "bf16": (dir, steps)} -> profiles/pmc_train.json (bench.py's training roofline blocks)."""
    import hashlib
    import json
    import subprocess
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(here, "pangu-pytorch_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip"))) + [os.path.join(csrc, "common.h"),
                                                               os.path.join(csrc, "..", "..", "include", "pangu_hip.h")]:
        h.update(open(f, "rb").read())
    out = {k: step_totals(d, n) for k, (d, n) in roots.items()}
    out["source_sha"] = h.hexdigest()[:16]
    try:
        commit = subprocess.run(["git", "-C", here, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        commit = ""
    out["commit"] = commit or os.environ.get("PANGU_COMMIT", "unknown")
    out["command"] = "python3 tools/profile_train.py <dtype> 2 1 under rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes; FETCH_SIZE x2)"
    json.dump(out, open(path, "w"), indent=1)


def main(root):
    rd, n1 = counters(os.path.join(root, "p1"))
    wr, _ = counters(os.path.join(root, "p2"))
    mf, _ = counters(os.path.join(root, "p3"))
    tot, cnt = durations(os.path.join(root, "p3"))
    print("| kernel | launches | avg us | read MB/launch | write MB/launch | HBM TB/s | MFMA busy | eff. clock GHz |")
    print("|---|---|---|---|---|---|---|---|")
    for k in sorted(tot, key=lambda k: -tot[k])[:40]:
        n = cnt[k]
        us = tot[k] / n
        r = 2.0 * rd[k].get("FETCH_SIZE", 0.0) * 1024 / max(n1.get(k, n), 1) / 1e6
        w = wr[k].get("WRITE_SIZE", 0.0) * 1024 / max(n1.get(k, n), 1) / 1e6
        gui = mf[k].get("GRBM_GUI_ACTIVE", 0.0)
        busy = mf[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 128) if gui else 0.0
        # effective shader clock under this kernel: GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back);
        # reads high on dispatches shorter than ~0.3 ms (the counter also ticks over the dispatch's ramp-up / drain)
        ghz = gui / 8.0 / (tot[k] * 1e3) if tot[k] else 0.0
        print(f"| {k} | {n} | {us:.0f} | {r:.0f} | {w:.0f} | {(r + w) / us:.2f} | {busy:.2f} | {ghz:.2f} |")


if __name__ == "__main__":
    if sys.argv[1] == "--json":          # --json <out.json> <f32 dir> <bf16 dir> <steps in each trace>
        write_json(sys.argv[2], {"f32": (sys.argv[3], int(sys.argv[5])), "bf16": (sys.argv[4], int(sys.argv[5]))})
    else:
        main(sys.argv[1])
