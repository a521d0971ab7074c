#!/bin/bash
# Exact per-step kernel list of the training step: kernel-trace statistics of (1 warm-up + 3 steps) minus those of the warm-up
# alone (model construction, weight init, shadow building, Adam state allocation), divided by 3.
#   bash tools/per_step_kernels.sh [bf16|f32] > profiles/<tag>_train_<dtype>_per_step.md      (through gpurun)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$(pwd)
export TMPDIR=/tmp
DT=${1:-bf16}
rm -rf /tmp/psk0 /tmp/psk3
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/psk0 -- python3 $ROOT/tools/profile_train.py $DT 0 1 > /tmp/psk0.log 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/psk3 -- python3 $ROOT/tools/profile_train.py $DT 3 1 > /tmp/psk3.log 2>&1 )
python3 - "$(find /tmp/psk0 -name '*kernel_stats.csv' | head -1)" "$(find /tmp/psk3 -name '*kernel_stats.csv' | head -1)" "$DT" <<'PY'
import csv, sys
def load(p):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(p))}
a, b = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k, (c, t) in b.items():
    c0, t0 = a.get(k, (0, 0.0))
    if c - c0 > 0:
        rows.append(((t - t0) / 3e3, (c - c0) / 3, k))
rows.sort(reverse=True)
print(f"Per-step kernels of the {sys.argv[3]} training step (tools/per_step_kernels.sh: trace of 1 + 3 steps minus trace of the warm-up; "
      f"durations under the profiler).  Sum: {sum(r[0] for r in rows) / 1e3:.2f} ms\n")
print("| us / step | launches / step | kernel |\n|---|---|---|")
for t, c, k in rows:
    k = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void at::native::", "at::")
    print(f"| {t:.1f} | {c:.1f} | `{k.split('(')[0][:110]}` |")
PY
