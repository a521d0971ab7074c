#!/usr/bin/env python3
"""Register / scratch / LDS budget of every kernel of libpangu_hip.so, from the compiler's own resource report:

  python tools/codeobj_resources.py > profiles/r03_codeobj_resources.md        (no GPU needed: hipcc cross-compiles gfx950)

Each csrc/*.hip is compiled with the Makefile's flags plus -Rpass-analysis=kernel-resource-usage (the numbers that end up in
the code object's kernel descriptors: .vgpr_count, .agpr_count, .private_segment_fixed_size, .group_segment_fixed_size).
ScratchSize > 0 = spills (or a runtime-indexed private array): the VERDICT r2 item 4 rows are marked."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pangu-pytorch_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage",
         "--cuda-device-only", "-c"]
KEYS = ["VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "SGPRs Spill", "VGPRs Spill", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]",
        "SGPRs"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.strip().split("\n")


def scratch_in_loops(src):
    """{mangled kernel name: (scratch instructions, of which inside a loop, [(MFMAs, scratch instructions) per loop])} from the
    assembly: a loop = a backward branch to a label; what matters for speed is whether the K / window loop touches scratch."""
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.run([HIPCC] + [f for f in FLAGS if f not in ("-c", "-Rpass-analysis=kernel-resource-usage")] + ["-S", src, "-o", tmp.name],
                       capture_output=True, text=True)
        lines = open(tmp.name).read().split("\n")
    out, cur, body = {}, None, []
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur, body = m.group(1), []
        elif cur is not None:
            body.append(l)
            if "s_endpgm" in l:
                labels = {mm.group(1): i for i, b in enumerate(body) for mm in [re.match(r"^(\.LBB\d+_\d+):", b)] if mm}
                loops = []
                for i, b in enumerate(body):
                    mm = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", b)
                    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                        loops.append((labels[mm.group(1)], i))
                sc = [i for i, b in enumerate(body) if "scratch_" in b]
                inl = [i for i in sc if any(a <= i <= e for a, e in loops)]
                out[cur] = (len(sc), len(inl), [(sum("v_mfma" in x for x in body[a:e]), sum("scratch_" in x for x in body[a:e])) for a, e in loops])
                cur = None
    return out


def main():
    rows = []
    loops = {}
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        loops.update(scratch_in_loops(src))
        with tempfile.NamedTemporaryFile(suffix=".o") as tmp:
            r = subprocess.run([HIPCC] + FLAGS + [src, "-o", tmp.name], capture_output=True, text=True)
        cur = None
        for line in r.stderr.splitlines():
            m = re.search(r"remark: (?:Function Name: (\S+)|\s+([^:]+): (\S+))", line)
            if not m:
                continue
            if m.group(1):
                cur = {"file": os.path.basename(src), "name": m.group(1)}
                rows.append(cur)
            elif cur is not None:
                cur[m.group(2).strip()] = m.group(3)
    names = demangle([r["name"] for r in rows])
    print("# Code-object resources of every kernel in libpangu_hip.so (gfx950, hipcc -O3; tools/codeobj_resources.py)\n")
    print("VGPR + AGPR share one 512-entry file per SIMD lane (allocation granule 8): waves/SIMD = min(8, 512 // alloc).  "
          "`scratch` = private_segment_fixed_size in bytes per lane (0 = no spill).\n")
    print("| file | kernel | VGPR | AGPR | scratch B/lane | VGPR spills | waves/SIMD | static LDS B | SGPR |")
    print("|---|---|---|---|---|---|---|---|---|")
    for r, n in zip(rows, names):
        n = n.replace("(anonymous namespace)::", "").replace("void ", "")
        n = n.split("(")[0]
        print(f"| {r['file']} | `{n[:90]}` | {r.get('VGPRs', '')} | {r.get('AGPRs', '')} | {r.get('ScratchSize [bytes/lane]', '')} | "
              f"{r.get('VGPRs Spill', '')} | {r.get('Occupancy [waves/SIMD]', '')} | {r.get('LDS Size [bytes/block]', '')} | {r.get('SGPRs', '')} |")
    spilled = [(r, n) for r, n in zip(rows, names) if r.get("ScratchSize [bytes/lane]", "0") not in ("0", "")]
    print(f"\n{len(rows)} kernels, {len(spilled)} with scratch:")
    print("(scratch instructions: total / inside a loop; per loop (MFMAs, scratch instructions) -- a kernel whose MFMA loops show 0 "
          "spills only in its prologue / epilogue)\n")
    for r, n in spilled:
        tot, inl, per = loops.get(r["name"], (None, None, []))
        per = [x for x in per if x[0] or x[1]]
        print(f"* `{n.replace('(anonymous namespace)::', '').split('(')[0][:100]}` ({r['file']}): {r['ScratchSize [bytes/lane]']} B/lane, "
              f"{r.get('VGPRs Spill', '?')} VGPRs spilled; scratch instructions {tot} / {inl} in loops; loops {per}")


if __name__ == "__main__":
    main()
