#!/usr/bin/env python3
"""Build-time check of the hand-placed LDS waits around the inline-asm transposing reads (`ds_read_b64_tr_b16` is inline
asm in csrc/wgrad_bf16_dma.hip, so the compiler knows neither that its destination registers are written asynchronously nor that an
`s_waitcnt lgkmcnt(N)` with N > 0 assumes LDS operations only -- scalar-memory loads return out of order).  Disassembles the file for
gfx950 (no GPU needed) and, for every kernel, walks each basic-block run that contains transposing reads with a model of the LGKM
queue:   * no s_load / s_buffer_load may be in flight while a transposing read is (an SMEM op under a counted lgkmcnt wait would
           make the count under-wait);
         * no instruction may read or overwrite a transposing read's destination registers before an s_waitcnt lgkmcnt(N) has
           retired it (LDS operations retire in issue order: a wait for N leaves the N youngest outstanding).
Usage: python tools/check_asm_waits.py [file.hip ...]   (exit code 1 on a violation)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pangu-pytorch_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def disassemble(src):
    out = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-S", "--cuda-device-only",
                          "-o", "-", src], capture_output=True, text=True, check=True)
    return out.stdout


def regs(tok):
    """v12 -> {12}; v[4:7] -> {4,5,6,7}; anything else -> empty."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def check_kernel(name, lines):
    errors = []
    queue = []                      # outstanding LGKM operations in issue order: (kind, dest registers, line no)
    n_tr = 0
    for no, raw in enumerate(lines):
        l = raw.split(";")[0].strip()
        if not l or l.endswith(":") or l.startswith("."):
            if l.endswith(":") and queue:
                # a label with operations in flight: keep the queue (fall-through is the common case); branch targets inside the
                # checked loops are loop headers reached with the same queue state
                pass
            continue
        op, _, rest = l.partition(" ")
        toks = [t.strip() for t in rest.split(",")] if rest else []
        used = set()
        for t in toks:
            used |= regs(t.split(" ")[0])
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", l)
            if m:
                keep = int(m.group(1))
                queue = queue[len(queue) - keep:] if keep else []
            continue
        pending = set()
        for kind, dst, _ in queue:
            if kind == "tr":
                pending |= dst
        if pending & used:
            errors.append(f"{name}: line {no}: `{l}` touches v{sorted(pending & used)} while a ds_read_b64_tr_b16 into them is still in flight")
        if op.startswith("ds_read_b64_tr") or op.startswith("ds_read_tr"):
            n_tr += 1
            if any(k == "smem" for k, _, _ in queue):
                errors.append(f"{name}: line {no}: transposing read issued with a scalar-memory load in flight (out-of-order return)")
            queue.append(("tr", regs(toks[0]), no))
        elif op.startswith("ds_"):                     # every other LDS operation counts in lgkmcnt, in order
            queue.append(("lds", set(), no))
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            if any(k == "tr" for k, _, _ in queue):
                errors.append(f"{name}: line {no}: `{op}` issued while transposing reads are in flight (SMEM returns out of order)")
            queue.append(("smem", set(), no))
        elif op in ("s_endpgm",):
            queue = []
    return n_tr, errors


def check_file(src):
    asm = disassemble(src)
    kernels = re.split(r"\n(?=[_A-Za-z0-9]+:\s*; @)", asm)
    total, errs, seen = 0, [], 0
    for k in kernels:
        head = k.split("\n", 1)[0]
        if "; @" not in head:
            continue
        body = k.split(".end_amdhsa_kernel")[0] if ".end_amdhsa_kernel" in k else k
        if "ds_read_b64_tr_b16" not in body and "ds_read_tr" not in body:
            continue
        seen += 1
        n, e = check_kernel(head.split(":")[0], body.split("\n"))
        total += n
        errs += e
    return seen, total, errs


if __name__ == "__main__":
    files = sys.argv[1:] or [os.path.join(CSRC, "wgrad_bf16_dma.hip")]
    bad = 0
    for f in files:
        seen, total, errs = check_file(f)
        print(f"{os.path.basename(f)}: {seen} kernels with transposing reads, {total} reads checked, {len(errs)} violations")
        for e in errs[:20]:
            print("  " + e)
        bad += len(errs)
    sys.exit(1 if bad else 0)
