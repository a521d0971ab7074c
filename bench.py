#!/usr/bin/env python3
"""Headline benchmark: PanguModel forward steps/sec on synthetic 721x1440x(13*5+4) fields, fp32, one sample per GPU.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A step = one `PanguModel.forward` (BASELINE.json configs[1]: single-GPU fp32 forward through the HIP path) on one
sample resident in HBM.  With N ranks every rank runs its own sample (pure data parallel, no data-path
collective in inference): value = N*K / max-over-ranks time, "scaling": "weak".

Rank 0 prints ONE JSON line with the contract keys plus
  roofline      live HIP-event timing of the dominant kernel (the f32 MFMA projection GEMM) over the timed region
  cpu_baseline  the CPU oracle (oracle/pangu_oracle.py, a port of the reference's algorithm) timed on this host on a
                bounded sample (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FWD_GFLOP = 8421.2          # SURVEY.md §8(a) ledger: dense-contraction GFLOP per forward step on the reference's PADDED shapes
FWD_GFLOP_EXEC = 8302.3     # executed here: the QKV / output projections skip the zero-pad rows (4 x 4.25 + 12 x 8.49 GFLOP less)
XGMI_LINK_GBS = 153.0       # one xGMI link, one direction (7 links per GPU, point to point)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
PEAK_HBM_GBS = 8000.0
ACHIEVABLE_HBM_GBS = 6290.0    # MI355X_MICROARCH.md: float4 copy
PEAK_BF16_MFMA_TFLOPS = 2500.0
SHARED_SOURCES = ["common.h", os.path.join("..", "..", "include", "pangu_hip.h")]      # hashed with every kernel family (tools/pmc_to_json.py too)


# executed dense-contraction GFLOP of ONE residual branch of one block, forward (projections on the unpadded tokens, the
# attention core on the padded windows): what a DropPath-dropped branch does NOT execute (layers.py:250-251)
_N0, _NP0, _N1, _NP1 = 521280, 535680, 131040, 138240
BRANCH_GFLOP = {192: (2e-9 * _N0 * 192 * 576 + 4e-9 * _NP0 * 144 * 192 + 2e-9 * _N0 * 192 * 192, 4e-9 * _N0 * 192 * 768),
                384: (2e-9 * _N1 * 384 * 1152 + 4e-9 * _NP1 * 144 * 384 + 2e-9 * _N1 * 384 * 384, 4e-9 * _N1 * 384 * 1536)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", choices=("auto", "sample", "full", "none"), default="auto",
                    help="sample: oracle block pairs on a half-longitude slice, extrapolated (~15 s); full: also the oracle's whole "
                         "forward once (45 s on a 256-core host, minutes on 8 cores); auto = full on hosts with >= 32 cores")
    ap.add_argument("--no-bf16", action="store_true", help="skip the secondary bf16 forward measurement")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary DDP training-step measurement")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the reference-call-convention timings (grad-enabled eval forward, the reference's own loop body)")
    ap.add_argument("--train-steps", type=int, default=6)
    ap.add_argument("--no-fed", action="store_true",
                    help="skip the host-fed training-step measurement (train_fed_from_host: pageable 573 MB batches through "
                         "data.DevicePrefetcher, beside the resident-batch step)")
    ap.add_argument("--rehearse-collective", action="store_true", help="(kept for old command lines: the rehearsal is ON by default at N = 1)")
    ap.add_argument("--no-rehearse", action="store_true",
                    help="N = 1 only: skip `allreduce_rehearsal_ms_per_step` (the training step timed again with a traffic generator "
                         "standing in for the bucketed all-reduce: dist.FlatGradSync(rehearse=..), per bucket two device-to-device copies "
                         "on a side stream confined to 32 / 64 workgroups)")
    ap.add_argument("--grad-sync", choices=("all_reduce", "reduce_scatter"), default="all_reduce",
                    help="the collective of dist.FlatGradSync (N > 1): bucketed all_reduce(AVG), or reduce_scatter + all_gather per bucket")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: THIS process (which has not touched the GPU -- torch is
    not even imported yet) starts `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a CHILD
    process, relays rank 0's JSON line and the child's return code.  No exec of a GPU-initialised process anywhere."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)      # stderr passes through
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    for ln in r.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return r.returncode if (r.returncode != 0 or lines) else 1


def cpu_baseline():
    """Time the CPU oracle on a half-longitude slice of each stage (10-20 s of CPU work) and extrapolate to one forward step."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))      # the ONLY place bench.py touches oracle/: the CPU baseline leg
    import cases
    import pangu_oracle as O
    # pick the thread count torch's CPU backend runs this workload fastest with on this host (the reference's
    # scripts pin 16: inference/test_main.py:60); huge hosts lose time to oversubscription at cpu_count threads
    st = cases.STAGES[384]
    xs, ps, pre = cases.block_input(384, 36), cases.block_params(384, False), cases.block_prefix(384, False)
    best = (1e30, 1)
    for nt in (8, 16, 32, 64, 128):
        if nt > (os.cpu_count() or 1):
            break
        torch.set_num_threads(nt)
        with torch.no_grad():
            O.earth_block(ps, pre, xs, st["Z"], st["H"], 36, st["heads"], False)
            t0 = time.perf_counter()
            O.earth_block(ps, pre, xs, st["Z"], st["H"], 36, st["heads"], False)
            best = min(best, (time.perf_counter() - t0, nt))
    threads = best[1]
    torch.set_num_threads(threads)
    t_total = 0.0
    parts = {}
    with torch.no_grad():
        for C, W, pairs, full_w in ((192, 180, 2, 360), (384, 96, 6, 180)):
            st = cases.STAGES[C]
            x = cases.block_input(C, W)
            t_pair = 0.0
            for roll in (False, True):
                p = cases.block_params(C, roll)
                pre = cases.block_prefix(C, roll)
                O.earth_block(p, pre, x[:, :st["Z"] * st["H"] * 12], st["Z"], st["H"], 12, st["heads"], roll)  # warm
                t0 = time.perf_counter()
                O.earth_block(p, pre, x, st["Z"], st["H"], W, st["heads"], roll)
                t_pair += time.perf_counter() - t0
            parts[C] = t_pair
            t_total += (full_w / W) * pairs * t_pair
    return {
        "value": 1.0 / t_total, "unit": "forward steps/s", "cores": threads, "host_cpus": os.cpu_count(), "kind": "port",
        "sample": ("oracle earth_block pairs (unshifted+shifted) on a half-longitude slice: C=192 at 8x181x180 "
                   f"({parts[192]:.2f}s), C=384 at 8x91x96 ({parts[384]:.2f}s); extrapolated by longitude (x2, x1.875) to the 4+12 blocks of one "
                   "forward (embed/recover/resample excluded, <4% of FLOPs)"),
    }


def cpu_baseline_full(threads):
    """The CPU oracle's WHOLE forward (oracle/pangu_oracle.forward, the restatement of reference pangu_model.py:50-87) timed
    once on this host with `threads` threads, on the bench's synthetic shapes."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cases
    import pangu_oracle as O
    import synth
    torch.set_num_threads(threads)
    p = {k: synth.synth_param(k, sh) for k, sh in cases.model_param_shapes().items()}
    inp, inp_s, stats, maps, const_h = cases.model_inputs()
    with torch.no_grad():
        t0 = time.perf_counter()
        out, _ = O.forward(p, inp, inp_s, stats, maps, const_h)
        t = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    return t


def rollout_vs_reference(dev):
    """BASELINE configs[4] pinned to the REFERENCE: the 7 x 24 h rollout on the goldens' closed-form weights and
    inputs, fp32 and bf16, against the fingerprints of seven chained reference forwards (tests/golden/rollout7.npz, written by
    oracle/gen_golden.py rollout7 in the build container).  Checker leg: uses oracle/ (closed-form generators, fingerprint compare)
    and runs with the CPU baseline only (rank 0, N = 1)."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cases
    import synth
    import pangu_pytorch_amd as P
    from pangu_pytorch_amd import rollout as R
    gpath = os.path.join(ROOT, "tests", "golden", "rollout7.npz")
    if not os.path.exists(gpath):
        return None
    g = np.load(gpath)
    m = P.PanguModel(device=dev).to(dev).eval()
    m.load_state_dict(synth.fill_state_dict(cases.model_param_shapes(), str(dev)))
    inp, inp_s, stats, maps, const_h = cases.model_inputs(str(dev))
    s_mean, s_std, u_mean, u_std = stats
    sl = (s_mean.view(1, 4, 1, 1), s_std.view(1, 4, 1, 1), u_mean.reshape(13, 5).flip(0).t().reshape(1, 5, 13, 1, 1).contiguous(),
          u_std.reshape(13, 5).flip(0).t().reshape(1, 5, 13, 1, 1).contiguous())
    res = {"note": "max relative fingerprint error (samples, column sums, mass; relative to the largest reference value) of each step's "
                   "normalised upper-air output against the REFERENCE's own 7-step rollout on the same closed-form weights / inputs "
                   "(tests/golden/rollout7.npz); these weights are deliberately not contractive, so one step's error is carried on"}
    for key, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        m.set_compute_dtype(dt)
        _, _, hist = R.rollout(m, inp, inp_s, stats, maps, const_h, sl, steps=7, graph=True, keep=True)
        res[key] = [cases.compare_summary(o, g, f"rollout.step{k + 1}.out", 1.0) for k, (o, _) in enumerate(hist)]
        del hist
    apath = os.path.join(ROOT, "tests", "golden", "autocast.npz")
    if os.path.exists(apath):      # the yardstick for the bf16 list: the reference's own CPU autocast(bfloat16) rollout, same metric
        ac = np.load(apath)
        res["bf16_reference_autocast"] = [float(ac[f"rollout.step{k + 1}.err"][0]) for k in range(7)]
    del m
    torch.cuda.empty_cache()
    return res


def pmc_traffic(dtype, family):
    """Per-launch HBM bytes / MFMA-busy of a kernel family from the committed rocprofv3 PMC run (tools/pmc_traffic.sh ->
    profiles/pmc_traffic_<dtype>.json).  Counters cannot be read from inside this process; the JSON records the sha256 of the
    kernel sources it was taken on: if a source changed since, the numbers are reported as stale (traffic null)."""
    import hashlib
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", f"pmc_traffic_{dtype}.json")))
        e = j[family]
        h = hashlib.sha256()
        for f in list(e["sources"]) + SHARED_SOURCES:        # the kernel files of the family + what every kernel includes
            h.update(open(os.path.join(ROOT, "pangu-pytorch_amd", "csrc", f), "rb").read())
        stale = h.hexdigest()[:16] != e["source_sha"]
        return (None if stale else e["hbm_bytes_per_launch"]), (None if stale else e["mfma_busy_frac"]), stale, j.get("commit")
    except (OSError, KeyError, ValueError):
        return None, None, None, None


def pmc_train(tag):
    """HBM bytes per training step (sum over every kernel of the step: FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc
    passes of tools/profile_train.py, written by tools/pmc_train_table.py --json -> profiles/pmc_train.json), or None when the
    table is missing or any kernel source changed since it was taken."""
    import glob
    import hashlib
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "pmc_train.json")))
        e = j[tag]
        h = hashlib.sha256()
        csrc = os.path.join(ROOT, "pangu-pytorch_amd", "csrc")
        for f in sorted(glob.glob(os.path.join(csrc, "*.hip"))) + [os.path.join(csrc, f) for f in SHARED_SOURCES]:
            h.update(open(f, "rb").read())
        stale = h.hexdigest()[:16] != j.get("source_sha")
        return (None if stale else e), stale, j.get("commit")
    except (OSError, KeyError, ValueError):
        return None, None, None


def xgmi_topology():
    """The node's GPU link table as rocm-smi reports it (rank 0, once): hop and weight matrices + link types, parsed from
    `rocm-smi --showtopo --json`; on any failure the reason (and the head of the raw output) instead -- never an exception."""
    import re
    import subprocess
    try:
        r = subprocess.run(["rocm-smi", "--showtopo", "--json"], capture_output=True, text=True, timeout=60)
        raw = r.stdout.strip()
        j = json.loads(raw[raw.index("{"):raw.rindex("}") + 1])
        mats = {}

        def walk(o):
            if isinstance(o, dict):
                for k, v in o.items():
                    m = re.search(r"(Weight|Hops|Link [Tt]ype)[^0-9]*(\d+)\D+(\d+)", str(k))
                    if m and not isinstance(v, (dict, list)):
                        mats.setdefault(m.group(1).lower().replace(" ", "_"), {})[(int(m.group(2)), int(m.group(3)))] = v
                    else:
                        walk(v)
            elif isinstance(o, list):
                for v in o:
                    walk(v)
        walk(j)
        n = 1 + max((max(a, b) for m in mats.values() for a, b in m), default=0)
        out = {"gpus": n, "source": "rocm-smi --showtopo --json"}
        for name, m in mats.items():
            mat = [[None] * n for _ in range(n)]
            for (a, b), v in m.items():
                try:
                    v = int(v)
                except (TypeError, ValueError):
                    pass
                mat[a][b] = v
                if mat[b][a] is None:
                    mat[b][a] = v
            out[name] = mat
        if not mats:
            out["raw_head"] = raw[:600]
        return out
    except Exception as e:
        return {"error": repr(e)[:200]}


def fed_from_host(model, opt, dev, consts, grad_sync, steps, rank):
    """The training step FED FROM THE HOST (reference models/pangu_sample.py:41-43 `.to(device)` of four pageable
    tensors = 573 MB per step, :77 `loss.item()` every step, era5_data/utils_data.py:117 level reversal in the reader): three
    distinct full-size pageable samples in FILE level order, cycled, through
      resident            the same samples already on the device (what `ddp_train*` / `ddp_samples_per_s` time), loss.item() per step
      fed_free_running    data.DevicePrefetcher(fuse_flip=True): worker-thread staging into page-locked slots, copy-engine upload two
                          batches ahead, the level flip an address inside patch_embed_gather / the loss kernel; no host sync
      fed_item_per_step   the same with the reference's per-step loss.item()
      reference_to_device the reference's own loop body: `.to(device)` of the four pageable tensors on the training thread, then the
                          step, then loss.item()  (its reader's host-side `[::-1]` copy is NOT timed: it runs in DataLoader workers)
    Returns ms per step of each and the pipeline's measured rates."""
    import torch
    from pangu_pytorch_amd import data, train
    stats, maps, const_h = consts
    g = torch.Generator().manual_seed(3000 + rank)
    shapes = ((1, 5, 13, 721, 1440), (1, 4, 721, 1440), (1, 5, 13, 721, 1440), (1, 4, 721, 1440))
    host = [tuple(torch.rand(sh, generator=g) * 2 - 1 for sh in shapes) for _ in range(3)]
    n_warm, n = 3, max(6, steps)
    sync = torch.cuda.synchronize

    def timed(batches, item, rev, to_device=False):
        torch.manual_seed(4242 + rank)          # every variant sees the SAME DropPath draws (a dropped stage-0 branch is worth ms)
        t0 = None
        for k, b in enumerate(batches):
            if k == n_warm:
                sync()
                t0 = time.perf_counter()
            if to_device:
                b = tuple(t.to(dev) for t in b)                      # pangu_sample.py:41-43 (pageable source: a blocking copy each)
            loss = train.train_step(model, opt, b, stats, maps, const_h, grad_sync=grad_sync, levels_reversed=rev)
            if item:
                loss.item()                                          # pangu_sample.py:77
        sync()
        return (time.perf_counter() - t0) / n * 1e3

    cyc = [host[i % 3] for i in range(n_warm + n)]
    res = {}
    resident = [tuple(t.to(dev) for t in b) for b in host]
    res["resident_item_per_step_ms"] = timed([resident[i % 3] for i in range(n_warm + n)], True, True)
    del resident
    pf = data.DevicePrefetcher(cyc, dev, flip_levels=True, fuse_flip=True, depth=2, reuse_device_buffers=True)
    res["fed_free_running_ms"] = timed(pf, False, pf.levels_reversed)
    res["fed_item_per_step_ms"] = timed(pf, True, pf.levels_reversed)
    sync()
    res["pipeline"] = pf.summary()
    res["reference_to_device_item_per_step_ms"] = timed(cyc, True, True, to_device=True)
    res["fed_over_resident"] = res["fed_item_per_step_ms"] / res["resident_item_per_step_ms"]
    res["bytes_per_step"] = sum(t.numel() * 4 for t in host[0])
    res["host_cpus"] = os.cpu_count()
    res["note"] = ("ms per training step, DropPath on with the SAME draws in all four loops (their own seed: compare the four numbers with each "
                   "other, not with the section's ms_per_step), three distinct pageable host samples in file (ascending) level order cycled; "
                   "needed host->device rate = bytes_per_step / step time; pipeline.host_stage_GBps is the page-locked staging copy "
                   "(pangu_host_copy, pipeline.copy_threads threads), pipeline.h2d_GBps the copy engine, consumer_wait what the training "
                   "thread waited for batches (both fed loops)")
    return res


def bf16_roofline(fams, eager_ms):
    """Per kernel family of the bf16 forward: live HIP-event launch times against the roof that BINDS it.
      mlp       mlp_ln_residual_bf16_kernel (whole MLP branch, hidden on chip)                  -> bf16 MFMA peak
      attn_qkv  window_attn_qkv_bf16_kernel (QKV projection + QK^T + bias + mask + softmax + PV) -> the SIMD's vector-issue port:
                SQ_INSTS_VALU (wave-level, from the committed PMC pass) minus the MFMAs and the exps, x 4 issue cycles, + the exps
                x 8, over 1024 SIMDs at the family's measured clock = the time the kernel cannot go below however idle the matrix pipe
      gemm_ln   gemm_ln_residual_bf16_kernel (output projection + LayerNorm + residual)         -> HBM (6.29 TB/s achievable)
      gemm      the plain projections (down / up-sampling, patch embed / recover)                -> HBM / MFMA ridge
    fams: name -> (total ms, total algorithmic FLOP, launches) of the instrumented pass."""
    pm_name = {"mlp": "mlp_fused", "attn_qkv": "attn_qkv", "gemm_ln": "gemm_ln", "gemm": "gemm"}
    out = {"peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "hbm_achievable_GBps": ACHIEVABLE_HBM_GBS,
           "timing": "HIP-event pairs around each launch in one more EAGER pass over the same K steps (the bf16 headline is the hipGraph replay)",
           "eager_ms_per_step": eager_ms}
    for name, (ms, flop, n) in fams.items():
        if not n:
            continue
        traffic, busy, stale, commit = pmc_traffic("bf16", pm_name[name])
        avg_ms = ms / n
        ach = flop / (ms * 1e-3) / 1e12
        e = {"launches": n, "avg_launch_ms": avg_ms, "algorithmic_flop_per_launch": flop / n, "achieved": ach,
             "frac": ach / PEAK_BF16_MFMA_TFLOPS, "traffic": traffic, "traffic_stale": stale, "mfma_busy_frac_pmc": busy,
             "hbm_floor_ms": traffic / (ACHIEVABLE_HBM_GBS * 1e9) * 1e3 if traffic else None, "bound": "mfma"}
        if e["hbm_floor_ms"] and e["hbm_floor_ms"] > flop / n / (PEAK_BF16_MFMA_TFLOPS * 1e12) * 1e3:
            e["bound"] = "hbm"
            e["x_hbm_floor"] = avg_ms / e["hbm_floor_ms"]
        if name == "attn_qkv":
            try:
                j = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_bf16.json")))["attn_qkv"]
                v, clk = j["valu_insts_per_launch"], j["eff_clock_ghz"]
                # per forward: 4 launches at C = 192 (6 heads, Np = 535680 padded tokens) and 12 at C = 384 (12 heads, Np = 138240);
                # one v_exp per score and lane-row: scores / 64 wave-level instructions; MFMAs: v_mfma_f32_16x16x32_bf16 = 16384 FLOP
                exps = (4 * _NP0 * 144 * 6 + 12 * _NP1 * 144 * 12) / 16 / 64
                mfma = flop / n / 16384.0
                cyc = ((v - mfma - exps) * 4.0 + exps * 8.0) / 1024.0
                e.update(bound="valu", valu_issue_bound_ms=cyc / (clk * 1e6), valu_insts_per_launch=v, mfma_insts_per_launch=mfma,
                         exp_insts_per_launch=exps, eff_clock_ghz=clk, x_valu_issue_bound=avg_ms / (cyc / (clk * 1e6)),
                         valu_note="issue cycles per SIMD = ((SQ_INSTS_VALU - MFMAs - exps) x 4 + exps x 8) / 1024 SIMDs; "
                                   "counters from profiles/pmc_traffic_bf16.json (stale flag as `traffic_stale`)")
            except (OSError, KeyError, ValueError, ZeroDivisionError):
                e.update(bound="valu", valu_issue_bound_ms=None,
                         valu_note="profiles/pmc_traffic_bf16.json carries no SQ_INSTS_VALU pass (tools/pmc_traffic.sh)")
        out[name] = e
    return out


def synthetic_inputs(dev, seed):
    """One synthetic ERA5-shaped sample, resident in HBM: upper-air (1,5,13,721,1440), surface (1,4,721,1440), O(1)
    values, non-trivial normalisation statistics, the three constant maps and const_h (reference pangu_model.py:60-66)."""
    import torch
    g = torch.Generator(device=dev).manual_seed(seed)
    u = lambda shape, scale=1.0, shift=0.0: (torch.rand(shape, generator=g, device=dev) * 2 - 1) * scale + shift
    inp, inp_s = u((1, 5, 13, 721, 1440)), u((1, 4, 721, 1440))
    stats = (u((4,), 0.3), u((4,), 0.2, 1.2), u((13, 1, 1, 5), 0.3), u((13, 1, 1, 5), 0.2, 1.2))
    return inp, inp_s, stats, u((1, 3, 724, 1440)), u((1, 1, 1, 13, 721, 1440))


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))          # plain `python bench.py --gpus N`: spawn the N ranks as a child process
    import torch
    n_dev = torch.cuda.device_count()
    local_rank = local_rank % max(n_dev, 1)      # (functional tests may run several ranks on one GPU with gloo)
    torch.cuda.set_device(local_rank)
    dist = None
    backend = None
    if world > 1 or os.environ.get("PANGU_DIST_FORCE") == "1":     # FORCE: run the RCCL path on a 1-rank communicator
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("PANGU_DIST_BACKEND", "nccl")      # "nccl" = RCCL over xGMI; "gloo" only for tests
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import pangu_pytorch_amd as P
    from pangu_pytorch_amd import ops

    dev = torch.device("cuda", local_rank)
    torch.manual_seed(0)
    model = P.PanguModel(device=dev).to(dev).eval()          # random-init weights of the real architecture
    inp, inp_s, stats, maps, const_h = synthetic_inputs(dev, seed=1000 + rank)   # a different sample per rank

    def step():
        with torch.no_grad():
            return model(inp, inp_s, stats, maps, const_h)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()               # the HEADLINE loop is un-instrumented: nothing but the K steps between the barriers
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    # second pass over the same K steps with HIP-event pairs around every GEMM / attention launch (~170 event records per
    # step on the launch stream): the per-kernel roofline numbers come from here, its step time is reported beside the headline
    ops.timing_start()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed_instr = time.perf_counter() - t1
    gemm_ms, gemm_flop, gemm_launches, other = ops.timing_stop("linear", also=("linear_ln", "attn"))
    fused_ms, fused_flop, fused_launches = other["linear_ln"]      # GEMMs with the fused LayerNorm+residual epilogue
    attn_ms, attn_flop, attn_launches = other["attn"]              # fused window attention (QK^T + bias + mask + softmax + PV)
    assert torch.isfinite(out[0]).all()

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    lsync = torch.cuda.synchronize      # no collectives inside the try blocks: a rank that fails must not desynchronise the others
    # ---- secondary metric: bf16 inference forward (BASELINE configs[2]/[4] precision), same inputs
    bf16_res = None
    if not args.no_bf16:
      try:
        model.set_compute_dtype(torch.bfloat16)
        for _ in range(2):
            step()
        lsync()
        tb = time.perf_counter()
        for _ in range(args.steps):
            out_b = step()
        lsync()
        tb = time.perf_counter() - tb
        ref = out[0].double()
        drift = ((out_b[0].double() - ref).norm() / ref.norm()).item()
        # the same step captured in a hipGraph (one launch per step), and the 7 x 24 h rollout of configs[4]
        from pangu_pytorch_amd import rollout as R
        gs = R.GraphedStep(model, inp, inp_s, stats, maps, const_h)
        gs.step()
        lsync()
        tg = time.perf_counter()
        for _ in range(args.steps):
            gs.step()
        lsync()
        tg = time.perf_counter() - tg
        sl = (stats[0].view(1, 4, 1, 1), stats[1].view(1, 4, 1, 1),
              stats[2].reshape(13, 5).flip(0).t().reshape(1, 5, 13, 1, 1).contiguous(),
              stats[3].reshape(13, 5).flip(0).t().reshape(1, 5, 13, 1, 1).contiguous())
        gr = R.GraphedStep(model, inp, inp_s, stats, maps, const_h, stats_last=sl, feed_back=True)
        lsync()
        tr = time.perf_counter()
        for _ in range(7):
            gr.step()
        lsync()
        tr = time.perf_counter() - tr
        # per-step drift of the bf16 rollout against the SAME rollout in fp32 on this GPU (the
        # ONNX reference is unavailable)
        del gr
        _, _, hist_b = R.rollout(model, inp, inp_s, stats, maps, const_h, sl, steps=7, graph=True, keep=True)
        model.set_compute_dtype(torch.float32)
        _, _, hist_f = R.rollout(model, inp, inp_s, stats, maps, const_h, sl, steps=7, graph=False, keep=True)
        model.set_compute_dtype(torch.bfloat16)
        roll_drift = [((hb[0].double() - hf[0].double()).norm() / hf[0].double().norm()).item()
                      for hb, hf in zip(hist_b, hist_f)]
        del hist_b, hist_f
        times = [tg, tb, tr]
        # per-family roofline of the bf16 forward: one more EAGER pass with HIP-event pairs around every launch
        ops.timing_start()
        for _ in range(args.steps):
            step()
        lin_ms, lin_flop, lin_n, fam = ops.timing_stop("linear_bf16", also=("attn_qkv_bf16", "linear_ln_bf16", "mlp_fused_bf16"))
        bf16_roof = bf16_roofline({"gemm": (lin_ms, lin_flop, lin_n), "attn_qkv": fam["attn_qkv_bf16"], "gemm_ln": fam["linear_ln_bf16"],
                                   "mlp": fam["mlp_fused_bf16"]}, tb / args.steps * 1e3)
        bf16_res = {"metric": "bf16 forward steps/s (bf16 activations+weights, fp32 LN/softmax/accumulate), hipGraph replay",
                    "value": world * args.steps / tg, "ms_per_step": tg / args.steps * 1e3,
                    "eager_ms_per_step": tb / args.steps * 1e3, "rel_l2_drift_vs_f32": drift,
                    "model_tflops": FWD_GFLOP_EXEC / (tg / args.steps * 1e3),
                    "frac_of_bf16_mfma_peak": FWD_GFLOP_EXEC / (tg / args.steps * 1e3) / 2500.0,
                    "rollout_7x24h_ms": tr * 1e3, "rollout_rel_l2_drift_vs_f32_per_step": roll_drift, "roofline": bf16_roof}
        model.set_compute_dtype(torch.float32)
        del out_b, gs
      except Exception as e:      # secondary metrics must never take the headline line down
        bf16_res = {"error": repr(e)[:300]}
        model.set_compute_dtype(torch.float32)
      if dist is not None:        # every rank gets here, failed or not: slowest rank's times, any rank's failure
        ok = "error" not in bf16_res
        t = torch.tensor((times if ok else [0.0, 0.0, 0.0]) + [0.0 if ok else 1.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if ok and t[3].item() > 0:
            bf16_res = {"error": "another rank failed the bf16 section"}
        elif ok:
            tg, tb, tr = t[:3].tolist()
            bf16_res.update(value=world * args.steps / tg, ms_per_step=tg / args.steps * 1e3,
                            eager_ms_per_step=tb / args.steps * 1e3, model_tflops=FWD_GFLOP_EXEC / (tg / args.steps * 1e3),
                            frac_of_bf16_mfma_peak=FWD_GFLOP_EXEC / (tg / args.steps * 1e3) / PEAK_BF16_MFMA_TFLOPS,
                            rollout_7x24h_ms=tr * 1e3)
            if isinstance(bf16_res.get("roofline"), dict):
                bf16_res["roofline"]["times_of"] = "rank 0 (per-launch HIP events are local); the step times above: max over ranks"

    # ---- secondary metric: DDP finetune step (BASELINE configs[3] shape: 1 sample/GPU, fwd + bwd + bucketed RCCL
    # gradient all-reduce overlapped with backward + Adam), reported beside the headline number
    train_res = {}
    if not args.no_train:
        from pangu_pytorch_amd import train
        from pangu_pytorch_amd.dist import FlatGradSync
        del out
        model.train()
        tgt, tgt_s, *_ = synthetic_inputs(dev, seed=2000 + rank)      # synthetic targets of the input's shape
        opt = train.make_optimizer(model)      # Adam(lr=5e-6, weight_decay=3e-6), reference finetune_fully.py:121
        sync = FlatGradSync(model, force_collective=True, mode=args.grad_sync, timing=True) if dist is not None else None
        other_mode = "reduce_scatter" if args.grad_sync == "all_reduce" else "all_reduce"
        batch = (inp, inp_s, tgt, tgt_s)
        exposed = []      # (event before finish(), event after): GPU time the compute stream spends waiting for the buckets

        def grad_sync():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            sync.finish()
            b.record()
            exposed.append((a, b))
        from pangu_pytorch_amd.layers import DropPath
        dps = [(blk.attention.dim, blk.drop_path) for layer in model.layers for blk in layer.blocks if isinstance(blk.drop_path, DropPath)]

        def dropped_gflop(before):
            """Forward GFLOP the DropPath-dropped branches of the steps since `before` did NOT execute (per-branch counters)."""
            return sum((d.n_dropped_branch[0] - b[0]) * BRANCH_GFLOP[C][0] + (d.n_dropped_branch[1] - b[1]) * BRANCH_GFLOP[C][1]
                       for (C, d), b in zip(dps, before))
        for tag, dt in (("ddp_train", torch.float32), ("ddp_train_bf16", torch.bfloat16)):
            try:
                model.set_compute_dtype(dt)
                torch.manual_seed(1234 + rank)            # DropPath draws (host RNG): tools/profile_train.py seeds the same way
                torch.cuda.reset_peak_memory_stats()
                for _ in range(2):                        # untimed: builds the weight shadows / Adam state, then one steady-state step
                    train.train_step(model, opt, batch, stats, maps, const_h, grad_sync=grad_sync if sync else None)
                lsync()
                del exposed[:]
                if sync:
                    sync.bucket_times_ms()                # (drops the warm-up steps' event pairs)
                copied0 = sync.copied_bytes if sync else 0
                drops0 = [tuple(d.n_dropped_branch) for _, d in dps]
                t1 = time.perf_counter()
                for _ in range(args.train_steps):
                    loss = train.train_step(model, opt, batch, stats, maps, const_h, grad_sync=grad_sync if sync else None)
                lsync()
                t_train = time.perf_counter() - t1
                bucket_ms = sync.bucket_times_ms() if sync else None
                exposed_main = sum(a.elapsed_time(b) for a, b in exposed) / max(len(exposed), 1) if sync else None
                copied_per_step = (sync.copied_bytes - copied0) / args.train_steps if sync else None
                skipped = dropped_gflop(drops0) / args.train_steps            # forward GFLOP per step that was NOT executed
                n_drop = sum(d.n_dropped_branch[0] + d.n_dropped_branch[1] - b[0] - b[1] for (_, d), b in zip(dps, drops0))
                exec_gflop = 3.0 * (FWD_GFLOP_EXEC - skipped)                 # fwd + bwd = 3 x forward
                what = ("fwd+bwd+bucketed grad all-reduce (RCCL, issued from the backward hooks)+Adam" if sync else
                        "fwd+bwd+Adam; ONE rank: no collective and no flat gradient buffer exist (see ddp_model for the all-reduce)")
                train_res[tag] = {
                    "metric": f"finetune samples/s ({what}; 1 sample/GPU, DropPath on), "
                              + ("fp32" if dt == torch.float32 else "bf16 compute / fp32 master weights+grads"),
                    "value": world * args.train_steps / t_train, "ms_per_step": t_train / args.train_steps * 1e3,
                    "steps": args.train_steps, "loss": float(loss),
                    "peak_mem_gb": torch.cuda.max_memory_allocated() / 2**30,
                    "model_tflops": exec_gflop / (t_train / args.train_steps * 1e3),
                    "dropped_branches_in_timed_steps": n_drop}
                # the same step with stochastic depth OFF (every branch computed: the full 3 x forward ledger), timed beside it
                saved_p = [d.drop_prob for _, d in dps]
                try:
                    for _, d in dps:
                        d.drop_prob = 0.0
                    train.train_step(model, opt, batch, stats, maps, const_h, grad_sync=grad_sync if sync else None)
                    lsync()
                    t2 = time.perf_counter()
                    n_off = max(2, args.train_steps // 2)
                    for _ in range(n_off):
                        train.train_step(model, opt, batch, stats, maps, const_h, grad_sync=grad_sync if sync else None)
                    lsync()
                    t_off = (time.perf_counter() - t2) / n_off
                finally:      # a failure in here must not leave stochastic depth off for the next dtype's "DropPath on" numbers
                    for (_, d), pr in zip(dps, saved_p):
                        d.drop_prob = pr
                # roofline block of the training step: EXECUTED work of the timed steps; bytes per step from the committed PMC table
                pm, pm_stale, pm_commit = pmc_train("bf16" if dt == torch.bfloat16 else "f32")
                t_s = t_train / args.train_steps
                flop = exec_gflop * 1e9
                full = 3 * FWD_GFLOP_EXEC * 1e9
                peak = PEAK_BF16_MFMA_TFLOPS if dt == torch.bfloat16 else PEAK_F32_MFMA_TFLOPS
                train_res[tag]["roofline"] = {
                    "executed_flop_per_step": flop, "mfma_peak": peak, "mfma_frac": flop / t_s / 1e12 / peak,
                    "droppath_skipped_flop_per_step": 3 * skipped * 1e9,
                    "droppath_off": {"ms_per_step": t_off * 1e3, "flop_per_step": full, "mfma_frac": full / t_off / 1e12 / peak,
                                     "steps": n_off},
                    "hbm_bytes_per_step": pm["hbm_bytes_per_step"] if pm else None,
                    "hbm_frac_of_8TBps": pm["hbm_bytes_per_step"] / t_s / 1e9 / PEAK_HBM_GBS if pm else None,
                    "hbm_frac_of_achievable_6.29TBps": pm["hbm_bytes_per_step"] / t_s / 1e9 / ACHIEVABLE_HBM_GBS if pm else None,
                    "bound": ("hbm" if pm and pm["hbm_bytes_per_step"] / PEAK_HBM_GBS / 1e9 > flop / peak / 1e12 else "mfma"),
                    "traffic_stale": pm_stale, "traffic_commit": pm_commit,
                    "traffic_unit": "HBM bytes per step = sum over all kernels of one step (FETCH_SIZE x2 + WRITE_SIZE; rocprofv3 --pmc "
                                    "passes of tools/profile_train.py, same DropPath seed as this loop -> profiles/pmc_train.json)"}
                if sync is None and not args.no_rehearse:
                    # the same step with a flat gradient buffer and, per bucket, a traffic generator on a side stream where the
                    # RCCL all-reduce would run: how much the overlapped collective slows the backward kernels (an upper bound)
                    reh = {}
                    for name, kw in (("flat_buffer_no_traffic", None), ("traffic_32wg_x2", {"workgroups": 32, "passes": 2}),
                                     ("traffic_64wg_x2", {"workgroups": 64, "passes": 2})):
                        fs = FlatGradSync(model, rehearse=kw)
                        evs = []

                        def fin(fs=fs, evs=evs):      # GPU time the compute stream waits for the side stream at finish()
                            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            a.record()
                            fs.finish()
                            b.record()
                            evs.append((a, b))
                        torch.manual_seed(1234 + rank)
                        for _ in range(2):
                            train.train_step(model, opt, batch, stats, maps, const_h, grad_sync=fin)
                        lsync()
                        del evs[:]
                        t3 = time.perf_counter()
                        for _ in range(args.train_steps):
                            train.train_step(model, opt, batch, stats, maps, const_h, grad_sync=fin)
                        lsync()
                        reh[name] = {"ms_per_step": (time.perf_counter() - t3) / args.train_steps * 1e3,
                                     "exposed_wait_ms_per_step": sum(a.elapsed_time(b) for a, b in evs) / max(len(evs), 1)}
                        fs.remove()
                        model.zero_grad(set_to_none=True)
                        del fs
                    reh["note"] = ("DropPath on, same seed; exposed_wait = what the optimizer waits for the last buckets' traffic (the tail a "
                                   "real collective would expose too); ms_per_step - exposed_wait - the no-traffic step = what the overlapped "
                                   "traffic costs the backward kernels; traffic = per bucket 2 x (read + write) of the bucket by a copy kernel "
                                   "confined to N workgroups on a side stream, released when the bucket's last gradient exists; the "
                                   "optimizer waits for it")
                    train_res[tag]["allreduce_rehearsal_ms_per_step"] = reh
                if sync:
                    train_res[tag]["exposed_allreduce_ms_per_step"] = exposed_main
                    train_res[tag]["grad_copy_fallback_mib"] = copied_per_step / 2**20          # per step, over the timed steps
                    train_res[tag]["grad_sync_mode"] = args.grad_sync
                    # per bucket (0 = the first one backward completes: the output layer; 19 = the input layer): launch -> its last
                    # collective done, mean over the timed steps -- events on a side stream that only waits for that collective
                    train_res[tag]["bucket_launch_to_done_ms"] = bucket_ms
                    train_res[tag]["bucket_mib"] = [b / 2**20 for b in sync.bucket_bytes()]
                    # the SAME step with the other collective, in this very invocation: one command yields the A/B
                    sync.remove()
                    model.zero_grad(set_to_none=True)
                    sync = FlatGradSync(model, force_collective=True, mode=other_mode, timing=True)
                    torch.manual_seed(1234 + rank)
                    for _ in range(2):
                        train.train_step(model, opt, batch, stats, maps, const_h, grad_sync=grad_sync)
                    lsync()
                    del exposed[:]
                    sync.bucket_times_ms()
                    t3 = time.perf_counter()
                    for _ in range(args.train_steps):
                        train.train_step(model, opt, batch, stats, maps, const_h, grad_sync=grad_sync)
                    lsync()
                    t_other = (time.perf_counter() - t3) / args.train_steps * 1e3
                    train_res[tag]["grad_sync_ab"] = {
                        args.grad_sync: {"ms_per_step": t_train / args.train_steps * 1e3, "exposed_ms_per_step": exposed_main,
                                         "bucket_launch_to_done_ms": bucket_ms},
                        other_mode: {"ms_per_step": t_other,
                                     "exposed_ms_per_step": sum(a.elapsed_time(b) for a, b in exposed) / max(len(exposed), 1),
                                     "bucket_launch_to_done_ms": sync.bucket_times_ms()},
                        "note": "same DropPath seed, same batch, this rank's times; all_reduce = one all_reduce(AVG) per bucket, "
                                "reduce_scatter = reduce_scatter(AVG) + all_gather per bucket in place"}
                    sync.remove()
                    model.zero_grad(set_to_none=True)
                    sync = FlatGradSync(model, force_collective=True, mode=args.grad_sync, timing=True)
                    del exposed[:]
                if not args.no_fed:
                    torch.manual_seed(1234 + rank)
                    train_res[tag]["train_fed_from_host"] = fed_from_host(model, opt, dev, (stats, maps, const_h),
                                                                          grad_sync if sync else None, args.train_steps, rank)
            except Exception as e:      # secondary metrics must never take the headline line down
                train_res[tag] = {"error": repr(e)[:300]}
            if dist is not None:        # collectives OUTSIDE the try: every rank gets here, failed or not
                ok = "error" not in train_res[tag]
                fed_keys = ("resident_item_per_step_ms", "fed_free_running_ms", "fed_item_per_step_ms", "reference_to_device_item_per_step_ms")
                fed = train_res[tag].get("train_fed_from_host") if ok else None
                t = torch.tensor([train_res[tag]["ms_per_step"] if ok else 0.0, 0.0 if ok else 1.0] +
                                 [fed[k] if fed else 0.0 for k in fed_keys], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                if ok and t[1].item() > 0:
                    train_res[tag] = {"error": "another rank failed this section"}
                    break
                if ok:
                    # slowest rank's step time; the FLOP-derived numbers follow it (executed FLOPs stay this rank's DropPath draws:
                    # `flop_counts_of: rank 0`)
                    ms_max = t[0].item()
                    r_ = train_res[tag]
                    scale = r_["ms_per_step"] / ms_max
                    r_.update(ms_per_step=ms_max, value=world / (ms_max * 1e-3), model_tflops=r_["model_tflops"] * scale)
                    rf = r_.get("roofline")
                    if rf:
                        rf["mfma_frac"] *= scale
                        for k in ("hbm_frac_of_8TBps", "hbm_frac_of_achievable_6.29TBps"):
                            if rf.get(k) is not None:
                                rf[k] *= scale
                        rf["flop_counts_of"] = "rank 0 (its DropPath draws); times: max over ranks"
                    if fed:
                        fed.update({k: t[2 + i].item() for i, k in enumerate(fed_keys)})
                        fed["fed_over_resident"] = fed["fed_item_per_step_ms"] / fed["resident_item_per_step_ms"]
                        fed["times"] = "max over ranks; pipeline rates: rank 0"
                else:
                    break
        model.set_compute_dtype(torch.float32)
        if sync is not None:
            sync.remove()
        del opt, sync

    # ---- the reference's OWN call conventions, timed (no collectives in here): what a maintainer gets from dropping the model in
    # without touching the scripts
    extras = None
    if not args.no_extras:
        extras = {}
        try:
            from pangu_pytorch_amd import train
            n_x = max(3, args.steps // 4)
            # (a) test() of models/pangu_sample.py:197-202: model.eval() and NO torch.no_grad() -> the autograd (activation-saving)
            # forward runs; the graph is dropped with the outputs
            model.eval()
            fe = {}
            for sfx, mode in (("", "recompute"), ("_saving_activations", "save")):
                model.eval_grad_mode = mode
                for key, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
                    model.set_compute_dtype(dt)
                    for _ in range(2):
                        o = model(inp, inp_s, stats, maps, const_h)
                        del o
                    lsync()
                    t1 = time.perf_counter()
                    for _ in range(n_x):
                        o = model(inp, inp_s, stats, maps, const_h)
                        del o
                    lsync()
                    fe[key + sfx] = (time.perf_counter() - t1) / n_x * 1e3
            model.eval_grad_mode = "recompute"
            fe["note"] = ("model.eval() with grads ENABLED, the call of reference models/pangu_sample.py:197-202 (no no_grad).  f32 / bf16: the "
                          "default (PanguModel.eval_grad_mode = 'recompute': inference kernels now, the autograd forward re-run only if a backward "
                          "arrives); *_saving_activations: eval_grad_mode = 'save' (the activation-saving training forward at once)")
            extras["forward_grad_enabled_eval_ms"] = fe
            # (b) the loop body of models/pangu_sample.py:45-77 as written there: optimizer.zero_grad(); model.train(); forward;
            # torch-op weighted L1; loss.backward(); optimizer.step() with torch.optim.Adam's defaults (finetune_fully.py:121)
            tgt, tgt_s, *_ = synthetic_inputs(dev, seed=2000 + rank)
            rl = {"note": "reference loop body unchanged: torch-op L1 loss + torch.optim.Adam(lr=5e-6, weight_decay=3e-6) (default "
                          "foreach implementation), DropPath on, dropped branches handed ZERO gradients (ops default 'zeros')"}
            for key, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
                model.set_compute_dtype(dt)
                opt_ref = torch.optim.Adam(model.parameters(), lr=5e-6, weight_decay=3e-6)
                torch.manual_seed(4321 + rank)

                def ref_step():
                    opt_ref.zero_grad()
                    model.train()
                    o, o_s = model(inp, inp_s, stats, maps, const_h)
                    loss = train._weighted_l1_loss_torch(o, o_s, tgt, tgt_s)
                    loss.backward()
                    opt_ref.step()
                    return loss
                for _ in range(2):
                    ref_step()
                lsync()
                t1 = time.perf_counter()
                for _ in range(n_x):
                    loss = ref_step()
                lsync()
                rl[key] = (time.perf_counter() - t1) / n_x * 1e3
                rl[key + "_loss"] = float(loss.detach())
                del opt_ref, loss
                model.zero_grad(set_to_none=True)
            extras["reference_loop_step_ms"] = rl
        except Exception as e:      # secondary metrics must never take the headline line down
            extras["error"] = repr(e)[:300]
        model.set_compute_dtype(torch.float32)
        model.eval()

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        ms_i = elapsed_instr / args.steps * 1e3          # step time of the instrumented pass (this rank)
        achieved = gemm_flop / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        # HBM bytes per launch of the same kernel family from the committed rocprofv3 PMC run (separate --pmc passes,
        # FETCH_SIZE doubled per MI355X_MICROARCH.md): counters cannot be read from inside this process
        traffic, mfma_busy, stale, pmc_commit = pmc_traffic("f32", "gemm")
        a_traffic, a_busy, a_stale, _ = pmc_traffic("f32", "attn")
        a_achieved = attn_flop / (attn_ms * 1e-3) / 1e12 if attn_ms > 0 else 0.0
        res = {
            "metric": "forward steps/sec (721x1440x13pl) per MI355X", "value": world * args.steps / elapsed,
            "unit": "forward steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "PanguModel fp32 forward, 1 sample/GPU, input (1,5,13,721,1440)+(1,4,721,1440), "
                                   "depths 2-6-6-2, dims 192/384, random-init weights (BASELINE configs[1])",
                       "parallelism": f"dp{world}"},
            "model_tflops": FWD_GFLOP_EXEC / ms,
            "model_flop_note": f"executed dense-contraction GFLOP per step {FWD_GFLOP_EXEC} (the reference's padded-shape ledger: {FWD_GFLOP})",
            "roofline": {"bound": "mfma", "kernel": "gemm_tn_f32_dma_kernel (plain projection GEMMs)", "achieved": achieved,
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS,
                         "traffic": traffic, "traffic_stale": stale, "traffic_commit": pmc_commit,
                         "traffic_unit": "HBM bytes/launch (rocprofv3 PMC passes of tools/pmc_traffic.sh -> profiles/pmc_traffic_f32.json; "
                                         "null + traffic_stale when the kernel source changed since)",
                         "algorithmic_flop_per_launch": gemm_flop / max(gemm_launches, 1),
                         "mfma_busy_frac_pmc": mfma_busy, "launches": gemm_launches, "avg_launch_ms": gemm_ms / max(gemm_launches, 1),
                         "share_of_step": gemm_ms / (ms_i * args.steps),
                         "timing_pass": {"note": "per-kernel durations come from a SECOND pass over the same K steps with HIP-event pairs around "
                                                 "each launch; the headline value / ms_per_step come from the un-instrumented loop",
                                         "instrumented_ms_per_step": ms_i, "event_overhead_ms_per_step": ms_i - ms,
                                         "event_pairs_per_step": (gemm_launches + fused_launches + attn_launches) / args.steps},
                         "fused_ln_gemm": {"kernel": "gemm_ln_residual_f32_dma_kernel (projection + LayerNorm + residual)",
                                           "launches": fused_launches, "avg_launch_ms": fused_ms / max(fused_launches, 1),
                                           "achieved": fused_flop / (fused_ms * 1e-3) / 1e12 if fused_ms > 0 else 0.0,
                                           "share_of_step": fused_ms / (ms_i * args.steps)},
                         "attention": {"kernel": "window_attn_f32_kernel (QK^T + Earth bias + shift mask + softmax + PV, one launch per block)",
                                       "bound": "mfma", "achieved": a_achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                       "frac": a_achieved / PEAK_F32_MFMA_TFLOPS, "launches": attn_launches,
                                       "avg_launch_ms": attn_ms / max(attn_launches, 1),
                                       "algorithmic_flop_per_launch": attn_flop / max(attn_launches, 1),
                                       "traffic": a_traffic, "traffic_stale": a_stale, "mfma_busy_frac_pmc": a_busy,
                                       "share_of_step": attn_ms / (ms_i * args.steps)}},
        }
        if bf16_res is not None:
            res["bf16_forward"] = bf16_res
        res.update(train_res)
        if extras:
            res.update(extras)
        # the metric's second half ("DDP samples/sec at 1/2/4/8") as TOP-LEVEL keys: measured in this very run by all `world` ranks
        # (fwd + bwd + the bucketed gradient all-reduce issued from the backward hooks + Adam; max-over-ranks step time)
        dd = {"unit": "samples/s", "measured": True, "ranks": world, "resident_batch": True,
              "resident_batch_note": "input and target of every step are already in HBM; the step fed from pageable host memory "
                                     "(573 MB per step) is ddp_train*.train_fed_from_host",
              "rccl_ranks": world if (dist is not None and backend == "nccl") else 0,
              "backend": (backend if dist is not None else None),
              "collective": ((("bucketed all_reduce(AVG)" if args.grad_sync == "all_reduce" else
                               "bucketed reduce_scatter(AVG) + all_gather, in place,") +
                              " of the flat fp32 gradient buffer, overlapped with backward") if dist is not None
                             else "none (one rank: no process group)")}
        if dist is not None:
            try:
                v = torch.cuda.nccl.version()
                dd["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
            except Exception as e:
                dd["rccl_version"] = repr(e)[:120]
            dd["xgmi_topology"] = xgmi_topology()
            dd["hip_devices_visible"] = n_dev
        for tag, key in (("ddp_train", "fp32"), ("ddp_train_bf16", "bf16")):
            tr_ = train_res.get(tag)
            if tr_ and "error" not in tr_:
                fed_ = tr_.get("train_fed_from_host") or {}
                dd[key] = {"value": tr_["value"], "ms_per_step": tr_["ms_per_step"],
                           "fed_from_host_value": (world / (fed_["fed_item_per_step_ms"] * 1e-3)) if "fed_item_per_step_ms" in fed_ else None,
                           "fed_from_host_ms_per_step": fed_.get("fed_item_per_step_ms"),
                           "exposed_allreduce_ms_per_step": tr_.get("exposed_allreduce_ms_per_step"),
                           "grad_copy_fallback_mib": tr_.get("grad_copy_fallback_mib"),
                           "bucket_launch_to_done_ms": tr_.get("bucket_launch_to_done_ms"),
                           "grad_sync_ab_ms_per_step": ({k: v["ms_per_step"] for k, v in tr_["grad_sync_ab"].items() if isinstance(v, dict)}
                                                        if "grad_sync_ab" in tr_ else None)}
            elif tr_:
                dd[key] = {"error": tr_["error"]}
        res["ddp_samples_per_s"] = dd
        # ---- data-parallel model: measured 1-rank step + MODELLED gradient all-reduce
        n_grad = sum(p.numel() for p in model.parameters())
        gbytes = n_grad * 4.0
        mdl = {"measured": False, "note": "MODELLED, not measured (one GPU per gpurun box): per step ONE averaged all-reduce of the flat fp32 gradient "
                       "buffer (reference era5_data/utils_dist.py:125-134 semantics), bucketed in reverse block order so it runs under "
                       "the remaining backward; xGMI is point to point, 7 links x 153 GB/s per direction per GPU",
               "grad_bytes": gbytes, "link_GBps": XGMI_LINK_GBS, "allreduce_ms": {}}
        for n in (2, 4, 8):
            ring = 2.0 * (n - 1) / n * gbytes / (XGMI_LINK_GBS * 1e9) * 1e3          # one link busy per GPU
            direct = 2.0 * gbytes / n / (XGMI_LINK_GBS * 1e9) * 1e3                    # reduce-scatter + all-gather, every peer link busy
            mdl["allreduce_ms"][str(n)] = {"measured": False, "ring_one_link": ring, "direct_all_links": direct}
        for tag in ("ddp_train", "ddp_train_bf16"):
            if tag in train_res and "error" not in train_res[tag]:
                t_step = train_res[tag]["ms_per_step"] if world == 1 else None
                if t_step:
                    d8 = mdl["allreduce_ms"]["8"]
                    mdl[tag] = {"measured_1gpu_ms_per_step": t_step,
                                "projected_8gpu_samples_per_s": {"measured": False, "allreduce_fully_overlapped": 8e3 / t_step,
                                                                 "direct_fully_exposed": 8e3 / (t_step + d8["direct_all_links"]),
                                                                 "ring_fully_exposed": 8e3 / (t_step + d8["ring_one_link"])}}
                if "exposed_allreduce_ms_per_step" in train_res[tag]:
                    mdl.setdefault(tag, {})["measured_exposed_allreduce_ms_this_run"] = train_res[tag]["exposed_allreduce_ms_per_step"]
        res["ddp_model"] = mdl
        mode = "none" if args.no_cpu_baseline else args.cpu_baseline
        if mode == "auto":
            mode = "full" if (os.cpu_count() or 1) >= 32 else "sample"
        if world == 1 and mode != "none":
            cb = cpu_baseline()
            if mode == "full":
                t_full = cpu_baseline_full(cb["cores"])
                cb["extrapolated_from_block_pairs"] = {"value": cb["value"], "sample": cb["sample"]}
                cb["value"] = 1.0 / t_full
                cb["sample"] = (f"the oracle's WHOLE forward (oracle/pangu_oracle.forward = reference pangu_model.py:50-87 restated), one "
                                f"step on the bench's shapes: {t_full:.1f} s with {cb['cores']} threads")
            res["cpu_baseline"] = cb
            try:
                rv = rollout_vs_reference(dev)
                if rv is not None:
                    res["rollout_7x24h_vs_reference"] = rv
            except Exception as e:      # a checker leg must never take the headline line down
                res["rollout_7x24h_vs_reference"] = {"error": repr(e)[:300]}
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
