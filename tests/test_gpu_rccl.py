"""The bench's multi-rank code path on real hardware: a 1-rank RCCL communicator (the GPU box has one GPU) driven by
torch.distributed.run exactly as the driver launches N>1 — process-group init with device_id, barriers, the MAX-over-ranks
timing reductions, the bucketed gradient all-reduce (AVG) issued from the backward hooks, hipGraph capture while the
RCCL watchdog thread is alive.  World-size-2 semantics are covered on CPU/gloo in test_dist_cpu.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_under_torchrun_one_rank_rccl():
    env = dict(os.environ, PANGU_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--train-steps", "1",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["value"] > 0
    assert "error" not in res["bf16_forward"], res["bf16_forward"]
    for tag in ("ddp_train", "ddp_train_bf16"):
        assert "error" not in res[tag], res[tag]
        assert res[tag]["loss"] == res[tag]["loss"]        # not NaN
        # what makes a first N > 1 run self-explaining, through REAL RCCL calls: per-bucket launch -> done
        # times from events on a side stream, and the other collective mode timed in the same invocation
        bt = res[tag]["bucket_launch_to_done_ms"]
        assert len(bt) == 20 and all(t is not None and t >= 0.0 for t in bt), bt
        ab = res[tag]["grad_sync_ab"]
        assert ab["all_reduce"]["ms_per_step"] > 0 and ab["reduce_scatter"]["ms_per_step"] > 0
        assert len(ab["reduce_scatter"]["bucket_launch_to_done_ms"]) == 20
        fed = res[tag]["train_fed_from_host"]
        assert fed["fed_item_per_step_ms"] > 0 and fed["pipeline"]["levels_reversed_fused"] is True
    dd = res["ddp_samples_per_s"]
    assert dd["resident_batch"] is True and isinstance(dd["rccl_version"], str) and dd["rccl_version"][0].isdigit(), dd.get("rccl_version")
    assert "xgmi_topology" in dd and ("error" in dd["xgmi_topology"] or dd["xgmi_topology"]["gpus"] >= 1)
    assert dd["bf16"]["fed_from_host_value"] > 0 and set(dd["bf16"]["grad_sync_ab_ms_per_step"]) == {"all_reduce", "reduce_scatter"}


def test_reduce_scatter_mode_on_a_one_rank_rccl_communicator():
    """`--grad-sync reduce_scatter` (FlatGradSync mode "reduce_scatter": per bucket an in-place reduce_scatter(AVG) into this rank's
    shard + an in-place all_gather, both issued asynchronously from the backward hooks) through REAL RCCL calls: on a 1-rank
    communicator the collectives are identities, so the training loss must equal the all_reduce mode's (to the run-to-run spread of
    the step's fp32 atomics: two optimizer steps, 1e-5) -- what the test buys is that RCCL accepts the in-place / aliased tensor arguments and the issue order (gloo, which runs the world-2 parity
    tests, takes a different code path)."""
    losses = {}
    for mode in ("all_reduce", "reduce_scatter"):
        env = dict(os.environ, PANGU_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
               os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--train-steps", "2",
               "--no-cpu-baseline", "--no-bf16", "--no-extras", "--no-fed", "--grad-sync", mode]
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        dd = res["ddp_samples_per_s"]
        assert dd["rccl_ranks"] == 1 and dd["backend"] == "nccl" and (("reduce_scatter" in dd["collective"]) == (mode == "reduce_scatter"))
        for tag in ("ddp_train", "ddp_train_bf16"):
            assert "error" not in res[tag], res[tag]
        losses[mode] = (res["ddp_train"]["loss"], res["ddp_train_bf16"]["loss"])
    for a, b in zip(losses["all_reduce"], losses["reduce_scatter"]):
        assert abs(a - b) <= 1e-5 * abs(a), losses
