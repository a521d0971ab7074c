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
