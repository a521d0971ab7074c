"""Multi-process (gloo, world_size 2, CPU) tests of the data-parallel gradient path: flat-buffer bucketed all-reduce
== the reference's intended gather_grad (era5_data/utils_dist.py:125-134: SUM then / world) == single-process mean of
per-sample gradients (the parity definition for the multi-GPU path)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _NoneGrad(torch.autograd.Function):
    """Like the HIP block function with a DropPath-dropped branch: the weight takes part in the graph but gets None."""

    @staticmethod
    def forward(ctx, x, w):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return g, None


class _Tiny(torch.nn.Module):
    """Same naming structure as PanguModel (so default_buckets groups per block), tiny sizes."""

    def __init__(self):
        super().__init__()
        mk = lambda: torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.LayerNorm(8))
        self._input_layer = torch.nn.Linear(6, 8)
        self.layers = torch.nn.ModuleDict({
            "EarthSpecificLayer0": torch.nn.ModuleDict({"blocks": torch.nn.ModuleDict(
                {"EarthSpecificBlock0": mk(), "EarthSpecificBlock1": mk()})}),
            "EarthSpecificLayer1": torch.nn.ModuleDict({"blocks": torch.nn.ModuleDict({"EarthSpecificBlock0": mk()})}),
        })
        self._output_layer = torch.nn.Linear(8, 3)

    def forward(self, x, drop=None):
        x = self._input_layer(x)
        for li, layer in self.layers.items():
            for bi, blk in layer["blocks"].items():
                if drop != (li, bi):
                    x = x + blk(x)
                else:
                    x = _NoneGrad.apply(x, blk[0].weight)
        return self._output_layer(x)


def _worker(rank, world, port, drop_on_rank1, q, mode="all_reduce"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import pangu_pytorch_amd as P
    from pangu_pytorch_amd import dist as D
    D.init_dist("pytorch", backend="gloo")
    assert D.get_dist_info() == (rank, world)
    torch.manual_seed(0)
    model = _Tiny()
    sync = D.FlatGradSync(model, mode=mode)
    assert len(sync.buckets) == 5           # output, L1.B0, L0.B1, L0.B0, input  (reverse execution order)
    xs = torch.randn(world, 5, 6, generator=torch.Generator().manual_seed(1))
    for step in range(2):                   # second step exercises grads-as-views accumulation
        sync.zero_grad()
        drop = ("EarthSpecificLayer0", "EarthSpecificBlock1") if (drop_on_rank1 and rank == 1) else None
        model(xs[rank], drop).pow(2).sum().backward()
        sync.finish()
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    assert all(p.grad.data_ptr() == sync._slot[p][1].data_ptr() for p in model.parameters())
    # slow baseline with the reference's per-parameter semantics
    m2 = _Tiny()
    m2.load_state_dict(model.state_dict())
    drop = ("EarthSpecificLayer0", "EarthSpecificBlock1") if (drop_on_rank1 and rank == 1) else None
    m2(xs[rank], drop).pow(2).sum().backward()
    for p in m2.parameters():
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    D.gather_grad(m2.parameters(), world)
    for (k, p) in m2.named_parameters():
        assert torch.allclose(grads[k], p.grad, rtol=1e-6, atol=1e-7), k
    if rank == 0:
        q.put({k: v.numpy() for k, v in grads.items()})      # plain bytes: no fd hand-off racing with process exit
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("mode", ["all_reduce", "reduce_scatter"])
@pytest.mark.parametrize("drop", [False, True])
def test_flat_grad_sync_matches_single_process_mean(drop, mode):
    """Both collectives of FlatGradSync (bucketed all_reduce; reduce_scatter + all_gather in place)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, drop, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    got = {k: torch.from_numpy(v) for k, v in q.get(timeout=120).items()}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-process reference: mean over the per-sample gradients
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pangu_oracle as O
    torch.manual_seed(0)
    model = _Tiny()
    xs = torch.randn(world, 5, 6, generator=torch.Generator().manual_seed(1))
    per_rank = []
    for r in range(world):
        model.zero_grad()
        d = ("EarthSpecificLayer0", "EarthSpecificBlock1") if (drop and r == 1) else None
        model(xs[r], d).pow(2).sum().backward()
        per_rank.append({k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p))
                         for k, p in model.named_parameters()})
    want = O.gather_grad_mean(per_rank)
    for k in want:
        assert torch.allclose(got[k], want[k], rtol=1e-5, atol=1e-6), k


def test_default_buckets_on_real_model():
    sys.path.insert(0, ROOT)
    import pangu_pytorch_amd as P
    from pangu_pytorch_amd import dist as D
    m = P.PanguModel(device="cpu")
    b = D.default_buckets(m)
    assert len(b) == 16 + 4                                   # 16 blocks + output, upsample, downsample, input
    assert sum(len(x) for x in b) == 223
    names = {id(p): n for n, p in m.named_parameters()}
    assert names[id(b[0][0])].startswith("_output_layer")
    assert names[id(b[1][0])].startswith("layers.EarthSpecificLayer3.blocks.EarthSpecificBlock1")
    assert names[id(b[2][0])].startswith("layers.EarthSpecificLayer3.blocks.EarthSpecificBlock0")
    assert names[id(b[3][0])].startswith("upsample")
    assert names[id(b[4][0])].startswith("layers.EarthSpecificLayer2.blocks.EarthSpecificBlock5")
    assert names[id(b[-2][0])].startswith("layers.EarthSpecificLayer0.blocks.EarthSpecificBlock0")
    assert names[id(b[-1][0])].startswith("_input_layer")
    # one ~62-64 MB bias table per block bucket
    for blk in b[1:3]:
        assert sum(p.numel() for p in blk) * 4 > 60e6


class _SlotWriter(torch.autograd.Function):
    """Like the HIP block function: its backward WRITES the parameter's gradient into the flat-buffer slot when
    ops.grad_slot hands one out (pangu-pytorch_amd/autograd.py: desb_out=ops.grad_slot(esb))."""

    @staticmethod
    def forward(ctx, x, w, k):
        ctx.k, ctx.w = k, w
        return x * 1.0

    @staticmethod
    def backward(ctx, g):
        from pangu_pytorch_amd import ops
        out = ops.grad_slot(ctx.w)
        gw = torch.full_like(ctx.w, float(ctx.k))
        if out is not None:
            out.copy_(gw)          # the kernel overwrites the slot
            gw = out
        return g, gw, None


def test_grad_slot_handed_out_once_per_backward():
    """one parameter feeding two slot-writing nodes in one backward (per-GPU batch B > 1, or a model applied
    twice) must accumulate g1 + g2, not 2 * g_last: the slot is handed out once until the parameter's gradient is accumulated."""
    from pangu_pytorch_amd import ops
    w = torch.nn.Parameter(torch.zeros(4))
    flat = torch.zeros(4)
    owner = object()
    ops.register_grad_slots({w: flat}, owner=owner)
    try:
        x = torch.ones(3, requires_grad=True)
        y = _SlotWriter.apply(x, w, 1.0) + _SlotWriter.apply(x, w, 10.0)
        hook = w.register_post_accumulate_grad_hook(lambda p: ops.release_grad_slot(p))
        y.sum().backward()
        assert torch.equal(w.grad, torch.full((4,), 11.0)), w.grad
        # released by the hook: the next backward (gradient cleared) may claim the slot again
        w.grad = None
        assert ops.grad_slot(w) is flat and ops.grad_slot(w) is None
        ops.release_grad_slot(w)
        # a parameter that already holds a gradient never gets the slot (autograd must add)
        w.grad = torch.ones(4)
        assert ops.grad_slot(w) is None
        hook.remove()
        # registries of two owners are independent
        w2, other = torch.nn.Parameter(torch.zeros(2)), object()
        ops.register_grad_slots({w2: torch.zeros(2)}, owner=other)
        ops.unregister_grad_slots(owner)
        w.grad = None
        assert ops.grad_slot(w) is None and ops.grad_slot(w2) is not None
        ops.unregister_grad_slots(other)
    finally:
        ops.unregister_grad_slots(owner)


def test_init_dist_slurm_launcher():
    """reference utils_dist.py:31-59: rank / world / master taken from the SLURM environment."""
    import subprocess
    env = dict(os.environ, SLURM_PROCID="0", SLURM_NTASKS="1", SLURM_NODELIST="localhost", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from pangu_pytorch_amd import dist as D\n"
            "D.init_dist('slurm', backend='gloo', port=%d)\n"
            "import os, torch.distributed as t\n"
            "assert D.get_dist_info() == (0, 1) and os.environ['RANK'] == '0' and os.environ['WORLD_SIZE'] == '1'\n"
            "t.destroy_process_group()\n"
            "try:\n    D.init_dist('mpi')\nexcept ValueError as e:\n    assert 'Invalid launcher' in str(e)\nelse:\n    raise SystemExit(3)\n"
            % (ROOT, _free_port()))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
