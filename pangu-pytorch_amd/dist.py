"""Data-parallel plumbing: one process per GPU, RCCL over xGMI (torch.distributed backend "nccl" on ROCm).

Counterpart of the reference's era5_data/utils_dist.py (init_dist :13-28, get_dist_info :65-76, and the intended but
never-called gradient averaging `gather_grad` :125-134 = all_reduce(SUM) then / world_size per parameter).

MI355X design: `gather_grad`'s 223 per-parameter messages (~140 of them < 8 KB) would be launch-latency bound, and
the payload is dominated by sixteen 62-64 MB earth_specific_bias gradients.  Here gradients live in ONE flat
fp32 buffer (1.107 GB) cut into buckets along block boundaries in reverse execution order; each bucket is
all-reduced (average) asynchronously as soon as backward has produced its last gradient, so the collective runs on
RCCL's stream underneath the remaining backward kernels.  After `finish()` every `p.grad` is a view into the flat
buffer, which the optimizer then reads in place.
"""
import os

import torch
import torch.distributed as tdist


def init_dist(launcher="pytorch", backend="nccl", port=None, **kwargs):
    """reference utils_dist.py:13-59: one GPU per process; `launcher` = "pytorch" (rank from the torch.distributed.run
    environment, :24-28) or "slurm" (rank / world / master from SLURM_PROCID / SLURM_NTASKS / SLURM_NODELIST, :31-59)."""
    if launcher == "slurm":
        import subprocess
        proc_id, ntasks = int(os.environ["SLURM_PROCID"]), int(os.environ["SLURM_NTASKS"])
        if port is not None:
            os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("MASTER_PORT", "29500")           # torch.distributed's default port, as the reference
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = subprocess.getoutput(
                f"scontrol show hostname {os.environ['SLURM_NODELIST']} | head -n1")
        os.environ["WORLD_SIZE"], os.environ["RANK"] = str(ntasks), str(proc_id)
        os.environ["LOCAL_RANK"] = str(proc_id % max(torch.cuda.device_count(), 1))
    elif launcher != "pytorch":
        raise ValueError(f"Invalid launcher type: {launcher}")
    rank = int(os.environ["RANK"])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", rank % max(torch.cuda.device_count(), 1)))
        torch.cuda.set_device(local)
        kwargs.setdefault("device_id", torch.device("cuda", local))
    tdist.init_process_group(backend=backend, **kwargs)


def get_dist_info():
    """reference utils_dist.py:65-76 -> (rank, world_size)."""
    if tdist.is_available() and tdist.is_initialized():
        return tdist.get_rank(), tdist.get_world_size()
    return 0, 1


_EXEC_POS = {"_input_layer": 0, "downsample": 2, "upsample": 5, "_output_layer": 7}
_LAYER_POS = {0: 1, 1: 3, 2: 4, 3: 6}       # forward order: embed, L0, down, L1, L2, up, L3, recover


def default_buckets(model):
    """Parameter groups in the order backward finishes them (reverse forward order): output layer, layer-3 blocks
    (last first), upsample, layer 2, layer 1, downsample, layer 0, input layer.  One bucket per block: each holds one
    62-64 MB earth_specific_bias gradient plus ~1.8-7 MB of projection weights."""
    groups, pos = {}, {}
    for n, p in model.named_parameters():
        parts = n.split(".")
        if parts[0] == "layers" and len(parts) > 4:
            k = ".".join(parts[:4])             # layers.EarthSpecificLayerI.blocks.EarthSpecificBlockJ
            li = int("".join(c for c in parts[1] if c.isdigit()) or 0)
            bj = int("".join(c for c in parts[3] if c.isdigit()) or 0)
            where = (_LAYER_POS.get(li, 10 + li), bj)
        else:
            k = parts[0]
            where = (_EXEC_POS.get(k, 100 + len(groups)), 0)
        if k not in groups:
            groups[k], pos[k] = [], where
        groups[k].append(p)
    return [groups[k] for k in sorted(groups, key=lambda k: pos[k], reverse=True)]


class FlatGradSync:
    """Bucketed, backward-overlapped gradient averaging over a flat buffer.

    usage:   sync = FlatGradSync(model);   loss.backward();   sync.finish();   optimizer.step()
    """

    def __init__(self, model, process_group=None, buckets=None, average=True, force_collective=False, mode="all_reduce",
                 rehearse=None, timing=False):
        """mode: "all_reduce" (default: one all_reduce(AVG) per bucket) or "reduce_scatter" (per bucket a reduce_scatter(AVG) into
        this rank's 1/world shard followed by an all_gather of the shards, both in place in the flat buffer: on the fully connected
        xGMI mesh every peer link carries 1/world of the bucket in each phase -- SURVEY section 5 -- instead of a ring's whole
        bucket over one link; same result, selectable so the first multi-GPU run can A/B the two)."""
        if mode not in ("all_reduce", "reduce_scatter"):
            raise ValueError("FlatGradSync mode: 'all_reduce' or 'reduce_scatter'")
        self.mode = mode
        # rehearse = {"workgroups": n, "passes": k} (ONE rank only; measurement aid, bench.py --rehearse-collective): every bucket
        # launch also starts k device-to-device copies of the bucket on a side stream, each confined to n workgroups -- the HBM
        # traffic and CU footprint a real RCCL all-reduce of that bucket would put next to the remaining backward kernels
        self.rehearse = dict(rehearse) if rehearse else None
        # timing: per bucket, an event pair (gradients complete on the compute stream = launch; the bucket's last collective done,
        # recorded on a side stream that only ever waits for that collective) -- `bucket_times_ms()`; what an N > 1 bench line needs
        # to explain its number (which buckets queue behind which, what the tail after the last backward kernel is)
        self.timing = bool(timing)
        self._tstream = None
        self._tev = []             # (bucket index, launch event, done event) of the steps since the last bucket_times_ms()
        self._side = None
        self.group = process_group
        self.world = tdist.get_world_size(process_group) if tdist.is_initialized() else 1
        self.rank = tdist.get_rank(process_group) if tdist.is_initialized() else 0
        self.average = average
        self.force = force_collective      # issue the collectives on a 1-rank group too (exercises the RCCL path)
        params = [p for p in model.parameters() if p.requires_grad]
        buckets = buckets if buckets is not None else default_buckets(model)
        buckets = [[p for p in b if p.requires_grad] for b in buckets]
        buckets = [b for b in buckets if b]
        assert sum(len(b) for b in buckets) == len(params), "buckets must cover every trainable parameter once"
        dev, dtype = params[0].device, params[0].dtype
        # every bucket's length is a multiple of (16 bytes / element size) * world elements (zero padding at its end): 16-B aligned
        # bucket starts and, in "reduce_scatter" mode, equal 16-B aligned shards per rank -- for 2-byte parameters too
        quantum = (16 // max(1, min(16, params[0].element_size()))) * max(self.world, 1)
        padded = lambda n: (n + quantum - 1) // quantum * quantum
        total = sum(padded(sum(p.numel() for p in b)) for b in buckets)
        self.flat = torch.zeros(total, dtype=dtype, device=dev)
        self.buckets = []          # (start, end incl. padding, [(param, view)])
        off = 0
        self._slot = {}
        for bi, b in enumerate(buckets):
            start = off
            views = []
            for p in b:
                v = self.flat[off:off + p.numel()].view_as(p)
                views.append((p, v))
                self._slot[p] = (bi, v)
                off += p.numel()
            off = start + padded(off - start)
            self.buckets.append((start, off, views))
        self._pending = [len(b[2]) for b in self.buckets]
        self._fired = set()
        self._next = 0             # buckets are launched strictly in order: identical collective order on every rank
        self._works = []
        self._gloo = tdist.is_initialized() and tdist.get_backend(process_group) == "gloo"
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params]
        # the kernels that WRITE a whole parameter gradient (the Earth-specific bias tables: 1.04 of the 1.107 GB) store it
        # straight into its flat slot (ops.grad_slot); what is left for _on_grad to copy are the small accumulated tensors
        self._cuda = dev.type == "cuda"
        if self._cuda:
            from . import ops
            self._ops = ops
            # (5-D: the bias tables the attention backward writes whole; the other matrices' slots are there for the explicit
            # zeros of a DropPath-dropped branch, ops.fill_dropped_grads: zeroed in place instead of a zeros tensor + a copy)
            ops.register_grad_slots({p: v for p, (_, v) in self._slot.items() if p.dim() >= 2}, owner=self)
        self.copied_bytes = 0      # bytes moved by the copy fallback since construction (diagnostic)

    # -- per-parameter hook: move the fresh gradient into its flat slot; launch every bucket that became complete
    def _on_grad(self, p):
        bi, view = self._slot[p]
        if self._cuda:
            self._ops.release_grad_slot(p)      # every node's gradient has arrived: the slot may be claimed again
        if p.grad is None:              # a custom Function returned None (DropPath-dropped branch): contributes zeros
            view.zero_()
            p.grad = view
        elif p.grad.data_ptr() != view.data_ptr():
            view.copy_(p.grad)
            self.copied_bytes += p.grad.numel() * p.grad.element_size()
            p.grad = view
        self._fired.add(p)
        self._pending[bi] -= 1
        while self._next < len(self.buckets) and self._pending[self._next] == 0:
            self._launch(self._next)
            self._next += 1

    def _launch(self, bi):
        if self.rehearse is not None and self.world == 1 and self._cuda:
            from . import _lib
            start, end, _ = self.buckets[bi]
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.flat.device)
                self._scratch = torch.empty_like(self.flat)
            ev = torch.cuda.Event()
            ev.record()                                     # the bucket's gradients are complete on the compute stream
            self._side.wait_event(ev)
            nbytes = (end - start) * self.flat.element_size()
            for _ in range(int(self.rehearse.get("passes", 2))):
                _lib.check(_lib.load().pangu_traffic_copy(self._side.cuda_stream, self.flat[start:end].data_ptr(),
                                                          self._scratch[start:end].data_ptr(), nbytes,
                                                          int(self.rehearse.get("workgroups", 32))), "traffic_copy")
        if self.world == 1 and not self.force:
            return
        n_before = len(self._works)
        e0 = None
        if self.timing and self._cuda:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        self._launch_collective(bi)
        if e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            if self._gloo or len(self._works) == n_before:
                # gloo (tests): works complete on host threads -- the done event is recorded by finish() after the host wait (or
                # right here when the phases ran synchronously)
                if len(self._works) == n_before:
                    e1.record()
                    self._tev.append((bi, e0, e1))
                else:
                    self._works[-1] = self._works[-1] + ((bi, e0, e1),)
            else:
                if self._tstream is None:
                    self._tstream = torch.cuda.Stream(device=self.flat.device)
                with torch.cuda.stream(self._tstream):
                    self._works[-1][0].wait()          # stream-side wait: the side stream blocks until this bucket's last collective is done
                    e1.record()
                self._tev.append((bi, e0, e1))

    def _launch_collective(self, bi):
        start, end, _ = self.buckets[bi]
        chunk = self.flat[start:end]
        if self.mode == "reduce_scatter":
            n = (end - start) // self.world
            shard = chunk[self.rank * n:(self.rank + 1) * n]
            if self._gloo:      # gloo runs async works on a thread pool, not in issue order: the two phases synchronously (tests only)
                tdist.reduce_scatter_tensor(shard, chunk, op=tdist.ReduceOp.SUM, group=self.group)
                if self.average:
                    shard.div_(self.world)
                tdist.all_gather_into_tensor(chunk, shard, group=self.group)
                return
            op = tdist.ReduceOp.AVG if self.average else tdist.ReduceOp.SUM
            # RCCL executes a communicator's collectives in issue order on its own stream: the gather follows the scatter
            self._works.append((tdist.reduce_scatter_tensor(shard, chunk, op=op, group=self.group, async_op=True), None))
            self._works.append((tdist.all_gather_into_tensor(chunk, shard, group=self.group, async_op=True), None))
            return
        if self.average and not self._gloo:
            self._works.append((tdist.all_reduce(chunk, op=tdist.ReduceOp.AVG, group=self.group, async_op=True), None))
        else:   # gloo has no AVG: SUM then divide (exactly gather_grad's arithmetic)
            self._works.append((tdist.all_reduce(chunk, op=tdist.ReduceOp.SUM, group=self.group, async_op=True),
                                chunk if self.average else None))

    def finish(self):
        """Launch what backward left incomplete (parameters without a gradient this step, e.g. a DropPath-dropped
        branch, contribute zeros), then wait for every bucket.  Afterwards each p.grad is a view of `flat`."""
        for bi in range(self._next, len(self.buckets)):
            for p, v in self.buckets[bi][2]:
                if p not in self._fired:
                    v.zero_()
                    p.grad = v
            self._launch(bi)
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)      # the optimizer waits for the rehearsal "collectives" as it would for RCCL's
        for work, chunk, *tev in self._works:
            work.wait()
            if chunk is not None:
                chunk.div_(self.world)
            if tev:                     # (gloo timing: the host wait above is the completion)
                tev[0][2].record()
                self._tev.append(tev[0])
        self._works = []
        self._fired = set()
        self._next = 0
        self._pending = [len(b[2]) for b in self.buckets]
        if self._cuda:
            for p in self._slot:
                self._ops.release_grad_slot(p)

    def bucket_times_ms(self):
        """timing=True: per bucket, the mean over the steps since the last call of (launch -> its last collective done) in ms,
        bucket 0 = the first one backward completes (the output layer); None for buckets never launched.  Synchronises the device."""
        if not self._tev:
            return None
        torch.cuda.synchronize(self.flat.device)
        acc = [[0.0, 0] for _ in self.buckets]
        for bi, e0, e1 in self._tev:
            acc[bi][0] += e0.elapsed_time(e1)
            acc[bi][1] += 1
        self._tev = []
        return [a / n if n else None for a, n in acc]

    def bucket_bytes(self):
        return [(end - start) * self.flat.element_size() for start, end, _ in self.buckets]

    def zero_grad(self):
        """Zero the flat buffer in one memset; gradients stay views into it."""
        self.flat.zero_()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self._cuda:
            self._ops.unregister_grad_slots(self)


def gather_grad(params, world_size=None):
    """reference utils_dist.py:125-134 verbatim semantics (per-parameter SUM then divide) — kept for API parity and as
    the slow baseline the flat-buffer path is checked against."""
    world_size = world_size or get_dist_info()[1]
    for p in params:
        if p.grad is not None:
            tdist.all_reduce(p.grad.data, op=tdist.ReduceOp.SUM)
            p.grad.data.div_(world_size)
