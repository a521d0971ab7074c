"""Input pipeline: keep the GPU fed once a step takes tens of milliseconds.

The reference's loop moves four pageable tensors per step with `.to(device)` (models/pangu_sample.py:41-43: 573 MB, on the
training thread, each copy staged through the runtime's own bounce buffers) and synchronises every step (`loss.item()`, :77);
its reader reverses the level axis on the host (`[::-1]`, era5_data/utils_data.py:117).  `DevicePrefetcher` is the idea of the
reference's (unused) DataPrefetcher (era5_data/utils_data.py:16-51) done for batches of that size:

  * a WORKER THREAD pulls the next batch from the loader, stages it into page-locked buffers that are allocated once
    (`pangu_host_copy`: the 573 MB split over a few host threads -- one thread copies at 2-3 GB/s, a 43 ms step needs 13.4 GB/s)
    and queues the host->device copies on a side stream; the training thread never touches host memory;
  * `depth` batches are staged AHEAD of the consumer, so a per-step `loss.item()` does not expose the copy of the next batch;
  * loaders that can write into a given buffer skip the staging copy altogether (`fill_pinned` protocol below);
  * the level reversal costs nothing: with `fuse_flip=True` the fields stay in file order and the first kernel that reads them
    (patch_embed_gather) / the loss kernel address level 12 - l (`levels_reversed=True` of PanguModel.forward / train.train_step).
"""
import os
import queue
import threading
import time

import torch


class PinnedFiller:
    """Protocol (duck-typed; subclassing is optional) of a loader that writes each batch straight into page-locked memory:

        spec          list of (shape, dtype), one per tensor of a batch (input, input_surface, target, target_surface, ...)
        fill_pinned(buffers) -> bool     write the next batch into `buffers` (CPU tensors of `spec`, page-locked); False = exhausted
        reset()       optional: rewind for another epoch (called at the start of every iteration)
        __len__()     optional

    A reader that decodes its file format into the buffers it is handed (numpy views of them: `buf.numpy()`) removes the only
    host-side copy of the pipeline."""
    spec = ()

    def fill_pinned(self, buffers):
        raise NotImplementedError


def default_copy_threads():
    """Host threads of one staging copy: up to 8, and never more than half of this rank's share of the host's cores."""
    world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1)
    return max(1, min(8, (os.cpu_count() or 1) // (2 * max(world, 1))))


def host_copy(dst, src, threads):
    """dst (page-locked CPU tensor) <- src (CPU tensor): the multi-threaded C-ABI copy for plain contiguous same-dtype tensors,
    `Tensor.copy_` (dtype conversion / strides) otherwise."""
    if (src.device.type == "cpu" and src.dtype == dst.dtype and src.shape == dst.shape and src.is_contiguous()
            and dst.is_contiguous()):
        from . import _lib
        _lib.check(_lib.load().pangu_host_copy(dst.data_ptr(), src.data_ptr(), src.numel() * src.element_size(), int(threads)),
                   "host_copy")
    else:
        dst.copy_(src)


_END = object()


class DevicePrefetcher:
    """Wrap an iterable of (input, input_surface, target, target_surface, *rest) CPU batches -- or a PinnedFiller.

    flip_levels   the reader delivers ascending levels: reverse the pressure-level axis (dim -3) of input / target.
    fuse_flip     with flip_levels: do NOT move any data; batches are yielded in file order and `self.levels_reversed` is True --
                  pass it on (`train_step(..., levels_reversed=pf.levels_reversed)`).  False: a device-side `flip` per field.
    depth         batches staged ahead of the consumer (>= 1; 2 hides the copy behind a per-step host sync).
    threaded      False: stage on the consumer's thread (the pre-round-6 behaviour; for debugging / comparison).
    copy_threads  host threads of one staging copy (default_copy_threads()).
    reuse_device_buffers   True: the device tensors of a batch are `depth + 3` STATIC buffer sets filled in turn (no allocator traffic
                  in the loop: a fresh 270 MB block per field and step can mean a synchronous hipMalloc when the caching allocator's
                  pool is fragmented) -- a yielded batch is valid until the consumer asks for the NEXT one (an event recorded then
                  orders the buffer's refill behind the consumer's work on it -- work enqueued on the stream that is CURRENT when the next
                  batch is requested); keep nothing from a batch across iterations.
                  False (default): every batch is freshly allocated and stays valid as long as it is referenced."""

    def __init__(self, loader, device, flip_levels=False, depth=2, fuse_flip=False, threaded=True, copy_threads=None,
                 reuse_device_buffers=False):
        self.loader, self.device = loader, torch.device(device)
        self.flip = bool(flip_levels) and not fuse_flip
        self.levels_reversed = bool(flip_levels) and bool(fuse_flip)
        self.depth = max(1, int(depth))
        self.threaded = bool(threaded)
        self.copy_threads = int(copy_threads) if copy_threads else default_copy_threads()
        # (high priority: the uploads are tiny next to a step's kernels and must not queue behind them)
        self.stream = torch.cuda.Stream(device=self.device, priority=-1)
        self._nslots = self.depth + 1          # `depth` queued + the one being staged
        self._pinned = [None] * self._nslots   # per slot: list of page-locked host buffers
        self._busy = [None] * self._nslots     # per slot: event of the last host->device copy that read those buffers
        self._slot = 0
        self.reuse = bool(reuse_device_buffers)
        self._ndev = self.depth + 3            # queued + being staged + held by the consumer + one whose release was just recorded
        self._dev = [None] * self._ndev        # per device slot: list of static device tensors
        self._consumed = [None] * self._ndev   # per device slot: event on the consumer's stream after its last use of that slot
        self._dslot = 0
        self._filler = hasattr(loader, "fill_pinned")
        # what the pipeline did, for bench.py: seconds in the staging copy / bytes staged / seconds the consumer waited for a batch
        self.stats = {"host_copy_s": 0.0, "bytes": 0, "batches": 0, "consumer_wait_s": 0.0, "h2d_events": []}

    def __len__(self):
        return len(self.loader)

    # ---- staging (worker thread, or the consumer's when threaded=False)
    def _slot_buffers(self, slot, spec):
        pins = self._pinned[slot]
        if pins is None or len(pins) != len(spec) or any(tuple(p.shape) != tuple(sh) or p.dtype != dt for p, (sh, dt) in zip(pins, spec)):
            pins = self._pinned[slot] = [torch.empty(tuple(sh), dtype=dt, pin_memory=True) for sh, dt in spec]
        return pins

    def _next_slot(self):
        slot = self._slot
        self._slot = (slot + 1) % self._nslots
        if self._busy[slot] is not None:
            self._busy[slot].synchronize()      # the async copy that last read this slot's buffers must be done (blocks the stager only)
        return slot

    def _upload(self, slot, pins, layout):
        """pins: the slot's page-locked tensors; layout: per batch element either an index into pins or ('obj', value)."""
        out, dslot = [], -1
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            if self.reuse:
                dslot = self._dslot
                self._dslot = (dslot + 1) % self._ndev
                dev = self._dev[dslot]
                if dev is None or len(dev) != len(pins) or any(d.shape != p.shape or d.dtype != p.dtype for d, p in zip(dev, pins)):
                    dev = self._dev[dslot] = [torch.empty(p.shape, dtype=p.dtype, device=self.device) for p in pins]
                if self._consumed[dslot] is not None:
                    self.stream.wait_event(self._consumed[dslot])      # the consumer's last kernels on this buffer set come first
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(self.stream)
            for i, item in enumerate(layout):
                if isinstance(item, tuple):
                    out.append(item[1])
                    continue
                if self.reuse:
                    d = dev[item]
                    d.copy_(pins[item], non_blocking=True)
                else:
                    d = pins[item].to(self.device, non_blocking=True)
                if self.flip and i in (0, 2) and d.dim() >= 4:
                    d = d.flip(-3)
                out.append(d)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(self.stream)
        self._busy[slot] = ev
        st = self.stats
        st["h2d_events"].append((e0, ev, sum(p.numel() * p.element_size() for p in pins)))
        del st["h2d_events"][:-16]
        return out, ev, dslot

    def _stage(self, batch):
        tensors = [t for t in batch if torch.is_tensor(t)]
        slot = self._next_slot()
        pins = self._slot_buffers(slot, [(t.shape, t.dtype) for t in tensors])
        t0 = time.perf_counter()
        for pin, t in zip(pins, tensors):
            host_copy(pin, t, self.copy_threads)              # host -> page-locked (the only host-side copy)
        st = self.stats
        st["host_copy_s"] += time.perf_counter() - t0
        st["bytes"] += sum(p.numel() * p.element_size() for p in pins)
        st["batches"] += 1
        layout, k = [], 0
        for t in batch:
            if torch.is_tensor(t):
                layout.append(k)
                k += 1
            else:
                layout.append(("obj", t))
        return self._upload(slot, pins, layout)

    def _stage_filled(self):
        slot = self._next_slot()
        pins = self._slot_buffers(slot, list(self.loader.spec))
        t0 = time.perf_counter()
        if not self.loader.fill_pinned(pins):
            return None
        st = self.stats
        st["host_copy_s"] += time.perf_counter() - t0          # (here: the loader's own decode-into-buffer time)
        st["bytes"] += sum(p.numel() * p.element_size() for p in pins)
        st["batches"] += 1
        return self._upload(slot, pins, list(range(len(pins))))

    def _staged(self):
        """Generator of staged (batch, event) pairs, in the calling thread."""
        if self._filler:
            if hasattr(self.loader, "reset"):
                self.loader.reset()
            while True:
                item = self._stage_filled()
                if item is None:
                    return
                yield item
        else:
            for batch in self.loader:
                yield self._stage(batch)

    def _worker(self, q, stop):
        try:
            for item in self._staged():
                while not stop.is_set():
                    try:
                        q.put(item, timeout=0.1)
                        break
                    except queue.Full:
                        continue
                if stop.is_set():
                    return
            q.put(_END)
        except BaseException as e:       # delivered to the consumer, which re-raises it
            q.put(e)

    def _hand_over(self, batch, ev, dslot=-1):
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        if dslot < 0:
            for t in batch:
                if torch.is_tensor(t):
                    t.record_stream(cur)
        return tuple(batch)

    def _release(self, dslot):
        """The consumer asked for the next batch: everything it enqueued on the batch in device slot `dslot` precedes this event."""
        if dslot >= 0:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._consumed[dslot] = ev

    def __iter__(self):
        if not self.threaded:
            # one batch staged ahead, on this thread (its host copy overlaps nothing but the GPU's queue)
            gen = self._staged()
            pending = next(gen, None)
            while pending is not None:
                batch, ev, dslot = pending
                pending = next(gen, None)
                yield self._hand_over(batch, ev, dslot)
                self._release(dslot)
            return
        q = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        th = threading.Thread(target=self._worker, args=(q, stop), name="pangu-prefetch", daemon=True)
        th.start()
        try:
            while True:
                t0 = time.perf_counter()
                while True:                   # never block forever on a worker that died without posting its end marker
                    try:
                        item = q.get(timeout=1.0)
                        break
                    except queue.Empty:
                        if not th.is_alive():
                            try:
                                item = q.get_nowait()
                                break
                            except queue.Empty:
                                raise RuntimeError("DevicePrefetcher: the staging thread ended without delivering a batch or its end marker")
                self.stats["consumer_wait_s"] += time.perf_counter() - t0
                if item is _END:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield self._hand_over(*item)
                self._release(item[2])
        finally:
            stop.set()
            while th.is_alive():          # unblock a worker stuck on a full queue (the consumer left early)
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.05)

    def summary(self):
        """Rates of the pipeline so far (call after a synchronize): staging copy GB/s (host), host->device GB/s (copy engine,
        over the last batches), seconds the consumer waited for batches."""
        st = self.stats
        h2d_ms = h2d_b = 0.0
        for e0, e1, nbytes in st["h2d_events"]:
            if e1.query():
                h2d_ms += e0.elapsed_time(e1)
                h2d_b += nbytes
        return {"batches": st["batches"], "bytes_per_batch": st["bytes"] / max(st["batches"], 1),
                "host_stage_GBps": st["bytes"] / st["host_copy_s"] / 1e9 if st["host_copy_s"] > 0 else None,
                "host_stage_ms_per_batch": st["host_copy_s"] / max(st["batches"], 1) * 1e3,
                "h2d_GBps": h2d_b / (h2d_ms * 1e-3) / 1e9 if h2d_ms > 0 else None,
                "consumer_wait_ms_per_batch": st["consumer_wait_s"] / max(st["batches"], 1) * 1e3,
                "copy_threads": self.copy_threads, "depth": self.depth, "threaded": self.threaded,
                "direct_fill": self._filler, "levels_reversed_fused": self.levels_reversed, "static_device_buffers": self.reuse}
