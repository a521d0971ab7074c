"""Input pipeline: keep the GPU fed once a step takes tens of milliseconds.

`DevicePrefetcher` is the idea of the reference's (unused) DataPrefetcher (era5_data/utils_data.py:16-51) done for 573 MB
batches: page-locked staging buffers allocated ONCE, the next batch copied host->device on a side stream while the
current step computes, double-buffered device tensors, and the level reversal of the reader
(`upper[:, ::-1]`, utils_data.py:117) moved onto the device so the host never makes the extra 286 MB copy."""
import torch


class DevicePrefetcher:
    """Wrap an iterable of (input, input_surface, target, target_surface, *rest) CPU batches.

    flip_levels=True reverses the pressure-level axis (dim -3) of input/target on the device (for readers that deliver
    ascending levels)."""

    def __init__(self, loader, device, flip_levels=False, depth=2):
        self.loader, self.device, self.flip = loader, torch.device(device), flip_levels
        self.stream = torch.cuda.Stream(device=self.device)
        self.depth = depth
        self._pinned = [None] * depth          # per slot: list of pinned host buffers
        self._busy = [None] * depth            # per slot: event of the last host->device copy that read those buffers
        self._slot = 0

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        slot = self._slot
        self._slot = (slot + 1) % self.depth
        tensors = [t for t in batch if torch.is_tensor(t)]
        if self._busy[slot] is not None:
            self._busy[slot].synchronize()      # the async copy of the batch staged here `depth` batches ago must be done
        if self._pinned[slot] is None or any(p.shape != t.shape or p.dtype != t.dtype
                                             for p, t in zip(self._pinned[slot], tensors)):
            self._pinned[slot] = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in tensors]
        out, k = [], 0
        with torch.cuda.stream(self.stream):
            for i, t in enumerate(batch):
                if not torch.is_tensor(t):
                    out.append(t)
                    continue
                pin = self._pinned[slot][k]
                k += 1
                pin.copy_(t)                                   # host -> pinned (the only host-side copy)
                d = pin.to(self.device, non_blocking=True)
                if self.flip and i in (0, 2) and d.dim() >= 4:
                    d = d.flip(-3)
                out.append(d)
        ev = torch.cuda.Event()
        ev.record(self.stream)
        self._busy[slot] = ev
        return out, ev

    def __iter__(self):
        it = iter(self.loader)
        pending = None
        try:
            pending = self._stage(next(it))
        except StopIteration:
            return
        while pending is not None:
            batch, ev = pending
            try:
                pending = self._stage(next(it))                # overlaps with the consumer's compute
            except StopIteration:
                pending = None
            torch.cuda.current_stream(self.device).wait_event(ev)
            for t in batch:
                if torch.is_tensor(t):
                    t.record_stream(torch.cuda.current_stream(self.device))
            yield tuple(batch)
