"""Autoregressive rollout (BASELINE configs[4]: 7 x 24 h) with the forward step captured in a hipGraph.

Shape of the loop: reference inference/inference_singleOutput.py:97-105 (the output of one 24 h step is the input of
the next).  The torch model returns NORMALISED fields (reference models/layers.py:531,542 leave the de-normalisation
commented out), so each step is followed by `normBackData` (reference era5_data/utils_data.py:324-330) before the
fields are fed back; here that de-normalisation is folded into the forward's last kernel (`pangu_patch_recover_scatter_denorm`
writes the physical fields straight into the step's static input buffers), inside the same captured graph, so the 286 MB state
never leaves the device, costs no extra pass, and one rollout step is ONE graph launch.
"""
import torch


def norm_back(upper, surface, stats_last):
    """reference era5_data/utils_data.py:324-330; stats_last = (s_mean(1,4,1,1), s_std, u_mean(1,5,13,1,1), u_std)."""
    s_mean, s_std, u_mean, u_std = stats_last
    return upper * u_std + u_mean, surface * s_std + s_mean


def norm_back_into(upper, surface, stats_last, dst_upper, dst_surface):
    """norm_back written straight into existing buffers (the rollout's next-step inputs): the same two roundings
    (multiply, then add) as the reference expression, without the temporaries and the copy pass."""
    s_mean, s_std, u_mean, u_std = stats_last
    torch.mul(upper, u_std, out=dst_upper).add_(u_mean)
    torch.mul(surface, s_std, out=dst_surface).add_(s_mean)


class GraphedStep:
    """One model step captured as a hipGraph (static shapes, B fixed, no host syncs inside the path).

    step(): input buffers -> model -> (optionally) de-normalised outputs copied back into the input buffers.
    """

    def __init__(self, model, inp, inp_surface, statistics, maps, const_h, stats_last=None, feed_back=False, warmup=2):
        assert inp.is_cuda and not any(p.requires_grad and torch.is_grad_enabled() for p in ())
        self.model = model
        self.inp = inp.clone()
        self.inp_surface = inp_surface.clone()
        self.consts = (statistics, maps, const_h)
        self.stats_last = stats_last
        self.feed_back = feed_back
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                 # warm-up outside capture: weight shadows, attribute calls, allocator
                self._body()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.out, self.out_surface = self._body()
        # warm-up advanced the state when feed_back is on: restore the caller's initial fields
        self.inp.copy_(inp)
        self.inp_surface.copy_(inp_surface)

    def _body(self):
        if self.feed_back:
            # normBackData folded into the forward's last kernel: the scatter writes the physical fields straight into the
            # input buffers (read by the first kernel of this same forward, long finished by then)
            from . import ops
            with ops.scatter_denorm(self.inp, self.inp_surface, self.stats_last):
                return self.model(self.inp, self.inp_surface, *self.consts)
        return self.model(self.inp, self.inp_surface, *self.consts)

    def load(self, inp, inp_surface):
        self.inp.copy_(inp)
        self.inp_surface.copy_(inp_surface)

    def step(self):
        """Replay the graph once; returns the (static) normalised output tensors of this step."""
        self.graph.replay()
        return self.out, self.out_surface


def rollout(model, inp, inp_surface, statistics, maps, const_h, stats_last, steps=7, graph=True, keep=False):
    """`steps` chained forwards. Returns the last step's physical-unit fields (and, with keep=True, a list of the
    normalised outputs of every step, cloned)."""
    history = []
    with torch.no_grad():
        if graph:
            g = GraphedStep(model, inp, inp_surface, statistics, maps, const_h, stats_last, feed_back=True)
            for _ in range(steps):
                out, out_s = g.step()
                if keep:
                    history.append((out.clone(), out_s.clone()))
            up, sf = g.inp.clone(), g.inp_surface.clone()
        else:
            up, sf = inp, inp_surface
            for _ in range(steps):
                out, out_s = model(up, sf, statistics, maps, const_h)
                if keep:
                    history.append((out.clone(), out_s.clone()))
                up, sf = norm_back(out, out_s, stats_last)
    return (up, sf, history) if keep else (up, sf)
