"""hipGraph capture of a whole model step (torch.cuda.CUDAGraph is hipGraph on ROCm).

The boundary network Dirichlet_BC_NN_Legacy_2 and the end-to-end Poisson_CNN_Legacy issue ~1000 kernels of 10-20 us per training step on
288 x 288 grids: on the host's launch path that is 60-70 ms of which the GPU computes a third.  A step on FIXED shapes is a fixed launch
sequence (libpcnn has no host-side data dependence: no readbacks, workspaces grow only until the largest shape has been seen), so it is
captured once and replayed: the launch-bound inner loop becomes one graph launch (MI355X guide: "capture launch-bound inner loops in
hipGraphs").  Replays are bit-identical to eager steps - the same kernels in the same order on the same buffers.

    step = GraphedTrainStep(model, ((bc, dx), target))        # warm-up + capture
    logs = step(((bc, dx), target))                            # copy inputs into the static buffers, replay, eager optimizer step

The optimizer step stays eager (one launch per parameter bucket): its learning rate and iteration count are host scalars that change from
step to step (ReduceLROnPlateau, Adam bias correction).  Data-parallel gradient reduction (model.grad_sync) runs eagerly between the two.
Shapes other than the captured ones need their own GraphedTrainStep (the reference's generators draw a new grid shape per batch: keep a
small dict of them, or use the eager path)."""
import torch


def _ctxs(model):
    return [m.ctx for m in ([model] + [getattr(model, n) for n in ('hpnn', 'dbcnn') if hasattr(model, n)]) if hasattr(m, 'ctx')]


def _static(v, device):
    if isinstance(v, torch.Tensor):
        return v.detach().to(device).clone()
    if hasattr(v, 'shape') and hasattr(v, 'dtype'):                 # numpy
        return torch.as_tensor(v).to(device).clone()
    return v                                                         # ints (x_output_resolution) are part of the captured shape


def _refill(static, new):
    for s, v in zip(static, new):
        if isinstance(s, torch.Tensor):
            v = v if isinstance(v, torch.Tensor) else torch.as_tensor(v)
            if tuple(v.shape) != tuple(s.shape):
                raise ValueError('this step was captured for shape %s, got %s: capture another GraphedTrainStep for it' % (tuple(s.shape), tuple(v.shape)))
            s.copy_(v, non_blocking=True)
        elif s != v:
            raise ValueError('this step was captured for %r, got %r' % (s, v))


class _Captured:
    def __init__(self, model, warmup):
        if not torch.cuda.is_available():
            raise RuntimeError('hipGraph capture needs the GPU')
        self.model, self.device = model, model.device
        for c in _ctxs(model):                                       # the weight-gradient side stream is an eager-mode overlap; one stream is captured
            c.side_allowed = False
            c.use_side = False
        self.stream = torch.cuda.Stream()
        self.warmup = max(1, int(warmup))

    def _capture(self, fn):
        torch.cuda.synchronize()
        with torch.cuda.stream(self.stream):                          # libpcnn handles are per stream: the warm-up creates this stream's handle,
            for _ in range(self.warmup):                              # its workspaces and every per-shape cache before anything is recorded
                fn()
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            out = fn()
        torch.cuda.synchronize()
        return out


class GraphedInference(_Captured):
    """model(inputs) on fixed shapes as one graph launch; returns the static output tensor (copy it if it must survive the next call)."""

    def __init__(self, model, example_inputs, warmup=2):
        super().__init__(model, warmup)
        self.static_in = [_static(v, self.device) for v in example_inputs]
        self.out = self._capture(lambda: model(self.static_in))

    def __call__(self, inputs):
        _refill(self.static_in, inputs)
        self.graph.replay()
        return self.out


class GraphedTrainStep(_Captured):
    """model.train_step(data) on fixed shapes: forward, loss, backward and the metrics replayed as one graph; gradient all-reduce (if the model
    is data-parallel) and the optimizer step eager."""

    def __init__(self, model, example_data, warmup=2):
        super().__init__(model, warmup)
        inputs, y = example_data
        self.static_in = [_static(v, self.device) for v in inputs]
        self.static_y = _static(y, self.device)
        opt = model.optimizer
        real_apply, real_sync = opt.apply_gradients, model.grad_sync
        snap = [(s, s.flat_w.clone(), s.flat_stats.clone()) for s in model.stores]
        opt.apply_gradients = lambda *a, **k: None
        model.grad_sync = None
        try:
            self.logs = self._capture(lambda: model.train_step((self.static_in, self.static_y)))
        finally:
            opt.apply_gradients, model.grad_sync = real_apply, real_sync
            del opt.__dict__['apply_gradients']                       # back to the class's method (the instance attribute shadowed it)
        for s, w, st in snap:                                         # warm-up and capture ran the step without an optimizer: only the BN statistics
            s.flat_w.copy_(w); s.flat_stats.copy_(st)                 # of a training-mode-BN model could have moved

    def __call__(self, data):
        inputs, y = data
        _refill(self.static_in, inputs)
        _refill([self.static_y], [y])
        self.graph.replay()
        m = self.model
        if m.grad_sync is not None:
            for s in m.stores:
                m.grad_sync(s.flat_g)
        m.optimizer.apply_gradients()
        return self.logs
