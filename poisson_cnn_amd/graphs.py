"""hipGraph capture of a whole model step (torch.cuda.CUDAGraph is hipGraph on ROCm).

A step on FIXED shapes is a fixed launch sequence (libpcnn has no host-side data dependence: no readbacks, workspaces grow only until the
largest shape has been seen), so it can be captured once and replayed as one graph launch; replays are bit-identical to eager steps - the
same kernels in the same order on the same buffers.  MEASURED BENEFIT ON THE SHIPPED MODELS: NONE (dbcnn.json 60.6 vs
59.4 ms, pcnn 68.7 vs 69.3 ms, hpnn unchanged) - rocprofv3 shows those steps are ~67 ms of kernel time, i.e. GPU-bound, not launch-bound;
the feature is kept for hosts whose launch path is slower than this one's (many small grids per step), not as an optimisation of the benchmarks.

    step = GraphedTrainStep(model, ((bc, dx), target))        # warm-up + capture
    logs = step(((bc, dx), target))                            # copy inputs into the static buffers, replay, eager optimizer step

The optimizer step stays eager (one launch per parameter bucket): its learning rate and iteration count are host scalars that change from
step to step (ReduceLROnPlateau, Adam bias correction).  Data-parallel collectives - the gradient all-reduce (model.grad_sync) AND the
2-element metric all-reduce (model.metric_sync) - are never recorded: they run eagerly after the replay, and the returned logs carry the
optimizer's CURRENT learning rate.
Shapes other than the captured ones need their own GraphedTrainStep (the reference's generators draw a new grid shape per batch: keep a
small dict of them, or use the eager path).  Scratch ownership: a capture keeps strong references to every model-level scratch buffer its
launches address (Context.ws / ws_side / the flipped-filter scratch, ops._default_ws) and switches the capture stream's libpcnn handle to
pcnn_set_workspace_retain, so a LATER capture or eager step on a larger shape - which makes those owners allocate bigger buffers - never
frees memory an earlier graph still replays into."""
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
        self.stream = torch.cuda.Stream()
        self.warmup = max(1, int(warmup))
        self._keep = []

    def _capture(self, fn):
        from ctypes import c_int
        from . import ops
        ctxs = _ctxs(self.model)
        saved = [(c, c.side_allowed, c.use_side) for c in ctxs]
        for c in ctxs:                                                # the weight-gradient side stream is an eager-mode overlap; ONE stream is
            c.side_allowed = False                                    # captured.  Restored below: later eager steps keep their overlap.
            c.use_side = False
        try:
            torch.cuda.synchronize()
            with torch.cuda.stream(self.stream):                      # libpcnn handles are per stream: the warm-up creates this stream's handle,
                for _ in range(self.warmup):                          # its workspaces and every per-shape cache before anything is recorded
                    fn()
                ops.handle().call('pcnn_set_workspace_retain', c_int(1))
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            # the flipped-filter table of the layer set the warm-up registered, built eagerly: the capture records one table launch (not one flip
            # per layer), and this graph owns a reference to the table and to every flipped filter it addresses
            self._keep += ops.prepare_flip_table_for_capture()
            ops.weights_changed()                                     # a replay must refresh the cached filter spectra (the weights move between replays,
            with torch.cuda.graph(self.graph, stream=self.stream):    # no host code runs): the refresh launch is recorded only if the version is new here
                out = fn()
            torch.cuda.synchronize()
        finally:
            for c, allowed, use in saved:
                c.side_allowed, c.use_side = allowed, use
        # the recorded launches address these buffers; their owners drop them when a larger shape comes along - the graph must not
        for c in ctxs:
            self._keep += c.all_scratch()
        self._keep.append(ops._default_ws.buf)
        self._keep = [b for b in self._keep if b is not None]
        return out

    def close(self):
        """Releases what the capture holds beyond the model itself: the graph, the references that kept its scratch alive, the capture
        stream's libpcnn handle (pcnn_destroy frees its workspaces and every buffer parked under pcnn_set_workspace_retain) and the per-stream
        scratch entries of the model's contexts.  A 'dict of graphs per shape' therefore costs device memory only while its entries are alive;
        an object that is dropped without close() is closed by its finaliser.  Idempotent; the object cannot be called afterwards."""
        if getattr(self, 'stream', None) is None:
            return
        from . import ops
        try:
            torch.cuda.synchronize()
            self.graph = None
            self._keep = []
            sp = self.stream.cuda_stream
            for c in _ctxs(self.model):
                c.drop_stream(sp)
            ops.release_stream_handle(sp)
        finally:
            self.stream = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 - interpreter shutdown: the driver reclaims everything anyway
            pass


class GraphedInference(_Captured):
    """model(inputs) on fixed shapes as one graph launch; returns the static output tensor (copy it if it must survive the next call)."""

    def __init__(self, model, example_inputs, warmup=2):
        super().__init__(model, warmup)
        self.static_in = [_static(v, self.device) for v in example_inputs]
        self.out = self._capture(lambda: model(self.static_in))

    def __call__(self, inputs):
        if self.stream is None:
            raise RuntimeError('this captured graph has been closed')
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
        real_apply, real_sync, real_metric = opt.apply_gradients, model.grad_sync, getattr(model, 'metric_sync', None)
        snap = [(s, s.flat_w.clone(), s.flat_stats.clone()) for s in model.stores]
        opt.apply_gradients = lambda *a, **k: None
        model.grad_sync = None
        model.metric_sync = None                                      # no collective inside the graph: the LOCAL loss / mse tensors are recorded
        try:
            self.logs = self._capture(lambda: model.train_step((self.static_in, self.static_y)))
        finally:
            opt.apply_gradients, model.grad_sync, model.metric_sync = real_apply, real_sync, real_metric
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
        logs = dict(self.logs)                                        # static local loss / mse tensors of the replay
        if getattr(m, 'metric_sync', None) is not None:               # data parallel: the global metrics, eagerly, as train_step reports them
            logs['loss'], logs['mse'] = m.metric_sync(logs['loss'], logs['mse'])
        if 'lr' in logs:
            logs['lr'] = m.optimizer.learning_rate                    # the host scalar as it is NOW (ReduceLROnPlateau), not at capture time
        return logs
