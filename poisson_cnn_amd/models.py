"""Homogeneous_Poisson_NN_Legacy on the libpcnn HIP kernels.

Drop-in for poisson_CNN/models/Homogeneous_Poisson_NN_Legacy.py:10-296: same constructor kwargs (the "model" section of
experiments/hpnn.json loads unchanged), `model([rhs, dx]) -> (N,1,H,W)`, `compile(loss, optimizer)`,
`train_step(((rhs, dx), y))`, `fit(sequence, epochs, callbacks)`, `trainable_variables`, `get_weights/set_weights`,
`save_weights/load_weights`, `summary()`.  The API is channels_first like the reference configs; inside, data is NHWC.
"""
import copy

import os

import numpy as np
import torch

from . import layers as L
from . import ops
from .utils import get_init_arguments_from_config


def process_normalizations(normalizations):
    """models/Homogeneous_Poisson_NN_Metalearning.py:27-45."""
    types, defaults = ['rhs_max_magnitude'], [False]
    if normalizations is None:
        return dict(zip(types, defaults))
    normalizations = dict(normalizations)
    for k, d in zip(types, defaults):
        normalizations.setdefault(k, d)
    if isinstance(normalizations['rhs_max_magnitude'], bool) and normalizations['rhs_max_magnitude']:
        normalizations['rhs_max_magnitude'] = 1.0
    return normalizations


def process_output_scaling_modes(output_scalings):
    """models/Homogeneous_Poisson_NN_Metalearning.py:47-57."""
    modes = ['rhs_max_magnitude', 'max_domain_size_squared', 'match_peak_laplacian_magnitude_to_peak_rhs', 'soln_max_magnitude']
    out = {m: False for m in modes}
    if isinstance(output_scalings, dict):
        out.update(copy.deepcopy(output_scalings))
    return out


def _as_device(t, device):
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(t))
    return t.to(device=device, dtype=torch.float32).contiguous()


def _host_logs(logs):
    """train_step's dict with device scalars -> Python floats through ONE device-to-host copy (each float(tensor) is a stream synchronise of its own)."""
    dev = [(k, v) for k, v in logs.items() if isinstance(v, torch.Tensor)]
    out = {k: (v if isinstance(v, float) else float(v)) for k, v in logs.items() if not isinstance(v, torch.Tensor)}
    if dev:
        vals = torch.stack([v.reshape(()).float() for _, v in dev]).tolist()
        out.update({k: x for (k, _), x in zip(dev, vals)})
    return {k: out[k] for k in logs}


class _Prefetcher:
    """fit()'s input pipeline: batch i + 1 is produced while step i runs - what Keras' OrderedEnqueuer does for the reference's Sequence generators
    (train/hpnn_legacy_train.py:60 `model.fit(dataset, ...)`).  The generators of poisson_cnn_amd.dataset draw O(10)-sized parameters on the host
    (13 ms per batch of 50 for the shipped config) and synthesise the fields on the device: run on a stream of their own, their uploads and kernels
    do not queue behind the training step, and the host work hides under the step's kernels.  The consumer stream waits for the producer's event;
    the tensors are handed over with record_stream so that the caching allocator does not recycle them early.  PCNN_PREFETCH=0: generate in line."""

    def __init__(self, dataset, device):
        self.dataset = dataset
        self.on = os.environ.get('PCNN_PREFETCH', '1') != '0' and torch.cuda.is_available() and torch.device(device).type == 'cuda'
        self.stream = torch.cuda.Stream(device=device) if self.on else None
        self.pending = None

    def request(self, idx):
        if not self.on:
            self.pending = (idx, None, None)
            return
        # (no wait for the training stream: the generator allocates fresh tensors, and blocks the consumer released are recycled by the caching
        # allocator only after the events record_stream left on them - so the synthesis kernels overlap the step as well)
        with torch.cuda.stream(self.stream):
            batch = self.dataset[idx]
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.pending = (idx, batch, ev)

    def close(self):
        """Drops a batch that was requested and never taken, and the libpcnn handle of the producer stream (its scratch and workspaces)."""
        self.pending = None
        if self.on and self.stream is not None:
            try:
                self.stream.synchronize()
                ops.release_stream_handle(self.stream.cuda_stream, torch.device(self.stream.device).index)
            finally:
                self.stream = None
                self.on = False

    def take(self):
        idx, batch, ev = self.pending
        self.pending = None
        if ev is None:
            return self.dataset[idx]
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)

        def hand_over(v):
            if isinstance(v, torch.Tensor) and v.is_cuda:
                v.record_stream(cur)
            elif isinstance(v, (list, tuple)):
                for u in v:
                    hand_over(u)
        hand_over(batch)
        return batch


class _ModelBase:
    """What the three model classes share: the Keras-style weight API over a ParamStore, compile(), fit()."""
    model_name = 'model'

    # ------------------------------------------------------------------ weights
    @property
    def trainable_variables(self):
        return [self.store.w[n] for n in self.store.trainable_names()]

    @property
    def weight_names(self):
        return self.store.names

    def get_weights(self):
        return [self.store.w[n].detach().cpu().numpy().copy() for n in self.store.names]

    def set_weights(self, weights):
        if isinstance(weights, dict):
            missing = set(self.store.names) - set(weights)
            if missing:
                raise ValueError('missing weights: %s' % sorted(missing)[:5])
            weights = [weights[n] for n in self.store.names]
        if len(weights) != len(self.store.names):
            raise ValueError('expected %d weight arrays, got %d' % (len(self.store.names), len(weights)))
        for n, v in zip(self.store.names, weights):
            v = np.asarray(v, dtype=np.float32)
            if tuple(v.shape) != tuple(self.store.w[n].shape):
                raise ValueError('shape mismatch for %s: %s vs %s' % (n, v.shape, tuple(self.store.w[n].shape)))
            self.store.w[n].copy_(torch.from_numpy(v))

    def save_weights(self, path, save_format=None):
        """Flat .npz keyed by parameter name (every model), or - save_format='tf' - the TensorFlow checkpoint pair <path>.index +
        <path>.data-00000-of-00001 that the reference's `model.save_weights(path)` writes (tf_checkpoint.py)."""
        if save_format in ('tf', 'tensorflow'):
            from .tf_checkpoint import save_tf_checkpoint
            return save_tf_checkpoint(self, str(path))
        if save_format not in (None, 'npz'):
            raise ValueError('save_format must be None / "npz" / "tf" (HDF5 is not supported)')
        np.savez(path, **{n.replace('/', '.'): w for n, w in zip(self.store.names, self.get_weights())})

    def load_weights(self, path):
        path = str(path)
        if os.path.exists(path + '.index'):
            from .tf_checkpoint import load_tf_checkpoint
            return load_tf_checkpoint(self, path)
        with np.load(path if path.endswith('.npz') else path + '.npz') as z:
            self.set_weights({n: z[n.replace('/', '.')] for n in self.store.names})

    def count_params(self):
        return self.store.n_trainable

    def summary(self, print_fn=print):
        print_fn('%s (MI355X / libpcnn)' % self.model_name)
        for name, shape, _, kind in self.store.specs:
            print_fn('  %-44s %-20s %s' % (name, shape, 'trainable' if kind in ('w', 'bn_gamma', 'bn_beta') else 'non-trainable'))
        print_fn('Trainable params: %d' % self.store.n_trainable)

    # ------------------------------------------------------------------ data_format
    # The network's boundary tensors all have ONE channel, so (N,1,H,W) and (N,H,W,1) are the same memory: data_format='channels_last' is a
    # reshape at the public entry points (__call__, train_step / fit); call() and everything inside stay channels_first.
    def _channels_last(self):
        return getattr(self, 'data_format', 'channels_first') == 'channels_last'

    def _cf(self, t):
        """one user tensor -> channels_first view: (N,H,W,1) -> (N,1,H,W), (N,L,1) -> (N,1,L); scalars / (N,k) arrays untouched"""
        if not self._channels_last() or not hasattr(t, 'shape') or len(t.shape) < 3:
            return t
        if t.shape[-1] != 1:
            raise ValueError('channels_last tensors of this model have one channel, got shape %s' % (tuple(t.shape),))
        t = torch.as_tensor(np.asarray(t)) if isinstance(t, np.ndarray) else t
        return t.reshape((t.shape[0], 1) + tuple(t.shape[1:-1]))

    def _cl(self, t):
        if not self._channels_last() or t.dim() < 3:
            return t
        return t.reshape((t.shape[0],) + tuple(t.shape[2:]) + (1,))

    def __call__(self, inp, training=False):
        return self._cl(self.call([self._cf(v) for v in inp], training=training))

    def train_step(self, data):
        inputs, y_true = data
        return self._train_step_cf(([self._cf(v) for v in inputs], self._cf(y_true)))

    @property
    def stores(self):
        return [self.store]

    def compile(self, loss, optimizer, max_input_shape=None):
        """max_input_shape = (per-process batch, H, W): the largest batch the training loop will see (train.main passes the maximum of the dataset's
        random_output_shape_range) - see presize()."""
        self.optimizer = optimizer
        self.loss_fn = loss
        optimizer.bind(self.stores)
        if max_input_shape is not None:
            self.presize(max_input_shape)

    def presize(self, max_input_shape):
        """One forward + backward on a synthetic batch of the LARGEST shape the training loop will see, without an optimizer step: the libpcnn handles'
        workspaces, the branch / side streams' scratch and torch's allocator pools all reach their high-water mark here, once.  The shipped training
        workload (experiments/hpnn.json:62-75: a new grid shape in [192, 384]^2 every batch, train/hpnn_legacy_train.py:26-60) otherwise re-grows them
        every time a larger shape comes along - a stream synchronise + free + allocate inside a convolution call, up to 0.6 s for one step
        (profiles/r06_train_shipped.txt).  Weights, optimizer state and BatchNormalization statistics are left exactly as they were."""
        batch = self._dummy_batch(tuple(int(v) for v in max_input_shape))
        if self.loss_fn is None:
            raise RuntimeError('presize() needs compile(loss, optimizer) first')
        stats = [(st, st.flat_stats.clone()) for st in self.stores if getattr(st, 'flat_stats', None) is not None]
        self._forward_backward(batch)
        for st in self.stores:
            st.flat_g.zero_()
        for st, snap in stats:
            st.flat_stats.copy_(snap)
        if getattr(self, '_acc', None) is not None:
            self._acc.zero_()
        torch.cuda.synchronize()

    def _dummy_batch(self, shape):
        raise NotImplementedError('%s: presize() is implemented for Homogeneous_Poisson_NN_Legacy' % type(self).__name__)

    metric_sync = None   # set by parallel.DataParallel.attach: (loss, mse) -> their GLOBAL values (one tiny all-reduce)

    def _logs(self, loss, mse):
        """What train_step returns.  Under data parallelism each rank's loss is its share of the global-batch loss (already divided by
        the global batch size, losses/loss_wrapper.py:46-49) and its mse the mean over its own samples; callbacks (ReduceLROnPlateau,
        TerminateOnNaN, ModelCheckpoint) must see the same GLOBAL numbers on every rank or the replicas' learning rates / stop decisions
        diverge - MirroredStrategy reduces the per-replica results the same way before Keras hands them to callbacks."""
        if self.metric_sync is not None:
            loss, mse = self.metric_sync(loss, mse)
        return {'loss': loss, 'mse': mse, 'lr': self.optimizer.learning_rate}

    def fit(self, dataset, epochs=1, callbacks=(), verbose=1, steps_per_epoch=None):
        """Minimal Keras-style loop over a Sequence-like dataset (`__len__`, `__getitem__` -> ([rhs, dx], soln)).

        Keras semantics for a custom train_step that returns plain values (models/Homogeneous_Poisson_NN_Legacy.py:291): the dict handed
        to on_epoch_end / History is the LAST batch's (tf.keras Model.fit: `epoch_logs = copy.copy(logs)`), not an epoch average - so that
        is what ReduceLROnPlateau and ModelCheckpoint monitor here too.  The epoch means are added under `loss_epoch_mean` / `mse_epoch_mean`.
        Under data parallelism the values are global (see _logs), so every rank's callbacks decide alike."""
        history = {'loss': [], 'mse': [], 'lr': [], 'loss_epoch_mean': [], 'mse_epoch_mean': []}
        self.stop_training = False
        for cb in callbacks:
            cb.set_model(self)
        fetch = _Prefetcher(dataset, self.device)             # one producer stream (and one libpcnn handle on it) per fit() call
        try:
            return self._fit_epochs(dataset, epochs, callbacks, verbose, steps_per_epoch, history, fetch)
        finally:
            fetch.close()

    def _fit_epochs(self, dataset, epochs, callbacks, verbose, steps_per_epoch, history, fetch):
        for epoch in range(epochs):
            n = steps_per_epoch if steps_per_epoch is not None else len(dataset)
            agg = {'loss': 0.0, 'mse': 0.0}
            logs = {'loss': float('nan'), 'mse': float('nan')}
            step = -1
            if n > 0:
                fetch.request(0)
            for step in range(n):
                inp, tar = fetch.take()
                logs = self.train_step((tuple(inp), tar))
                if step + 1 < n:                               # the next batch is generated (host draws + uploads + synthesis kernels on a stream of its
                    fetch.request(step + 1)                    # own) while this step's kernels run; dataset[step] is still called once per step, in order
                logs = _host_logs(logs)                        # the step's one host round trip
                for k in agg:
                    agg[k] += logs[k]
                for cb in callbacks:
                    cb.on_batch_end(step, logs)
                if self.stop_training:
                    break
            logs = {'loss': logs['loss'], 'mse': logs['mse'], 'lr': self.optimizer.learning_rate,
                    'loss_epoch_mean': agg['loss'] / max(step + 1, 1), 'mse_epoch_mean': agg['mse'] / max(step + 1, 1)}
            for k in history:
                history[k].append(logs[k])
            if verbose:
                print('Epoch %d/%d - loss: %.6g - mse: %.6g - lr: %.3g (epoch mean loss %.6g)'
                      % (epoch + 1, epochs, logs['loss'], logs['mse'], logs['lr'], logs['loss_epoch_mean']), flush=True)
            for cb in callbacks:
                cb.on_epoch_end(epoch, logs)
            if hasattr(dataset, 'on_epoch_end'):
                dataset.on_epoch_end()
            if self.stop_training:
                break
        return history


_BRANCH_STREAMS = os.environ.get('PCNN_BRANCH_STREAMS', '1') != '0'      # 0: the coarse bottleneck branches stay on the main stream


class Homogeneous_Poisson_NN_Legacy(_ModelBase):
    model_name = 'Homogeneous_Poisson_NN_Legacy'

    def __init__(self, data_format='channels_first', final_convolutions_config=None, pre_bottleneck_convolutions_config=None,
                 bottleneck_deconv_config=None, bottleneck_multilinear_config=None, input_normalization=None, output_scaling=None,
                 use_batchnorm=False, postsmoother_iterations=5, use_scaling=False, use_positional_embeddings=True, scaling_config=None,
                 gradient_accumulation_steps=None, bc_type='dirichlet', device=None, seed=0, batchnorm_training=False):
        if data_format not in ('channels_first', 'channels_last'):
            raise ValueError('data_format must be channels_first or channels_last')
        if pre_bottleneck_convolutions_config is None:
            raise ValueError('Provide a config for pre bottleneck convolutions')
        if bottleneck_deconv_config is None or bottleneck_multilinear_config is None:
            raise ValueError('Provide a config for bottleneck blocks')
        if final_convolutions_config is None:
            raise ValueError('Provide a config for final convolutions')
        if bc_type.lower() not in ('dirichlet', 'neumann'):
            raise ValueError('bc_type can only be neumann or dirichlet.')
        if device is None and not torch.cuda.is_available():
            raise RuntimeError('Homogeneous_Poisson_NN_Legacy needs an AMD GPU: the HIP kernels are the only compute path '
                               '(device="cpu" builds the parameter structure only; calling the model will raise)')
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.ndims = 2
        self.data_format = data_format
        self.gradient_accumulation_steps = gradient_accumulation_steps
        self.input_normalization = process_normalizations(input_normalization)
        self.output_scaling = process_output_scaling_modes(output_scaling)
        self.use_batchnorm = use_batchnorm
        self.use_positional_embeddings = use_positional_embeddings
        self.neumann = bc_type.lower() == 'neumann'
        self.store = S = L.ParamStore()
        # BatchNormalization mode inside train_step: False = moving statistics (what the reference's train_step most likely does,
        # SURVEY.md row H4), True = batch statistics + moving-average update (Keras training=True semantics, per replica)
        S.bn_training = bool(batchnorm_training)
        self.ctx = C = L.Context()
        C.enable_side_stream()

        # pre-bottleneck convolutions (reference :41-57)
        pre = copy.deepcopy(pre_bottleneck_convolutions_config)
        mode, val = pre.pop('padding_mode', 'CONSTANT'), pre.pop('constant_padding_value', 0.0)
        self.pre = []
        cin = 3 if use_positional_embeddings else 1
        for k in range(len(pre['filters'])):
            a = get_init_arguments_from_config(pre, k, ['filters', 'kernel_sizes'], ['filters', 'kernel_size'])
            self.pre.append(L.ConvUnit(S, C, 'pre/conv%d' % k, a['kernel_size'], cin, a['filters'], padding_mode=mode, pad_value=val,
                                       activation=a.get('activation', 'linear'), use_bias=a.get('use_bias', True),
                                       bn_name=('pre/bn%d' % k) if use_batchnorm else None))
            cin = a['filters']
        c0 = cin
        # bottleneck blocks (:59-69), sorted by descending downsampling factor
        assert bottleneck_deconv_config['filters'] == bottleneck_multilinear_config['filters']
        F = self.filters = bottleneck_deconv_config['filters']
        dfields = ['downsampling_factors', 'upsampling_factors', 'conv_kernel_sizes', 'deconv_kernel_sizes', 'n_convs']
        dargs = ['downsampling_factor', 'upsampling_factor', 'conv_kernel_size', 'deconv_kernel_size', 'n_convs']
        dcfgs = [get_init_arguments_from_config(bottleneck_deconv_config, k, dfields, dargs) for k in range(len(bottleneck_deconv_config['downsampling_factors']))]
        dcfgs = sorted(dcfgs, key=lambda a: a['downsampling_factor'], reverse=True)
        self.bottleneck_deconv_blocks = [L.bottleneck_block_deconvupsample(S, C, 'deconv_f%d' % a['downsampling_factor'], c0, use_batchnorm=use_batchnorm, **a)
                                         for a in dcfgs]
        mfields = ['downsampling_factors', 'upsampling_factors', 'conv_kernel_sizes', 'n_convs'] + (['resize_methods'] if 'resize_methods' in bottleneck_multilinear_config else [])
        margs = ['downsampling_factor', 'upsampling_factor', 'conv_kernel_size', 'n_convs'] + (['resize_method'] if 'resize_methods' in bottleneck_multilinear_config else [])
        mcfgs = [get_init_arguments_from_config(bottleneck_multilinear_config, k, mfields, margs) for k in range(len(bottleneck_multilinear_config['downsampling_factors']))]
        mcfgs = sorted(mcfgs, key=lambda a: a['downsampling_factor'], reverse=True)
        self.bottleneck_multilinear_blocks = [L.bottleneck_block_multilinearupsample(S, C, 'multilinear_f%d' % a['downsampling_factor'], c0, use_batchnorm=use_batchnorm, **a)
                                              for a in mcfgs]
        # merge + post-merge (:71-76)
        self.non_bottleneck_conv = L.ConvUnit(S, C, 'non_bottleneck_conv', 5, c0, F, pad='same', activation='leaky_relu')
        self.post_merge_conv = L.ConvUnit(S, C, 'post_merge_conv', 7, 2 * F, F, pad='same', activation='leaky_relu')
        self.post_merge_resnet = L.resnet(S, C, 'post_merge_resnet', F, 7, activation='leaky_relu')
        # final convolutions (:78-96)
        fc = copy.deepcopy(final_convolutions_config)
        nst = len(fc['filters'])
        fmode, fval = fc.pop('padding_mode', 'CONSTANT'), fc.pop('constant_padding_value', 0.0)
        nreg = fc.pop('final_regular_conv_stages', 2)
        self.final = []
        cin = F
        for k in range(nst - nreg):
            a = get_init_arguments_from_config(fc, k, ['filters', 'kernel_sizes'], ['filters', 'kernel_size'])
            act, ub = a.get('activation', 'linear'), a.get('use_bias', True)
            self.final.append(L.ConvUnit(S, C, 'final/stage%d/conv' % k, a['kernel_size'], cin, a['filters'], padding_mode=fmode, pad_value=fval,
                                         activation=act, use_bias=ub))
            self.final.append(L.resnet(S, C, 'final/stage%d/res' % k, a['filters'], a['kernel_size'], padding_mode='constant', activation=act, use_bias=ub))
            cin = a['filters']
        for j, k in enumerate(range(nst - nreg, nst)):
            self.final.append(L.ConvUnit(S, C, 'final/out%d' % j, fc['kernel_sizes'][k], cin, fc['filters'][k], pad='same', activation='linear',
                                         use_bias=fc.get('use_bias', True)))
            cin = fc['filters'][k]
        if cin != 1:
            raise ValueError('the last final convolution must have 1 filter')
        # dx dense layers (:98-102)
        units = [100, 100, F]
        acts = ['leaky_relu', 'leaky_relu', 'linear']
        self.dx_dense_layers = []
        din = 3
        for i, (u, a) in enumerate(zip(units, acts)):
            self.dx_dense_layers.append(L.Dense(S, 'dx_dense%d' % i, din, u, a))
            din = u
        self.postsmoother = L.JacobiIterationLayer(postsmoother_iterations) if postsmoother_iterations > 0 else None
        self.scaling = L.Scaling(S, C, 'scaling', **scaling_config) if use_scaling else None
        S.finalize(self.device)
        S.initialize(seed)
        self.optimizer = None
        self.loss_fn = None
        self.grad_sync = None    # set by parallel.DataParallel: called with the flat gradient bucket before the optimizer step
        self._acc = None

    COARSE_FACTOR = int(os.environ.get('PCNN_COARSE_FACTOR', '2'))     # bottleneck branches from this down-sampling factor on (default: all of them) run their
                                                                        # convolution stages on streams of their own; measured at 8 x 1024^2: none 243, >= 8: 238.9, all: 236 ms per step

    # ------------------------------------------------------------------ forward
    def _pool_pyramid(self, x, blocks):
        """The eight branches average-pool the SAME tensor by 2, 3, 4, 8, ..., 128 (blocks/bottleneck_block.py:36-41): eight full-resolution
        reads.  Where a factor f and an already pooled factor g | f both divide the image (no SAME padding), pool_f = pool_{f/g} o pool_g
        exactly (equal-weight means of equal-size blocks), so only the factors without a divisor read the full tensor.  {f: (tensor, parent g)}"""
        N, H, W, _ = x.shape
        fs = sorted({b.f for b in blocks if b.down_conv is None and b.pool in ('average', 'avg') and H % b.f == 0 and W % b.f == 0})
        if os.environ.get('PCNN_POOL_PYRAMID', '1') == '0':
            fs = []
        pyr = {}
        for f in fs:
            parent = max([g for g in pyr if f % g == 0], default=None)
            pyr[f] = (ops.pool2d_fwd(x if parent is None else pyr[parent][0], f if parent is None else f // parent, 'average'), parent)
        return pyr

    def call(self, inp, training=False):
        """reference :183-257.  inp = [rhs (N,1,H,W), dx (N,1)]; returns (N,1,H,W) (torch CUDA tensor)."""
        rhs, dx = inp
        rhs = _as_device(rhs, self.device)
        dx = _as_device(dx, self.device)
        if rhs.dim() != 4 or rhs.shape[1] != 1:
            raise ValueError('rhs must have shape (N,1,H,W)')
        dx = dx.reshape(dx.shape[0], -1)[:, :1].contiguous()
        N, _, H, W = rhs.shape
        S = self.store
        S.refresh_bn()
        rhs_hw = rhs.view(N, H, W)
        x = ops.assemble_input(rhs_hw, self.use_positional_embeddings)
        # dense input [dx, Lx, Ly] (:193,:202): tiny (N,3) host-side assembly
        dense_inp = torch.cat([dx, dx * float(H - 1), dx * float(W - 1)], 1).contiguous()
        hint = None                                               # max|x| travels with the tensor from layer to layer (weight-gradient scaling)
        for c in self.pre:
            x = c.forward(x, training=training, x_absmax=hint)
            hint = c.out_absmax
        initial = x
        F = self.filters
        cat = ops.empty((N, H, W, 2 * F), self.device)       # [non_bottleneck_conv | merged] (tf.concat axis=1, :224)
        merged = cat[..., F:]
        blocks = self.bottleneck_deconv_blocks + self.bottleneck_multilinear_blocks
        alpha = 1.0 / float(len(blocks) * F)                     # :222
        pyr = self._pool_pyramid(initial, blocks)
        # The eight branches are independent between the pooling pyramid and the merge buffer, and the coarse ones (factor >= 8: images of 128^2 and
        # below at 1024^2) are chains of launches that fill a fraction of the chip - 7.9 ms per training step back to back (tools/probe_branches.py).
        # Their convolution stages run on streams of their own (each stream has its libpcnn handle and its scratch, layers.Context), so that one
        # branch's small or tailing launches fill the gaps of another's; only the up-sampling, which accumulates into the shared merge buffer, stays on
        # the main stream.  The accumulation order is the same with and without the streams, so the result does not depend on the mode.
        coarse = [b for b in blocks if b.f >= self.COARSE_FACTOR]
        order = [b for b in blocks if b not in coarse] + coarse
        pending = {}
        self._branch_streams = {}
        if coarse and self.ctx.use_side and _BRANCH_STREAMS:
            main = torch.cuda.current_stream()
            ready = torch.cuda.Event()
            ready.record(main)
            for k, b in enumerate(coarse):
                # a branch that pools the full-resolution tensor itself (its factor does not divide the image) accumulates its input gradient
                # into d_initial in the backward pass: all such branches share stream 0, so that they do so one after the other
                st = self.ctx.branch_stream(k + 1 if b.f in pyr else 0)
                with torch.cuda.stream(st):
                    st.wait_event(ready)
                    o = b.forward_convs(initial, training, pyr[b.f][0] if b.f in pyr else None)
                    done = torch.cuda.Event()
                    done.record(st)
                pending[b] = (o, done)
                if training:
                    self._branch_streams[b] = st
        # The eight branches accumulate into the merge slice (8 x 1024^2 x 32 channels: 1.07 GB, read and written once per branch); the resize branches at the end of
        # the order are summed in ONE read-modify-write pass (ops.resize_fwd_multi: the arithmetic and order of their separate calls, bit for bit).  (Measured and
        # rejected, round 6: running the accumulation over batch chunks of <= 160 MB of merge slice, every branch per chunk, so that a chunk would stay in the 256 MB
        # Infinity Cache between its passes - the single-stream step did not move, 195.4 vs 195.2 ms.)
        ntail = 0
        while ntail < min(3, len(order) - 1) and hasattr(order[len(order) - 1 - ntail], 'method'):
            ntail += 1
        if ntail < 2:
            ntail = 0
        tail = []
        for i, b in enumerate(order):
            if b in pending:                                          # (a branch is waited for right before ITS up-sampling: the later branches' convolution
                o, done = pending[b]                                  # stages keep running on their streams under the earlier branches' up-sampling)
                torch.cuda.current_stream().wait_event(done)
                o.record_stream(torch.cuda.current_stream())
            else:
                o = b.forward_convs(initial, training, pyr[b.f][0] if b.f in pyr else None)
            b.note_upsampled(o, training)
            if i >= len(order) - ntail:
                tail.append((b, o))
            else:
                b.up_into(o, (H, W), merged, alpha, 0.0 if i == 0 else 1.0)
        if tail:
            beta0 = 0.0 if ntail == len(order) else 1.0
            if ops.resize_fwd_multi([o for _, o in tail], (H, W), [b.method for b, _ in tail], alpha=alpha, beta=beta0, out=merged) is None:
                for k, (b, o) in enumerate(tail):
                    b.up_into(o, (H, W), merged, alpha, beta0 if k == 0 else 1.0)
        self._pyr = pyr if training else None
        self.non_bottleneck_conv.forward(initial, out=cat[..., :F], training=training)
        x = self.post_merge_conv.forward(cat, training=training)
        x = self.post_merge_resnet.forward(x, training=training, x_absmax=self.post_merge_conv.out_absmax)
        d = dense_inp
        for lyr in self.dx_dense_layers:
            d = lyr.forward(d, training=training)
        xs = ops.channel_scale_fwd(x, d)                          # :231
        if training:
            self._saved = {'initial': initial, 'scale_in': x, 'dx_info': d, 'shape': (N, H, W)}
        x = xs
        hint = None
        for lyr in self.final:
            x = lyr.forward(x, training=training, x_absmax=hint)
            hint = lyr.out_absmax
        if self.scaling is not None:
            x = self.scaling.forward(x, rhs_hw.view(N, H, W, 1), training=training)
        x = ops.bc_ring_fwd(x, self.neumann)                      # :251
        if self.postsmoother is not None:
            dx2 = torch.cat([dx, dx], 1).contiguous()
            x = self.postsmoother.forward(x, rhs_hw.view(N, H, W, 1), dx2, training=training)
        return x.view(N, 1, H, W)

    # ------------------------------------------------------------------ backward
    def backward(self, dpred):
        """Back-propagates dL/dpred (N,1,H,W) through the graph saved by call(training=True); fills store.flat_g."""
        sv = self._saved
        self._saved = None
        N, H, W = sv['shape']
        F = self.filters
        d = dpred.contiguous().view(N, H, W, 1)
        if self.postsmoother is not None:
            d = self.postsmoother.backward(d)
        d = ops.bc_ring_bwd(d, self.neumann)
        if self.scaling is not None:
            d = self.scaling.backward(d)
        ready = False                                              # the gradient arriving at a layer already went through its activation backward
        for i in reversed(range(len(self.final))):                 # (fused into the data-gradient kernel of the layer after it, layers.ConvUnit.post_spec)
            post = self.final[i - 1].post_spec() if i > 0 else None
            d = self.final[i].backward(d, inplace=True, dz_ready=ready, post=post)
            ready = post is not None and post.applied
        # the einsum's input is post_merge_resnet's activation output: its adjoint also applies that layer's activation backward (one pass of a full-
        # resolution 32-channel tensor instead of four; ops.channel_scale_bwd_post declines where the offer does not fit)
        ps = self.post_merge_resnet.post_spec()
        fused = ops.channel_scale_bwd_post(sv['scale_in'], sv['dx_info'], d, ps, ws=self.ctx.ws)
        if fused is not None:
            d, ds = fused
        else:
            d, ds = ops.channel_scale_bwd(sv['scale_in'], sv['dx_info'], d, ws=self.ctx.ws)
        dd = ds
        for i, lyr in enumerate(reversed(self.dx_dense_layers)):
            dd = lyr.backward(dd, need_dx=(i < len(self.dx_dense_layers) - 1))
        post = self.post_merge_conv.post_spec()
        d = self.post_merge_resnet.backward(d, inplace=True, dz_ready=fused is not None, post=post)
        dcat = self.post_merge_conv.backward(d, inplace=True, dz_ready=post is not None and post.applied)
        d_initial = self.non_bottleneck_conv.backward(dcat[..., :F], inplace=False)
        dmerged = dcat[..., F:]
        blocks = self.bottleneck_deconv_blocks + self.bottleneck_multilinear_blocks
        alpha = 1.0 / float(len(blocks) * F)
        grads = {}
        late = []
        main = torch.cuda.current_stream()
        for b in blocks:
            st = self._branch_streams.get(b)
            if st is None:
                # a branch that pools the full-resolution tensor itself ADDS into d_initial: on the main stream that must come after every
                # stream-run branch that did the same (they share stream 0 among themselves; ADVICE r4: non-default PCNN_COARSE_FACTOR / factor orders)
                for _, g_prev, done_prev in late:
                    if g_prev is None:
                        main.wait_event(done_prev)
                g = b.backward_from(dmerged, alpha, d_initial)
                if g is not None:
                    grads[b.f] = g if b.f not in grads else ops.axpby(1.0, g, 1.0, grads[b.f])
                continue
            dco = b.backward_up(dmerged, alpha)                       # reads the shared gradient: main stream
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(st):                               # the branch's convolution stages: the stream its saved activations live on
                st.wait_event(ev)
                dco.record_stream(st)
                g = b.backward_convs(dco, d_initial)                 # pyramid-fed: returns the pooled gradient; else adds into d_initial (stream 0 only)
                done = torch.cuda.Event()
                done.record(st)
            late.append((b, g, done))
        for b, g, done in late:
            main.wait_event(done)
            if g is not None:
                g.record_stream(main)
                grads[b.f] = g if b.f not in grads else ops.axpby(1.0, g, 1.0, grads[b.f])
        self._branch_streams = {}
        for f in sorted(grads, reverse=True):                      # the pyramid's adjoint: coarsest level first, each into its parent
            parent = self._pyr[f][1]
            if parent is None:
                ops.pool2d_bwd(sv['initial'], grads[f], f, 'average', dx=d_initial, accumulate=True)
            else:
                if parent not in grads:
                    grads[parent] = torch.zeros_like(self._pyr[parent][0])
                ops.pool2d_bwd(self._pyr[parent][0], grads[f], f // parent, 'average', dx=grads[parent], accumulate=True)
        self._pyr = None
        d = d_initial
        for i, c in enumerate(reversed(self.pre)):
            d = c.backward(d, need_dx=(i < len(self.pre) - 1), inplace=True)
        self.ctx.join()                                            # weight gradients of the side stream
        self.store.finish_bn_grads()

    # ------------------------------------------------------------------ training (reference :259-296)
    def _loss_and_grads(self, rhs, dx, y_true):
        pred = self.call([rhs, dx], training=True)
        loss, dpred = self.loss_fn.value_and_grad(y_true, pred, rhs, torch.cat([dx, dx], 1))
        self.backward(dpred)
        return loss, pred

    def _dummy_batch(self, shape):
        N, H, W = shape
        g = torch.Generator(device='cpu').manual_seed(0)
        rhs = (torch.rand((N, 1, H, W), generator=g) * 2 - 1).to(self.device)
        return (rhs, torch.full((N, 1), 0.02, device=self.device)), (torch.rand((N, 1, H, W), generator=g) * 0.1).to(self.device)

    def _train_step_cf(self, data):
        loss, gt, pred = self._forward_backward(data)
        if self.grad_sync is not None:
            self.grad_sync(self.store.flat_g)
        self.optimizer.apply_gradients()
        return self._logs(loss, self.loss_fn.mse_metric(gt, pred))

    def _forward_backward(self, data):
        """forward, loss and backward of one batch (with the reference's gradient accumulation, :275-289): store.flat_g holds the gradient."""
        (rhs, dx), y_true = data
        rhs, dx, y_true = _as_device(rhs, self.device), _as_device(dx, self.device), _as_device(y_true, self.device)
        dx = dx.reshape(dx.shape[0], -1)[:, :1].contiguous()
        S = self.store
        if self.gradient_accumulation_steps is None:
            loss, pred = self._loss_and_grads(rhs, dx, y_true)
            gt = y_true
        else:
            from .utils import split_indices
            steps = int(self.gradient_accumulation_steps)
            idx = split_indices(rhs.shape[0], steps)
            if self._acc is None:
                self._acc = torch.zeros_like(S.flat_g)
            for s in range(steps):
                a, b = int(idx[s]), int(idx[s + 1])
                loss, pred = self._loss_and_grads(rhs[a:b].contiguous(), dx[a:b].contiguous(), y_true[a:b].contiguous())
                ops.axpby_flat(1.0 / steps, S.flat_g, 0.0 if s == 0 else 1.0, self._acc)     # grads = sum / steps (:287)
                gt = y_true[a:b]
            ops.axpby_flat(1.0, self._acc, 0.0, S.flat_g)
        return loss, gt, pred


# =====================================================================================================================
class Dirichlet_BC_NN_Legacy_2(_ModelBase):
    """Drop-in for poisson_CNN/models/Dirichlet_BC_NN_Legacy.py:14-187: same constructor kwargs (the "model" section of
    experiments/dbcnn.json loads unchanged), `model([bc (N,1,L), dx (N,1), x_output_resolution]) -> (N,1,X,L)`,
    `train_step(((bc, dx), y))`.  1-D tensors are NHWC (N,1,L,C) inside; Conv1D kernels are stored as (1,k,Cin,Cout)."""
    model_name = 'Dirichlet_BC_NN_Legacy_2'

    def __init__(self, data_format='channels_first', boundary_conv_config=None, spp_config=None, domain_info_mlp_config=None,
                 final_convolutions_config=None, postsmoother_iterations=0, use_batchnorm=False, device=None, seed=0, batchnorm_training=False):
        if data_format not in ('channels_first', 'channels_last'):
            raise ValueError('data_format must be channels_first or channels_last')
        if boundary_conv_config is None:
            raise ValueError('Provide a config for the boundary convolutions.')
        if spp_config is None:
            raise ValueError('Provide a config for the Spatial Pyramid Pooling.')
        if final_convolutions_config is None:
            raise ValueError('Provide a config for the domain convolutions.')
        if domain_info_mlp_config is None:
            raise ValueError('Provide a config for the domain info MLP.')
        if device is None and not torch.cuda.is_available():
            raise RuntimeError('Dirichlet_BC_NN_Legacy_2 needs an AMD GPU: the HIP kernels are the only compute path '
                               '(device="cpu" builds the parameter structure only; calling the model will raise)')
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.ndims, self.data_format, self.use_batchnorm = 2, data_format, use_batchnorm
        assert boundary_conv_config['filters'][-1] == domain_info_mlp_config['units'][-1]       # reference :39
        self.x_dir_nmodes = M = domain_info_mlp_config['units'][-1]
        if M > 27:
            import warnings
            warnings.warn('%d sinh modes chosen may lead to NaN values with float32 precision. Consider using fewer than 28 when using float32.' % M)
        self.store = S = L.ParamStore()
        S.bn_training = bool(batchnorm_training)
        self.ctx = C = L.Context()
        C.enable_side_stream()
        # boundary convolutions (:44-63): per stage conv (+BN) then a 1-D resnet
        bcc = copy.deepcopy(boundary_conv_config)
        mode, val = bcc.pop('padding_mode', 'CONSTANT'), bcc.pop('constant_padding_value', 0.0)
        self.boundary = []
        cin = 3
        for k in range(len(bcc['filters'])):
            a = get_init_arguments_from_config(bcc, k, ['filters', 'kernel_sizes'], ['filters', 'kernel_size'])
            act, ub = a.get('activation', 'linear'), a.get('use_bias', True)
            self.boundary.append(L.ConvUnit(S, C, 'bc/stage%d/conv' % k, (1, a['kernel_size']), cin, a['filters'], padding_mode=mode, pad_value=val,
                                            activation=act, use_bias=ub, bn_name=('bc/stage%d/bn' % k) if use_batchnorm else None))
            self.boundary.append(L.resnet(S, C, 'bc/stage%d/res' % k, a['filters'], (1, a['kernel_size']), use_batchnorm=use_batchnorm,
                                          padding_mode=mode, constant_padding_value=val, activation=act, use_bias=ub))
            cin = a['filters']
        # SPP (:68) + domain-info MLP (:69-74)
        self.spp_levels = [lv if isinstance(lv, int) else lv[0] for lv in spp_config['levels']]
        kind = spp_config.get('pooling_type', 'average').lower()
        if kind not in ('average', 'avg', 'max'):
            raise ValueError('spp_config pooling_type must be "average" or "max" (layers/SpatialPyramidPool.py:17-24)')
        self.spp_max = kind == 'max'
        din = 3 + sum(self.spp_levels)
        self.mlp = []
        for k, (u, a) in enumerate(zip(domain_info_mlp_config['units'], domain_info_mlp_config['activations'])):
            self.mlp.append(L.Dense(S, 'mlp/dense%d' % k, din, u, a))
            din = u
        # final convolutions (:76-97)
        fc = copy.deepcopy(final_convolutions_config)
        nst = len(fc['filters'])
        fmode, fval = fc.pop('padding_mode', 'CONSTANT'), fc.pop('constant_padding_value', 0.0)
        nreg = fc.pop('final_regular_conv_stages', 2)
        self.final = []
        cin = M + 2
        for k in range(nst - nreg):
            a = get_init_arguments_from_config(fc, k, ['filters', 'kernel_sizes'], ['filters', 'kernel_size'])
            act, ub = a.get('activation', 'linear'), a.get('use_bias', True)
            self.final.append(L.ConvUnit(S, C, 'final/stage%d/conv' % k, a['kernel_size'], cin, a['filters'], padding_mode=fmode, pad_value=fval,
                                         activation=act, use_bias=ub))
            self.final.append(L.resnet(S, C, 'final/stage%d/res' % k, a['filters'], a['kernel_size'], padding_mode='constant', activation=act, use_bias=ub))
            cin = a['filters']
        for j, k in enumerate(range(nst - nreg, nst)):
            self.final.append(L.ConvUnit(S, C, 'final/out%d' % j, fc['kernel_sizes'][k], cin, fc['filters'][k], pad='same', activation='tanh',
                                         use_bias=fc.get('use_bias', True)))
            cin = fc['filters'][k]
        if cin != 1:
            raise ValueError('the last final convolution must have 1 filter')
        self.postsmoother = L.JacobiIterationLayer(postsmoother_iterations) if postsmoother_iterations > 0 else None
        S.finalize(self.device)
        S.initialize(seed)
        self.optimizer = self.loss_fn = self.grad_sync = None
        self._bins, self._sinh = {}, {}

    def _bin_table(self, Lh):
        if Lh not in self._bins:
            from .utils import split_indices
            bins = []
            for lv in self.spp_levels:
                ix = split_indices(Lh, lv)
                if (np.diff(ix) <= 0).any():
                    raise ValueError('boundary too short for the spatial pyramid: %d bins over %d points' % (lv, Lh))
                bins += [[0, 1, ix[b], ix[b + 1]] for b in range(lv)]
            self._bins[Lh] = torch.tensor(np.array(bins, dtype=np.int32), device=self.device)
        return self._bins[Lh]

    def _sinh_table(self, X):
        """build_series_x_dir_components (:106-111): input-independent, so tabulated on the host (fp64, rounded once)."""
        if X not in self._sinh:
            xbar = np.linspace(0.0, 1.0, X)
            v = np.sinh(np.outer(np.arange(1, self.x_dir_nmodes + 1, dtype=np.float64), np.pi * (xbar - 1.0)))
            v = v / np.abs(v).max(axis=1, keepdims=True)
            self._sinh[X] = torch.from_numpy(v.astype(np.float32)).to(self.device).contiguous()
        return self._sinh[X]

    def call(self, inp, training=False):
        """reference :126-170."""
        bc, dx, X = inp
        X = int(X)
        bc, dx = _as_device(bc, self.device), _as_device(dx, self.device)
        if bc.dim() != 3 or bc.shape[1] != 1:
            raise ValueError('bc must have shape (N,1,L)')
        dx = dx.reshape(dx.shape[0], -1)[:, :1].contiguous()
        N, _, Lh = bc.shape
        S = self.store
        S.refresh_bn()
        bc2 = bc.reshape(N, Lh)
        o = ops.dbc_assemble_input(bc2)
        hint = None
        for lyr in self.boundary:
            o = lyr.forward(o, training=training, x_absmax=hint)
            hint = lyr.out_absmax
        bc_conv = o                                                            # (N,1,L,M)
        bins = self._bin_table(Lh)
        spp_arg = None
        if self.spp_max:
            feats, spp_arg = ops.spp_max_fwd(bc_conv.contiguous(), bins)
        else:
            feats = ops.spp_avg_fwd(bc_conv, bins)
        ds = torch.cat([dx * float(X - 1), dx * float(Lh - 1)], 1)           # compute_domain_sizes (:131); tiny (N,2) host-side assembly
        d = torch.cat([dx, ds / ds.amax(dim=1, keepdim=True), feats], 1).contiguous()
        for lyr in self.mlp:
            d = lyr.forward(d, training=training)
        sh = self._sinh_table(X)
        x = ops.dbc_expand_fwd(bc_conv, sh, d)                                # (N,X,L,M+2)
        hint = None
        for lyr in self.final:
            x = lyr.forward(x, training=training, x_absmax=hint)
            hint = lyr.out_absmax
        pre = x.view(N, X, Lh)
        out, _ = ops.set_max_magnitude_fwd(pre, 1.0)                          # (:163)
        ops.set_first_row(out, bc2)                                           # (:165-166)
        if training:
            self._saved = {'bc_conv': bc_conv, 'mlp_out': d, 'sinh': sh, 'pre': pre, 'bins': bins, 'spp_arg': spp_arg, 'shape': (N, X, Lh)}
        if self.postsmoother is not None:
            dx2 = torch.cat([dx, dx], 1).contiguous()
            out = self.postsmoother.forward(out.view(N, X, Lh, 1), torch.zeros_like(out).view(N, X, Lh, 1), dx2, training=training).view(N, X, Lh)
        return out.view(N, 1, X, Lh)

    def backward(self, dpred):
        sv = self._saved
        self._saved = None
        N, X, Lh = sv['shape']
        d = dpred.contiguous().view(N, X, Lh)
        if self.postsmoother is not None:
            d = self.postsmoother.backward(d.view(N, X, Lh, 1)).view(N, X, Lh)
        else:
            d = d.clone()
        ops.set_first_row(d, None)                                            # the first row is the (constant) boundary input
        d = ops.set_max_magnitude_bwd(sv['pre'], d, 1.0).view(N, X, Lh, 1)
        for lyr in reversed(self.final):
            d = lyr.backward(d, inplace=True)
        dbc_conv, dd = ops.dbc_expand_bwd(d, sv['bc_conv'], sv['sinh'], sv['mlp_out'], ws=self.ctx.ws)
        for lyr in reversed(self.mlp):
            dd = lyr.backward(dd, need_dx=True)
        dfeats = dd[:, 3:].contiguous()                                       # [dx, domain sizes] carry no parameters upstream
        dspp = (ops.spp_max_bwd(sv['spp_arg'], dfeats, tuple(sv['bc_conv'].shape)) if self.spp_max
                else ops.spp_avg_bwd(sv['bins'], dfeats, tuple(sv['bc_conv'].shape)))
        dbc_conv = ops.axpby(1.0, dspp, 1.0, dbc_conv)
        d = dbc_conv
        for i, lyr in enumerate(reversed(self.boundary)):
            last = i == len(self.boundary) - 1
            d = lyr.backward(d, need_dx=not last, inplace=True) if isinstance(lyr, L.ConvUnit) else lyr.backward(d, inplace=True)
        self.ctx.join()                                            # weight gradients of the side stream
        self.store.finish_bn_grads()

    def _train_step_cf(self, data):
        """reference :172-187: the loss sees rhs = 0 and dx repeated for both axes."""
        (bc, dx), y_true = data
        bc, dx, y_true = _as_device(bc, self.device), _as_device(dx, self.device), _as_device(y_true, self.device)
        dx = dx.reshape(dx.shape[0], -1)[:, :1].contiguous()
        pred = self.call([bc, dx, y_true.shape[2]], training=True)
        loss, dpred = self.loss_fn.value_and_grad(y_true, pred, torch.zeros_like(y_true), torch.cat([dx, dx], 1))
        self.backward(dpred)
        if self.grad_sync is not None:
            self.grad_sync(self.store.flat_g)
        self.optimizer.apply_gradients()
        return self._logs(loss, self.loss_fn.mse_metric(y_true, pred))


    # ------------------------------------------------------------------ used by Poisson_CNN_Legacy: several calls per step share the weights
    def _stateful(self):
        objs = [self]
        for lyr in self.boundary + self.final:
            objs += [lyr.c0, lyr.c1, lyr.c2] if isinstance(lyr, L.resnet) else [lyr]
        return objs + self.mlp + ([self.postsmoother] if self.postsmoother is not None else [])

    def snapshot(self):
        """What backward() needs from the last call(training=True)."""
        return [(o, {k: getattr(o, k) for k in ('saved', '_saved', 'dx2') if hasattr(o, k)}) for o in self._stateful()]

    @staticmethod
    def restore(snap):
        for o, attrs in snap:
            for k, v in attrs.items():
                setattr(o, k, v)


# =====================================================================================================================
class Poisson_CNN_Legacy(_ModelBase):
    """Drop-in for poisson_CNN/models/Poisson_CNN_Legacy.py:5-71: `Poisson_CNN_Legacy(hpnn, dbcnn)`,
    `model([rhs (N,1,H,W), left (N,1,W), top (N,1,H), right (N,1,W), bottom (N,1,H), dx (N,1)]) -> (N,1,H,W)`: the homogeneous
    solution plus one Dirichlet_BC_NN_Legacy_2 pass per edge (left/right and top/bottom are batched: same weights, same shapes),
    rotated/flipped into place by flip_and_rotate_tensor and un-normalised by the per-sample scaling factors."""
    model_name = 'Poisson_CNN_Legacy'

    def __init__(self, hpnn, dbcnn, jacobi_iterations=0):
        if jacobi_iterations > 0:
            # the reference constructor dereferences an unimported module name here (Poisson_CNN_Legacy.py:11) and raises NameError
            raise NotImplementedError('Poisson_CNN_Legacy: jacobi_iterations > 0 is unreachable in the reference (NameError at construction)')
        self.hpnn, self.dbcnn = hpnn, dbcnn
        self.device = hpnn.device
        self.data_format = getattr(hpnn, 'data_format', 'channels_first')
        self.optimizer = self.loss_fn = self.grad_sync = None

    # weights: the two sub-models' lists, hpnn first (Keras tracks attributes in assignment order)
    @property
    def stores(self):
        return [self.hpnn.store, self.dbcnn.store]

    @property
    def weight_names(self):
        return ['hpnn/' + n for n in self.hpnn.weight_names] + ['dbcnn/' + n for n in self.dbcnn.weight_names]

    @property
    def trainable_variables(self):
        return self.hpnn.trainable_variables + self.dbcnn.trainable_variables

    def get_weights(self):
        return self.hpnn.get_weights() + self.dbcnn.get_weights()

    def set_weights(self, weights):
        if isinstance(weights, dict):
            self.hpnn.set_weights({n[5:]: v for n, v in weights.items() if n.startswith('hpnn/')})
            self.dbcnn.set_weights({n[6:]: v for n, v in weights.items() if n.startswith('dbcnn/')})
        else:
            k = len(self.hpnn.weight_names)
            self.hpnn.set_weights(list(weights[:k]))
            self.dbcnn.set_weights(list(weights[k:]))

    def save_weights(self, path, save_format=None):
        if save_format in ('tf', 'tensorflow'):
            from .tf_checkpoint import save_tf_checkpoint
            return save_tf_checkpoint(self, str(path))
        np.savez(path, **{n.replace('/', '.'): w for n, w in zip(self.weight_names, self.get_weights())})

    def load_weights(self, path):
        if os.path.exists(str(path) + '.index'):
            from .tf_checkpoint import load_tf_checkpoint
            return load_tf_checkpoint(self, str(path))
        with np.load(path if str(path).endswith('.npz') else str(path) + '.npz') as z:
            self.set_weights({n: z[n.replace('/', '.')] for n in self.weight_names})

    def count_params(self):
        return self.hpnn.count_params() + self.dbcnn.count_params()

    def summary(self, print_fn=print):
        self.hpnn.summary(print_fn)
        self.dbcnn.summary(print_fn)

    # (transpose, flip rows, flip columns) that flip_and_rotate_tensor applies to each edge's result (:36-46) ...
    _PLACE = {'left': (False, False, False), 'top': (True, False, True), 'right': (False, True, False), 'bottom': (True, False, False)}
    # ... and their adjoints (a transpose swaps the roles of the two flips)
    _ADJ = {'left': (False, False, False), 'top': (True, True, False), 'right': (False, True, False), 'bottom': (True, False, False)}

    def call(self, inp, training=False):
        rhs, left, top, right, bottom, dx = [_as_device(v, self.device) for v in inp]
        dx = dx.reshape(dx.shape[0], -1)[:, :1].contiguous()
        N, _, H, W = rhs.shape
        rhs_n, rhs_f = ops.set_max_magnitude_fwd(rhs.reshape(N, H * W), 1.0)                 # :24
        edges = {}
        for name, v in (('left', left), ('top', top), ('right', right), ('bottom', bottom)):
            edges[name] = ops.set_max_magnitude_fwd(v.reshape(N, -1), 1.0)                   # :25-28
        h = self.hpnn.call([rhs_n.view(N, 1, H, W), dx], training=training)
        dmax = torch.cat([dx * float(H - 1), dx * float(W - 1)], 1).amax(dim=1)
        scale_h = (dmax * dmax / rhs_f).contiguous()                                          # :30
        pred = ops.flip_rotate(h.view(N, H, W), alpha=scale_h)
        dx2 = torch.cat([dx, dx], 0).contiguous()
        snaps, scales = {}, {}
        for pair, X in ((('left', 'right'), H), (('top', 'bottom'), W)):
            bc = torch.cat([edges[pair[0]][0], edges[pair[1]][0]], 0)
            out = self.dbcnn.call([bc.view(2 * N, 1, -1), dx2, X], training=training)        # (2N,1,X,L)
            out = out.view(2 * N, X, -1)
            for j, name in enumerate(pair):
                scales[name] = (1.0 / edges[name][1]).contiguous()                            # :33-46
                t, fy, fx = self._PLACE[name]
                ops.flip_rotate(out[j * N:(j + 1) * N], transpose=t, flip_y=fy, flip_x=fx, alpha=scales[name], out=pred, accumulate=True)
            if training:
                snaps[pair] = self.dbcnn.snapshot()
        if training:
            self._saved = {'snaps': snaps, 'scales': scales, 'scale_h': scale_h, 'shape': (N, H, W)}
        return pred.view(N, 1, H, W)

    def backward(self, dpred):
        sv = self._saved
        self._saved = None
        N, H, W = sv['shape']
        d = dpred.contiguous().view(N, H, W)
        self.hpnn.backward(ops.flip_rotate(d, alpha=sv['scale_h']).view(N, 1, H, W))
        acc = None
        for pair, X in ((('top', 'bottom'), W), (('left', 'right'), H)):
            Lh = H if X == W else W
            dout = ops.empty((2 * N, X, Lh), self.device)
            for j, name in enumerate(pair):
                t, fy, fx = self._ADJ[name]
                ops.flip_rotate(d, transpose=t, flip_y=fy, flip_x=fx, alpha=sv['scales'][name], out=dout[j * N:(j + 1) * N])
            self.dbcnn.restore(sv['snaps'][pair])
            self.dbcnn.backward(dout.view(2 * N, 1, X, Lh))
            g = self.dbcnn.store.flat_g
            if acc is None:
                acc = g.clone()
            else:
                ops.axpby_flat(1.0, acc, 1.0, g)                                              # the two passes share the weights

    def _train_step_cf(self, data):
        """reference :56-66."""
        inputs, y_true = data
        inputs = [_as_device(v, self.device) for v in inputs]
        y_true = _as_device(y_true, self.device)
        dx = inputs[5].reshape(inputs[5].shape[0], -1)[:, :1].contiguous()
        pred = self.call(inputs, training=True)
        loss, dpred = self.loss_fn.value_and_grad(y_true, pred, inputs[0], torch.cat([dx, dx], 1))
        self.backward(dpred)
        if self.grad_sync is not None:
            for s in self.stores:
                self.grad_sync(s.flat_g)
        self.optimizer.apply_gradients()
        return self._logs(loss, self.loss_fn.mse_metric(y_true, pred))
