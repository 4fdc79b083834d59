"""Dirichlet_BC_NN_Metalearning (models/Dirichlet_BC_NN_Metalearning.py:13-208) on the libpcnn HIP kernels: the
boundary-condition network whose every convolution filter is emitted per sample by a hyper-network.

    bc (N,1,L) + [1 | cos(pi y)]  ->  1-D metalearning_conv + 1-D metalearning_resnet per stage, hyper input [dx / sizes, sizes / max]  (:43-58,:143-146)
    ->  SpatialPyramidPool (1-D)  ->  [dx, sizes, spp] -> Dense / LayerNormalization chain (:61-70,:149-153)
    ->  einsum('bmy,mx,bm->bmxy') with the normalised sinh(m pi (xbar - 1)) basis (:103-110,:159)  ->  | position embeddings
    ->  2-D metalearning_conv + metalearning_resnet stages, then plain tanh convolutions (:72-93,:163-169)
    ->  set_max_magnitude_in_batch, first row := bc, optional Jacobi post-smoother (:171-181)

Built AS WRITTEN: every key of boundary_conv_config / final_convolutions_config other than the per-stage lists is handed to each stage's
metalearning_conv and metalearning_resnet (get_init_arguments_from_config, :41-42,:46-47) - padding_mode, conv_activation, dense_activations,
pre_output_dense_units, use_layernorm, use_bias ... as in the reference's own __main__ configs (:219-250) - and `dense_inp` is rebound at :150 before
the 2-D stages, so their hyper-networks read [dx, sizes, spp(boundary features)] - the gradient of every 2-D filter flows back into the boundary
convolutions through the pyramid pooling - while the 1-D stages read [dx / sizes, sizes / max].
BatchNormalization runs on its moving statistics (the train_step calls the model without training=True, :197).  There is no TF here and the
reference has no test or config for this model: PARITY UNPINNED, checked against the fp64 autograd restatement in oracle/dbcnn_metalearning.py.

    model([bc (N,1,L), dx (N,2), x_output_resolution]) -> (N,1,X,L);   compile(loss, optimizer);   train_step(((bc, dx (N,1)), y))
"""
import copy

import numpy as np
import torch

from . import layers as L
from . import metalearning as M
from . import ops
from .models import _ModelBase, _as_device
from .utils import get_init_arguments_from_config, split_indices

_CONV_FIELDS = (['filters', 'kernel_sizes'], ['filters', 'kernel_size'])


class Dirichlet_BC_NN_Metalearning(_ModelBase):
    model_name = 'Dirichlet_BC_NN_Metalearning'

    def __init__(self, ndims=2, data_format='channels_first', boundary_conv_config=None, spp_config=None, domain_info_mlp_config=None,
                 final_convolutions_config=None, postsmoother_iterations=0, use_batchnorm=False, device=None, seed=0):
        if ndims != 2:
            raise NotImplementedError('ndims = 2 only (1-D boundaries of 2-D domains)')
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
            raise RuntimeError('Dirichlet_BC_NN_Metalearning needs an AMD GPU: the HIP kernels are the only compute path')
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.ndims, self.data_format, self.use_batchnorm = 2, data_format, use_batchnorm
        assert boundary_conv_config['filters'][-1] == domain_info_mlp_config['units'][-1]       # reference :33
        self.x_dir_nmodes = nm = int(boundary_conv_config['filters'][-1])
        if nm > 27:
            import warnings
            warnings.warn('%d sinh modes chosen may lead to NaN values with float32 precision. Consider using fewer than 28 when using float32.' % nm)
        self.store = S = L.ParamStore()
        self.ctx = C = L.Context()
        # (ndims-1)-dimensional convolutions on the BC info (:40-58); hyper input = dense_inp of :133 (4 features)
        self.boundary = []
        cin = 3
        for k in range(len(boundary_conv_config['filters'])):
            a = get_init_arguments_from_config(boundary_conv_config, k, *_CONV_FIELDS)
            self.boundary.append(M.metalearning_conv(dimensions=1, padding='same', previous_layer_filters=cin, dense_input_features=4, store=S, ctx=C,
                                                     name='bc/stage%d/conv' % k, **a))
            cin = int(a['filters'])
            self.boundary.append(M.metalearning_resnet(dimensions=1, use_batchnorm=use_batchnorm, previous_layer_filters=cin, dense_input_features=4,
                                                       store=S, ctx=C, name='bc/stage%d/res' % k, **a))
        # SPP (:61) + the domain-info dense chain, a LayerNormalization before every Dense but the first (:62-70)
        self.spp_levels = [lv if isinstance(lv, int) else lv[0] for lv in spp_config['levels']]
        kind = spp_config.get('pooling_type', 'average').lower()                            # layers/SpatialPyramidPool.py:6 default
        if kind not in ('average', 'avg', 'max'):
            raise ValueError('spp_config pooling_type must be "average" or "max" (layers/SpatialPyramidPool.py:17-24)')
        self.spp_max = kind == 'max'
        self.dense_features = din = 4 + sum(self.spp_levels)                  # every bin pools over the channels too (layers/SpatialPyramidPool.py:43-44)
        self.mlp = []
        for k in range(len(domain_info_mlp_config['units'])):
            a = get_init_arguments_from_config(domain_info_mlp_config, k, ['units', 'activations'], ['units', 'activation'])
            if k != 0:
                self.mlp.append(L.LayerNormalization(S, 'mlp/ln%d' % k, din))
            self.mlp.append(L.Dense(S, 'mlp/dense%d' % k, din, int(a['units']), a['activation']))
            din = int(a['units'])
        # convolutions on the assembled tensor (:72-93); hyper input = the REBOUND dense_inp of :150
        fc = copy.deepcopy(final_convolutions_config)
        nst = len(fc['filters'])
        self.final_regular_conv_stages = nreg = fc.pop('final_regular_conv_stages', 2)
        self.final_meta, self.final_regular = [], []
        cin = nm + 2
        for k in range(nst - nreg):
            a = get_init_arguments_from_config(fc, k, *_CONV_FIELDS)
            self.final_meta.append(M.metalearning_conv(dimensions=2, padding='same', previous_layer_filters=cin, dense_input_features=self.dense_features,
                                                       store=S, ctx=C, name='final/stage%d/conv' % k, **a))
            cin = int(a['filters'])
            self.final_meta.append(M.metalearning_resnet(dimensions=2, use_batchnorm=use_batchnorm, previous_layer_filters=cin,
                                                         dense_input_features=self.dense_features, store=S, ctx=C, name='final/stage%d/res' % k, **a))
        for j, k in enumerate(range(nst - nreg, nst)):
            self.final_regular.append(L.ConvUnit(S, C, 'final/out%d' % j, fc['kernel_sizes'][k], cin, fc['filters'][k], pad='same', activation='tanh',
                                                 use_bias=fc['use_bias']))
            cin = int(fc['filters'][k])
        if cin != 1:
            raise ValueError('the last final convolution must have 1 filter')
        self.postsmoother = L.JacobiIterationLayer(postsmoother_iterations) if postsmoother_iterations > 0 else None
        S.finalize(self.device)
        S.initialize(seed)
        self.optimizer = self.loss_fn = self.grad_sync = None
        self._bins, self._sinh = {}, {}

    def _bin_table(self, Lh):
        if Lh not in self._bins:
            bins = []
            for lv in self.spp_levels:
                ix = split_indices(Lh, lv)
                if (np.diff(ix) <= 0).any():
                    raise ValueError('boundary too short for the spatial pyramid: %d bins over %d points' % (lv, Lh))
                bins += [[0, 1, ix[b], ix[b + 1]] for b in range(lv)]
            self._bins[Lh] = torch.tensor(np.array(bins, dtype=np.int32), device=self.device)
        return self._bins[Lh]

    def _sinh_table(self, X):
        """build_series_x_dir_components (:103-110): input-independent, tabulated on the host in fp64 and rounded once."""
        if X not in self._sinh:
            xbar = np.linspace(0.0, 1.0, X)
            v = np.sinh(np.outer(np.arange(1, self.x_dir_nmodes + 1, dtype=np.float64), np.pi * (xbar - 1.0)))
            v = v / np.abs(v).max(axis=1, keepdims=True)
            self._sinh[X] = torch.from_numpy(v.astype(np.float32)).to(self.device).contiguous()
        return self._sinh[X]

    def call(self, inp, training=False):
        """reference :124-183."""
        bc, dx, X = inp
        X = int(X)
        bc, dx = _as_device(bc, self.device), _as_device(dx, self.device)
        if bc.dim() != 3 or bc.shape[1] != 1:
            raise ValueError('bc must have shape (N,1,L)')
        N, _, Lh = bc.shape
        dx = dx.reshape(N, -1)
        if dx.shape[1] != 2:
            raise ValueError('dx must have shape (N,2) (the train step tiles the (N,1) grid spacing, reference :189)')
        self.store.refresh_bn()
        # tiny (N,2) / (N,4) host-side assemblies (:128-134)
        domain_sizes = dx * torch.tensor([float(X - 1), float(Lh - 1)], device=self.device)
        hyper_in = torch.cat([dx / domain_sizes, domain_sizes / domain_sizes.amax(dim=1, keepdim=True)], 1).contiguous()
        bc2 = bc.reshape(N, Lh)
        o = ops.dbc_assemble_input(bc2)                                               # (N,1,L,3) = [bc | 1 | cos(pi y)] (:136-141)
        for lyr in self.boundary:
            o = lyr.forward(o, hyper_in, training)
        bc_conv = o                                                                   # (N,1,L,M)
        bins = self._bin_table(Lh)
        spp_arg = None
        if self.spp_max:
            feats, spp_arg = ops.spp_max_fwd(bc_conv.contiguous(), bins)
        else:
            feats = ops.spp_avg_fwd(bc_conv, bins)
        dense_inp = torch.cat([dx, domain_sizes, feats], 1).contiguous()              # (:150)
        d = dense_inp
        for lyr in self.mlp:
            d = lyr.forward(d, training)
        sh = self._sinh_table(X)
        x = ops.dbc_expand_fwd(bc_conv, sh, d)                                        # (N,X,L,M+2): the series | position embeddings (:159-164)
        for lyr in self.final_meta:
            x = lyr.forward(x, dense_inp, training)
        for lyr in self.final_regular:
            x = lyr.forward(x, training=training)
        pre = x.view(N, X, Lh)
        out, _ = ops.set_max_magnitude_fwd(pre, 1.0)                                  # (:171-172)
        ops.set_first_row(out, bc2)                                                   # (:173-176)
        if training:
            self._saved = {'bc_conv': bc_conv, 'mlp_out': d, 'sinh': sh, 'pre': pre, 'bins': bins, 'spp_arg': spp_arg, 'shape': (N, X, Lh)}
        if self.postsmoother is not None:
            out = self.postsmoother.forward(out.view(N, X, Lh, 1), torch.zeros_like(out).view(N, X, Lh, 1), dx.contiguous(), training=training).view(N, X, Lh)
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
        ops.set_first_row(d, None)                                                    # the first row is the (constant) boundary input
        d = ops.set_max_magnitude_bwd(sv['pre'], d, 1.0).view(N, X, Lh, 1)
        for lyr in reversed(self.final_regular):
            d = lyr.backward(d, inplace=True)
        d_dense = None                                                                # gradient at the rebound dense_inp: every 2-D hyper-network + the MLP
        for lyr in reversed(self.final_meta):
            d, dd = lyr.backward(d)
            d_dense = dd if d_dense is None else ops.axpby_flat(1.0, dd, 1.0, d_dense)
        dbc_conv, dd = ops.dbc_expand_bwd(d, sv['bc_conv'], sv['sinh'], sv['mlp_out'], ws=self.ctx.ws)
        for lyr in reversed(self.mlp):
            dd = lyr.backward(dd, need_dx=True)
        d_dense = dd if d_dense is None else ops.axpby_flat(1.0, dd, 1.0, d_dense)
        dfeats = d_dense[:, 4:].contiguous()                                          # [dx, domain sizes] are inputs
        dspp = (ops.spp_max_bwd(sv['spp_arg'], dfeats, tuple(sv['bc_conv'].shape)) if self.spp_max
                else ops.spp_avg_bwd(sv['bins'], dfeats, tuple(sv['bc_conv'].shape)))
        d = ops.axpby(1.0, dspp, 1.0, dbc_conv)
        for i, lyr in enumerate(reversed(self.boundary)):
            d, _ = lyr.backward(d, need_dx=(i < len(self.boundary) - 1))              # the 1-D hyper input is a function of dx and the shape only
        self.ctx.join()
        self.store.finish_bn_grads()

    def _train_step_cf(self, data):
        """reference :185-205: dx tiled to both axes, the loss sees rhs = 0."""
        (bc, dx), y_true = data
        bc, dx, y_true = _as_device(bc, self.device), _as_device(dx, self.device), _as_device(y_true, self.device)
        dx = dx.reshape(dx.shape[0], -1)
        if dx.shape[1] == 1:
            dx = dx.repeat(1, 2)                                                      # tf.tile(dx, [1, ndims]) (:192)
        dx = dx.contiguous()
        pred = self.call([bc, dx, y_true.shape[2]], training=True)
        loss, dpred = self.loss_fn.value_and_grad(y_true, pred, torch.zeros_like(y_true), dx)
        self.backward(dpred)
        if self.grad_sync is not None:
            self.grad_sync(self.store.flat_g)
        self.optimizer.apply_gradients()
        return self._logs(loss, self.loss_fn.mse_metric(y_true, pred))
