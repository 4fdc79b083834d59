"""CPU oracle: restatement of Dirichlet_BC_NN_Metalearning (models/Dirichlet_BC_NN_Metalearning.py:13-208) on the fp64 autograd twin
(oracle/torch_twin.py), composed of oracle/metalearning.py's per-sample-filter layers.

TEST INFRASTRUCTURE ONLY - see oracle/np_ops.py header.  PARITY UNPINNED (no TF here; the reference has no test, config or training script
for this model - its only user is the __main__ block at :210-256).

Followed AS WRITTEN: the stage constructors receive every non-list key of their config (get_init_arguments_from_config,
models/Homogeneous_Poisson_NN_Metalearning.py:10-25, called at :41-42,:46-47,:82-83) on top of metalearning_conv's defaults
(layers/metalearning_conv.py:53: zero padding, linear activations, dense widths [8, 16], biases, no layer norm); the 2-D stages' hyper-networks
read the `dense_inp` REBOUND at :150 ([dx, domain sizes, pooled boundary features]), the 1-D stages the one of :133.
1-D tensors (N, C, L) are carried as (N, C, 1, L).  Parameter names are those poisson_cnn_amd/dbcnn_models.py registers.
"""
import numpy as np
import torch

from . import torch_twin as T
from .dbcnn import position_embeddings, sinh_basis
from .metalearning import mconv, mresnet

def _stage_kwargs(cfg):
    """What a stage config hands to metalearning_conv beyond filters / kernel_size -> (dense activations, oracle.metalearning.mconv kwargs)."""
    units = cfg.get('pre_output_dense_units', [8, 16])
    acts = cfg.get('dense_activations', 'linear')
    acts = list(acts) if isinstance(acts, (list, tuple)) else [acts] * (len(units) + 1)
    assert len(acts) == len(units) + 1
    return acts, dict(mode=cfg.get('padding_mode', 'constant').upper(), value=cfg.get('constant_padding_value', 0.0), act=cfg.get('conv_activation', 'linear'),
                      use_bias=cfg.get('use_bias', True), use_layernorm=cfg.get('use_layernorm', False))


def _layernorm(p, name, x):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + 1e-3) * p[name + '/gamma'] + p[name + '/beta']      # tf.keras.layers.LayerNormalization(): epsilon 1e-3


def forward(cfg, p, bc, dx, X, taps=None):
    """cfg: the constructor kwargs (boundary_conv_config, spp_config, domain_info_mlp_config, final_convolutions_config, use_batchnorm,
    postsmoother_iterations).  bc (N,1,L), dx (N,2), X = x_output_resolution.  Returns (N,1,X,L)."""
    bcc, mlp, fc = cfg['boundary_conv_config'], cfg['domain_info_mlp_config'], cfg['final_convolutions_config']
    use_bn = cfg.get('use_batchnorm', False)
    bc, dx = T.asarray(bc), T.asarray(dx)
    N, _, Lh = bc.shape
    nm = bcc['filters'][-1]
    domain_sizes = dx * T.asarray(np.array([X - 1.0, Lh - 1.0]))                                       # compute_domain_sizes (:130)
    hyper_in = T.concat([dx / domain_sizes, domain_sizes / domain_sizes.max(dim=1, keepdim=True).values], 1)      # (:133-134)
    pos = position_embeddings(N, X, Lh)
    o = T.concat([bc[:, :, None, :], T.asarray(pos[:, :, 0:1, :])], 1)                                  # (N,3,1,L) (:136-141)
    cin = 3
    acts, ckw = _stage_kwargs(bcc)
    for i, (f, k) in enumerate(zip(bcc['filters'], bcc['kernel_sizes'])):                               # (:143-146)
        o = mconv(p, 'bc/stage%d/conv' % i, o, hyper_in, k, cin, f, acts, kh=1, **ckw)
        o = mresnet(p, 'bc/stage%d/res' % i, o, hyper_in, k, f, acts, use_bn, kh=1, **ckw)
        cin = f
    bc_conv = o                                                                                         # (N,M,1,L)
    sp = cfg['spp_config']
    levels = [[1, lv] if isinstance(lv, int) else [1, lv[0]] for lv in sp['levels']]
    feats = T.spatial_pyramid_pool(bc_conv, levels, 'avg' if sp.get('pooling_type', 'average').lower() in ('average', 'avg') else 'max')   # (:149)
    dense_inp = T.concat([dx, domain_sizes, feats], 1)                                                  # (:150)
    d = dense_inp
    for i, (u, a) in enumerate(zip(mlp['units'], mlp['activations'])):                                  # (:151-153; layers :62-70)
        if i != 0:
            d = _layernorm(p, 'mlp/ln%d' % i, d)
        d = T.dense(d, p['mlp/dense%d/kernel' % i], p['mlp/dense%d/bias' % i], a)
    if taps is not None:
        taps['bc_conv'], taps['mlp'], taps['dense_inp'] = bc_conv, d, dense_inp
    out = T.einsum('bmy,mx,bm->bmxy', bc_conv[:, :, 0, :], T.asarray(sinh_basis(nm, X)), d)             # (:156-159)
    out = T.concat([out, T.asarray(pos)], 1)                                                            # (:163-164)
    nreg = fc.get('final_regular_conv_stages', 2)
    nst = len(fc['filters'])
    cin = nm + 2
    acts, ckw = _stage_kwargs(fc)
    for i in range(nst - nreg):                                                                         # (:165-166)
        f, k = fc['filters'][i], fc['kernel_sizes'][i]
        out = mconv(p, 'final/stage%d/conv' % i, out, dense_inp, k, cin, f, acts, **ckw)
        out = mresnet(p, 'final/stage%d/res' % i, out, dense_inp, k, f, acts, use_bn, **ckw)
        cin = f
    for j, i in enumerate(range(nst - nreg, nst)):                                                      # (:168-169; layers :91-93)
        out = T.same_conv2d(out, p['final/out%d/kernel' % j], p['final/out%d/bias' % j] if fc['use_bias'] else None, 'tanh')
    if taps is not None:
        taps['pre_norm'] = out
    out = T.set_max_magnitude_in_batch(out, 1.0)                                                        # (:171-172)
    out = T.concat([bc[:, :, None, :], out[:, :, 1:, :]], 2)                                            # (:173-176)
    nit = cfg.get('postsmoother_iterations', 0)
    if nit > 0:
        out = T.jacobi_iterations(out, T.zeros_like(out), dx, nit)                                      # (:179-181)
    return out
