"""Optimizers, callbacks and the training entry point (drop-in for poisson_CNN/train/hpnn_legacy_train.py and train/utils.py).

python -m poisson_cnn_amd.train config.json [--checkpoint_dir D] [--continue_from_checkpoint D] [--dataset_type analytical|numerical]
                                            [--learning_rate X|from_json]
Under `torchrun` (WORLD_SIZE > 1) the batch is sharded over ranks and gradients are all-reduced over RCCL
(parallel.DataParallel), replacing the reference's tf.distribute.MirroredStrategy (train/hpnn_legacy_train.py:37-38).
"""
import argparse
import json
import math
import os

import numpy as np
import torch

from . import ops


class _Optimizer:
    def bind(self, store):
        """store: one ParamStore or a list of them (a composite model such as Poisson_CNN_Legacy trains several buckets)."""
        self.stores = list(store) if isinstance(store, (list, tuple)) else [store]
        self.store = self.stores[0]
        self._init_state()

    def _init_state(self):
        pass


def _reject_unsupported_optimizer_kwargs(who, kwargs):
    """tf.keras optimizers also take clipnorm / clipvalue / decay (/ lr as an alias): options that change the update must not be
    swallowed silently - a config that sets them would train differently from the reference."""
    for k, v in kwargs.items():
        if k in ('clipnorm', 'clipvalue', 'global_clipnorm') and v is None:
            continue
        if k == 'decay' and not v:
            continue
        raise NotImplementedError('%s: optimizer option %r=%r is not implemented' % (who, k, v))


class Adam(_Optimizer):
    """tf.keras.optimizers.Adam defaults (train/utils.py:3-8; experiments/hpnn.json optimizer_parameters)."""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False, name='Adam', **kwargs):
        _reject_unsupported_optimizer_kwargs('Adam', kwargs)
        self.amsgrad = bool(amsgrad)
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = float(learning_rate), float(beta_1), float(beta_2), float(epsilon)
        self.iterations = 0

    def _init_state(self):
        import torch
        self.ms = [torch.zeros_like(s.flat_w) for s in self.stores]
        self.vs = [torch.zeros_like(s.flat_w) for s in self.stores]
        self.vhats = [torch.zeros_like(s.flat_w) if self.amsgrad else None for s in self.stores]
        self.m, self.v = self.ms[0], self.vs[0]

    def apply_gradients(self, grad_scale=1.0):
        self.iterations += 1
        for s, m, v, vh in zip(self.stores, self.ms, self.vs, self.vhats):
            ops.adam_step(s.flat_w, s.flat_g, m, v, self.learning_rate, self.beta_1, self.beta_2, self.epsilon, self.iterations, grad_scale, vhat=vh)


class SGD(_Optimizer):
    def __init__(self, learning_rate=0.01, momentum=0.0, nesterov=False, name='SGD', **kwargs):
        _reject_unsupported_optimizer_kwargs('SGD', kwargs)
        self.learning_rate, self.momentum, self.nesterov = float(learning_rate), float(momentum), bool(nesterov)
        self.iterations = 0

    def _init_state(self):
        self.vs = [torch.zeros_like(s.flat_w) for s in self.stores] if self.momentum != 0.0 else None

    def apply_gradients(self, grad_scale=1.0):
        self.iterations += 1
        for k, s in enumerate(self.stores):
            if self.momentum != 0.0:
                ops.sgd_momentum_step(s.flat_w, s.flat_g, self.vs[k], self.learning_rate, self.momentum, self.nesterov, grad_scale)
            else:
                ops.sgd_step(s.flat_w, s.flat_g, self.learning_rate, grad_scale)


def choose_optimizer(name):
    """train/utils.py:3-8."""
    name = name.lower()
    if name == 'adam':
        return Adam
    if name == 'sgd':
        return SGD
    raise ValueError('unknown optimizer ' + name)


# ----------------------------------------------------------------------------- Keras-like callbacks (train/hpnn_legacy_train.py:46-50)
class Callback:
    def set_model(self, model):
        self.model = model

    def on_batch_end(self, batch, logs):
        pass

    def on_epoch_end(self, epoch, logs):
        pass


class ModelCheckpoint(Callback):
    def __init__(self, filepath, save_weights_only=True, save_best_only=True, monitor='loss', save_format=None):
        self.filepath, self.best, self.monitor, self.save_best_only = filepath, math.inf, monitor, save_best_only
        self.save_format = save_format                       # None: flat .npz; 'tf': TensorFlow checkpoint files (tf_checkpoint.py)

    def on_epoch_end(self, epoch, logs):
        v = logs[self.monitor]
        if not self.save_best_only or v < self.best:
            self.best = min(self.best, v)
            if int(os.environ.get('RANK', '0')) == 0:
                self.model.save_weights(self.filepath, **({'save_format': self.save_format} if self.save_format else {}))


class ReduceLROnPlateau(Callback):
    """tf.keras.callbacks.ReduceLROnPlateau, mode 'min' (train/hpnn_legacy_train.py:48 passes patience and min_lr; everything else is
    the Keras default: factor 0.1, min_delta 1e-4, cooldown 0).  An epoch counts as an improvement only if monitor < best - min_delta.
    The monitored value is what fit() hands to on_epoch_end: the last batch's (global) loss, as in Keras."""

    def __init__(self, monitor='loss', factor=0.1, patience=10, verbose=0, mode='auto', min_delta=1e-4, cooldown=0, min_lr=0.0):
        if factor >= 1.0:
            raise ValueError('ReduceLROnPlateau does not support a factor >= 1.0.')
        if mode not in ('auto', 'min'):
            raise NotImplementedError("ReduceLROnPlateau: only mode 'min' / 'auto' on a loss is implemented")
        self.monitor, self.factor, self.patience, self.verbose = monitor, factor, patience, verbose
        self.min_delta, self.cooldown, self.min_lr = min_delta, cooldown, min_lr
        self.best, self.wait, self.cooldown_counter = math.inf, 0, 0

    def on_epoch_end(self, epoch, logs):
        current = logs[self.monitor]
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.wait = 0
        if current < self.best - self.min_delta:
            self.best, self.wait = current, 0
        elif self.cooldown_counter <= 0:
            self.wait += 1
            if self.wait >= self.patience:
                opt = self.model.optimizer
                if opt.learning_rate > self.min_lr:
                    opt.learning_rate = max(opt.learning_rate * self.factor, self.min_lr)
                    if self.verbose:
                        print('Epoch %d: ReduceLROnPlateau reducing learning rate to %g.' % (epoch + 1, opt.learning_rate))
                    self.cooldown_counter = self.cooldown
                    self.wait = 0


class TerminateOnNaN(Callback):
    def on_batch_end(self, batch, logs):
        if not math.isfinite(logs['loss']):
            print('Batch %d: Invalid loss, terminating training' % batch)
            self.model.stop_training = True


def latest_checkpoint(checkpoint_dir):
    """tf.train.latest_checkpoint: the prefix named by `model_checkpoint_path` in <dir>/checkpoint, or None."""
    state = os.path.join(checkpoint_dir, 'checkpoint')
    if os.path.exists(state):
        for line in open(state):
            if line.startswith('model_checkpoint_path:'):
                name = line.split(':', 1)[1].strip().strip('"')
                return name if os.path.isabs(name) else os.path.join(checkpoint_dir, name)
    return None


def load_model_checkpoint(model, checkpoint_path, **unused):
    """train/utils.py:10-29.  A directory holding TensorFlow checkpoint files (a `checkpoint` state file naming the prefix, as Keras'
    ModelCheckpoint leaves it) is read through tf_checkpoint.py; otherwise the flat .npz this package writes by default."""
    if checkpoint_path is not None:
        path = checkpoint_path
        if os.path.isdir(path):
            path = latest_checkpoint(path) or os.path.join(path, 'chkpt.checkpoint.npz')
        print('Attempting to load checkpoint from ' + path)
        model.load_weights(path)


def main(argv=None):
    from . import configs
    from .utils import convert_tf_object_names
    from .models import Homogeneous_Poisson_NN_Legacy, Dirichlet_BC_NN_Legacy_2, Poisson_CNN_Legacy
    from .losses import loss_wrapper
    from .dataset import numerical_dataset_generator, reverse_poisson_dataset_generator, reverse_poisson_dataset_generator_homogeneous_neumann
    from . import parallel
    p = argparse.ArgumentParser(description='Train the Homogeneous Poisson NN')
    p.add_argument('config', type=str)
    p.add_argument('--checkpoint_dir', type=str, default='.')
    p.add_argument('--continue_from_checkpoint', type=str, default=None)
    p.add_argument('--dataset_type', type=lambda x: str(x).lower(), default='analytical')
    p.add_argument('--learning_rate', type=str, default=None)
    p.add_argument('--epochs', type=int, default=None)
    p.add_argument('--model', type=lambda x: str(x).lower(), default='hpnn', choices=['hpnn', 'dbcnn', 'pcnn'],
                   help='hpnn: train/hpnn_legacy_train.py (train/hpnn_train.py when the model section carries model_type); dbcnn: train/dbcnn_legacy_train.py; '
                        'pcnn: train/pcnn_end_to_end.py')
    args = p.parse_args(argv)
    if args.dataset_type not in ('numerical', 'analytical'):
        raise ValueError('Invalid dataset type. Received: ' + args.dataset_type)
    config = convert_tf_object_names(configs.load_config(args.config))
    if config['training'].get('precision', 'float32') != 'float32':
        raise NotImplementedError('the HIP kernels compute in float32')
    dp = parallel.DataParallel.from_env()
    gbs = config['dataset']['batch_size']
    dcfg = dict(config['dataset'])
    dcfg['batch_size'] = dp.local_batch(gbs)
    if dp.world_size > 1:       # every rank draws the step's grid shape from one shared stream and its samples from its own (dataset._streams)
        dcfg['shard'] = (dp.rank, dp.world_size)
    if args.model == 'dbcnn':      # train/dbcnn_legacy_train.py:26-31: one non-zero edge, zero right-hand side
        dataset = numerical_dataset_generator(randomize_boundary_smoothness=True, exclude_zero_boundaries=True, nonzero_boundaries=['left'], rhses='zero',
                                              return_boundaries=True, return_dx=True, return_rhs=False, **dcfg)
        model = Dirichlet_BC_NN_Legacy_2(**config['model'])
    elif args.model == 'pcnn':     # train/pcnn_end_to_end.py:28-34: all four edges + a random right-hand side, both sub-models trained jointly
        dataset = numerical_dataset_generator(randomize_boundary_smoothness=True, exclude_zero_boundaries=False, nonzero_boundaries=['left', 'right', 'top', 'bottom'],
                                              rhses='random', return_boundaries=True, return_dx=True, return_rhs=True, **dcfg)
        model = Poisson_CNN_Legacy(Homogeneous_Poisson_NN_Legacy(**config['hpnn_model']), Dirichlet_BC_NN_Legacy_2(**config['dbcnn_model']))
    elif 'model_type' in config['model']:      # train/hpnn_train.py:23-33: the config names the model class; analytic (reverse) dataset
        from .hpnn_models import Homogeneous_Poisson_NN, Homogeneous_Poisson_NN_Metalearning
        mcfg = dict(config['model'])
        model_type = mcfg.pop('model_type')
        classes = {'cnn_metalearning': Homogeneous_Poisson_NN_Metalearning, 'cnn': Homogeneous_Poisson_NN}
        if model_type not in classes:
            raise NotImplementedError('model_type %r (built: %s)' % (model_type, sorted(classes)))
        dataset = reverse_poisson_dataset_generator(**dcfg)
        model = classes[model_type](**mcfg)
    else:
        neumann = config['model'].get('bc_type', 'dirichlet').lower() == 'neumann'
        if args.dataset_type == 'numerical':
            dataset = numerical_dataset_generator(**dcfg)
        elif neumann:
            dataset = reverse_poisson_dataset_generator_homogeneous_neumann(**dcfg)
        else:
            dataset = reverse_poisson_dataset_generator(**dcfg)
        model = Homogeneous_Poisson_NN_Legacy(**config['model'])
    optimizer = choose_optimizer(config['training']['optimizer'])(**config['training']['optimizer_parameters'])
    loss = loss_wrapper(global_batch_size=gbs, **config['training']['loss_parameters'])
    # the largest batch this run will see, for model.presize(): only where the generator draws its grid shape from a range (the analytic generators)
    rng_ = config['dataset'].get('random_output_shape_range')
    presize = None
    if rng_ is not None and isinstance(model, Homogeneous_Poisson_NN_Legacy) and os.environ.get('PCNN_PRESIZE', '1') != '0':
        r = np.asarray(rng_, dtype=np.int64)
        r = np.tile(r[None], (2, 1)) if r.ndim == 1 else r
        presize = (dcfg['batch_size'], int(r[0, 1]), int(r[1, 1]))
    model.compile(loss=loss, optimizer=optimizer, max_input_shape=presize)
    dp.attach(model)
    cb = [ModelCheckpoint(args.checkpoint_dir + '/chkpt.checkpoint'), ReduceLROnPlateau(patience=4, min_lr=config['training']['min_learning_rate']),
          TerminateOnNaN()]
    load_model_checkpoint(model, args.continue_from_checkpoint)
    if args.learning_rate is not None:
        model.optimizer.learning_rate = config['training']['optimizer_parameters']['learning_rate'] if args.learning_rate.lower() == 'from_json' else float(args.learning_rate)
    if dp.rank == 0:
        model.summary()
    model.fit(dataset, epochs=args.epochs or config['training']['n_epochs'], callbacks=cb, verbose=1 if dp.rank == 0 else 0)


if __name__ == '__main__':
    main()
