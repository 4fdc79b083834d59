"""Host-side helpers mirroring poisson_CNN/utils (reference paths relative to poisson_CNN/)."""
import numpy as np

_ACTIVATIONS = {
    'linear': 'linear', 'tf.keras.activations.linear': 'linear', None: 'linear',
    'tf.nn.leaky_relu': 'leaky_relu', 'leaky_relu': 'leaky_relu', 'tf.keras.layers.leakyrelu': 'leaky_relu',
    'tf.nn.tanh': 'tanh', 'tf.math.tanh': 'tanh', 'tf.keras.activations.tanh': 'tanh', 'tanh': 'tanh',
    'tf.nn.relu': 'relu', 'tf.keras.activations.relu': 'relu', 'relu': 'relu',
}


_SOFTMAX = ('softmax', 'tf.nn.softmax', 'tf.keras.activations.softmax')


def canonical_activation(a, dense=False):
    """Name -> kernel activation enum name.  Replaces the reference's eval() of "tf.nn.*" strings
    (utils/convert_tf_object_names.py:13-18) with a lookup table.  'softmax' exists for Dense layers only (dense=True)."""
    key = a.lower() if isinstance(a, str) else a
    if dense and key in _SOFTMAX:
        return 'softmax'
    if key not in _ACTIVATIONS:
        raise ValueError('unsupported activation %r (supported: %s)' % (a, sorted(k for k in _ACTIVATIONS if k)))
    return _ACTIVATIONS[key]


def convert_tf_object_names(x):
    """utils/convert_tf_object_names.py:3-21.  The reference turns strings containing "tf." into TF objects with eval();
    here the strings are validated and kept as names (the kernels take enums), so a reference JSON loads unchanged."""
    if isinstance(x, list):
        return [convert_tf_object_names(i) if isinstance(i, (list, dict)) else _check(i) for i in x]
    if isinstance(x, dict):
        return {k: (convert_tf_object_names(v) if isinstance(v, (list, dict)) else _check(v)) for k, v in x.items()}
    raise ValueError('The input must be a list or dict')


def _check(item):
    if isinstance(item, str) and 'tf.' in item:
        canonical_activation(item, dense=True)   # raises for anything this build cannot map
    return item


def get_init_arguments_from_config(cfg, k, fields_in_cfg, fields_in_args):
    """models/Homogeneous_Poisson_NN_Metalearning.py:10-25."""
    out = {key: cfg[key] for key in cfg if key not in fields_in_cfg}
    out.update({a: cfg[c][k] for a, c in zip(fields_in_args, fields_in_cfg)})
    return out


def advanced_pad_amounts(k):
    """utils/apply_advanced_padding_and_call_conv_layer.py:9-10."""
    return k // 2, k // 2 - (1 - k % 2)


def same_pad_amounts(k):
    """Keras Conv2D(padding='same', strides=1)."""
    return (k - 1) // 2, k - 1 - (k - 1) // 2


def split_indices(n, sections):
    """dataset/utils/split_indices.py:4-26."""
    per, extra = divmod(int(n), int(sections))
    return np.cumsum([0] + [per + 1] * extra + [per] * (sections - extra))


def glorot_limit(shape):
    if len(shape) == 1:
        fi = fo = shape[0]
    elif len(shape) == 2:
        fi, fo = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fi, fo = shape[-2] * rf, shape[-1] * rf
    return float(np.sqrt(6.0 / (fi + fo)))


def choose_optimizer(name):
    """train/utils.py:3-8."""
    from .train import Adam, SGD
    name = name.lower()
    if name == 'adam':
        return Adam
    if name == 'sgd':
        return SGD
    raise ValueError('unknown optimizer ' + name)
