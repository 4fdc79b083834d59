"""Consumes the files tools/tf_side_fixture.py emits on a machine that has TensorFlow: a TensorFlow-written checkpoint of the
shipped hpnn.json model and one forward output at 128^2.  Absent files -> skipped: nothing here pins parity today, it makes pinning a five-minute
job for whoever holds TensorFlow."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIR = os.path.join(ROOT, 'tests', 'golden', 'tf_side')
have = os.path.exists(os.path.join(DIR, 'ckpt.index')) and os.path.exists(os.path.join(DIR, 'forward_128.npz'))


@pytest.mark.skipif(not have, reason='no TensorFlow-made fixture under tests/golden/tf_side (run tools/tf_side_fixture.py where TensorFlow exists)')
def test_tf_written_checkpoint_is_readable_and_names_every_variable():
    from poisson_cnn_amd import tf_checkpoint
    tensors = tf_checkpoint.read_bundle(os.path.join(DIR, 'ckpt'), verify=True)            # CRC-32C of every tensor checked
    assert sum(int(np.prod(v.shape)) for k, v in tensors.items() if k.endswith('.ATTRIBUTES/VARIABLE_VALUE')) >= 5556956


@pytest.mark.gpu
@pytest.mark.skipif(not have, reason='no TensorFlow-made fixture under tests/golden/tf_side')
def test_forward_from_tf_weights_matches_tf_output_at_128():
    """THE pinning test of the TF-op arithmetic: TensorFlow's own weights, TensorFlow's own output, north_star tolerance 1e-5 rel-L2."""
    from poisson_cnn_amd import configs, tf_checkpoint
    from poisson_cnn_amd.models import Homogeneous_Poisson_NN_Legacy
    fx = np.load(os.path.join(DIR, 'forward_128.npz'))
    model = Homogeneous_Poisson_NN_Legacy(**configs.hpnn()['model'])
    tf_checkpoint.load_tf_checkpoint(model, os.path.join(DIR, 'ckpt'))
    y = model([fx['rhs'], fx['dx']]).cpu().numpy()
    err = np.linalg.norm(y - fx['out']) / np.linalg.norm(fx['out'])
    assert err < 1e-5, err
