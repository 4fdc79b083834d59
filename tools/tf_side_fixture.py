"""For a third party WITH TensorFlow 2.x and the reference checkout (this repo's build container has neither): emits the two files
that would pin the TF-op arithmetic and the checkpoint interchange of SURVEY 8c / 8f-2.  Nothing of the reference is copied - it is imported.

    PYTHONPATH=/path/to/poisson_CNN python tools/tf_side_fixture.py /path/to/poisson_CNN/poisson_CNN/experiments/hpnn.json tests/golden/tf_side

writes   tests/golden/tf_side/ckpt.index + ckpt.data-00000-of-00001   model.save_weights() of Homogeneous_Poisson_NN_Legacy(**hpnn.json['model'])
         tests/golden/tf_side/forward_128.npz                         rhs (1,1,128,128), dx (1,1), out = model([rhs, dx]) - float32, seed 0
tests/test_tf_side_fixture.py consumes them when present (load_tf_checkpoint -> forward on the MI355X within 1e-5 rel-L2 of `out`) and skips otherwise."""
import json
import os
import sys

import numpy as np
import tensorflow as tf
from poisson_CNN.models import Homogeneous_Poisson_NN_Legacy

cfg_path, out_dir = sys.argv[1], sys.argv[2]
os.makedirs(out_dir, exist_ok=True)
tf.keras.backend.set_floatx('float32')
tf.random.set_seed(0)
model = Homogeneous_Poisson_NN_Legacy(**json.load(open(cfg_path))['model'])
rng = np.random.default_rng(0)
rhs = rng.uniform(-1, 1, (1, 1, 128, 128)).astype(np.float32)
dx = np.full((1, 1), 0.02, np.float32)
out = model([tf.constant(rhs), tf.constant(dx)]).numpy()       # builds the variables, then IS the fixture's expected output
model.save_weights(os.path.join(out_dir, 'ckpt'))
np.savez_compressed(os.path.join(out_dir, 'forward_128.npz'), rhs=rhs, dx=dx, out=out, tf_version=tf.__version__)
print('wrote', out_dir, 'output rms', float(np.sqrt((out ** 2).mean())))
