"""TensorFlow checkpoint (TensorBundle V2) interchange, tf_checkpoint.py: the format's fixed points and the round trip.  No TensorFlow exists
here, so cross-reading with TensorFlow itself is NOT covered (the module header says "parity unpinned")."""
import os
import struct

import numpy as np
import pytest

from poisson_cnn_amd import tf_checkpoint as T


def test_crc32c_known_answers_and_mask():
    # RFC 3720 B.4 vectors
    assert T.crc32c(b'123456789') == 0xE3069283
    assert T.crc32c(bytes(32)) == 0x8A9136AA
    assert T.crc32c(bytes([0xff] * 32)) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    # running form: Extend(Extend(0, a), b) == Value(a + b)
    assert T.crc32c(b'6789', T.crc32c(b'12345')) == 0xE3069283
    for v in (0, 1, 0xE3069283, 0xffffffff):
        assert T.unmask(T.mask(v)) == v
    assert T.mask(0xE3069283) == (((0xE3069283 >> 15) | (0xE3069283 << 17)) + 0xa282ead8) & 0xffffffff


def test_bundle_round_trip_and_layout(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {'a/b/kernel' + T.SUFFIX: rng.standard_normal((3, 3, 2, 4)).astype(np.float32),
               'a/b/bias' + T.SUFFIX: rng.standard_normal(4).astype(np.float32),
               'a/c/0/gamma' + T.SUFFIX: rng.standard_normal(7).astype(np.float64),
               'step' + T.SUFFIX: np.array(12345, dtype=np.int64)}
    prefix = str(tmp_path / 'chkpt.checkpoint')
    T.write_bundle(prefix, tensors)
    assert sorted(os.listdir(tmp_path)) == ['checkpoint', 'chkpt.checkpoint.data-00000-of-00001', 'chkpt.checkpoint.index']
    idx = open(prefix + '.index', 'rb').read()
    assert struct.unpack('<Q', idx[-8:])[0] == 0xdb4775248b80fb57                     # table magic
    got = T.read_bundle(prefix)
    assert sorted(got) == sorted(tensors)
    for k in tensors:
        assert got[k].dtype == tensors[k].dtype and got[k].shape == tensors[k].shape
        np.testing.assert_array_equal(got[k], tensors[k])
    # the data shard is the raw little-endian bytes back to back in key order (object graph included)
    keys = sorted(list(tensors) + [T.OBJECT_GRAPH_KEY])
    data = open(prefix + '.data-00000-of-00001', 'rb').read()
    off = 0
    for k in keys:
        if k != T.OBJECT_GRAPH_KEY:
            raw = tensors[k].tobytes()
            assert data[off:off + len(raw)] == raw
            off += len(raw)
        else:
            n, p = T._read_varint(data, off)
            off = p + 4 + n
    assert off == len(data)
    # header entry: key "", num_shards 1, version.producer 1
    items = T._read_table(prefix + '.index')
    assert items[0][0] == b'' and [k for k, _ in items] == sorted(k for k, _ in items)
    hdr = {n: v for n, _, v in T._parse(items[0][1])}
    assert hdr[1] == 1 and T._parse(hdr[3]) == [(1, 0, 1)]
    # state file
    assert 'model_checkpoint_path: "chkpt.checkpoint"' in open(tmp_path / 'checkpoint').read()


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / 'c')
    T.write_bundle(prefix, {'w' + T.SUFFIX: np.arange(100, dtype=np.float32)})
    raw = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    raw[-17] ^= 0x40                                                # inside the float tensor (the object graph sorts first)
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(raw))
    with pytest.raises(ValueError, match='checksum mismatch'):
        T.read_bundle(prefix)
    assert T.read_bundle(prefix, verify=False)['w' + T.SUFFIX].shape == (100,)
    idx = bytearray(open(prefix + '.index', 'rb').read())
    idx[3] ^= 0x01
    open(prefix + '.index', 'wb').write(bytes(idx))
    with pytest.raises(ValueError, match='block checksum'):
        T.read_bundle(prefix)
    open(prefix + '.index', 'wb').write(b'not a table' * 10)
    with pytest.raises(ValueError, match='bad magic'):
        T.read_bundle(prefix)


def test_many_keys_span_several_table_blocks(tmp_path):
    tensors = {'layer/%04d/kernel%s' % (i, T.SUFFIX): np.full((i % 5 + 1,), i, dtype=np.float32) for i in range(700)}
    prefix = str(tmp_path / 'big')
    T.write_bundle(prefix, tensors)
    # force small blocks as well: restart points + prefix compression + several data blocks
    items = [(k.encode(), b'v' * 10) for k in sorted(tensors)]
    T._write_table(str(tmp_path / 't.index'), items, block_size=512)
    assert T._read_table(str(tmp_path / 't.index')) == items
    got = T.read_bundle(prefix)
    assert len(got) == 700 and all(np.array_equal(got[k], tensors[k]) for k in tensors)


def test_object_graph_names_every_variable():
    keys = ['a/b/kernel' + T.SUFFIX, 'a/b/bias' + T.SUFFIX, 'a/c/0/gamma' + T.SUFFIX]
    g = T._object_graph(keys)
    nodes = [v for n, _, v in T._parse(g) if n == 1]
    # root -> a -> {b -> {kernel, bias}, c -> 0 -> gamma}: 1 + 1 + 2 + 2 + 1 + 1 nodes
    assert len(nodes) == 8
    root_children = [dict((n, v) for n, _, v in T._parse(c))[2] for n, _, c in T._parse(nodes[0]) if n == 1]
    assert root_children == [b'a']
    found = []
    for nd in nodes:
        for n, _, v in T._parse(nd):
            if n == 2:
                a = {k: val for k, _, val in T._parse(v)}
                assert a[1] == b'VARIABLE_VALUE'
                found.append(a[3].decode())
    assert sorted(found) == sorted(keys)


def test_hpnn_checkpoint_uses_the_reference_object_paths(tmp_path):
    """Variables land at the attribute paths of models/Homogeneous_Poisson_NN_Legacy.py:41-115 and come back bit-exact through
    save_weights(save_format='tf') / load_weights / load_model_checkpoint(directory)."""
    from poisson_cnn_amd import configs, train
    from poisson_cnn_amd.models import Homogeneous_Poisson_NN_Legacy
    cfg = configs.hpnn()['model']
    m = Homogeneous_Poisson_NN_Legacy(**cfg, device='cpu', seed=1)
    paths = T.keras_object_paths(m)
    assert paths['pre/conv1/kernel'] == 'pre_bottleneck_convolutions/2/kernel'          # conv, BN, conv, BN, ... in one list
    assert paths['pre/bn1/moving_variance'] == 'pre_bottleneck_convolutions/3/moving_variance'
    assert paths['deconv_f16/res1/bn0/gamma'] == 'bottleneck_deconv_blocks/0/conv_layers/2/batchnorm0/gamma'
    assert paths['deconv_f2/deconv/kernel'] == 'bottleneck_deconv_blocks/4/upsample_layer/kernel'
    assert paths['multilinear_f32/conv0/bias'] == 'bottleneck_multilinear_blocks/2/conv_layers/0/bias'
    assert paths['post_merge_resnet/conv2/kernel'] == 'post_merge_resnet/conv_layers/2/kernel'
    assert paths['final/stage3/res/conv1/bias'] == 'final_convolutions/7/conv_layers/1/bias'
    assert paths['final/out1/kernel'] == 'final_convolutions/15/kernel'
    assert paths['dx_dense2/kernel'] == 'dx_dense_layers/2/kernel'
    assert paths['scaling/conv2/bias'] == 'scaling/stages/4/bias' and paths['scaling/dense1/kernel'] == 'scaling/dense_1/kernel'
    d = tmp_path / 'ck'
    d.mkdir()
    m.save_weights(str(d / 'chkpt.checkpoint'), save_format='tf')
    m2 = Homogeneous_Poisson_NN_Legacy(**cfg, device='cpu', seed=2)
    assert any(not np.array_equal(a, b) for a, b in zip(m.get_weights(), m2.get_weights()))
    train.load_model_checkpoint(m2, str(d))
    for a, b in zip(m.get_weights(), m2.get_weights()):
        np.testing.assert_array_equal(a, b)
    # a checkpoint lacking a variable is an error, extra variables (optimizer slots) are ignored
    t = T.read_bundle(str(d / 'chkpt.checkpoint'))
    t['optimizer/iter' + T.SUFFIX] = np.array(3, dtype=np.int64)
    T.write_bundle(str(d / 'extra'), t)
    m2.load_weights(str(d / 'extra'))
    del t['post_merge_conv/kernel' + T.SUFFIX]
    T.write_bundle(str(d / 'short'), t)
    with pytest.raises(ValueError, match='post_merge_conv/kernel'):
        m2.load_weights(str(d / 'short'))
    # a variable with the right element count but another layout (Cin / Cout swapped) must be refused, not reshaped
    t = T.read_bundle(str(d / 'chkpt.checkpoint'))
    key = 'final_convolutions/4/kernel' + T.SUFFIX                  # final/stage2/conv: (9, 9, 28, 24)
    assert t[key].shape == (9, 9, 28, 24)
    t[key] = np.ascontiguousarray(t[key].transpose(0, 1, 3, 2))
    T.write_bundle(str(d / 'swapped'), t)
    with pytest.raises(ValueError, match=r'\(9, 9, 24, 28\)'):
        m2.load_weights(str(d / 'swapped'))


def test_dbcnn_and_pcnn_checkpoints_use_the_reference_object_paths(tmp_path):
    """Dirichlet_BC_NN_Legacy_2 (models/Dirichlet_BC_NN_Legacy.py:47-95) and Poisson_CNN_Legacy (self.hpnn / self.dbcnn): object paths, Conv1D
    kernel shapes as TensorFlow holds them, the object-graph proto lists every variable, bit-exact round trips."""
    from poisson_cnn_amd import configs
    from poisson_cnn_amd.models import Dirichlet_BC_NN_Legacy_2, Homogeneous_Poisson_NN_Legacy, Poisson_CNN_Legacy
    dcfg = configs.dbcnn_tiny()['model']
    db = Dirichlet_BC_NN_Legacy_2(**dcfg, device='cpu', seed=1)
    paths = T.keras_object_paths(db)
    per = 3 if db.use_batchnorm else 2
    assert paths['bc/stage0/conv/kernel'] == 'boundary_convolutions/0/kernel'
    assert paths['bc/stage1/res/conv2/bias'] == 'boundary_convolutions/%d/conv_layers/2/bias' % (per + per - 1)
    if db.use_batchnorm:
        assert paths['bc/stage1/bn/moving_mean'] == 'boundary_convolutions/%d/moving_mean' % (per + 1)
        assert paths['bc/stage2/res/bn1/gamma'] == 'boundary_convolutions/%d/batchnorm1/gamma' % (2 * per + per - 1)
    assert paths['mlp/dense1/kernel'] == 'domain_info_dense_layers/1/kernel'
    assert paths['final/stage0/conv/bias'] == 'final_convolutions/0/bias' and paths['final/stage0/res/conv1/kernel'] == 'final_convolutions/1/conv_layers/1/kernel'
    assert paths['final/out1/kernel'] == 'final_convolutions/3/kernel'
    d = tmp_path / 'db'
    d.mkdir()
    db.save_weights(str(d / 'chkpt.checkpoint'), save_format='tf')
    graph = []
    t = T.read_bundle(str(d / 'chkpt.checkpoint'), graph_out=graph)
    assert t['boundary_convolutions/0/kernel' + T.SUFFIX].ndim == 3                     # tf.keras.layers.Conv1D kernel: (k, Cin, Cout)
    nodes = T.parse_object_graph(graph[0])
    for n in db.weight_names:                                                            # the object graph lists every variable of the model
        assert T.resolve_checkpoint_key(nodes, paths[n]) == paths[n] + T.SUFFIX
    db2 = Dirichlet_BC_NN_Legacy_2(**dcfg, device='cpu', seed=2)
    db2.load_weights(str(d / 'chkpt.checkpoint'))
    for a, b in zip(db.get_weights(), db2.get_weights()):
        np.testing.assert_array_equal(a, b)
    # the loader follows the checkpoint's OWN graph: a checkpoint whose keys are not the literal `<path>/.ATTRIBUTES/VARIABLE_VALUE` strings
    # (TensorFlow names a key after the first path its traversal finds) loads as long as its graph leads there
    renamed = {}
    for k, v in t.items():
        renamed[k.replace('boundary_convolutions/', 'boundary_convolution_ops/')] = v
    # re-encode: children edges keep the attribute names, the leaves' checkpoint_key point at the renamed tensors
    def graph_with_keys(keys_by_path):
        nds = [{'children': [], 'attr': None}]
        index = {'': 0}
        for path, key in keys_by_path.items():
            cur, prefix = 0, ''
            for comp in path.split('/'):
                prefix = comp if not prefix else prefix + '/' + comp
                if prefix not in index:
                    index[prefix] = len(nds)
                    nds.append({'children': [], 'attr': None})
                    nds[cur]['children'].append((index[prefix], comp))
                cur = index[prefix]
            nds[cur]['attr'] = (path, key)
        out = b''
        for nd in nds:
            body = b''.join(T._msg(1, T._field(1, 0, T._varint(i)) + T._msg(2, name.encode())) for i, name in nd['children'])
            if nd['attr'] is not None:
                body += T._msg(2, T._msg(1, b'VARIABLE_VALUE') + T._msg(2, nd['attr'][0].encode()) + T._msg(3, nd['attr'][1].encode()))
            out += T._msg(1, body)
        return out
    g2 = graph_with_keys({paths[n]: (paths[n] + T.SUFFIX).replace('boundary_convolutions/', 'boundary_convolution_ops/') for n in db.weight_names})
    orig = T._object_graph
    T._object_graph = lambda keys: g2
    try:
        T.write_bundle(str(d / 'renamed'), renamed)
    finally:
        T._object_graph = orig
    db3 = Dirichlet_BC_NN_Legacy_2(**dcfg, device='cpu', seed=3)
    db3.load_weights(str(d / 'renamed'))
    for a, b in zip(db.get_weights(), db3.get_weights()):
        np.testing.assert_array_equal(a, b)
    # Poisson_CNN_Legacy: the two sub-models under their attribute names
    full = configs.hpnn_tiny()['model']
    pc = Poisson_CNN_Legacy(Homogeneous_Poisson_NN_Legacy(**full, device='cpu', seed=4), Dirichlet_BC_NN_Legacy_2(**dcfg, device='cpu', seed=5))
    pp = T.keras_object_paths(pc)
    assert pp['hpnn/pre/conv0/kernel'] == 'hpnn/pre_bottleneck_convolutions/0/kernel' and pp['dbcnn/mlp/dense0/bias'] == 'dbcnn/domain_info_dense_layers/0/bias'
    pc.save_weights(str(d / 'pcnn'), save_format='tf')
    pc2 = Poisson_CNN_Legacy(Homogeneous_Poisson_NN_Legacy(**full, device='cpu', seed=6), Dirichlet_BC_NN_Legacy_2(**dcfg, device='cpu', seed=7))
    pc2.load_weights(str(d / 'pcnn'))
    for a, b in zip(pc.get_weights(), pc2.get_weights()):
        np.testing.assert_array_equal(a, b)


def test_dbcnn_metalearning_checkpoint_object_paths(tmp_path):
    """Dirichlet_BC_NN_Metalearning (models/Dirichlet_BC_NN_Metalearning.py:43-93): hyper-network variables under <stage layer>/dense_layers/<i>, the
    optional LayerNormalization as the last entry of that list, metalearning_resnet members conv0..2 / batchnorm0..1, the domain-info chain
    [Dense, LayerNormalization, Dense, ...]; the object graph lists every variable and the round trip is bit-exact."""
    from poisson_cnn_amd.dbcnn_models import Dirichlet_BC_NN_Metalearning
    kw = dict(ndims=2, use_batchnorm=True, postsmoother_iterations=0,
              boundary_conv_config={'filters': [4, 6], 'kernel_sizes': [7, 5], 'use_layernorm': True, 'pre_output_dense_units': [6, 10]},
              spp_config={'levels': [[2], 3]}, domain_info_mlp_config={'units': [14, 9, 6], 'activations': ['tf.nn.leaky_relu', 'tf.nn.tanh', 'softmax']},
              final_convolutions_config={'filters': [5, 4, 2, 1], 'kernel_sizes': [5, 3, 3, 3], 'final_regular_conv_stages': 2, 'use_bias': True})
    m = Dirichlet_BC_NN_Metalearning(**kw, device='cpu', seed=1)
    paths = T.keras_object_paths(m)
    assert paths['bc/stage0/conv/dense0/kernel'] == 'boundary_convolutions/0/dense_layers/0/kernel'
    assert paths['bc/stage0/conv/layernorm/gamma'] == 'boundary_convolutions/0/dense_layers/3/gamma'
    assert paths['bc/stage1/res/conv2/dense1/bias'] == 'boundary_convolutions/3/conv2/dense_layers/1/bias'
    assert paths['bc/stage1/res/conv0/layernorm/beta'] == 'boundary_convolutions/3/conv0/dense_layers/3/beta'
    assert paths['bc/stage1/res/bn1/moving_variance'] == 'boundary_convolutions/3/batchnorm1/moving_variance'
    assert paths['mlp/dense0/kernel'] == 'domain_info_dense_layers/0/kernel' and paths['mlp/ln1/gamma'] == 'domain_info_dense_layers/1/gamma'
    assert paths['mlp/dense2/bias'] == 'domain_info_dense_layers/4/bias'
    assert paths['final/stage1/conv/dense2/kernel'] == 'final_convolutions/2/dense_layers/2/kernel'
    assert paths['final/stage1/res/conv1/dense0/kernel'] == 'final_convolutions/3/conv1/dense_layers/0/kernel'
    assert paths['final/out0/kernel'] == 'final_convolutions/4/kernel' and paths['final/out1/bias'] == 'final_convolutions/5/bias'
    m.save_weights(str(tmp_path / 'chkpt.checkpoint'), save_format='tf')
    graph = []
    T.read_bundle(str(tmp_path / 'chkpt.checkpoint'), graph_out=graph)
    nodes = T.parse_object_graph(graph[0])
    for n in m.weight_names:
        assert T.resolve_checkpoint_key(nodes, paths[n]) == paths[n] + T.SUFFIX
    m2 = Dirichlet_BC_NN_Metalearning(**kw, device='cpu', seed=2)
    m2.load_weights(str(tmp_path / 'chkpt.checkpoint'))
    for a, b in zip(m.get_weights(), m2.get_weights()):
        np.testing.assert_array_equal(a, b)
