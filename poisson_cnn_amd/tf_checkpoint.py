"""TensorFlow checkpoint interchange: reader and writer of the TensorBundle V2 container that
`model.save_weights(prefix)` / `tf.train.Checkpoint` produce and `train/utils.py:10-29` (`load_model_checkpoint`) consumes -
`<prefix>.index` + `<prefix>.data-00000-of-00001` (+ the `checkpoint` state file) - without TensorFlow.

Format, restated from the published TensorFlow sources (tensorflow/core/util/tensor_bundle/tensor_bundle.{h,cc}, core/lib/io/table*.cc,
core/protobuf/tensor_bundle.proto, core/protobuf/trackable_object_graph.proto; the repository pins tensorflow-gpu==2.4):
  * the index file is a LevelDB-format table (prefix-compressed data blocks with restart arrays, a 5-byte trailer per block = compression
    type 0 + masked CRC-32C, an empty metaindex block, an index block, a 48-byte footer ending in the magic 0xdb4775248b80fb57);
    key "" holds a BundleHeaderProto (num_shards = 1, little endian, version.producer = 1), every other key a BundleEntryProto
    (dtype, shape, shard_id, offset, size, masked crc32c of the bytes);
  * the data shard is the tensors' raw little-endian bytes back to back in key order;
  * Keras writes one variable per key `<object path>/.ATTRIBUTES/VARIABLE_VALUE` plus `_CHECKPOINTABLE_OBJECT_GRAPH`, a string tensor holding a
    TrackableObjectGraph proto (nodes, named child edges, one SerializedTensor attribute per variable) that `load_weights` walks from the
    root by attribute name.
PARITY UNPINNED: no TensorFlow exists in the build environment, so files written here have never been read by TensorFlow and vice versa; what
is tested is the format's fixed points (magic, CRC-32C known answers, the masking formula) and the round trip.  The object paths
(`keras_object_paths`) restate the attribute names of models/Homogeneous_Poisson_NN_Legacy.py:41-115, blocks/*.py and layers/Scaling.py.
"""
import os
import struct
from ctypes import c_size_t, c_uint32, c_void_p

import numpy as np

MAGIC = 0xdb4775248b80fb57
DT = {np.dtype('float32'): 1, np.dtype('float64'): 2, np.dtype('int32'): 3, np.dtype('int64'): 9}
DT_INV = {v: k for k, v in DT.items()}
DT_STRING = 7
OBJECT_GRAPH_KEY = '_CHECKPOINTABLE_OBJECT_GRAPH'
SUFFIX = '/.ATTRIBUTES/VARIABLE_VALUE'


# ----------------------------------------------------------------------------------------------------------------- crc32c
def crc32c(data, crc=0):
    from . import _lib
    lib = _lib.load()
    lib.pcnn_crc32c.restype = c_uint32
    lib.pcnn_crc32c.argtypes = [c_void_p, c_size_t, c_uint32]
    buf = bytes(data)
    return int(lib.pcnn_crc32c(buf, len(buf), crc))


def mask(crc):
    """tensorflow/core/lib/hash/crc32c.h Mask(): rotate right by 15, add a constant."""
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


def unmask(m):
    rot = (m - 0xa282ead8) & 0xffffffff
    return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ----------------------------------------------------------------------------------------------------------------- protobuf (the few messages needed)
def _varint(n):
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7f
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift = val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7f) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _field(num, wire, payload):
    return _varint((num << 3) | wire) + payload


def _msg(num, payload):
    return _field(num, 2, _varint(len(payload)) + payload)


def _parse(buf):
    """-> list of (field number, wire type, value) with value int (varint / fixed) or bytes (length-delimited)."""
    out, pos = [], 0
    while pos < len(buf):
        tag, pos = _read_varint(buf, pos)
        num, wire = tag >> 3, tag & 7
        if wire == 0:
            v, pos = _read_varint(buf, pos)
        elif wire == 2:
            n, pos = _read_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wire == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        elif wire == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        else:
            raise ValueError('unsupported protobuf wire type %d' % wire)
        out.append((num, wire, v))
    return out


def _header_proto():
    return _field(1, 0, _varint(1)) + _msg(3, _field(1, 0, _varint(1)))            # num_shards = 1, endianness = LITTLE (default 0), version {producer: 1}


def _shape_proto(shape):
    return b''.join(_msg(2, _field(1, 0, _varint(int(d)))) for d in shape)          # repeated Dim dim = 2 { int64 size = 1 }


def _entry_proto(dtype, shape, offset, size, crc):
    p = _field(1, 0, _varint(dtype)) + _msg(2, _shape_proto(shape))
    if offset:
        p += _field(4, 0, _varint(offset))
    p += _field(5, 0, _varint(size)) + _field(6, 5, struct.pack('<I', crc))
    return p                                                                         # shard_id = 0 is the default and omitted


# ----------------------------------------------------------------------------------------------------------------- LevelDB-format table
class _BlockBuilder:
    def __init__(self, restart_interval=16):
        self.buf, self.restarts, self.count, self.last, self.interval = bytearray(), [0], 0, b'', restart_interval

    def add(self, key, value):
        shared = 0
        if self.count < self.interval:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        self.last = key
        self.count += 1

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))


def _write_block(f, contents):
    """-> BlockHandle (offset, size); trailer = type byte 0 (no compression) + masked crc32c over contents + type."""
    off = f.tell()
    f.write(contents)
    f.write(b'\x00' + struct.pack('<I', mask(crc32c(contents + b'\x00'))))
    return off, len(contents)


def _write_table(path, items, block_size=262144):
    """items: sorted list of (key bytes, value bytes)."""
    with open(path, 'wb') as f:
        index = _BlockBuilder(restart_interval=1)
        blk, last = _BlockBuilder(), None
        for k, v in items:
            blk.add(k, v)
            last = k
            if len(blk.buf) >= block_size:
                h = _write_block(f, blk.finish())
                index.add(last, _varint(h[0]) + _varint(h[1]))
                blk = _BlockBuilder()
        if blk.buf or last is None:
            h = _write_block(f, blk.finish())
            index.add(last if last is not None else b'', _varint(h[0]) + _varint(h[1]))
        meta = _write_block(f, _BlockBuilder().finish())
        idx = _write_block(f, index.finish())
        handles = _varint(meta[0]) + _varint(meta[1]) + _varint(idx[0]) + _varint(idx[1])
        f.write(handles + b'\x00' * (40 - len(handles)) + struct.pack('<Q', MAGIC))


def _read_block(buf, off, size, verify=True):
    contents = buf[off:off + size]
    if verify:
        typ = buf[off + size:off + size + 1]
        stored = struct.unpack_from('<I', buf, off + size + 1)[0]
        if typ != b'\x00':
            raise ValueError('compressed table blocks are not supported (TensorBundle writes none)')
        if unmask(stored) != crc32c(contents + typ):
            raise ValueError('table block checksum mismatch')
    nrest = struct.unpack_from('<I', contents, len(contents) - 4)[0]
    end = len(contents) - 4 - 4 * nrest
    pos, key, out = 0, b'', []
    while pos < end:
        shared, pos = _read_varint(contents, pos)
        non_shared, pos = _read_varint(contents, pos)
        vlen, pos = _read_varint(contents, pos)
        key = key[:shared] + contents[pos:pos + non_shared]
        pos += non_shared
        out.append((key, contents[pos:pos + vlen]))
        pos += vlen
    return out


def _read_table(path, verify=True):
    buf = open(path, 'rb').read()
    if len(buf) < 48 or struct.unpack_from('<Q', buf, len(buf) - 8)[0] != MAGIC:
        raise ValueError('%s is not a TensorBundle index (bad magic)' % path)
    footer = buf[-48:]
    _, p = _read_varint(footer, 0)
    _, p = _read_varint(footer, p)
    ioff, p = _read_varint(footer, p)
    isz, p = _read_varint(footer, p)
    items = []
    for _, handle in _read_block(buf, ioff, isz, verify):
        off, q = _read_varint(handle, 0)
        sz, q = _read_varint(handle, q)
        items += _read_block(buf, off, sz, verify)
    return items


# ----------------------------------------------------------------------------------------------------------------- object graph
def _object_graph(keys):
    """TrackableObjectGraph for variables at `<path>/.ATTRIBUTES/VARIABLE_VALUE`: one node per path prefix, children named by the path
    components in first-seen order, one attribute (name VARIABLE_VALUE, full_name = path, checkpoint_key = key) per leaf."""
    nodes = [{'children': [], 'attr': None}]
    index = {'': 0}
    for key in keys:
        path = key[:-len(SUFFIX)]
        cur, prefix = 0, ''
        for comp in path.split('/'):
            prefix = comp if not prefix else prefix + '/' + comp
            if prefix not in index:
                index[prefix] = len(nodes)
                nodes.append({'children': [], 'attr': None})
                nodes[cur]['children'].append((index[prefix], comp))
            cur = index[prefix]
        nodes[cur]['attr'] = (path, key)
    out = b''
    for nd in nodes:
        body = b''.join(_msg(1, _field(1, 0, _varint(i)) + _msg(2, name.encode())) for i, name in nd['children'])
        if nd['attr'] is not None:
            body += _msg(2, _msg(1, b'VARIABLE_VALUE') + _msg(2, nd['attr'][0].encode()) + _msg(3, nd['attr'][1].encode()))
        out += _msg(1, body)
    return out


# ----------------------------------------------------------------------------------------------------------------- bundle
def write_bundle(prefix, tensors, with_object_graph=True):
    """tensors: dict checkpoint key -> numpy array (float32 / float64 / int32 / int64).  Writes <prefix>.index, <prefix>.data-00000-of-00001
    and the `checkpoint` state file next to them."""
    keys = sorted(tensors)
    blobs = {}
    for k in keys:
        a = np.asarray(tensors[k])
        if a.dtype not in DT:
            raise ValueError('unsupported dtype %s for %s' % (a.dtype, k))
        raw = a.astype(a.dtype.newbyteorder('<')).tobytes()
        blobs[k] = (DT[a.dtype], a.shape, raw, mask(crc32c(raw)))                 # tobytes() is C order whatever the strides
    if with_object_graph:
        graph = _object_graph([k for k in keys if k.endswith(SUFFIX)])
        # tensor_bundle.cc WriteStringTensor: [varint64 length per element][masked crc32c of the lengths, 4 bytes][the bytes]; the lengths'
        # checksum runs over each length as a little-endian uint32 (uint64 above 4 GiB), and the entry's checksum continues that running
        # value over the 4 stored checksum bytes and then the string bytes
        lcrc = crc32c(struct.pack('<I', len(graph)))
        lbytes = struct.pack('<I', mask(lcrc))
        blobs[OBJECT_GRAPH_KEY] = (DT_STRING, (), _varint(len(graph)) + lbytes + graph, mask(crc32c(graph, crc32c(lbytes, lcrc))))
    items = [(b'', _header_proto())]
    offset = 0
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        for k in sorted(blobs):
            dt, shape, raw, crc = blobs[k]
            f.write(raw)
            items.append((k.encode(), _entry_proto(dt, shape, offset, len(raw), crc)))
            offset += len(raw)
    _write_table(prefix + '.index', sorted(items))
    base = os.path.basename(prefix)
    with open(os.path.join(os.path.dirname(prefix) or '.', 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))


def read_bundle(prefix, verify=True, graph_out=None):
    """-> dict checkpoint key -> numpy array (the object-graph entry is skipped; its serialized proto is appended to `graph_out` if given)."""
    items = _read_table(prefix + '.index', verify)
    if not items or items[0][0] != b'':
        raise ValueError('TensorBundle header entry missing')
    hdr = {n: v for n, _, v in _parse(items[0][1])}
    nshards = hdr.get(1, 1)
    if hdr.get(2, 0) != 0:
        raise ValueError('big-endian bundles are not supported')
    shards = {}
    out = {}
    for key, val in items[1:]:
        e = {}
        shape = []
        for n, _, v in _parse(val):
            if n == 2:
                shape = [dict((a, b) for a, _, b in _parse(dim)).get(1, 0) for num, _, dim in _parse(v) if num == 2]
            elif n == 7:
                raise ValueError('sliced (partitioned) variables are not supported: %s' % key.decode())
            else:
                e[n] = v
        dtype = e.get(1, 0)
        if key.decode() == OBJECT_GRAPH_KEY and graph_out is not None:
            sid = e.get(3, 0)
            if sid not in shards:
                shards[sid] = np.memmap('%s.data-%05d-of-%05d' % (prefix, sid, nshards), dtype=np.uint8, mode='r')
            raw = bytes(shards[sid][e.get(4, 0):e.get(4, 0) + e.get(5, 0)])
            n, q = _read_varint(raw, 0)                             # string tensor: [varint length][4-byte masked crc of the lengths][bytes]
            graph_out.append(raw[q + 4:q + 4 + n])
        if dtype == DT_STRING or key.decode() == OBJECT_GRAPH_KEY:
            continue
        if dtype not in DT_INV:
            raise ValueError('unsupported dtype %d for %s' % (dtype, key.decode()))
        sid = e.get(3, 0)
        if sid not in shards:
            shards[sid] = np.memmap('%s.data-%05d-of-%05d' % (prefix, sid, nshards), dtype=np.uint8, mode='r')
        raw = bytes(shards[sid][e.get(4, 0):e.get(4, 0) + e.get(5, 0)])
        if verify and unmask(e.get(6, 0)) != crc32c(raw):
            raise ValueError('checksum mismatch for %s' % key.decode())
        out[key.decode()] = np.frombuffer(raw, dtype=DT_INV[dtype].newbyteorder('<')).astype(DT_INV[dtype]).reshape(shape)
    return out


def parse_object_graph(blob):
    """TrackableObjectGraph -> list of nodes {children: {local_name: node_id}, attributes: {name: checkpoint_key}}."""
    nodes = []
    for num, _, body in _parse(blob):
        if num != 1:
            continue
        nd = {'children': {}, 'attributes': {}}
        for n, _, v in _parse(body):
            if n == 1:
                f = {a: b for a, _, b in _parse(v)}
                nd['children'][f.get(2, b'').decode()] = f.get(1, 0)
            elif n == 2:
                f = {a: b for a, _, b in _parse(v)}
                nd['attributes'][f.get(1, b'').decode()] = f.get(3, b'').decode()
        nodes.append(nd)
    return nodes


def resolve_checkpoint_key(nodes, object_path):
    """Walks `children.local_name` edges from the root along the components of a Keras object path and returns the leaf's VARIABLE_VALUE
    checkpoint key - the key TensorFlow itself chose for that variable (the first path its traversal found), whatever string that is.
    None when the graph has no such path."""
    cur = 0
    for comp in object_path.split('/'):
        nxt = nodes[cur]['children'].get(comp)
        if nxt is None or nxt >= len(nodes):
            return None
        cur = nxt
    return nodes[cur]['attributes'].get('VARIABLE_VALUE')


# ----------------------------------------------------------------------------------------------------------------- Keras object paths of the models
def _unit(m, names, unit, theirs, bn_theirs=None):
    """A ConvUnit (kernel, bias) and, when it carries one, its BatchNormalization (gamma, beta, moving_mean, moving_variance)."""
    for n in names:
        if n.startswith(unit.name + '/'):
            m[n] = theirs + n[len(unit.name):]
        elif getattr(unit, 'bn_name', None) is not None and n.startswith(unit.bn_name + '/'):
            m[n] = bn_theirs + n[len(unit.bn_name):]


def _resnet(m, names, r, theirs):
    """blocks/resnet.py:12-27: self.conv_layers[0..2], self.batchnorm0, self.batchnorm1."""
    for i, c in enumerate((r.c0, r.c1, r.c2)):
        _unit(m, names, c, '%s/conv_layers/%d' % (theirs, i), '%s/batchnorm%d' % (theirs, i))


def _block(m, names, blk, theirs):
    """blocks/bottleneck_block.py:26-66,97: self.downsample_layer (strided convolution), self.conv_layers (convolutions, their BatchNormalization
    layers and resnets in ONE list, in call order), self.upsample_layer (deconvupscale; the multilinear Upsample has no variables)."""
    from . import layers as L
    if blk.down_conv is not None:
        _unit(m, names, blk.down_conv, theirs + '/downsample_layer')
    idx = 0
    for st in blk.stages:
        if isinstance(st, L.resnet):
            _resnet(m, names, st, '%s/conv_layers/%d' % (theirs, idx))
            idx += 1
        else:
            _unit(m, names, st, '%s/conv_layers/%d' % (theirs, idx), '%s/conv_layers/%d' % (theirs, idx + 1))
            idx += 2 if st.bn_name is not None else 1
    for n in names:
        if n.startswith(blk.name + '/deconv/'):
            m[n] = theirs + '/upsample_layer' + n[len(blk.name + '/deconv'):]


def _dbcnn_paths(model):
    """Dirichlet_BC_NN_Legacy_2 (models/Dirichlet_BC_NN_Legacy.py:47-95): self.boundary_convolutions = [conv, (BatchNormalization), resnet] per stage,
    self.domain_info_dense_layers, self.final_convolutions = [conv, resnet] per stage followed by the regular convolutions.  (The parallel
    *_ops lists hold the same layer objects; TensorFlow names a variable after the first attribute path that reaches it.)"""
    import re
    names = model.weight_names
    per = 3 if model.use_batchnorm else 2
    nst = 1 + max([int(re.match(r'final/stage(\d+)/', n).group(1)) for n in names if n.startswith('final/stage')] + [-1])
    m = {}
    for n in names:
        q = n.split('/')
        if q[0] == 'bc':
            k = int(q[1][5:])
            if q[2] == 'conv':
                m[n] = 'boundary_convolutions/%d/%s' % (per * k, q[3])
            elif q[2] == 'bn':
                m[n] = 'boundary_convolutions/%d/%s' % (per * k + 1, q[3])
            else:                                                       # res/conv{i} | res/bn{i}
                sub = 'conv_layers/%s' % q[3][4:] if q[3].startswith('conv') else 'batchnorm%s' % q[3][2:]
                m[n] = 'boundary_convolutions/%d/%s/%s' % (per * k + per - 1, sub, q[4])
        elif q[0] == 'mlp':
            m[n] = 'domain_info_dense_layers/%s/%s' % (q[1][5:], q[2])
        elif q[0] == 'final' and q[1].startswith('stage'):
            k = int(q[1][5:])
            if q[2] == 'conv':
                m[n] = 'final_convolutions/%d/%s' % (2 * k, q[3])
            else:
                sub = 'conv_layers/%s' % q[3][4:] if q[3].startswith('conv') else 'batchnorm%s' % q[3][2:]
                m[n] = 'final_convolutions/%d/%s/%s' % (2 * k + 1, sub, q[4])
        elif q[0] == 'final' and q[1].startswith('out'):
            m[n] = 'final_convolutions/%d/%s' % (2 * nst + int(q[1][3:]), q[2])
    return m


def _dbcnn_meta_paths(model):
    """Dirichlet_BC_NN_Metalearning (models/Dirichlet_BC_NN_Metalearning.py:43-93): self.boundary_convolutions = [metalearning_conv, metalearning_resnet]
    per stage, self.domain_info_dense_layers = [Dense, (LayerNormalization, Dense)...], self.final_convolutions = [metalearning_conv,
    metalearning_resnet] per stage followed by the regular convolutions.  A metalearning_conv keeps its hyper-network in self.dense_layers
    (layers/metalearning_conv.py:127-129: the Dense layers, then the optional LayerNormalization); a metalearning_resnet is conv0 / conv1 / conv2 /
    batchnorm0 / batchnorm1 (blocks/metalearning_resnet.py:10-25)."""
    nmeta = len(model.final_meta)

    def hyper(q):                                       # [..., 'dense<i>' | 'layernorm', var] -> dense_layers/<i>/<var>
        if q[0] == 'layernorm':
            ndense = 1 + max(int(n.split('/')[-2][5:]) for n in model.weight_names if n.startswith(prefix) and n.split('/')[-2].startswith('dense'))
            return 'dense_layers/%d/%s' % (ndense, q[1])
        return 'dense_layers/%s/%s' % (q[0][5:], q[1])

    m = {}
    for n in model.weight_names:
        q = n.split('/')
        if q[0] in ('bc', 'final') and q[1].startswith('stage'):
            k = int(q[1][5:])
            top = 'boundary_convolutions' if q[0] == 'bc' else 'final_convolutions'
            if q[2] == 'conv':
                prefix = '/'.join(q[:3]) + '/'
                m[n] = '%s/%d/%s' % (top, 2 * k, hyper(q[3:]))
            elif q[3].startswith('bn'):
                m[n] = '%s/%d/batchnorm%s/%s' % (top, 2 * k + 1, q[3][2:], q[4])
            else:
                prefix = '/'.join(q[:4]) + '/'
                m[n] = '%s/%d/%s/%s' % (top, 2 * k + 1, q[3], hyper(q[4:]))
        elif q[0] == 'final':
            m[n] = 'final_convolutions/%d/%s' % (nmeta + int(q[1][3:]), q[2])
        elif q[0] == 'mlp':
            k = int(q[1][5:] if q[1].startswith('dense') else q[1][2:])
            m[n] = 'domain_info_dense_layers/%d/%s' % (2 * k if q[1].startswith('dense') else 2 * k - 1, q[2])
    return m


def keras_object_paths(model):
    """Our parameter name -> the reference model's Keras object path: Homogeneous_Poisson_NN_Legacy (attribute names of
    models/Homogeneous_Poisson_NN_Legacy.py:41-115), Dirichlet_BC_NN_Legacy_2, and Poisson_CNN_Legacy (self.hpnn / self.dbcnn,
    models/Poisson_CNN_Legacy.py:8-9) - the three models the reference's training scripts save through ModelCheckpoint
    (train/hpnn_legacy_train.py, train/dbcnn_legacy_train.py, train/pcnn_end_to_end.py; train/utils.py:10-29)."""
    from . import layers as L
    from .models import Homogeneous_Poisson_NN_Legacy, Dirichlet_BC_NN_Legacy_2, Poisson_CNN_Legacy
    if isinstance(model, Poisson_CNN_Legacy):
        m = {'hpnn/' + n: 'hpnn/' + v for n, v in keras_object_paths(model.hpnn).items()}
        m.update({'dbcnn/' + n: 'dbcnn/' + v for n, v in keras_object_paths(model.dbcnn).items()})
        return m
    from .dbcnn_models import Dirichlet_BC_NN_Metalearning
    if isinstance(model, (Dirichlet_BC_NN_Legacy_2, Dirichlet_BC_NN_Metalearning)):
        m = _dbcnn_paths(model) if isinstance(model, Dirichlet_BC_NN_Legacy_2) else _dbcnn_meta_paths(model)
        missing = [n for n in model.weight_names if n not in m]
        if missing or len(set(m.values())) != len(m):
            raise RuntimeError('no unique Keras object path for %s' % (missing[:5],))
        return m
    if not isinstance(model, Homogeneous_Poisson_NN_Legacy):
        raise NotImplementedError('TensorFlow-format checkpoints are mapped for Homogeneous_Poisson_NN_Legacy, Dirichlet_BC_NN_Legacy_2, Poisson_CNN_Legacy and '
                                  'Dirichlet_BC_NN_Metalearning - the models the reference can construct; the two train/hpnn_train.py classes raise NameError in the '
                                  'reference, so no TensorFlow checkpoint of them can exist (flat .npz works for every model)')
    names = model.weight_names
    m = {}
    idx = 0
    for c in model.pre:                                            # :43-57 conv, [BatchNormalization], conv, ... in one list
        _unit(m, names, c, 'pre_bottleneck_convolutions/%d' % idx, 'pre_bottleneck_convolutions/%d' % (idx + 1))
        idx += 2 if c.bn_name is not None else 1
    for i, b in enumerate(model.bottleneck_deconv_blocks):         # :63-69, already sorted by downsampling factor (descending) like the reference
        _block(m, names, b, 'bottleneck_deconv_blocks/%d' % i)
    for i, b in enumerate(model.bottleneck_multilinear_blocks):
        _block(m, names, b, 'bottleneck_multilinear_blocks/%d' % i)
    _unit(m, names, model.non_bottleneck_conv, 'non_bottleneck_conv')
    _unit(m, names, model.post_merge_conv, 'post_merge_conv')
    _resnet(m, names, model.post_merge_resnet, 'post_merge_resnet')
    for i, lyr in enumerate(model.final):                          # :79-96 [conv, resnet] per stage, then the regular convolutions
        if isinstance(lyr, L.resnet):
            _resnet(m, names, lyr, 'final_convolutions/%d' % i)
        else:
            _unit(m, names, lyr, 'final_convolutions/%d' % i)
    for n in names:
        p = n.split('/')
        if p[0].startswith('dx_dense'):                            # :102
            m[n] = 'dx_dense_layers/%s/%s' % (p[0][8:], p[1])
        elif p[0] == 'scaling':                                    # layers/Scaling.py:23-33: stages = [conv, pool, conv, pool, ...], dense_0..2
            m[n] = 'scaling/stages/%d/%s' % (2 * int(p[1][4:]), p[2]) if p[1].startswith('conv') else 'scaling/dense_%s/%s' % (p[1][5:], p[2])
    missing = [n for n in names if n not in m]
    if missing:
        raise RuntimeError('no Keras object path for %s' % missing[:5])
    if len(set(m.values())) != len(m):
        raise RuntimeError('Keras object paths are not unique')
    return m


def _tf_shape(name, w):
    """The tensor as TensorFlow holds it: our 1-D boundary convolutions keep their Conv1D kernels (k, Cin, Cout) as (1, k, Cin, Cout)."""
    w = np.asarray(w)
    return w.reshape(w.shape[1:]) if (name.endswith('/kernel') and w.ndim == 4 and w.shape[0] == 1 and '/bc/' in '/' + name) else w


def save_tf_checkpoint(model, prefix):
    paths = keras_object_paths(model)
    write_bundle(prefix, {paths[n] + SUFFIX: _tf_shape(n, w) for n, w in zip(model.weight_names, model.get_weights())})


def load_tf_checkpoint(model, prefix):
    """Loads variables by Keras object path.  The key of each variable is looked up in the checkpoint's own object graph (walk of the
    `children.local_name` edges along the path -> the VARIABLE_VALUE attribute's checkpoint_key): TensorFlow names a key after the first path
    its traversal finds, which need not be the literal `<path>/.ATTRIBUTES/VARIABLE_VALUE`; only a checkpoint without a graph entry falls back
    to that literal.  Optimizer slots and anything else in the checkpoint are ignored.  (Cross-reading with TensorFlow itself is untested here:
    there is no TensorFlow in the build environment - DESIGN.md section 7.)"""
    paths = keras_object_paths(model)
    graph = []
    tensors = read_bundle(prefix, graph_out=graph)
    nodes = parse_object_graph(graph[0]) if graph else None
    weights = {}
    ours = dict(zip(model.weight_names, model.get_weights()))
    for n in model.weight_names:
        key = resolve_checkpoint_key(nodes, paths[n]) if nodes else None
        if key is None or key not in tensors:
            key = paths[n] + SUFFIX
        if key not in tensors:
            raise ValueError('checkpoint %s has no variable %s (expected for %s)' % (prefix, key, n))
        t = tensors[key]
        want = tuple(ours[n].shape)
        # shapes must MATCH - equal element counts are not enough (a kernel with Cin / Cout swapped, or another stage's kernel with the same
        # product, would load scrambled).  The one exception is the rule _tf_shape applies on save: our (1, k, Cin, Cout) holds a Conv1D's (k, Cin, Cout).
        if not (tuple(t.shape) == want or (len(want) == 4 and want[0] == 1 and tuple(t.shape) == want[1:])):
            raise ValueError('checkpoint variable %s has shape %s, the model\'s %s has %s' % (key, tuple(t.shape), n, want))
        weights[n] = t.reshape(want)
    model.set_weights(weights)
