"""ctypes binding of libpcnn.so (include/pcnn.h).  There is NO fallback: if the HIP library is missing or a call
fails, the product path raises."""
import ctypes
import os

import torch  # noqa: F401  - MUST be imported before libpcnn.so is dlopen'ed: both then share torch's HIP runtime
#                            (libamdhip64.so.7 is resolved by soname to the copy torch already loaded), so device pointers and
#                            streams are valid on both sides.  Loading libpcnn first would start a second, separate runtime.
from ctypes import POINTER, Structure, byref, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PCNN_LIBRARY') or os.path.join(_HERE, 'libpcnn.so')     # PCNN_LIBRARY: another build of the same library (developer A/B runs)


class ConvDesc(Structure):
    _fields_ = [('N', c_int), ('H', c_int), ('W', c_int), ('Cin', c_int), ('ldx', c_int),
                ('Ho', c_int), ('Wo', c_int), ('Cout', c_int), ('ldy', c_int),
                ('kh', c_int), ('kw', c_int), ('pad_top', c_int), ('pad_left', c_int),
                ('pad_mode', c_int), ('pad_value', c_float), ('act', c_int), ('act_alpha', c_float),
                ('ld_res', c_int), ('ld_act_out', c_int)]


_lib = None


# ---------------------------------------------------------------------------------------------------------------- prototypes from include/pcnn.h
# Every entry point gets its argtypes / restype from the header's own declaration when the library is loaded: without
# them ctypes passes a bare Python int as a C int - a 64-bit count or size silently loses its upper half - and a float as a double.  The
# converters below are lenient about HOW a caller spells a value (a Python number or any ctypes scalar) and strict about WHAT arrives: the
# declared C type, range-checked; a value that does not fit, or a float where an integer is declared, raises instead of corrupting memory.
def _scalar(ctype, name, lo=None, hi=None):
    is_float = ctype in (c_float, ctypes.c_double)

    class _Arg(ctype):
        @classmethod
        def from_param(cls, v):
            if isinstance(v, ctypes._SimpleCData):
                v = v.value
            if is_float:
                if isinstance(v, (str, bytes, bool)) or not hasattr(v, '__float__'):
                    raise TypeError('%s argument: %r is not a number' % (name, v))
                return ctype(float(v))
            if isinstance(v, bool):
                v = int(v)
            if not isinstance(v, int):
                if hasattr(v, '__index__'):
                    v = v.__index__()
                else:
                    raise TypeError('%s argument: %r is not an integer' % (name, v))
            if not lo <= v <= hi:
                raise OverflowError('%s argument: %d does not fit' % (name, v))
            return ctype(v)
    _Arg.__name__ = 'arg_' + name
    return _Arg


_C_SCALARS = {
    'int': _scalar(c_int, 'int', -2 ** 31, 2 ** 31 - 1), 'unsigned': _scalar(ctypes.c_uint, 'unsigned', 0, 2 ** 32 - 1),
    'int32_t': _scalar(c_int32, 'int32_t', -2 ** 31, 2 ** 31 - 1), 'uint32_t': _scalar(ctypes.c_uint32, 'uint32_t', 0, 2 ** 32 - 1),
    'int64_t': _scalar(c_int64, 'int64_t', -2 ** 63, 2 ** 63 - 1), 'uint64_t': _scalar(ctypes.c_uint64, 'uint64_t', 0, 2 ** 64 - 1),
    'long long': _scalar(ctypes.c_longlong, 'long long', -2 ** 63, 2 ** 63 - 1),
    'size_t': _scalar(c_size_t, 'size_t', 0, 2 ** 64 - 1), 'float': _scalar(c_float, 'float'), 'double': _scalar(ctypes.c_double, 'double'),
}
_C_RESTYPES = {'int': c_int, 'size_t': c_size_t, 'uint32_t': ctypes.c_uint32, 'const char*': c_char_p, 'void': None, 'int64_t': c_int64, 'float': c_float, 'double': ctypes.c_double}


def header_prototypes(path=None):
    """{name: (return type string, [argument type strings])} of every `pcnn_*` function declared in include/pcnn.h (comments stripped;
    a pointer argument is reported as '<base>*')."""
    import re
    path = path or os.path.join(os.path.dirname(_HERE), 'include', 'pcnn.h')
    with open(path) as f:
        src = f.read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    src = re.sub(r'//[^\n]*', ' ', src)
    protos = {}
    for m in re.finditer(r'([A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)(pcnn_[a-z0-9_]+)\s*\(([^()]*)\)\s*;', src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        ret = ' '.join(ret.replace('*', ' * ').split()).replace(' *', '*')
        if ret.startswith('typedef') or 'struct' in ret:
            continue
        types = []
        args = ' '.join(args.split())
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a or '[' in a:
                    base = a[:a.rindex('*')] if '*' in a else a[:a.index('[')].rsplit(' ', 1)[0]
                    types.append(' '.join(base.replace('*', ' * ').split()).replace(' *', '*') + '*')
                else:
                    parts = a.split()
                    types.append(' '.join(parts[:-1]) if len(parts) > 1 else parts[0])          # drop the parameter name
        protos[name] = (ret, types)
    return protos


def _bind_prototypes(lib):
    bound = {}
    for name, (ret, types) in header_prototypes().items():
        fn = getattr(lib, name, None)
        if fn is None:
            continue                                               # tests/test_host_logic.py checks that every declared symbol is exported
        argtypes = []
        for t in types:
            if t.endswith('*') or t == 'pcnn_handle':
                argtypes.append(c_void_p)                          # accepts None, an address, c_void_p, byref(struct), ctypes arrays / pointers
            else:
                key = t.replace('const ', '').strip()
                if key == 'unsigned int':
                    key = 'unsigned'
                if key not in _C_SCALARS:
                    raise RuntimeError('include/pcnn.h: %s has an argument of type %r that _lib.py cannot bind' % (name, t))
                argtypes.append(_C_SCALARS[key])
        if ret not in _C_RESTYPES:
            raise RuntimeError('include/pcnn.h: %s returns %r, which _lib.py cannot bind' % (name, ret))
        fn.argtypes, fn.restype = argtypes, _C_RESTYPES[ret]
        bound[name] = (ret, types)
    return bound


def source_hash():
    """sha256 over the kernel sources this tree builds libpcnn.so from (csrc/*.hip, csrc/*.h, include/pcnn.h): the stamp that ties a
    committed counter summary (profiles/*pmc_summary*.json) to the kernels it was measured on - bench.py reports `traffic` only when
    the stamp matches the tree it runs from."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, 'csrc', '*.hip')) + glob.glob(os.path.join(_HERE, 'csrc', '*.h')))
    files.append(os.path.join(os.path.dirname(_HERE), 'include', 'pcnn.h'))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load():
    """Loads libpcnn.so (built by `make -C poisson_cnn_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('libpcnn.so not found at %s: build it with `make -C poisson_cnn_amd/csrc` '
                           '(there is no CPU fallback)' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib._pcnn_prototypes = _bind_prototypes(lib)                  # argtypes + restype of every declared entry point, from the header itself
    _lib = lib
    return lib


class Handle:
    """One libpcnn handle per (process, device, stream)."""

    def __init__(self, device=0, stream=0):
        self.lib = load()
        self._h = c_void_p()
        rc = self.lib.pcnn_create(int(device), c_void_p(stream), byref(self._h))
        if rc != 0:
            raise RuntimeError('pcnn_create(device=%d) failed with code %d' % (device, rc))
        self.device, self.stream_ptr = device, int(stream or 0)

    def set_stream(self, stream):
        self.check(self.lib.pcnn_set_stream(self._h, c_void_p(stream)), 'pcnn_set_stream')
        self.stream_ptr = int(stream or 0)

    def check(self, rc, what):
        if rc != 0:
            raise RuntimeError('%s failed: %s' % (what, self.lib.pcnn_last_error(self._h).decode()))

    def call(self, name, *args):
        fn = getattr(self.lib, name)
        self.check(fn(self._h, *args), name)

    def __del__(self):
        try:
            if self._h:
                self.lib.pcnn_destroy(self._h)
        except Exception:
            pass
