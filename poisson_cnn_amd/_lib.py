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
    lib.pcnn_last_error.restype = c_char_p
    lib.pcnn_last_error.argtypes = [c_void_p]
    lib.pcnn_create.argtypes = [c_int, c_void_p, POINTER(c_void_p)]
    for name in ('pcnn_conv2d_wgrad_workspace', 'pcnn_colsum_workspace', 'pcnn_deconv_wgrad_workspace', 'pcnn_channel_scale_workspace',
                 'pcnn_dbc_expand_bwd_workspace'):
        if hasattr(lib, name):
            getattr(lib, name).restype = c_size_t
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
