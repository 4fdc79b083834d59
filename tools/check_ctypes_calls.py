"""Static check of every libpcnn call site in poisson_cnn_amd/ against the prototypes of include/pcnn.h: argument count, and - where the
call site wraps an argument in a ctypes scalar - that the wrapper's kind (integer / float / pointer) is the declared one.
tests/test_host_logic.py runs it (no GPU needed)."""
import ast
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

INT_WRAPPERS = {'c_int', 'c_int32', 'c_int64', 'c_size_t', 'c_uint32', 'c_uint', 'c_longlong', 'c_uint64'}
FLOAT_WRAPPERS = {'c_float', 'c_double'}
PTR_WRAPPERS = {'c_void_p', 'byref', 'c_char_p', 'POINTER'}


def kind_of_decl(t):
    if t.endswith('*') or t == 'pcnn_handle':
        return 'ptr'
    return 'float' if t.replace('const ', '') in ('float', 'double') else 'int'


def kind_of_arg(node):
    if isinstance(node, ast.Call):
        f = node.func
        name = f.attr if isinstance(f, ast.Attribute) else (f.id if isinstance(f, ast.Name) else None)
        if name in INT_WRAPPERS:
            return 'int'
        if name in FLOAT_WRAPPERS:
            return 'float'
        if name in PTR_WRAPPERS or name in ('_p', 'ptr', '_ptr'):
            return 'ptr'
    if isinstance(node, ast.Constant):
        if node.value is None:
            return 'ptr'
        if isinstance(node.value, float):
            return 'float'
    return None


def check():
    from poisson_cnn_amd import _lib
    protos = _lib.header_prototypes()
    problems, sites = [], 0
    for path in sorted(glob.glob(os.path.join(ROOT, 'poisson_cnn_amd', '**', '*.py'), recursive=True)):
        tree = ast.parse(open(path).read(), path)
        for node in ast.walk(tree):
            if not isinstance(node, ast.Call):
                continue
            f = node.func
            name, args, via_call = None, None, False
            if isinstance(f, ast.Attribute) and f.attr == 'call' and node.args and isinstance(node.args[0], ast.Constant) and str(node.args[0].value).startswith('pcnn_'):
                name, args, via_call = node.args[0].value, node.args[1:], True
            elif isinstance(f, ast.Attribute) and f.attr.startswith('pcnn_'):
                name, args = f.attr, node.args
            if name is None:
                continue
            sites += 1
            where = '%s:%d %s' % (os.path.relpath(path, ROOT), node.lineno, name)
            if name not in protos:
                problems.append(where + ': not declared in include/pcnn.h')
                continue
            decl = protos[name][1][1:] if via_call else protos[name][1]            # Handle.call supplies the handle
            if any(isinstance(a, ast.Starred) for a in args):
                continue
            if len(args) != len(decl):
                problems.append('%s: %d arguments, the header declares %d' % (where, len(args), len(decl)))
                continue
            for i, (a, t) in enumerate(zip(args, decl)):
                k = kind_of_arg(a)
                if k is not None and k != kind_of_decl(t) and not (k == 'int' and kind_of_decl(t) == 'float'):
                    problems.append('%s: argument %d is spelled as %s, the header declares %s' % (where, i + (1 if via_call else 0), k, t))
    return sites, problems


if __name__ == '__main__':
    n, probs = check()
    print('%d call sites checked' % n)
    for p in probs:
        print('  ' + p)
    sys.exit(1 if probs else 0)
