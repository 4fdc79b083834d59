"""Generates tests/golden/fd_coefficients.json by loading the ONE reference module that is importable in the
build container without TensorFlow (dataset/utils/get_fd_coefficients.py: numpy + scipy only) straight from
/root/reference.  Run in the build container only; the GPU box has no /root/reference and only reads the JSON."""
import importlib.util
import json
import os

REF = '/root/reference/poisson_CNN/dataset/utils/get_fd_coefficients.py'
spec = importlib.util.spec_from_file_location('ref_get_fd_coefficients', REF)
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

cases = []
for pos, order in [([-1, 0, 1], 2), ([-2, -1, 0, 1, 2], 2), ([-1, 0, 1], 1), ([-2, -1, 0, 1, 2], 1), ([-3, -2, -1, 0, 1, 2, 3], 2),
                   ([-3, -2, -1, 0, 1], 2), ([0, 1, 2, 3], 1), ([-2, -1, 0, 1, 2], 4), ([-4, -3, -2, -1, 0, 1, 2, 3, 4], 2)]:
    cases.append({'stencil_positions': pos, 'order': order, 'coefficients': [float(v) for v in mod.get_fd_coefficients(pos, order)]})
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fd_coefficients.json'), 'w') as f:
    json.dump(cases, f, indent=1)
print(len(cases), 'cases written')
