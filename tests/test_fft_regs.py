"""The in-register FFTs of the spectral transform kernels (poisson_cnn_amd/csrc/fft_regs.h) on the host: the header is plain C++ behind `__device__`
guards, so the very source the HIP kernels include is compiled with g++ and checked against a double-precision DFT (tests/native/test_fft_regs.cpp:
complex DIF of 2 ... 64 points both signs and its undo, the real-input forward and the half-complex inverse).  Tolerance: rel-L2 5e-7 (fp32, six
butterfly levels; measured 1.7e-7).  -ffp-contract=off as in the kernels' build."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('g++') is None, reason='needs g++')
def test_fft_regs_against_double_dft(tmp_path):
    exe = tmp_path / 'test_fft_regs'
    src = os.path.join(ROOT, 'tests', 'native', 'test_fft_regs.cpp')
    subprocess.run(['g++', '-O2', '-std=c++17', '-ffp-contract=off', '-o', str(exe), src], check=True, cwd=ROOT)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r'worst rel-L2 error ([0-9.e+-]+)', r.stdout)
    assert m and float(m.group(1)) < 5e-7, r.stdout


@pytest.mark.skipif(shutil.which('g++') is None, reason='needs g++')
def test_fft_regs_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """The same source under AddressSanitizer + UndefinedBehaviorSanitizer on the host (GPU sanitizers are not available on the pool): the template recursion
    indexes its register arrays with compile-time arithmetic - an off-by-one there would be silent on the device.  Any report aborts the run."""
    exe = tmp_path / 'test_fft_regs_san'
    src = os.path.join(ROOT, 'tests', 'native', 'test_fft_regs.cpp')
    r = subprocess.run(['g++', '-O1', '-std=c++17', '-ffp-contract=off', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-o', str(exe), src],
                       capture_output=True, text=True, cwd=ROOT)
    if r.returncode != 0 and ('asan' in r.stderr.lower() or 'ubsan' in r.stderr.lower() or 'sanitize' in r.stderr.lower()):
        pytest.skip('this toolchain has no sanitizer runtime: ' + r.stderr.strip().splitlines()[-1])
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900, env=dict(os.environ, ASAN_OPTIONS='detect_leaks=0'))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert 'worst rel-L2 error' in r.stdout

