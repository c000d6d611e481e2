import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `-m gpu` on the GPU box)')


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build the HIP library (hipcc cross-compiles gfx950
    without a GPU) and the C oracle once, so that the suites do not depend on `__graft_entry__.build()` having run first."""
    import subprocess
    lib = os.path.join(ROOT, 'symmer_amd', 'libsymgpu.so')
    if not os.path.exists(lib):
        subprocess.run(['make', '-s', '-j4', '-C', os.path.join(ROOT, 'symmer_amd', 'csrc')], check=False)
    if not os.path.exists(os.path.join(ROOT, 'oracle', 'liboracle.so')):
        subprocess.run(['make', '-s', '-C', os.path.join(ROOT, 'oracle'), 'liboracle.so'], check=False)
