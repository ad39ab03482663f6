import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the C-ABI library is built in-tree and git-ignored: build it on a fresh checkout (hipcc cross-compiles gfx950
    # without a GPU).  On the GPU box the prebuilt .so travels with the snapshot.
    so = os.path.join(ROOT, 'graspnerf_amd', 'csrc', 'libgnr.so')
    if not os.path.exists(so):
        import shutil
        import subprocess
        if shutil.which('hipcc'):
            subprocess.check_call(['bash', os.path.join(ROOT, 'graspnerf_amd', 'csrc', 'build.sh')])


@pytest.fixture(scope='session')
def weights_np():
    return dict(np.load(os.path.join(GOLDEN, 'weights_seed0.npz')))


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN, f'golden_{name}.npz')))
    return load
