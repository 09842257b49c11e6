import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle():
    """The CPU oracle (test infrastructure); built on demand with gcc."""
    import oracle as _oracle
    _oracle.build()
    return _oracle


@pytest.fixture(scope='session')
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from dirt_amd import _lib
    _lib.load()  # the HIP extension must be present on a GPU box: fail loudly, never fall back
    return torch.device('cuda:0')
