import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


class _OracleWithMass:
    """The oracle package with `backward` returning the per-element L1 masses by default (tests/parity.py)."""

    def __init__(self, module):
        self._module = module

    def __getattr__(self, name):
        return getattr(self._module, name)

    def backward(self, *args, **kw):
        kw.setdefault('want_mass', True)
        return self._module.backward(*args, **kw)


@pytest.fixture(scope='session')
def oracle():
    """The CPU oracle (test infrastructure); built on demand with gcc."""
    import oracle as _oracle
    _oracle.build()
    return _OracleWithMass(_oracle)


@pytest.fixture(scope='session')
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from dirt_amd import _lib
    _lib.load()  # the HIP extension must be present on a GPU box: fail loudly, never fall back
    return torch.device('cuda:0')
