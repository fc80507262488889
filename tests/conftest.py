import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_golden.npz'))


@pytest.fixture(scope='session')
def body():
    from avatarcap_amd import synthetic as syn
    return syn.synthetic_body()
