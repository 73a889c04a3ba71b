import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle():
    """The CPU oracle (oracle/libmzoracle.so), built on demand.  Test infrastructure only."""
    sys.path.insert(0, os.path.join(REPO, 'oracle'))
    import oracle as _oracle

    _oracle.build()
    return _oracle
