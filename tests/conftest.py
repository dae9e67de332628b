import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'reference: needs the read-only reference checkout (build container only)')


@pytest.fixture(scope='session')
def reference_pkg():
    from oracle import ref_shims
    if not ref_shims.reference_available():
        pytest.skip('reference checkout not present (GPU box / CI): covered by tests/golden fixtures')
    return ref_shims.install()
