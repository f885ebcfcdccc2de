import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """Device context of the HIP library; GPU tests fail (not skip) if it cannot be created."""
    from string_grouper_amd import _native as N
    return N.default_context(0)
