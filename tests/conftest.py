import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "iridium-sniffer_amd"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import orc
    return orc.lib()


@pytest.fixture(scope="session")
def reflib():
    import orc
    r = orc.ref()
    if r is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    return r
