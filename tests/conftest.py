import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "spicedb-kubeapi-proxy_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` via gpurun)")


@pytest.fixture(scope="session")
def aclgpu_lib():
    """Builds (hipcc cross-compile, no GPU needed) and loads libaclgpu.so."""
    import aclgpu
    aclgpu._lib.build()
    return aclgpu._lib.load()
