import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "oracle"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure the native pieces exist (they are built in-tree by __graft_entry__.build())."""
    from treeqp_amd import build, capi
    if not capi.library_path().exists() or not build.oracle_library().exists():
        build.build_product()
        build.build_oracle()
    return True


@pytest.fixture(scope="session")
def orc(built):
    import oracle_py
    return oracle_py


@pytest.fixture(scope="session")
def capi(built):
    from treeqp_amd import capi as c
    c.lib()
    return c
